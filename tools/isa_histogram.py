#!/usr/bin/env python3
"""Instruction histogram of k_closest_hit<false> per region (refill / node step / leaf step / loop control), from the
gfx950 assembly of the shipped source and flags (hipcc -S).  The regions are delimited by the `; RTMARK <name>` comments
the kernel emits through empty inline asm.

  python tools/isa_histogram.py [--out profiles/r03_isa_closest_hit]   ->  <out>.txt (table) and <out>.json

Per region: instructions by class, in total and on the straight-line path (`always`: not skippable by a forward
s_cbranch inside the region), and VALU pipe cycles: a wave64 VALU instruction occupies its SIMD-32 pipe for 2 cycles,
transcendental f32 and f64 arithmetic for 4, f64 transcendental 8 (MI355X_MICROARCH.md, cycle constants).
bench.py multiplies the `always` VALU cycles of a region by the number of times the counters build saw the waves enter it:
a LOWER bound of the VALU pipe cycles a launch issues, hence a roofline fraction that cannot exceed 1."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer-rust_amd", "csrc")
TRANS32 = ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")


def flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^CXXFLAGS\s*=\s*(.*)$", mk, re.M)
    return m.group(1).split()


def classify(op):
    if op.startswith("v_"):
        if op.startswith("v_mfma") or op.startswith("v_smfma"):
            return "mfma", 0
        if "f64" in op:
            return "valu_f64", 8 if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")) else 4
        if op.startswith(TRANS32):
            return "valu_trans", 4
        return "valu", 2
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 0
    if op.startswith("ds_"):
        return "lds", 0
    if op.startswith(("s_load", "s_buffer_load", "s_scratch")):
        return "smem", 0
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call")):
        return "branch", 0
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep", "s_setprio")):
        return "wait", 0
    if op.startswith("s_"):
        return "salu", 0
    return "other", 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_isa_closest_hit"))
    ap.add_argument("--kernel", default="k_closest_hitILb0ELb1ELb0E")   # <COUNT = false, TRIS = true, GEN = false>
    ap.add_argument("--keep-asm", default=None)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        asm = args.keep_asm or os.path.join(td, "rt_kernels.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + flags() + ["--cuda-device-only", "-S", "-o", asm,
                               os.path.join(CSRC, "rt_kernels.hip")], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN.*%s.*:\s*(;.*)?$" % args.kernel, l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    region = "prologue"
    pending = set()          # forward branch targets not yet reached: code in between can be skipped
    seen_labels = set()
    hist = {}
    meta = {}
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t:
            continue
        m = re.match(r"^;\s*RTMARK\s+(\w+)", t)
        if m:
            region = m.group(1)
            pending = set()      # a marker sits on the straight-line path of its region
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            seen_labels.add(m.group(1))
            pending.discard(m.group(1))
            continue
        if t.startswith((";", ".", "#")):
            continue
        op = t.split()[0]
        cls, cyc = classify(op)
        h = hist.setdefault(region, {})
        e = h.setdefault(cls, {"total": 0, "always": 0, "cycles_total": 0, "cycles_always": 0})
        e["total"] += 1
        e["cycles_total"] += cyc
        if not pending:
            e["always"] += 1
            e["cycles_always"] += cyc
        if cls == "branch":
            tgt = t.split()[-1]
            if tgt.startswith(".LBB") and tgt not in seen_labels:
                pending.add(tgt)
    for l in lines[end:end + 400]:
        for key in (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".group_segment_fixed_size"):
            if l.strip().startswith(key + ":") and key not in meta:
                meta[key] = int(l.split(":")[1])
        if ".name:" in l and meta:
            break
    out = {"kernel": "k_closest_hit<false, true, false>", "flags": " ".join(flags()), "regions": {}, "meta": meta}
    txt = ["k_closest_hit<false, true, false>, gfx950, flags: %s" % out["flags"], "registers: %s" % meta, "",
           "%-10s %-11s %7s %7s %9s %9s" % ("region", "class", "total", "always", "cyc_total", "cyc_alw")]
    for r, h in hist.items():
        reg = {c: e for c, e in sorted(h.items())}
        reg["valu_cycles_always"] = sum(e["cycles_always"] for c, e in h.items() if c.startswith("valu"))
        reg["valu_cycles_total"] = sum(e["cycles_total"] for c, e in h.items() if c.startswith("valu"))
        reg["valu_insts_always"] = sum(e["always"] for c, e in h.items() if c.startswith("valu"))
        reg["valu_insts_total"] = sum(e["total"] for c, e in h.items() if c.startswith("valu"))
        out["regions"][r] = reg
        for c, e in sorted(h.items()):
            txt.append("%-10s %-11s %7d %7d %9d %9d" % (r, c, e["total"], e["always"], e["cycles_total"], e["cycles_always"]))
        txt.append("%-10s %-11s %7d %7d %9d %9d" % (r, "VALU (all)", reg["valu_insts_total"], reg["valu_insts_always"], reg["valu_cycles_total"], reg["valu_cycles_always"]))
        txt.append("")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    open(args.out + ".txt", "w").write("\n".join(txt) + "\n")
    json.dump(out, open(args.out + ".json", "w"), indent=1)
    print("\n".join(txt))


if __name__ == "__main__":
    main()
