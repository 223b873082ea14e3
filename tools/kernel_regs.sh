#!/bin/bash
# tools/kernel_regs.sh [extra hipcc flags]: VGPRs, spills and occupancy of every kernel of rt_kernels.hip (compiler remarks, no GPU needed)
cd "$(dirname "$0")/../raytracer-rust_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fno-slp-vectorize -fPIC "$@" -Rpass-analysis=kernel-resource-usage -c -o /dev/null rt_kernels.hip 2>&1 |
  python3 -c '
import re,sys
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r"remark: .*Function Name: (\S+)",l)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r"remark: .*?\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPR Spill|SGPR Spill): (\d+)",l)
    if m and cur: rows[cur][m.group(1)]=m.group(2)
import subprocess
for k,v in rows.items():
    name=subprocess.run(["c++filt",k],capture_output=True,text=True).stdout.strip()
    name=name.replace("rtamd::(anonymous namespace)::","").replace("void ","")
    name=re.sub(r"\(rtamd::.*","",name)
    print("%-34s VGPR %3s  spill %3s  scratch %4s  occ %s  LDS %s"%(name[:34],v.get("VGPRs"),v.get("VGPR Spill","0"),v.get("ScratchSize [bytes/lane]"),v.get("Occupancy [waves/SIMD]"),v.get("LDS Size [bytes/block]")))
'
