#!/usr/bin/env python3
"""Soak run of tests/test_gpu_parity.py::test_random_soups_match_the_oracle over many more seeds than the suite holds:
python tools/soak_soups.py <first seed> <last seed>.  Prints one line per seed; exits non-zero on the first failure."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_gpu_parity as T
ctx = T.rt.Context(0)
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    t = time.time()
    try:
        T.test_random_soups_match_the_oracle(ctx, seed)
        print("seed %d ok (%.1f s)" % (seed, time.time() - t), flush=True)
    except Exception:
        bad += 1
        print("seed %d FAILED" % seed, flush=True)
        traceback.print_exc()
        if bad >= 3:
            break
sys.exit(1 if bad else 0)
