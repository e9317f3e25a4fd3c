#!/usr/bin/env python3
"""Attribute k_rewrite's time to its parts: run the config-3 micro-benchmark (one 240 Mb contig) against the
timing-only ablation builds of libmsim (`make -C mutation-simulator_amd/csrc ablate`; MSIM_ABL in apply.hip).

    python mutation-simulator_amd/tools/apply_ablation.py [reps]
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
LIB = ROOT / "mutation-simulator_amd" / "lib"
CHILD = r'''
import sys
sys.path[:0] = [%(root)r, %(root)r + "/mutation-simulator_amd", %(root)r + "/tests", %(root)r + "/tests/golden"]
import bench
from mutation_simulator_amd import _ffi, mutator as mm
L = 240_000_000
sim = bench.workload_settings([L], snp=0.005, titv=1.0, extra=bench.C3_FLAGS)
eng = _ffi.Engine(0)
eng.seed(42, 42)
eng.set_params(mm.params_descriptor(sim))
cid = eng.add_contig_synthetic(L, 5)
eng.plan_contig(cid, mm.plan_descriptors(sim.chromosomes[0]))
eng.apply_contig(cid)
eng.reset_stats()
reps = %(reps)d
for _ in range(reps):
    eng.apply_contig(cid)
st = eng.stats()
alg = st["bytes_in"] + st["bytes_out"] + 16 * st["records"]
ms = st["apply_kernel_ms"] / reps
print(f"k_rewrite {ms*1e3:8.1f} us  {alg/reps/ms/1e6:8.1f} GB/s")
'''

# (sub-directory of lib/, MSIM_REWRITE, label)
RUNS = [("ld0", "old", "k_rewrite, round-1 byte-shifted 20-B loads, 8 waves/SIMD"),
        ("ld0w6", "old", "  same, compiled for 6 waves/SIMD"),
        ("ld0w4", "old", "  same, compiled for 4 waves/SIMD"),
        ("ld1w8", "old", "k_rewrite, two aligned 16-B loads per group, 8 waves/SIMD"),
        ("ld1w6", "old", "  same, 6 waves/SIMD"),
        ("ld1w5", "old", "  same, 5 waves/SIMD"),
        ("ld1w4", "old", "  same, 4 waves/SIMD"),
        ("ld1w4", "span", "k_rewrite_span (input span staged in LDS), 4 workgroups/CU")]

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    for sub, mode, what in RUNS:
        lib = LIB / sub / "libmsim.so"
        if not lib.exists():
            continue
        env = dict(os.environ, MSIM_LIB=str(lib), MSIM_REWRITE=mode)
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": str(ROOT), "reps": reps}], env=env,
                           capture_output=True, text=True)
        print(f"{what:52s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
