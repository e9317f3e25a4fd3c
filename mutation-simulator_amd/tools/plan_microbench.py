#!/usr/bin/env python3
"""GPU sampler micro-benchmark: plan ONE contig repeatedly (no other contigs in flight), for
per-kernel timings under rocprofv3:   rocprofv3 --kernel-trace --stats -- python3 .../plan_microbench.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "mutation-simulator_amd", ROOT / "tests", ROOT / "tests" / "golden"):
    sys.path.insert(0, str(p))

import bench  # noqa: E402
from mutation_simulator_amd import _ffi  # noqa: E402
from mutation_simulator_amd import mutator as mm  # noqa: E402

if __name__ == "__main__":
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 240_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    sim = bench.workload_settings([L])
    eng = _ffi.Engine(0, _ffi.PLAN_GPU)
    eng.set_params(mm.params_descriptor(sim))
    cid = eng.add_contig_synthetic(L, 5)
    desc = mm.plan_descriptors(sim.chromosomes[0])
    for r in range(reps):
        eng.seed(42, 42)
        t0 = time.perf_counter()
        eng.plan_contig(cid, desc)
        eng.sync()
        t1 = time.perf_counter()
        eng.apply_contig(cid)
        eng.sync()
        print(f"rep {r}: plan {1e3*(t1-t0):.3f} ms, apply {1e3*(time.perf_counter()-t1):.3f} ms")
    eng.close()
