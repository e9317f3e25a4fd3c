#!/usr/bin/env python3
"""Host-side cost of enqueueing one config-2 step (plan + apply calls return without waiting): if this is close
to the step time, the step is bound by HIP API calls on the host, not by the GPU."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "mutation-simulator_amd", ROOT / "tests", ROOT / "tests" / "golden"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from mutation_simulator_amd import _ffi  # noqa: E402
from mutation_simulator_amd import mutator as mm  # noqa: E402

lengths = bench.contig_lengths(3_000_000_000)
sim = bench.workload_settings(lengths)
eng = _ffi.Engine(0)
eng.set_params(mm.params_descriptor(sim))
cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
descs = [mm.plan_descriptors(ch) for ch in sim.chromosomes]
for rep in range(4):
    eng.seed(42, 42)
    t0 = time.perf_counter()
    tp = ta = 0.0
    for i in range(len(lengths)):
        a = time.perf_counter()
        eng.plan_contig(cids[i], descs[i])
        b = time.perf_counter()
        eng.apply_contig(cids[i])
        c = time.perf_counter()
        tp += b - a
        ta += c - b
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.2f} ms (plan calls {1e3*tp:.2f}, apply calls {1e3*ta:.2f}); then sync waited {1e3*(t2-t1):.2f} ms")
