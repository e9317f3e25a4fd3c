import sys
sys.path[:0]=["/root/repo","/root/repo/mutation-simulator_amd","/root/repo/tests","/root/repo/tests/golden"]
import bench
from mutation_simulator_amd import _ffi, mutator as mm
lengths = bench.contig_lengths(3_000_000_000)
sim = bench.workload_settings(lengths)
eng = _ffi.Engine(0)
eng.set_params(mm.params_descriptor(sim))
cids=[eng.add_contig_synthetic(L, 1000+i) for i,L in enumerate(lengths)]
descs=[mm.plan_descriptors(ch) for ch in sim.chromosomes]
import time
for rep in range(3):
    eng.seed(42,42)
    t=time.perf_counter()
    for i,ch in enumerate(sim.chromosomes):
        eng.plan_contig(cids[i], descs[i])
        eng.sync()          # isolate: nothing of the next contig overlaps
    print("plan-only, sync per contig:", round((time.perf_counter()-t)*1e3,2),"ms")
