#!/usr/bin/env python3
"""k_rewrite micro-benchmark: one 240 Mb contig, several record loads, GB/s of algorithmic traffic.

    python mutation-simulator_amd/tools/apply_microbench.py [reps]
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "mutation-simulator_amd", ROOT / "tests", ROOT / "tests" / "golden"):
    sys.path.insert(0, str(p))

import bench  # noqa: E402
from mutation_simulator_amd import _ffi  # noqa: E402
from mutation_simulator_amd import mutator as mm  # noqa: E402

C3 = ["-in", "0.001", "-inmin", "1", "-inmax", "50", "-de", "0.001", "-demin", "1", "-demax", "50",
      "-du", "0.0005", "-dumin", "50", "-dumax", "500", "-iv", "0.0005", "-ivmin", "50", "-ivmax", "500"]


def run(name, L, sim, reps):
    eng = _ffi.Engine(0)
    eng.seed(42, 42)
    eng.set_params(mm.params_descriptor(sim))
    cid = eng.add_contig_synthetic(L, 5)
    eng.plan_contig(cid, mm.plan_descriptors(sim.chromosomes[0]) if sim.has_mutations else [])
    eng.apply_contig(cid)
    eng.reset_stats()
    for _ in range(reps):
        eng.apply_contig(cid)
    st = eng.stats()
    alg = st["bytes_in"] + st["bytes_out"] + 16 * st["records"]
    ms = st["apply_kernel_ms"] / reps
    print(f"{name:28s} L={L/1e6:.0f} Mb records={st['records']//reps:>9d} k_rewrite {ms*1e3:8.1f} us "
          f"{alg/reps/ms/1e6:8.1f} GB/s   all apply kernels {st['apply_ms']/reps*1e3:8.1f} us")
    eng.close()


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    L = 240_000_000
    class NoMut:
        has_mutations = False
        mut_block = {}
        titv = 1.0
        chromosomes = []
    run("no records (copy ceiling)", L, NoMut, reps)
    run("config 2: -sn 0.01", L, bench.workload_settings([L]), reps)
    run("-sn 0.001", L, bench.workload_settings([L], snp=0.001), reps)
    run("-sn 0.05", L, bench.workload_settings([L], snp=0.05), reps)
    run("config 3: SV mix", L, bench.workload_settings([L], snp=0.005, titv=1.0, extra=C3), reps)
