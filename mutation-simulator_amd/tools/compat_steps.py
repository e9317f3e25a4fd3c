"""Steps of one bench workload through the compatible (bit-exact) engines, for profiling and A/B runs:
    rocprofv3 --kernel-trace --stats -d out -- python3 mutation-simulator_amd/tools/compat_steps.py c2 5
Prints ms per step (bench.one_step: plan + apply every contig, one synchronisation) and what the host spends enqueuing one."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "mutation-simulator_amd", ROOT / "tests", ROOT / "tests" / "golden"):
    sys.path.insert(0, str(p))

import bench  # noqa: E402
from mutation_simulator_amd import _ffi  # noqa: E402
from mutation_simulator_amd import mutator as mm  # noqa: E402
from mutation_simulator_amd.sharding import run_sharded_pass  # noqa: E402


def main():
    w = sys.argv[1] if len(sys.argv) > 1 else "c2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    total = int(float(sys.argv[3])) if len(sys.argv) > 3 else 3_000_000_000
    lengths = bench.contig_lengths(total)
    sim = bench.build_settings(w, lengths)
    _ffi.warm_up_async(0, pin=True).join()              # (as the CLI and bench.py: the calling thread on the GPU's NUMA node)
    eng = _ffi.Engine(0)
    cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
    eng.set_params(mm.params_descriptor(sim))
    mine = list(range(len(lengths)))
    if len(sys.argv) > 4:                               # only a sharded rank's steps: it owns every argv[4]-th contig (0: none)
        stride = int(sys.argv[4])
        owned = mine[::stride] if stride else []
        for _ in range(3):
            bench.one_step(eng, sim, cids, owned, 42, mm.plan_table, lengths)
        t0 = time.perf_counter()
        for _ in range(steps):
            bench.one_step(eng, sim, cids, owned, 42, mm.plan_table, lengths)
        print(f"{w} a rank that owns {len(owned)} of {len(mine)} contigs: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")
        eng.close()
        return
    for _ in range(3):
        bench.one_step(eng, sim, cids, mine, 42, mm.plan_table)
    eng.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        bench.one_step(eng, sim, cids, mine, 42, mm.plan_table)
    dt = time.perf_counter() - t0
    st = eng.stats()
    print(f"{w}: {dt / steps * 1e3:.3f} ms per step, {sum(lengths) * steps / dt / 1e9:.1f} Gbases/s, plan_gpu {st['plan_gpu_ms'] / steps:.3f} ms, "
          f"rewrite {st['apply_kernel_ms'] / steps:.3f} ms in {st['apply_launches'] // steps} launches, ahead {st.get('snp_samples_ahead', 0) // steps} of {st['contigs_snp'] // steps}")
    for owned in ([], mine[::8], mine[::2]):             # a rank of a sharded step: the others' contigs through msim_plan_chain
        for _ in range(2):
            bench.one_step(eng, sim, cids, owned, 42, mm.plan_table, lengths)
        t0 = time.perf_counter()
        for _ in range(steps):
            bench.one_step(eng, sim, cids, owned, 42, mm.plan_table, lengths)
        print(f"{w} a rank that owns {len(owned)} of {len(mine)} contigs: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")
    eng.seed(42, 42)
    t0 = time.perf_counter()
    run_sharded_pass(eng, sim, cids, mine, mm.plan_table)
    t1 = time.perf_counter()
    eng.sync()
    print(f"{w} host enqueue of one step: {(t1 - t0) * 1e3:.3f} ms, then {(time.perf_counter() - t1) * 1e3:.3f} ms until the device is done")
    eng.close()


if __name__ == "__main__":
    main()
