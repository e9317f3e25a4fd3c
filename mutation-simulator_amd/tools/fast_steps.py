"""Steps of one bench workload through the counter-based PLAN engine (--rng fast), for profiling:
    rocprofv3 --kernel-trace --stats -d out -- python3 mutation-simulator_amd/tools/fast_steps.py c3 5
Prints ms per step and the engine's stage times."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "mutation-simulator_amd", ROOT / "tests", ROOT / "tests" / "golden"):
    sys.path.insert(0, str(p))

import bench  # noqa: E402
from mutation_simulator_amd import _ffi  # noqa: E402
from mutation_simulator_amd import mutator as mm  # noqa: E402


def main():
    w = sys.argv[1] if len(sys.argv) > 1 else "c2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    total = int(float(sys.argv[3])) if len(sys.argv) > 3 else 3_000_000_000
    lengths = bench.contig_lengths(total)
    sim = bench.build_settings(w, lengths)
    _ffi.warm_up_async(0, pin=True).join()              # (as the CLI and bench.py: the calling thread on the GPU's NUMA node)
    eng = _ffi.Engine(0, _ffi.RNG_FAST)
    cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
    tables = [mm.plan_table(ch) for ch in sim.chromosomes]
    eng.set_params(mm.params_descriptor(sim))

    def step(apply=True):
        eng.set_fast_key(42)
        for ch, t in zip(sim.chromosomes, tables):
            eng.plan_contig(cids[ch.number], t)
            if apply:
                eng.apply_contig(cids[ch.number])
        eng.sync()
    for _ in range(3):
        step()
    for label, ap in (("plan+apply", True), ("plan only", False)):
        eng.reset_stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(ap)
        dt = time.perf_counter() - t0
        st = eng.stats()
        print(f"{w} {label}: {dt / steps * 1e3:.3f} ms per step, {sum(lengths) * steps / dt / 1e9:.1f} Gbases/s, "
              f"apply kernels {st['apply_ms'] / steps:.3f} ms, rewrite {st['apply_kernel_ms'] / steps:.3f} ms, records {st['records'] // max(steps, 1)}")
    # host enqueue cost alone: time the calls without the final sync
    eng.set_fast_key(42)
    t0 = time.perf_counter()
    for ch, t in zip(sim.chromosomes, tables):
        eng.plan_contig(cids[ch.number], t)
        eng.apply_contig(cids[ch.number])
    t1 = time.perf_counter()
    eng.sync()
    print(f"{w} host enqueue of one step: {(t1 - t0) * 1e3:.3f} ms, then {(time.perf_counter() - t1) * 1e3:.3f} ms until the device is done")
    eng.close()


if __name__ == "__main__":
    main()
