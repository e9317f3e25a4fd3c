"""How a process's NUMBER of hardware queues changes the latency of a kernel launch (MI355X, ROCm 7.2):
    python3 mutation-simulator_amd/tools/hw_queue_probe.py
Every line is a fresh process: a libmsim context (plan stream: high priority; emission stream: normal), then n more streams at a
priority, each used once; then 4000 dependent one-lane launches on the plan stream, timed.  The runtime hands out up to four
hardware queues per priority; the step from eight to nine queues in the process is what cost the SV-mix engine 5 ms per 3 Gb step
behind passes that had created side streams (NOTES.md section 10)."""
import ctypes as C
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "mutation-simulator_amd"):
    sys.path.insert(0, str(p))


def one(spec):
    from mutation_simulator_amd import _ffi
    eng = _ffi.Engine(0)
    eng.lib.msim_dbg_queue_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    eng.lib.msim_dbg_queue_probe.restype = C.c_int
    ns = C.c_double()
    total = 2                                              # the context's plan + emission stream
    for part in spec.split("+"):
        if not part:
            continue
        n, prio = part.split(":")
        assert eng.lib.msim_dbg_queue_probe(eng.h, int(n), {"h": -1, "n": 0, "l": 1}[prio], 8, 0, C.byref(ns)) == 0
        total += int(n)
    best = 1e18
    for _ in range(5):
        assert eng.lib.msim_dbg_queue_probe(eng.h, 0, 0, 4000, 0, C.byref(ns)) == 0
        best = min(best, ns.value)
    bursts = []
    for idle in (100, 1000, 5000):
        assert eng.lib.msim_dbg_queue_probe(eng.h, 0, 0, 200, idle, C.byref(ns)) == 0
        bursts.append(ns.value / 1e3)
    print(f"{spec or '(none)':22s} streams {total:3d}   {best / 1e3:6.2f} us per back-to-back launch | a burst of 4 launches + wait after "
          f"100 us / 1 ms / 5 ms of idle queue: {bursts[0]:6.1f} / {bursts[1]:6.1f} / {bursts[2]:6.1f} us", flush=True)
    eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        specs = ["", "1:n", "2:n", "3:n", "3:n+1:l", "3:n+2:l", "3:n+3:l", "3:n+4:l", "6:n", "6:n+3:l", "3:n+3:l+1:h", "3:n+3:l+3:h",
                 "3:h", "4:h", "8:n+8:l+8:h"]
        for spec in specs:
            subprocess.run([sys.executable, __file__, "--one", spec], check=False)
