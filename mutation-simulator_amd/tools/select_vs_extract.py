#!/usr/bin/env python3
"""A/B of the host-chain engine's sampling step on the REAL c4sv range table (see select_vs_extract.cpp): writes the (n, k) pairs
of the 3 Gb bench genome's drawing ranges, compiles the C++ benchmark for this host and runs it.

    python mutation-simulator_amd/tools/select_vs_extract.py [reps]
"""
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "mutation-simulator_amd", ROOT / "tests", ROOT / "tests" / "golden"):
    sys.path.insert(0, str(p))

import bench  # noqa: E402
from mutation_simulator_amd import mutator as mm  # noqa: E402

lengths = bench.contig_lengths(3_000_000_000)
sim = bench.build_settings("c4sv", lengths)
pairs = []
for ch in sim.chromosomes:
    t = mm.plan_table(ch)
    k = t["k"].astype(np.int64)
    n = (t["stop"].astype(np.int64) - (k - 1)) - t["start"].astype(np.int64)      # d = 1 (default blocks)
    keep = (k > 0) & (n > t["setsize"])                                            # set path (the pool-path ranges are tiny)
    pairs.append(np.stack([n[keep], k[keep]], axis=1))
nk = np.concatenate(pairs).astype(np.uint32)
td = Path(tempfile.mkdtemp())
(td / "ranges.bin").write_bytes(nk.tobytes())
exe = td / "sve"
subprocess.check_call(["g++", "-O3", "-march=native", "-o", str(exe), str(Path(__file__).with_suffix(".cpp"))])
print(subprocess.run(["grep", "-m1", "model name", "/proc/cpuinfo"], capture_output=True, text=True).stdout.strip())
sys.exit(subprocess.call([str(exe), str(td / "ranges.bin")] + sys.argv[1:2]))
