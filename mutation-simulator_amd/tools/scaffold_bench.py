#!/usr/bin/env python3
"""End-to-end CLI time on an assembly-like input: N scaffolds of L bases each (file in -> Fasta + VCF files out).

    python mutation-simulator_amd/tools/scaffold_bench.py [n_scaffolds] [scaffold_len]
"""
import contextlib
import io
import random
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "mutation-simulator_amd")]


def write_genome(path: Path, n: int, L: int, bpl: int = 60) -> int:
    rs = np.random.RandomState(7)
    lines = (L + bpl - 1) // bpl
    with open(path, "wb") as fh:
        for i0 in range(0, n, 512):
            m = min(512, n - i0)
            bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, size=(m, lines * bpl))]
            for k in range(m):
                body = np.full((lines, bpl + 1), 10, dtype=np.uint8)
                body[:, :bpl] = bases[k].reshape(lines, bpl)
                txt = body.tobytes()
                cut = L + L // bpl if L % bpl else L + L // bpl - 1      # no '\n' beyond the last base's line end
                fh.write(b">scaf%06d\n" % (i0 + k) + txt[:cut] + b"\n")
    return n * L


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
    from mutation_simulator_amd import __main__ as cli
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        total = write_genome(td / "asm.fa", n, L)
        for rep in range(2):
            random.seed(1)
            np.random.seed(1)
            t0 = time.perf_counter()
            with contextlib.redirect_stderr(io.StringIO()):
                cli.main(["-q", "-o", str(td / f"out{rep}"), str(td / "asm.fa"), "args", "-sn", "0.01", "-titv", "2.0"])
            dt = time.perf_counter() - t0
            sz = (td / f"out{rep}_ms.fa").stat().st_size
            print(f"run {rep}: {n} scaffolds x {L} b = {total/1e6:.0f} Mb in {dt:.2f} s = {total/dt/1e6:.1f} Mbases/s end to end "
                  f"(Fasta {sz/1e6:.0f} MB, VCF {(td / f'out{rep}_ms.vcf').stat().st_size/1e6:.1f} MB)", flush=True)


if __name__ == "__main__":
    main()
