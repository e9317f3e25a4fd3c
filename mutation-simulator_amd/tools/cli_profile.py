#!/usr/bin/env python3
"""End-to-end profile of the CLI on a synthetic FASTA (file in -> Fasta + VCF files out).

    python mutation-simulator_amd/tools/cli_profile.py --mb 600 --contigs 4 -- args -sn 0.01 -titv 2.0

Prints the wall time of the whole run and the cProfile top list, i.e. where the host-side text / IO work
around the device path goes."""
import argparse
import cProfile
import pstats
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "mutation-simulator_amd")]


def write_fasta(path: Path, mb: int, contigs: int, bpl: int = 60) -> None:
    rs = np.random.RandomState(5)
    with open(path, "wb") as f:
        for c in range(contigs):
            L = mb * 1_000_000 // contigs
            bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, L)]
            f.write(f">chr{c+1} synthetic\n".encode())
            full = L // bpl
            block = np.empty((full, bpl + 1), dtype=np.uint8)
            block[:, :bpl] = bases[:full * bpl].reshape(full, bpl)
            block[:, bpl] = 10
            f.write(block.tobytes())
            if L % bpl:
                f.write(bases[full * bpl:].tobytes() + b"\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=600)
    ap.add_argument("--contigs", type=int, default=4)
    ap.add_argument("--top", type=int, default=22)
    ap.add_argument("--sort", default="cumulative", help="cProfile sort key (cumulative, tottime)")
    ap.add_argument("--dir", default=None, help="directory for the input and the outputs (default: the system's temp dir)")
    ap.add_argument("--runs", type=int, default=1, help="repeat the CLI run (outputs removed in between); the last one is profiled")
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    rest = [x for x in a.rest if x != "--"] or ["args", "-sn", "0.01", "-titv", "2.0"]
    from mutation_simulator_amd.__main__ import main as cli_main
    with tempfile.TemporaryDirectory(dir=a.dir) as td:
        fa = Path(td) / "in.fa"
        t0 = time.perf_counter()
        write_fasta(fa, a.mb, a.contigs)
        print(f"input: {fa.stat().st_size/1e6:.0f} MB written in {time.perf_counter()-t0:.1f} s", flush=True)
        argv = ["--seed", "42", "-q", "-o", str(Path(td) / "out"), str(fa)] + rest
        for _ in range(a.runs - 1):
            t0 = time.perf_counter()
            cli_main(argv)
            print(f"(unprofiled run: {time.perf_counter()-t0:.3f} s)", flush=True)
            for o in Path(td).glob("out*"):
                o.unlink()
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        print(f"(profiled run starts at t={time.monotonic():.4f})", flush=True)
        pr.enable()
        cli_main(argv)
        pr.disable()
        dt = time.perf_counter() - t0
        print(f"(profiled run ends at t={time.monotonic():.4f})", flush=True)
        outs = sorted(Path(td).glob("out*"))
        print(f"CLI wall {dt:.2f} s  ({a.mb/dt:.1f} Mbases/s end to end); outputs: "
              + ", ".join(f"{o.name} {o.stat().st_size/1e6:.0f} MB" for o in outs), flush=True)
        pstats.Stats(pr).sort_stats(a.sort).print_stats(a.top)


if __name__ == "__main__":
    main()
