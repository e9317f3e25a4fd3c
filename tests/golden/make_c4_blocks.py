#!/usr/bin/env python3
"""BASELINE configs[3] workload from the reference's own example: the block list of
``/root/reference/data/Example RMT files/Homo_sapiens.rmt`` made usable.

As shipped the example's ranges overlap, and the reference crashes on it (ValueError from random.sample,
SURVEY.md section 2).  This script -- build container only, the reference never travels -- does what the survey
prescribes: per chromosome 1-22, X, Y it interval-merges the ``a-b None`` gene blocks, rescales them from GRCh38
coordinates to the bench's synthetic contig lengths (bench.contig_lengths(3e9), same chromosome order), merges
again what the rescaling made touch, and adds hot (``sn 0.05``), cold (``sn 0.001``) and 1 kb pool-path hot-spot
(``sn 0.2``) ranges inside gaps, chosen by a fixed-seed generator.  The result is a DATA fixture
(tests/golden/c4_blocks.npz: per contig starts / ends (1-based inclusive) / kinds) that bench.py turns back into RMT
text; nothing of the reference's file text is kept.

    python tests/golden/make_c4_blocks.py
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path[:0] = [str(ROOT)]
import bench  # noqa: E402

SRC = Path("/root/reference/data/Example RMT files/Homo_sapiens.rmt")
KIND_NONE, KIND_HOT, KIND_COLD, KIND_SPOT = 0, 1, 2, 3


def merge(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1] + 2:            # overlapping, touching or leaving a gap too small for a filler range
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def main():
    blocks = {}
    cur = None
    for line in SRC.read_text().splitlines():
        t = line.split("#")[0].split()
        if not t:
            continue
        if t[0].lower() == "chr":
            cur = int(t[1])
            blocks[cur] = []
        elif cur is not None and "-" in t[0] and t[1].lower() == "none":
            a, b = t[0].split("-")
            blocks[cur].append((int(a), int(b)))
    lengths = bench.contig_lengths(3_000_000_000)
    rs = np.random.RandomState(2024)
    out = {"lengths": np.array(lengths, dtype=np.int64)}
    n_tot = 0
    for ci, L in enumerate(lengths):
        scale = L / bench.GRCH38[ci]
        iv = [(max(2, int(a * scale)), min(L - 2, int(b * scale))) for a, b in merge(blocks[ci + 1])]
        iv = [list(x) for x in merge([x for x in iv if x[1] >= x[0]])]
        rows = [(a, b, KIND_NONE) for a, b in iv]
        # hot / cold / hot-spot ranges inside gaps (the gap keeps >= 2 free bases on both sides)
        for (a0, b0), (a1, _) in zip(iv[:-1], iv[1:]):
            gap_lo, gap_hi = b0 + 3, a1 - 3
            glen = gap_hi - gap_lo + 1
            u = rs.random_sample()
            if u < 0.015 and glen > 3_000:
                n = int(min(glen - 10, rs.randint(1_000, 100_000)))
                s = gap_lo + int(rs.randint(0, glen - n))
                rows.append((s, s + n - 1, KIND_HOT))
            elif u < 0.03 and glen > 120_000:
                n = int(min(glen - 10, rs.randint(100_000, 1_000_000)))
                s = gap_lo + int(rs.randint(0, glen - n))
                rows.append((s, s + n - 1, KIND_COLD))
            elif u < 0.04 and glen > 1_500:
                s = gap_lo + int(rs.randint(0, glen - 1_000))
                rows.append((s, s + 999, KIND_SPOT))
        rows.sort()
        arr = np.array(rows, dtype=np.int64)
        assert np.all(arr[1:, 0] > arr[:-1, 1] + 1) and arr[0, 0] >= 2 and arr[-1, 1] <= L - 2
        out[f"s{ci}"] = arr[:, 0].astype(np.uint32)
        out[f"e{ci}"] = arr[:, 1].astype(np.uint32)
        out[f"k{ci}"] = arr[:, 2].astype(np.uint8)
        n_tot += len(rows)
        blocked = int((arr[arr[:, 2] == 0, 1] - arr[arr[:, 2] == 0, 0] + 1).sum())
        print(f"chr{ci+1}: {len(blocks[ci+1])} listed -> {len(iv)} merged blocks, {blocked / L:.1%} blocked, {len(rows) - len(iv)} hot/cold/spot")
    np.savez_compressed(HERE / "c4_blocks.npz", **out)
    print(f"{n_tot} ranges -> {HERE / 'c4_blocks.npz'} ({(HERE / 'c4_blocks.npz').stat().st_size} bytes)")


if __name__ == "__main__":
    main()
