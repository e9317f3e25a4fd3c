#!/usr/bin/env python3
"""Opt-in differential fuzz of the whole product path (CLI -> device engines -> APPLY -> device text) against
the CPU oracle, with settings shaped to reach all three device PLAN engines in AUTO mode.  Not collected by
pytest (GPU minutes are budgeted); run it by hand on a GPU box:

    python tests/fuzz_engines.py [iterations] [seed] [round3]

Every iteration builds a small genome (1-4 contigs of 0.3-4 Mb), random ARGS or RMT settings whose SNP block
equals the minimum block (the device engines' precondition), runs the CLI and the oracle on the same seeds and
compares Fasta + VCF bytes.  Prints the failing configuration and exits non-zero on the first difference."""
from __future__ import annotations

import sys
import tempfile
import traceback
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "mutation-simulator_amd", ROOT / "tests", ROOT / "tests" / "golden"):
    sys.path.insert(0, str(p))

from test_gpu_parity import _product_vs_oracle  # noqa: E402


def args_settings(rs):
    dmin = int(rs.choice([1, 1, 1, 2, 3]))
    kind = rs.choice(["snp", "sv", "sv", "sv_sparse"])
    argv = ["args", "-titv", repr(float(rs.choice([0.0, 0.5, 1.0, 2.0, 7.5])))]
    if kind == "snp":
        argv += ["-sn", repr(float(rs.choice([0.004, 0.01, 0.03, 0.1])))]
    else:
        scale = 0.002 if kind == "sv_sparse" else float(rs.choice([0.004, 0.01, 0.03]))
        argv += ["-sn", repr(scale * float(rs.uniform(0.2, 1.0)))]
        for flag in ("in", "de", "du", "iv"):
            if rs.rand() < 0.25:
                continue
            lo = int(rs.randint(2 if flag == "iv" else 1, 40))
            hi = lo + int(rs.choice([0, 1, 10, 80, 600, 6000]))
            argv += [f"-{flag}", repr(scale * float(rs.uniform(0.02, 0.4))), f"-{flag}min", str(lo), f"-{flag}max", str(hi)]
    for flag in ("sn", "in", "de", "du", "iv", "tl"):
        b = dmin if flag == "sn" else dmin + int(rs.choice([0, 0, 1, 4, 30]))
        argv += [f"-{flag}b", str(b)]
    return argv, None


def rmt_settings(rs, lengths):
    rows = ["titv = " + repr(float(rs.choice([0.0, 1.0, 2.0]))), "", "std", "it None",
            "sn " + repr(float(rs.choice([0.005, 0.01, 0.02]))), ""]
    for ci, L in enumerate(lengths):
        if rs.rand() < 0.3:
            continue                                     # unlisted: std only (one big range -> SNP sampler)
        rows.append(f"chr {ci + 1}")
        n_blocks = int(rs.choice([3, 20, 150, 600]))
        cuts = np.sort(rs.choice(np.arange(2, L - 2), size=min(2 * n_blocks, (L - 4) // 2 * 2), replace=False))
        for a, b in zip(cuts[0::2], cuts[1::2]):
            what = rs.choice(["None", "None", "None", "sn 0.05", "sn 0.001", "sn 0.2", "sn 0.3"])
            rows.append(f"{int(a)}-{int(b)} {what}")
    return [], "\n".join(rows) + "\n"


def _sv_tokens(rs, scale, with_tl, widths=None):
    """RMT / ARGS style (flag, rate, min, max) tuples of a random SV mix; widths: force these max - min + 1 per type."""
    out = []
    for flag in ("in", "de", "du", "iv") + (("tl",) if with_tl else ()):
        if rs.rand() < 0.2:
            continue
        lo = int(rs.randint(2 if flag == "iv" else 1, 40))
        w = int(widths[flag]) if widths else 1 + int(rs.choice([0, 1, 10, 80, 600]))
        out.append((flag, scale * float(rs.uniform(0.02, 0.4)), lo, lo + w - 1))
    return out


def args_settings_round3(rs):
    """ARGS settings of the shapes the round-3 engines took over: translocations on one big range (SV-mix engine with
    __link_tls), five SV types with five different length widths (wide accept tables), SNP block above the sampling
    distance (host-chain engine with every candidate on the chain)."""
    dmin = int(rs.choice([1, 1, 2]))
    scale = float(rs.choice([0.004, 0.01, 0.03]))
    widths = None
    if rs.rand() < 0.4:
        w = [int(x) for x in rs.choice([2, 3, 7, 20, 55, 130, 700, 3000], size=5, replace=False)]
        widths = dict(zip(("in", "de", "du", "iv", "tl"), w))
    argv = ["args", "-titv", repr(float(rs.choice([0.0, 1.0, 2.0]))), "-sn", repr(scale * float(rs.uniform(0.2, 1.0)))]
    for flag, rate, lo, hi in _sv_tokens(rs, scale, with_tl=rs.rand() < 0.8, widths=widths):
        argv += [f"-{flag}", repr(rate), f"-{flag}min", str(lo), f"-{flag}max", str(hi)]
    sn_b = dmin + (int(rs.choice([1, 3])) if rs.rand() < 0.2 else 0)
    for flag in ("sn", "in", "de", "du", "iv", "tl"):
        b = sn_b if flag == "sn" else dmin + int(rs.choice([0, 0, 1, 4, 30]))
        argv += [f"-{flag}b", str(b)]
    return argv, None


def rmt_settings_round3(rs, lengths):
    """RMT files of the mainstream shape: an SV `std` line (sometimes with translocations), gene blocks, hot / cold ranges
    with SV settings and token order of their own, sometimes `sn_block` above the minimum block (host-chain engine)."""
    def line(tokens, sn):
        parts = [f"sn {sn!r}"] if sn else []
        for flag, rate, lo, hi in tokens:
            parts.append(f"{flag} {rate!r} {flag}min {lo} {flag}max {hi}")
        order = list(rs.permutation(len(parts)))
        return " ".join(parts[i] for i in order) or "None"
    meta = ["titv = " + repr(float(rs.choice([0.0, 1.0, 2.0])))]
    if rs.rand() < 0.25:
        meta.append(f"sn_block = {int(rs.choice([2, 3, 6]))}")
    if rs.rand() < 0.3:
        meta.append(f"de_block = {int(rs.choice([2, 5, 40]))}")
    with_tl = rs.rand() < 0.5
    std = line(_sv_tokens(rs, 0.01, with_tl), float(rs.choice([0.004, 0.008])))
    rows = meta + ["", "std", "it None", std, ""]
    for ci, L in enumerate(lengths):
        if rs.rand() < 0.25:
            continue
        if L < 8:                                          # (a contig of a few bases: no room for a range line; std settings)
            continue
        rows.append(f"chr {ci + 1}")
        n_blocks = int(rs.choice([3, 20, 150]))
        cuts = np.sort(rs.choice(np.arange(2, L - 2), size=min(2 * n_blocks, (L - 4) // 2 * 2), replace=False))
        for a, b in zip(cuts[0::2], cuts[1::2]):
            r = rs.rand()
            if r < 0.6:
                what = "None"
            elif r < 0.8:
                what = "sn " + repr(float(rs.choice([0.05, 0.001, 0.2])))
            else:
                what = line(_sv_tokens(rs, float(rs.choice([0.01, 0.03])), with_tl and rs.rand() < 0.5), float(rs.choice([0.0, 0.01, 0.02])))
            rows.append(f"{int(a)}-{int(b)} {what}")
    return [], "\n".join(rows) + "\n"


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    round3 = len(sys.argv) > 3 and sys.argv[3] == "round3"            # the shapes of args/rmt_settings_round3 instead
    rs = np.random.RandomState(seed)
    for it in range(iters):
        lengths = [int(rs.choice([300_000, 700_000, 1_500_000, 4_000_000]) + rs.randint(0, 5000))
                   for _ in range(int(rs.randint(1, 5)))]
        if rs.rand() < 0.15:
            lengths.append(int(rs.randint(1, 3000)))     # a tiny contig in the middle of the stream chain
        mode = rs.choice(["args", "args", "rmt"])
        if round3:
            argv, rmt = args_settings_round3(rs) if mode == "args" else rmt_settings_round3(rs, lengths)
        else:
            argv, rmt = args_settings(rs) if mode == "args" else rmt_settings(rs, lengths)
        spec = {"contigs": [{"defline": f"f{it}_{i} fuzz", "length": L, "bpl": int(rs.choice([50, 60, 61, 80])),
                             "seed": 10_000 * seed + 10 * it + i} for i, L in enumerate(lengths)]}
        sp, sn = int(rs.randint(0, 1 << 30)), int(rs.randint(0, 1 << 30))
        try:
            with tempfile.TemporaryDirectory() as td:
                _product_vs_oracle(Path(td), spec, argv, sp, sn, rmt_text=rmt)
        except (ValueError, KeyError) as e:              # over-dense settings: both sides raise the reference's error
            print(f"it {it:3d} {mode} lengths={lengths} -> {type(e).__name__} (reference error path) ", flush=True)
            continue
        except Exception:
            print(f"FAIL it {it} seed {seed}: lengths={lengths} argv={argv} seeds=({sp},{sn})\nrmt={rmt}", flush=True)
            traceback.print_exc()
            sys.exit(1)
        print(f"it {it:3d} ok  {mode:4s} lengths={lengths} {' '.join(argv[:12])}", flush=True)
    print(f"fuzz: {iters} iterations identical to the oracle")


if __name__ == "__main__":
    main()
