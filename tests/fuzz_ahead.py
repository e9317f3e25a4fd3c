#!/usr/bin/env python3
"""Opt-in differential fuzz of the SNP sampler's anchored windows (tests/test_gpu_ahead.py has the fixed cases) against the
sequential host planner.  Not collected by pytest; run it by hand on a GPU box:

    python tests/fuzz_ahead.py [iterations] [seed] [big]

("big": contigs of 2-8 Mb throughout -- most samples are then larger than twice the uncertainty of their start and go ahead.)

Every iteration: a genome of 2-40 contigs (0.3-8 Mb; SNP rates up to k/n = 0.24; some contigs with two ranges or none, which stay
on the chain), random titv / sampling distance / emission group size, a random subset walked with msim_plan_chain, every contig
planned before the first read; records, insert pools and both streams' final positions must equal the host planner's."""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "mutation-simulator_amd", ROOT / "tests", ROOT / "tests" / "golden"):
    sys.path.insert(0, str(p))

from mutation_simulator_amd import _ffi  # noqa: E402
from test_gpu_ahead import _run_then_fetch, _same  # noqa: E402
from test_gpu_sampler import _params, _snp_range  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    always_big = len(sys.argv) > 3 and sys.argv[3] == "big"
    rs = np.random.RandomState(seed)
    ahead_total = 0
    worst = 0
    for it in range(iters):
        d = int(rs.choice([1, 1, 2, 3]))
        titv = float(rs.choice([0.0, 0.5, 1.0, 2.0, 7.5, 1e9]))
        n_contigs = int(rs.choice([2, 3, 5, 9, 24, 40]))
        big = always_big or rs.rand() < 0.3
        contigs = []
        for _ in range(n_contigs):
            L = int(rs.randint(2_000_000 if always_big else 300_000, 8_000_000 if big else 2_500_000))
            u = rs.rand()
            if u < 0.08:
                contigs.append((L, []))
                continue
            rate = float(rs.choice([0.004, 0.01, 0.03, 0.1, 0.19]))
            if u < 0.2:                                          # two ranges: the chain's own path, inside the estimate
                a = L // 3
                k1, k2 = max(4096, int(a * rate / d)), max(4096, int((L - a - 10) * rate / d))
                if a - k1 * d > 4 * k1 and (L - a - 10) - k2 * d > 4 * k2:
                    contigs.append((L, [_snp_range(0, a - 1, k1), _snp_range(a + 5, L - 1, k2, True)]))
                    continue
            k = max(4096, int(L * rate / d))
            if L - k * d <= 4 * k:
                k = max(4096, (L // (5 * d)))
            if L - k * d <= 4 * k:
                contigs.append((L, []))
                continue
            contigs.append((L, [_snp_range(0, L - 1, k)]))
        blocks = {t: d for t in ("SN", "IN", "DE", "IV", "DU", "TL", "TLI")}
        params = _params(blocks, titv=titv)
        os.environ["MSIM_AHEAD"] = "2"
        os.environ["MSIM_EMIT_GROUP"] = str(int(rs.choice([1, 2, 2, 3, 4])))
        chain_only = [i for i in range(n_contigs) if rs.rand() < (0.5 if rs.rand() < 0.5 else 0.0)]
        seeds = (int(rs.randint(1, 1 << 30)), int(rs.randint(1, 1 << 30)))
        what = f"it {it} d={d} titv={titv} contigs={[(L, [int(r.k) for r in rr]) for L, rr in contigs]} chain_only={chain_only} seeds={seeds} group={os.environ['MSIM_EMIT_GROUP']}"
        try:
            host, hs, hst = _run_then_fetch(_ffi.PLAN_HOST, contigs, params, seeds)
            gpu, gs, gst = _run_then_fetch(_ffi.PLAN_AUTO, contigs, params, seeds, chain_only=chain_only)
            _same(host, gpu, hs, gs, hst, gst)
        except BaseException:
            print("FAILED", what)
            raise
        ahead_total += gst["snp_samples_ahead"]
        worst = max(worst, gst.get("snp_ahead_margin_permille", 0))
        print(f"it {it} ok  {n_contigs} contigs, {gst['snp_samples_ahead']} samples ahead, {len(chain_only)} walked", flush=True)
    print(f"fuzz: {iters} iterations identical to the host planner, {ahead_total} samples ahead of the chain; the worst start used "
          f"{worst / 10:.1f} % of the deviation the host allowed for (8 sigma + 256 words; 100 % = the soft edge = an error)")


if __name__ == "__main__":
    main()
