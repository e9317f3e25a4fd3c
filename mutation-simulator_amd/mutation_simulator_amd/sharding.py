"""Contig sharding for multi-GPU runs (one process per GPU).

Contigs are independent units of APPLY, so they are spread over ranks by longest-processing-time
bin packing on their lengths.  PLAN is *not* sharded in compatible mode: the two
MT19937 streams are chained across contigs, so every rank replays the whole PLAN (deterministic,
cheap on the GPU sampler) and applies only the contigs it owns.  No data-path collective is needed
for results that stay in HBM.
"""
from __future__ import annotations


def lpt_partition(lengths: list[int], n_ranks: int) -> list[list[int]]:
    """Owner lists per rank: contigs sorted by length descending, each to the least-loaded rank."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    bins: list[list[int]] = [[] for _ in range(n_ranks)]
    load = [0] * n_ranks
    for idx in sorted(range(len(lengths)), key=lambda i: (-lengths[i], i)):
        b = min(range(n_ranks), key=lambda r: (load[r], r))
        bins[b].append(idx)
        load[b] += lengths[idx]
    return [sorted(b) for b in bins]


def imbalance(lengths: list[int], parts: list[list[int]]) -> float:
    """max rank load / mean rank load."""
    loads = [sum(lengths[i] for i in p) for p in parts]
    mean = sum(loads) / len(loads) if loads else 0
    return max(loads) / mean if mean else 1.0


def run_sharded_pass(engine, sim, contig_ids, owned, plan_descriptors, apply: bool = True) -> None:
    """One PLAN + APPLY pass of a rank: plan every contig in index order (stream chaining), apply
    the owned ones.  ``engine`` is an ``_ffi.Engine``; ``contig_ids[i]`` the libmsim id of contig i."""
    mine = set(owned)
    for chrom in sim.chromosomes:
        i = chrom.number
        engine.plan_contig(contig_ids[i], plan_descriptors(chrom))
        if apply and i in mine:
            engine.apply_contig(contig_ids[i])
