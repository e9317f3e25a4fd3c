"""Contig sharding for multi-GPU runs (one process per GPU).

Contigs are independent units of APPLY, so they are spread over ranks by longest-processing-time
bin packing on their lengths.  PLAN is *not* sharded in compatible mode: the two
MT19937 streams are chained across contigs, so every rank walks the whole chain (deterministic;
for contigs it does not own only the stream positions, ``msim_plan_chain``) and plans with
emission + applies only the contigs it owns.  No data-path collective is needed
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


def run_sharded_pass(engine, sim, contig_ids, owned, plan_descriptors, apply: bool = True, lengths=None) -> None:
    """One PLAN + APPLY pass of a rank over contigs resident in HBM: every contig in index order (the streams chain
    across contigs) -- the owned ones planned with emission and applied, the others only walked (``msim_plan_chain``:
    stream positions, no records / SNP outcomes / insert pool) when their ``lengths`` are given, else planned in full.
    ``engine`` is an ``_ffi.Engine``; ``contig_ids[i]`` the libmsim id of contig i."""
    mine = set(owned)
    for chrom in sim.chromosomes:
        i = chrom.number
        if i in mine or lengths is None:
            engine.plan_contig(contig_ids[i], plan_descriptors(chrom))
            if apply and i in mine:
                engine.apply_contig(contig_ids[i])
        else:
            engine.plan_chain(lengths[i], plan_descriptors(chrom))
