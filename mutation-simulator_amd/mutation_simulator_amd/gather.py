"""Final gather of sharded results to rank 0 over RCCL (xGMI inside a node).

Contigs are applied on the rank that owns them (``sharding.lpt_partition``); a consumer that wants the
whole mutated genome on one GPU gathers them with point-to-point transfers: every peer sends its
contigs straight to rank 0 over its own direct link (a ring collective would be bound by one link;
sizes differ per rank anyway).  The transport is ``torch.distributed`` with backend ``nccl`` -- which
is RCCL on ROCm -- used purely as plumbing: libmsim's device buffers are wrapped zero-copy through
``__cuda_array_interface__``; nothing here touches the bytes.

This module is the only place the package imports torch, and only when a gather is requested.
"""
from __future__ import annotations


class _DeviceBytes:
    """Minimal ``__cuda_array_interface__`` view of ``n`` bytes at device address ``addr``."""

    def __init__(self, addr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (addr, False),
                                         "version": 2, "strides": None}


def as_tensor(addr: int, n: int, device):
    import torch
    if n == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.as_tensor(_DeviceBytes(addr, n), device=device)


def gather_to_root(engine, contig_ids, parts, lengths_out, rank: int, world: int, device, root: int = 0):
    """Move every applied contig's mutated stream to ``root``.

    ``parts[r]`` = contig indices owned by rank r, ``lengths_out[i]`` = mutated length of contig i
    (known on every rank: PLAN is replicated).  Returns, on root, a dict contig -> uint8 tensor on
    ``device`` (own contigs are views of libmsim's buffers, received ones are fresh tensors); on other
    ranks an empty dict.  All transfers are posted before any is waited for.
    """
    import torch
    import torch.distributed as dist
    out = {}
    if world == 1:
        for i in parts[0]:
            addr, n = engine.result_device_ptr(contig_ids[i])
            out[i] = as_tensor(addr, n, device)
        return out
    ops, recv = [], {}
    if rank == root:
        for r in range(world):
            for i in parts[r]:
                if r == root:
                    addr, n = engine.result_device_ptr(contig_ids[i])
                    out[i] = as_tensor(addr, n, device)
                else:
                    buf = torch.empty(lengths_out[i], dtype=torch.uint8, device=device)
                    recv[i] = buf
                    ops.append(dist.P2POp(dist.irecv, buf, r))
    else:
        for i in parts[rank]:
            addr, n = engine.result_device_ptr(contig_ids[i])
            ops.append(dist.P2POp(dist.isend, as_tensor(addr, n, device), root))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    out.update(recv)
    return out
