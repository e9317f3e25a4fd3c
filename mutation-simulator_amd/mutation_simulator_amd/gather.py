"""Final gather of sharded results to rank 0: control plane in Python, data plane in libmsim.

Contigs are applied on the rank that owns them (``sharding.lpt_partition``); a consumer that wants the whole
mutated genome on one GPU gathers them with point-to-point transfers -- every peer sends its contigs straight to
the root over its own direct xGMI link (a ring collective would be bound by one link; sizes differ per rank).

* data plane: ``msim_gather_to_root`` (csrc/comm.cpp) -- libmsim's own RCCL communicator, grouped
  ``ncclSend`` / ``ncclRecv`` on its device buffers.  No torch tensor ever holds genome bytes.
* control plane: whatever process group the launcher provides (``torch.distributed`` with the CPU backend
  ``gloo`` under ``torch.distributed.run``): it carries the 128-byte ``ncclUniqueId`` once and, only for contigs
  whose mutated length PLAN does not fix (host-planned tables), their lengths.

``HostTransport`` executes the same transfer list (``msim_gather_plan``) over the control plane's CPU send / recv:
the stand-in used by the multi-process CPU tests, where no GPU exists.
"""
from __future__ import annotations

import numpy as np

from . import _ffi


def owners_of(parts, n_contigs: int):
    owner = [0] * n_contigs
    for r, p in enumerate(parts):
        for i in p:
            owner[i] = r
    return owner


class Communicator:
    """One per rank.  ``dist`` = an initialised ``torch.distributed`` module or None for world 1.  The control plane moves CPU
    objects and CPU tensors (the ``ncclUniqueId``, the sizes exchange): its process group needs a CPU-capable backend -- ``gloo``,
    what ``bench.py`` and ``multi_gpu.py`` initialise.  Under a device-only group (``nccl``) the sizes go by ``all_gather_object``
    instead of the packed ``all_reduce`` (slower, pickles; correct)."""

    def __init__(self, engine, rank: int, world: int, dist=None):
        self.eng, self.rank, self.world, self.dist = engine, rank, world, dist
        self.live = False
        if world > 1:
            # rank 0 ALWAYS takes part in the broadcast: if it cannot produce the id (no librccl, dlopen failure) it
            # broadcasts the reason instead, so that every rank raises the same error before any later collective --
            # leaving the broadcast on one rank only would hang the others until the control plane times out
            box = [None]
            if rank == 0:
                try:
                    box = [("id", _ffi.comm_unique_id())]
                except Exception as e:  # noqa: BLE001
                    box = [("error", f"{type(e).__name__}: {e}")]
            dist.broadcast_object_list(box, src=0)
            kind, payload = box[0]
            if kind == "error":
                raise _ffi.MsimError(f"rank 0 could not open an RCCL communicator: {payload}")
            engine.comm_init(payload, rank, world)
            self.live = True

    def describe(self) -> str:
        return ("libmsim RCCL communicator (dlopen librccl): grouped ncclSend/ncclRecv, every peer -> rank 0 over its own "
                "xGMI link; ncclUniqueId bootstrapped over the gloo control plane")

    def sizes(self, contig_ids, parts):
        """(owner, mutated length, records, insert pool bytes) of every contig, on every rank.  The owner knows them once it
        has applied the contig; the others learn them over the control plane (in compatible mode a rank only walks the stream
        chain through contigs it does not own -- msim_plan_chain leaves no table --, in fast mode it skips them altogether)."""
        owner = owners_of(parts, len(contig_ids))
        n = len(contig_ids)
        if self.world > 1:
            # one packed int64 tensor, summed over the ranks (every contig has exactly one owner): no pickling on the control
            # plane inside a timed step (an all_gather_object of dicts was 0.3-1 ms per call)
            import torch
            t = torch.zeros(3 * n, dtype=torch.int64)
            for i, cid in enumerate(contig_ids):
                if owner[i] == self.rank:
                    a, b, c = self.eng.result_sizes(cid)
                    t[i], t[n + i], t[2 * n + i] = int(a), int(b), int(c)
            if "gloo" in str(self.dist.get_backend()):
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
                v = t.tolist()
            else:                                          # a device-only backend cannot reduce a CPU tensor
                rows = [None] * self.world
                self.dist.all_gather_object(rows, t.tolist())
                v = [sum(col) for col in zip(*rows)]
            return owner, v[:n], v[n:2 * n], v[2 * n:]
        lens, nrec, pool = [0] * n, [0] * n, [0] * n
        for i, cid in enumerate(contig_ids):
            lens[i], nrec[i], pool[i] = (int(x) for x in self.eng.result_sizes(cid))
        return owner, lens, nrec, pool

    def gather_to_root(self, contig_ids, parts, root: int = 0, records: bool = True):
        """Every contig's mutated stream (and, ``records``, its record table + insert pool: the binary VCF) on ``root``.
        Returns (per slot the device addresses (stream, records, pool) on this rank, lengths, record counts, pool bytes)."""
        owner, lens, nrec, pool = self.sizes(contig_ids, parts)
        addrs = self.eng.gather_to_root(contig_ids, owner, lens, nrec if records else None, pool if records else None, root)
        return addrs, lens, nrec, pool

    def close(self):
        if self.live:
            self.eng.comm_destroy()
            self.live = False


class HostTransport:
    """Executes ``msim_gather_plan`` with CPU buffers over ``torch.distributed`` send / recv (gloo): the same
    bookkeeping as the RCCL path -- who sends what, sizes, posting order -- without a GPU."""

    def __init__(self, rank: int, world: int, dist):
        self.rank, self.world, self.dist = rank, world, dist

    def gather_to_root(self, payload: dict, owner, out_len, root: int = 0, n_records=None, pool_len=None):
        """payload: slot -> uint8 array (the stream), or slot -> (stream, record bytes, pool bytes) when ``n_records`` /
        ``pool_len`` are given, for the slots this rank owns.  Returns the same on root (parts as uint8 arrays)."""
        import torch
        three = n_records is not None
        ops = _ffi.gather_plan(owner, out_len, self.rank, self.world, root, n_records, pool_len)
        out, reqs, keep = {}, [], []

        def part_of(slot, part):
            v = payload[slot]
            return np.ascontiguousarray(v[part] if three else v).view(np.uint8).reshape(-1)
        for kind, slot, part, peer, nbytes in ops:
            if kind == 2:
                out[(slot, part)] = part_of(slot, part)
            elif kind == 0:
                t = torch.from_numpy(part_of(slot, part).copy())
                assert t.numel() == nbytes
                keep.append(t)
                reqs.append(self.dist.isend(t, peer, tag=3 * slot + part))
            else:
                t = torch.empty(nbytes, dtype=torch.uint8)
                out[(slot, part)] = t
                reqs.append(self.dist.irecv(t, peer, tag=3 * slot + part))
        for r in reqs:
            r.wait()
        got = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in out.items()}
        if not three:
            return {slot: v for (slot, part), v in got.items()}
        res = {}
        for (slot, part), v in got.items():
            res.setdefault(slot, [np.zeros(0, np.uint8)] * 3)
            res[slot] = list(res[slot])
            res[slot][part] = v
        return {slot: tuple(v) for slot, v in res.items()}
