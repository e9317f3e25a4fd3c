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
    """One per rank.  ``dist`` = an initialised ``torch.distributed`` module (any backend) or None for world 1."""

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

    def lengths(self, contig_ids, parts):
        """Mutated length of every contig, on every rank.  PLAN is replayed everywhere, so lengths it fixes are
        local knowledge; the rest is exchanged over the control plane by the owning rank."""
        owner = owners_of(parts, len(contig_ids))
        lens, missing = [], []
        for i, cid in enumerate(contig_ids):
            n, known = self.eng.planned_out_len(cid)
            if not known:
                missing.append(i)
                n = self.eng.result_sizes(cid)[0] if owner[i] == self.rank else 0
            lens.append(n)
        if missing and self.world > 1:
            mine = {i: lens[i] for i in missing if owner[i] == self.rank}
            every = [None] * self.world
            self.dist.all_gather_object(every, mine)
            for d in every:
                for i, n in d.items():
                    lens[i] = n
        return owner, lens

    def gather_to_root(self, contig_ids, parts, root: int = 0):
        owner, lens = self.lengths(contig_ids, parts)
        return self.eng.gather_to_root(contig_ids, owner, lens, root), lens

    def close(self):
        if self.live:
            self.eng.comm_destroy()
            self.live = False


class HostTransport:
    """Executes ``msim_gather_plan`` with CPU buffers over ``torch.distributed`` send / recv (gloo): the same
    bookkeeping as the RCCL path -- who sends what, sizes, posting order -- without a GPU."""

    def __init__(self, rank: int, world: int, dist):
        self.rank, self.world, self.dist = rank, world, dist

    def gather_to_root(self, payload: dict, owner, out_len, root: int = 0):
        """payload: slot -> uint8 numpy array for the slots this rank owns.  Returns slot -> array on root."""
        import torch
        ops = _ffi.gather_plan(owner, out_len, self.rank, self.world, root)
        out, reqs, keep = {}, [], []
        for kind, slot, peer, nbytes in ops:
            if kind == 2:
                out[slot] = payload[slot]
            elif kind == 0:
                t = torch.from_numpy(np.ascontiguousarray(payload[slot]))
                assert t.numel() == nbytes
                keep.append(t)
                reqs.append(self.dist.isend(t, peer, tag=slot))
            else:
                t = torch.empty(nbytes, dtype=torch.uint8)
                out[slot] = t
                reqs.append(self.dist.irecv(t, peer, tag=slot))
        for r in reqs:
            r.wait()
        return {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in out.items()}
