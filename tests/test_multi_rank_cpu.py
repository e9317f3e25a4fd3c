"""N > 1 path on CPU: two processes over gloo (127.0.0.1).  Checks what must hold for the sharded
run to be correct by construction: (1) the LPT partition covers every contig exactly once and is
balanced, (2) every rank's replicated PLAN lands on identical records and identical stream
positions (so any rank may apply any contig), (3) the union of the per-rank apply sets is the
genome, (4) the gather's bookkeeping (libmsim's msim_gather_plan) delivers every contig to rank 0
when driven by a host-memory transport over gloo.  PLAN runs through libmsim's host-only context (no GPU here); APPLY itself is covered by the
single-GPU parity tests -- contigs are independent, sharding does not change a contig's result."""
from __future__ import annotations

import hashlib
import os
import socket
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str):
    for p in (ROOT, ROOT / "mutation-simulator_amd", ROOT / "tests", ROOT / "tests" / "golden"):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    import numpy as np
    import torch.distributed as dist

    import bench
    from mutation_simulator_amd import _ffi
    from mutation_simulator_amd import mutator as mm
    from mutation_simulator_amd.sharding import lpt_partition, run_sharded_pass

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths = bench.contig_lengths(6_000_000)            # the bench genome, scaled down 500x
        sim = bench.workload_settings(lengths, extra=["-in", "0.001", "-inmax", "8", "-de", "0.001", "-demax", "9"])
        parts = lpt_partition(lengths, world)
        eng = _ffi.Engine(device=-1)
        eng.seed(42, 42)
        eng.set_params(mm.params_descriptor(sim))
        cids = [eng.add_contig(np.zeros(L, dtype=np.uint8)) for L in lengths]
        run_sharded_pass(eng, sim, cids, parts[rank], mm.plan_descriptors, apply=False)
        digests = []
        for cid in cids:
            recs, pool = eng.fetch_records(cid)
            digests.append(hashlib.sha256(recs.tobytes() + pool.tobytes()).hexdigest())
        state = [(hashlib.sha256(eng.get_mt_state(s)[0].tobytes()).hexdigest(), eng.get_mt_state(s)[1]) for s in (0, 1)]
        # the pass as the product and the bench run it: contigs of OTHER ranks only walked (msim_plan_chain) -- the owned
        # ones must come out as above and both streams must end where the full replay ends
        eng.seed(42, 42)
        run_sharded_pass(eng, sim, cids, parts[rank], mm.plan_table, apply=False, lengths=lengths)
        for i in parts[rank]:
            recs, pool = eng.fetch_records(cids[i])
            assert hashlib.sha256(recs.tobytes() + pool.tobytes()).hexdigest() == digests[i]
        assert state == [(hashlib.sha256(eng.get_mt_state(s)[0].tobytes()).hexdigest(), eng.get_mt_state(s)[1]) for s in (0, 1)]
        # ---- the gather's bookkeeping (msim_gather_plan: who sends what, sizes, posting order), driven end to end
        # with a host-memory transport: every rank "applies" its contigs (deterministic stand-in bytes whose LENGTH
        # differs from the input length, like a mutated contig's), rank 0 must end up with every contig intact
        # The three parts of a slot travel: the mutated stream, the record table and the insert pool (the binary VCF).  Records
        # and pools are the REAL ones of the contigs this rank owns (sharded pass above); the root must be able to render every
        # contig's VCF lines from what it received -- byte-identical to what a single rank renders from its own full replay.
        from mutation_simulator_amd.gather import HostTransport, owners_of
        owner = owners_of(parts, len(lengths))
        out_len = [L + (i * 37) % 101 - 50 for i, L in enumerate(lengths)]

        def stand_in(i):
            return np.random.RandomState(1000 + i).randint(0, 256, out_len[i], dtype=np.uint8)

        def bases_of(i):
            return np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.RandomState(77 + i).randint(0, 4, lengths[i])]
        tables = {i: eng.fetch_records(cids[i]) for i in parts[rank]}
        n_rec_mine = {i: len(tables[i][0]) for i in tables}
        pool_mine = {i: len(tables[i][1]) for i in tables}
        every = [None] * world
        dist.all_gather_object(every, (n_rec_mine, pool_mine))          # sizes over the control plane, as gather.Communicator does
        n_records, pool_len = [0] * len(lengths), [0] * len(lengths)
        for a, b in every:
            for i, v in a.items():
                n_records[i] = v
            for i, v in b.items():
                pool_len[i] = v
        payload = {i: (stand_in(i), tables[i][0].copy(), tables[i][1].copy()) for i in parts[rank]}
        got = HostTransport(rank, world, dist).gather_to_root(payload, owner, out_len, root=0, n_records=n_records, pool_len=pool_len)
        gather_ok = vcf_ok = None
        if rank == 0:
            gather_ok = sorted(got) == list(range(len(lengths))) and all(np.array_equal(got[i][0], stand_in(i)) for i in got)
            # the single-rank answer: a full replay of PLAN on this rank (records of every contig)
            eng.seed(42, 42)
            run_sharded_pass(eng, sim, cids, list(range(len(lengths))), mm.plan_descriptors, apply=False)
            vcf_ok = True
            for i in range(len(lengths)):
                recs, pool = eng.fetch_records(cids[i])
                want = _ffi.render_vcf(recs, pool, bases_of(i), f"chr{i + 1}")
                g_recs = np.frombuffer(got[i][1].tobytes(), dtype=_ffi.RECORD_DTYPE)
                have = _ffi.render_vcf(g_recs, got[i][2], bases_of(i), f"chr{i + 1}")
                vcf_ok = vcf_ok and have == want and len(g_recs) == n_records[i] and len(want) > 0
        else:
            assert got == {}
        ops = _ffi.gather_plan(owner, out_len, rank, world, 0, n_records, pool_len)
        gathered = [None] * world
        dist.all_gather_object(gathered, {"rank": rank, "owned": parts[rank], "digests": digests, "state": state,
                                          "gather_ok": gather_ok, "vcf_ok": vcf_ok, "ops": ops})
        if rank == 0:
            import json
            Path(out_dir, "result.json").write_text(json.dumps({"parts": parts, "gathered": gathered,
                                                                "n_contigs": len(lengths)}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_gloo(tmp_path):
    import json

    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = json.loads((tmp_path / "result.json").read_text())
    g = res["gathered"]
    assert len(g) == 2 and {x["rank"] for x in g} == {0, 1}
    owned = sorted(i for x in g for i in x["owned"])
    assert owned == list(range(res["n_contigs"]))                 # every contig applied exactly once
    assert g[0]["digests"] == g[1]["digests"]                     # replicated PLAN is identical ...
    assert g[0]["state"] == g[1]["state"]                         # ... and so are both stream positions
    by_rank = {x["rank"]: x for x in g}
    assert by_rank[0]["gather_ok"] is True                        # root holds every contig, byte for byte
    assert by_rank[0]["vcf_ok"] is True                           # ... and renders every contig's VCF lines from what it received
    sends = [tuple(o) for o in by_rank[1]["ops"]]
    recvs = [tuple(o) for o in by_rank[0]["ops"] if o[0] == 1]
    assert [o[0] for o in sends] == [0] * len(sends) and sorted({o[1] for o in sends}) == by_rank[1]["owned"]
    assert {o[2] for o in sends} == {0, 1, 2}                     # streams, record tables and insert pools travel
    # every send of rank 1 meets a receive of rank 0 for the same slot, part and size, in the same order
    assert [(o[1], o[2], o[4]) for o in sends] == [(o[1], o[2], o[4]) for o in recvs] and all(o[3] == 1 for o in recvs)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_lpt_partition_covers_and_balances(world):
    import bench
    from mutation_simulator_amd.sharding import imbalance, lpt_partition
    lengths = bench.contig_lengths(3_000_000_000)
    parts = lpt_partition(lengths, world)
    assert sorted(i for p in parts for i in p) == list(range(len(lengths)))
    assert imbalance(lengths, parts) < 1.06                       # split further only above 5 %
    assert parts == lpt_partition(lengths, world)                 # deterministic on every rank
