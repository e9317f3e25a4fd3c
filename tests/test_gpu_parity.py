"""Parity tests proper: the HIP path, called through the C-ABI, against the reference goldens and
against the CPU oracle on seeded inputs.  Bar: bit-exact (integer / byte work)."""
from __future__ import annotations

import contextlib
import io
import random

import numpy as np
import pytest

from helpers import (CASES, all_case_names, case_input_bytes, case_meta, mask_vcf,
                     parse_fasta_bytes, sha256)
from mutation_simulator_amd import _ffi
from oracle import oracle as orc
from pipeline import run_product_case

pytestmark = pytest.mark.gpu

RUNNABLE = [n for n in all_case_names() if case_meta(n).get("sim") is not None]


@pytest.fixture(scope="module")
def engine():
    eng = _ffi.Engine(0)
    yield eng
    eng.close()


def test_device_is_mi355x(engine):
    assert "gfx950" in engine.device_name()


@pytest.mark.parametrize("name", RUNNABLE)
def test_cli_matches_reference_golden(name, tmp_path):
    """Whole CLI on the GPU == what the reference wrote for the same argv + seeds."""
    meta = case_meta(name)
    res = run_product_case(meta, tmp_path)
    if meta["exception"] is not None:
        assert type(res["exception"]).__name__ == meta["exception"]["type"]
        if meta["exception"]["type"] == "KeyError":
            assert repr(res["exception"].args[0]) == meta["exception"]["repr_args"][0]
        else:
            assert str(res["exception"]) == meta["exception"]["message"]
        return
    assert res["exception"] is None and res["exit_code"] is None, (res["exception"], res["stderr"])
    if "fasta_len" in meta:                     # (absent in `it` mode: no mutation pass, no _ms files)
        assert len(res["fasta"]) == meta["fasta_len"] and sha256(res["fasta"]) == meta["fasta_sha256"]
        assert len(res["vcf"]) == meta["vcf_len"] and sha256(res["vcf"]) == meta["vcf_sha256"]
        if meta["store"] == "full":
            assert res["fasta"] == (CASES / name / "expected_ms.fa").read_bytes()
            assert res["vcf"] == (CASES / name / "expected_ms.vcf").read_bytes()
    else:
        assert res["fasta"] is None and res["vcf"] is None
    if "it_fasta_len" in meta:                  # the interchromosomal-translocation pass: its Fasta and its BEDPE
        assert len(res["it_fasta"]) == meta["it_fasta_len"] and sha256(res["it_fasta"]) == meta["it_fasta_sha256"]
        assert len(res["bedpe"]) == meta["bedpe_len"] and sha256(res["bedpe"]) == meta["bedpe_sha256"]
        if meta["store"] == "full":
            assert res["it_fasta"] == (CASES / name / "expected_ms_it.fa").read_bytes()
            assert res["bedpe"] == (CASES / name / "expected_ms_it.bedpe").read_bytes()
    else:
        assert res["it_fasta"] is None and res["bedpe"] is None
    assert res["stderr"] == meta["stderr"]
    assert [random.getrandbits(32) for _ in range(4)] == meta["py_next_words_after"]


_OVERFLOW_SCRIPT = r"""
import json, sys, tempfile
from pathlib import Path
sys.path[:0] = {paths!r}
from helpers import CASES, case_meta, sha256
from pipeline import run_product_case
from mutation_simulator_amd import mutator
meta = case_meta({name!r})
res = run_product_case(meta, Path(tempfile.mkdtemp()))
print(json.dumps(dict(exc=repr(res["exception"]), code=res["exit_code"], fasta=sha256(res["fasta"]), vcf=sha256(res["vcf"]),
                      stderr=res["stderr"], replanned=mutator.REPLANNED_CONTIGS)))
"""


@pytest.mark.parametrize("name,nth", [("c1_snp_1mb", 1), ("svmix_1mb", 1), ("rmt_blocks_3mb", 1), ("rmt_blocks_3mb", 2),
                                      ("rmt_svmix_blocks_3mb", 2), ("rmt_svstd_blocks_3mb", 1),
                                      ("rmt_snblock_svstd_1500k", 2)])
def test_window_overflow_is_replanned_on_the_host(name, nth):
    """A device PLAN engine that reports a 16-sigma window overflow (forced here by the MSIM_DBG_FORCE_OVERFLOW test hook,
    one case per engine: SNP sampler, SV mix, host-cut, host-chain) must not cost the run: the host package puts the streams back and
    plans that contig with the sequential host planner -- the files still equal the reference's.  nth = 2: the overflow
    hits a later contig, so the streams are first advanced over the earlier ones again."""
    import json
    import os
    import subprocess
    import sys
    meta = case_meta(name)
    env = dict(os.environ, MSIM_DBG_FORCE_OVERFLOW=str(nth))
    out = subprocess.run([sys.executable, "-c", _OVERFLOW_SCRIPT.format(paths=sys.path[:8], name=name)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["exc"] == "None" and got["code"] is None
    assert got["replanned"] == 1
    assert got["fasta"] == meta["fasta_sha256"] and got["vcf"] == meta["vcf_sha256"]
    assert got["stderr"] == meta["stderr"]


# ------------------------------------------------------------------ seeded sweeps vs the oracle
def _sim_dump_from(sim):
    from test_host_settings import dump_sim
    return dump_sim(sim)


def _product_vs_oracle(tmp_path, spec, argv_tail, seed_py, seed_np, rmt_text=None):
    import inputs as gin
    import mutation_simulator_amd as msa
    from mutation_simulator_amd import __main__ as msa_main
    infile = gin.write_input(spec, tmp_path / "in.fa")
    tail = list(argv_tail)
    if rmt_text is not None:
        (tmp_path / "c.rmt").write_text(rmt_text)
        tail = ["rmt", str(tmp_path / "c.rmt")]
    argv = ["-q", "-o", str(tmp_path / "out"), str(infile)] + tail
    random.seed(seed_py)
    np.random.seed(seed_np)
    with contextlib.redirect_stderr(io.StringIO()):
        msa_main.main(argv)
    got_fa = (tmp_path / "out_ms.fa").read_bytes()
    got_vcf = mask_vcf((tmp_path / "out_ms.vcf").read_bytes())
    # oracle on the same settings tree
    with contextlib.redirect_stderr(io.StringIO()):
        args = msa.get_args(argv)
        fasta = msa.load_fasta(args.infile)
        sim = (msa.SimulationSettings.from_args(args, fasta, True) if args.mode == "args"
               else msa.SimulationSettings.from_rmt(args.rmtfile, fasta, True))
    o = orc.Oracle()
    o.seed(seed_py, seed_np)
    contigs = parse_fasta_bytes(infile.read_bytes())
    fa, vcf, _, _ = o.run_genome(contigs, _sim_dump_from(sim), infile.name)
    assert got_fa == fa
    assert got_vcf == vcf


SWEEP = [
    ("snp_5mb", [5_000_000], ["args", "-sn", "0.01", "-titv", "2.0"]),
    ("sv_3mb", [2_000_000, 1_000_003], ["args", "-sn", "0.005", "-in", "0.001", "-inmin", "1", "-inmax", "50",
                                        "-de", "0.001", "-demin", "1", "-demax", "50", "-du", "0.0005",
                                        "-dumin", "50", "-dumax", "500", "-iv", "0.0005", "-ivmin", "50",
                                        "-ivmax", "500"]),
    ("dense_snp", [300_000], ["args", "-sn", "0.3", "-titv", "0.7"]),            # > 1024 records / tile
    ("dense_mix", [200_000, 77], ["args", "-sn", "0.1", "-in", "0.1", "-inmax", "9", "-de", "0.1", "-demax", "7",
                                  "-du", "0.05", "-dumax", "11", "-iv", "0.05", "-ivmax", "13"]),
    ("long_sv", [400_000], ["args", "-de", "0.0002", "-demin", "500", "-demax", "40000", "-du", "0.0002",
                            "-dumin", "300", "-dumax", "30000", "-iv", "0.0002", "-ivmin", "100", "-ivmax",
                            "20000", "-in", "0.0002", "-inmin", "200", "-inmax", "5000"]),
    ("tl_mix", [300_000, 5_000, 40], ["args", "-tl", "0.02", "-tlmin", "1", "-tlmax", "300", "-tlb", "2", "-sn", "0.01",
                                      "-in", "0.003", "-inmax", "7", "-du", "0.002", "-dumax", "50", "-iv", "0.002",
                                      "-ivmax", "40", "-de", "0.002", "-demax", "30"]),
    ("tiny", [1, 2, 15, 16, 17, 31, 33, 64, 4095, 4096, 4097, 16383, 16384, 16385],
     ["args", "-sn", "0.2", "-in", "0.05", "-de", "0.05"]),
]


@pytest.mark.parametrize("name,lengths,argv", SWEEP, ids=[s[0] for s in SWEEP])
@pytest.mark.parametrize("seed", [1, 2])
def test_sweep_vs_oracle(name, lengths, argv, seed, tmp_path):
    spec = {"contigs": [{"defline": f"s{i} d", "length": L, "bpl": 60 + 7 * (i % 3), "seed": 100 * seed + i,
                         "decorate": (i % 2 == 1) and L > 2000} for i, L in enumerate(lengths)]}
    _product_vs_oracle(tmp_path, spec, argv, seed, seed + 17)


def test_random_parameter_sweep_vs_oracle(tmp_path):
    """Randomised (fixed-seed) argument sets: rates, length bounds, blocks, titv, odd contig sizes."""
    rs = np.random.RandomState(2024)
    for it in range(12):
        rates = rs.random_sample(5) * rs.choice([0.002, 0.02, 0.08])
        rates[rs.randint(0, 5)] = 0.0
        argv = ["args", "-titv", str(round(float(rs.random_sample() * 4), 3))]
        for flag, r in zip(["sn", "in", "de", "iv", "du"], rates):
            argv += [f"-{flag}", repr(float(r))]
            if flag != "sn":
                lo = int(rs.randint(2 if flag == "iv" else 1, 30))
                argv += [f"-{flag}min", str(lo), f"-{flag}max", str(lo + int(rs.randint(0, 200)))]
            argv += [f"-{flag}b", str(int(rs.randint(1, 8)))]
        lengths = [int(rs.randint(1, 400_000)) for _ in range(int(rs.randint(1, 4)))]
        spec = {"contigs": [{"defline": f"r{it}_{i}", "length": L, "bpl": int(rs.randint(20, 100)),
                             "seed": 1000 * it + i, "decorate": bool(rs.randint(0, 2)) and L > 2000}
                            for i, L in enumerate(lengths)]}
        d = tmp_path / f"it{it}"
        d.mkdir()
        _product_vs_oracle(d, spec, argv, 10 + it, 90 + it)


def test_rmt_many_ranges_vs_oracle(tmp_path):
    rs = np.random.RandomState(7)
    L = 600_000
    cuts = sorted(set(int(x) for x in rs.randint(1, L, 120)))
    rows = []
    for a, b in zip(cuts[::2], cuts[1::2]):
        kind = rs.randint(0, 4)
        if b - a < 3:
            continue
        if kind == 0:
            rows.append(f"{a+1}-{b} None")
        elif kind == 1:
            rows.append(f"{a+1}-{b} sn 0.05")
        elif kind == 2:
            rows.append(f"{a+1}-{b} sn 0.3 in 0.02 inmin 1 inmax 4")
        else:
            rows.append(f"{a+1}-{b} sn 0.001 de 0.002 demin 3 demax 60 iv 0.001 ivmin 5 ivmax 90 du 0.001 dumin 4 dumax 80")
    text = "titv = 1.7\nsn_block = 2\nstd\nit None\nsn 0.01\nchr 1\n" + "\n".join(rows) + "\n"
    spec = {"contigs": [{"defline": "big rmt", "length": L, "bpl": 60, "seed": 5},
                        {"defline": "other", "length": 50_000, "bpl": 60, "seed": 6}]}
    _product_vs_oracle(tmp_path, spec, [], 3, 4, rmt_text=text)


# ------------------------------------------------------------------ device helpers
from oracle.support import checksum_host, mix64, synth_host  # noqa: E402,F401  (host twins of k_synth / k_checksum)


@pytest.mark.parametrize("length", [1, 31, 32, 33, 1000, 1 << 20, (1 << 20) + 17])
def test_synthetic_contig_and_checksum(engine, length):
    engine.clear()
    cid = engine.add_contig_synthetic(length, 99)
    got = engine.read_contig(cid)
    want = synth_host(length, 99)
    assert np.array_equal(got, want)
    p = _ffi.Params()
    for i in range(8):
        p.block[i] = 1
    engine.set_params(p)
    engine.plan_contig(cid, [])
    engine.apply_contig(cid)
    assert np.array_equal(engine.fetch_sequence(cid), want)          # no records: identity
    assert engine.result_checksum(cid) == checksum_host(want)
    engine.clear()


def _plan_apply_snp_contig(engine, seed=5):
    engine.clear()
    engine.seed(seed, seed + 1)
    p = _ffi.Params()
    for i in range(8):
        p.block[i] = 1
    p.ti_lim = (1 << 52) + 1
    engine.set_params(p)
    cid = engine.add_contig_synthetic(3_000_001, 4)
    r = _ffi.Range()
    r.start, r.stop, r.k, r.setsize, r.n_types = 0, 3_000_000, 30_000, 262165, 1
    r.types[0] = 1
    r.cdf_thr[0] = 1 << 53
    engine.plan_contig(cid, [r])
    engine.apply_contig(cid)
    return cid


def test_gather_single_rank_and_rccl_loopback(engine):
    """World 1: the gather is the identity (device address of the contig's own buffer).  Then libmsim's RCCL path
    for real: librccl dlopen()ed, a 1-rank communicator, and the mutated contig sent to and received from rank 0
    itself (grouped ncclSend + ncclRecv) -- the bytes that come back must be the mutated contig."""
    import ctypes as C
    from mutation_simulator_amd.gather import Communicator
    cid = _plan_apply_snp_contig(engine)
    want = engine.fetch_sequence(cid)
    comm = Communicator(engine, 0, 1, None)
    addrs, lens, nrec, pool = comm.gather_to_root([cid], [[0]])
    assert lens == [len(want)] and addrs[0][0] == engine.result_device_ptr(cid)[0]
    recs, _ = engine.fetch_records(cid)
    assert nrec == [len(recs)] and pool == [0] and addrs[0][2] == 0
    assert np.array_equal(engine.gather_fetch(addrs[0][1], 16 * nrec[0], _ffi.RECORD_DTYPE).view(np.uint8), recs.view(np.uint8))
    engine.comm_init(_ffi.comm_unique_id(), 0, 1)
    lib = _ffi.load()
    lib.msim_dbg_comm_loopback.restype = C.c_int
    lib.msim_dbg_comm_loopback.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    s = C.c_uint64()
    rc = lib.msim_dbg_comm_loopback(engine.h, cid, C.byref(s))
    assert rc == 0, lib.msim_last_error(engine.h).decode()
    assert s.value == checksum_host(want) == engine.result_checksum(cid)
    engine.comm_destroy()
    engine.clear()


def _sharded_worker(rank, world, port, out_dir, transport, one_device):
    """One rank of BASELINE configs[4]'s shape: ONE genome, contigs LPT-sharded, every rank runs ``bench.one_step`` exactly as
    ``bench.py --gpus N`` does (plan + apply what it owns, ``msim_plan_chain`` through the rest, one synchronisation, no read in
    between), then everything goes to rank 0 -- ``transport`` "rccl": libmsim's grouped ncclSend / ncclRecv over xGMI into device
    buffers; "host": fetched and sent over the gloo control plane (so that the same steps and checks run with both ranks on ONE
    GPU, where RCCL refuses to open a communicator).  Rank 0 holds every contig against the CPU ORACLE (mutated stream + VCF
    lines rendered from the gathered records and pools), for -sn 0.01 -titv 2.0 and for the full SV mix; then the same genome
    in fast mode (``--rng fast``: a rank plans and applies only what it owns) against the numpy twin and a 1-rank run."""
    import hashlib
    import json
    import os
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for q in (root, root / "mutation-simulator_amd", root / "tests", root / "tests" / "golden"):
        if str(q) not in sys.path:
            sys.path.insert(0, str(q))
    import numpy as np
    import torch.distributed as dist
    import bench
    import fast_twin as ft
    from mutation_simulator_amd import _ffi as ffi
    from mutation_simulator_amd import mutator as mm
    from mutation_simulator_amd.gather import Communicator
    from mutation_simulator_amd.sharding import lpt_partition
    from oracle import oracle as orc
    from test_fast_host import _twin_ranges, assert_plan_equals_twin
    from test_gpu_parity import synth_host
    from test_host_settings import dump_sim
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    report = {"ok": True, "checked": []}
    try:
        lengths = bench.contig_lengths(60_000_000)
        parts = lpt_partition(lengths, world)
        owner = {i: r for r, p in enumerate(parts) for i in p}
        device = 0 if one_device else rank

        def gather(eng, cids, comm):
            """rank 0: per contig (mutated stream, record table, insert pool) as host arrays; other ranks: None"""
            if transport == "rccl":
                addrs, lens, nrec, pool = comm.gather_to_root(cids, parts)
                if rank:
                    return None
                return [(eng.gather_fetch(a, n), eng.gather_fetch(ra, 16 * nr, ffi.RECORD_DTYPE), eng.gather_fetch(pa, pl))
                        for (a, ra, pa), n, nr, pl in zip(addrs, lens, nrec, pool)]
            mine = {}
            for i in parts[rank]:
                recs, pl = eng.fetch_records(cids[i])
                mine[i] = (eng.fetch_sequence(cids[i]), recs, pl)
            every = [None] * world if rank == 0 else None
            dist.gather_object(mine, every, dst=0)
            if rank:
                return None
            merged = {}
            for d in every:
                merged.update(d)
            return [merged[i] for i in range(len(lengths))]

        # ---- compatible streams: the step bench.py times, sharded
        for workload in ("c2", "c3"):
            sim = bench.build_settings(workload, lengths)
            eng = ffi.Engine(device)
            eng.set_params(mm.params_descriptor(sim))
            cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
            comm = Communicator(eng, rank, world, dist) if transport == "rccl" else None
            bench.one_step(eng, sim, cids, parts[rank], 42, mm.plan_table, lengths)
            got = gather(eng, cids, comm)
            if rank == 0:
                dump = dump_sim(sim)
                o = orc.Oracle()
                o.seed(42, 42)
                o.configure(dump)
                by_number = {ch["number"]: ch for ch in dump["chromosomes"]}
                for chrom in sim.chromosomes:
                    i = chrom.number
                    name = f"chr{i + 1}"
                    bases = synth_host(lengths[i], 1000 + i)
                    fa, vcf, _ = o.mutate_contig_stream(bases, name, f"{name} synthetic", 60, by_number[i]["ranges"])
                    body = np.frombuffer(fa, dtype=np.uint8)[len(f">{name} synthetic\n"):]
                    seq, recs, pool = got[i]
                    same = np.array_equal(body[body != 10], seq) and ffi.render_vcf(recs, pool, bases, name) == bytes(vcf)
                    report["ok"] = report["ok"] and bool(same)
                    report["checked"].append([workload, i, owner[i], bool(same)])
            dist.barrier()
            if comm is not None:
                comm.close()
            eng.close()
        # ---- fast mode: a rank plans and applies only what it owns (bench.fast_rng_sharded's step)
        sim = bench.build_settings("c3", lengths)
        params = mm.params_descriptor(sim)
        tables = [mm.plan_descriptors(ch) for ch in sim.chromosomes]
        eng = ffi.Engine(device, ffi.RNG_FAST)
        eng.set_params(params)
        cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
        comm = Communicator(eng, rank, world, dist) if transport == "rccl" else None
        eng.set_fast_key(42)
        for ch, t in zip(sim.chromosomes, tables):
            if ch.number in parts[rank]:
                eng.plan_contig(cids[ch.number], t)
                eng.apply_contig(cids[ch.number])
            else:
                eng.plan_chain(lengths[ch.number], t)
        eng.sync()
        got = gather(eng, cids, comm)
        if rank == 0:
            blocks = {t: int(params.block[t]) for t in range(1, 8)}
            eng.set_fast_key(42)                                       # the 1-rank answer: everything planned + applied here
            for ch, t in zip(sim.chromosomes, tables):
                eng.plan_contig(cids[ch.number], t)
                eng.apply_contig(cids[ch.number])
            eng.sync()
            for ch, t in zip(sim.chromosomes, tables):
                i = ch.number
                seq, recs, pool = got[i]
                twin = ft.plan(lengths[i], _twin_ranges(t), blocks, int(params.ti_lim), 42, i)
                same = True
                try:
                    assert_plan_equals_twin(recs, pool, len(recs) == 0, twin)
                except AssertionError:
                    same = False
                same = same and np.array_equal(eng.fetch_sequence(cids[i]), seq)
                report["ok"] = report["ok"] and bool(same)
                report["checked"].append(["fast c3", i, owner[i], bool(same)])
        dist.barrier()
        if comm is not None:
            comm.close()
        eng.close()
        if rank == 0:
            Path(out_dir, "sharded.json").write_text(json.dumps(report))
    finally:
        dist.destroy_process_group()


def _run_sharded(tmp_path, transport, one_device):
    """Two worker interpreters, started as plain subprocesses: this process must NOT import torch -- the PyTorch-ROCm wheel
    brings a HIP runtime of its own, and a second runtime beside the one libmsim has already loaded here ends the test run
    with a double free at interpreter exit (the workers import torch first, as bench.py does)."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path

    from mutation_simulator_amd.multi_gpu import die_with_parent
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    here = str(Path(__file__).resolve().parent)
    code = ("import sys; sys.path.insert(0, {here!r}); import conftest  # noqa: F401 (sys.path of the test tree)\n"
            "from test_gpu_parity import _sharded_worker\n"
            "_sharded_worker({rank}, 2, {port}, {out!r}, {transport!r}, {one!r})\n")
    procs = [subprocess.Popen([sys.executable, "-c", code.format(here=here, rank=r, port=port, out=str(tmp_path), transport=transport,
                                                                  one=one_device)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, preexec_fn=die_with_parent)
             for r in range(2)]
    try:
        outs = [p.communicate(timeout=540)[0] for p in procs]
    finally:                                               # (a worker that outlives its test would keep the GPU)
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    res = json.loads((tmp_path / "sharded.json").read_text())
    bad = [c for c in res["checked"] if not c[3]]
    assert res["ok"] and not bad, bad
    assert len(res["checked"]) == 3 * 24
    assert {c[2] for c in res["checked"]} == {0, 1}                    # contigs of both ranks were held against the oracle


def test_two_ranks_one_gpu_sharded_step_vs_oracle(tmp_path):
    """The sharded step of ``bench.py --gpus 2`` with both ranks on ONE GPU (what the 1-GPU boxes of this pool can run): the
    same worker as the 2-GPU test below, results exchanged over the control plane instead of RCCL."""
    _run_sharded(tmp_path, "host", True)


def test_two_gpu_sharded_apply_and_rccl_gather(tmp_path):
    """BASELINE configs[4]'s shape on two GPUs: one genome, contigs LPT-sharded, ``msim_plan_chain`` for contigs a rank does
    not own, streams + record tables + insert pools gathered to rank 0 over libmsim's RCCL communicator -- compat streams
    against the oracle, fast mode against the twin.  Needs two visible GPUs."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    n = C.c_int()
    if hip.hipGetDeviceCount(C.byref(n)) != 0 or n.value < 2:
        pytest.skip("needs two GPUs")
    _run_sharded(tmp_path, "rccl", False)


def test_rmt_overlapping_large_snp_ranges_vs_oracle(tmp_path):
    """Two overlapping SN-only ranges, each large enough for the device SNP sampler (k >= 4096), std None:
    the reference does not raise (the negative-length filler has no mutations) and merges the two per-range
    dicts with update() -- later range wins (mutator.py:121).  Only the host planner reproduces that, so the
    device engines must decline the contig (round-1 advisor finding: they did not)."""
    text = "std\nit None\nNone\nchr 1\n1-600000 sn 0.02\n400001-1000000 sn 0.02\n"
    spec = {"contigs": [{"defline": "ovl big", "length": 1_200_000, "bpl": 60, "seed": 11}]}
    _product_vs_oracle(tmp_path, spec, [], 8, 9, rmt_text=text)


def test_svmix_lengths_beyond_table_entries_vs_oracle(tmp_path):
    """randint widths of 2^24 and more do not fit an entry of the boundary walk's next-accepted-draw tables: the SV-mix
    engine then walks tempered words with the retry loop inside (chain_boundary_host).  Deletions / duplications that long
    are clamped to the contig end (mutator.py:253-264), so the case is small; checked against the oracle."""
    argv = ["args", "-sn", "0.004", "-de", "0.00002", "-demin", "1", "-demax", "20000000",
            "-du", "0.00002", "-dumin", "5", "-dumax", "17000000", "-in", "0.0005", "-inmin", "1", "-inmax", "9"]
    spec = {"contigs": [{"defline": "wide lens", "length": 2_400_000, "bpl": 60, "seed": 21},
                        {"defline": "second", "length": 1_100_000, "bpl": 70, "seed": 22}]}
    _product_vs_oracle(tmp_path, spec, argv, 5, 6)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_it_mode_random_sweep_vs_oracle(seed, tmp_path):
    """`mutation-simulator file it <rate>` on random contig sets (1-base and 3-base contigs, odd counts, lower case and
    ambiguity codes in the input, rates from "no breakpoint" to "ValueError on the shorter contig") against the oracle's
    restatement of it_mutator.py: _ms_it Fasta, BEDPE, warnings and the generator's position."""
    import inputs as gin
    from mutation_simulator_amd import __main__ as msa_main
    rs = np.random.RandomState(500 + seed)
    for it in range(6):
        n = int(rs.randint(2, 10))
        lengths = [int(rs.choice([1, 2, 3, 5, 40, 700, 9_000, 120_000, 400_000])) + int(rs.randint(0, 50)) * (rs.rand() < 0.7)
                   for _ in range(n)]
        spec = {"contigs": [{"defline": f"z{it}_{i} it fuzz", "length": int(L), "bpl": int(rs.choice([50, 60, 61, 80])),
                             "seed": 7_000 * seed + 20 * it + i, "decorate": bool(rs.rand() < 0.4) and L > 2000}
                            for i, L in enumerate(lengths)]}
        rate = float(rs.choice([1e-7, 1e-4, 2e-3, 0.03, 0.3, 0.5]))
        d = tmp_path / f"it{it}"
        d.mkdir()
        infile = gin.write_input(spec, d / "in.fa")
        sp = int(rs.randint(0, 1 << 30))
        random.seed(sp)
        err = io.StringIO()
        argv = ["-o", str(d / "out"), str(infile), "it", repr(rate)]
        code = None
        with contextlib.redirect_stderr(err), contextlib.redirect_stdout(io.StringIO()):
            try:
                msa_main.main(argv)
            except SystemExit as e:                   # fewer than two usable contigs: ITNotEnoughAvailChromsError -> exit(1)
                code = e.code
        contigs = parse_fasta_bytes(infile.read_bytes())
        usable = [c for c in contigs if len(c["bases"]) > 2]
        if len(usable) < 2:
            assert code == 1
            continue
        assert code is None, err.getvalue()
        nxt = [random.getrandbits(32) for _ in range(4)]
        o = orc.Oracle()
        o.seed(sp, 0)
        fa, bedpe, ws = o.it_pass(contigs, [rate] * len(contigs))
        assert (d / "out_ms_it.fa").read_bytes() == fa
        assert (d / "out_ms_it.bedpe").read_bytes() == bedpe
        assert err.getvalue() == "".join(f"WARNING: {w}\n" for w in ws)
        assert nxt == o.py_words32(4)
