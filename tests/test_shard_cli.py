"""``--gpus N`` in the product CLI (multi_gpu.py): one worker process per GPU, the chain of both MT19937 streams walked
by every worker (msim_plan_chain for contigs it does not own), the owners' part files concatenated by the parent.

CPU tier: the workers run on host-only contexts (MSIM_SHARD_HOST_ONLY=1: PLAN through libmsim's host planner, VCF text
by the host renderer, no sequence) -- unit assignment, the chain-only advance, assembly order, warnings, error
propagation and the hand-back of the generator states are the same code as on GPUs.  GPU tier: the ranks share the
box's one device (MSIM_SHARD_DEVICES=0,0,..): complete Fasta + VCF bytes against the reference's goldens."""
from __future__ import annotations

import random

import numpy as np
import pytest

from helpers import CASES, case_meta, sha256
from pipeline import run_product_case


def _run_sharded(name, tmp_path, monkeypatch, gpus, host_only):
    import pipeline
    meta = case_meta(name)
    monkeypatch.setenv("MSIM_SHARD_DEVICES", ",".join(["0"] * gpus))
    if host_only:
        monkeypatch.setenv("MSIM_SHARD_HOST_ONLY", "1")
    real_prepare = pipeline.prepare
    monkeypatch.setattr(pipeline, "prepare", lambda m, t: ["--gpus", str(gpus)] + real_prepare(m, t))
    return meta, run_product_case(meta, tmp_path)


def _check_vcf_rng_stderr(meta, res):
    assert res["exception"] is None and res["exit_code"] is None, (res["exception"], res["stderr"])
    assert len(res["vcf"]) == meta["vcf_len"] and sha256(res["vcf"]) == meta["vcf_sha256"]
    assert res["stderr"] == meta["stderr"]
    assert [random.getrandbits(32) for _ in range(4)] == meta["py_next_words_after"]
    assert [int(x) for x in np.random.randint(0, 4294967296, size=4, dtype=np.uint32)] == meta["np_next_words_after"]


@pytest.mark.parametrize("name,gpus", [("svmix_2ctg_200k", 2), ("many_scaffolds", 2), ("many_scaffolds", 3),
                                       ("rmt_svstd_blocks_3mb", 2), ("rmt_quiet_none_std", 2), ("tl_rmt", 2),
                                       ("tiny_contigs", 4)])
def test_sharded_cli_host_only_workers(name, gpus, tmp_path, monkeypatch):
    meta, res = _run_sharded(name, tmp_path, monkeypatch, gpus, host_only=True)
    _check_vcf_rng_stderr(meta, res)
    # the Fasta holds every header in contig order (bodies need a GPU)
    heads = [l for l in res["fasta"].split(b"\n") if l.startswith(b">")]
    assert heads == [b">" + c["long_name"].encode() for c in meta["contigs"]]


def test_sharded_cli_propagates_the_references_value_error(tmp_path, monkeypatch):
    """Overlapping RMT ranges: PLAN raises the reference's ValueError on every rank at the same contig; the parent
    re-raises it once."""
    meta, res = _run_sharded("err_rmt_overlap", tmp_path, monkeypatch, 2, host_only=True)
    assert meta["exception"]["type"] == "ValueError"
    assert type(res["exception"]).__name__ == "ValueError" and str(res["exception"]) == meta["exception"]["message"]


def test_unit_owners_cover_every_unit_once():
    from mutation_simulator_amd.multi_gpu import unit_owners
    sizes = [5, 900_000, 37, 37, 250_000_000, 12, 90_000_000, 3]
    for world in (1, 2, 3, 8):
        own = unit_owners(sizes, world)
        assert len(own) == len(sizes) and set(own) <= set(range(world))
        loads = [sum(s for s, o in zip(sizes, own) if o == r) for r in range(world)]
        assert max(loads) >= sum(sizes) / world
    assert unit_owners(sizes, 2)[4] != unit_owners(sizes, 2)[6]          # the two big ones go to different ranks


GPU_CASES = [("svmix_2ctg_200k", 2), ("many_scaffolds", 2), ("rmt_svstd_blocks_3mb", 2), ("rmt_blocks_3mb", 3),
             ("c1_snp_1mb", 2), ("snp_titv2_2ctg", 2), ("tiny_contigs", 3), ("rmt_snblock_svstd_1500k", 2)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,gpus", GPU_CASES)
def test_sharded_cli_matches_reference_golden(name, gpus, tmp_path, monkeypatch):
    """Whole CLI with --gpus N (the ranks share device 0): files == what the reference wrote for the same argv + seeds."""
    meta, res = _run_sharded(name, tmp_path, monkeypatch, gpus, host_only=False)
    _check_vcf_rng_stderr(meta, res)
    assert len(res["fasta"]) == meta["fasta_len"] and sha256(res["fasta"]) == meta["fasta_sha256"]
    if meta["store"] == "full":
        assert res["fasta"] == (CASES / name / "expected_ms.fa").read_bytes()


@pytest.mark.gpu
def test_sharded_mutation_pass_then_it_pass(tmp_path, monkeypatch):
    """--gpus 2 on an RMT with mutations and `it` lines: the workers write the mutation pass, the parent takes the generator
    back and runs the interchromosomal-translocation pass over the assembled _ms Fasta -- all four files as the reference's."""
    name = "it_rmt_mutations"
    meta, res = _run_sharded(name, tmp_path, monkeypatch, 2, host_only=False)
    _check_vcf_rng_stderr(meta, res)
    assert res["fasta"] == (CASES / name / "expected_ms.fa").read_bytes()
    assert res["it_fasta"] == (CASES / name / "expected_ms_it.fa").read_bytes()
    assert res["bedpe"] == (CASES / name / "expected_ms_it.bedpe").read_bytes()


@pytest.mark.gpu
def test_sharded_cli_key_error_leaves_what_one_gpu_leaves(tmp_path, monkeypatch):
    """A transversion on a base outside AGTCN (mutator.py:449-455) in a contig owned by some rank: the KeyError reaches
    the caller and the files hold what the 1-GPU run leaves behind."""
    meta = case_meta("err_snp_on_U")
    (tmp_path / "one").mkdir()
    (tmp_path / "two").mkdir()
    one = run_product_case(meta, tmp_path / "one")
    meta, res = _run_sharded("err_snp_on_U", tmp_path / "two", monkeypatch, 2, host_only=False)
    assert type(res["exception"]).__name__ == "KeyError"
    assert repr(res["exception"].args[0]) == meta["exception"]["repr_args"][0]
    assert res["fasta"] == one["fasta"] and res["vcf"] == one["vcf"]
