"""A run's MT19937 streams have no length limit in the reference (mutator.py:105-142 draws from the two global generators for as
long as the genome lasts; util.py:104-109).  The device streams are generated from jump-ahead tables that reach 8192 chunks
(1.31 G words) from a session's origin -- ``-sn 0.01`` gets there near 20 Gb of genome.  ``gpu_plan_make_room`` therefore
RE-BASES the session in front of a contig that would not fit: the states at both streams' exact positions become the next
session's origin.  ``MSIM_DBG_JUMP_MAX_CHUNKS`` (read when a context is created) shrinks the span so that small genomes re-base
every few contigs; everything below is compared with the CPU ORACLE (or with the reference's own goldens through the CLI),
which walks both streams sequentially and knows nothing of sessions."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import numpy as np
import pytest

import bench
from helpers import case_meta
from mutation_simulator_amd import _ffi
from mutation_simulator_amd import mutator as mm
from test_gpu_bench_order import _bench_order_vs_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload,engine,chunks,n,lo,hi", [
    ("c2", "contigs_snp", 4, 10, 3_000_000, 5_000_000),            # 40 Mb of -sn 0.01 -titv 2.0: the SNP sampler
    ("c3", "contigs_svmix", 3, 12, 2_000_000, 4_000_000),           # the SV mix: boundary windows, insert pool (NumPy stream)
    ("c4", "contigs_hostcut", 3, 12, 4_000_000, 6_000_000),        # RMT gene blocks: one window per contig, host cuts
    ("c4sv", "contigs_hostchain", 4, 14, 4_000_000, 6_000_000),    # RMT + SV std line: samples + chains in one window
])
def test_small_span_rebases_vs_oracle(workload, engine, chunks, n, lo, hi, monkeypatch):
    """Every device engine across >= 3 re-bases, in the bench's order (nothing read between contigs): Fasta body + VCF text of
    every contig and both final stream positions equal the oracle's."""
    monkeypatch.setenv("MSIM_DBG_JUMP_MAX_CHUNKS", str(chunks))
    rs = np.random.RandomState(chunks * 100 + n)
    lengths = [int(x) for x in rs.randint(lo, hi, size=n)]
    st = _bench_order_vs_oracle(workload, lengths)
    assert st[engine] == n, {k: v for k, v in st.items() if k.startswith("contigs_") and v}
    assert st["stream_rebases"] >= 3, st["stream_rebases"]


def test_default_span_does_not_rebase():
    """The shipped span (8192 chunks): a 100 Mb genome is nowhere near it."""
    st = _bench_order_vs_oracle("c2", [30_000_000, 50_000_000, 20_000_000])
    assert st["stream_rebases"] == 0 and st["contigs_snp"] == 3


def test_contig_beyond_any_span_goes_to_the_host_planner(monkeypatch):
    """A contig whose windows no span holds (here: a 2-chunk span and a 20 Mb contig) is planned on the host in AUTO mode --
    same bytes -- and refused where a device engine is forced."""
    monkeypatch.setenv("MSIM_DBG_JUMP_MAX_CHUNKS", "2")
    lengths = [2_000_000, 20_000_000, 2_000_000]
    st = _bench_order_vs_oracle("c2", lengths)
    assert st["contigs_host"] == 1 and st["contigs_snp"] == 2
    sim = bench.build_settings("c2", lengths)
    eng = _ffi.Engine(0, _ffi.PLAN_GPU)
    eng.seed(1, 1)
    eng.set_params(mm.params_descriptor(sim))
    cid = eng.add_contig_synthetic(lengths[1], 5)
    with pytest.raises(_ffi.MsimUnsupported):
        eng.plan_contig(cid, mm.plan_table(sim.chromosomes[1]))
    eng.close()


def test_non_owning_rank_rebases_along_the_chain(monkeypatch):
    """``msim_plan_chain`` (contigs another rank owns: stream positions only) re-bases like a full plan: a rank that owns every
    third contig still produces the oracle's bytes for those."""
    monkeypatch.setenv("MSIM_DBG_JUMP_MAX_CHUNKS", "4")
    rs = np.random.RandomState(11)
    lengths = [int(x) for x in rs.randint(3_000_000, 5_000_000, size=9)]
    for workload in ("c2", "c3"):
        st = _bench_order_vs_oracle(workload, lengths, owned=[1, 4, 7])
        assert st["stream_rebases"] >= 3


_CLI_SCRIPT = """
import sys, json, tempfile
sys.path[:0] = {paths!r}
from pathlib import Path
from helpers import case_meta, sha256
from pipeline import run_product_case
from mutation_simulator_amd import mutator
meta = case_meta({name!r})
td = Path(tempfile.mkdtemp())
res = run_product_case(meta, td, extra_argv=["--bench-json", str(td / "stats.json")])
st = json.loads((td / "stats.json").read_text())
print(json.dumps(dict(exc=repr(res["exception"]), code=res["exit_code"], fasta=sha256(res["fasta"]), vcf=sha256(res["vcf"]),
                      stderr=res["stderr"], replanned=mutator.REPLANNED_CONTIGS, rebases=st.get("stream_rebases"))))
"""


@pytest.mark.parametrize("name,chunks,min_rebases", [("rmt_blocks_3mb", 2, 1), ("rmt_svstd_blocks_3mb", 2, 0),
                                                     ("snp_titv2_2ctg", 2, 0), ("readme_mix_tl", 2, 0), ("many_scaffolds", 2, 0)])
def test_cli_goldens_with_a_small_span(name, chunks, min_rebases):
    """The reference's own outputs (goldens) through the whole CLI with a 2-chunk span: device engines, batches of small contigs
    and the host planner alternate along one pair of streams, sessions end wherever the next contig would not fit."""
    meta = case_meta(name)
    env = dict(os.environ, MSIM_DBG_JUMP_MAX_CHUNKS=str(chunks))
    out = subprocess.run([sys.executable, "-c", _CLI_SCRIPT.format(paths=sys.path[:8], name=name)], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["exc"] == "None" and got["code"] is None
    assert got["replanned"] == 0                                   # (no window overflowed: the re-base came first)
    assert got["fasta"] == meta["fasta_sha256"] and got["vcf"] == meta["vcf_sha256"]
    assert got["stderr"] == meta["stderr"]
    assert got["rebases"] is not None and got["rebases"] >= min_rebases, got
