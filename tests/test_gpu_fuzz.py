"""Bounded differential fuzz of the whole product path on the GPU against the CPU ORACLE (not the product's own host
planner): random genomes and random ARGS / RMT settings shaped to reach every device PLAN engine in AUTO mode
(SNP sampler, SV mix, host-sampled RMT contigs, host planner for the rest), Fasta + VCF bytes compared.  The
generators are those of tests/fuzz_engines.py (the opt-in long-running variant); seeds are fixed, so the test is
deterministic.  Settings the reference rejects (over-dense ranges -> ValueError) must be rejected identically."""
from __future__ import annotations

import numpy as np
import pytest

from fuzz_engines import args_settings, args_settings_round3, rmt_settings, rmt_settings_round3
from test_gpu_parity import _product_vs_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_device_engines_vs_oracle_fuzz(seed, tmp_path):
    rs = np.random.RandomState(seed)
    ran = 0
    for it in range(8):
        lengths = [int(rs.choice([300_000, 700_000, 1_500_000, 4_000_000]) + rs.randint(0, 5000))
                   for _ in range(int(rs.randint(1, 4)))]
        if rs.rand() < 0.2:
            lengths.append(int(rs.randint(1, 3000)))     # a tiny contig in the middle of the stream chain
        mode = rs.choice(["args", "args", "rmt"])
        argv, rmt = args_settings(rs) if mode == "args" else rmt_settings(rs, lengths)
        spec = {"contigs": [{"defline": f"f{it}_{i} fuzz", "length": L, "bpl": int(rs.choice([50, 60, 61, 80])),
                             "seed": 10_000 * seed + 10 * it + i} for i, L in enumerate(lengths)]}
        sp, sn = int(rs.randint(0, 1 << 30)), int(rs.randint(0, 1 << 30))
        d = tmp_path / f"it{it}"
        d.mkdir()
        try:
            _product_vs_oracle(d, spec, argv, sp, sn, rmt_text=rmt)
            ran += 1
        except (ValueError, KeyError):
            # the product raised the reference's exception before the oracle was consulted: the oracle must raise too
            with pytest.raises((ValueError, KeyError)):
                _oracle_only(d, spec, argv, sp, sn, rmt)
    assert ran >= 4


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_round3_engine_shapes_vs_oracle_fuzz(seed, tmp_path):
    """The same against the shapes the round-3 engines took over: translocations (SV-mix engine + __link_tls), five length
    widths (wide accept tables), SV std lines over gene blocks with per-range SV settings in token order, `sn_block` above
    the minimum block (host-chain engine) -- whole CLI vs the ORACLE."""
    rs = np.random.RandomState(seed)
    ran = 0
    for it in range(8):
        lengths = [int(rs.choice([300_000, 700_000, 1_500_000, 4_000_000]) + rs.randint(0, 5000))
                   for _ in range(int(rs.randint(1, 4)))]
        if rs.rand() < 0.2:
            lengths.append(int(rs.randint(1, 3000)))
        mode = rs.choice(["args", "rmt", "rmt"])
        argv, rmt = args_settings_round3(rs) if mode == "args" else rmt_settings_round3(rs, lengths)
        spec = {"contigs": [{"defline": f"g{it}_{i} fuzz", "length": L, "bpl": int(rs.choice([50, 60, 61, 80])),
                             "seed": 20_000 * seed + 10 * it + i} for i, L in enumerate(lengths)]}
        sp, sn = int(rs.randint(0, 1 << 30)), int(rs.randint(0, 1 << 30))
        d = tmp_path / f"it{it}"
        d.mkdir()
        try:
            _product_vs_oracle(d, spec, argv, sp, sn, rmt_text=rmt)
            ran += 1
        except (ValueError, KeyError):
            with pytest.raises((ValueError, KeyError)):
                _oracle_only(d, spec, argv, sp, sn, rmt)
    assert ran >= 4


def _oracle_only(tmp_path, spec, argv_tail, seed_py, seed_np, rmt_text):
    import contextlib
    import io

    import inputs as gin
    import mutation_simulator_amd as msa
    from helpers import parse_fasta_bytes
    from oracle import oracle as orc
    from test_host_settings import dump_sim
    infile = gin.write_input(spec, tmp_path / "in2.fa")
    tail = list(argv_tail)
    if rmt_text is not None:
        (tmp_path / "c2.rmt").write_text(rmt_text)
        tail = ["rmt", str(tmp_path / "c2.rmt")]
    argv = ["-q", "-o", str(tmp_path / "out2"), str(infile)] + tail
    with contextlib.redirect_stderr(io.StringIO()):
        args = msa.get_args(argv)
        fasta = msa.load_fasta(args.infile)
        sim = (msa.SimulationSettings.from_args(args, fasta, True) if args.mode == "args"
               else msa.SimulationSettings.from_rmt(args.rmtfile, fasta, True))
    o = orc.Oracle()
    o.seed(seed_py, seed_np)
    o.run_genome(parse_fasta_bytes(infile.read_bytes()), dump_sim(sim), infile.name)
