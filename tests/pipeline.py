"""Drives the PRODUCT (mutation-simulator_amd host package + libmsim) for golden cases.

``run_product_case``: whole CLI path (argv -> settings -> Mutator on the GPU -> files).
``plan_only_vcf``   : host-only context (no GPU): settings -> msim_plan_contig -> rendered VCF.
"""
from __future__ import annotations

import contextlib
import io
import random
from pathlib import Path

import numpy as np

import mutation_simulator_amd as msa
from mutation_simulator_amd import _ffi
from mutation_simulator_amd import __main__ as msa_main
from mutation_simulator_amd import mutator as msa_mutator

from helpers import CASES, case_input_bytes, mask_vcf


def prepare(meta: dict, tmp: Path):
    infile = tmp / meta["infile_name"]
    infile.write_bytes(case_input_bytes(meta))
    tail = list(meta["argv_tail"])
    if tail[:1] == ["rmt"]:
        rmt = tmp / "case.rmt"
        rmt.write_text((CASES / meta["name"] / "case.rmt").read_text())
        tail = ["rmt", str(rmt)]
    return ["-o", str(tmp / "out"), str(infile)] + tail


def build_settings(argv):
    args = msa.get_args(argv)
    fasta = msa.load_fasta(args.infile)
    if args.mode == "args":
        sim = msa.SimulationSettings.from_args(args, fasta, args.ignore_warnings)
    elif args.mode == "it":
        sim = msa.SimulationSettings.from_it(args.interchromosomalrate, fasta, args.ignore_warnings)
    else:
        sim = msa.SimulationSettings.from_rmt(args.rmtfile, fasta, args.ignore_warnings)
    return args, fasta, sim


def plan_only_vcf(meta: dict, tmp: Path):
    """Returns (vcf_bytes_masked, empty_contigs, engine) using a host-only libmsim context."""
    argv = prepare(meta, tmp)
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        args, fasta, sim = build_settings(argv)
    random.seed(meta["seed_py"])
    np.random.seed(meta["seed_np"])
    eng = _ffi.Engine(device=-1)
    msa_mutator.export_python_streams(eng)
    eng.set_params(msa_mutator.params_descriptor(sim))
    vw = msa.VcfWriter(tmp / "plan_only.vcf")
    vw.write_header(args.infile.name, fasta, sim.assembly_name, sim.species_name, sim.sample_name)
    empty = []
    for chrom in sim.chromosomes:
        rec = fasta[chrom.number]
        cid = eng.add_contig(rec.bases)
        eng.plan_contig(cid, msa_mutator.plan_descriptors(chrom))
        if eng.plan_was_empty(cid):
            empty.append(chrom.number)
        recs, pool = eng.fetch_records(cid)
        vw.write_raw(_ffi.render_vcf(recs, pool, rec.bases, rec.name))
        eng.clear()
    vw.close()
    msa_mutator.import_python_streams(eng)
    return mask_vcf((tmp / "plan_only.vcf").read_bytes()), empty, eng


def run_product_case(meta: dict, tmp: Path, extra_argv=()):
    """Whole CLI on the GPU.  Returns dict(exit_code, exception, stdout, stderr, fasta, vcf).  ``extra_argv``: options of this
    package's own (``--bench-json path``, ``--rng fast`` ...) in front of the reference's argv."""
    argv = list(extra_argv) + prepare(meta, tmp)
    out, err = io.StringIO(), io.StringIO()
    code, exc = None, None
    random.seed(meta["seed_py"])
    np.random.seed(meta["seed_np"])
    try:
        with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
            try:
                msa_main.main(argv)
            except SystemExit as e:
                code = e.code
    except BaseException as e:  # noqa: BLE001
        exc = e
    res = {"exit_code": code, "exception": exc, "stdout": out.getvalue(), "stderr": err.getvalue(),
           "fasta": None, "vcf": None, "it_fasta": None, "bedpe": None}
    suffix = Path(meta["infile_name"]).suffix
    fa, vcf = tmp / f"out_ms{suffix}", tmp / "out_ms.vcf"
    if fa.exists():
        res["fasta"] = fa.read_bytes()
    if vcf.exists():
        res["vcf"] = mask_vcf(vcf.read_bytes())
    it_fa, bedpe = tmp / f"out_ms_it{suffix}", tmp / "out_ms_it.bedpe"      # the interchromosomal-translocation pass
    if it_fa.exists():
        res["it_fasta"] = it_fa.read_bytes()
    if bedpe.exists():
        res["bedpe"] = bedpe.read_bytes()
    return res
