#!/usr/bin/env python3
"""Golden-vector generator: runs the REAL reference (imported from /root/reference) under fixed
seeds and records what it produced.  Runs ONLY in the build container (the reference never
travels to the GPU box); its outputs under tests/golden/ are committed as data fixtures.

    python tests/golden/make_goldens.py            # regenerate everything

What gets pinned:
  rng_kat.json        raw MT19937 words for both seedings, randbelow/sample/randint/uniform,
                      numpy choice(p) / choice("ATGC"), sample_with_minimum_distance
  settings.json       argv / RMT text -> the settings tree the reference derives from it
  plan.json           Mutator._Mutator__get_mutations outputs + RNG stream positions after them
  apply.json          Mutator._Mutator__mutate_sequence on hand-built mutation dicts (edge cases)
  errors.json         exception type + message / exit code + stderr for bad inputs
  cases/<name>/       whole-CLI runs: meta.json (+ input.fa, expected .fa / .vcf for small cases,
                      SHA-256 + head/tail for the 1 Mb ones)

The reference has no seed flag; seeding is ``random.seed(s); numpy.random.seed(t)`` before
``main()``.  ``pyfaidx`` is absent from this image: ``pyfaidx_standin.py`` (ours, sequence access
only) is injected as ``sys.modules['pyfaidx']``.  The ``##filedate=`` VCF line is wall-clock and is
masked to ``##filedate=MASKED``.
"""
from __future__ import annotations

import contextlib
import hashlib
import io
import json
import os
import platform
import random
import shutil
import sys
import tempfile
from argparse import Namespace
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import inputs as gin  # noqa: E402
import pyfaidx_standin  # noqa: E402

REFERENCE_ROOT = Path("/root/reference")
sys.modules["pyfaidx"] = pyfaidx_standin
sys.path.insert(0, str(REFERENCE_ROOT))
import mutation_simulator as ref  # noqa: E402
from mutation_simulator import __main__ as ref_main  # noqa: E402
from mutation_simulator import util as ref_util  # noqa: E402
from mutation_simulator.mutator import Mutation, Mutator  # noqa: E402
from mutation_simulator.rmt import SimulationSettings  # noqa: E402

MutType = ref.MutType
ENV = {
    "python": platform.python_version(),
    "numpy": np.__version__,
    "reference_version": ref.__version__,
    "pyfaidx": "stand-in (tests/golden/pyfaidx_standin.py)",
}

C3_FLAGS = ("-sn 0.005 -in 0.001 -inmin 1 -inmax 50 -de 0.001 -demin 1 -demax 50 "
            "-du 0.0005 -dumin 50 -dumax 500 -iv 0.0005 -ivmin 50 -ivmax 500").split()


# ----------------------------------------------------------------------------- helpers
def sha256(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def mask_vcf(b: bytes) -> bytes:
    out = []
    for line in b.split(b"\n"):
        if line.startswith(b"##filedate="):
            line = b"##filedate=MASKED"
        out.append(line)
    return b"\n".join(out)


def py_next_words(n=4):
    st = random.getstate()
    w = [random.getrandbits(32) for _ in range(n)]
    random.setstate(st)
    return w


def np_next_words(n=4):
    st = np.random.get_state()
    w = [int(x) for x in np.random.randint(0, 4294967296, size=n, dtype=np.uint32)]
    np.random.set_state(st)
    return w


def py_words_consumed(state_before, limit=50_000_000):
    """Number of 32-bit words the global ``random`` consumed since ``state_before``."""
    target = random.getstate()
    probe = random.Random()
    probe.setstate(state_before)
    # compare by the next 6 outputs; advance one word at a time
    want = py_next_words(6)
    window = [probe.getrandbits(32) for _ in range(6)]
    n = 0
    while window != want:
        window.pop(0)
        window.append(probe.getrandbits(32))
        n += 1
        if n > limit:
            raise RuntimeError("could not locate stream position")
    random.setstate(target)
    return n


def np_words_consumed(state_before, limit=50_000_000):
    target = np.random.get_state()
    want = np_next_words(6)
    probe = np.random.RandomState()
    probe.set_state(state_before)
    window = [int(x) for x in probe.randint(0, 4294967296, size=6, dtype=np.uint32)]
    n = 0
    while window != want:
        window.pop(0)
        window.append(int(probe.randint(0, 4294967296, dtype=np.uint32)))
        n += 1
        if n > limit:
            raise RuntimeError("could not locate stream position")
    np.random.set_state(target)
    return n


def run_reference_cli(argv: list[str], seed_py: int, seed_np: int):
    """Run the reference's ``main()``; returns (exit_code|None, stdout, stderr, exception)."""
    old_argv = sys.argv
    sys.argv = ["mutation-simulator"] + argv
    out, err = io.StringIO(), io.StringIO()
    code, exc = None, None
    random.seed(seed_py)
    np.random.seed(seed_np)
    try:
        with contextlib.redirect_stdout(out), _capture_ref_stderr(err):
            try:
                ref_main.main()
            except SystemExit as e:  # exit_with_error -> exit(1); argparse -> exit(2)
                code = e.code
    except BaseException as e:  # uncaught traceback in the reference (ValueError, KeyError ...)
        exc = e
    finally:
        sys.argv = old_argv
    return code, out.getvalue(), err.getvalue(), exc


@contextlib.contextmanager
def _capture_ref_stderr(buf):
    """The reference binds ``stderr`` at import time (``from sys import stderr``): patch those
    module globals as well as ``sys.stderr``."""
    from mutation_simulator import mutator as ref_mutator
    saved = (ref_util.stderr, ref_mutator.stderr)
    ref_util.stderr = buf
    ref_mutator.stderr = buf
    try:
        with contextlib.redirect_stderr(buf):
            yield
    finally:
        ref_util.stderr, ref_mutator.stderr = saved


def mask_runtime(s: str) -> str:
    import re
    return re.sub(r"finished in: [0-9.]+s", "finished in: MASKEDs", s)


# ----------------------------------------------------------------------------- CLI cases
def cli_case(name: str, spec: dict, argv_tail: list[str], seed_py: int, seed_np: int,
             store: str = "full", rmt_text: str | None = None, infile_name: str = "input.fa",
             notes: str = ""):
    """Run one whole-CLI case in a scratch dir and write its fixture directory."""
    case_dir = HERE / "cases" / name
    if case_dir.exists():
        shutil.rmtree(case_dir)
    case_dir.mkdir(parents=True)
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        infile = td / infile_name
        gin.write_input(spec, infile)
        argv_mode = list(argv_tail)
        if rmt_text is not None:
            (td / "case.rmt").write_text(rmt_text)
            argv_mode = ["rmt", str(td / "case.rmt")]
        argv = ["-o", str(td / "out"), str(infile)] + argv_mode
        code, so, se, exc = run_reference_cli(argv, seed_py, seed_np)
        stem = Path(infile_name).suffix
        out_fa = td / f"out_ms{stem}"
        out_vcf = td / "out_ms.vcf"
        meta = {
            "name": name, "env": ENV, "notes": notes,
            "input_spec": spec, "infile_name": infile_name,
            "argv_tail": argv_tail if rmt_text is None else ["rmt", "case.rmt"],
            "seed_py": seed_py, "seed_np": seed_np, "store": store,
            "exit_code": code,
            "exception": None if exc is None else {"type": type(exc).__name__, "message": str(exc),
                                                   "repr_args": [repr(a) for a in exc.args]},
            "stdout": mask_runtime(so), "stderr": se.replace(str(td), "<TMP>"),
            "input_sha256": sha256(infile.read_bytes()),
            "py_next_words_after": py_next_words(), "np_next_words_after": np_next_words(),
        }
        if rmt_text is not None:
            (case_dir / "case.rmt").write_text(rmt_text)
        if code is None:
            # the settings tree the reference derived for this run (rebuilt; RNG-free)
            old = sys.argv
            sys.argv = ["mutation-simulator"] + argv
            try:
                with _capture_ref_stderr(io.StringIO()):
                    a2 = ref.get_args()
                    f2 = ref_util.load_fasta(a2.infile)
                    sim2 = (SimulationSettings.from_args(a2, f2, True) if a2.mode == "args"
                            else SimulationSettings.from_it(a2.interchromosomalrate, f2, True) if a2.mode == "it"
                            else SimulationSettings.from_rmt(a2.rmtfile, f2, True))
                meta["sim"] = dump_sim(sim2)
                meta["contigs"] = [{"name": f2[k].name, "long_name": f2[k].long_name,
                                    "length": len(f2[k]), "lenc": f2.faidx.index[k].lenc}
                                   for k in f2.keys()]
            finally:
                sys.argv = old
        it_fa, it_bedpe = td / f"out_ms_it{stem}", td / "out_ms_it.bedpe"
        if exc is None and code is None and it_fa.exists():
            # the interchromosomal-translocation pass (it_mutator.py): its Fasta and its BEDPE
            fa_it, bedpe = it_fa.read_bytes(), it_bedpe.read_bytes()
            meta.update({"it_fasta_sha256": sha256(fa_it), "it_fasta_len": len(fa_it),
                         "bedpe_sha256": sha256(bedpe), "bedpe_len": len(bedpe),
                         "bedpe_head": bedpe[:400].decode(), "it_fasta_head": fa_it[:200].decode()})
            if store == "full":
                (case_dir / "expected_ms_it.fa").write_bytes(fa_it)
                (case_dir / "expected_ms_it.bedpe").write_bytes(bedpe)
                if not out_fa.exists():
                    (case_dir / "input.fa").write_bytes(infile.read_bytes())
        if exc is None and code is None and out_fa.exists():
            fa = out_fa.read_bytes()
            vcf = mask_vcf(out_vcf.read_bytes())
            vcf_lines = vcf.split(b"\n")
            body = [l for l in vcf_lines if l and not l.startswith(b"#")]
            meta.update({
                "fasta_sha256": sha256(fa), "fasta_len": len(fa),
                "vcf_sha256": sha256(vcf), "vcf_len": len(vcf), "vcf_records": len(body),
                "vcf_head": [l.decode() for l in body[:8]],
                "vcf_tail": [l.decode() for l in body[-4:]],
                "fasta_head": fa[:200].decode(), "fasta_tail": fa[-120:].decode(),
            })
            counts = {}
            for l in body:
                info = l.split(b"\t")[7]
                key = "SNP" if info == b"." else info.split(b";")[0].split(b"=")[1].decode()
                counts[key] = counts.get(key, 0) + 1
            meta["vcf_type_counts"] = counts
            if store == "full":
                (case_dir / "input.fa").write_bytes(infile.read_bytes())
                (case_dir / "expected_ms.fa").write_bytes(fa)
                (case_dir / "expected_ms.vcf").write_bytes(vcf)
        (case_dir / "meta.json").write_text(json.dumps(meta, indent=1) + "\n")
    status = "ok" if exc is None and code is None else f"exit={code} exc={type(exc).__name__ if exc else None}"
    print(f"  case {name:32s} {status}")
    return meta


RMT_SMALL = """\
# small RMT exercising: meta block values, titv, None blocks, a hot pool-path range,
# cold ranges, SV types inside ranges (token order), END, an unlisted contig (std only)
fasta = Input.FA
species_name = Testus Maximus
assembly_name = TM1
sample_name = S1
titv = 2.5
sn_block = 3
in_block = 0
de_block = 2

std
it None
sn 0.01

chr 1
1-1000 None
2001-3000 du 0.001 dumin 5 dumax 9 sn 0.2 in 0.01 inmin 1 inmax 3
5001-9000 sn 0.001
12001-20000 sn 0.05 de 0.002 demin 2 demax 30 iv 0.001 ivmin 4 ivmax 40
30001-END None
chr 3
101-200 None
501-40000 in 0.004 inmin 2 inmax 12 du 0.002 dumin 10 dumax 60 sn 0.004
"""


def make_cli_cases():
    print("CLI cases")
    one_mb = {"contigs": [{"defline": "contig1 synthetic 1Mb", "length": 1_000_000, "bpl": 60,
                           "seed": 1234}]}
    cli_case("c1_snp_1mb", one_mb, ["args", "-sn", "0.01"], 42, 42, store="hash",
             notes="BASELINE config 1")
    two = {"contigs": [
        {"defline": "chrA first contig", "length": 120_000, "bpl": 60, "seed": 11},
        {"defline": "chrB", "length": 80_037, "bpl": 70, "seed": 12}]}
    cli_case("snp_titv2_2ctg", two, ["args", "-sn", "0.01", "-titv", "2.0"], 7, 7,
             notes="BASELINE config 2 shape at 200 kb")
    sv2 = {"contigs": [
        {"defline": "sv1 150k", "length": 150_000, "bpl": 60, "seed": 21},
        {"defline": "sv2", "length": 50_000, "bpl": 80, "seed": 22}]}
    cli_case("svmix_2ctg_200k", sv2, ["args"] + C3_FLAGS, 42, 42,
             notes="BASELINE config 3 flags at 200 kb")
    cli_case("svmix_1mb", one_mb, ["args"] + C3_FLAGS, 42, 42, store="hash",
             notes="BASELINE config 3 flags at 1 Mb")
    iu = {"contigs": [
        {"defline": "iu1 with N runs and IUPAC", "length": 60_000, "bpl": 60, "seed": 31,
         "decorate": True},
        {"defline": "iu2", "length": 30_011, "bpl": 50, "seed": 32, "decorate": True}]}
    cli_case("svmix_iupac", iu,
             ["args", "-sn", "0.02", "-titv", "0.5", "-in", "0.004", "-inmin", "1", "-inmax",
              "20", "-de", "0.004", "-demin", "1", "-demax", "25", "-du", "0.002", "-dumin", "3",
              "-dumax", "40", "-iv", "0.002", "-ivmin", "2", "-ivmax", "35"], 3, 5,
             notes="N runs, IUPAC codes, lower-case input, different seeds per stream")
    cli_case("blocks_nondefault", two,
             ["args", "-sn", "0.03", "-snb", "4", "-in", "0.01", "-inmax", "6", "-inb", "7",
              "-de", "0.01", "-demax", "9", "-deb", "3", "-iv", "0.004", "-ivmax", "12",
              "-ivb", "2", "-du", "0.004", "-dumax", "15", "-dub", "5", "-tlb", "2",
              "-a", "asmX", "-s", "Homo sapiens", "-n", "sampleZ"], 99, 100,
             notes="non-default block values: min block 2 spacing, SNPs blocked by earlier muts")
    tiny = {"contigs": [
        {"defline": "t1 single line", "length": 37, "bpl": 60, "seed": 41},
        {"defline": "t2 exact multiple", "length": 120, "bpl": 60, "seed": 42},
        {"defline": "t3", "length": 1, "bpl": 60, "seed": 43},
        {"defline": "t4 longer", "length": 1234, "bpl": 61, "seed": 44}]}
    cli_case("tiny_contigs", tiny, ["args", "-sn", "0.05", "-in", "0.02", "-de", "0.02"], 5, 6,
             notes="contigs shorter than a line, exact multiple of bpl, 1 base, no-mutation warning")
    hundred = {"contigs": [{"defline": "r1", "length": 100_000, "bpl": 60, "seed": 51}]}
    cli_case("readme_mix_no_tl", hundred,
             ["args", "-sn", "0.01", "-in", "0.01", "-de", "0.01", "-du", "0.01", "-iv", "0.01"],
             1, 2, notes="README perf flags minus -tl (default lengths 1-2, iv 2-3)")
    cli_case("readme_mix_tl", hundred,
             ["args", "-sn", "0.01", "-in", "0.01", "-de", "0.01", "-du", "0.01", "-iv", "0.01",
              "-tl", "0.01"], 1, 2,
             notes="README perf flags incl. translocations (SURVEY 8(f)3 'next' row)")
    cli_case("titv0_dense", hundred, ["args", "-sn", "0.2", "-titv", "0"], 8, 9,
             notes="titv 0 -> transitions only when m == 0; dense SNPs")
    rmt_in = {"contigs": [
        {"defline": "c1 rmt target", "length": 45_000, "bpl": 60, "seed": 61},
        {"defline": "c2 std only", "length": 20_000, "bpl": 60, "seed": 62},
        {"defline": "c3", "length": 40_000, "bpl": 75, "seed": 63}]}
    cli_case("rmt_small", rmt_in, [], 42, 43, rmt_text=RMT_SMALL, infile_name="input.fa",
             notes="RMT mode: blocked/hot(pool path)/cold ranges, block meta, END, token order")
    # --- failures that end in an uncaught exception or exit(1) in the reference
    cli_case("err_rmt_overlap", rmt_in, [], 42, 42, notes="overlapping ranges -> ValueError",
             rmt_text="std\nit None\nsn 0.01\n\nchr 1\n1001-5000 None\n3000-3500 None\n")
    cli_case("err_rates_too_high", hundred, ["args", "-sn", "0.3", "-in", "0.3"], 1, 1)
    cli_case("err_rate_above_one", hundred, ["args", "-sn", "1.5"], 1, 1)
    cli_case("err_rates_zero", hundred, ["args"], 1, 1)
    cli_case("err_rate_negative", hundred, ["args", "-sn", "0.1", "-de", "-0.01"], 1, 1)
    cli_case("err_iv_min", hundred, ["args", "-iv", "0.01", "-ivmin", "1"], 1, 1)
    cli_case("err_min_gt_max", hundred, ["args", "-de", "0.01", "-demin", "9", "-demax", "3"], 1, 1)
    cli_case("err_in_min_zero", hundred, ["args", "-sn", "0.01", "-inmin", "0"], 1, 1,
             notes="every non-SN type present in the dict is validated even at rate 0")
    cli_case("err_titv_negative", hundred, ["args", "-sn", "0.01", "-titv", "-1"], 1, 1)
    cli_case("err_rmt_no_std", rmt_in, [], 1, 1, rmt_text="titv = 1\nchr 1\n1-100 None\n")
    cli_case("err_rmt_bad_float", rmt_in, [], 1, 1,
             rmt_text="std\nit None\nsn abc\n")
    cli_case("err_rmt_chrom_missing", rmt_in, [], 1, 1,
             rmt_text="std\nit None\nsn 0.01\nchr 7\n1-100 None\n")
    cli_case("err_rmt_out_of_bounds", rmt_in, [], 1, 1,
             rmt_text="std\nit None\nsn 0.01\nchr 2\n1-100 None\n500-20002 None\n")
    cli_case("err_rmt_missing_len", rmt_in, [], 1, 1,
             rmt_text="std\nit None\nsn 0.01\nchr 2\n1-100 in 0.01\n")
    cli_case("err_dup_header", {"contigs": [
        {"defline": "same x", "length": 100, "bpl": 60, "seed": 1},
        {"defline": "same y", "length": 100, "bpl": 60, "seed": 2}]}, ["args", "-sn", "0.01"], 1, 1)
    cli_case("err_snp_on_U", {"contigs": [
        {"defline": "rna", "length": 64, "bpl": 60, "seed": 1, "literal": "ACGU" * 16}]},
        ["args", "-sn", "0.4", "-titv", "0"], 1, 1,
        notes="transversion on a base outside AGTCN -> uncaught KeyError")
    cli_case("warn_rmt_meta", rmt_in, [], 4, 4,
             rmt_text="fasta = other.fa\nmd5 = 00ff\ntl_block = -3\nstd\nit None\nsn 0.002\n",
             notes="meta mismatch warnings + block clamp warning; values are lower-cased")
    tl_in = {"contigs": [
        {"defline": "tl1 translocations", "length": 90_000, "bpl": 60, "seed": 81, "decorate": True},
        {"defline": "tl2", "length": 151, "bpl": 50, "seed": 82},
        {"defline": "tl3", "length": 30_000, "bpl": 70, "seed": 83}]}
    cli_case("tl_heavy", tl_in,
             ["args", "-tl", "0.03", "-tlmin", "5", "-tlmax", "60", "-tlb", "3", "-sn", "0.01", "-in", "0.002",
              "-inmax", "5", "-de", "0.002", "-demax", "9", "-iv", "0.002", "-ivmax", "30", "-titv", "1.3"], 12, 13,
             notes="translocation-heavy mix: linked TL/TLI pairs, reversed and forward copies, IUPAC inside spans, "
                   "a tiny contig where TL/TLI counts differ (fix_tl_amount) or TLIs stay unlinked")
    cli_case("tl_rmt", tl_in, [], 21, 22,
             rmt_text="tl_block = 2\nstd\nit None\ntl 0.01 tlmin 2 tlmax 40\nchr 1\n1-30000 sn 0.02 tl 0.04 tlmin 10 tlmax 200\n"
                      "50001-60000 None\nchr 2\n1-151 tl 0.2 tlmin 1 tlmax 3\n",
             notes="RMT with translocations: token-order chances, TL pairs linked across ranges of one contig")
    cli_case("rmt_quiet_none_std", rmt_in, [], 4, 4,
             rmt_text="std\nit None\nNone\nchr 2\n11-5000 sn 0.01 IN 0.002 INMIN 1 INMAX 4\n",
             notes="std None: only the listed range mutates; upper-case keywords")


# ----------------------------------------------------------------------------- RNG KATs
def make_rng_kat():
    print("RNG KATs")
    kat: dict = {"env": ENV}
    kat["py_seed_words"] = {}
    for s in [0, 1, 42, 2**31, 2**32 - 1, 2**32, 2**40 + 5, 123456789012345678901234567890]:
        random.seed(s)
        kat["py_seed_words"][str(s)] = [random.getrandbits(32) for _ in range(8)]
    kat["np_seed_words"] = {}
    for s in [0, 1, 42, 2**31, 2**32 - 1]:
        np.random.seed(s)
        kat["np_seed_words"][str(s)] = [int(x) for x in
                                        np.random.randint(0, 4294967296, size=8, dtype=np.uint32)]
    # state hand-off: raw 624-word state + position after N draws
    random.seed(42)
    [random.getrandbits(32) for _ in range(1000)]
    st = random.getstate()
    kat["py_state_after_1000"] = {"pos": st[1][-1], "crc_first8": list(st[1][:8]),
                                  "next": py_next_words(4)}
    # randbelow via randrange / randint
    rb = []
    for seed, n, cnt in [(1, 1, 5), (1, 2, 12), (2, 3, 12), (3, 50, 20), (4, 451, 20),
                         (5, 990000, 10), (6, 2**28 - 3, 10), (7, 2**32 - 1, 6),
                         (8, 2**32, 6), (9, 2**33 + 17, 6), (10, 2**64 + 1, 4)]:
        random.seed(seed)
        st0 = random.getstate()
        vals = [random.randrange(n) for _ in range(cnt)]
        rb.append({"seed": seed, "n": n, "values": vals, "words": py_words_consumed(st0),
                   "next": py_next_words(2)})
    kat["randbelow"] = rb
    ri = []
    for seed, a, b, cnt in [(11, 0, 1, 16), (12, 5, 5, 4), (13, 100, 149, 10), (14, 49, 499, 10)]:
        random.seed(seed)
        st0 = random.getstate()
        vals = [random.randint(a, b) for _ in range(cnt)]
        ri.append({"seed": seed, "a": a, "b": b, "values": vals, "words": py_words_consumed(st0)})
    kat["randint"] = ri
    random.seed(21)
    kat["uniform"] = {"seed": 21, "values_hex": [random.uniform(0, 1).hex() for _ in range(8)]}
    # sample(range(n), k): set path and pool path
    smp = []
    for seed, n, k in [(31, 100, 0), (31, 10, 3), (32, 21, 5), (33, 22, 5), (34, 85, 6), (35, 86, 6),
                       (36, 789, 211), (37, 1045, 211), (38, 1046, 211), (39, 5000, 40),
                       (40, 990000, 10000), (41, 17, 17), (42, 300, 299), (43, 65557, 10000),
                       (44, 65558, 10000)]:
        random.seed(seed)
        st0 = random.getstate()
        vals = random.sample(range(n), k)
        from math import ceil, log
        setsize = 21 + (4 ** ceil(log(k * 3, 4)) if k > 5 else 0)
        entry = {"seed": seed, "n": n, "k": k, "setsize": setsize,
                 "path": "pool" if n <= setsize else "set",
                 "words": py_words_consumed(st0), "next": py_next_words(2),
                 "sha256_sorted": sha256(np.sort(np.array(vals, dtype=np.int64)).tobytes()),
                 "sha256_order": sha256(np.array(vals, dtype=np.int64).tobytes())}
        if k <= 300:
            entry["values"] = vals
        else:
            entry["first"] = vals[:8]
        smp.append(entry)
    kat["sample"] = smp
    swmd = []
    for seed, start, stop, k, d in [(3, 0, 99, 5, 3), (42, 0, 999_999, 10_000, 1),
                                    (5, 2000, 2999, 211, 1), (6, 500, 1499, 40, 4),
                                    (7, 0, 36, 1, 1), (8, 10, 10, 0, 1), (9, 0, 59, 20, 2)]:
        random.seed(seed)
        st0 = random.getstate()
        vals = ref_util.sample_with_minimum_distance(start, stop, k, d)
        e = {"seed": seed, "start": start, "stop": stop, "k": k, "d": d,
             "words": py_words_consumed(st0),
             "sha256": sha256(np.array(vals, dtype=np.int64).tobytes()),
             "first": vals[:8], "last": vals[-2:]}
        if k <= 300:
            e["values"] = vals
        swmd.append(e)
    kat["sample_with_minimum_distance"] = swmd
    # numpy choice with p
    ch = []
    for seed, p, size in [(1, [1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], 12),
                          (2, [0.625, 0.125, 0.125, 0.0625, 0.0625, 0.0, 0.0], 40),
                          (3, [0.2, 0.3, 0.5], 40), (4, [0.0, 0.0, 1.0], 6),
                          (5, [1 / 3, 1 / 3, 1 / 3], 40), (6, [0.1] * 10, 40)]:
        np.random.seed(seed)
        st0 = np.random.get_state()
        from numpy.random import choice
        idx = [int(x) for x in choice(list(range(len(p))), p=p, size=size)]
        cdf = np.cumsum(np.array(p, dtype=np.float64))
        cdf /= cdf[-1]
        ch.append({"seed": seed, "p_hex": [float(x).hex() for x in p], "size": size, "idx": idx,
                   "cdf_hex": [float(x).hex() for x in cdf],
                   "words": np_words_consumed(st0)})
    kat["np_choice_p"] = ch
    np.random.seed(77)
    st0 = np.random.get_state()
    from numpy.random import choice
    s = "".join(choice(["A", "T", "G", "C"], 64))
    kat["np_choice_atgc"] = {"seed": 77, "size": 64, "value": s, "words": np_words_consumed(st0)}
    (HERE / "rng_kat.json").write_text(json.dumps(kat, indent=1) + "\n")


# ----------------------------------------------------------------------------- settings goldens
def dump_sim(sim) -> dict:
    def ms(m):
        if m.mut_rates is None:
            return None
        return {"rates": [[t.name, r] for t, r in m.mut_rates.items()],
                "rates_hex": [[t.name, float(r).hex()] for t, r in m.mut_rates.items()],
                "chances_hex": [[t.name, float(c).hex()] for t, c in m.mut_chances.items()],
                "rate_sum_hex": float(sum(m.mut_rates.values())).hex(),
                "min": {t.name: v for t, v in m.mut_lengs["min"].items()} if m.mut_lengs else None,
                "max": {t.name: v for t, v in m.mut_lengs["max"].items()} if m.mut_lengs else None,
                "has_mutations": m.has_mutations}
    return {
        "mut_block": [[t.name, v] for t, v in sim.mut_block.items()],
        "titv": sim.titv, "fasta": sim.fasta if sim.fasta is None else str(sim.fasta),
        "md5": sim.md5, "species_name": sim.species_name, "assembly_name": sim.assembly_name,
        "sample_name": sim.sample_name, "has_mutations": sim.has_mutations, "has_it": sim.has_it,
        "chromosomes": [{"number": c.number, "it_rate": c.it_rate,
                         "ranges": [{"start": r.start, "stop": r.stop,
                                     "settings": ms(r.mutation_settings)}
                                    for r in c.range_definitions]} for c in sim.chromosomes],
    }


def make_settings():
    print("settings goldens")
    out = {"env": ENV, "cases": []}
    spec = {"contigs": [
        {"defline": "c1 x", "length": 45_000, "bpl": 60, "seed": 61},
        {"defline": "c2", "length": 20_000, "bpl": 60, "seed": 62},
        {"defline": "c3", "length": 40_000, "bpl": 75, "seed": 63}]}
    rmts = {
        "rmt_small": RMT_SMALL,
        "rmt_end_only_last": "std\nit None\nsn 0.01\nchr 2\n1-100 None\n201-END sn 0.02\n",
        "rmt_unsorted": "std\nit None\nsn 0.01\nchr 3\n5001-6000 None\n1-1000 sn 0.1\nchr 1\n7-9 None\n",
        "rmt_tl": "tl_block = 4\nstd\nit 0.1\nsn 0.01 tl 0.02 tlmin 3 tlmax 30\nchr 2\nit None\n1-50 None\n",
        "rmt_full_cover": "std\nit None\nNone\nchr 2\n1-20000 sn 0.004\n",
        "rmt_comments": "# c\n\n  titv = 3 # trailing\nunknown_key = 5\nstd # s\nit None\nsn 0.01 foo 3\n",
        "rmt_blocks_zero": "sn_block=0\nin_block = 5\nstd\nit None\nsn 0.01\n",
    }
    argvs = {
        "args_c2": ["args", "-sn", "0.01", "-titv", "2.0"],
        "args_c3": ["args"] + C3_FLAGS,
        "args_blocks": ["args", "-sn", "0.03", "-snb", "4", "-inb", "0", "-deb", "-2", "-tl", "0.02",
                        "-tlmin", "5", "-tlmax", "9"],
        "args_sum_half": ["args", "-sn", "0.25", "-in", "0.25"],
    }
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        infile = gin.write_input(spec, td / "Input.FA")
        fasta = ref_util.load_fasta(infile)
        for name, text in rmts.items():
            p = td / f"{name}.rmt"
            p.write_text(text)
            err = io.StringIO()
            entry = {"name": name, "kind": "rmt", "rmt_text": text, "input_spec": spec}
            try:
                with _capture_ref_stderr(err):
                    sim = SimulationSettings.from_rmt(p, fasta, False)
                entry["sim"] = dump_sim(sim)
            except Exception as e:  # noqa: BLE001
                entry["exception"] = {"type": type(e).__name__,
                                      "message": str(e).replace(str(td), "<TMP>")}
            entry["stderr"] = err.getvalue()
            out["cases"].append(entry)
        for name, argv in argvs.items():
            old = sys.argv
            sys.argv = ["mutation-simulator", str(infile)] + argv
            err = io.StringIO()
            entry = {"name": name, "kind": "args", "argv_tail": argv, "input_spec": spec}
            try:
                with _capture_ref_stderr(err):
                    args = ref.get_args()
                    sim = SimulationSettings.from_args(args, fasta, args.ignore_warnings)
                entry["sim"] = dump_sim(sim)
                entry["outfasta"] = str(args.outfasta)
                entry["outvcf"] = str(args.outvcf)
            except Exception as e:  # noqa: BLE001
                entry["exception"] = {"type": type(e).__name__, "message": str(e)}
            finally:
                sys.argv = old
            entry["stderr"] = err.getvalue()
            out["cases"].append(entry)
        # output naming (argument_parser.add_outfile_names)
        naming = []
        for o, inf in [(None, "genome.fa"), ("res", "genome.fasta"), ("dir/sub/base", "g.fa.gz"),
                       ("x.y", "in.fna"), (".", "a.b.fa")]:
            old = sys.argv
            sys.argv = ["mutation-simulator"] + ([] if o is None else ["-o", o]) + [inf, "args"]
            try:
                a = ref.get_args()
                naming.append({"o": o, "infile": inf, "outbase": str(a.outbase),
                               "outfasta": str(a.outfasta), "outvcf": str(a.outvcf)})
            finally:
                sys.argv = old
        out["naming"] = naming
    (HERE / "settings.json").write_text(json.dumps(out, indent=1) + "\n")


# ----------------------------------------------------------------------------- plan / apply goldens
class _Scratch:
    """A Mutator wired to scratch output files (the reference opens writers in __init__)."""

    def __init__(self, spec, argv_tail, td: Path):
        self.td = td
        self.infile = gin.write_input(spec, td / "input.fa")
        old = sys.argv
        sys.argv = ["mutation-simulator", "-o", str(td / "out"), str(self.infile)] + argv_tail
        try:
            self.args = ref.get_args()
        finally:
            sys.argv = old
        self.args.no_progress = True
        self.args.no_color = True
        self.fasta = ref_util.load_fasta(self.infile)
        self.sim = SimulationSettings.from_args(self.args, self.fasta, True)
        self.mutator = Mutator(self.args, self.fasta, self.sim)

    def outputs(self):
        self.mutator.close()
        fa = Path(self.args.outfasta).read_bytes()
        vcf = mask_vcf(Path(self.args.outvcf).read_bytes())
        return fa, vcf


def make_plan():
    print("plan goldens")
    out = {"env": ENV, "cases": []}
    cases = [
        ("snp_only_50k", [50_000], ["args", "-sn", "0.01"], 42, 42),
        ("svmix_120k", [120_000], ["args"] + C3_FLAGS, 42, 42),
        ("svmix_two_contigs", [30_000, 41_111], ["args"] + C3_FLAGS, 9, 10),
        ("iv_near_end", [2_000], ["args", "-iv", "0.05", "-ivmin", "100", "-ivmax", "400"], 3, 3),
        ("du_de_clamp", [1_500], ["args", "-du", "0.02", "-dumin", "50", "-dumax", "400", "-de",
                                  "0.02", "-demin", "50", "-demax", "400"], 4, 4),
        ("blocks", [20_000], ["args", "-sn", "0.05", "-snb", "6", "-in", "0.02", "-inmax", "4",
                              "-inb", "9", "-de", "0.02", "-demax", "5", "-deb", "3", "-dub", "3",
                              "-ivb", "3", "-tlb", "3"], 5, 5),
        ("dense_pool_path", [900], ["args", "-sn", "0.2", "-in", "0.011", "-inmax", "3"], 6, 6),
    ]
    for name, lengths, argv, sp, sn in cases:
        spec = {"contigs": [{"defline": f"p{i}", "length": L, "bpl": 60, "seed": 70 + i}
                            for i, L in enumerate(lengths)]}
        with tempfile.TemporaryDirectory() as td:
            sc = _Scratch(spec, argv, Path(td))
            random.seed(sp)
            np.random.seed(sn)
            entry = {"name": name, "input_spec": spec, "argv_tail": argv, "seed_py": sp,
                     "seed_np": sn, "mut_block": [[t.name, v] for t, v in sc.sim.mut_block.items()],
                     "sim": dump_sim(sc.sim), "contigs": []}
            for chrom in sc.sim.chromosomes:
                L = len(sc.fasta[chrom.number])
                for rng in chrom.range_definitions:
                    if not rng.mutation_settings.has_mutations:
                        continue
                    s_py, s_np = random.getstate(), np.random.get_state()
                    muts, tls, tlis = sc.mutator._Mutator__get_mutations(rng, L)
                    entry["contigs"].append({
                        "number": chrom.number, "length": L, "start": rng.start, "stop": rng.stop,
                        "range_index": chrom.range_definitions.index(rng),
                        "n_kept": len(muts),
                        "records": [[m.start, m.type.name, m.stop] for _, m in sorted(muts.items())]
                        if len(muts) <= 2500 else None,
                        "records_sha256": sha256(np.array(
                            [[m.start, m.type.value, m.stop] for _, m in sorted(muts.items())],
                            dtype=np.int64).tobytes()),
                        "py_words": py_words_consumed(s_py), "np_words": np_words_consumed(s_np),
                        "py_next": py_next_words(2), "np_next": np_next_words(2)})
            sc.outputs()
        out["cases"].append(entry)
    (HERE / "plan.json").write_text(json.dumps(out, indent=1) + "\n")


def make_apply():
    print("apply goldens")
    out = {"env": ENV, "cases": []}
    T = {"SN": MutType.SN, "IN": MutType.IN, "DE": MutType.DE, "IV": MutType.IV, "DU": MutType.DU}
    cases = [
        ("survey_edge", "ACGTNRYKMACGTACGTACGTNNACGTACGTACGTAC", 10, 1.0, 1, 1,
         [("IN", 0, 2), ("SN", 4, 4), ("SN", 5, 5), ("IV", 6, 9), ("DU", 12, 14), ("DE", 20, 23),
          ("SN", 22, 22), ("DE", 30, 37)]),
        ("del_at_zero", "ACGTACGTAC", 60, 1.0, 1, 1, [("DE", 0, 2)]),
        ("del_to_end", "ACGTACGTACGG", 5, 1.0, 2, 2, [("DE", 8, 11)]),
        ("ins_at_zero_and_last", "ACGTACGTAC", 4, 1.0, 3, 3, [("IN", 0, 4), ("IN", 9, 9)]),
        ("inv_palindrome_suppressed", "GGAATTCCACGTAAGCTTGG", 60, 1.0, 4, 4,
         [("IV", 2, 5), ("IV", 8, 11), ("IV", 13, 18)]),
        ("inv_iupac", "AAKSYMWRBDHV-NUXACGT", 7, 1.0, 5, 5, [("IV", 2, 15)]),
        ("dup_raw_iupac", "ACRYKMNNACGTAC", 60, 1.0, 6, 6, [("DU", 2, 6), ("DU", 13, 13)]),
        ("snp_all_bases_ti", "ACGTNKSYMWRBDHV-", 60, 1e9, 7, 7,
         [("SN", i, i) for i in range(16)]),
        ("snp_all_bases_tv", "ACGTNKSYMWRBDHV-ACGTNACGTN", 60, 0.0, 8, 8,
         [("SN", i, i) for i in range(26)]),
        ("snp_titv_mix", "ACGT" * 40, 60, 2.0, 9, 9, [("SN", i, i) for i in range(0, 160, 2)]),
        ("covered_skips", "ACGTACGTACGTACGTACGTACGTACGTACGT", 60, 1.0, 10, 10,
         [("DU", 2, 9), ("SN", 5, 5), ("IN", 7, 9), ("IV", 12, 20), ("DE", 15, 25),
          ("SN", 20, 20), ("SN", 21, 21), ("DE", 26, 31)]),
        ("adjacent_everything", "ACGTTGCAACGTTGCAACGTTGCAACGT", 9, 0.7, 11, 11,
         [("SN", 0, 0), ("IN", 1, 3), ("DE", 2, 3), ("IV", 4, 5), ("DU", 6, 6), ("SN", 7, 7),
          ("IN", 8, 8), ("DE", 9, 9), ("DU", 10, 12), ("IV", 13, 27)]),
    ]
    for name, seq, bpl, titv, sp, sn, muts in cases:
        spec = {"contigs": [{"defline": "edge case", "length": len(seq), "bpl": max(bpl, 1),
                             "seed": 0, "literal": seq}]}
        with tempfile.TemporaryDirectory() as td:
            sc = _Scratch(spec, ["args", "-sn", "0.01"], Path(td))
            md = {s: Mutation(T[t], s, e) for t, s, e in muts}
            random.seed(sp)
            np.random.seed(sn)
            s_py, s_np = random.getstate(), np.random.get_state()
            fw = sc.mutator._Mutator__fasta_writer
            fw.set_bpl(bpl)
            fw.write_header("edge case")
            exc = None
            try:
                sc.mutator._Mutator__mutate_sequence(sc.fasta[0], md, titv)
            except Exception as e:  # noqa: BLE001
                exc = {"type": type(e).__name__, "message": str(e)}
            pyw, npw = py_words_consumed(s_py), np_words_consumed(s_np)
            fa, vcf = sc.outputs()
            body = [l.decode() for l in vcf.split(b"\n") if l and not l.startswith(b"#")]
            out["cases"].append({"name": name, "sequence": seq, "bpl": bpl, "titv": titv,
                                 "seed_py": sp, "seed_np": sn, "muts": muts, "exception": exc,
                                 "fasta": fa.decode(), "vcf_body": body,
                                 "py_words": pyw, "np_words": npw})
    (HERE / "apply.json").write_text(json.dumps(out, indent=1) + "\n")


def _rmt_gene_blocking(lengths, seed):
    """A non-overlapping gene-blocking RMT (std `sn 0.01`, `None` blocks, hot / cold ranges, 1 kb `sn 0.2`
    hot spots whose sample takes CPython's pool path) from this file's own deterministic generator."""
    rs = np.random.RandomState(seed)
    out = ["titv = 2.0", "", "std", "it None", "sn 0.01", ""]
    for ci, L in enumerate(lengths):
        if L is None:
            continue                                   # unlisted contig: std only
        out.append(f"chr {ci + 1}")
        n_blocks = max(4, L // 20_000)
        kinds = rs.choice(4, size=n_blocks, p=[0.85, 0.06, 0.05, 0.04])
        blen = np.minimum(np.exp(rs.normal(np.log(2500.0), 1.3, n_blocks)).astype(np.int64) + 30, 60_000)
        blen[kinds == 3] = 1000
        blen[kinds == 2] = rs.randint(20_000, 120_000, int((kinds == 2).sum()))
        free = L - int(blen.sum()) - 2 * n_blocks - 500
        while free < L // 3:
            blen = np.maximum(blen // 2, 30)
            free = L - int(blen.sum()) - 2 * n_blocks - 500
        gaps = rs.dirichlet(np.ones(n_blocks + 1)) * free
        at = 1
        for b in range(n_blocks):
            at += int(gaps[b]) + 2
            a, e = at, at + int(blen[b]) - 1
            out.append(f"{a}-{e} " + ("None", "sn 0.05", "sn 0.001", "sn 0.2")[kinds[b]])
            at = e + 1
        assert at < L
    return "\n".join(out) + "\n"


RMT_SV_STD = """\
# SV mix on every contig through the std line (one range per contig), non-default blocks
titv = 2.0
du_block = 20
iv_block = 5
de_block = 3

std
it None
sn 0.004 in 0.001 inmin 1 inmax 30 de 0.001 demin 1 demax 80 du 0.0005 dumin 20 dumax 300 iv 0.0005 ivmin 20 ivmax 300
"""


def _rmt_gene_blocking_sv(lengths, seed, meta=("titv = 2.0",)):
    """The same gene-blocking layout with an SV `std` line -- the mainstream RMT shape: every gap between two blocked
    genes is a drawing range of the std settings -- plus hot ranges with their OWN SV settings (second settings object,
    chances in token order), cold SNP-only ranges and 1 kb pool-path hot spots."""
    rs = np.random.RandomState(seed)
    out = list(meta) + ["", "std", "it None",
                        "sn 0.005 in 0.001 inmin 1 inmax 50 de 0.001 demin 1 demax 50 du 0.0005 dumin 50 dumax 500 "
                        "iv 0.0005 ivmin 50 ivmax 500", ""]
    lines = ("None", "de 0.01 demin 1 demax 50 sn 0.03 du 0.002 dumin 50 dumax 500", "sn 0.001", "sn 0.2 in 0.01 inmin 1 inmax 50")
    for ci, L in enumerate(lengths):
        if L is None:
            continue                                   # unlisted contig: std only
        out.append(f"chr {ci + 1}")
        n_blocks = max(4, L // 20_000)
        kinds = rs.choice(4, size=n_blocks, p=[0.8, 0.08, 0.07, 0.05])
        blen = np.minimum(np.exp(rs.normal(np.log(2500.0), 1.3, n_blocks)).astype(np.int64) + 30, 60_000)
        blen[kinds == 3] = 1000
        blen[kinds == 2] = rs.randint(20_000, 120_000, int((kinds == 2).sum()))
        free = L - int(blen.sum()) - 2 * n_blocks - 500
        while free < L // 3:
            blen = np.maximum(blen // 2, 30)
            free = L - int(blen.sum()) - 2 * n_blocks - 500
        gaps = rs.dirichlet(np.ones(n_blocks + 1)) * free
        at = 1
        for b in range(n_blocks):
            at += int(gaps[b]) + (2 if b % 7 else 0)   # every seventh block touches the range before it
            a, e = at, at + int(blen[b]) - 1
            out.append(f"{a}-{e} " + lines[kinds[b]])
            at = e + 1
        assert at < L
    return "\n".join(out) + "\n"


def make_cli_cases_scaffolds():
    """An assembly-like input: hundreds of small scaffolds (the batch path of the product, msim_batch_run), with a
    large contig in the middle so that the stream chain crosses batch -> device engine -> batch."""
    print("CLI cases (many scaffolds)")
    rs = np.random.RandomState(77)
    contigs = []
    for i in range(320):
        L = int(rs.choice([37, 400, 1_500, 4_000, 9_000, 15_000]) + rs.randint(0, 300))
        contigs.append({"defline": f"scaf{i:04d} len={L}", "length": L, "bpl": int(rs.choice([60, 60, 70, 80])), "seed": 900 + i,
                        "decorate": bool(rs.randint(0, 4) == 0) and L > 2000})
    contigs.insert(150, {"defline": "chrBig in the middle", "length": 900_000, "bpl": 60, "seed": 899})
    cli_case("many_scaffolds", {"contigs": contigs},
             ["args", "-sn", "0.01", "-titv", "2.0", "-in", "0.002", "-inmax", "6", "-de", "0.002", "-demax", "9",
              "-du", "0.001", "-dumax", "20", "-iv", "0.001", "-ivmax", "15"], 21, 22, store="hash",
             notes="321 contigs: 320 scaffolds of 37 b - 15 kb (batched) around one 900 kb contig (SV-mix engine)")
    cli_case("many_scaffolds_tl", {"contigs": contigs[:60]},
             ["args", "-sn", "0.01", "-tl", "0.004", "-tlmin", "2", "-tlmax", "40", "-de", "0.002", "-demax", "9"], 23, 24,
             store="hash", notes="60 scaffolds with translocations (TL / TLI records inside a batch)")


def make_cli_cases_engines():
    """Mid-size cases that reach the device PLAN engines in AUTO mode (k >= 4096 on one range, or many
    deterministic-SNP ranges): the reference's own output pins them, not only the host planner."""
    print("CLI cases (device engines)")
    spec = {"contigs": [{"defline": "g1 gene blocking 3Mb", "length": 3_000_000, "bpl": 60, "seed": 71},
                        {"defline": "g2 std only", "length": 1_200_000, "bpl": 80, "seed": 72},
                        {"defline": "g3 dense blocks", "length": 400_000, "bpl": 60, "seed": 73}]}
    cli_case("rmt_blocks_3mb", spec, [], 42, 43, store="hash",
             rmt_text=_rmt_gene_blocking([3_000_000, None, 400_000], 5),
             notes="RMT gene blocking: ~170 drawing ranges incl. pool-path hot spots (host-sampled engine), an unlisted "
                   "contig (SNP sampler), titv 2")
    spec = {"contigs": [{"defline": "s1 3Mb", "length": 3_000_000, "bpl": 60, "seed": 81},
                        {"defline": "s2", "length": 700_011, "bpl": 70, "seed": 82}]}
    cli_case("rmt_svmix_blocks_3mb", spec, [], 11, 12, store="hash", rmt_text=RMT_SV_STD,
             notes="SV mix via the RMT std line, du/iv/de blocks 20/5/3 (SV-mix engine, 21 k + 4.9 k candidates)")
    spec = {"contigs": [{"defline": "v1 gene blocking + SV std 3Mb", "length": 3_000_000, "bpl": 60, "seed": 91},
                        {"defline": "v2 std only", "length": 600_000, "bpl": 80, "seed": 92},
                        {"defline": "v3 dense blocks", "length": 500_000, "bpl": 60, "seed": 93}]}
    cli_case("rmt_svstd_blocks_3mb", spec, [], 31, 32, store="hash",
             rmt_text=_rmt_gene_blocking_sv([3_000_000, None, 500_000], 6),
             notes="RMT gene blocking with an SV std line: ~170 drawing ranges of two SV settings objects, cold SNP ranges, "
                   "pool-path hot spots with insertions, touching ranges (host-chain engine); an unlisted contig (SV-mix engine)")
    spec = {"contigs": [{"defline": "w1 sn_block 3", "length": 1_500_000, "bpl": 60, "seed": 95},
                        {"defline": "w2 std only", "length": 300_000, "bpl": 70, "seed": 96}]}
    cli_case("rmt_snblock_svstd_1500k", spec, [], 33, 34, store="hash",
             rmt_text=_rmt_gene_blocking_sv([1_500_000, None], 7, meta=("titv = 0.5", "sn_block = 3", "du_block = 9")),
             notes="the same shape with sn_block 3 > min(block): kept SNPs block their successors, every candidate is on the "
                   "boundary chain (host-chain engine, both contigs)")


def make_reference_timing():
    """Wall time of the REAL reference CLI (file in -> Fasta + VCF out, one core) in THIS container, on the
    BASELINE config-1 shape (1 Mb, -sn 0.01) and on 10 Mb of the headline settings (-sn 0.01 -titv 2.0).
    bench.py carries the result in `cpu_baseline.reference_survey_mbases_s` (the reference itself never travels
    to the GPU box).  Opt-in (`make_goldens.py timing`): wall-clock numbers are not reproducible fixtures.
    Caveat: sequence access goes through the in-memory pyfaidx stand-in, which is faster than real pyfaidx."""
    import time
    print("reference timing")
    cpu = "unknown"
    for line in Path("/proc/cpuinfo").read_text().splitlines():
        if line.startswith("model name"):
            cpu = line.split(":", 1)[1].strip()
            break
    runs = []
    for name, length, tail in [("c1_1mb_sn0.01", 1_000_000, ["args", "-sn", "0.01"]),
                               ("10mb_sn0.01_titv2", 10_000_000, ["args", "-sn", "0.01", "-titv", "2.0"]),
                               ("1mb_c3_svmix", 1_000_000, ["args"] + C3_FLAGS)]:
        with tempfile.TemporaryDirectory() as td:
            spec = {"contigs": [{"defline": "contig1 synthetic", "length": length, "bpl": 60, "seed": 1234}]}
            infile = gin.write_input(spec, Path(td) / "in.fa")
            t0 = time.perf_counter()
            code, _, err, exc = run_reference_cli(["-q", "-o", str(Path(td) / "out"), str(infile)] + tail, 42, 42)
            dt = time.perf_counter() - t0
            assert code is None and exc is None, (code, err, exc)
        runs.append({"name": name, "bases": length, "argv": tail, "seconds": round(dt, 3),
                     "mbases_per_s": round(length / dt / 1e6, 4)})
        print(f"  {name}: {dt:.2f} s = {length / dt / 1e6:.3f} Mbases/s")
    out = {"_env": ENV, "cpu_model": cpu, "cores_used": 1, "host_cores": os.cpu_count(),
           "what": "mutation_simulator.__main__.main() of the real reference, whole CLI incl. file I/O, seeds 42/42",
           "published_readme_mbases_s": 0.19, "runs": runs}
    (HERE / "reference_timing.json").write_text(json.dumps(out, indent=1) + "\n")


RMT_IT = """\
# mutations AND interchromosomal translocations: the second pass reads what the first one wrote
titv = 2.0

std
it 0.0004
sn 0.01 in 0.002 inmin 1 inmax 9 de 0.002 demin 1 demax 12

chr 2
it None
chr 3
it 0.002
1-3000 None
3001-END sn 0.02
chr 5
it 0
"""


def make_cli_cases_readme():
    """The commands of the reference README's "How Tos" (README.md:205-240) at a size where the device PLAN engines engage."""
    print("CLI cases (README how-tos)")
    spec = {"contigs": [{"defline": "h1 how-to", "length": 800_000, "bpl": 60, "seed": 401},
                        {"defline": "h2", "length": 400_000, "bpl": 70, "seed": 402}]}
    cli_case("readme_howto_sn", spec, ["args", "-sn", "0.05"], 91, 92, store="hash",
             notes="README.md:210: SNPs every 20th base (SNP sampler engine)")
    cli_case("readme_howto_sn_titv", spec, ["args", "-sn", "0.05", "-titv", "0.5"], 93, 94, store="hash",
             notes="README.md:218: ti/tv 0.5")
    cli_case("readme_howto_snb", spec, ["args", "-sn", "0.05", "-snb", "10", "-titv", "0.5"], 95, 96, store="hash",
             notes="README.md:225: SNP block 10 -- above the sampling distance: every candidate is on the boundary chain "
                   "(host-chain engine, one range per contig)")
    cli_case("readme_howto_sv", spec, ["args", "-sn", "0.01", "-in", "0.01", "-de", "0.01", "-du", "0.01", "-iv", "0.01",
                                       "-tl", "0.01", "-inmin", "10", "-inmax", "100"], 97, 98, store="hash",
             notes="README.md:236: every type at 0.01 with inserts of 10-100 bases (SV-mix engine with translocations: 48 k + "
                   "24 k candidates)")


def make_cli_cases_it():
    """The `it` sub-command and RMT files with `it` lines (it_mutator.py, bedpe_writer.py)."""
    print("CLI cases (interchromosomal translocations)")
    four = {"contigs": [{"defline": "a1 first", "length": 50_000, "bpl": 60, "seed": 301},
                        {"defline": "a2", "length": 30_011, "bpl": 70, "seed": 302},
                        {"defline": "a3 third one", "length": 20_000, "bpl": 60, "seed": 303, "decorate": True},
                        {"defline": "a4", "length": 41_234, "bpl": 80, "seed": 304}]}
    cli_case("it_only_4ctg", four, ["it", "0.0005"], 51, 52, notes="`it` mode: two pairs, 17-22 breakpoints each; no _ms files")
    odd = {"contigs": [{"defline": "b1", "length": 9_000, "bpl": 60, "seed": 311},
                       {"defline": "b2 two bases", "length": 2, "bpl": 60, "seed": 312},
                       {"defline": "b3", "length": 12_345, "bpl": 50, "seed": 313},
                       {"defline": "b4", "length": 7_000, "bpl": 60, "seed": 314},
                       {"defline": "b5 three bases", "length": 3, "bpl": 60, "seed": 315},
                       {"defline": "b6", "length": 15_000, "bpl": 61, "seed": 316},
                       {"defline": "b7", "length": 4_321, "bpl": 60, "seed": 317}]}
    for sp, sn_ in ((61, 62), (63, 64), (65, 66)):
        cli_case(f"it_only_odd_s{sp}", odd, ["it", "0.002"], sp, sn_,
                 notes="`it` mode: a 2-base contig is left out, six take part -- who stays single depends on the walk over "
                       "the list the reference removes from; a 3-base contig can get a partner")
    cli_case("it_rate_high_low", {"contigs": [{"defline": "c1", "length": 100, "bpl": 60, "seed": 321},
                                               {"defline": "c2", "length": 300, "bpl": 60, "seed": 322},
                                               {"defline": "c3", "length": 280, "bpl": 60, "seed": 323},
                                               {"defline": "c4", "length": 90, "bpl": 60, "seed": 324}]},
             ["it", "0.5"], 71, 72, notes="rate 0.5 on short contigs: sample() raises on the shorter one of each pair -- first or "
                                        "second, after the longer one already drew -- both warnings, full copies written")
    cli_case("it_rate_tiny", four, ["it", "0.0000001"], 73, 74, notes="no breakpoint at all: the 'rates too low' warnings")
    spec = {"contigs": [{"defline": "m1", "length": 60_000, "bpl": 60, "seed": 331},
                        {"defline": "m2 no it", "length": 25_000, "bpl": 60, "seed": 332},
                        {"defline": "m3", "length": 33_333, "bpl": 70, "seed": 333},
                        {"defline": "m4", "length": 47_000, "bpl": 60, "seed": 334},
                        {"defline": "m5 it 0", "length": 52_000, "bpl": 80, "seed": 335},
                        {"defline": "m6", "length": 8_000, "bpl": 60, "seed": 336}]}
    cli_case("it_rmt_mutations", spec, [], 81, 82, rmt_text=RMT_IT,
             notes="RMT with mutations and it rates (std 0.0004, one contig None, one 0.002, one 0): mutation pass, then the IT "
                   "pass over the mutated Fasta; _ms and _ms_it files")


def main():
    os.chdir(HERE)
    which = set(sys.argv[1:]) or {"rng", "settings", "plan", "apply", "cli", "engines", "it", "readme"}
    if "timing" in which:
        make_reference_timing()
    if "it" in which:
        make_cli_cases_it()
    if "readme" in which:
        make_cli_cases_readme()
    if "engines" in which:
        make_cli_cases_engines()
    if "scaffolds" in which or "engines" in which:
        make_cli_cases_scaffolds()
    if "rng" in which:
        make_rng_kat()
    if "settings" in which:
        make_settings()
    if "plan" in which:
        make_plan()
    if "apply" in which:
        make_apply()
    if "cli" in which:
        make_cli_cases()
    print("done")


if __name__ == "__main__":
    main()
