"""Pins the CPU oracle (oracle/msim_oracle.c) against goldens captured from the real reference.

Every later parity claim (HIP path == oracle) rests on these: the oracle has to reproduce, bit for
bit, what ``/root/reference`` produced under fixed seeds (tests/golden/make_goldens.py).
"""
from __future__ import annotations

import numpy as np
import pytest

from helpers import (CASES, all_case_names, case_input_bytes, case_meta, load_json,
                     parse_fasta_bytes, sha256)
from oracle import oracle as orc

KAT = load_json("rng_kat.json")


# ------------------------------------------------------------------------------ RNG KATs
@pytest.mark.parametrize("seed", list(KAT["py_seed_words"].keys()))
def test_py_seed_words(seed):
    o = orc.Oracle()
    o.seed(int(seed), 0)
    assert o.py_words32(8) == KAT["py_seed_words"][seed]


@pytest.mark.parametrize("seed", list(KAT["np_seed_words"].keys()))
def test_np_seed_words(seed):
    o = orc.Oracle()
    o.seed(0, int(seed))
    assert o.np_words32(8) == KAT["np_seed_words"][seed]


def test_state_roundtrip():
    import random
    random.seed(42)
    [random.getrandbits(32) for _ in range(1000)]
    st = random.getstate()[1]
    o = orc.Oracle()
    o.set_state(0, st[:624], st[624])
    assert st[624] == KAT["py_state_after_1000"]["pos"]
    assert o.py_words32(4) == KAT["py_state_after_1000"]["next"]


@pytest.mark.parametrize("e", KAT["randbelow"], ids=lambda e: f"n={e['n']}")
def test_randbelow(e):
    if e["n"] >= 2**63:
        pytest.skip("oracle supports n < 2**63 (contigs >= 4.29 Gb are out of scope, SURVEY 7.3 H1)")
    o = orc.Oracle()
    o.seed(e["seed"], 0)
    assert [o.randbelow(e["n"]) for _ in e["values"]] == e["values"]
    assert o.words(0) == e["words"]
    assert o.py_words32(2) == e["next"]


@pytest.mark.parametrize("e", KAT["randint"], ids=lambda e: f"{e['a']}-{e['b']}")
def test_randint(e):
    o = orc.Oracle()
    o.seed(e["seed"], 0)
    assert [o.randint(e["a"], e["b"]) for _ in e["values"]] == e["values"]
    assert o.words(0) == e["words"]


def test_uniform():
    o = orc.Oracle()
    o.seed(KAT["uniform"]["seed"], 0)
    assert [o.uniform01().hex() for _ in KAT["uniform"]["values_hex"]] == KAT["uniform"]["values_hex"]


@pytest.mark.parametrize("e", KAT["sample"], ids=lambda e: f"n{e['n']}k{e['k']}")
def test_sample(e):
    o = orc.Oracle()
    o.seed(e["seed"], 0)
    assert orc.lib().orc_setsize(e["k"]) == e["setsize"]
    vals = o.sample(e["n"], e["k"])
    assert sha256(np.array(vals, dtype=np.int64).tobytes()) == e["sha256_order"]
    if "values" in e:
        assert vals == e["values"]
    assert o.words(0) == e["words"]
    assert o.py_words32(2) == e["next"]


@pytest.mark.parametrize("e", KAT["sample_with_minimum_distance"], ids=lambda e: f"k{e['k']}d{e['d']}")
def test_sample_min_dist(e):
    o = orc.Oracle()
    o.seed(e["seed"], 0)
    vals = o.sample_min_dist(e["start"], e["stop"], e["k"], e["d"])
    assert sha256(vals.astype(np.int64).tobytes()) == e["sha256"]
    assert o.words(0) == e["words"]
    if e["k"] > 1:
        assert np.all(np.diff(vals) >= 1 + e["d"])


def test_sample_value_error():
    o = orc.Oracle()
    o.seed(1, 1)
    with pytest.raises(ValueError, match="Sample larger than population or is negative"):
        o.sample(5, 6)
    with pytest.raises(ValueError):
        o.sample(5, -1)
    with pytest.raises(ValueError):
        o.sample_min_dist(100, 98, 3, 1)      # negative-length filler range (rmt.py:243-255)


@pytest.mark.parametrize("e", KAT["np_choice_p"], ids=lambda e: f"seed{e['seed']}")
def test_np_choice_p(e):
    o = orc.Oracle()
    o.seed(0, e["seed"])
    p = [float.fromhex(h) for h in e["p_hex"]]
    assert o.choice_p(p, e["size"]) == e["idx"]
    assert o.words(1) == e["words"]


def test_np_choice_atgc():
    e = KAT["np_choice_atgc"]
    o = orc.Oracle()
    o.seed(0, e["seed"])
    assert o.choice_atgc(e["size"]) == e["value"]
    assert o.words(1) == e["words"]


# ------------------------------------------------------------------------------ plan goldens
PLAN = load_json("plan.json")


@pytest.mark.parametrize("case", PLAN["cases"], ids=lambda c: c["name"])
def test_plan(case):
    o = orc.Oracle()
    o.seed(case["seed_py"], case["seed_np"])
    o.configure(case["sim"])
    for g in case["contigs"]:
        rd = case["sim"]["chromosomes"][g["number"]]["ranges"][g["range_index"]]
        w0, w1 = o.words(0), o.words(1)
        recs = o.get_mutations(rd, g["length"])
        assert len(recs) == g["n_kept"]
        arr = np.array(recs, dtype=np.int64).reshape(-1, 3)
        assert sha256(arr.tobytes()) == g["records_sha256"]
        if g["records"] is not None:
            assert [[p, orc.TYPE_NAME[t], s] for p, t, s in recs] == g["records"]
        assert o.words(0) - w0 == g["py_words"]
        assert o.words(1) - w1 == g["np_words"]


# ------------------------------------------------------------------------------ apply goldens
APPLY = load_json("apply.json")


@pytest.mark.parametrize("case", APPLY["cases"], ids=lambda c: c["name"])
def test_apply(case):
    o = orc.Oracle()
    o.seed(case["seed_py"], case["seed_np"])
    o.configure({"mut_block": [], "titv": case["titv"]})
    muts = [(orc.TYPE_ID[t], s, e) for t, s, e in case["muts"]]
    o.mutate_sequence(case["sequence"].encode(), "edge", "edge case", case["bpl"], muts)
    fa, vcf = o.outputs()
    assert fa.decode() == case["fasta"]
    assert [l for l in vcf.decode().split("\n") if l] == case["vcf_body"]
    assert o.words(0) == case["py_words"]
    assert o.words(1) == case["np_words"]


# ------------------------------------------------------------------------------ whole-CLI goldens
# (cases with an interchromosomal-translocation pass: tests/test_it_host.py -- mutation pass included where there is one)
RUNNABLE = [n for n in all_case_names() if case_meta(n).get("sim") is not None and "it_fasta_len" not in case_meta(n)]


@pytest.mark.parametrize("name", RUNNABLE)
def test_cli_case(name):
    meta = case_meta(name)
    contigs = parse_fasta_bytes(case_input_bytes(meta))
    for c, g in zip(contigs, meta["contigs"]):
        assert (c["name"], c["long_name"], len(c["bases"]), c["lenc"]) == \
               (g["name"], g["long_name"], g["length"], g["lenc"])
    o = orc.Oracle()
    o.seed(meta["seed_py"], meta["seed_np"])
    if meta["exception"] is not None:
        exc = {"ValueError": ValueError, "KeyError": KeyError}[meta["exception"]["type"]]
        with pytest.raises(exc) as ei:
            o.run_genome(contigs, meta["sim"], meta["infile_name"])
        if exc is KeyError:
            assert repr(ei.value.args[0]) == meta["exception"]["repr_args"][0]
        else:
            assert str(ei.value) == meta["exception"]["message"]
        return
    fa, vcf, empty, _ = o.run_genome(contigs, meta["sim"], meta["infile_name"])
    assert len(fa) == meta["fasta_len"] and sha256(fa) == meta["fasta_sha256"]
    assert len(vcf) == meta["vcf_len"] and sha256(vcf) == meta["vcf_sha256"]
    if meta["store"] == "full":
        assert fa == (CASES / name / "expected_ms.fa").read_bytes()
        assert vcf == (CASES / name / "expected_ms.vcf").read_bytes()
    warned = [int(l.split("sequence ")[1].split(" ")[0]) - 1
              for l in meta["stderr"].splitlines() if "No mutations could be generated" in l]
    assert empty == warned
    # the streams must stand where the reference left them
    assert o.py_words32(4) == meta["py_next_words_after"]
    assert o.np_words32(4) == meta["np_next_words_after"]
