"""``MSIM_RNG_FAST`` (``--rng fast``): the counter-based generator mode.  It is NOT stream-compatible with the reference by
design, so there is no golden to compare with; what is checked instead:

* the device against a numpy restatement of the same algorithm (Philox4x32-10 draws, rounds until k distinct, rank shift,
  SNP outcome) -- exact equality of every record;
* the properties the reference's construction guarantees (util.py:93-109, mutator.py:428-455): exactly k = int(len * rate)
  positions per range, inside the range, at least d + 1 apart, uniform over the range (chi-square), transition share p_ti,
  the two transversion columns equally likely;
* APPLY + device text on these records against the ORACLE's ``__mutate_sequence`` / VCF writer (parity of everything
  behind PLAN does not depend on where the records came from);
* determinism, key sensitivity, contig ordinals (``msim_plan_chain`` keeps ranks aligned), refusal of SV settings.
"""
from __future__ import annotations

import numpy as np
import pytest

from inputs import random_bases
from mutation_simulator_amd import _ffi
from oracle import oracle as orc
from test_gpu_sampler import _params, _snp_range, _sv_range, C3_CHANCES, C3_LENS

pytestmark = pytest.mark.gpu

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox(c0, c1, c2, c3, key):
    """Philox4x32-10 over numpy uint64 arrays holding 32-bit values (Salmon et al. 2011)."""
    c = [np.asarray(x, dtype=np.uint64) for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = key & MASK, key >> 32
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c[0], np.uint64(M1) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & np.uint64(MASK),
             (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & np.uint64(MASK)]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c


def restate(ranges, d, ti_lim, key, seq):
    """(positions, aux) as plan_kernels.h computes them: per range, rounds of draws j = done .. done + rem - 1."""
    pos_all = []
    for r, (start, stop, k) in enumerate(ranges):
        n = (stop - (k - 1) * d) - start
        have = np.zeros(0, dtype=np.int64)
        done, rem = 0, k
        while rem:
            j = np.arange(done, done + rem, dtype=np.uint64)
            x = philox(j, r, seq, 0, key)
            x64 = [int(a) | (int(b) << 32) for a, b in zip(x[0], x[1])]
            v = np.array([(w * n) >> 64 for w in x64], dtype=np.int64)
            have = np.union1d(have, v)
            done += rem
            rem = k - len(have)
        pos_all.append(start + have + d * np.arange(k, dtype=np.int64))
    pos = np.concatenate(pos_all) if pos_all else np.zeros(0, dtype=np.int64)
    x = philox(np.arange(len(pos), dtype=np.uint64), 0, seq, 1, key)
    u53 = ((x[1] << np.uint64(32)) | x[0]) >> np.uint64(11)
    aux = np.where(u53 < np.uint64(ti_lim), 0, 1 + (x[2] & np.uint64(1))).astype(np.uint8)
    return pos, aux


def restate_one_big_range(start, stop, k, d, ti_lim, key, seq):
    """One range of >= 65536 positions: the contig's own word stream (word 4 i + q = lane q of philox((i, 0, seq, 2))) through
    ``random.sample``'s set path as the binned sampler runs it -- tempered word >> (32 - bits) accepted below n, the first k
    distinct accepted values."""
    n = (stop - (k - 1) * d) - start
    bits = int(n).bit_length()
    m = 4 * k + 65536
    x = philox(np.arange((m + 3) // 4, dtype=np.uint64), 0, seq, 2, key)
    w = np.stack(x, axis=1).reshape(-1)[:m]
    y = w ^ (w >> np.uint64(11))
    y = y ^ ((y << np.uint64(7)) & np.uint64(0x9D2C5680))
    y = y ^ ((y << np.uint64(15)) & np.uint64(0xEFC60000))
    y = (y ^ (y >> np.uint64(18))) & np.uint64(MASK)
    v = (y >> np.uint64(32 - bits)).astype(np.int64)
    acc = v[v < n]
    _, first = np.unique(acc, return_index=True)
    cut = np.sort(first)[k - 1]                       # where the k-th distinct value appears
    have = np.unique(acc[:cut + 1])
    assert len(have) == k
    pos = start + have + d * np.arange(k, dtype=np.int64)
    x = philox(np.arange(k, dtype=np.uint64), 0, seq, 1, key)
    u53 = ((x[1] << np.uint64(32)) | x[0]) >> np.uint64(11)
    aux = np.where(u53 < np.uint64(ti_lim), 0, 1 + (x[2] & np.uint64(1))).astype(np.uint8)
    return pos, aux


def test_one_big_range_equals_numpy_restatement():
    params = _params(titv=2.0)
    eng = _ffi.Engine(0, _ffi.RNG_FAST)
    eng.set_params(params)
    eng.set_fast_key(99)
    for seq, (L, k) in enumerate([(3_000_000, 70_000), (40_000_000, 400_000), (1_000_000, 300_000)]):
        _, recs = _plan(eng, L, [_snp_range(0, L - 1, k)])
        pos, aux = restate_one_big_range(0, L - 1, k, 1, _ti_lim(params), 99, seq)
        assert np.array_equal(recs["pos"], pos) and np.array_equal(recs["aux"], aux)
        eng.clear()
    eng.close()


def _plan(eng, L, ranges, bases=None):
    cid = eng.add_contig(bases) if bases is not None else eng.add_contig_synthetic(L, 7)
    eng.plan_contig(cid, ranges)
    recs, pool = eng.fetch_records(cid)
    return cid, recs.copy()


def _ti_lim(params):
    return int(params.ti_lim)


@pytest.mark.parametrize("key", [1, 0xDEADBEEFCAFEF00D])
def test_device_equals_numpy_restatement(key):
    params = _params(titv=2.0)
    eng = _ffi.Engine(0, _ffi.RNG_FAST)
    eng.set_params(params)
    eng.set_fast_key(key)
    shapes = [(300_000, [(0, 299_999, 3_000)]),
              (500_000, [(1_000, 49_999, 4_900), (60_000, 60_400, 100), (100_000, 499_999, 2_000)]),   # dense: many rounds
              (70_000, [(10, 69_000, 1)])]
    for seq, (L, rs) in enumerate(shapes):
        _, recs = _plan(eng, L, [_snp_range(a, b, k) for a, b, k in rs])
        pos, aux = restate(rs, 1, _ti_lim(params), key, seq)
        assert np.array_equal(recs["pos"], pos) and np.array_equal(recs["stop"], pos)
        assert np.array_equal(recs["aux"], aux)
        assert (recs["type"] == 1).all()
        eng.clear()
    assert eng.stats()["contigs_fast"] == 3
    eng.close()


def test_properties_at_scale():
    L = 60_000_000
    blocks = {t: 3 for t in ("SN", "IN", "DE", "IV", "DU", "TL", "TLI")}          # min distance d = 3
    params = _params(blocks, titv=2.0)
    rs = [(0, 19_999_999, 200_000), (20_000_000, 20_004_999, 600), (25_000_000, L - 1, 350_000)]
    eng = _ffi.Engine(0, _ffi.RNG_FAST)
    eng.set_params(params)
    eng.set_fast_key(77)
    _, recs = _plan(eng, L, [_snp_range(a, b, k) for a, b, k in rs])
    pos = recs["pos"].astype(np.int64)
    assert len(pos) == sum(k for _, _, k in rs)
    assert np.all(np.diff(pos) > 3)                                          # sorted, distinct, at least d + 1 apart
    at = 0
    for a, b, k in rs:
        p = pos[at:at + k]
        at += k
        assert p[0] >= a and p[-1] <= b
        # uniformity of the underlying sample (positions minus their rank shift) over 64 bins
        v = p - a - 3 * np.arange(k)
        n = (b - (k - 1) * 3) - a
        if k >= 64 * 50:
            counts = np.bincount((v * 64 // n).astype(np.int64), minlength=64)
            chi2 = float(((counts - k / 64) ** 2 / (k / 64)).sum())
            assert chi2 < 150, chi2                                          # 63 degrees of freedom: mean 63, sd 11
    aux = recs["aux"]
    p_ti = 2.0 * (1 / 3.0)
    share = float((aux == 0).mean())
    assert abs(share - p_ti) < 5 * np.sqrt(p_ti * (1 - p_ti) / len(aux))
    tv = aux[aux > 0]
    assert abs(float((tv == 1).mean()) - 0.5) < 5 * 0.5 / np.sqrt(len(tv))
    eng.close()


def test_apply_and_text_vs_oracle_on_fast_records():
    """Everything behind PLAN (rewrite kernels, framing, VCF text) against the oracle fed with the very same records."""
    L = 2_000_000
    bases = random_bases(L, 9)
    params = _params(titv=1.0)
    eng = _ffi.Engine(0, _ffi.RNG_FAST)
    eng.set_params(params)
    eng.set_fast_key(5)
    cid, recs = _plan(eng, L, [_snp_range(0, L - 1, 20_000)], bases=bases)
    eng.apply_contig(cid)
    got = eng.fetch_sequence(cid)
    # the SNP outcome is part of the record (aux); the host renderer (pinned to the reference's goldens) turns it into the
    # ALT base of every line -- the rewrite kernel must have put exactly those bases into the sequence
    host_vcf = _ffi.render_vcf(recs, np.zeros(0, dtype=np.uint8), bases, "chrF")
    want = bases.copy()
    n_lines = 0
    for line in host_vcf.split(b"\n"):
        if not line:
            continue
        f = line.split(b"\t")
        p = int(f[1]) - 1
        assert f[3] == bytes([bases[p]]) and len(f[4]) == 1 and f[4] != f[3]
        want[p] = f[4][0]
        n_lines += 1
    assert n_lines == len(recs)
    assert np.array_equal(got, want)
    vcf = eng.render_vcf_device(cid, "chrF").tobytes()
    assert vcf == host_vcf
    eng.close()


def test_determinism_keys_and_ordinals():
    params = _params(titv=2.0)
    r = [_snp_range(0, 999_999, 10_000)]

    def run(key, skip_first):
        eng = _ffi.Engine(0, _ffi.RNG_FAST)
        eng.set_params(params)
        eng.set_fast_key(key)
        out = []
        for i in range(3):
            if i == 0 and skip_first:
                eng.plan_chain(1_000_000, r)                # a contig another rank owns: only its ordinal is consumed
                out.append(None)
                continue
            _, recs = _plan(eng, 1_000_000, r)
            out.append(recs)
            eng.clear()
        eng.close()
        return out
    a, b, c, d = run(11, False), run(11, False), run(12, False), run(11, True)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))                    # same key: same records
    assert not np.array_equal(a[0]["pos"], a[1]["pos"])                              # contig ordinals differ
    assert not np.array_equal(a[0]["pos"], c[0]["pos"])                              # keys differ
    assert np.array_equal(a[1].view(np.uint8), d[1].view(np.uint8)) and np.array_equal(a[2].view(np.uint8), d[2].view(np.uint8))


def test_refuses_what_it_does_not_cover():
    eng = _ffi.Engine(0, _ffi.RNG_FAST)
    eng.set_params(_params(titv=1.0))
    cid = eng.add_contig_synthetic(1_000_000, 1)
    with pytest.raises(_ffi.MsimError, match="fast RNG mode"):
        eng.plan_contig(cid, [_sv_range(0, 999_999, 9_000, C3_CHANCES, C3_LENS)])       # SV types
    with pytest.raises(_ffi.MsimError, match="fast RNG mode"):
        eng.plan_contig(cid, [_snp_range(0, 999_999, 600_000)])                          # denser than every second slot
    eng.set_params(_params({"SN": 4}, titv=1.0))
    with pytest.raises(_ffi.MsimError, match="fast RNG mode"):
        eng.plan_contig(cid, [_snp_range(0, 999_999, 5_000)])                            # SNP block above the minimum block
    eng.close()


def test_cli_rng_fast_end_to_end(tmp_path):
    """``--rng fast`` through the CLI: k SNPs per contig exactly where int(len * rate) says, files consistent with each other
    (every VCF line's REF is the input base, its ALT the output base; nothing else changed), reproducible under --seed,
    different under another seed, identical with --gpus 2; SV flags are refused with the library's message."""
    import contextlib
    import io

    import inputs as gin
    from helpers import parse_fasta_bytes
    from mutation_simulator_amd import __main__ as cli
    spec = {"contigs": [{"defline": "f1 fast", "length": 900_000, "bpl": 60, "seed": 1},
                        {"defline": "f2", "length": 5_000, "bpl": 50, "seed": 2},
                        {"defline": "f3", "length": 300_011, "bpl": 70, "seed": 3}]}
    infile = gin.write_input(spec, tmp_path / "in.fa")
    src = parse_fasta_bytes(infile.read_bytes())

    def run(tag, *extra):
        with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
            cli.main(["-q", "--rng", "fast", *extra, "-o", str(tmp_path / tag), str(infile), "args", "-sn", "0.01", "-titv", "2.0"])
        return (tmp_path / f"{tag}_ms.fa").read_bytes(), (tmp_path / f"{tag}_ms.vcf").read_bytes()
    fa, vcf = run("a", "--seed", "7")
    out = parse_fasta_bytes(fa)
    lines = [l.split(b"\t") for l in vcf.split(b"\n") if l and not l.startswith(b"#")]
    for c_in, c_out in zip(src, out):
        mine = [l for l in lines if l[0] == c_in["name"].encode()]
        assert len(mine) == int(len(c_in["bases"]) * 0.01)
        changed = np.flatnonzero(c_in["bases"] != c_out["bases"])
        assert [int(l[1]) - 1 for l in mine] == changed.tolist()
        assert all(l[3] == bytes([c_in["bases"][p]]) and l[4] == bytes([c_out["bases"][p]]) for l, p in zip(mine, changed))
    assert run("b", "--seed", "7") == (fa, vcf)
    fa2, vcf2 = run("c", "--seed", "8")
    assert fa2 != fa
    import os
    os.environ["MSIM_SHARD_DEVICES"] = "0,0"
    try:
        assert run("d", "--seed", "7", "--gpus", "2") == (fa, vcf)
    finally:
        del os.environ["MSIM_SHARD_DEVICES"]
    err = io.StringIO()
    with pytest.raises(SystemExit), contextlib.redirect_stderr(err):
        cli.main(["-q", "--rng", "fast", "-o", str(tmp_path / "e"), str(infile), "args", "-sn", "0.01", "-de", "0.001"])
    assert "fast RNG mode" in err.getvalue()


@pytest.mark.parametrize("workload,engines", [("c2", 24), ("c4", 24)])
def test_full_genome_fast_properties(workload, engines):
    """The bench genome (3 Gb, 24 contigs) through the fast mode with the settings of BASELINE configs[1] / configs[3]: per
    contig exactly sum(k) records, sorted, at least d + 1 apart, every range holding exactly its k positions; APPLY changes
    exactly those bases (checksum of the changed positions against the records' outcomes is the rewrite kernels' own test:
    here only the count of differing bases is taken, on the largest contig)."""
    import bench
    from mutation_simulator_amd import mutator as mm
    lengths = bench.contig_lengths(3_000_000_000)
    sim = bench.build_settings(workload, lengths)
    eng = _ffi.Engine(0, _ffi.RNG_FAST)
    eng.set_params(mm.params_descriptor(sim))
    eng.set_fast_key(2024)
    total = 0
    for chrom in sim.chromosomes:
        L = lengths[chrom.number]
        table = mm.plan_table(chrom)
        cid = eng.add_contig_synthetic(L, 1000 + chrom.number)
        eng.plan_contig(cid, table)
        recs, _ = eng.fetch_records(cid)
        pos = recs["pos"].astype(np.int64)
        k = table["k"].astype(np.int64)
        assert len(pos) == int(k.sum())
        assert np.all(np.diff(pos) > 1)
        drawing = table[k > 0]
        edges = np.searchsorted(pos, drawing["start"].astype(np.int64))
        counts = np.diff(np.concatenate((edges, [len(pos)])))
        assert np.array_equal(counts, drawing["k"].astype(np.int64))                    # every range: exactly its k positions ...
        last = pos[np.cumsum(counts) - 1]
        assert np.all(last <= drawing["stop"].astype(np.int64))                          # ... none beyond its stop
        if chrom.number == 0:
            before = eng.read_contig(cid)
            eng.apply_contig(cid)
            after = eng.fetch_sequence(cid)
            assert int((before != after).sum()) == len(pos)
        total += len(pos)
        eng.clear()
    assert eng.stats()["contigs_fast"] == engines and total > 10_000_000
    eng.close()
