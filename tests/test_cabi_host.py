"""C-ABI checks that need no GPU: the library loads, exports every symbol include/msim.h declares,
fails loudly without a device, and its host planner + VCF renderer reproduce the reference goldens
(through a host-only context, which can plan and render but cannot produce a sequence)."""
from __future__ import annotations

import random
import re
from pathlib import Path

import numpy as np
import pytest

from helpers import CASES, ROOT, all_case_names, case_meta, sha256
from mutation_simulator_amd import _ffi
from mutation_simulator_amd import mutator as mm
from pipeline import plan_only_vcf


def test_header_symbols_exported_and_bound():
    header = (ROOT / "include" / "msim.h").read_text()
    declared = set(re.findall(r"\b(msim_[a-z0-9_]+)\s*\(", header))
    lib = _ffi.load()
    bound = {name for name, _, _ in _ffi.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name)
    assert lib.msim_abi_version() == 1


def test_struct_layouts_match_header():
    assert _ffi.C.sizeof(_ffi.Record) == 16 and _ffi.RECORD_DTYPE.itemsize == 16
    assert _ffi.C.sizeof(_ffi.Range) == 8 * 4 + 4 + 4 * 8 + 4 + 8 * 8 * 3
    assert _ffi.C.sizeof(_ffi.Params) == 72


def test_no_silent_cpu_fallback():
    """Without a GPU the product must refuse, not compute on the CPU."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); from mutation_simulator_amd import _ffi\n"
            "import ctypes\n"
            "try:\n    _ffi.Engine(0); print('CREATED')\nexcept _ffi.MsimError as e:\n    print('REFUSED', e)\n"
            % str(ROOT / "mutation-simulator_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         env={"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": "", "PATH": "/usr/bin:/bin"})
    assert "REFUSED" in out.stdout or "CREATED" in out.stdout
    if "CREATED" in out.stdout:
        pytest.skip("a GPU is visible even with empty visibility masks")
    eng = _ffi.Engine(device=-1)
    cid = eng.add_contig(np.frombuffer(b"ACGT" * 10, dtype=np.uint8))
    p = _ffi.Params()
    for i in range(8):
        p.block[i] = 1
    eng.set_params(p)
    eng.plan_contig(cid, [])
    with pytest.raises(_ffi.MsimError, match="needs the GPU"):
        eng.apply_contig(cid)
    with pytest.raises(_ffi.MsimError, match="needs the GPU"):
        eng.add_contig_synthetic(100, 1)


def test_mt_state_roundtrip_with_python_generators():
    random.seed(1234)
    np.random.seed(4321)
    [random.random() for _ in range(700)]
    np.random.random_sample(333)
    eng = _ffi.Engine(device=-1)
    mm.export_python_streams(eng)
    want_py = random.getrandbits(32)
    want_np = int(np.random.randint(0, 4294967296, dtype=np.uint32))
    random.seed(0)
    np.random.seed(0)
    mm.import_python_streams(eng)
    assert random.getrandbits(32) == want_py
    assert int(np.random.randint(0, 4294967296, dtype=np.uint32)) == want_np


def test_threshold_arithmetic_matches_float_compare():
    """Integer thresholds == the reference's double comparisons, probed at the boundaries."""
    rng = np.random.RandomState(5)
    for _ in range(200):
        p = rng.random_sample(rng.randint(1, 8))
        p[rng.randint(0, len(p))] = 0.0 if len(p) > 1 else p[0]
        p = p / p.sum()
        cdf = np.cumsum(p.astype(np.float64))
        cdf /= cdf[-1]
        thr = [mm._ceil_scaled(float(c)) for c in cdf]
        probes = set()
        for t in thr:
            probes.update(m for m in (t - 1, t, t + 1) if 0 <= m < (1 << 53))
        probes.update(int(x) for x in rng.randint(0, 2**53, 20, dtype=np.int64))
        for m in probes:
            u = m / 9007199254740992.0
            assert sum(1 for t in thr if t <= m) == int(np.searchsorted(cdf, u, side="right"))
    for titv in [0.0, 0.5, 1.0, 2.0, 2.5, 1e9, 1e-9, 3.0, float("inf")]:
        p_ti = titv * (1 / (titv + 1))
        class S:  # noqa: D401
            mut_block = {}
        S.titv = titv
        lim = mm.params_descriptor(S).ti_lim
        for m in {0, 1, lim - 1, lim, lim + 1, (1 << 53) - 1} - {-1}:
            if 0 <= m < (1 << 53):
                assert (m < lim) == (m / 9007199254740992.0 <= p_ti), (titv, m)


def test_setsize_matches_cpython():
    for k in [0, 1, 5, 6, 7, 21, 22, 85, 86, 211, 341, 342, 1365, 1366, 10000, 2_480_000, 87381, 87382]:
        want = 21 + (4 ** __import__("math").ceil(__import__("math").log(k * 3, 4)) if k > 5 else 0)
        assert mm.sample_setsize(k) == want


RUNNABLE = [n for n in all_case_names() if case_meta(n).get("sim") is not None]


@pytest.mark.parametrize("name", RUNNABLE)
def test_host_planner_vcf_matches_reference(name, tmp_path):
    """settings -> msim_plan_contig (host-only ctx) -> msim_render_vcf == the reference's VCF."""
    meta = case_meta(name)
    if meta["exception"] is not None and meta["exception"]["type"] == "ValueError":
        with pytest.raises(ValueError) as ei:
            plan_only_vcf(meta, tmp_path)
        assert str(ei.value) == meta["exception"]["message"]
        return
    if meta["exception"] is not None:
        pytest.skip("KeyError surfaces in APPLY (GPU test)")
    vcf, empty, eng = plan_only_vcf(meta, tmp_path)
    assert len(vcf) == meta["vcf_len"] and sha256(vcf) == meta["vcf_sha256"]
    if meta["store"] == "full":
        assert vcf == (CASES / name / "expected_ms.vcf").read_bytes()
    warned = [int(l.split("sequence ")[1].split(" ")[0]) - 1
              for l in meta["stderr"].splitlines() if "No mutations could be generated" in l]
    assert empty == warned
    # both Python generators stand where the reference left them
    assert [random.getrandbits(32) for _ in range(4)] == meta["py_next_words_after"]
    assert [int(x) for x in np.random.randint(0, 4294967296, size=4, dtype=np.uint32)] == \
        meta["np_next_words_after"]
