"""csrc/mt_jump_table.h -- 265 generated GF(2) polynomials, 2 MB, the origin of every device stream session -- is what its
generator says it is (a stale or hand-edited header must not ship).

* `check_mt_jump.check_quick` (what `__graft_entry__.build()` runs): constants vs `gen_mt_jump.py`, three polynomials against
  sequential generation, a sample by composition -- numpy only.
* EVERY polynomial against plain sequential generation: the oracle's MT19937 (CPython's `_randommodule.c` restated,
  oracle/msim_oracle.c) steps through the 7168 chunks = 1.145 G words the table reaches, and at each of the 265 multiples its state
  must equal the polynomial applied to the start state (the reference consumes the same sequence word by word:
  `random.sample` / `randint`, util.py:104, mutator.py:246-262).
* a flipped bit anywhere in a checked polynomial is caught."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "mutation-simulator_amd" / "tools"))
import check_mt_jump as cj  # noqa: E402


def test_quick_check_passes_on_the_committed_header():
    assert cj.check_quick() >= 9


def test_every_polynomial_against_sequential_generation_by_the_oracle():
    from oracle import oracle as orc
    consts, P = cj.parse_header()
    cj.check_constants(consts)
    chunk, deg = consts["MT_CHUNK_WORDS"], consts["MT_POLY_DEG"]
    mult = cj.multiples(consts["MT_JUMP_RADIX"])
    assert len(mult) == len(P) == 265 and max(mult) == 7168
    rs = np.random.RandomState(99)
    state = rs.randint(0, 2 ** 32, size=cj.N, dtype=np.uint64).astype(np.uint32)
    head = cj.raw_sequence(state, deg + 2 * cj.N)
    o = orc.Oracle()
    o.set_state(0, [int(x) for x in state], cj.N)          # idx 624: the next word regenerates -- x[624] is the first output
    want_at = {}
    need = set(mult)
    for c in range(1, max(mult) + 1):
        o.skip_words(0, chunk)                             # (the outputs themselves are not needed: the state behind them is)
        if c in need:
            mt, idx = o.get_state(0)
            assert idx == cj.N
            want_at[c] = np.array(mt, dtype=np.uint32)     # = x[c * chunk .. c * chunk + 624)
    bad = [i for i, m in enumerate(mult) if not cj.same_state(cj.apply_jump(P[i], head), want_at[m])]
    assert not bad, f"polynomials {bad[:8]} are not jumps by their multiples of {chunk} words"


def test_a_flipped_coefficient_is_caught(tmp_path):
    text = cj.HEADER.read_text()
    at = text.index("0x", text.index("MT_JUMP_POLY[MT_JUMP_POLYS][MT_POLY_WORDS]"))
    word = int(text[at + 2:at + 10], 16) ^ 0x10
    bad = tmp_path / "mt_jump_table.h"
    bad.write_text(text[:at + 2] + f"{word:08x}" + text[at + 10:])
    with pytest.raises(AssertionError, match="not a jump by 1 chunk"):
        cj.check_quick(bad)
    stale = tmp_path / "stale.h"
    stale.write_text(text.replace("MT_JUMP_RADIX[MT_JUMP_LEVELS] = {256, 4, 8}", "MT_JUMP_RADIX[MT_JUMP_LEVELS] = {16, 16, 32}"))
    with pytest.raises(AssertionError, match="regenerate the header"):
        cj.check_quick(stale)
