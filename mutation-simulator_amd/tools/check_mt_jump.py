#!/usr/bin/env python3
"""Checks csrc/mt_jump_table.h -- 2 MB of generated data the device streams are built from -- without trusting its generator's
own run: a stale or hand-edited header must not ship.

    python3 mutation-simulator_amd/tools/check_mt_jump.py            # the quick check `__graft_entry__.build()` runs

`check_quick` (numpy only, ~1 s): the header's constants against `gen_mt_jump.py`'s; the first three polynomials against plain
sequential MT19937 generation (jump(state, J) == J steps); a sample of the others -- the last of every level and a few seeded
picks -- by composition (jump by a, then by b == jump by a + b, with a and b taken from the table).
`tests/test_jump_table.py` (CPU tier) checks ALL 265 polynomials against sequential generation of 1.1 G words by the oracle's
MT19937 (the tests may use the oracle; a build may not).

What a jump is (gen_mt_jump.py): with the raw sequence x[t] of a state x[0 .. 624), the window J words ahead is
    x[J + m] = XOR over { i : coefficient i of g_J is 1 } of x[i + m],   g_J = x^J mod phi   (phi: MT19937's characteristic polynomial)
-- the bits of word 0 below its top bit are not part of the generator's state and are ignored."""
from __future__ import annotations

import re
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
HEADER = HERE.parent / "csrc" / "mt_jump_table.h"
N, M = 624, 397


def parse_header(path: Path = HEADER):
    """(constants dict, polynomials as uint32[n_polys, words])."""
    text = path.read_text()
    consts = {k: int(v) for k, v in re.findall(r"constexpr int (MT_\w+) = (\d+);", text)}
    m = re.search(r"MT_JUMP_RADIX\[MT_JUMP_LEVELS\] = \{([^}]*)\}", text)
    consts["MT_JUMP_RADIX"] = [int(x) for x in m.group(1).split(",")]
    body = text[text.index("MT_JUMP_POLY[MT_JUMP_POLYS][MT_POLY_WORDS]"):]
    words = np.array([int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{8})u", body)], dtype=np.uint32)
    n_polys, n_words = consts["MT_JUMP_POLYS"], consts["MT_POLY_WORDS"]
    if words.size != n_polys * n_words:
        raise AssertionError(f"{path.name}: {words.size} words for {n_polys} x {n_words}")
    return consts, words.reshape(n_polys, n_words)


def regenerate(x: np.ndarray) -> np.ndarray:
    """The next 624 raw words behind the block x (one MT19937 state regeneration, vectorised over its three independent runs)."""
    new = np.empty(N, dtype=np.uint32)
    up, lo, mag = np.uint32(0x80000000), np.uint32(0x7FFFFFFF), np.uint32(0x9908B0DF)

    def tw(a, b, c):
        y = (a & up) | (b & lo)
        return c ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), mag, np.uint32(0))
    new[0:N - M] = tw(x[0:N - M], x[1:N - M + 1], x[M:N])
    for a in range(N - M, N - 1, N - M):
        b = min(a + (N - M), N - 1)
        new[a:b] = tw(x[a:b], x[a + 1:b + 1], new[a - (N - M):b - (N - M)])
    new[N - 1] = tw(x[N - 1:N], new[0:1], new[M - 1:M])[0]
    return new


def raw_sequence(state: np.ndarray, count: int) -> np.ndarray:
    """x[0 .. count) with x[0 .. 624) = state."""
    blocks = [np.asarray(state, dtype=np.uint32)]
    while len(blocks) * N < count:
        blocks.append(regenerate(blocks[-1]))
    return np.concatenate(blocks)[:count]


def support(poly_words: np.ndarray) -> np.ndarray:
    bits = np.unpackbits(poly_words.astype("<u4").view(np.uint8), bitorder="little")
    return np.flatnonzero(bits)


def apply_jump(poly_words: np.ndarray, seq: np.ndarray, n_out: int = N) -> np.ndarray:
    """out[m] = XOR_{i in supp(g)} seq[i + m], m < n_out (seq: at least deg + n_out raw words from the window's start)."""
    idx = support(poly_words)
    win = np.lib.stride_tricks.sliding_window_view(seq, n_out)
    return np.bitwise_xor.reduce(win[idx], axis=0)


def same_state(a: np.ndarray, b: np.ndarray) -> bool:
    return bool(np.array_equal(a[1:], b[1:]) and ((int(a[0]) ^ int(b[0])) & 0x80000000) == 0)


def multiples(radix):
    """Chunk multiple of every polynomial, in the header's order (level-major)."""
    out, stride = [], 1
    for r in radix:
        out += [m * stride for m in range(1, r)]
        stride *= r
    return out


def check_constants(consts):
    sys.path.insert(0, str(HERE))
    import gen_mt_jump as gen
    want = {"MT_CHUNK_BLOCKS": gen.CHUNK_BLOCKS, "MT_CHUNK_WORDS": gen.CHUNK, "MT_JUMP_LEVELS": len(gen.RADIX),
            "MT_JUMP_RADIX": list(gen.RADIX), "MT_JUMP_POLYS": sum(r - 1 for r in gen.RADIX),
            "MT_JUMP_MAX_CHUNKS": int(np.prod(gen.RADIX)), "MT_POLY_WORDS": (gen.DEG + 31) // 32, "MT_POLY_DEG": gen.DEG}
    for k, v in want.items():
        if consts.get(k) != v:
            raise AssertionError(f"mt_jump_table.h: {k} = {consts.get(k)}, tools/gen_mt_jump.py says {v} -- regenerate the header")


def check_quick(path: Path = HEADER, n_random: int = 6, seed: int = 20240607) -> int:
    """Raises AssertionError on the first polynomial that is not x^(multiple * CHUNK) mod phi; returns how many were checked."""
    consts, P = parse_header(path)
    check_constants(consts)
    chunk, deg = consts["MT_CHUNK_WORDS"], consts["MT_POLY_DEG"]
    if any(int(p[-1]) >> (deg % 32) for p in P):
        raise AssertionError("mt_jump_table.h: a polynomial of degree >= 19937")
    mult = multiples(consts["MT_JUMP_RADIX"])
    index_of = {m: i for i, m in enumerate(mult)}
    rs = np.random.RandomState(seed)
    state = rs.randint(0, 2 ** 32, size=N, dtype=np.uint64).astype(np.uint32)
    seq = raw_sequence(state, 3 * chunk + deg + 2 * N)
    checked = 0
    for m in (1, 2, 3):                                   # against plain sequential generation
        got = apply_jump(P[index_of[m]], seq)
        if not same_state(got, seq[m * chunk:m * chunk + N]):
            raise AssertionError(f"mt_jump_table.h: polynomial {index_of[m]} is not a jump by {m} chunk(s) (sequential generation disagrees)")
        checked += 1
    # composition: jump by a, then by b == jump by a + b (x^(a+b) = x^a x^b mod phi), both parts from the table
    picks, at = [], 0
    for r in consts["MT_JUMP_RADIX"]:
        at += r - 1
        picks.append(at - 1)                               # the last polynomial of every level
    picks += [int(i) for i in rs.choice(len(mult), size=n_random, replace=False)]
    head = seq[:deg + 2 * N]
    for i in picks:
        m = mult[i]
        if m <= 3:
            continue
        a = max(x for x in index_of if x < m and (m - x) in index_of)
        mid = apply_jump(P[index_of[a]], head)
        mid[0] &= np.uint32(0x80000000)
        got = apply_jump(P[index_of[m - a]], raw_sequence(mid, deg + 2 * N), 8)
        want = apply_jump(P[i], head, 8)
        if not np.array_equal(got[1:], want[1:]):
            raise AssertionError(f"mt_jump_table.h: polynomial {i} (x{m} chunks) != jump x{a} then x{m - a}")
        checked += 1
    return checked


if __name__ == "__main__":
    n = check_quick()
    print(f"mt_jump_table.h: constants match gen_mt_jump.py, {n} polynomials verified (sequential generation / composition)")
