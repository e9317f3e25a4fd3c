#!/usr/bin/env python3
"""Generates csrc/mt_jump_table.h: GF(2) jump-ahead polynomials for MT19937.

The reference draws every random number from two sequential MT19937 streams (CPython `random`,
NumPy `RandomState`).  To produce those exact streams on a GPU, thousands of stream chunks have to
be generated concurrently, which needs the generator state at the start of every chunk.  For a
linear generator with characteristic polynomial phi, the state J steps ahead is g_J(A) applied to
the current state, g_J = x^J mod phi (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer 2008).
With the raw output sequence x[t] this reads, word for word,

    x[J + m] = XOR over { i : coefficient i of g_J is 1 } of x[i + m]

so a jump is a GF(2) convolution of ~20 k freshly generated raw words with the polynomial -- a good
GPU job.  This script computes phi (Berlekamp-Massey on one output bit), then the polynomials of a
MIXED-RADIX cascade: level l turns n_l = prod(RADIX[:l]) chunk-start states into n_l * RADIX[l] by applying,
to every one of them, the jumps m * n_l * CHUNK for m = 1 .. RADIX[l]-1 -- all of one level in a single
kernel launch, so the cascade is len(RADIX) dependent launches deep (a radix-2 doubling cascade with the
same reach was 13 deep, and every small level costs ~250 us of latency on the critical path of a step).
g_{l,m} = x^(m * n_l * CHUNK) mod phi; each is checked against plain sequential generation or against
the composition of two smaller jumps.  Pure integer arithmetic on Python ints; runs in about a minute.
"""
from __future__ import annotations

import sys
from pathlib import Path

N, M = 624, 397
DEG = 19937
CHUNK_BLOCKS = 256                 # one stream chunk = 256 regenerations = 159 744 words
CHUNK = N * CHUNK_BLOCKS
# 8192 chunks = 1.3 G words per cascade, 3 dependent launches.  The first level is the wide one: a jump is one 88 KB workgroup, a
# CU holds one, so 255 jumps are ONE round of the device (~65 us) -- and the 689 jumps of a 3 Gb -sn 0.01 session (255 + 2 x 256)
# three rounds in two launches (rounds 1-4: [16, 16, 16, 2] -- 15 + 240 + 512 jumps in three launches, the last one beside the
# first contigs' chains).  Sessions of up to 41 M words need the first level only.
RADIX = [256, 4, 8]


def twist(a, b, c):
    y = (a & 0x80000000) | (b & 0x7FFFFFFF)
    return c ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)


def raw_sequence(state, count):
    """x[0..623] = state; returns x[0 .. count)."""
    x = list(state)
    while len(x) < count:
        t = len(x) - N
        x.append(twist(x[t], x[t + 1], x[t + M]))
    return x[:count]


def berlekamp_massey(bits):
    """Minimal LFSR connection polynomial C (as int, bit i = coeff of x^i) and its length."""
    c, b = 1, 1
    L, m = 0, 1
    s = 0                                  # s holds bits seen so far, newest at bit 0 (reversed window)
    for n, bit in enumerate(bits):
        s = (s << 1) | bit
        # discrepancy = sum_{i=0..L} c_i * bits[n-i]; bits[n-i] is bit i of s
        d = bin(c & s & ((1 << (L + 1)) - 1)).count("1") & 1
        if d:
            t = c
            c ^= b << m
            if 2 * L <= n:
                L = n + 1 - L
                b = t
                m = 1
            else:
                m += 1
        else:
            m += 1
    return c, L


def poly_square(a: int) -> int:
    """Square in GF(2)[x]: spread the bits."""
    return int("0".join(bin(a)[2:]), 2) if a else 0


def poly_mod(a: int, phi: int, deg: int) -> int:
    while True:
        bl = a.bit_length()
        if bl <= deg:
            return a
        a ^= phi << (bl - 1 - deg)


def poly_powx(e: int, phi: int, deg: int) -> int:
    """x^e mod phi."""
    r = 1
    for bit in bin(e)[2:]:
        r = poly_mod(poly_square(r), phi, deg)
        if bit == "1":
            r = poly_mod(r << 1, phi, deg)
    return r


def apply_jump(g: int, x):
    """Window J ahead from x[0 ..] : out[m] = XOR_{i in supp(g)} x[i + m]."""
    idx = [i for i in range(g.bit_length()) if (g >> i) & 1]
    out = []
    for m in range(N):
        acc = 0
        for i in idx:
            acc ^= x[i + m]
        out.append(acc)
    return out


def main():
    out_path = Path(__file__).resolve().parent.parent / "csrc" / "mt_jump_table.h"
    # a state with plenty of bits set
    import random
    random.seed(20240607)
    state = [random.getrandbits(32) for _ in range(N)]
    seq = raw_sequence(state, N + 2 * DEG + 64)
    bits = [(w >> 7) & 1 for w in seq[N:N + 2 * DEG + 40]]
    c, L = berlekamp_massey(bits)
    assert L == DEG, L
    # connection polynomial C(x) = sum c_i x^i with s_n = sum_{i>=1} c_i s_{n-i}; the characteristic
    # polynomial is its reciprocal: phi(x) = x^L * C(1/x)
    phi = int(bin(c)[2:].zfill(L + 1)[::-1][::-1], 2)
    phi = sum(((c >> i) & 1) << (L - i) for i in range(L + 1))
    assert phi.bit_length() == DEG + 1
    # sanity: phi annihilates the sequence: XOR_{i in supp(phi)} x[t+i] == 0 for t >= 1
    supp = [i for i in range(DEG + 1) if (phi >> i) & 1]
    for t in (1, 2, 17):
        acc = 0
        for i in supp:
            acc ^= seq[t + i]
        assert acc == 0, "characteristic polynomial check failed"
    print(f"phi: degree {DEG}, weight {len(supp)}", file=sys.stderr)

    polys = []                                             # level-major: level l holds RADIX[l]-1 polynomials
    stride = 1
    by_mult = {}                                           # multiple of CHUNK -> polynomial
    for radix in RADIX:
        for m in range(1, radix):
            g = poly_powx(CHUNK * m * stride, phi, DEG)
            polys.append(g)
            by_mult[m * stride] = g
        stride *= radix
    # verify the smallest jumps against sequential generation
    long_seq = raw_sequence(state, 3 * CHUNK + DEG + 2 * N)
    for mult in (1, 2, 3):
        J = CHUNK * mult
        got = apply_jump(by_mult[mult], long_seq)
        want = long_seq[J:J + N]
        assert got[1:] == want[1:] and (got[0] ^ want[0]) & 0x80000000 == 0, f"jump x{mult} mismatch"
    # every other polynomial: x^(a+b) = x^a * x^b mod phi, checked as "jump by a, then by b" == "jump by a+b"
    # on the first words of the window (apply_jump needs DEG + N words after the start: regenerate from the window)
    def jump_window(g, window):
        seq2 = raw_sequence(window, DEG + 2 * N)
        idx = [i for i in range(g.bit_length()) if (g >> i) & 1]
        out = []
        for m_ in range(8):
            acc = 0
            for i in idx:
                acc ^= seq2[i + m_]
            out.append(acc)
        return out
    def full_window(g, window):
        seq2 = raw_sequence(window, DEG + 2 * N)
        return apply_jump(g, seq2)
    mults = sorted(by_mult)
    for mult in mults:
        if mult <= 3:
            continue
        # split into two known multiples
        a_ = max(x for x in mults if x < mult and (mult - x) in by_mult)
        b_ = mult - a_
        mid = full_window(by_mult[a_], state)
        mid[0] = (mid[0] & 0x80000000) | (mid[0] & 0x7FFFFFFF)
        got = jump_window(by_mult[b_], mid)
        want = jump_window(by_mult[mult], state)
        assert got[1:] == want[1:], f"composition check failed for x{mult} = x{a_} + x{b_}"
    print(f"{len(polys)} jump polynomials verified (sequential generation / composition)", file=sys.stderr)

    words_per_poly = (DEG + 31) // 32
    with open(out_path, "w") as fh:
        fh.write("// GENERATED by tools/gen_mt_jump.py -- do not edit.\n")
        fh.write("// MT19937 jump-ahead polynomials of a mixed-radix cascade: level l (n_l = product of the radices below it)\n")
        fh.write("// holds g_{l,m}(x) = x^(m * n_l * CHUNK) mod phi(x) for m = 1 .. RADIX[l]-1, level-major; little-endian\n")
        fh.write("// 32-bit limbs (bit i of limb j = coefficient of x^(32 j + i)).\n")
        fh.write("#pragma once\n#include <stdint.h>\n\nnamespace msim {\n")
        fh.write(f"constexpr int MT_CHUNK_BLOCKS = {CHUNK_BLOCKS};\n")
        fh.write(f"constexpr int MT_CHUNK_WORDS = {CHUNK};\n")
        fh.write(f"constexpr int MT_JUMP_LEVELS = {len(RADIX)};\n")
        fh.write("constexpr int MT_JUMP_RADIX[MT_JUMP_LEVELS] = {" + ", ".join(str(r) for r in RADIX) + "};\n")
        fh.write(f"constexpr int MT_JUMP_POLYS = {len(polys)};\n")
        total = 1
        for r in RADIX:
            total *= r
        fh.write(f"constexpr int MT_JUMP_MAX_CHUNKS = {total};\n")
        fh.write(f"constexpr int MT_POLY_WORDS = {words_per_poly};\n")
        fh.write(f"constexpr int MT_POLY_DEG = {DEG};\n")
        fh.write("static const uint32_t MT_JUMP_POLY[MT_JUMP_POLYS][MT_POLY_WORDS] = {\n")
        for g in polys:
            limbs = [(g >> (32 * j)) & 0xFFFFFFFF for j in range(words_per_poly)]
            fh.write("  {")
            for j, w in enumerate(limbs):
                if j % 8 == 0:
                    fh.write("\n    ")
                fh.write(f"0x{w:08x}u,")
            fh.write("\n  },\n")
        fh.write("};\n}  // namespace msim\n")
    print(f"wrote {out_path} ({len(polys)} polynomials, weights "
          f"{[bin(g).count('1') for g in polys[:4]]}...)", file=sys.stderr)


if __name__ == "__main__":
    main()
