"""Deterministic synthetic FASTA inputs shared by the golden generator and the tests.

Nothing here comes from the reference: inputs are produced by this generator (NumPy's legacy
``RandomState`` stream is frozen by NumPy policy, so they are reproducible on any box).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_bases(length: int, seed: int) -> np.ndarray:
    """uint8 array of i.i.d. uniform A/C/G/T (``b"ACGT"[RandomState(seed).randint(0, 4, L)]``)."""
    return ACGT[np.random.RandomState(seed).randint(0, 4, length)]


def decorate(bases: np.ndarray, seed: int, n_runs: int = 3, iupac: int = 40,
             lower: int = 3) -> np.ndarray:
    """Sprinkle N runs, IUPAC ambiguity codes ('-' included) and lower-case stretches."""
    rs = np.random.RandomState(seed)
    out = bases.copy()
    L = len(out)
    for _ in range(n_runs):
        a = int(rs.randint(0, max(1, L - 200)))
        out[a:a + int(rs.randint(5, 150))] = ord("N")
    codes = np.frombuffer(b"KSYMWRBDHV-N", dtype=np.uint8)
    pos = rs.randint(0, L, iupac)
    out[pos] = codes[rs.randint(0, len(codes), iupac)]
    for _ in range(lower):
        a = int(rs.randint(0, max(1, L - 500)))
        b = a + int(rs.randint(10, 400))
        seg = out[a:b]
        is_up = (seg >= 65) & (seg <= 90)
        seg[is_up] += 32
    return out


def fasta_bytes(contigs) -> bytes:
    """``contigs``: iterable of (defline_without_gt, uint8 bases, bases_per_line)."""
    parts: list[bytes] = []
    for defline, bases, bpl in contigs:
        parts.append(b">" + defline.encode() + b"\n")
        raw = bases.tobytes()
        for i in range(0, len(raw), bpl):
            parts.append(raw[i:i + bpl] + b"\n")
    return b"".join(parts)


def build_input(spec: dict) -> bytes:
    """Build FASTA bytes from a JSON-able spec:
    {"contigs": [{"defline": str, "length": int, "bpl": int, "seed": int, "decorate": bool}]}
    """
    contigs = []
    for c in spec["contigs"]:
        b = random_bases(c["length"], c["seed"])
        if c.get("decorate"):
            b = decorate(b, c["seed"] + 7919)
        if "literal" in c:
            b = np.frombuffer(c["literal"].encode(), dtype=np.uint8)
        contigs.append((c["defline"], b, c["bpl"]))
    return fasta_bytes(contigs)


def write_input(spec: dict, path: Path) -> Path:
    path.write_bytes(build_input(spec))
    return path
