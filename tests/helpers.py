"""Shared test helpers (golden loading, FASTA parsing for tests, masking)."""
from __future__ import annotations

import hashlib
import json
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
CASES = GOLDEN / "cases"


def sha256(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def load_json(name: str):
    return json.loads((GOLDEN / name).read_text())


def case_meta(name: str) -> dict:
    return json.loads((CASES / name / "meta.json").read_text())


def all_case_names() -> list[str]:
    return sorted(p.name for p in CASES.iterdir() if (p / "meta.json").exists())


def case_input_bytes(meta: dict) -> bytes:
    import inputs as gin
    b = gin.build_input(meta["input_spec"])
    assert sha256(b) == meta["input_sha256"], "synthetic input generator drifted"
    return b


def parse_fasta_bytes(data: bytes):
    """Minimal FASTA parser for tests: list of {name,long_name,lenc,bases(np.uint8 upper)}."""
    contigs = []
    cur = None
    for line in data.split(b"\n"):
        line = line.rstrip(b"\r")
        if line.startswith(b">"):
            long_name = line[1:].decode()
            cur = {"name": long_name.split()[0] if long_name.split() else "",
                   "long_name": long_name, "lenc": None, "chunks": []}
            contigs.append(cur)
        elif line and cur is not None:
            if cur["lenc"] is None:
                cur["lenc"] = len(line)
            cur["chunks"].append(line)
    for c in contigs:
        raw = b"".join(c.pop("chunks")).upper()
        c["bases"] = np.frombuffer(raw, dtype=np.uint8).copy()
        if c["lenc"] is None:
            c["lenc"] = 0
    return contigs


def mask_vcf(b: bytes) -> bytes:
    out = []
    for line in b.split(b"\n"):
        if line.startswith(b"##filedate="):
            line = b"##filedate=MASKED"
        out.append(line)
    return b"\n".join(out)
