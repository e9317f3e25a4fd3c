"""Console helpers and FASTA loading (reference util.py:17-91; message texts are user-visible
behaviour and therefore identical)."""
from __future__ import annotations

import hashlib
import sys
from pathlib import Path

from .colors import Colors
from .fasta_io import Fasta


class FastaDuplicateHeaderError(Exception):
    """The input FASTA holds the same record name twice."""


def _paint(text: str, colour: str, no_color: bool) -> str:
    return text if no_color else f"{colour}{text}{Colors.norm}"


def exit_with_error(e: Exception, no_color: bool):
    print(_paint(f"ERROR: {e}", Colors.error, no_color), file=sys.stderr)
    sys.exit(1)


def format_warning(msg: str, no_color: bool) -> str:
    return _paint(f"WARNING: {msg}", Colors.warn, no_color)


def print_warning(msg: str, no_color: bool):
    print(format_warning(msg, no_color), file=sys.stderr)


def print_success(msg, no_color):
    print(_paint(msg, Colors.ok, no_color))


def get_md5(fname: Path) -> str:
    digest = hashlib.md5()
    with open(fname, "rb") as fh:
        while True:
            block = fh.read(1 << 20)
            if not block:
                break
            digest.update(block)
    return digest.hexdigest()


def load_fasta(fname: Path) -> Fasta:
    """Load the input genome (replaces the reference's pyfaidx call, util.py:77-91)."""
    try:
        return Fasta(str(Path(fname).resolve()))
    except ValueError:
        raise FastaDuplicateHeaderError(f"Fasta {fname} contains duplicate header")
