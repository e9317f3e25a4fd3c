"""Default option values (reference defaults.py:7-42)."""
import sys
from pathlib import Path

from .mut_types import MutType

_tty = sys.stdout.isatty() and sys.stderr.isatty()


class Defaults:
    OUTBASE = Path(".")
    IGNORE_WARNINGS = False
    QUIET = False
    NO_COLOR = not _tty
    NO_PROGRESS = not _tty

    SPECIES_NAME = "Unknown"
    ASSEMBLY_NAME = "Unknown"
    SAMPLE_NAME = "Unknown"

    TITV = 1
    RATE = 0
    BLOCK = 1
    MINLEN = 1
    MAXLEN = 2
    IV_MINLEN = 2
    IV_MAXLEN = 3

    # insertion order as in the reference (SN, IN, DE, IV, DU, TL, TLI)
    MUT_BLOCK = dict.fromkeys((MutType.SN, MutType.IN, MutType.DE, MutType.IV, MutType.DU,
                               MutType.TL, MutType.TLI), BLOCK)
