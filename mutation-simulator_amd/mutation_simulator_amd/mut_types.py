"""Mutation type identifiers -- numerically identical to the reference enum (mut_types.py:4-12).

The integer values cross the C-ABI (``msim_record.type``), so they are spelled out.
"""
from enum import Enum


class MutType(Enum):
    SN = 1   # single nucleotide polymorphism
    IN = 2   # insertion
    DE = 3   # deletion
    DU = 4   # tandem duplication
    IV = 5   # inversion
    TL = 6   # translocation: excised copy
    TLI = 7  # translocation: insert site
