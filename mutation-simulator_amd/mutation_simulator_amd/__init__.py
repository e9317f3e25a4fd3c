"""MI355X-native drop-in for Mutation-Simulator's ``args`` / ``rmt`` mutation pass.

Exports the names the reference package exports (reference __init__.py:1-18) for the path this
build covers; the work behind ``Mutator`` runs as hand-written HIP kernels through libmsim.
"""
from ._version import __version__
from .argument_parser import get_args
from .bedpe_writer import BedpeWriter, BedpeWriterError
from .colors import Colors
from .fasta_io import FastaIndexingError, FastaNotFoundError
from .fasta_writer import FastaWriter, FastaWriterError
from .it_mutator import ITMutator
from .mut_types import MutType
from .mutator import Mutation, Mutator
from .rmt import (ChromNotExistError, ITNotEnoughAvailChromsError, ItRateTooHighError,
                  ItRateTooLowError, MinimumLengthHigherThanMaximumError,
                  MinimumLengthTooLowError, MissingLengthError, RangeDefinitionOutOfBoundsError,
                  RatesTooHighError, RatesTooLowError, RMTParseError, SimulationSettings,
                  TitvTooLowError)
from .util import (FastaDuplicateHeaderError, exit_with_error, format_warning, get_md5,
                   load_fasta, print_success, print_warning)
from .vcf_writer import VcfRecord, VcfWriter, VcfWriterError
