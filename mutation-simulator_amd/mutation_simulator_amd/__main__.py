"""``mutation-simulator file {args,rmt}`` (reference __main__.py:34-111) on an MI355X."""
from __future__ import annotations

import json
import random
import sys
from timeit import default_timer as timer

from . import *  # noqa: F401,F403  (same style of namespace as the reference entry point)
from ._ffi import MsimError, MsimUnsupported

_INIT_ERRORS = (FileNotFoundError, ITNotEnoughAvailChromsError, RatesTooHighError, RatesTooLowError,
                FastaIndexingError, FastaNotFoundError, ItRateTooHighError, ItRateTooLowError,
                RMTParseError, MissingLengthError, MinimumLengthTooLowError, TitvTooLowError,
                ChromNotExistError, RangeDefinitionOutOfBoundsError, FastaDuplicateHeaderError,
                MinimumLengthHigherThanMaximumError)


STAGES: dict = {}          # wall seconds of the last run's stages (--bench-json: cli_s)


def initialize(argv=None):
    t0 = timer()
    args = get_args(argv)
    STAGES["parse_args"] = timer() - t0
    if (args.gpus or 1) <= 1:
        try:                       # the GPU comes up while the FASTA is read and indexed (never in the parent of --gpus N)
            from ._ffi import warm_up_async
            warm_up_async(args.device or 0, pin=True)
        except Exception:  # noqa: BLE001  (no library: Mutator reports it properly)
            pass
    try:
        t0 = timer()
        fasta = load_fasta(args.infile)
        STAGES["load_index"] = timer() - t0
        if args.mode == "args":
            sim = SimulationSettings.from_args(args, fasta, args.ignore_warnings)
        elif args.mode == "it":
            sim = SimulationSettings.from_it(args.interchromosomalrate, fasta, args.ignore_warnings)
        else:
            sim = SimulationSettings.from_rmt(args.rmtfile, fasta, args.ignore_warnings)
    except _INIT_ERRORS as e:
        exit_with_error(e, args.no_color)
    if not args.ignore_warnings:
        warn_user(args, sim)
    return args, fasta, sim


def warn_user(args, sim):
    if sim.fasta and args.infile.name != sim.fasta:
        print_warning("Fasta filename does not match RMT", args.no_color)
    if sim.md5 and get_md5(args.infile) != sim.md5:
        print_warning("Fasta md5 hash does not match RMT", args.no_color)


def main(argv=None):
    start = timer()
    args, fasta, sim = initialize(argv)
    loaded = timer()
    if args.seed is not None:
        import numpy
        random.seed(args.seed)
        numpy.random.seed(args.seed)
    if sim.has_mutations:
        try:
            t0 = timer()
            mutator = Mutator(args, fasta, sim)
            t1 = timer()
            try:
                mutator.mutate()
            finally:
                t2 = timer()
                mutator.close()            # (also on the reference's KeyError / ValueError: the files are complete as far as they go)
            fasta.close()
            if args.bench_json:
                stats = dict(mutator.stats)
                from . import mutator as _m
                stats["replanned_contigs"] = _m.REPLANNED_CONTIGS     # device window overflows recovered on the host (expected: 0)
                stats["cli_s"] = {"load_index_settings": round(loaded - start, 4), "mutate_and_write": round(timer() - loaded, 4),
                                  "parse_args": round(STAGES.get("parse_args", 0.0), 4), "load_index": round(STAGES.get("load_index", 0.0), 4),
                                  "open_writers": round(t1 - t0, 4), "mutate": round(t2 - t1, 4), "close": round(timer() - t2, 4)}
                args.bench_json.write_text(json.dumps(stats, indent=1) + "\n")
        except (FastaWriterError, VcfWriterError, MsimError) as e:
            exit_with_error(e, args.no_color)
    if sim.has_it:                         # the second pass reads what the first one wrote (reference __main__.py:88-102)
        if sim.has_mutations:
            try:
                fasta = load_fasta(args.outfasta)
            except (FastaDuplicateHeaderError, FastaIndexingError, FastaNotFoundError) as e:
                exit_with_error(e, args.no_color)
        try:
            it_mutator = ITMutator(args, fasta, sim)
            try:
                it_mutator.mutate()
            finally:
                it_mutator.close()
            fasta.close()
        except (FastaWriterError, BedpeWriterError, MsimError) as e:
            exit_with_error(e, args.no_color)
    runtime = round(timer() - start, 4)
    if not args.quiet:
        print_success(f"Mutation-Simulator finished in: {runtime}s", args.no_color)


if __name__ == "__main__":
    main()
