"""ctypes binding of libmsim.so (include/msim.h) -- the only way the host package reaches the GPU.

There is no CPU fallback: if the library is missing or no MI355X is visible, ``Engine()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
LIB_CANDIDATES = [PKG_DIR.parent / "lib" / "libmsim.so"]

OK, ERR_ARG, ERR_HIP, ERR_VALUE, ERR_KEY, ERR_UNSUPPORTED, ERR_NOMEM, ERR_IO = range(8)
PLAN_AUTO, PLAN_HOST, PLAN_GPU = 0, 1, 2
RNG_FAST = 4           # Engine flag: counter-based generator, NOT stream-compatible with the reference (msim.h: MSIM_RNG_FAST)


class MsimError(RuntimeError):
    """A libmsim call failed (HIP error, bad argument, unsupported input)."""


class MsimUnsupported(MsimError):
    """Valid for the reference, outside this build's scope (translocations, >= 4 GiB contigs)."""


class Record(C.Structure):           # msim_record, 16 bytes
    _fields_ = [("pos", C.c_uint32), ("stop", C.c_uint32), ("extra", C.c_uint32),
                ("type", C.c_uint8), ("aux", C.c_uint8), ("rsv", C.c_uint16)]


RECORD_DTYPE = np.dtype([("pos", "<u4"), ("stop", "<u4"), ("extra", "<u4"), ("type", "u1"),
                         ("aux", "u1"), ("rsv", "<u2")])


class Range(C.Structure):            # msim_range
    _fields_ = [("start", C.c_int64), ("stop", C.c_int64), ("k", C.c_int64),
                ("setsize", C.c_int64), ("n_types", C.c_int32), ("types", C.c_int32 * 8),
                ("cdf_thr", C.c_uint64 * 8), ("min_len", C.c_int64 * 8), ("max_len", C.c_int64 * 8)]


class SettingsDesc(C.Structure):     # msim_settings_desc
    _fields_ = [("rate_sum", C.c_double), ("n_types", C.c_int32), ("types", C.c_int32 * 8), ("chances", C.c_double * 8),
                ("min_len", C.c_int64 * 8), ("max_len", C.c_int64 * 8)]


class Params(C.Structure):           # msim_params
    _fields_ = [("block", C.c_int64 * 8), ("ti_lim", C.c_uint64)]


class BatchContig(C.Structure):      # msim_batch_contig
    _fields_ = [("body", C.c_void_p), ("body_bytes", C.c_uint64), ("n_bases", C.c_uint64),
                ("lenc", C.c_uint32), ("lenb", C.c_uint32), ("ranges", C.POINTER(Range)),
                ("n_ranges", C.c_int32), ("name", C.c_char_p), ("header", C.c_char_p),
                ("name_len", C.c_uint32), ("header_len", C.c_uint32)]


# the same two structs as numpy record dtypes: a batch over an assembly builds its tables with array operations
RANGE_DTYPE = np.dtype([("start", "<i8"), ("stop", "<i8"), ("k", "<i8"), ("setsize", "<i8"), ("n_types", "<i4"),
                        ("types", "<i4", (8,)), ("_pad", "<i4"), ("cdf_thr", "<u8", (8,)), ("min_len", "<i8", (8,)),
                        ("max_len", "<i8", (8,))])
BATCH_CONTIG_DTYPE = np.dtype([("body", "<u8"), ("body_bytes", "<u8"), ("n_bases", "<u8"), ("lenc", "<u4"), ("lenb", "<u4"),
                               ("ranges", "<u8"), ("n_ranges", "<i4"), ("_pad", "<i4"), ("name", "<u8"), ("header", "<u8"),
                               ("name_len", "<u4"), ("header_len", "<u4")])


class Timing(C.Structure):           # msim_timing
    _fields_ = [("plan_host_ms", C.c_double), ("plan_gpu_ms", C.c_double),
                ("upload_ms", C.c_double), ("apply_ms", C.c_double),
                ("apply_kernel_ms", C.c_double), ("apply_launches", C.c_uint64),
                ("bytes_in", C.c_uint64), ("bytes_out", C.c_uint64), ("records", C.c_uint64),
                ("py_words", C.c_uint64), ("np_words", C.c_uint64),
                ("contigs_snp", C.c_uint64), ("contigs_svmix", C.c_uint64), ("contigs_hostcut", C.c_uint64),
                ("contigs_hostchain", C.c_uint64), ("contigs_host", C.c_uint64), ("contigs_batch", C.c_uint64),
                ("contigs_fast", C.c_uint64), ("stream_rebases", C.c_uint64),
                ("snp_samples_ahead", C.c_uint64),
                ("host_walk_run_ms", C.c_double), ("host_walk_wait_ms", C.c_double),
                ("host_walk_candidates", C.c_uint64), ("host_cut_words", C.c_uint64),
                ("snp_ahead_margin_permille", C.c_uint64)]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


# every symbol include/msim.h declares: (name, restype, argtypes)
_VP, _U8P, _U64P, _U32P, _IP = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), \
    C.POINTER(C.c_uint32), C.POINTER(C.c_int)
SYMBOLS = [
    ("msim_abi_version", C.c_int, []),
    ("msim_warm_up", C.c_int, [C.c_int]),
    ("msim_create", C.c_int, [C.c_int, C.c_uint32, C.POINTER(_VP)]),
    ("msim_destroy", None, [_VP]),
    ("msim_last_error", C.c_char_p, [_VP]),
    ("msim_device_name", C.c_int, [_VP, C.c_char_p, C.c_int]),
    ("msim_sync", C.c_int, [_VP]),
    ("msim_set_plan_mode", C.c_int, [_VP, C.c_uint32]),
    ("msim_seed", C.c_int, [_VP, _U32P, C.c_int, C.c_uint32]),
    ("msim_set_mt_state", C.c_int, [_VP, C.c_int, _U32P, C.c_int]),
    ("msim_get_mt_state", C.c_int, [_VP, C.c_int, _U32P, _IP]),
    ("msim_reserve_streams", C.c_int, [_VP, C.c_uint64, C.c_uint64]),
    ("msim_add_contig", C.c_int, [_VP, _VP, C.c_uint64, _IP]),
    ("msim_add_contig_synthetic", C.c_int, [_VP, C.c_uint64, C.c_uint64, _IP]),
    ("msim_contig_length", C.c_int, [_VP, C.c_int, _U64P]),
    ("msim_read_contig", C.c_int, [_VP, C.c_int, C.c_uint64, C.c_uint64, _VP]),
    ("msim_clear", C.c_int, [_VP]),
    ("msim_set_params", C.c_int, [_VP, C.POINTER(Params)]),
    ("msim_plan_contig", C.c_int, [_VP, C.c_int, C.POINTER(Range), C.c_int]),
    ("msim_plan_chain", C.c_int, [_VP, C.c_uint64, C.POINTER(Range), C.c_int]),
    ("msim_plan_was_empty", C.c_int, [_VP, C.c_int, _IP]),
    ("msim_apply_contig", C.c_int, [_VP, C.c_int]),
    ("msim_key_error", C.c_int, [_VP, C.c_int, _U8P, _U64P]),
    ("msim_result_sizes", C.c_int, [_VP, C.c_int, _U64P, _U64P, _U64P]),
    ("msim_fetch_sequence", C.c_int, [_VP, C.c_int, C.c_uint64, C.c_uint64, _VP]),
    ("msim_fetch_records", C.c_int, [_VP, C.c_int, _VP, _VP]),
    ("msim_result_checksum", C.c_int, [_VP, C.c_int, _U64P]),
    ("msim_release_result", C.c_int, [_VP, C.c_int]),
    ("msim_result_device_ptr", C.c_int, [_VP, C.c_int, _U64P, _U64P]),
    ("msim_render_vcf", C.c_int, [_VP, C.c_uint64, _VP, _VP, C.c_uint64, C.c_char_p, _VP,
                                  C.c_uint64, _U64P]),
    ("msim_render_vcf_device", C.c_int, [_VP, C.c_int, C.c_char_p, _VP, C.c_uint64, _U64P]),
    ("msim_fetch_sequence_framed", C.c_int, [_VP, C.c_int, C.c_uint32, _VP, C.c_uint64, _U64P]),
    ("msim_render_vcf_device_file", C.c_int, [_VP, C.c_int, C.c_char_p, C.c_int, C.c_uint64, _U64P]),
    ("msim_fetch_sequence_framed_file", C.c_int, [_VP, C.c_int, C.c_uint32, C.c_int, C.c_uint64, _U64P]),
    ("msim_file_wait", C.c_int, [_VP]),
    ("msim_device_host_cpus", C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    ("msim_batch_fetch_file", C.c_int, [_VP, C.c_int, C.c_uint64, C.c_int, C.c_uint64]),
    ("msim_add_contig_text", C.c_int, [_VP, _VP, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _IP]),
    ("msim_splice_contigs", C.c_int, [_VP, C.c_int, C.c_int, C.c_uint64, _U64P, _U64P, _IP]),
    ("msim_set_fast_key", C.c_int, [_VP, C.c_uint64]),
    ("msim_sample_min_distance", C.c_int, [_VP, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _VP]),
    ("msim_host_alloc", C.c_int, [_VP, C.c_uint64, C.POINTER(_VP)]),
    ("msim_host_free", C.c_int, [_VP, _VP]),
    ("msim_batch_run", C.c_int, [_VP, C.POINTER(BatchContig), C.c_int]),
    ("msim_batch_sizes", C.c_int, [_VP, C.c_int, _U64P, _U64P, C.POINTER(C.c_int32), _U64P]),
    ("msim_batch_fetch", C.c_int, [_VP, _VP, C.c_uint64, _VP, C.c_uint64]),
    ("msim_batch_view", C.c_int, [_VP, C.POINTER(C.c_void_p), _U64P, C.POINTER(C.c_void_p), _U64P, _U64P]),
    ("msim_batch_key_contig", C.c_int, [_VP, _IP]),
    ("msim_fasta_index", C.c_int, [_VP, C.c_uint64, _VP, C.c_uint64, _U64P]),
    ("msim_build_ranges", C.c_int, [_VP, C.c_int, _VP, _VP, _VP, C.c_int64, _VP]),
    ("msim_comm_unique_id", C.c_int, [_VP]),
    ("msim_comm_init", C.c_int, [_VP, _VP, C.c_int, C.c_int]),
    ("msim_comm_destroy", C.c_int, [_VP]),
    ("msim_planned_out_len", C.c_int, [_VP, C.c_int, _U64P, _IP]),
    ("msim_gather_to_root", C.c_int, [_VP, C.c_int, _IP, _IP, _U64P, _U64P, _U64P, C.c_int, _U64P]),
    ("msim_gather_plan", C.c_int, [C.c_int, _IP, _U64P, _U64P, _U64P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), _IP]),
    ("msim_gather_fetch", C.c_int, [_VP, C.c_uint64, C.c_uint64, _VP]),
    ("msim_stats", C.c_int, [_VP, C.POINTER(Timing)]),
    ("msim_reset_stats", C.c_int, [_VP]),
]

_lib = None


def library_path() -> Path:
    env = os.environ.get("MSIM_LIB")
    if env:
        return Path(env)
    for p in LIB_CANDIDATES:
        if p.exists():
            return p
    raise MsimError(
        f"libmsim.so not found (looked in {[str(p) for p in LIB_CANDIDATES]}); build it with "
        "`make -C mutation-simulator_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`")


def load():
    """dlopen libmsim.so and declare every prototype.  Never touches the GPU by itself."""
    global _lib
    if _lib is None:
        lib = C.CDLL(str(library_path()))
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.msim_abi_version() != 8:
            raise MsimError("libmsim ABI version mismatch")
        _lib = lib
    return _lib


def parse_cpulist(text: str) -> set:
    """A Linux cpulist ("0-63,128-191") as a set of ints."""
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def warm_up_async(device: int = 0, pin: bool = False):
    """Start the HIP runtime on a helper thread (the first HIP call of a process costs ~0.2 s) while the caller reads and
    indexes its input; returns the thread (join it, or just create the Engine: HIP initialisation is serialised inside).
    ``pin``: once the runtime is up, move the CALLING thread onto the CPUs of the GPU's NUMA node (what the CLI does:
    on a two-socket host the copies to and from the GPU then cross the inter-socket links once; ``MSIM_NO_PIN=1`` or a
    cpuset that does not contain those CPUs leaves it where it is)."""
    import threading
    lib = load()
    caller = threading.get_native_id()

    def work():
        lib.msim_warm_up(int(device))
        if not pin or os.environ.get("MSIM_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
            return
        try:
            buf = C.create_string_buffer(4096)
            if lib.msim_device_host_cpus(int(device), buf, 4096) != OK or not buf.value:
                return
            cpus = parse_cpulist(buf.value.decode()) & os.sched_getaffinity(caller)
            if cpus:
                os.sched_setaffinity(caller, cpus)
        except (OSError, ValueError):
            pass
    t = threading.Thread(target=work, name="msim-warm-up", daemon=True)
    t.start()
    return t


def _ptr(a: np.ndarray):
    return C.c_void_p(a.ctypes.data)


class Engine:
    """One libmsim context (one GPU).  Thin, typed wrappers -- no logic lives here."""

    def __init__(self, device: int = 0, flags: int = PLAN_AUTO):
        self.lib = load()
        self.h = C.c_void_p()
        if flags == PLAN_AUTO and os.environ.get("MSIM_PLAN_MODE"):      # diagnosis: 1 = host planner everywhere, 2 = device engines only
            flags = int(os.environ["MSIM_PLAN_MODE"])
        rc = self.lib.msim_create(device, flags, C.byref(self.h))
        if rc != OK:
            msg = self.lib.msim_last_error(None).decode()
            self.h = None
            raise MsimError(f"msim_create failed ({rc}): {msg}")
        self._pinned: list = []

    def close(self):
        if getattr(self, "h", None):
            for p in getattr(self, "_pinned", []):
                self.lib.msim_host_free(self.h, C.c_void_p(p))
            self._pinned = []
            self.lib.msim_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ errors
    def _check(self, rc: int, contig: int | None = None):
        if rc == OK:
            return
        msg = self.lib.msim_last_error(self.h).decode()
        if rc == ERR_VALUE:
            raise ValueError(msg)                      # the reference's own exception (util.py:104)
        if rc == ERR_KEY:
            base = C.c_uint8()
            pos = C.c_uint64()
            self.lib.msim_key_error(self.h, -1, C.byref(base),
                                    C.byref(pos))
            raise KeyError(chr(base.value))            # mutator.py:449-455
        if rc == ERR_UNSUPPORTED:
            raise MsimUnsupported(msg)
        raise MsimError(f"libmsim error {rc}: {msg}")

    # ------------------------------------------------------------------ device / rng
    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self._check(self.lib.msim_device_name(self.h, buf, 256))
        return buf.value.decode()

    def sync(self):
        self._check(self.lib.msim_sync(self.h))

    def seed(self, py_seed: int, np_seed: int):
        n = abs(int(py_seed))
        key = []
        while True:
            key.append(n & 0xFFFFFFFF)
            n >>= 32
            if not n:
                break
        arr = (C.c_uint32 * len(key))(*key)
        self._check(self.lib.msim_seed(self.h, arr, len(key), int(np_seed) & 0xFFFFFFFF))

    def set_mt_state(self, stream: int, mt, pos: int):
        arr = np.ascontiguousarray(mt, dtype=np.uint32)
        assert arr.shape == (624,)
        self._check(self.lib.msim_set_mt_state(self.h, stream, arr.ctypes.data_as(_U32P), int(pos)))

    def get_mt_state(self, stream: int):
        arr = np.zeros(624, dtype=np.uint32)
        pos = C.c_int()
        self._check(self.lib.msim_get_mt_state(self.h, stream, arr.ctypes.data_as(_U32P),
                                               C.byref(pos)))
        return arr, pos.value

    def set_fast_key(self, key: int):
        """Key of the counter-based generator of a ``RNG_FAST`` context (restarts its contig ordinal)."""
        self._check(self.lib.msim_set_fast_key(self.h, int(key) & 0xFFFFFFFFFFFFFFFF))

    def reserve_streams(self, py_words: int, np_words: int = 0):
        self._check(self.lib.msim_reserve_streams(self.h, int(py_words), int(np_words)))

    # ------------------------------------------------------------------ genome
    def add_contig(self, bases: np.ndarray) -> int:
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        cid = C.c_int()
        self._check(self.lib.msim_add_contig(self.h, _ptr(bases), bases.shape[0], C.byref(cid)))
        return cid.value

    def add_contig_text(self, body: np.ndarray, n_bases: int, lenc: int, lenb: int) -> int:
        """One FASTA record straight from file text (bytes after the header line); the device skips the
        line terminators and upper-cases."""
        body = np.ascontiguousarray(body, dtype=np.uint8)
        cid = C.c_int()
        self._check(self.lib.msim_add_contig_text(self.h, _ptr(body), body.shape[0], n_bases, lenc, lenb, C.byref(cid)))
        return cid.value

    def sample_min_distance(self, start: int, stop: int, k: int, d: int, setsize: int) -> np.ndarray:
        """``sample_with_minimum_distance`` (util.py:93-109) on this context's CPython stream; ValueError as CPython's."""
        out = np.empty(max(int(k), 0), dtype=np.int64)
        self._check(self.lib.msim_sample_min_distance(self.h, int(start), int(stop), int(k), int(d), int(setsize),
                                                      C.c_void_p(out.ctypes.data)))
        return out

    def splice_contigs(self, a: int, b: int, bp_a, bp_b) -> int:
        """Interchromosomal translocation of contig ``a`` with partner ``b`` (it_mutator.py:121-146): a new contig made of
        a's and b's segments between the breakpoints, taken alternately.  No breakpoints: a copy of ``a``."""
        bp_a = np.ascontiguousarray(bp_a, dtype=np.uint64)
        bp_b = np.ascontiguousarray(bp_b, dtype=np.uint64)
        if bp_a.shape != bp_b.shape or bp_a.ndim != 1:
            raise ValueError("both contigs need the same number of breakpoints")
        cid = C.c_int()
        n = int(bp_a.shape[0])
        self._check(self.lib.msim_splice_contigs(self.h, a, b, n, C.cast(C.c_void_p(bp_a.ctypes.data), _U64P) if n else None,
                                                 C.cast(C.c_void_p(bp_b.ctypes.data), _U64P) if n else None, C.byref(cid)))
        return cid.value

    def host_buffer(self, nbytes: int) -> np.ndarray:
        """A page-locked uint8 buffer owned by this engine (freed at ``close``): ingest / egress through it runs at the
        link's speed (no staging pass, no page faults of a fresh allocation)."""
        p = _VP()
        self._check(self.lib.msim_host_alloc(self.h, int(nbytes), C.byref(p)))
        self._pinned.append(p.value)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(int(nbytes), 1),))[:nbytes]

    def add_contig_synthetic(self, length: int, seed: int) -> int:
        cid = C.c_int()
        self._check(self.lib.msim_add_contig_synthetic(self.h, length, seed, C.byref(cid)))
        return cid.value

    def contig_length(self, contig: int) -> int:
        n = C.c_uint64()
        self._check(self.lib.msim_contig_length(self.h, contig, C.byref(n)))
        return n.value

    def read_contig(self, contig: int, offset: int = 0, n: int | None = None) -> np.ndarray:
        if n is None:
            n = self.contig_length(contig) - offset
        out = np.empty(n, dtype=np.uint8)
        self._check(self.lib.msim_read_contig(self.h, contig, offset, n, _ptr(out)))
        return out

    def clear(self):
        self._check(self.lib.msim_clear(self.h))

    # ------------------------------------------------------------------ plan / apply
    def set_params(self, params: Params):
        self._check(self.lib.msim_set_params(self.h, C.byref(params)))

    @staticmethod
    def range_table(ranges: list):
        """The ctypes array `plan_contig` passes on; a caller that plans the same contig repeatedly may build it once."""
        arr = (Range * max(len(ranges), 1))(*ranges)
        arr.n_ranges = len(ranges)
        return arr

    @staticmethod
    def _ranges_arg(ranges):
        """(pointer, count, keep-alive) of a range table given as a list of ``Range``, a ctypes array or a numpy
        ``RANGE_DTYPE`` table (``mutator.plan_table``)."""
        if isinstance(ranges, np.ndarray):
            if ranges.dtype != RANGE_DTYPE:
                raise MsimError("range table of the wrong dtype")
            ranges = np.ascontiguousarray(ranges)
            return C.cast(C.c_void_p(ranges.ctypes.data), C.POINTER(Range)), int(ranges.shape[0]), ranges
        arr = ranges if isinstance(ranges, C.Array) else Engine.range_table(ranges)
        return arr, getattr(arr, "n_ranges", len(arr)), arr

    def plan_contig(self, contig: int, ranges):
        ptr, n, keep = self._ranges_arg(ranges)
        self._check(self.lib.msim_plan_contig(self.h, contig, ptr, n), contig)
        del keep

    def plan_chain(self, length: int, ranges):
        """Advance both streams over a contig this process does not mutate (multi-GPU: a rank that does not own it)."""
        ptr, n, keep = self._ranges_arg(ranges)
        self._check(self.lib.msim_plan_chain(self.h, length, ptr, n))
        del keep

    def set_plan_mode(self, mode: int):
        """0 = AUTO, PLAN_HOST = force the sequential host planner, PLAN_GPU = force a device engine."""
        self._check(self.lib.msim_set_plan_mode(self.h, mode))

    def plan_was_empty(self, contig: int) -> bool:
        e = C.c_int()
        self._check(self.lib.msim_plan_was_empty(self.h, contig, C.byref(e)))
        return bool(e.value)

    def apply_contig(self, contig: int):
        self._check(self.lib.msim_apply_contig(self.h, contig), contig)

    def result_sizes(self, contig: int, applied: bool = True):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.msim_result_sizes(self.h, contig, C.byref(a) if applied else None,
                                               C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def fetch_sequence(self, contig: int, offset: int = 0, n: int | None = None) -> np.ndarray:
        if n is None:
            n = self.result_sizes(contig)[0] - offset
        out = np.empty(n, dtype=np.uint8)
        self._check(self.lib.msim_fetch_sequence(self.h, contig, offset, n, _ptr(out)))
        return out

    def fetch_records(self, contig: int):
        _, n_rec, n_pool = self.result_sizes(contig, applied=False)
        recs = np.zeros(n_rec, dtype=RECORD_DTYPE)
        pool = np.zeros(n_pool, dtype=np.uint8)
        self._check(self.lib.msim_fetch_records(self.h, contig, _ptr(recs), _ptr(pool)))
        return recs, pool

    def render_vcf_device(self, contig: int, seq_name: str, guess: int = 0) -> np.ndarray:
        """VCF record lines of the contig rendered on the device (uint8 array of text).  With a size ``guess`` the
        text comes back in ONE synchronising call when it fits (the library keeps the rendered text, so a second
        call with the right size only copies)."""
        need = C.c_uint64()
        name = seq_name.encode("utf-8", "replace")
        if guess > 0:
            out = np.empty(guess, dtype=np.uint8)
            rc = self.lib.msim_render_vcf_device(self.h, contig, name, _ptr(out), guess, C.byref(need))
            if rc == OK:
                return out[:need.value]
            if rc != ERR_ARG or need.value <= guess:
                self._check(rc, contig)
        else:
            self._check(self.lib.msim_render_vcf_device(self.h, contig, name, None, 0, C.byref(need)), contig)
        out = np.empty(need.value, dtype=np.uint8)
        if need.value:
            self._check(self.lib.msim_render_vcf_device(self.h, contig, name, _ptr(out), need.value, C.byref(need)), contig)
        return out

    def render_vcf_device_size(self, contig: int, seq_name: str) -> int:
        """Render the contig's VCF record lines on the device and report their size (the text stays in HBM for
        ``render_vcf_device_into``)."""
        need = C.c_uint64()
        self._check(self.lib.msim_render_vcf_device(self.h, contig, seq_name.encode("utf-8", "replace"), None, 0,
                                                    C.byref(need)), contig)
        return need.value

    def render_vcf_device_into(self, contig: int, seq_name: str, out: np.ndarray) -> int:
        """... and copy them into the caller's buffer -- any writable uint8 array, e.g. a mapped span of the output file."""
        need = C.c_uint64()
        self._check(self.lib.msim_render_vcf_device(self.h, contig, seq_name.encode("utf-8", "replace"), _ptr(out),
                                                    out.shape[0], C.byref(need)), contig)
        return need.value

    def fetch_sequence_framed_size(self, contig: int, bpl: int) -> int:
        need = C.c_uint64()
        self._check(self.lib.msim_fetch_sequence_framed(self.h, contig, bpl, None, 0, C.byref(need)), contig)
        return need.value

    def fetch_sequence_framed_into(self, contig: int, bpl: int, out: np.ndarray) -> int:
        need = C.c_uint64()
        self._check(self.lib.msim_fetch_sequence_framed(self.h, contig, bpl, _ptr(out), out.shape[0], C.byref(need)), contig)
        return need.value

    def file_wait(self):
        """Everything queued by the ``..._to_file`` calls is in its file (raises the first failure of a queued write)."""
        self._check(self.lib.msim_file_wait(self.h))

    def fetch_sequence_framed_to_file(self, contig: int, bpl: int, fd: int, offset: int) -> int:
        """The framed body QUEUED for bytes [offset, offset + n) of the open regular file ``fd`` (libmsim's output channel
        writes it while the caller goes on; ``file_wait`` / ``sync`` / ``close`` join); returns n.  ``MsimUnsupported``:
        ``fd`` is no regular file -- use ``..._into`` + write()."""
        n = C.c_uint64()
        self._check(self.lib.msim_fetch_sequence_framed_file(self.h, contig, bpl, int(fd), int(offset), C.byref(n)), contig)
        return n.value

    def render_vcf_device_to_file(self, contig: int, seq_name: str, fd: int, offset: int) -> int:
        """The contig's VCF record lines into bytes [offset, offset + n) of the open file ``fd``; returns n."""
        n = C.c_uint64()
        self._check(self.lib.msim_render_vcf_device_file(self.h, contig, seq_name.encode("utf-8", "replace"), int(fd),
                                                         int(offset), C.byref(n)), contig)
        return n.value

    def fetch_sequence_framed(self, contig: int, bpl: int, guess_len: int | None = None) -> np.ndarray:
        """The mutated contig as FASTA body text (newline after every ``bpl`` bases).  With ``guess_len`` (e.g. the
        input length: an SNP-only table never changes it) the text comes back in ONE synchronising call when it
        fits; the library keeps the framed text, so a retry with the exact size only copies."""
        need = C.c_uint64()
        if guess_len is not None:
            cap = guess_len + guess_len // bpl + guess_len // 8 + 64
            out = np.empty(cap, dtype=np.uint8)
            rc = self.lib.msim_fetch_sequence_framed(self.h, contig, bpl, _ptr(out), cap, C.byref(need))
            if rc == OK:
                return out[:need.value]
            if rc != ERR_ARG or need.value <= cap:
                self._check(rc, contig)
        else:
            self._check(self.lib.msim_fetch_sequence_framed(self.h, contig, bpl, None, 0, C.byref(need)), contig)
        out = np.empty(need.value, dtype=np.uint8)
        if need.value:
            self._check(self.lib.msim_fetch_sequence_framed(self.h, contig, bpl, _ptr(out), need.value, C.byref(need)), contig)
        return out

    def result_checksum(self, contig: int) -> int:
        s = C.c_uint64()
        self._check(self.lib.msim_result_checksum(self.h, contig, C.byref(s)))
        return s.value

    def result_device_ptr(self, contig: int):
        """(device address, length) of the mutated stream, for GPU-to-GPU transports."""
        a, n = C.c_uint64(), C.c_uint64()
        self._check(self.lib.msim_result_device_ptr(self.h, contig, C.byref(a), C.byref(n)), contig)
        return a.value, n.value

    # ------------------------------------------------------------------ many small contigs in one pass
    def batch_run(self, items):
        """``items``: list of (body uint8 array, n_bases, lenc, lenb, [Range], name, header).  Returns (fasta_text, vcf_text,
        per-contig empty flags, bases on the last FASTA line): the texts are uint8 VIEWS of libmsim's buffers (valid until
        the next batch); fasta_text is the complete run of records ('>' header lines included) as FastaWriter writes them."""
        n = len(items)
        arr = (BatchContig * n)()
        keep = []
        for i, (body, n_bases, lenc, lenb, ranges, name, header) in enumerate(items):
            body = np.ascontiguousarray(body, dtype=np.uint8)
            ra = (Range * max(len(ranges), 1))(*ranges)
            nm = name.encode("utf-8", "replace")
            hd = header.encode("utf-8", "replace")
            keep.append((body, ra, nm, hd))
            q = arr[i]
            q.body = body.ctypes.data
            q.body_bytes = body.shape[0]
            q.n_bases = n_bases
            q.lenc, q.lenb = lenc, lenb
            q.ranges = ra
            q.n_ranges = len(ranges)
            q.name = nm
            q.header = hd
        self._check(self.lib.msim_batch_run(self.h, arr, n))
        return self._batch_result(n)          # (frames the FASTA text now: the deflines in `keep` die with this call)

    def batch_run_table(self, table: np.ndarray, keep=(), defer_fasta: bool = False):
        """The same from a ready ``BATCH_CONTIG_DTYPE`` table (pointers into buffers the caller keeps alive -- ``keep`` --
        for the duration of the call): an assembly's batch is built with array operations, not per contig.
        ``defer_fasta``: the first result is the FASTA text's SIZE; ``batch_fetch_fasta(dst)`` then frames it straight into
        ``dst`` (a mapped span of the output file) -- the deflines the table points at must still be alive then."""
        table = np.ascontiguousarray(table)
        n = int(table.shape[0])
        self._check(self.lib.msim_batch_run(self.h, C.cast(C.c_void_p(table.ctypes.data), C.POINTER(BatchContig)), n))
        del keep
        return self._batch_result(n, defer_fasta)

    def batch_fetch_fasta(self, dst: np.ndarray):
        """Frame the last batch's FASTA text into ``dst`` (writable, contiguous uint8, at least the text's size)."""
        if dst.dtype != np.uint8 or not dst.flags.c_contiguous or not dst.flags.writeable:
            raise ValueError("batch_fetch_fasta needs a writable contiguous uint8 array")
        self._check(self.lib.msim_batch_fetch(self.h, C.c_void_p(dst.ctypes.data), dst.shape[0], None, 0))

    def batch_fetch_to_files(self, fasta_fd: int, fasta_offset: int, vcf_fd: int, vcf_offset: int):
        """Queue the last batch's FASTA text (framed by libmsim's host threads) and VCF text for their files' spans at the
        given offsets (-1: not that one); ``file_wait`` joins."""
        self._check(self.lib.msim_batch_fetch_file(self.h, int(fasta_fd), int(fasta_offset), int(vcf_fd), int(vcf_offset)))

    def _batch_result(self, n: int, defer_fasta: bool = False):
        em = (C.c_int32 * n)()
        self._check(self.lib.msim_batch_sizes(self.h, n, None, None, em, None))
        fp, vp = C.c_void_p(), C.c_void_p()
        fn, vn, last = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.msim_batch_view(self.h, None if defer_fasta else C.byref(fp), C.byref(fn), C.byref(vp), C.byref(vn),
                                             C.byref(last)))

        def view(ptr, nbytes):
            if not nbytes:
                return np.empty(0, dtype=np.uint8)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))
        return (fn.value if defer_fasta else view(fp, fn.value)), view(vp, vn.value), np.frombuffer(em, dtype=np.int32) != 0, last.value

    # ------------------------------------------------------------------ multi-GPU (csrc/comm.cpp)
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.msim_comm_init(self.h, buf, rank, world))

    def comm_destroy(self):
        self._check(self.lib.msim_comm_destroy(self.h))

    def planned_out_len(self, contig: int):
        """(mutated length, known): known = PLAN alone fixes the length (every rank can compute it)."""
        n, k = C.c_uint64(), C.c_int()
        self._check(self.lib.msim_planned_out_len(self.h, contig, C.byref(n), C.byref(k)))
        return n.value, bool(k.value)

    def gather_to_root(self, contig_ids, owner, out_len, n_records=None, pool_len=None, root: int = 0):
        """Moves every slot -- mutated stream, and with ``n_records`` / ``pool_len`` its record table and insert pool -- to
        ``root`` over RCCL; returns per slot the device addresses (stream, records, pool) on this rank (0: not here / empty)."""
        n = len(contig_ids)
        ids = (C.c_int * max(n, 1))(*contig_ids)
        own = (C.c_int * max(n, 1))(*owner)
        lens = (C.c_uint64 * max(n, 1))(*out_len)
        nrec = (C.c_uint64 * max(n, 1))(*n_records) if n_records is not None else None
        pool = (C.c_uint64 * max(n, 1))(*pool_len) if pool_len is not None else None
        addrs = (C.c_uint64 * (3 * max(n, 1)))()
        self._check(self.lib.msim_gather_to_root(self.h, n, ids, own, lens, nrec, pool, root, addrs))
        return [tuple(addrs[3 * i + j] for j in range(3)) for i in range(n)]

    def gather_fetch(self, device_addr: int, nbytes: int, dtype=np.uint8) -> np.ndarray:
        """A gathered part (address from ``gather_to_root``) as a host array."""
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        if nbytes:
            self._check(self.lib.msim_gather_fetch(self.h, device_addr, nbytes, C.c_void_p(out.ctypes.data)))
        return out

    def release_result(self, contig: int):
        self._check(self.lib.msim_release_result(self.h, contig))

    def stats(self) -> dict:
        t = Timing()
        self._check(self.lib.msim_stats(self.h, C.byref(t)))
        return t.as_dict()

    def reset_stats(self):
        self._check(self.lib.msim_reset_stats(self.h))


def comm_unique_id() -> bytes:
    """ncclGetUniqueId as 128 opaque bytes (call on one rank, broadcast over the control plane)."""
    lib = load()
    # RCCL's bootstrap looks for a network interface; on a box without one (or without name resolution) that can
    # take a minute.  Inside one node the loopback interface is all it needs.
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    buf = (C.c_uint8 * 128)()
    rc = lib.msim_comm_unique_id(buf)
    if rc != OK:
        raise MsimError(f"msim_comm_unique_id failed ({rc}): {lib.msim_last_error(None).decode()}")
    return bytes(buf)


def gather_plan(owner, out_len, rank: int, world: int, root: int = 0, n_records=None, pool_len=None):
    """The transfers ``rank`` posts for a gather: list of (kind, slot, part, peer, bytes); kind 0 send, 1 recv, 2 local;
    part 0 the mutated stream, 1 the record table, 2 the insert pool."""
    lib = load()
    n = len(owner)
    own = (C.c_int * max(n, 1))(*owner)
    lens = (C.c_uint64 * max(n, 1))(*out_len)
    nrec = (C.c_uint64 * max(n, 1))(*n_records) if n_records is not None else None
    pool = (C.c_uint64 * max(n, 1))(*pool_len) if pool_len is not None else None
    ops = (C.c_int64 * (15 * max(n, 1)))()
    k = C.c_int()
    rc = lib.msim_gather_plan(n, own, lens, nrec, pool, rank, world, root, ops, C.byref(k))
    if rc != OK:
        raise MsimError(f"msim_gather_plan failed ({rc})")
    return [tuple(int(ops[5 * i + j]) for j in range(5)) for i in range(k.value)]


def render_vcf(recs: np.ndarray, pool: np.ndarray, bases: np.ndarray, seq_name: str) -> bytes:
    """Record table -> VCF record lines (host helper; no GPU, no context)."""
    lib = load()
    recs = np.ascontiguousarray(recs)
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    need = C.c_uint64()
    name = seq_name.encode("utf-8", "replace")
    rc = lib.msim_render_vcf(_ptr(recs), recs.shape[0], _ptr(pool), _ptr(bases), bases.shape[0],
                             name, None, 0, C.byref(need))
    if rc != OK:
        raise MsimError(f"msim_render_vcf failed ({rc})")
    out = np.empty(need.value, dtype=np.uint8)
    rc = lib.msim_render_vcf(_ptr(recs), recs.shape[0], _ptr(pool), _ptr(bases), bases.shape[0],
                             name, _ptr(out), need.value, C.byref(need))
    if rc != OK:
        raise MsimError(f"msim_render_vcf failed ({rc})")
    return out.tobytes()


FASTA_RECORD_DTYPE = np.dtype([("h0", "<u8"), ("h1", "<u8"), ("b0", "<u8"), ("b1", "<u8"), ("n_bases", "<u8"),
                               ("lenc", "<u4"), ("lenb", "<u4"), ("flags", "<u4"), ("rsv", "<u4")])   # msim_fasta_record
FASTA_HAS_BODY, FASTA_BAD_LINES, FASTA_NONUNIFORM = 1, 2, 4


def fasta_index(text: np.ndarray):
    """Index pass over a whole FASTA text (uint8): structured array of msim_fasta_record, or None when sequence text
    precedes the first defline (pyfaidx: FastaIndexingError)."""
    lib = load()
    text = np.ascontiguousarray(text, dtype=np.uint8)
    n = C.c_uint64()
    cap = 4096                                         # one pass for anything but a large assembly (the defline scan reads
    while True:                                        # the whole text: do not run it twice just to learn the count)
        out = np.zeros(cap, dtype=FASTA_RECORD_DTYPE)
        rc = lib.msim_fasta_index(_ptr(text), text.shape[0], out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
        if rc == 3:                                    # MSIM_ERR_VALUE
            return None
        if rc == ERR_ARG and n.value > cap:
            cap = n.value
            continue
        if rc:
            raise MsimError(f"msim_fasta_index failed ({rc})")
        out = out[:n.value]
        break
    return out
