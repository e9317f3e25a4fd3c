"""ctypes front-end of the CPU oracle (oracle/msim_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; nothing under ``mutation-simulator_amd/`` does.  The oracle consumes a *settings tree* in
the neutral dict form ``tests/golden/make_goldens.py:dump_sim`` writes (so it can be driven by
trees dumped from the real reference as well as by trees our host package derives).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "_build" / "libmsim_oracle.so"

TYPE_ID = {"SN": 1, "IN": 2, "DE": 3, "DU": 4, "IV": 5, "TL": 6, "TLI": 7}
TYPE_NAME = {v: k for k, v in TYPE_ID.items()}


class OracleValueError(ValueError):
    """The reference would raise ValueError('Sample larger than population or is negative')."""


class OrcRange(C.Structure):
    _fields_ = [("start", C.c_int64), ("stop", C.c_int64), ("rate_sum", C.c_double),
                ("n_types", C.c_int32), ("types", C.c_int32 * 8), ("chances", C.c_double * 8),
                ("min_len", C.c_int64 * 8), ("max_len", C.c_int64 * 8)]


class OrcRec(C.Structure):
    _fields_ = [("pos", C.c_int64), ("type", C.c_int32), ("rev", C.c_int32), ("start", C.c_int64),
                ("stop", C.c_int64), ("ins_pos", C.c_int64)]


def build(force: bool = False) -> Path:
    src = HERE / "msim_oracle.c"
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB_PATH))
        L.orc_new.restype = C.c_void_p
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_seed_py.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
        L.orc_seed_np.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_set_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_int]
        L.orc_get_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.orc_words.argtypes = [C.c_void_p, C.c_int]
        L.orc_words.restype = C.c_uint64
        L.orc_set_block.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_set_titv.argtypes = [C.c_void_p, C.c_double]
        L.orc_py_next32.argtypes = [C.c_void_p]
        L.orc_py_next32.restype = C.c_uint32
        L.orc_np_next32.argtypes = [C.c_void_p]
        L.orc_np_next32.restype = C.c_uint32
        L.orc_skip_words.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.orc_skip_words.restype = None
        L.orc_randbelow.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_randbelow.restype = C.c_uint64
        L.orc_randint.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.orc_randint.restype = C.c_int64
        L.orc_random.argtypes = [C.c_void_p]
        L.orc_random.restype = C.c_double
        L.orc_setsize.argtypes = [C.c_int64]
        L.orc_setsize.restype = C.c_int64
        L.orc_sample.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        L.orc_sample_min_dist.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                          C.POINTER(C.c_int64)]
        L.orc_np_double.argtypes = [C.c_void_p]
        L.orc_np_double.restype = C.c_double
        L.orc_choice_p.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int64,
                                   C.POINTER(C.c_int32)]
        L.orc_choice_atgc.argtypes = [C.c_void_p, C.c_int64, C.c_char_p]
        L.orc_get_mutations.argtypes = [C.c_void_p, C.POINTER(OrcRange), C.c_int64,
                                        C.POINTER(C.POINTER(OrcRec)), C.POINTER(C.c_int64),
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mutate_sequence.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p,
                                          C.POINTER(OrcRec), C.c_int64]
        L.orc_mutate_contig.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_char_p,
                                        C.c_int64, C.POINTER(OrcRange), C.c_int,
                                        C.POINTER(C.POINTER(OrcRec)), C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int)]
        L.orc_vcf_header.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p),
                                     C.POINTER(C.c_int64), C.c_char_p, C.c_char_p, C.c_char_p,
                                     C.c_char_p]
        L.orc_release.argtypes = [C.c_void_p]
        L.orc_set_bpl.argtypes = [C.c_void_p, C.c_int64]
        L.orc_write_header.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_fasta.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_fasta.restype = C.c_void_p
        L.orc_vcf.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_vcf.restype = C.c_void_p
        L.orc_clear_outputs.argtypes = [C.c_void_p]
        L.orc_key_error_base.argtypes = [C.c_void_p]
        L.orc_key_error_base.restype = C.c_char
        _lib = L
    return _lib


def seed_key(seed: int) -> list[int]:
    """random.seed(int): abs(seed) as little-endian 32-bit limbs, at least one limb."""
    n = abs(int(seed))
    key = []
    while True:
        key.append(n & 0xFFFFFFFF)
        n >>= 32
        if not n:
            break
    return key


def range_from_dump(rd: dict) -> OrcRange | None:
    """``{"start","stop","settings":{chances_hex, rate_sum_hex, min, max}}`` -> OrcRange."""
    st = rd["settings"]
    if st is None or not st["has_mutations"]:
        return None
    r = OrcRange()
    r.start, r.stop = rd["start"], rd["stop"]
    r.rate_sum = float.fromhex(st["rate_sum_hex"])
    r.n_types = len(st["chances_hex"])
    for j, (name, hx) in enumerate(st["chances_hex"]):
        r.types[j] = TYPE_ID[name]
        r.chances[j] = float.fromhex(hx)
    for name, v in (st["min"] or {}).items():
        r.min_len[TYPE_ID[name]] = v
    for name, v in (st["max"] or {}).items():
        r.max_len[TYPE_ID[name]] = v
    return r


class Oracle:
    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_new())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_free(self.h)
            self.h = None

    # ---- RNG
    def seed(self, py_seed: int, np_seed: int):
        key = seed_key(py_seed)
        arr = (C.c_uint32 * len(key))(*key)
        self.L.orc_seed_py(self.h, arr, len(key))
        self.L.orc_seed_np(self.h, np_seed & 0xFFFFFFFF)

    def set_state(self, stream: int, mt, idx: int):
        arr = (C.c_uint32 * 624)(*[int(x) for x in mt])
        self.L.orc_set_state(self.h, stream, arr, idx)

    def get_state(self, stream: int):
        arr = (C.c_uint32 * 624)()
        idx = C.c_int()
        self.L.orc_get_state(self.h, stream, arr, C.byref(idx))
        return list(arr), idx.value

    def words(self, stream: int) -> int:
        return self.L.orc_words(self.h, stream)

    def py_words32(self, n):
        return [self.L.orc_py_next32(self.h) for _ in range(n)]

    def np_words32(self, n):
        return [self.L.orc_np_next32(self.h) for _ in range(n)]

    def skip_words(self, stream: int, n: int):
        """n words of the stream drawn and dropped (sequential generation, in C)."""
        self.L.orc_skip_words(self.h, stream, n)

    def randbelow(self, n):
        return self.L.orc_randbelow(self.h, n)

    def randint(self, a, b):
        return self.L.orc_randint(self.h, a, b)

    def uniform01(self):
        return self.L.orc_random(self.h)

    def sample(self, n, k):
        out = (C.c_int64 * max(k, 1))()
        rc = self.L.orc_sample(self.h, n, k, out)
        if rc == 1:
            raise OracleValueError("Sample larger than population or is negative")
        assert rc == 0
        return list(out[:k])

    def sample_min_dist(self, start, stop, k, d):
        out = (C.c_int64 * max(k, 1))()
        rc = self.L.orc_sample_min_dist(self.h, start, stop, k, d, out)
        if rc == 1:
            raise OracleValueError("Sample larger than population or is negative")
        assert rc == 0
        return np.ctypeslib.as_array(out)[:k].copy()

    def choice_p(self, p, size):
        arr = (C.c_double * len(p))(*p)
        out = (C.c_int32 * size)()
        self.L.orc_choice_p(self.h, arr, len(p), size, out)
        return list(out)

    def choice_atgc(self, n):
        buf = C.create_string_buffer(n)
        self.L.orc_choice_atgc(self.h, n, buf)
        return buf.raw[:n].decode()

    # ---- settings
    def configure(self, sim: dict):
        blk = (C.c_int64 * 8)(*([1] * 8))
        for name, v in sim["mut_block"]:
            blk[TYPE_ID[name]] = v
        self.L.orc_set_block(self.h, blk)
        self.L.orc_set_titv(self.h, float(sim["titv"]))

    # ---- plan / apply
    def get_mutations(self, rd: dict, chrom_len: int):
        r = range_from_dump(rd)
        recs = C.POINTER(OrcRec)()
        n = C.c_int64()
        rc = self.L.orc_get_mutations(self.h, C.byref(r), chrom_len, C.byref(recs), C.byref(n),
                                      None, None, None, None)
        if rc == 1:
            raise OracleValueError("Sample larger than population or is negative")
        assert rc == 0, rc
        out = [(recs[i].pos, recs[i].type, recs[i].stop) for i in range(n.value)]
        if n.value:
            self.L.orc_release(recs)
        return out

    def mutate_sequence(self, seq: bytes, name: str, long_name: str, bpl: int, muts):
        """``muts``: iterable of (type_id, start, stop) keyed by start (hand-built dict)."""
        muts = sorted(muts, key=lambda m: m[1])
        arr = (OrcRec * max(len(muts), 1))()
        for i, (t, s, e) in enumerate(muts):
            arr[i].pos, arr[i].type, arr[i].start, arr[i].stop = s, t, s, e
        self.L.orc_set_bpl(self.h, bpl)
        self.L.orc_write_header(self.h, long_name.encode())
        buf = np.frombuffer(seq, dtype=np.uint8)
        rc = self.L.orc_mutate_sequence(self.h, buf.ctypes.data, len(seq), name.encode(), arr,
                                        len(muts))
        self._raise(rc)

    def _raise(self, rc):
        if rc == 1:
            raise OracleValueError("Sample larger than population or is negative")
        if rc == 2:
            raise KeyError(self.L.orc_key_error_base(self.h).decode())
        if rc:
            raise RuntimeError(f"oracle rc={rc}")

    def outputs(self):
        n = C.c_uint64()
        p = self.L.orc_fasta(self.h, C.byref(n))
        fa = C.string_at(p, n.value) if n.value else b""
        p = self.L.orc_vcf(self.h, C.byref(n))
        vcf = C.string_at(p, n.value) if n.value else b""
        return fa, vcf

    def mutate_contig_stream(self, bases: np.ndarray, name: str, long_name: str, lenc: int, ranges: list):
        """One iteration of mutate()'s contig loop (mutator.py:111-141) with the outputs cleared first: returns
        (fasta text of this contig incl. its header line, its VCF record lines, had_mutations).  The RNG streams
        continue across calls like the reference's; memory stays at one contig (full-size parity tests)."""
        self.L.orc_clear_outputs(self.h)
        rs = [r for r in (range_from_dump(rd) for rd in ranges) if r is not None]
        arr = (OrcRange * max(len(rs), 1))(*rs)
        had = C.c_int()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        rc = self.L.orc_mutate_contig(self.h, bases.ctypes.data, len(bases), name.encode(), long_name.encode(), lenc,
                                      arr, len(rs), None, None, C.byref(had))
        self._raise(rc)
        fa, vcf = self.outputs()
        return fa, vcf, bool(had.value)

    def run_genome(self, contigs, sim: dict, infile_name: str, date: str = "MASKED",
                   keep_records: bool = False):
        """Whole ``Mutator.__init__`` + ``mutate()`` (mutator.py:79-142).

        ``contigs``: list of dicts {name, long_name, lenc, bases(np.uint8, upper-cased)}.
        Returns (fasta_bytes, vcf_bytes, no_mutation_contig_numbers, records_per_contig|None).
        """
        self.configure(sim)
        self.L.orc_clear_outputs(self.h)
        names = (C.c_char_p * len(contigs))(*[c["name"].encode() for c in contigs])
        lens = (C.c_int64 * len(contigs))(*[len(c["bases"]) for c in contigs])
        self.L.orc_vcf_header(self.h, infile_name.encode(), len(contigs), names, lens,
                              sim["assembly_name"].encode(), sim["species_name"].encode(),
                              sim["sample_name"].encode(), date.encode())
        empty, recs_out = [], []
        for chrom in sim["chromosomes"]:
            c = contigs[chrom["number"]]
            rs = [r for r in (range_from_dump(rd) for rd in chrom["ranges"]) if r is not None]
            arr = (OrcRange * max(len(rs), 1))(*rs)
            recs = C.POINTER(OrcRec)()
            n = C.c_int64()
            had = C.c_int()
            bases = np.ascontiguousarray(c["bases"], dtype=np.uint8)
            rc = self.L.orc_mutate_contig(self.h, bases.ctypes.data, len(bases), c["name"].encode(),
                                          c["long_name"].encode(), c["lenc"], arr, len(rs),
                                          C.byref(recs), C.byref(n), C.byref(had))
            self._raise(rc)
            if not had.value:
                empty.append(chrom["number"])
            if keep_records:
                a = np.zeros((n.value, 6), dtype=np.int64)
                for i in range(n.value):
                    r = recs[i]
                    a[i] = (r.pos, r.type, r.start, r.stop, r.rev, r.ins_pos)
                recs_out.append(a)
            if recs:
                self.L.orc_release(recs)
        fa, vcf = self.outputs()
        return fa, vcf, empty, (recs_out if keep_records else None)

    # ---- interchromosomal translocations: the reference's second pass
    def shuffle(self, items: list) -> list:
        """Lib/random.py shuffle, by randbelow (msim_oracle.c: orc_shuffle does the same over int64)."""
        x = list(items)
        for i in range(len(x) - 1, 0, -1):
            j = self.randbelow(i + 1)
            x[i], x[j] = x[j], x[i]
        return x

    def it_pass(self, contigs, it_rates, ignore_warnings: bool = False):
        """``ITMutator.__init__`` + ``mutate()`` (it_mutator.py:24-220, bedpe_writer.py:36-55, fasta_writer.py:31-58).

        ``contigs``: list of dicts {name, long_name, lenc, bases(np.uint8, upper-cased)} -- the Fasta the pass reads;
        ``it_rates``: per contig a float or None.  Draws from THIS oracle's CPython stream.
        Returns (fasta_bytes, bedpe_bytes, [warning texts])."""
        warnings = []

        def warn(text):
            if not ignore_warnings:
                warnings.append(text)
        # it_mutator.py:50-57  contigs with a rate (0 counts) and more than two bases
        avail = [i for i, r in enumerate(it_rates) if r is not None and len(contigs[i]["bases"]) > 2]
        # it_mutator.py:59-71  shuffle, then a walk over the very list that is being emptied
        avail = self.shuffle(avail)
        partners = {}
        i = 0
        while i < len(avail):
            chrom = avail[i]
            i += 1
            avail.remove(chrom)
            if avail:
                partner = avail[self.randbelow(len(avail))]            # random.choice
                partners[partner] = chrom
                partners[chrom] = partner
                avail.remove(partner)
        # it_mutator.py:73-83  one contig per pair
        once = list(partners.keys())
        for chrom, partner in partners.items():
            if chrom in once:
                once.remove(partner)
        # it_mutator.py:158-190  breakpoints of every pair
        bps = {}
        for chrom in once:
            partner = partners[chrom]
            l1, l2 = len(contigs[chrom]["bases"]), len(contigs[partner]["bases"])
            amount = int((l1 + l2 - 4) / 2 * ((it_rates[chrom] + it_rates[partner]) / 2))      # it_mutator.py:96
            b1, b2 = [], []
            try:
                b1 = [int(x) for x in self.sample_min_dist(1, l1, amount, 1)]
                b2 = [int(x) for x in self.sample_min_dist(1, l2, amount, 1)]
            except OracleValueError:
                warn(f"Interchromosomal translocation rate too high for sequence {chrom+1} and {partner+1}.")
            if b1 and b2:
                bps[chrom] = (b1, b2)
                bps[partner] = (b2, b1)
            else:
                warn(f"No interchromosomal translocations could be generated between sequence {chrom+1} and {partner+1} "
                     "(it rates too low).")
        # it_mutator.py:192-216  write every contig, in file order
        fa, bedpe = bytearray(), bytearray()
        written = 0
        for k, c in enumerate(contigs):
            if written:                                                  # fasta_writer.py:31-38
                fa += b"\n"
            fa += b">" + c["long_name"].encode() + b"\n"
            written = 0
            if k in bps:
                own, other = bps[k]
                p = contigs[partners[k]]
                ca, cb = [0] + own + [len(c["bases"])], [0] + other + [len(p["bases"])]
                parts = [(p["bases"][cb[j]:cb[j + 1]] if j % 2 else c["bases"][ca[j]:ca[j + 1]]) for j in range(len(ca) - 1)]
                seq = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
                n = len(own)                                             # bedpe_writer.py:44-55
                for j in range(0, n, 2):
                    if j != n - 1:
                        bedpe += f"{c['name']}\t{own[j]}\t{own[j+1]}\t{p['name']}\t{other[j]}\t{other[j+1]}\n".encode()
                    elif n % 2:
                        bedpe += f"{c['name']}\t{own[j]}\t{len(c['bases'])}\t{p['name']}\t{other[j]}\t{len(p['bases'])}\n".encode()
            else:
                # __write_chrom_full (it_mutator.py:148-156) writes the header itself -- after __mutate_sequence already
                # did (it_mutator.py:199-202): a contig without breakpoints carries its defline twice
                fa += b">" + c["long_name"].encode() + b"\n"
                seq = c["bases"]
            bpl = c["lenc"]
            raw = bytes(np.asarray(seq, dtype=np.uint8))
            for a in range(0, len(raw), bpl):                            # fasta_writer.py:40-58
                line = raw[a:a + bpl]
                fa += line
                if len(line) == bpl:
                    fa += b"\n"
                    written = 0
                else:
                    written = len(line)
        return bytes(fa), bytes(bedpe), warnings

