// APPLY: records + uint8 genome in HBM -> mutated uint8 stream in HBM.   gfx950 (MI355X) only.
//
// Takes over Mutator.__mutate_sequence (mutator.py:318-426), i.e. the per-base Python loop that is
// ~96 % of the reference's wall time, as an output-centric streaming rewrite:
//
//   1. k_delta_* : per-record length delta -> exclusive scan -> output offset of every record
//      (the "wavefront scan for the length delta"; three small kernels, records only).
//   2. k_tile_index : for every 16 KiB output tile the last record starting at or before it.
//   3. k_rewrite<CAP> / k_rewrite_snp : one workgroup per output tile, the tile assembled in LDS and streamed out
//      with aligned nontemporal 16-B stores.
//        - k_rewrite_snp (SNP-only tables, offset == position): aligned 16-B loads into the LDS tile, ONE LANE PER
//          RECORD patches its byte through an LDS LUT.
//        - k_rewrite<CAP> (indels / SVs): the tile's record window (offset / segment end / run source / type /
//          segment source, 20 B per record) in LDS; an index table (LDS atomicMax + max-scan) gives every 16-byte
//          group its governing record; structural (non-SNP) records own pieces -- a segment (insert-pool bytes,
//          duplication copy, reverse complement, TLI span) and the copy run after it; pass A resolves the piece
//          covering each group's first byte, pass B one lane per piece that starts inside a group (masked dword
//          LDS merges), pass B2 one lane per SNP record.  Every piece load is TWO ALIGNED 16-B loads + a byte
//          funnel (round 2: the byte-shifted 20-B loads of round 1 cost 23 % of the kernel).
//      The kernels are HBM-bound: per output byte one input byte read, one output byte written (+ 20 B per record).
//
// Integer/byte work only -- no MFMA.  Roofline: HBM bandwidth.
#include <algorithm>
#include <vector>

#include "ctx.h"

// Timing-only ablations of k_rewrite (tools/apply_ablation.py; `make ablate`).  0 in every shipped build: any other
// value produces WRONG bytes and exists only to attribute the kernel's time to its parts on the GPU.
//   1 aligned loads (no byte shift, no fifth dword)   2 no structural fix-ups (pass B)   4 no SNP pass (B2)
//   8 no index table (every group = plain copy)      16 XCD-contiguous tile order
#ifndef MSIM_ABL
#define MSIM_ABL 0
#endif

namespace msim {

namespace {

constexpr int THREADS = 256;
constexpr int GROUP = 16;                              // output bytes per lane per iteration
#ifndef MSIM_ITERS
#define MSIM_ITERS 4                                   // (A/B builds: 8 = 32 KiB tiles, four per CU -- see the Makefile's `ablate`)
#endif
constexpr int ITERS = MSIM_ITERS;
constexpr int TILE = THREADS * GROUP * ITERS;          // 16384 output bytes per workgroup
constexpr int REC_CAP = 1024;                          // records staged in LDS per tile (dense tables)
constexpr int REC_CAP_SMALL = 35 * ITERS;              // sparse tables: window (140 entries) + tile + LUT = 20 KB of LDS -> 8 tiles per CU
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_BLOCK = THREADS * SCAN_ITEMS;

// LUT layout in the 1280-byte table (built on the host, ctx creation):
//   [0,256)     snp, aux 0 : transition(conv(x))                      mutator.py:77,463
//   [256,512)   snp, aux 1 : transversion column 0 of conv(x), 0 = KeyError   mutator.py:449-455
//   [512,768)   snp, aux 2 : transversion column 1
//   [768,1024)  conv(x)    : non_ambiguous                            mutator.py:75
//   [1024,1280) comp(conv(x))                                         mutator.py:76,383
constexpr int LUT_BYTES = 1280;

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void rec_lengths(const msim_record &r, uint32_t &outlen, uint32_t &inlen) {
    const uint32_t len = r.stop - r.pos + 1;
    switch (r.type) {
        case MSIM_SN: outlen = 1; inlen = 1; break;
        case MSIM_IN: outlen = len + 1; inlen = 1; break;          // insert, then the base itself
        case MSIM_DE: outlen = 0; inlen = len; break;
        case MSIM_IV: outlen = len; inlen = len; break;
        case MSIM_DU: outlen = 2 * len; inlen = len; break;
        case MSIM_TL: outlen = 0; inlen = len; break;               // excised like a deletion
        case MSIM_TLI: {                                            // seq[start : stop+1], then the base itself
            const uint32_t ilen = r.stop + 1 > r.extra ? r.stop + 1 - r.extra : 0;
            outlen = ilen + 1; inlen = 1; break;
        }
        default: outlen = 0; inlen = 0; break;
    }
}

// ------------------------------------------------------------------ 1. offsets (delta scan)
__global__ __launch_bounds__(THREADS) void k_delta_reduce(const msim_record *__restrict__ recs, uint32_t n,
                                                          long long *__restrict__ block_sums) {
    __shared__ long long red[THREADS / 64];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    long long s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) {
            uint32_t ol, il;
            rec_lengths(recs[base + i], ol, il);
            s += (long long)ol - (long long)il;
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < THREADS / 64; w++) t += red[w];
        block_sums[blockIdx.x] = t;
    }
}

// exclusive scan of block_sums[0..nb) in place, total to block_sums[nb]; single workgroup
__global__ __launch_bounds__(1024) void k_scan_sums(long long *__restrict__ block_sums, uint32_t nb) {
    __shared__ long long buf[1024];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const long long v = i < nb ? block_sums[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            long long t = threadIdx.x >= (unsigned)o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        const long long incl = buf[threadIdx.x];
        const long long c = carry;
        if (i < nb) block_sums[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[nb] = carry;
}

__global__ __launch_bounds__(THREADS) void k_offsets(const msim_record *__restrict__ recs, uint32_t n,
                                                     const long long *__restrict__ block_sums,
                                                     uint32_t *__restrict__ off) {
    __shared__ long long part[THREADS];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    long long d[SCAN_ITEMS];
    long long s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        d[i] = 0;
        if (base + i < n) {
            uint32_t ol, il;
            rec_lengths(recs[base + i], ol, il);
            d[i] = (long long)ol - (long long)il;
        }
        s += d[i];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < THREADS; o <<= 1) {
        long long t = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    long long run = block_sums[blockIdx.x] + part[threadIdx.x] - s;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) off[base + i] = (uint32_t)((long long)recs[base + i].pos + run);
        run += d[i];
    }
}

// ------------------------------------------------------------------ 2. tile index
// first[t] = index of the last record whose output offset is <= t*TILE, or -1
// (off == nullptr: SNP-only table, a record's output offset is its position)
// dyn != nullptr (tables of the counter-based PLAN engine, plan_fast.hip): the record count is only known on the device --
// dyn[0] = records, dyn[1] = mutated length; n / n_entries are then upper bounds the grid was sized with
__global__ __launch_bounds__(THREADS) void k_tile_index(const uint32_t *__restrict__ off,
                                                        const msim_record *__restrict__ recs, uint32_t n,
                                                        int32_t *__restrict__ first, uint32_t n_entries,
                                                        const uint32_t *__restrict__ dyn, unsigned long long *__restrict__ err) {
    const uint32_t t = blockIdx.x * THREADS + threadIdx.x;
    if (t == 0) *err = ~0ull;                              // the contig's KeyError word: none so far (the rewrite kernel follows)
    if (t >= n_entries) return;
    if (dyn) n = dyn[0];
    const uint64_t target = (uint64_t)t * TILE;
    uint32_t lo = 0, hi = n;                               // upper bound: first index with off > target
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t o = off ? off[mid] : recs[mid].pos;
        if ((uint64_t)o <= target) lo = mid + 1; else hi = mid;
    }
    first[t] = (int32_t)lo - 1;
}

// The same for up to 32 contigs in ONE launch (the counter-based engine applies a batch of contigs back to back: 24 launches
// of 10 us each for a few thousand threads of work were a tenth of its step).  Jobs travel as kernel arguments.
struct TileJob {
    const uint32_t *off; const msim_record *recs; const uint32_t *dyn; int32_t *first; unsigned long long *err;
    uint32_t n, n_entries, entry_base, rsv;
};
struct TileJobs { TileJob j[32]; uint32_t n_jobs, total; };
__global__ __launch_bounds__(THREADS) void k_tile_index_batch(TileJobs J) {
    const uint32_t g = blockIdx.x * THREADS + threadIdx.x;
    if (g >= J.total) return;
    uint32_t k = 0;
    for (uint32_t q = 1; q < J.n_jobs; q++) if (J.j[q].entry_base <= g) k = q;
    const TileJob &T = J.j[k];
    const uint32_t t = g - T.entry_base;
    if (t == 0) *T.err = ~0ull;
    uint32_t n = T.dyn ? T.dyn[0] : T.n;
    const uint64_t target = (uint64_t)t * TILE;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t o = T.off ? T.off[mid] : T.recs[mid].pos;
        if ((uint64_t)o <= target) lo = mid + 1; else hi = mid;
    }
    T.first[t] = (int32_t)lo - 1;
}

// ------------------------------------------------------------------ 3. rewrite
__device__ __forceinline__ uint32_t get_byte(const u32x4 &v, uint32_t idx) {
    const uint32_t w = (idx >> 2) == 0 ? v.x : (idx >> 2) == 1 ? v.y : (idx >> 2) == 2 ? v.z : v.w;
    return (w >> ((idx & 3) * 8)) & 0xffu;
}

// How a record shapes the output stream: at most ONE special segment [o, e) followed by a plain copy
// run out[e + t] = in[s + t] that lasts until the next record's o.
//   SN : no segment (e = o), run continues at in[pos]; the byte at o is patched      mutator.py:334-341
//   IN : segment = insert pool bytes, run continues at in[pos]                       mutator.py:343-358
//   DE : no segment, run continues at in[stop + 1]                                   mutator.py:360-377
//   IV : segment = reverse complement of in[pos..stop], run continues at in[stop+1]  mutator.py:379-387
//   DU : segment = in[pos..stop] (first copy); the second copy simply IS the run, which restarts at
//        in[pos]                                                                      mutator.py:389-399
__device__ __forceinline__ void rec_view(const msim_record &r, uint32_t o, uint32_t &e, uint32_t &s) {
    const uint32_t len = r.stop - r.pos + 1;
    switch (r.type) {
        case MSIM_IN: e = o + len; s = r.pos; break;
        case MSIM_DE: e = o; s = r.stop + 1; break;
        case MSIM_IV: e = o + len; s = r.stop + 1; break;
        case MSIM_DU: e = o + len; s = r.pos; break;
        case MSIM_TL: e = o; s = r.stop + 1; break;       // DEL:ME, excised span         mutator.py:360-377
        case MSIM_TLI:                                    // INS:ME: (converted / rev-comp) copy of the linked
            e = o + (r.stop + 1 > r.extra ? r.stop + 1 - r.extra : 0);   // TL span, then in[pos]   :401-421
            s = r.pos; break;
        default: e = o; s = r.pos; break;                 // SN
    }
}

template <int CAP>
struct RecWin {                 // the tile's record window in LDS
    uint32_t o[CAP];            // output offset of the record's segment
    uint32_t e[CAP];            // end of the segment
    uint32_t s[CAP];            // input position where the copy run after the segment starts
    uint32_t m[CAP];            // type | aux << 8
    uint32_t x[CAP];            // segment source: record.extra (insert pool offset, TLI span start), or
};                              //   record.stop for a reversed TLI (its span is read backwards)

template <bool IN_LDS, int CAP>
struct RecAccess {
    const RecWin<CAP> *win;
    const msim_record *recs;
    const uint32_t *off;
    int32_t r_lo;
    __device__ __forceinline__ uint32_t O(int32_t j) const {
        return IN_LDS ? win->o[j - r_lo] : (off ? off[j] : recs[j].pos);
    }
    __device__ __forceinline__ void all(int32_t j, uint32_t &o, uint32_t &e, uint32_t &s, uint32_t &m) const {
        if (IN_LDS) {
            const int32_t q = j - r_lo;
            o = win->o[q]; e = win->e[q]; s = win->s[q]; m = win->m[q];
        } else {
            const msim_record r = recs[j];
            o = off ? off[j] : r.pos;
            rec_view(r, o, e, s);
            m = (uint32_t)r.type | ((uint32_t)r.aux << 8);
        }
    }
    // segment source (see RecWin::x) without a global round trip
    __device__ __forceinline__ uint32_t ext(int32_t j) const {
        if (IN_LDS) return win->x[j - r_lo];
        const msim_record r = recs[j];
        return (r.type == MSIM_TLI && (r.aux & 1)) ? r.stop : r.extra;
    }
};

__device__ __forceinline__ void report_key_error(unsigned long long *err, uint64_t pos, uint32_t conv_base) {
    atomicMin(err, (unsigned long long)((pos << 8) | conv_base));
}

// Four bases through one of the reference's translation tables (mutator.py:75-76):
//   mode 1: convert(x) (ambiguity codes -> bases), mode 2: complement(convert(x)).
// A/C/G/T have distinct (x >> 1) & 3 codes (0,1,3,2): v_perm_b32 turns codes back into letters (check:
// all four bytes plain ACGT -> convert is the identity) or into complements; anything else (N, IUPAC,
// U ...) takes the 256-entry LDS table.
__device__ __forceinline__ uint32_t map4(uint32_t x, uint32_t mode, const uint8_t *lut) {
    const uint32_t codes = (x >> 1) & 0x03030303u;
    if (__builtin_amdgcn_perm(0u, 0x47544341u, codes) == x)            // "ACTG"[code] == byte ?
        return mode == 2 ? __builtin_amdgcn_perm(0u, 0x43414754u, codes) : x;   // "TGAC"[code]
    const uint8_t *t = lut + (mode == 2 ? 1024 : 768);
    return (uint32_t)t[x & 0xff] | ((uint32_t)t[(x >> 8) & 0xff] << 8) | ((uint32_t)t[(x >> 16) & 0xff] << 16) |
           ((uint32_t)t[x >> 24] << 24);
}

// Where the 16 output bytes of the group starting at absolute offset G come from, as produced by ONE
// piece of record j: its segment (in_seg) or the copy run after it (has = false: the run before the
// first record).  Bytes outside the piece are garbage the caller masks or overwrites.  All output
// offsets are < 2^32 (checked on the host), so the bookkeeping is 32-bit; only sources are 64-bit.
struct PieceSrc {
    const uint8_t *ptr;      // source byte address of the group's first byte (may be unaligned)
    uint32_t mode;           // 0 raw, 1 convert, 2 complement(convert); bit 8: reversed
};

__device__ __forceinline__ PieceSrc piece_src(uint32_t G, bool has, bool in_seg, uint32_t oj, uint32_t ej,
                                              uint32_t sj, uint32_t mj, uint32_t xj,
                                              const uint8_t *__restrict__ in, const uint8_t *__restrict__ pool) {
    PieceSrc p;
    p.mode = 0;
    const uint8_t *sp = in;
    int64_t so;
    if (in_seg) {
        const uint32_t type = mj & 0xff;
        const int64_t rel = (int64_t)G - (int64_t)oj;    // segment offset of the group's first byte
        if (type == MSIM_DU) {
            so = (int64_t)sj + rel;
        } else if (type == MSIM_IN) {
            sp = pool;
            so = (int64_t)xj + rel;
        } else if (type == MSIM_IV) {                    // out[P] = rc(in[stop - (P - o)]), stop = s - 1
            so = (int64_t)sj - 16 - rel;
            p.mode = 2 | 256;
        } else {                                         // MSIM_TLI: copy of the linked TL span in[extra .. stop]
            if ((mj >> 8) & 1) { so = (int64_t)xj - 15 - rel; p.mode = 2 | 256; }
            else { so = (int64_t)xj + rel; p.mode = 1; }
        }
    } else {
        so = has ? (int64_t)sj + ((int64_t)G - (int64_t)ej) : (int64_t)G;
    }
    p.ptr = sp + so;
    return p;
}

// The 16 bytes at an arbitrary byte address.  MSIM_LOAD 1 (default): TWO ALIGNED dwordx4 loads (the 32 bytes
// around them; the second is the next lane's first, an L1 hit) and a byte funnel -- round-2 ablation: the round-1
// form (MSIM_LOAD 0: dwordx4 at a 4-byte aligned address + one more dword) cost 23 % of the kernel, a dwordx4 that
// is not 16-byte aligned being split by the texture addresser.  `s` = byte offset of the wanted bytes in d[].
#ifndef MSIM_LOAD
#define MSIM_LOAD 1
#endif
#ifndef MSIM_WPE
#define MSIM_WPE 8                                     // waves per SIMD k_rewrite<140> is compiled for (register budget 512 / WPE)
#endif
struct Raw5 { u32x4 a, b; uint32_t s; };              // 32 (or 20) bytes around the wanted 16; s = their byte offset in a|b
__device__ __forceinline__ Raw5 piece_load(const PieceSrc &p) {
    Raw5 r;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p.ptr);
    r.b = u32x4{0, 0, 0, 0};
    if (MSIM_ABL & 1) {
        r.a = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a & ~(uintptr_t)15));
        r.s = 0;
        return r;
    }
    if (MSIM_LOAD == 1) {
        const u32x4 *al = reinterpret_cast<const u32x4 *>(a & ~(uintptr_t)15);
        r.a = al[0];
        r.b = al[1];
        r.s = (uint32_t)(a & 15);
        return r;
    }
    const uint8_t *al = reinterpret_cast<const uint8_t *>(a & ~(uintptr_t)3);
    const u32x4_a4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4 *>(al));
    r.a.x = v.x; r.a.y = v.y; r.a.z = v.z; r.a.w = v.w;
    r.b.x = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(al + 16));
    r.s = (uint32_t)(a & 3);
    return r;
}
__device__ __forceinline__ u32x4 piece_finish(const Raw5 &r, uint32_t mode, const uint8_t *lut) {
    uint32_t f0 = r.a.x, f1 = r.a.y, f2 = r.a.z, f3 = r.a.w, f4 = r.b.x;
    if (MSIM_LOAD == 1 && !(MSIM_ABL & 1)) {             // dword part of the shift: two select stages over the 8 dwords
        const bool q2 = r.s & 8, q1 = r.s & 4;
        const uint32_t e0 = q2 ? r.a.z : r.a.x, e1 = q2 ? r.a.w : r.a.y, e2 = q2 ? r.b.x : r.a.z, e3 = q2 ? r.b.y : r.a.w,
                       e4 = q2 ? r.b.z : r.b.x, e5 = q2 ? r.b.w : r.b.y;
        f0 = q1 ? e1 : e0; f1 = q1 ? e2 : e1; f2 = q1 ? e3 : e2; f3 = q1 ? e4 : e3; f4 = q1 ? e5 : e4;
    }
    const uint32_t sh = r.s & 3;
    u32x4 pv;
    pv.x = __builtin_amdgcn_alignbyte(f1, f0, sh);
    pv.y = __builtin_amdgcn_alignbyte(f2, f1, sh);
    pv.z = __builtin_amdgcn_alignbyte(f3, f2, sh);
    pv.w = __builtin_amdgcn_alignbyte(f4, f3, sh);
    if (mode & 256) {
        const u32x4 w = pv;
        pv.x = __builtin_bswap32(w.w); pv.y = __builtin_bswap32(w.z);
        pv.z = __builtin_bswap32(w.y); pv.w = __builtin_bswap32(w.x);
    }
    if (mode & 3) {
        pv.x = map4(pv.x, mode & 3, lut); pv.y = map4(pv.y, mode & 3, lut);
        pv.z = map4(pv.z, mode & 3, lut); pv.w = map4(pv.w, mode & 3, lut);
    }
    return pv;
}
__device__ __forceinline__ uint32_t snp_patch(uint32_t x, uint32_t mj, uint32_t pos, const uint8_t *lut,
                                              unsigned long long *err) {
    const uint32_t nb = lut[(mj >> 8) * 256 + x];
    if (nb == 0 && (mj >> 8) != 0) { report_key_error(err, (uint64_t)pos, lut[768 + x]); return x; }
    return nb;
}

// Pass B helper: bytes [p, q) (same 16-B group) of one piece -> LDS tile, byte by byte
__device__ __forceinline__ void fix_bytes(uint8_t *tile, uint32_t tile0, uint32_t p, uint32_t q, const u32x4 &pv,
                                          int patch_idx, uint32_t patch_val) {
    const uint32_t G = p & ~15u;
    for (uint32_t b = p; b < q; b++) {
        uint32_t x = get_byte(pv, b - G);
        if ((int)(b - G) == patch_idx) x = patch_val;
        tile[b - tile0] = (uint8_t)x;
    }
}

template <bool IN_LDS, int CAP>
__device__ __forceinline__ void rewrite_tile(const RecAccess<IN_LDS, CAP> &A, int32_t r_hi, bool any_rec, int32_t cnt,
                                             uint8_t *tile, const uint8_t *__restrict__ in,
                                             uint8_t *__restrict__ out, const msim_record *__restrict__ recs,
                                             uint32_t n_rec, const uint8_t *__restrict__ pool, const uint8_t *lut,
                                             uint64_t tile0_64, uint64_t L_out, unsigned long long *err) {
    const int32_t r_lo = A.r_lo;
    const uint32_t INF = 0xffffffffu;
    const uint32_t tile0 = (uint32_t)tile0_64;
    const uint32_t tile_end = (uint32_t)min<uint64_t>(tile0_64 + TILE, L_out);
    // ---- resolve: (A) every group from the piece that covers its first byte; (B) one lane per PIECE (2 per
    // record: segment, copy run) owns the bytes from the piece start to the end of its 16-B group.  All sources
    // are resolved first (LDS searches), then ALL loads are issued together -- the fix-up loads do not wait for
    // a second HBM round trip behind a barrier.
    // (This is the rare path -- a tile with more structural records than the LDS window holds: one group at a time, no arrays of
    //  sources and loads in flight.  Unrolled four-fold like the LDS path it needed 47 spilled registers, and a kernel with ANY
    //  scratch pays for it at every dispatch -- the queue's scratch is set up before the launch.)
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        const uint32_t g = it * (THREADS * GROUP) + threadIdx.x * GROUP;
        const uint32_t O = tile0 + g;
        if (O >= tile_end) continue;
        int32_t j = r_lo - 1;
        if (any_rec) {
            int32_t lo = r_lo, hi = r_hi + 1;
            while (lo < hi) {
                const int32_t mid = (lo + hi) >> 1;
                if (A.O(mid) <= O) lo = mid + 1; else hi = mid;
            }
            j = lo - 1;
        }
        const bool has = any_rec && j >= r_lo;
        uint32_t oj = 0, ej = 0, sj = 0, mj = 0, xj = 0;
        if (has) A.all(j, oj, ej, sj, mj);
        const bool in_seg = has && O < ej;
        if (in_seg) xj = A.ext(j);
        const PieceSrc ps = piece_src(O, has, in_seg, oj, ej, sj, mj, xj, in, pool);
        u32x4 pv = piece_finish(piece_load(ps), ps.mode, lut);
        if (!in_seg && has && (mj & 0xff) == MSIM_SN && oj == O)        // the group starts on an SNP byte
            pv.x = (pv.x & ~0xffu) | snp_patch(pv.x & 0xff, mj, sj, lut, err);
        *reinterpret_cast<u32x4 *>(tile + g) = pv;
    }
    __syncthreads();
    // (B) fix-up pieces: one lane per PIECE (2 per record: segment, copy run) owns the bytes from the piece start to the end
    // of its 16-B group and overwrites them
    struct Fix { PieceSrc ps; uint32_t p, end, mj, sj; bool on; };
    auto resolve_fix = [&](int32_t q2, Fix &f) {
        f.on = false;
        f.ps.ptr = in; f.ps.mode = 0; f.p = f.end = f.mj = f.sj = 0;
        if (q2 >= 2 * cnt) return;
        const int32_t q = q2 >> 1;
        const bool run = q2 & 1;
        const int32_t j = r_lo + q;
        uint32_t oj, ej, sj, mj;
        A.all(j, oj, ej, sj, mj);
        uint32_t p, end;
        if (!run) {                                       // segment piece [o, e)
            p = oj;
            end = ej;
        } else {                                          // copy run piece [e, next o)
            uint32_t next_o = INF;
            if (q + 1 < cnt) next_o = A.O(j + 1);
            else if ((uint32_t)(j + 1) < n_rec) next_o = A.off ? A.off[j + 1] : recs[j + 1].pos;
            p = ej;
            end = next_o;
        }
        if (end <= p || p < tile0 || p >= tile_end || !(p & 15u)) return;
        const uint32_t xj = run ? 0u : A.ext(j);
        f.on = true;
        f.p = p;
        f.end = min(min(end, (p | 15u) + 1u), tile_end);
        f.mj = run ? mj : 0u;                             // only a copy run can start on an SNP byte
        f.sj = sj;
        f.ps = piece_src(p & ~15u, true, !run, oj, ej, sj, mj, xj, in, pool);
    };
    auto apply_fix = [&](const Fix &f, const Raw5 &raw) {
        if (!f.on) return;
        const u32x4 pv = piece_finish(raw, f.ps.mode, lut);
        int pidx = -1;
        uint32_t pval = 0;
        if ((f.mj & 0xff) == MSIM_SN) {                   // the run starts on the SNP byte
            pidx = (int)(f.p & 15u);
            pval = snp_patch(get_byte(pv, (uint32_t)pidx), f.mj, f.sj, lut, err);
        }
        fix_bytes(tile, tile0, f.p, f.end, pv, pidx, pval);
    };
#pragma unroll 1
    for (int32_t q2 = (int32_t)threadIdx.x; q2 < 2 * cnt; q2 += THREADS) {
        Fix f;
        resolve_fix(q2, f);
        if (f.on) apply_fix(f, piece_load(f.ps));
    }
    __syncthreads();
    // ---- pass C: stream the tile out
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t g = it * (THREADS * GROUP) + threadIdx.x * GROUP;
        if (tile0 + g < tile_end)
            __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(tile + g),
                                        reinterpret_cast<u32x4 *>(out + tile0_64 + g));
    }
}


// ---- LDS path (the tile's records fit the window) --------------------------------------------------------
// The FIRST version of this routine was VALU-issue bound on SV mixes (round-1 PMC: ~1000 VALU instructions per wave, VALU
// pipes ~70 % busy), so everything here is about instruction count.  What it bought, measured on the round-3/4 kernel
// (profiles/r04_pmc_sq_k_rewrite.txt, k_rewrite<140> on a 240 Mb contig of the c3 mix): 564 VALU instructions per wave; of
// the waves' cycles 62.5 % wait for memory (SQ_WAIT_ANY), 19.8 % wait to issue (SQ_WAIT_INST_ANY), 17.7 % execute
// (SQ_ACTIVE_INST_ANY; VALU 9 %, LDS 1.5 % with a bank conflict in 13 % of its cycles) -- the kernel now waits for HBM
// like k_rewrite_snp does (85 % SQ_WAIT_ANY), it no longer issues instructions against it.  The means:
//   * governing record of every 16-B group from an INDEX TABLE (one LDS atomicMax per record + a max-scan,
//     aliased onto the tile buffer) instead of a binary search per group
//   * SNPs do not break a copy run (the run after an SNP continues the same source stream), so only the
//     STRUCTURAL records (everything but SNPs; compacted in order while the window is filled) own pieces and
//     boundary fix-ups; SNP bytes are patched afterwards, one lane per record, like k_rewrite_snp does
//   * fix-up bytes are merged with masked dword LDS atomics, not byte loops
__device__ __forceinline__ void fix_merge(uint8_t *tile, uint32_t tile0, uint32_t p, uint32_t end, const u32x4 &pv) {
    const uint32_t G = p & ~15u;
    const int a = (int)(p - G), b = (int)(end - G);       // bytes [a, b) of the group, 0 < a < b <= 16
    uint32_t *grp = reinterpret_cast<uint32_t *>(tile + (G - tile0));
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const int lo = max(a - 4 * d, 0), hi = min(b - 4 * d, 4);
        if (hi > lo) {
            const uint32_t w = d == 0 ? pv.x : d == 1 ? pv.y : d == 2 ? pv.z : pv.w;
            if (hi - lo == 4) {
                grp[d] = w;
            } else {
                const uint32_t mask = ((1u << (8 * (hi - lo))) - 1u) << (8 * lo);
                atomicAnd(&grp[d], ~mask);                // other lanes own the other bytes of this dword
                atomicOr(&grp[d], w & mask);
            }
        }
    }
}

template <int CAP>
__device__ __forceinline__ void rewrite_tile_lds(const RecWin<CAP> &win, int32_t n_win, uint8_t *tile,
                                                 uint32_t *wtmp, const uint8_t *__restrict__ in,
                                                 uint8_t *__restrict__ out, const uint8_t *__restrict__ pool,
                                                 const uint8_t *lut, uint64_t tile0_64, uint64_t L_out,
                                                 unsigned long long *err, const msim_record *__restrict__ recs,
                                                 const uint32_t *__restrict__ off, int32_t r_lo, int32_t r_hi,
                                                 uint32_t my_snp_o, uint32_t my_snp_aux, uint32_t my_snp_pos) {
    const int32_t n_struct = n_win;
    const uint32_t INF = 0xffffffffu;
    const uint32_t tile0 = (uint32_t)tile0_64;
    const uint32_t tile_end = (uint32_t)min<uint64_t>(tile0_64 + TILE, L_out);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- index table: idx[g] = 1 + window index of the last record whose output offset is <= the start of
    // group g (0: none).  Aliases the tile buffer, which is not written before the barrier below.
    uint32_t *idx = reinterpret_cast<uint32_t *>(tile);      // (zeroed by the caller, in front of the barrier behind the window fill)
    for (int32_t q = threadIdx.x; q < n_win; q += THREADS) {
        const uint32_t o = win.o[q];
        const uint32_t g = o <= tile0 ? 0u : (o - tile0 + 15u) >> 4;
        if (g < (uint32_t)(TILE / GROUP)) atomicMax(&idx[g], (uint32_t)q + 1u);
    }
    __syncthreads();
    {
        uint32_t v[ITERS];                                 // this thread's ITERS consecutive entries
#pragma unroll
        for (int q = 0; q < ITERS; q += 4) {
            const u32x4 w = *reinterpret_cast<u32x4 *>(idx + ITERS * threadIdx.x + q);
            v[q] = w.x; v[q + 1] = w.y; v[q + 2] = w.z; v[q + 3] = w.w;
        }
#pragma unroll
        for (int q = 1; q < ITERS; q++) v[q] = max(v[q], v[q - 1]);
        uint32_t incl = v[ITERS - 1];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= o) incl = max(incl, t);
        }
        uint32_t ex = __shfl_up(incl, 1, 64);
        if (lane == 0) ex = 0;
        if (lane == 63) wtmp[wave] = incl;
        __syncthreads();
        for (int w = 0; w < wave; w++) ex = max(ex, wtmp[w]);
#pragma unroll
        for (int q = 0; q < ITERS; q += 4)
            *reinterpret_cast<u32x4 *>(idx + ITERS * threadIdx.x + q) = u32x4{max(v[q], ex), max(v[q + 1], ex), max(v[q + 2], ex), max(v[q + 3], ex)};
        __syncthreads();
    }
    // ---- resolve: (A) every group from the piece that covers its first byte
    PieceSrc src[ITERS];
    bool live[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t g = it * (THREADS * GROUP) + threadIdx.x * GROUP;
        const uint32_t O = tile0 + g;
        live[it] = O < tile_end;
        const uint32_t j1 = (MSIM_ABL & 8) ? 0u : idx[it * THREADS + threadIdx.x];
        const bool has = j1 != 0;
        const uint32_t q = has ? j1 - 1u : 0u;
        const uint32_t oj = win.o[q], ej = win.e[q], sj = win.s[q], mj = win.m[q] & 0xffffu;
        const bool in_seg = has && O < ej;
        const uint32_t xj = win.x[q];
        src[it] = piece_src(O, has, in_seg, oj, ej, sj, mj, xj, in, pool);
        if (!live[it]) { src[it].ptr = in; src[it].mode = 0; }
    }
    // ---- resolve: (B) a structural piece that starts inside a group owns the bytes up to the group's end.
    // piece u: record sidx[u >> 1]; even = its segment [o, e), odd = its copy run [e, next structural o)
    struct Fix { PieceSrc ps; uint32_t p, end; bool on; };
    auto resolve_fix = [&](int32_t u, Fix &f) {
        f.on = false;
        f.ps.ptr = in; f.ps.mode = 0; f.p = f.end = 0;
        if (u >= 2 * n_struct || (MSIM_ABL & 2)) return;
        const int32_t k = u >> 1;
        const bool run = u & 1;
        const uint32_t q = (uint32_t)k;                    // (the window holds the structural records only, in order)
        const uint32_t oj = win.o[q], ej = win.e[q], sj = win.s[q], mj = win.m[q] & 0xffffu;
        uint32_t p = oj, end = ej;
        if (run) {
            p = ej;
            end = k + 1 < n_struct ? win.o[k + 1] : INF;   // later records start beyond this tile
        }
        if (end <= p || p < tile0 || p >= tile_end || !(p & 15u)) return;
        f.on = true;
        f.p = p;
        f.end = min(min(end, (p | 15u) + 1u), tile_end);
        f.ps = piece_src(p & ~15u, true, !run, oj, ej, sj, mj, win.x[q], in, pool);
    };
    Fix fx;
    resolve_fix((int32_t)threadIdx.x, fx);
    // ---- all loads in flight together; the index table is dead from here on
    Raw5 raw[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) raw[it] = piece_load(src[it]);
    const Raw5 fraw = piece_load(fx.ps);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        if (!live[it]) continue;
        const uint32_t g = it * (THREADS * GROUP) + threadIdx.x * GROUP;
        *reinterpret_cast<u32x4 *>(tile + g) = piece_finish(raw[it], src[it].mode, lut);
    }
    __syncthreads();
    // ---- pass B: structural boundaries
    if (fx.on) fix_merge(tile, tile0, fx.p, fx.end, piece_finish(fraw, fx.ps.mode, lut));
    for (int32_t u = (int32_t)threadIdx.x + THREADS; u < 2 * n_struct; u += THREADS) {   // dense tiles
        Fix f;
        resolve_fix(u, f);
        if (f.on) fix_merge(tile, tile0, f.p, f.end, piece_finish(piece_load(f.ps), f.ps.mode, lut));
    }
    __syncthreads();
    // ---- pass B2: SNP bytes, one lane per record (mutator.py:334-341, 428-463).  The SNPs are not in the window: an
    // SNP does not break a copy run, so nothing above needs it -- a lane reads its record (and offset) from the table,
    // coalesced and fresh in L2 from the window fill.  A tile of a hot spot with a thousand SNPs costs four rounds here.
    // (the first THREADS records of the tile are still in registers from the window fill: my_snp_*)
    if (!(MSIM_ABL & 4) && my_snp_aux != 0xffffffffu && my_snp_o >= tile0 && my_snp_o < tile_end) {
        const uint32_t x = tile[my_snp_o - tile0];
        const uint32_t nb = lut[my_snp_aux * 256 + x];
        if (nb == 0 && my_snp_aux != 0) report_key_error(err, (uint64_t)my_snp_pos, lut[768 + x]);
        else tile[my_snp_o - tile0] = (uint8_t)nb;
    }
    for (int32_t j = r_lo + THREADS + (int32_t)threadIdx.x; j <= ((MSIM_ABL & 4) ? r_lo - 1 : r_hi); j += THREADS) {
        const msim_record r = recs[j];
        if (r.type != MSIM_SN) continue;
        const uint32_t o = off ? off[j] : r.pos;
        if (o < tile0 || o >= tile_end) continue;
        const uint32_t x = tile[o - tile0];
        const uint32_t nb = lut[(uint32_t)r.aux * 256 + x];
        if (nb == 0 && r.aux != 0) report_key_error(err, (uint64_t)r.pos, lut[768 + x]);
        else tile[o - tile0] = (uint8_t)nb;
    }
    __syncthreads();
    // ---- pass C: stream the tile out
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t g = it * (THREADS * GROUP) + threadIdx.x * GROUP;
        if (tile0 + g < tile_end)
            __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(tile + g),
                                        reinterpret_cast<u32x4 *>(out + tile0_64 + g));
    }
}

// One tile of one contig (blk of nwg: the workgroup's place in its contig's grid).
template <int CAP>
__device__ __forceinline__ void rewrite_one_tile(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                 const msim_record *__restrict__ recs, const uint32_t *__restrict__ off,
                                                 const int32_t *__restrict__ first, uint32_t n_rec, uint64_t L_out,
                                                 const uint8_t *__restrict__ pool, const uint8_t *__restrict__ lut_g,
                                                 unsigned long long *err, const uint32_t *__restrict__ dyn, uint32_t blk, uint32_t nwg) {
    __shared__ RecWin<CAP> win;
    __shared__ __attribute__((aligned(16))) uint8_t tile[TILE];
    __shared__ __attribute__((aligned(16))) uint8_t lut[LUT_BYTES];
    uint32_t t = blk;
    if (MSIM_ABL & 16) {                                   // XCD x gets a contiguous range of tiles (bijective for any grid)
        const uint32_t q8 = nwg / 8, r8 = nwg % 8, xcd = t % 8, k8 = t / 8;
        t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k8;
    }
    if (dyn) {                                             // sizes from the device (see k_tile_index); surplus TILES leave (the
        n_rec = dyn[0];                                    // tile a workgroup maps to decides, not its place in the grid)
        L_out = dyn[1];
        if ((uint64_t)t * TILE >= L_out) return;
    }
    const uint64_t tile0 = (uint64_t)t * TILE;
    for (int i = threadIdx.x; i < LUT_BYTES / 4; i += THREADS)
        reinterpret_cast<uint32_t *>(lut)[i] = reinterpret_cast<const uint32_t *>(lut_g)[i];
    const int32_t f0 = first[t], f1 = first[t + 1];
    const bool any_rec = f1 >= 0 && n_rec > 0;
    const int32_t r_lo = f0 < 0 ? 0 : f0;
    const int32_t r_hi = f1;
    const int32_t cnt = any_rec ? r_hi - r_lo + 1 : 0;
    __shared__ uint32_t wtmp[THREADS / 64];
    // The window holds the tile's ANCHOR (record r_lo, whatever its type: the run that enters the tile starts there) and
    // the structural (non-SNP) records behind it, in order -- compacted while the table is read (ballot + popcount).
    // SNPs stay out: they do not break a copy run, and pass B2 patches them straight from the table.  So the window's
    // capacity bounds the STRUCTURAL records of a tile; an SNP hot spot (hundreds of SNPs in a tile) costs no LDS.
    int32_t n_win = 0;
    uint32_t my_snp_o = 0, my_snp_aux = 0xffffffffu, my_snp_pos = 0;   // this lane's SNP among the tile's first THREADS records
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int32_t q0 = 0; q0 < cnt; q0 += THREADS) {
            const int32_t q = q0 + (int32_t)threadIdx.x;
            bool keep = false;
            msim_record r{};
            uint32_t o = 0;
            if (q < cnt) {
                r = recs[r_lo + q];
                o = off ? off[r_lo + q] : r.pos;
                keep = q == 0 || r.type != MSIM_SN;
                if (q0 == 0 && r.type == MSIM_SN) { my_snp_o = o; my_snp_aux = r.aux; my_snp_pos = r.pos; }
            }
            const unsigned long long bal = __ballot(keep);
            if (lane == 0) wtmp[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t base = 0, total = 0;
            for (int w = 0; w < THREADS / 64; w++) { if (w < wave) base += wtmp[w]; total += wtmp[w]; }
            const int32_t k = n_win + (int32_t)base + __popcll(bal & ((1ull << lane) - 1ull));
            if (keep && k < CAP) {
                uint32_t e, sgm;
                rec_view(r, o, e, sgm);
                win.o[k] = o;
                win.e[k] = e;
                win.s[k] = sgm;
                win.m[k] = (uint32_t)r.type | ((uint32_t)r.aux << 8);
                win.x[k] = (r.type == MSIM_TLI && (r.aux & 1)) ? r.stop : r.extra;
            }
            n_win += (int32_t)total;
            if (q0 + THREADS < cnt) __syncthreads();         // (wtmp is reused by the next round -- a tile with more than 256 records)
        }
    }
    const bool in_lds = n_win <= CAP;
#pragma unroll
    for (int q = 0; q < ITERS; q += 4)                       // the LDS path's index table (aliases the tile buffer)
        *reinterpret_cast<u32x4 *>(tile + 4 * (ITERS * threadIdx.x + q)) = u32x4{0, 0, 0, 0};
    __syncthreads();
    if (in_lds) {
        rewrite_tile_lds<CAP>(win, n_win, tile, wtmp, in, out, pool, lut, tile0, L_out, err, recs, off, r_lo, r_hi,
                              my_snp_o, my_snp_aux, my_snp_pos);
    } else {
        RecAccess<false, CAP> A{&win, recs, off, r_lo};
        rewrite_tile<false, CAP>(A, r_hi, any_rec, cnt, tile, in, out, recs, n_rec, pool, lut, tile0, L_out, err);
    }
}
template <int CAP>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(CAP <= 512 ? MSIM_WPE : 4, 8))) void k_rewrite(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                     const msim_record *__restrict__ recs,
                                                     const uint32_t *__restrict__ off,
                                                     const int32_t *__restrict__ first, uint32_t n_rec,
                                                     uint64_t L_out, const uint8_t *__restrict__ pool,
                                                     const uint8_t *__restrict__ lut_g,
                                                     unsigned long long *err, const uint32_t *__restrict__ dyn) {
    rewrite_one_tile<CAP>(in, out, recs, off, first, n_rec, L_out, pool, lut_g, err, dyn, blockIdx.x, gridDim.x);
}

// The rewrite of several contigs in ONE launch: a batch of the counter-based engine (or an emission group of the SNP sampler)
// applies its contigs back to back, and every launch boundary between two rewrite kernels is 5-10 us in which an HBM-bound
// machine moves nothing (24 of them per genome: a seventh of the APPLY phase).  Jobs travel as kernel arguments; a workgroup
// finds its contig by its index (constant indices and scalar selects: no copy of the table, no LDS).
struct RwJob {
    const uint8_t *in; uint8_t *out; const msim_record *recs; const uint32_t *off; const int32_t *first; const uint8_t *pool;
    unsigned long long *err; const uint32_t *dyn; uint64_t L_out; uint32_t n_rec, n_tiles, tile_base, rsv;
};
constexpr int RW_JOBS = 16;
struct RwJobs { RwJob j[RW_JOBS]; uint32_t n, total; };
__device__ __forceinline__ RwJob rw_job_of(const RwJobs &J, uint32_t blk) {
    RwJob T = J.j[0];
#pragma unroll
    for (int q = 1; q < RW_JOBS; q++)
        if ((uint32_t)q < J.n && J.j[q].tile_base <= blk) T = J.j[q];
    return T;
}
template <int CAP>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(CAP <= 512 ? MSIM_WPE : 4, 8))) void k_rewrite_b(RwJobs J, const uint8_t *__restrict__ lut_g) {
    const RwJob T = rw_job_of(J, blockIdx.x);
    const uint32_t blk = blockIdx.x - T.tile_base;
    if (blk >= T.n_tiles) return;
    rewrite_one_tile<CAP>(T.in, T.out, T.recs, T.off, T.first, T.n_rec, T.L_out, T.pool, lut_g, T.err, T.dyn, blk, T.n_tiles);
}

// ------------------------------------------------------------------ 3b. rewrite, SNP-only tables
// No length change: output offset == input position.  The tile is staged in LDS (aligned 16-B loads,
// ds_write_b128), then ONE LANE PER RECORD patches its byte through the LDS LUT -- no per-lane record
// search -- and the tile streams out with aligned 16-B stores.  Any number of records per tile.
__device__ __forceinline__ void rewrite_snp_one_tile(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                     const msim_record *__restrict__ recs, const int32_t *__restrict__ first,
                                                     uint32_t n_rec, uint64_t L, const uint8_t *__restrict__ lut_g,
                                                     unsigned long long *err, const uint32_t *__restrict__ dyn, uint32_t blk) {
    if (dyn) n_rec = dyn[0];
    __shared__ __attribute__((aligned(16))) uint8_t tile[TILE];
    __shared__ __attribute__((aligned(16))) uint8_t lut[1024];
    const uint64_t tile0 = (uint64_t)blk * TILE;
    u32x4 v[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t o = it * (THREADS * GROUP) + threadIdx.x * GROUP;
        v[it] = u32x4{0, 0, 0, 0};
        if (tile0 + o < L) v[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in + tile0 + o));
    }
    reinterpret_cast<uint32_t *>(lut)[threadIdx.x] = reinterpret_cast<const uint32_t *>(lut_g)[threadIdx.x];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t o = it * (THREADS * GROUP) + threadIdx.x * GROUP;
        *reinterpret_cast<u32x4 *>(tile + o) = v[it];
    }
    const int32_t f0 = first[blk], f1 = first[blk + 1];
    __syncthreads();
    // records with tile0 <= pos < tile0 + TILE are (f0 or f0+1) .. f1
    if (f1 >= 0 && n_rec) {
        const int32_t lo = f0 < 0 ? 0 : f0;
        for (int32_t j = lo + (int32_t)threadIdx.x; j <= f1; j += THREADS) {
            const msim_record r = recs[j];
            const uint64_t p = r.pos;
            if (p < tile0 || p >= tile0 + TILE) continue;
            const uint32_t idx = (uint32_t)(p - tile0);
            const uint32_t x = tile[idx];
            const uint32_t nb = lut[(uint32_t)r.aux * 256 + x];
            if (nb == 0 && r.aux != 0) report_key_error(err, p, lut[768 + x]);
            else tile[idx] = (uint8_t)nb;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t o = it * (THREADS * GROUP) + threadIdx.x * GROUP;
        if (tile0 + o < L)
            __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(tile + o), reinterpret_cast<u32x4 *>(out + tile0 + o));
    }
}
__global__ __launch_bounds__(THREADS) void k_rewrite_snp(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                         const msim_record *__restrict__ recs,
                                                         const int32_t *__restrict__ first, uint32_t n_rec,
                                                         uint64_t L, const uint8_t *__restrict__ lut_g,
                                                         unsigned long long *err, const uint32_t *__restrict__ dyn) {
    rewrite_snp_one_tile(in, out, recs, first, n_rec, L, lut_g, err, dyn, blockIdx.x);
}
__global__ __launch_bounds__(THREADS) void k_rewrite_snp_b(RwJobs J, const uint8_t *__restrict__ lut_g) {
    const RwJob T = rw_job_of(J, blockIdx.x);
    const uint32_t blk = blockIdx.x - T.tile_base;
    if (blk >= T.n_tiles) return;
    rewrite_snp_one_tile(T.in, T.out, T.recs, T.first, T.n_rec, T.L_out, lut_g, T.err, T.dyn, blk);
}

// single-lane epilogue: small results go to the pinned host mailbox (no blit-kernel D2H copy)
__global__ void k_publish_u64(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ mailbox) {
    *mailbox = *src;
    __threadfence_system();
}

// ------------------------------------------------------------------ synthetic genome / checksum
__device__ __host__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// base(i) = "ACGT"[(mix64(seed + (i >> 5)) >> (2 * (i & 31))) & 3]; one lane makes 32 bases
__global__ __launch_bounds__(THREADS) void k_synth(uint8_t *__restrict__ dst, uint64_t len, uint64_t seed) {
    const uint64_t g = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    const uint64_t i0 = g * 32;
    if (i0 >= len) return;
    uint64_t bits = mix64(seed + g);
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t x = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t code = (uint32_t)(bits & 3);
            bits >>= 2;
            x |= (uint32_t)("ACGT"[code]) << (q * 8);
        }
        w[k] = x;
    }
    if (i0 + 32 <= len) {
        u32x4 *p = reinterpret_cast<u32x4 *>(dst + i0);
        u32x4 a, b;
        a.x = w[0]; a.y = w[1]; a.z = w[2]; a.w = w[3];
        b.x = w[4]; b.y = w[5]; b.z = w[6]; b.w = w[7];
        p[0] = a; p[1] = b;
    } else {
        for (uint64_t i = i0; i < len; i++) dst[i] = (uint8_t)(w[(i - i0) >> 2] >> (((i - i0) & 3) * 8));
    }
}

// checksum = sum_k mix64(word_k + k * GOLD) over the little-endian 8-byte words of the stream
// (zero padded), wrapping uint64; order sensitive, parallel.  Same function in tests/.
__global__ __launch_bounds__(THREADS) void k_checksum(const uint8_t *__restrict__ src, uint64_t len,
                                                      unsigned long long *__restrict__ sum) {
    const uint64_t nwords = (len + 7) / 8;
    unsigned long long acc = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * THREADS + threadIdx.x; k < nwords; k += (uint64_t)gridDim.x * THREADS) {
        uint64_t w;
        if (k * 8 + 8 <= len) {
            w = *reinterpret_cast<const uint64_t *>(src + k * 8);
        } else {
            w = 0;
            for (uint64_t i = k * 8; i < len; i++) w |= (uint64_t)src[i] << ((i - k * 8) * 8);
        }
        acc += mix64(w + k * 0x9E3779B97F4A7C15ull);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, acc);
}

}  // namespace

int dev_reserve(Ctx *c, void **p, size_t *cap, size_t want_bytes) {
    if (*p && *cap >= want_bytes) return MSIM_OK;
    if (*p) MSIM_HIP(c, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t sz = want_bytes + (want_bytes >> 4) + 256;
    MSIM_HIP(c, hipMalloc(p, sz));
    *cap = sz;
    return MSIM_OK;
}

int ensure_scratch(Ctx *c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return MSIM_OK;
    MSIM_HIP(c, wait_stream(c->stream));          // the old block may still be in use
    MSIM_HIP(c, wait_stream(c->emit_stream));
    if (c->d_scratch) MSIM_HIP(c, hipFree(c->d_scratch));
    c->d_scratch = nullptr;
    c->scratch_bytes = 0;
    size_t want = bytes + (bytes >> 2) + 4096;
    MSIM_HIP(c, hipMalloc(&c->d_scratch, want));
    c->scratch_bytes = want;
    return MSIM_OK;
}

int synth_contig_device(Ctx *c, uint8_t *d_dst, uint64_t len, uint64_t seed) {
    if (len == 0) return MSIM_OK;
    const uint64_t lanes = (len + 31) / 32;
    const uint32_t blocks = (uint32_t)((lanes + THREADS - 1) / THREADS);
    hipLaunchKernelGGL(k_synth, dim3(blocks), dim3(THREADS), 0, c->stream, d_dst, len, seed);
    MSIM_HIP(c, hipGetLastError());
    return MSIM_OK;
}

int checksum_device(Ctx *c, const uint8_t *d_src, uint64_t len, uint64_t *sum) {
    int rc = ensure_scratch(c, 64);
    if (rc) return rc;
    unsigned long long *d_sum = reinterpret_cast<unsigned long long *>(c->d_scratch);
    MSIM_HIP(c, hipMemsetAsync(d_sum, 0, 8, c->stream));
    if (len) {
        const uint64_t nwords = (len + 7) / 8;
        uint32_t blocks = (uint32_t)std::min<uint64_t>((nwords + THREADS - 1) / THREADS, 256 * 16);
        hipLaunchKernelGGL(k_checksum, dim3(blocks), dim3(THREADS), 0, c->stream, d_src, len, d_sum);
        MSIM_HIP(c, hipGetLastError());
    }
    unsigned long long h = 0;
    MSIM_HIP(c, hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, c->stream));
    MSIM_HIP(c, wait_stream(c->stream));
    *sum = h + len * 0x9E3779B97F4A7C15ull;
    return MSIM_OK;
}

// LUT upload lives with the ctx (msim_api.hip); declared here
extern uint8_t *ctx_lut(Ctx *c);

// KeyError words + length-check words of every contig: two asynchronous copies into pinned memory on the emit stream
static int enqueue_err_copies(Ctx *c) {
    const size_t nc = c->contigs.size();
    if (c->cap_h_errs < 2 * nc) {
        if (c->h_errs) MSIM_HIP(c, hipHostFree(c->h_errs));
        c->h_errs = nullptr; c->cap_h_errs = 0;
        MSIM_HIP(c, hipHostMalloc(&c->h_errs, (2 * nc + 64) * sizeof(unsigned long long), hipHostMallocDefault));
        c->cap_h_errs = 2 * nc + 64;
    }
    MSIM_HIP(c, hipMemcpyAsync(c->h_errs, c->d_errs, nc * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->emit_stream));
    MSIM_HIP(c, hipMemcpyAsync(c->h_errs + nc, c->d_errs + MAX_CONTIGS, nc * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->emit_stream));
    return MSIM_OK;
}

int apply_finish(Ctx *c) {
    bool have_errs = false;
    {   // sizes the counter-based engine left on the device (synchronises its streams; no-op when nothing is pending).  The
        // KeyError words ride behind its lanes' work, under the same host wait.
        int state = 0;
        const bool ride = c->fast && !c->pending_apply.empty();
        // The copies ride on the emit stream behind the lanes whose set is still pending (fast_plan_collect joins those).  An
        // APPLY enqueued on a lane whose set was collected earlier (plan, sync, msim_apply_contig, plan more) is on none of them:
        // the emit stream joins every such launch through the event behind it, so that no KeyError / length word is copied
        // before the rewrite that writes it has run.  (A contig whose rewrite rode in another's batched launch -- timing_shared --
        // is covered by that launch's event: the batch's contigs are all pending.)
        if (ride) {                                        // (the last launch of every such stream is enough: a stream runs in order)
            std::vector<hipStream_t> seen;
            for (auto it = c->pending_apply.rbegin(); it != c->pending_apply.rend(); ++it) {
                if (*it < 0 || (size_t)*it >= c->contigs.size()) continue;
                const Contig &g = c->contigs[(size_t)*it];
                hipStream_t s = g.apply_stream;
                if (!g.apply_pending || g.timing_shared || !g.ea2 || !s || s == c->emit_stream) continue;
                if (std::find(seen.begin(), seen.end(), s) != seen.end()) continue;
                seen.push_back(s);
                if (fast_lane_joined_at_collect(c, s)) continue;       // fast_plan_collect joins this lane itself
                MSIM_HIP(c, hipStreamWaitEvent(c->emit_stream, g.ea2, 0));
            }
        }
        const int rc = fast_plan_collect(c, ride ? enqueue_err_copies : nullptr, &state);
        if (rc) { c->pending_apply.clear(); for (auto &g : c->contigs) g.apply_pending = g.dyn_applied = false; return rc; }
        have_errs = state == 1;
    }
    if (c->pending_apply.empty()) return MSIM_OK;
    {   // APPLYs that ran on a stream of their own (counter-based engine)
        hipStream_t last = nullptr;
        for (int idx : c->pending_apply) {
            if (idx < 0 || (size_t)idx >= c->contigs.size()) continue;
            hipStream_t s = c->contigs[(size_t)idx].apply_stream;
            if (s && s != last && s != c->emit_stream) { MSIM_HIP(c, wait_stream(s)); last = s; }
        }
    }
    const size_t nc = c->contigs.size();
    if (!have_errs) {                                      // (two blocking hipMemcpy cost ~40 us each at every step boundary)
        const int rc = enqueue_err_copies(c);
        if (rc) return rc;
        MSIM_HIP(c, wait_stream(c->emit_stream));
    }
    unsigned long long *errs = c->h_errs, *deltas = c->h_errs + nc;
    bool delta_mismatch = false;
    hipEvent_t origin = nullptr;
    std::vector<std::pair<float, float>> spans_all, spans_k;
    auto union_ms = [](std::vector<std::pair<float, float>> &v) {
        std::sort(v.begin(), v.end());
        double sum = 0;
        float lo = 0, hi = 0;
        bool open = false;
        for (const auto &iv : v) {
            if (open && iv.first <= hi) { hi = std::max(hi, iv.second); continue; }
            if (open) sum += hi - lo;
            lo = iv.first; hi = iv.second; open = true;
        }
        if (open) sum += hi - lo;
        return sum;
    };
    for (int idx : c->pending_apply) {
        if (idx < 0 || (size_t)idx >= c->contigs.size()) continue;
        Contig &g = c->contigs[(size_t)idx];
        if (!g.apply_pending) continue;
        g.apply_pending = false;
        if (g.dyn_applied) {                              // (sizes collected above)
            g.dyn_applied = false;
            c->t.bytes_out += g.out_len;
            c->t.records += g.n_rec;
        }
        if (!g.timing_shared) {                            // (else: its rewrite ran inside another contig's batched launch)
            // Launches of the counter-based engine's batches run on lanes of their own and can overlap: the stage times are
            // the time during which at least one launch was in flight (the union of the spans), not the sum of the spans --
            // two rewrites side by side would otherwise be booked twice (1.27 ms of rewriting read 2.4 ms).
            hipEvent_t e0 = g.ea0_is_ea1 ? g.ea1 : g.ea0;
            if (!origin) origin = e0;
            float t0 = 0, t1 = 0, t2 = 0;
            MSIM_HIP(c, hipEventElapsedTime(&t0, origin, e0));
            MSIM_HIP(c, hipEventElapsedTime(&t1, origin, g.ea1));
            MSIM_HIP(c, hipEventElapsedTime(&t2, origin, g.ea2));
            spans_all.push_back({t0, t2});
            spans_k.push_back({t1, t2});
        }
        g.timing_shared = false;
        const unsigned long long h_err = errs[(size_t)idx];
        g.key_error = h_err != ~0ull;
        if (g.key_error) { g.key_pos = h_err >> 8; g.key_base = (uint8_t)(h_err & 0xff); g.key_reported = false; }
        // (off_ready: the offsets came from the planner itself -- there is no second, independent sum to compare)
        if (g.delta_known && !g.off_ready && !g.all_snp && g.n_rec && (long long)deltas[(size_t)idx] != g.known_delta) delta_mismatch = true;
    }
    c->pending_apply.clear();
    c->t.apply_ms += union_ms(spans_all);
    c->t.apply_kernel_ms += union_ms(spans_k);
    if (delta_mismatch) return fail(c, MSIM_ERR_HIP, "internal: planner and device disagree on the mutated length");
    return MSIM_OK;
}

// SNP-only tables: fully asynchronous (no length change, so nothing has to come back to the host
// before the kernel can be launched); the KeyError word and the timings are collected by
// apply_finish at the next synchronising call.  Tables with indels need the scanned total length to
// size the output, so that path synchronises once.
// APPLY of contigs the counter-based engine planned as one batch (plan_fast.hip): their tile indices in one launch, then the
// rewrite kernels back to back on the batch's stream.
// While apply_batch_device applies its contigs, apply_contig_device hands their rewrite launches over instead of making them
// (variant: 0 SNP-only, 1 small window, 2 large window): they go out as ONE launch per variant.
// (the collector hangs on the context -- Ctx::rw_collect -- so that two contexts driven from two threads never see each other's)
struct RwPending { RwJob job; int variant; int contig; };

static_assert((TILE & (TILE - 1)) == 0, "the expansion's tile index shifts by log2(TILE)");
int apply_prepare_tile_index(Ctx *c, Contig &g, hipStream_t st, int32_t **first, uint32_t *n_tiles, uint32_t *tile_shift,
                             unsigned long long **err) {
    const uint32_t nt = (uint32_t)((g.len + TILE - 1) / TILE);      // (SNP-only: the mutated contig is as long as the contig)
    if (g.cap_first < (size_t)(nt + 1) * sizeof(int32_t)) MSIM_HIP(c, wait_stream(st));
    const int rc = dev_reserve(c, (void **)&g.d_first, &g.cap_first, (size_t)(nt + 1) * sizeof(int32_t));
    if (rc) return rc;
    *first = g.d_first;
    *n_tiles = nt;
    *tile_shift = (uint32_t)__builtin_ctz((unsigned)TILE);
    *err = c->d_errs + g.index;
    g.tile_index_by_plan = nt != 0;
    return MSIM_OK;
}

int apply_batch_device(Ctx *c, const std::vector<int> &ids, bool batch_rewrites) {
    TileJobs J;
    J.n_jobs = 0; J.total = 0;
    hipStream_t st = nullptr;
    auto launch = [&]() {
        if (J.n_jobs) hipLaunchKernelGGL(k_tile_index_batch, dim3((J.total + THREADS - 1) / THREADS), dim3(THREADS), 0, st, J);
        J.n_jobs = 0; J.total = 0;
    };
    std::vector<int> marked;
    int batch_first = -1;
    bool any_tile_job = false;                             // (none: every contig's tile index comes with its records -- no start event)
    for (int id : ids) {
        const Contig &g = c->contigs[(size_t)id];
        if (!(g.tile_index_by_plan && g.all_snp && !g.d_dyn)) any_tile_job = true;
    }
    for (int id : ids) {
        Contig &g = c->contigs[(size_t)id];
        const uint32_t *dyn = g.d_dyn;
        const uint32_t n = (uint32_t)(dyn ? g.n_rec_cap : g.n_rec);
        const uint64_t out_bound = g.all_snp ? g.len : dyn ? g.out_cap_len
                                   : (g.off_ready && g.delta_known) ? (uint64_t)((long long)g.len + g.known_delta) : 0;
        // (only what a device engine planned: a device-sized table, an SNP-only one, or one that came with its offsets and its
        //  length -- the host-chain engines' --, on the batch's stream, not in flight)
        if (!g.apply_stream || !n || !out_bound || g.apply_pending || (st && g.apply_stream != st)) continue;
        // (a mutated contig of 4 GiB or more, or a negative length: left unmarked -- apply_contig_device takes it alone and
        //  refuses it before anything is sized from the value)
        if (!g.all_snp && !dyn && ((long long)g.len + g.known_delta < 0 || (long long)g.len + g.known_delta >= (1ll << 32))) continue;
        st = g.apply_stream;
        const uint32_t n_tiles = (uint32_t)((out_bound + TILE - 1) / TILE);
        if (g.cap_first < (size_t)(n_tiles + 1) * sizeof(int32_t)) MSIM_HIP(c, wait_stream(st));
        int rc = dev_reserve(c, (void **)&g.d_first, &g.cap_first, (size_t)(n_tiles + 1) * sizeof(int32_t));
        if (rc) return rc;
        if (!g.ea0) {
            MSIM_HIP(c, hipEventCreate(&g.ea0));
            MSIM_HIP(c, hipEventCreate(&g.ea1));
            MSIM_HIP(c, hipEventCreate(&g.ea2));
        }
        // ONE start event for the batch (on its first contig, in front of the tile index): an event per contig is a packet per
        // contig on the queue -- twelve of them stood between a batch's tile index and its rewrite launch, 50 us of an idle device
        g.ea0_is_ea1 = false;
        if (marked.empty()) {
            if (any_tile_job) MSIM_HIP(c, hipEventRecord(g.ea0, st));
            else g.ea0_is_ea1 = true;
            batch_first = id;
        }
        if (g.tile_index_by_plan && g.all_snp && !dyn) {   // its tile index is already on its way (k_bitmap_expand_tiles_b)
            g.tile_index_by_plan = false;
            g.tile_index_done = true;
            marked.push_back(id);
            continue;
        }
        TileJob &T = J.j[J.n_jobs++];
        T.off = g.all_snp ? nullptr : g.d_off; T.recs = g.d_recs; T.dyn = dyn; T.first = g.d_first; T.err = c->d_errs + g.index;
        T.n = n; T.n_entries = n_tiles + 1; T.entry_base = J.total; T.rsv = 0;
        J.total += n_tiles + 1;
        g.tile_index_done = true;
        marked.push_back(id);
        if (J.n_jobs == 32) launch();
    }
    launch();
    MSIM_HIP(c, hipGetLastError());
    int rc = MSIM_OK;
    std::vector<RwPending> rw;
    static const bool no_rw_batch = getenv("MSIM_NO_REWRITE_BATCH") != nullptr;
    c->rw_collect = (no_rw_batch || !batch_rewrites) ? nullptr : &rw;
    for (int id : ids) {
        Contig &g = c->contigs[(size_t)id];
        if (!rc) rc = apply_contig_device(c, g);
        g.tile_index_done = false;
    }
    c->rw_collect = nullptr;
    for (int id : marked) c->contigs[(size_t)id].tile_index_done = false;
    if (rc) return rc;
    // ---- the collected rewrites: one launch per kernel variant and RW_JOBS contigs; the first contig's events time the launch
    for (int variant = 0; variant < 3; variant++) {
        RwJobs R;
        R.n = 0; R.total = 0;
        int leader = -1;
        auto go = [&]() -> int {
            if (!R.n) return MSIM_OK;
            Contig &L = c->contigs[(size_t)leader];
            if (leader != batch_first) {                    // (a second kernel variant's launch: its own span)
                if (any_tile_job) MSIM_HIP(c, hipEventRecord(L.ea0, st));
                else L.ea0_is_ea1 = true;
            }
            MSIM_HIP(c, hipEventRecord(L.ea1, st));
            if (variant == 0) hipLaunchKernelGGL(k_rewrite_snp_b, dim3(R.total), dim3(THREADS), 0, st, R, ctx_lut(c));
            else if (variant == 1) hipLaunchKernelGGL(k_rewrite_b<REC_CAP_SMALL>, dim3(R.total), dim3(THREADS), 0, st, R, ctx_lut(c));
            else hipLaunchKernelGGL(k_rewrite_b<REC_CAP>, dim3(R.total), dim3(THREADS), 0, st, R, ctx_lut(c));
            MSIM_HIP(c, hipGetLastError());
            MSIM_HIP(c, hipEventRecord(L.ea2, st));
            L.timing_shared = false;
            c->t.apply_launches += 1;
            R.n = 0; R.total = 0; leader = -1;
            return MSIM_OK;
        };
        for (const RwPending &p : rw) {
            if (p.variant != variant) continue;
            if (leader < 0) leader = p.contig;
            RwJob &T = R.j[R.n++];
            T = p.job;
            T.tile_base = R.total;
            R.total += T.n_tiles;
            if (R.n == RW_JOBS && (rc = go())) return rc;
        }
        if ((rc = go())) return rc;
    }
    return MSIM_OK;
}

int apply_contig_device(Ctx *c, Contig &g) {
    // (d_dyn: planned by the counter-based engine with types beyond SNPs -- the record count and the mutated length sit in
    //  device memory; `n` and the output length below are then the bounds the plan allocated for, and the kernels read
    //  the exact values themselves)
    const uint32_t *dyn = g.d_dyn;
    const uint32_t n = (uint32_t)(dyn ? g.n_rec_cap : g.n_rec);
    hipStream_t st = g.apply_stream ? g.apply_stream : c->emit_stream;
    if (!g.ea0) {
        MSIM_HIP(c, hipEventCreate(&g.ea0));
        MSIM_HIP(c, hipEventCreate(&g.ea1));
        MSIM_HIP(c, hipEventCreate(&g.ea2));
    }
    if (g.apply_pending) {                                 // same contig applied again before collection
        int rc = apply_finish(c);
        if (rc) return rc;
    }
    unsigned long long *d_err = c->d_errs + g.index;
    if (!(c->rw_collect && g.tile_index_done)) { MSIM_HIP(c, hipEventRecord(g.ea0, st)); g.ea0_is_ea1 = false; }   // (batched: apply_batch_device recorded the batch's)
    // ---- 1. output offsets (skipped for an SNP-only table: no length change, offset == position)
    long long total_delta = 0;
    const uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    const uint32_t *d_off = nullptr;
    if (dyn) {
        total_delta = g.all_snp ? 0 : (long long)g.out_cap_len - (long long)g.len;
        d_off = g.all_snp ? nullptr : g.d_off;
    } else if (n && !g.all_snp && g.off_ready) {           // planned on the device: the offsets came with the records
        total_delta = g.known_delta;
        d_off = g.d_off;
    } else if (n && !g.all_snp) {
        int rc = dev_reserve(c, (void **)&g.d_off, &g.cap_off, (size_t)n * sizeof(uint32_t));
        if (rc) return rc;
        rc = ensure_scratch(c, (size_t)(nb + 1) * sizeof(long long));
        if (rc) return rc;
        long long *d_sums = reinterpret_cast<long long *>(c->d_scratch);
        hipLaunchKernelGGL(k_delta_reduce, dim3(nb), dim3(THREADS), 0, st, g.d_recs, n, d_sums);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, d_sums, nb);
        hipLaunchKernelGGL(k_offsets, dim3(nb), dim3(THREADS), 0, st, g.d_recs, n, d_sums, g.d_off);
        MSIM_HIP(c, hipGetLastError());
        if (g.delta_known) {
            // the planner summed the length changes while it ran the boundary chain: no round trip.  The
            // device total goes to this contig's check word and is compared when the APPLY is collected.
            total_delta = g.known_delta;
            hipLaunchKernelGGL(k_publish_u64, dim3(1), dim3(1), 0, st,
                               reinterpret_cast<const unsigned long long *>(d_sums + nb),
                               c->d_errs + MAX_CONTIGS + g.index);
            MSIM_HIP(c, hipGetLastError());
        } else {
            hipLaunchKernelGGL(k_publish_u64, dim3(1), dim3(1), 0, st,
                               reinterpret_cast<const unsigned long long *>(d_sums + nb), c->h_mail);
            MSIM_HIP(c, hipGetLastError());
            MSIM_HIP(c, wait_stream(st));
            total_delta = (long long)*c->h_mail;
        }
        d_off = g.d_off;
    }
    const long long out_len_ll = (long long)g.len + total_delta;
    if (out_len_ll < 0 || (uint64_t)out_len_ll >= (1ull << 32))
        return fail(c, MSIM_ERR_UNSUPPORTED, "mutated contig of 4 GiB or more");
    g.out_len = (uint64_t)out_len_ll;
    if (g.cap_out < g.out_len + PAD) {                     // replacing a buffer: nothing may be in flight on it
        MSIM_HIP(c, wait_stream(st));
        int rc = dev_reserve(c, (void **)&g.d_out, &g.cap_out, g.out_len + PAD);
        if (rc) return rc;
    }
    // ---- 2. tile index
    const uint32_t n_tiles = (uint32_t)((g.out_len + TILE - 1) / TILE);
    int32_t *d_first = nullptr;
    if (n_tiles) {
        if (g.apply_stream) {
            if (g.cap_first < (size_t)(n_tiles + 1) * sizeof(int32_t)) MSIM_HIP(c, wait_stream(st));
            int rc = dev_reserve(c, (void **)&g.d_first, &g.cap_first, (size_t)(n_tiles + 1) * sizeof(int32_t));
            if (rc) return rc;
            d_first = g.d_first;
        } else {
            int rc = ensure_scratch(c, (size_t)(n_tiles + 1) * sizeof(int32_t) + 64);
            if (rc) return rc;
            d_first = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(c->d_scratch) + 64);
        }
        if (!g.tile_index_done) {                          // (else: apply_batch_device did it for the whole batch)
            hipLaunchKernelGGL(k_tile_index, dim3((n_tiles + 1 + THREADS - 1) / THREADS), dim3(THREADS), 0, st,
                               d_off, g.d_recs, n, d_first, n_tiles + 1, dyn, d_err);
            MSIM_HIP(c, hipGetLastError());
        }
    } else {
        MSIM_HIP(c, hipMemsetAsync(d_err, 0xff, 8, st));
    }
    // ---- 3. rewrite
    std::vector<RwPending> *collect = static_cast<std::vector<RwPending> *>(c->rw_collect);
    const bool hand_over = collect && g.tile_index_done && n_tiles;     // (apply_batch_device launches it with its batch)
    g.timing_shared = hand_over;
    if (!hand_over) MSIM_HIP(c, hipEventRecord(g.ea1, st));
    if (hand_over) {
        const bool small_win = dyn ? g.n_struct_est * 2 < (uint64_t)n_tiles * REC_CAP_SMALL
                                   : (uint64_t)n * 5 + 64 * 4 < (uint64_t)n_tiles * REC_CAP_SMALL * 4;
        RwPending p;
        memset(&p, 0, sizeof p);
        p.job.in = g.d_in + PAD; p.job.out = g.d_out; p.job.recs = g.d_recs; p.job.off = d_off;
        p.job.first = d_first; p.job.pool = g.d_pool ? g.d_pool + PAD : nullptr; p.job.err = d_err; p.job.dyn = dyn;
        p.job.L_out = g.out_len; p.job.n_rec = n; p.job.n_tiles = n_tiles;
        p.variant = g.all_snp ? 0 : (small_win ? 1 : 2);
        p.contig = g.index;
        collect->push_back(p);
    } else if (n_tiles) {
        // small window: mean records per tile < 80 % of it (dyn: only the STRUCTURAL records take window slots, and their
        // expected number is what the host knows -- mean structural candidates per tile below half the window)
        const bool small_win = dyn ? g.n_struct_est * 2 < (uint64_t)n_tiles * REC_CAP_SMALL
                                   : (uint64_t)n * 5 + 64 * 4 < (uint64_t)n_tiles * REC_CAP_SMALL * 4;
        if (g.all_snp)
            hipLaunchKernelGGL(k_rewrite_snp, dim3(n_tiles), dim3(THREADS), 0, st, g.d_in + PAD, g.d_out, g.d_recs,
                               d_first, n, g.out_len, ctx_lut(c), d_err, dyn);
        else if (small_win)
            hipLaunchKernelGGL(k_rewrite<REC_CAP_SMALL>, dim3(n_tiles), dim3(THREADS), 0, st, g.d_in + PAD, g.d_out, g.d_recs,
                               d_off, d_first, n, g.out_len, g.d_pool + PAD, ctx_lut(c), d_err, dyn);
        else
            hipLaunchKernelGGL(k_rewrite<REC_CAP>, dim3(n_tiles), dim3(THREADS), 0, st, g.d_in + PAD, g.d_out, g.d_recs,
                               d_off, d_first, n, g.out_len, g.d_pool + PAD, ctx_lut(c), d_err, dyn);
        MSIM_HIP(c, hipGetLastError());
    }
    if (!hand_over) {
        MSIM_HIP(c, hipEventRecord(g.ea2, st));
        c->t.apply_launches += n_tiles ? 1 : 0;
    }
    c->t.bytes_in += g.len;
    if (!dyn) {                                            // (dyn: counted when the sizes are collected, apply_finish)
        c->t.bytes_out += g.out_len;
        c->t.records += n;
    }
    g.applied = true;
    g.apply_pending = true;
    g.dyn_applied = dyn != nullptr;
    g.key_error = false;
    c->pending_apply.push_back(g.index);
    if (g.all_snp || g.delta_known || dyn) return MSIM_OK; // asynchronous
    int rc = apply_finish(c);
    if (rc) return rc;
    if (g.key_error) { g.key_reported = true; return fail(c, MSIM_ERR_KEY, std::string("KeyError: '") + (char)g.key_base + "'"); }
    return MSIM_OK;
}

}  // namespace msim
