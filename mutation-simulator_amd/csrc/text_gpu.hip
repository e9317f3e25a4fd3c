// Text on the device: VCF record lines, FASTA line framing (egress) and FASTA body gathering (ingest).
// gfx950 (MI355X) only.                                                   SURVEY.md section 8(f) rows 1-2.
//
// The record table and the mutated stream live in HBM; for an SV mix the VCF text is as large as the
// genome, so rendering it on the host means one core pushing gigabytes through a
// byte-wise formatter.  Here:
//   VCF    k_vcf_lines<false> : a wave per 64 consecutive records.  Short lines -- every SNP, every record spanning <= 24
//                               bases -- are formatted by ONE lane each (LSink), long REF / ALT by the whole wave, one
//                               record after the other (WSink: scalar fields by the low lanes, copies 4 bytes per lane)
//                               -> length of every line (0 = suppressed, REF == ALT)
//          k_len_* / k_scan_u64: exclusive u64 scan -> byte offset of every line
//          k_vcf_lines<true>  : same walk, writing (raw, ambiguity-converted, or reverse-complemented REF / ALT)
//          follows mutator.py:334-421 (record construction) and vcf_writer.py:44-52,118-126 (the line)
//   FASTA  k_frame  : mutated stream -> text with '\n' after every `bpl` bases (fasta_writer.py:40-58); 16 output bytes
//                     per lane from two unaligned 16-byte loads, the newline shifted in
//          k_gather : FASTA body text (uniform line width, as pyfaidx requires) -> upper-cased uint8 bases
//                     (what pyfaidx hands the reference with sequence_always_upper=True, util.py:84-88); 16 bases per lane,
//                     the one line terminator a group can straddle squeezed out
//                               the write pass formats a wave's lines into LDS where all 64 are short and stores the stretch
//                               16 aligned bytes per lane; REF / ALT of SNP records come from the length pass (no second gather)
//   (round 4, profiles/r04_kernel_stats_cli_*.txt: k_gather / k_frame 4.8-5.2 TB/s; k_vcf_lines' write pass on SNP tables
//    4.0 TB/s [1.26 with a lane's own byte stores and the general kernel's registers], its length pass 1.4 TB/s -- it gathers
//    one base per record from the contig)
//   IT     k_splice : interchromosomal translocation of one contig -- segments of two contigs taken alternately
//                     (it_mutator.py:121-146 __write_with_bp)
// Byte/integer work, HBM-bound; no MFMA.
#include <cstring>

#include "ctx.h"

namespace msim {

uint8_t *ctx_lut(Ctx *c);             // msim_api.hip: 1280-byte translation table (layout in apply.hip)

namespace {

constexpr int TX_THREADS = 256;
constexpr int TX_WAVES = TX_THREADS / 64;

__device__ __forceinline__ int ndigits(unsigned long long v) {
    int k = 1;
    unsigned long long p = 10;
    while (v >= p && k < 20) { p *= 10; k++; }
    return k;
}

// Two sinks with one interface.  WSink: wave-uniform -- every lane carries the same offset n; small fields are written by
// the low lanes, bulk copies by all 64 (4 bytes per lane while 256 are left).  LSink: one lane formats its own line
// sequentially -- for records whose REF / ALT are a few bases (the reference's default SV lengths are 1..3, defaults.py:29-32).
template <bool WRITE>
struct WSink {
    char *p;
    unsigned long long n;
    uint32_t lane;
    __device__ __forceinline__ uint32_t first() const { return lane; }
    __device__ __forceinline__ uint32_t step() const { return 64; }
    __device__ __forceinline__ bool any(bool d) const { return __ballot(d) != 0ull; }
    __device__ __forceinline__ void put(char c) {
        if (WRITE && lane == 0) p[n] = c;
        n++;
    }
    template <int N>
    __device__ __forceinline__ void lit(const char (&s)[N]) {            // N - 1 characters
        if (WRITE && lane < (uint32_t)(N - 1)) p[n + lane] = s[lane];
        n += N - 1;
    }
    __device__ __forceinline__ void num(unsigned long long v) {
        const int k = ndigits(v);
        if (WRITE && lane < (uint32_t)k) {
            unsigned long long q = v;
            for (int i = 0; i < k - 1 - (int)lane; i++) q /= 10;
            p[n + lane] = (char)('0' + (int)(q % 10));
        }
        n += (unsigned)k;
    }
    // mode 0 raw, 1 conv(x), 2 comp(conv(x)) read BACKWARDS from src (src points at the LAST source byte)
    __device__ __forceinline__ void bulk(const uint8_t *__restrict__ src, unsigned long long len, int mode,
                                         const uint8_t *lut) {
        if (WRITE) {
            unsigned long long i = 0;
            for (; i + 256 <= len; i += 256) {                           // unaligned dword per lane (gfx950 global accesses may be)
                const unsigned long long k = i + 4ull * lane;
                uint32_t w;
                if (mode == 2) {
                    uint32_t v;
                    __builtin_memcpy(&v, src - k - 3, 4);
                    w = (uint32_t)lut[1024 + (v >> 24)] | ((uint32_t)lut[1024 + ((v >> 16) & 255)] << 8) |
                        ((uint32_t)lut[1024 + ((v >> 8) & 255)] << 16) | ((uint32_t)lut[1024 + (v & 255)] << 24);
                } else {
                    __builtin_memcpy(&w, src + k, 4);
                    if (mode == 1)
                        w = (uint32_t)lut[768 + (w & 255)] | ((uint32_t)lut[768 + ((w >> 8) & 255)] << 8) |
                            ((uint32_t)lut[768 + ((w >> 16) & 255)] << 16) | ((uint32_t)lut[768 + (w >> 24)] << 24);
                }
                __builtin_memcpy(p + n + k, &w, 4);
            }
            for (i += lane; i < len; i += 64) {
                uint8_t c;
                if (mode == 0) c = src[i];
                else if (mode == 1) c = lut[768 + src[i]];
                else c = lut[1024 + *(src - i)];
                p[n + i] = (char)c;
            }
        }
        n += len;
    }
};

template <bool WRITE>
struct LSink {
    char *p;
    unsigned long long n;
    __device__ __forceinline__ uint32_t first() const { return 0; }
    __device__ __forceinline__ uint32_t step() const { return 1; }
    __device__ __forceinline__ bool any(bool d) const { return d; }
    __device__ __forceinline__ void put(char c) {
        if (WRITE) p[n] = c;
        n++;
    }
    template <int N>
    __device__ __forceinline__ void lit(const char (&s)[N]) {
        if (WRITE) {
#pragma unroll
            for (int i = 0; i < N - 1; i++) p[n + i] = s[i];
        }
        n += N - 1;
    }
    __device__ __forceinline__ void num(unsigned long long v) {
        const int k = ndigits(v);
        if (WRITE)
            for (int q = k - 1; q >= 0; q--) { p[n + q] = (char)('0' + (int)(v % 10)); v /= 10; }
        n += (unsigned)k;
    }
    __device__ __forceinline__ void bulk(const uint8_t *__restrict__ src, unsigned long long len, int mode,
                                         const uint8_t *lut) {
        if (WRITE) {
            for (unsigned long long i = 0; i < len; i++) {
                uint8_t c;
                if (mode == 0) c = src[i];
                else if (mode == 1) c = lut[768 + src[i]];
                else c = lut[1024 + *(src - i)];
                p[n + i] = (char)c;
            }
        }
        n += len;
    }
};

template <class S>
__device__ __forceinline__ void line_head(S &s, const uint8_t *name, uint32_t name_len, unsigned long long start) {
    s.bulk(name, name_len, 0, nullptr);
    s.put('\t');
    s.num(start);
    s.lit("\t.\t");
}
// svtype: 0 none (SNP), 1 INS, 2 DEL, 3 INV, 4 DUP, 5 INS:ME, 6 DEL:ME
template <class S>
__device__ __forceinline__ void line_tail(S &s, int svtype, unsigned long long end, unsigned long long len) {
    s.lit("\t.\t.\t");
    if (svtype) {
        s.lit("SVTYPE=");
        switch (svtype) {
            case 1: s.lit("INS"); break;
            case 2: s.lit("DEL"); break;
            case 3: s.lit("INV"); break;
            case 4: s.lit("DUP"); break;
            case 5: s.lit("INS:ME"); break;
            default: s.lit("DEL:ME"); break;
        }
        s.lit(";END=");
        s.num(end);
        s.lit(";SVLEN=");
        s.num(len);
    } else {
        s.put('.');
    }
    s.lit("\tGT\t1\n");
}

// One record's line into a sink (mutator.py:334-421 builds the record, vcf_writer.py:118-126 the line).
template <class S>
__device__ __forceinline__ void format_record(S &s, const msim_record &r, const uint8_t *__restrict__ pool,
                                              const uint8_t *__restrict__ in, unsigned long long L,
                                              const uint8_t *__restrict__ name, uint32_t name_len, const uint8_t *lut) {
    const unsigned long long pos = r.pos, stop = r.stop;
    switch (r.type) {
        case MSIM_SN: {                                                  // mutator.py:334-341
            const uint8_t x = in[pos];
            const uint8_t ref = lut[768 + x];
            const uint8_t alt = lut[(uint32_t)r.aux * 256 + x];          // ti / tv column of conv(x)
            if (ref == alt) break;                                       // vcf_writer.py:123
            line_head(s, name, name_len, pos + 1);
            s.put((char)ref); s.put('\t'); s.put((char)alt);
            line_tail(s, 0, 0, 0);
            break;
        }
        case MSIM_IN: {                                                  // mutator.py:343-358
            const unsigned long long len = stop + 1 - pos;
            const uint8_t *ins = pool + r.extra;
            if (pos > 0) {
                const char ref = (char)lut[768 + in[pos - 1]];
                line_head(s, name, name_len, pos);
                s.put(ref); s.put('\t'); s.put(ref); s.bulk(ins, len, 0, lut);
                line_tail(s, 1, pos, len);
            } else {
                const char ref = (char)lut[768 + in[0]];
                line_head(s, name, name_len, 1);
                s.put(ref); s.put('\t'); s.bulk(ins, len, 0, lut); s.put(ref);
                line_tail(s, 1, 1, len);
            }
            break;
        }
        case MSIM_TLI: {                                                 // mutator.py:401-421
            const unsigned long long src = r.extra;
            const unsigned long long hi = stop + 1 < L ? stop + 1 : L;
            const unsigned long long ilen = hi > src ? hi - src : 0;
            const bool rev = r.aux & 1, after = r.aux & 2;
            if (ilen == 0) break;                                        // REF == ALT: suppressed
            const unsigned long long start = after ? pos : pos + 1;
            const char ref = (char)lut[768 + in[after ? pos - 1 : pos]];
            line_head(s, name, name_len, start);
            s.put(ref); s.put('\t');
            if (after) s.put(ref);
            if (rev) s.bulk(in + hi - 1, ilen, 2, lut);
            else s.bulk(in + src, ilen, 1, lut);
            if (!after) s.put(ref);
            line_tail(s, 5, start, ilen);
            break;
        }
        case MSIM_TL:
        case MSIM_DE: {                                                  // mutator.py:360-377
            unsigned long long start = pos, end = stop + 1, lo = pos - 1;
            if (pos == 0) { start = 1; end = stop + 2; lo = 0; }
            const unsigned long long hi = end < L ? end : L;             // slice clamps at len(sequence)
            line_head(s, name, name_len, start);
            s.bulk(in + lo, hi - lo, 1, lut);
            s.put('\t');
            s.put((char)lut[768 + in[pos > 0 ? lo : hi - 1]]);           // REF[0] / REF[-1]
            line_tail(s, r.type == MSIM_DE ? 2 : 6, end, stop - pos + 1);
            break;
        }
        case MSIM_IV: {                                                  // mutator.py:379-387
            const unsigned long long len = stop - pos + 1;
            bool diff = false;                                           // REF == ALT (palindrome): suppressed
            for (unsigned long long q = s.first(); q < len; q += s.step())
                diff |= lut[768 + in[pos + q]] != lut[1024 + in[stop - q]];
            if (!s.any(diff)) break;
            line_head(s, name, name_len, pos + 1);
            s.bulk(in + pos, len, 1, lut);
            s.put('\t');
            s.bulk(in + stop, len, 2, lut);
            line_tail(s, 3, stop + 1, 0);
            break;
        }
        case MSIM_DU: {                                                  // mutator.py:389-399 (REF not converted)
            const unsigned long long len = stop - pos + 1;
            line_head(s, name, name_len, pos + 1);
            s.bulk(in + pos, len, 0, lut); s.put('\t');
            s.bulk(in + pos, len, 0, lut); s.bulk(in + pos, len, 0, lut);
            line_tail(s, 4, pos + len, len);
            break;
        }
        default: break;
    }
}

constexpr uint32_t VCF_LANE_SPAN = 24;        // records spanning at most this many bases are formatted by one lane
constexpr uint32_t VCF_STAGE = 4096;          // LDS per wave for the lines of its 64 records (write pass)

// the line of an SNP record (mutator.py:334-341, vcf_writer.py:118-126) at p: name \t POS \t . \t REF \t ALT \t . \t . \t . \t GT \t 1 \n
template <class P>
__device__ __forceinline__ void snp_line(P p, const uint8_t *__restrict__ name, uint32_t name_len, unsigned long long start, int nd,
                                         uint8_t ref, uint8_t alt) {
    for (uint32_t q = 0; q < name_len; q++) p[q] = (char)name[q];
    p += name_len;
    *p++ = '\t';
    unsigned long long v = start;
    for (int q = nd - 1; q >= 0; q--) { p[q] = (char)('0' + (int)(v % 10)); v /= 10; }
    p += nd;
    const char tail[18] = {'\t', '.', '\t', (char)ref, '\t', (char)alt, '\t', '.', '\t', '.', '\t', '.', '\t', 'G', 'T', '\t', '1', '\n'};
#pragma unroll
    for (int q = 0; q < 18; q++) p[q] = tail[q];
}

// len_io: the lines' lengths -- written by the length pass, read by the write pass.  ra: REF | ALT << 8 of every SNP record, left
// by the length pass (which has to look at the base anyway: REF == ALT is suppressed) so that the write pass does not gather
// the contig's bases at 2 M scattered positions a second time.
// ALL_SNP (Contig::all_snp: the table holds SNP records only -- the SNP sampler's, `-sn` alone): the kernel without the other
// types' formatter -- 52 registers instead of 139, eight waves per SIMD instead of three.
template <bool WRITE, bool ALL_SNP>
__global__ __launch_bounds__(TX_THREADS) void k_vcf_lines(const msim_record *__restrict__ recs, uint32_t n_rec,
                                                          const uint8_t *__restrict__ pool,
                                                          const uint8_t *__restrict__ in, unsigned long long L,
                                                          const uint8_t *__restrict__ name, uint32_t name_len,
                                                          const uint8_t *__restrict__ lut_g,
                                                          uint32_t *__restrict__ len_io, uint16_t *__restrict__ ra,
                                                          const unsigned long long *__restrict__ off,
                                                          char *__restrict__ text) {
    __shared__ uint8_t lut[1280];
    __shared__ __attribute__((aligned(16))) char stage[WRITE ? TX_WAVES : 1][WRITE ? VCF_STAGE : 16];
    for (int i = threadIdx.x; i < 1280 / 4; i += TX_THREADS)
        reinterpret_cast<uint32_t *>(lut)[i] = reinterpret_cast<const uint32_t *>(lut_g)[i];
    __syncthreads();
    // A wave takes 64 consecutive records.  Short lines -- every SNP (~25 bytes) and every record of a few bases -- are formatted by
    // one LANE each (a wave per such line -- round 1 -- spent ~15 instructions of 64 lanes on 25 bytes: 135 GB/s, 1.7 % of HBM, and
    // a twelfth of a CLI run with -sn 0.01); records with a long REF / ALT are taken by the whole wave, one after the other, their
    // fields broadcast.
    // Write pass: the lines of a wave's records are adjacent in the text.  Where all of them are short, the lanes format into
    // LDS and the wave stores the stretch 16 aligned bytes per lane (a lane's own 25-byte line goes out as 25 one-byte stores
    // otherwise: 64 bytes per store instruction instead of 1024).
    const uint32_t lane_id = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base_rec = (blockIdx.x * TX_WAVES + wave) * 64;
    if (base_rec >= n_rec) return;
    const uint32_t mine = base_rec + lane_id;
    const bool valid = mine < n_rec;
    msim_record my{};
    if (valid) my = recs[mine];
    bool by_lane = false;
    if (valid && (ALL_SNP || my.type == MSIM_SN)) {
        by_lane = true;
    } else if (valid) {
        const unsigned long long hi = (unsigned long long)my.stop + 1 < L ? (unsigned long long)my.stop + 1 : L;
        const unsigned long long span = my.type == MSIM_TLI ? (hi > my.extra ? hi - my.extra : 0)
                                                            : (unsigned long long)my.stop - my.pos + 1;
        by_lane = span <= VCF_LANE_SPAN;
    }
    unsigned long long my_off = 0, start0 = 0;
    uint32_t my_len = 0, stretch = 0, phase = 0;
    bool staged = false;
    if (WRITE) {
        if (valid) { my_off = off[mine]; my_len = len_io[mine]; }
        const uint32_t last = min(63u, n_rec - 1 - base_rec);
        start0 = __shfl(my_off, 0, 64);
        const unsigned long long end = __shfl(my_off + my_len, (int)last, 64);
        stretch = (uint32_t)min(end - start0, (unsigned long long)(2 * VCF_STAGE));
        phase = (uint32_t)(start0 & 15);
        staged = __ballot(valid && !by_lane) == 0ull && phase + stretch <= VCF_STAGE;
    }
    if (valid && (ALL_SNP || my.type == MSIM_SN)) {                      // mutator.py:334-341
        const unsigned long long start = (unsigned long long)my.pos + 1;
        const int nd = ndigits(start);
        if (!WRITE) {
            const uint8_t x = in[my.pos];
            const uint8_t ref = lut[768 + x], alt = lut[(uint32_t)my.aux * 256 + x];      // ti / tv column of conv(x)
            len_io[mine] = ref == alt ? 0u : name_len + (uint32_t)nd + 19u;                 // vcf_writer.py:123: REF == ALT suppressed
            ra[mine] = (uint16_t)((uint32_t)ref | ((uint32_t)alt << 8));
        } else if (my_len) {
            const uint32_t r2 = ra[mine];
            if (staged) snp_line(&stage[wave][phase + (uint32_t)(my_off - start0)], name, name_len, start, nd, (uint8_t)r2, (uint8_t)(r2 >> 8));
            else snp_line(text + my_off, name, name_len, start, nd, (uint8_t)r2, (uint8_t)(r2 >> 8));
        }
    } else if (!ALL_SNP && valid && by_lane) {
        LSink<WRITE> s;
        s.n = 0;
        s.p = !WRITE ? nullptr : staged ? &stage[wave][phase + (uint32_t)(my_off - start0)] : text + my_off;
        format_record(s, my, pool, in, L, name, name_len, lut);
        if (!WRITE) len_io[mine] = (uint32_t)s.n;
    }
    if (WRITE && staged) {                                               // (wave-uniform)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (flat stores into LDS complete out of order with ds reads: wait for both)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        char *g0 = text + (start0 - phase);                              // 16-byte aligned (the text buffer is)
        const uint32_t lim = phase + stretch;
        for (uint32_t k = lane_id * 16; k < lim; k += 64 * 16) {
            if (k >= phase && k + 16 <= lim) {
                *reinterpret_cast<uint4 *>(g0 + k) = *reinterpret_cast<const uint4 *>(&stage[wave][k]);
            } else {
                const uint32_t lo = max(k, phase), hi = min(k + 16, lim);
                for (uint32_t q = lo; q < hi; q++) g0[q] = stage[wave][q];
            }
        }
        return;
    }
    if (ALL_SNP) return;
    unsigned long long todo = __ballot(valid && !by_lane);
    while (todo) {
        const int src_lane = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t i = base_rec + (uint32_t)src_lane;
        msim_record r;
        r.pos = (uint32_t)__shfl((int)my.pos, src_lane, 64);
        r.stop = (uint32_t)__shfl((int)my.stop, src_lane, 64);
        r.extra = (uint32_t)__shfl((int)my.extra, src_lane, 64);
        const uint32_t ta = (uint32_t)__shfl((int)((uint32_t)my.type | ((uint32_t)my.aux << 8)), src_lane, 64);
        r.type = (uint8_t)ta; r.aux = (uint8_t)(ta >> 8); r.rsv = 0;
        WSink<WRITE> s;
        s.lane = lane_id;
        s.n = 0;
        s.p = WRITE ? text + __shfl(my_off, src_lane, 64) : nullptr;
        format_record(s, r, pool, in, L, name, name_len, lut);
        if (!WRITE && s.lane == 0) len_io[i] = (uint32_t)s.n;
    }
}

// ---- exclusive u64 scan of u32 lengths (2048 per workgroup)
constexpr int LS_ITEMS = 8;
constexpr int LS_BLOCK = TX_THREADS * LS_ITEMS;

__global__ __launch_bounds__(TX_THREADS) void k_len_reduce(const uint32_t *__restrict__ len, uint32_t n,
                                                           unsigned long long *__restrict__ sums) {
    __shared__ unsigned long long red[TX_WAVES];
    const uint32_t i0 = blockIdx.x * LS_BLOCK + threadIdx.x * LS_ITEMS;
    unsigned long long s = 0;
#pragma unroll
    for (int q = 0; q < LS_ITEMS; q++) if (i0 + q < n) s += len[i0 + q];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(1024) void k_scan_u64(unsigned long long *__restrict__ a, uint32_t n,
                                                   unsigned long long *__restrict__ mailbox) {
    __shared__ unsigned long long buf[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < n ? a[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const unsigned long long t = threadIdx.x >= (unsigned)o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        const unsigned long long incl = buf[threadIdx.x], c = carry;
        if (i < n) a[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) { a[n] = carry; *mailbox = carry; __threadfence_system(); }
}

__global__ __launch_bounds__(TX_THREADS) void k_len_offsets(const uint32_t *__restrict__ len, uint32_t n,
                                                            const unsigned long long *__restrict__ sums,
                                                            unsigned long long *__restrict__ off) {
    __shared__ unsigned long long wsum[TX_WAVES];
    const uint32_t i0 = blockIdx.x * LS_BLOCK + threadIdx.x * LS_ITEMS;
    uint32_t l[LS_ITEMS];
    unsigned long long s = 0;
#pragma unroll
    for (int q = 0; q < LS_ITEMS; q++) { l[q] = i0 + q < n ? len[i0 + q] : 0; s += l[q]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned long long run = sums[blockIdx.x] + incl - s;
    for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int q = 0; q < LS_ITEMS; q++) {
        if (i0 + q < n) off[i0 + q] = run;
        run += l[q];
    }
}

// ---- FASTA egress: '\n' after every bpl bases (no newline after a partial last line)
__global__ __launch_bounds__(TX_THREADS) void k_frame(const uint8_t *__restrict__ seq, unsigned long long L,
                                                      uint32_t bpl, unsigned long long text_len,
                                                      uint8_t *__restrict__ text) {
    const unsigned long long t0 = ((unsigned long long)blockIdx.x * TX_THREADS + threadIdx.x) * 16;
    if (t0 >= text_len) return;
    const uint32_t stride = bpl + 1;
    unsigned long long line = t0 / stride;
    uint32_t col = (uint32_t)(t0 % stride);
    if (bpl >= 16 && t0 + 16 <= text_len) {
        // Lines of 16 bases or more: at most ONE newline falls into these 16 bytes, at k = bpl - col.  Bytes before it are source
        // bytes s0 .. (16 at the output's source offset), bytes behind it the same stream one byte later in the output: two
        // 16-byte loads and a mask (round 1's byte-by-byte walk ran at 2.0 TB/s: 16 byte loads per thread).
        const unsigned long long s0 = t0 - line;           // source index of output byte t0 (of the byte after a leading '\n')
        const uint32_t k = bpl - col;                      // col <= bpl
        unsigned __int128 a, out;
        __builtin_memcpy(&a, seq + s0, 16);
        if (k >= 16) out = a;
        else {
            unsigned __int128 cshift = 0;
            if (s0) __builtin_memcpy(&cshift, seq + s0 - 1, 16);      // cshift[q] = seq[s0 + q - 1]
            const unsigned __int128 ones = ~(unsigned __int128)0;
            const unsigned __int128 low = k ? ones >> (128 - 8 * k) : 0;              // bytes [0, k)
            const unsigned __int128 upto = k == 15 ? ones : ones >> (128 - 8 * (k + 1));   // bytes [0, k]
            out = (a & low) | ((unsigned __int128)'\n' << (8 * k)) | (cshift & ~upto);
        }
        __builtin_memcpy(text + t0, &out, 16);
        return;
    }
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 16; q++) {
        uint8_t c = 0;
        if (t0 + q < text_len) c = col == bpl ? (uint8_t)'\n' : seq[line * bpl + col];
        w[q >> 2] |= (uint32_t)c << ((q & 3) * 8);
        if (++col == stride) { col = 0; line++; }
    }
    if (t0 + 16 <= text_len) {
        uint4 v; v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        *reinterpret_cast<uint4 *>(text + t0) = v;
    } else {
        for (int q = 0; t0 + q < text_len; q++) text[t0 + q] = (uint8_t)(w[q >> 2] >> ((q & 3) * 8));
    }
}

// ---- interchromosomal translocation: output byte o belongs to segment j = the last one with seg_out[j] <= o; even segments
// are cut from contig a, odd ones from contig b, seg_src[j] is where the segment starts in its contig.  16 output bytes per
// thread: one binary search, then a forward walk (segments are at least two bases long -- breakpoints keep a base between them)
__global__ __launch_bounds__(TX_THREADS) void k_splice(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                       const uint32_t *__restrict__ seg_out, const uint32_t *__restrict__ seg_src,
                                                       uint32_t n_seg, unsigned long long out_len, uint8_t *__restrict__ dst) {
    const unsigned long long o0 = ((unsigned long long)blockIdx.x * TX_THREADS + threadIdx.x) * 16;
    if (o0 >= out_len) return;
    uint32_t lo = 0, hi = n_seg;                           // seg_out[lo] <= o0 < seg_out[hi]   (seg_out[n_seg] = out_len)
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if ((unsigned long long)seg_out[mid] <= o0) lo = mid; else hi = mid;
    }
    uint32_t j = lo;
    unsigned long long next = seg_out[j + 1];
    const uint8_t *src = ((j & 1u) ? b : a) + seg_src[j] - seg_out[j];      // src[o] is output byte o while o is in segment j
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const unsigned long long o = o0 + q;
        uint8_t c = 0;
        if (o < out_len) {
            while (o >= next) {                            // (a while: an empty segment can not occur, but costs nothing to allow)
                j++;
                next = seg_out[j + 1];
                src = ((j & 1u) ? b : a) + seg_src[j] - seg_out[j];
            }
            c = src[o];
        }
        w[q >> 2] |= (uint32_t)c << ((q & 3) * 8);
    }
    if (o0 + 16 <= out_len) {
        uint4 v; v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        *reinterpret_cast<uint4 *>(dst + o0) = v;
    } else {
        for (int q = 0; o0 + q < out_len; q++) dst[o0 + q] = (uint8_t)(w[q >> 2] >> ((q & 3) * 8));
    }
}

// ---- FASTA ingest: base i of the record sits at body[(i / lenc) * lenb + i % lenc]; a-z -> A-Z
__global__ __launch_bounds__(TX_THREADS) void k_gather(const uint8_t *__restrict__ body, unsigned long long n_bases,
                                                       uint32_t lenc, uint32_t lenb, uint8_t *__restrict__ dst) {
    const unsigned long long i0 = ((unsigned long long)blockIdx.x * TX_THREADS + threadIdx.x) * 16;
    if (i0 >= n_bases) return;
    unsigned long long line = i0 / lenc;
    uint32_t col = (uint32_t)(i0 % lenc);
    if (lenc >= 16 && i0 + 16 <= n_bases) {
        // Lines of 16 bases or more: these 16 bases cross at most one line terminator (lenb - lenc bytes), after kk = lenc - col
        // of them: the bases before it sit at the text offset, the ones behind it lenb - lenc bytes further -- two 16-byte
        // loads and a mask, upper-casing on all 16 bytes at once.
        const unsigned long long b0 = line * lenb + col;
        const uint32_t kk = lenc - col;
        unsigned __int128 a, x;
        __builtin_memcpy(&a, body + b0, 16);
        if (kk >= 16) x = a;
        else {
            unsigned __int128 b;
            __builtin_memcpy(&b, body + b0 + (lenb - lenc), 16);      // (the body's buffer carries 64 bytes of slack behind the text)
            const unsigned __int128 low = (~(unsigned __int128)0) >> (128 - 8 * kk);   // kk >= 1
            x = (a & low) | (b & ~low);
        }
        // a-z -> A-Z per byte: bit 7 of (c + 0x1f) says c >= 'a', of (c + 0x05) c > 'z' (7-bit part: no carry into a neighbour)
        unsigned long long h[2] = {(unsigned long long)x, (unsigned long long)(x >> 64)};
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const unsigned long long v = h[q], lo7 = v & 0x7f7f7f7f7f7f7f7full;
            const unsigned long long is_lower = (lo7 + 0x1f1f1f1f1f1f1f1full) & ~(lo7 + 0x0505050505050505ull) & ~v & 0x8080808080808080ull;
            h[q] = v - (is_lower >> 2);
        }
        uint4 o;
        o.x = (uint32_t)h[0]; o.y = (uint32_t)(h[0] >> 32); o.z = (uint32_t)h[1]; o.w = (uint32_t)(h[1] >> 32);
        *reinterpret_cast<uint4 *>(dst + i0) = o;
        return;
    }
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 16; q++) {
        uint8_t c = 0;
        if (i0 + q < n_bases) {
            c = body[line * lenb + col];
            if (c >= 'a' && c <= 'z') c -= 32;
        }
        w[q >> 2] |= (uint32_t)c << ((q & 3) * 8);
        if (++col == lenc) { col = 0; line++; }
    }
    if (i0 + 16 <= n_bases) {
        uint4 v; v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        *reinterpret_cast<uint4 *>(dst + i0) = v;
    } else {
        for (int q = 0; i0 + q < n_bases; q++) dst[i0 + q] = (uint8_t)(w[q >> 2] >> ((q & 3) * 8));
    }
}

}  // namespace

// VCF text of one contig into the context's text buffer; returns its size.
int vcf_render_device(Ctx *c, Contig &g, const char *seq_name, uint64_t *bytes, uint8_t **buf, size_t *cap) {
    const uint32_t n = (uint32_t)g.n_rec;
    *bytes = 0;
    const bool own = buf == nullptr;
    if (own) { buf = &c->d_text; cap = &c->cap_text; }
    if (!n) { if (own) c->text_len = 0; return MSIM_OK; }
    hipStream_t st = c->stream;
    const size_t name_len = strlen(seq_name);
    const uint32_t nb = (n + LS_BLOCK - 1) / LS_BLOCK;
    // scratch: name | lens u32[n] | sums u64[nb+1] | off u64[n] | REF/ALT of the SNP records u16[n]
    const size_t o_len = (name_len + 255) & ~(size_t)255;
    const size_t o_sums = o_len + (((size_t)n * 4 + 255) & ~(size_t)255);
    const size_t o_off = o_sums + (((size_t)(nb + 1) * 8 + 255) & ~(size_t)255);
    const size_t o_ra = o_off + (((size_t)n * 8 + 255) & ~(size_t)255);
    int rc = dev_reserve(c, (void **)&c->d_text_scratch, &c->cap_text_scratch, o_ra + (size_t)n * 2);
    if (rc) return rc;
    uint8_t *base = c->d_text_scratch;
    uint8_t *d_name = base;
    uint32_t *d_len = reinterpret_cast<uint32_t *>(base + o_len);
    unsigned long long *d_sums = reinterpret_cast<unsigned long long *>(base + o_sums);
    unsigned long long *d_off = reinterpret_cast<unsigned long long *>(base + o_off);
    uint16_t *d_ra = reinterpret_cast<uint16_t *>(base + o_ra);
    MSIM_HIP(c, hipMemcpyAsync(d_name, seq_name, name_len, hipMemcpyHostToDevice, st));
    const uint8_t *in = g.d_in + PAD;
    const uint8_t *pool = g.d_pool ? g.d_pool + PAD : nullptr;
    const dim3 grid((n + 64 * TX_WAVES - 1) / (64 * TX_WAVES));           // a wave takes 64 consecutive records
#define MSIM_VCF_LINES(W, ...) do { if (g.all_snp) hipLaunchKernelGGL((k_vcf_lines<W, true>), __VA_ARGS__); \
                                    else hipLaunchKernelGGL((k_vcf_lines<W, false>), __VA_ARGS__); } while (0)
    MSIM_VCF_LINES(false, grid, dim3(TX_THREADS), 0, st, g.d_recs, n, pool, in, (unsigned long long)g.len,
                   d_name, (uint32_t)name_len, ctx_lut(c), d_len, d_ra, (const unsigned long long *)nullptr, (char *)nullptr);
    hipLaunchKernelGGL(k_len_reduce, dim3(nb), dim3(TX_THREADS), 0, st, d_len, n, d_sums);
    hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, st, d_sums, nb, c->h_mail);
    hipLaunchKernelGGL(k_len_offsets, dim3(nb), dim3(TX_THREADS), 0, st, d_len, n, d_sums, d_off);
    MSIM_HIP(c, hipGetLastError());
    MSIM_HIP(c, hipStreamSynchronize(st));
    const uint64_t total = *c->h_mail;
    rc = dev_reserve(c, (void **)buf, cap, total + 64);
    if (rc) return rc;
    if (total) {
        MSIM_VCF_LINES(true, grid, dim3(TX_THREADS), 0, st, g.d_recs, n, pool, in, (unsigned long long)g.len,
                       d_name, (uint32_t)name_len, ctx_lut(c), d_len, d_ra, d_off, reinterpret_cast<char *>(*buf));
#undef MSIM_VCF_LINES
        MSIM_HIP(c, hipGetLastError());
    }
    if (own) c->text_len = total;
    *bytes = total;
    return MSIM_OK;
}

// Mutated stream of one contig as FASTA body text into the context's text buffer.
int fasta_frame_device(Ctx *c, Contig &g, uint32_t bpl, uint64_t *bytes, uint8_t **buf, size_t *cap) {
    const uint64_t L = g.out_len;
    const uint64_t total = L + L / bpl;
    *bytes = total;
    if (!buf) { buf = &c->d_text; cap = &c->cap_text; c->text_len = total; }
    if (!total) return MSIM_OK;
    int rc = dev_reserve(c, (void **)buf, cap, total + 64);
    if (rc) return rc;
    const uint64_t groups = (total + 15) / 16;
    hipLaunchKernelGGL(k_frame, dim3((uint32_t)((groups + TX_THREADS - 1) / TX_THREADS)), dim3(TX_THREADS), 0, c->stream,
                       g.d_out, (unsigned long long)L, bpl, (unsigned long long)total, *buf);
    MSIM_HIP(c, hipGetLastError());
    return MSIM_OK;
}

// FASTA body text (host memory) -> upper-cased bases of a new contig's input buffer.
int fasta_gather_device(Ctx *c, const uint8_t *body, uint64_t body_bytes, uint64_t n_bases, uint32_t lenc,
                        uint32_t lenb, uint8_t *d_dst) {
    if (!n_bases) return MSIM_OK;
    int rc = dev_reserve(c, (void **)&c->d_text, &c->cap_text, body_bytes + 64);
    if (rc) return rc;
    MSIM_HIP(c, hipMemcpyAsync(c->d_text, body, body_bytes, hipMemcpyHostToDevice, c->stream));
    const uint64_t groups = (n_bases + 15) / 16;
    hipLaunchKernelGGL(k_gather, dim3((uint32_t)((groups + TX_THREADS - 1) / TX_THREADS)), dim3(TX_THREADS), 0, c->stream,
                       c->d_text, (unsigned long long)n_bases, lenc, lenb, d_dst);
    MSIM_HIP(c, hipGetLastError());
    MSIM_HIP(c, hipStreamSynchronize(c->stream));
    c->text_len = 0;
    return MSIM_OK;
}

// Segments of the inputs of two contigs, taken alternately, as the mutated stream of `dst` (seg tables: see k_splice; host
// memory, n_seg + 1 and n_seg entries).  b may be nullptr when no odd segment exists.
int splice_device(Ctx *c, const Contig &a, const Contig *b, const uint32_t *seg_out, const uint32_t *seg_src, uint32_t n_seg,
                  Contig &dst) {
    const uint64_t out_len = seg_out[n_seg];
    int rc = dev_reserve(c, (void **)&dst.d_out, &dst.cap_out, out_len + PAD);
    if (rc) return rc;
    dst.out_len = out_len;
    if (!out_len) return MSIM_OK;
    uint32_t *d_tab = nullptr;
    const size_t tab_bytes = ((size_t)2 * n_seg + 1) * sizeof(uint32_t);
    MSIM_HIP(c, hipMalloc(&d_tab, tab_bytes));
    hipError_t e = hipMemcpyAsync(d_tab, seg_out, ((size_t)n_seg + 1) * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tab + n_seg + 1, seg_src, (size_t)n_seg * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const uint64_t groups = (out_len + 15) / 16;
        hipLaunchKernelGGL(k_splice, dim3((uint32_t)((groups + TX_THREADS - 1) / TX_THREADS)), dim3(TX_THREADS), 0, c->stream,
                           a.d_in + PAD, b ? b->d_in + PAD : a.d_in + PAD, d_tab, d_tab + n_seg + 1, n_seg,
                           (unsigned long long)out_len, dst.d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_tab);
    if (e != hipSuccess) return hip_fail(c, e, "splice");
    return MSIM_OK;
}

}  // namespace msim
