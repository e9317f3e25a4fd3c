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

// Two sinks with one interface, both used by ONE lane for its own record.  CSink counts (the length pass); PSink writes the
// line's short fields -- name, numbers, separators, INFO -- itself and turns every REF / ALT copy longer than VCF_INLINE bytes
// into a PIECE (destination offset, source, length, mode) in the wave's LDS list: the wave copies all pieces of its 64 records
// together afterwards, 16 destination-aligned bytes per lane and step (vcf_copy_pieces).
constexpr uint32_t VCF_INLINE = 25;           // copies up to this length are made by the formatting lane (a DE of span 24 reads 25)
constexpr uint32_t VCF_LANE_SPAN = 24;        // records spanning at most this many bases have no piece ("short": candidates for staging)
constexpr uint32_t VCF_STAGE = 4096;          // LDS per wave for the lines of its 64 records where all of them are short
constexpr int VCF_SLOTS = 4;                  // pieces a record can have: a long name, REF, ALT (twice for a duplication)

struct VcfPiece { unsigned long long dst; const uint8_t *src; uint32_t len, mode; };      // mode: see bulk()

struct CSink {
    unsigned long long n;
    __device__ __forceinline__ void put(char) { n++; }
    template <int N>
    __device__ __forceinline__ void lit(const char (&)[N]) { n += N - 1; }
    __device__ __forceinline__ void num(unsigned long long v) { n += (unsigned)ndigits(v); }
    __device__ __forceinline__ void bulk(const uint8_t *, unsigned long long len, int, const uint8_t *) { n += len; }
};

struct PSink {
    char *text;                           // the contig's VCF text
    unsigned long long n;                 // absolute offset of the next byte
    VcfPiece *slots;                      // this lane's VCF_SLOTS entries of the wave's piece list (len == 0: unused)
    uint32_t np;
    __device__ __forceinline__ void put(char c) { text[n++] = c; }
    template <int N>
    __device__ __forceinline__ void lit(const char (&s)[N]) {            // N - 1 characters, four per store where there are four
        constexpr int M = N - 1;
        char *q = text + n;
#pragma unroll
        for (int i = 0; i + 4 <= M; i += 4) {
            const uint32_t w = (uint32_t)(uint8_t)s[i] | ((uint32_t)(uint8_t)s[i + 1] << 8) | ((uint32_t)(uint8_t)s[i + 2] << 16) |
                               ((uint32_t)(uint8_t)s[i + 3] << 24);
            __builtin_memcpy(q + i, &w, 4);                              // (unaligned dword store)
        }
#pragma unroll
        for (int i = M & ~3; i < M; i++) q[i] = s[i];
        n += M;
    }
    __device__ __forceinline__ void num(unsigned long long v) {
        const int k = ndigits(v);
        for (int q = k - 1; q >= 0; q--) { text[n + q] = (char)('0' + (int)(v % 10)); v /= 10; }
        n += (unsigned)k;
    }
    // mode 0 raw, 1 conv(x), 2 comp(conv(x)) read BACKWARDS from src (src points at the LAST source byte)
    __device__ __forceinline__ void bulk(const uint8_t *__restrict__ src, unsigned long long len, int mode, const uint8_t *lut) {
        if (len <= VCF_INLINE || np >= (uint32_t)VCF_SLOTS || len >= (1ull << 32)) {
            for (unsigned long long i = 0; i < len; i++) {
                uint8_t c;
                if (mode == 0) c = src[i];
                else if (mode == 1) c = lut[768 + src[i]];
                else c = lut[1024 + *(src - i)];
                text[n + i] = (char)c;
            }
        } else {
            VcfPiece pc;
            pc.dst = n; pc.src = src; pc.len = (uint32_t)len; pc.mode = (uint32_t)mode;
            slots[np++] = pc;
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
// CHECK: evaluate the suppression rules (REF == ALT, vcf_writer.py:123) -- the length pass does, the write pass skips the
// records whose length came out 0.
template <bool CHECK, class S>
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
            if (CHECK) {                                                 // REF == ALT (a reverse-complement palindrome): suppressed
                bool diff = false;                                       // (three of four random pairs differ: ~1.3 steps)
                for (unsigned long long q = 0; q < len && !diff; q++)
                    diff = lut[768 + in[pos + q]] != lut[1024 + in[stop - q]];
                if (!diff) break;
            }
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

// 16 source bytes through a 256-entry LDS table
__device__ __forceinline__ uint4 lut16(uint4 w, const uint8_t *t) {
    uint32_t v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int d = 0; d < 4; d++)
        v[d] = (uint32_t)t[v[d] & 255] | ((uint32_t)t[(v[d] >> 8) & 255] << 8) | ((uint32_t)t[(v[d] >> 16) & 255] << 16) |
               ((uint32_t)t[v[d] >> 24] << 24);
    return uint4{v[0], v[1], v[2], v[3]};
}
// ... and the same with the byte order reversed (reverse complement: the piece is read backwards)
__device__ __forceinline__ uint4 lut16_rev(uint4 w, const uint8_t *t) {
    uint32_t v[4] = {w.w, w.z, w.y, w.x};
#pragma unroll
    for (int d = 0; d < 4; d++)
        v[d] = (uint32_t)t[v[d] >> 24] | ((uint32_t)t[(v[d] >> 16) & 255] << 8) | ((uint32_t)t[(v[d] >> 8) & 255] << 16) |
               ((uint32_t)t[v[d] & 255] << 24);
    return uint4{v[0], v[1], v[2], v[3]};
}

// The pieces of a wave's records, copied by the whole wave: a UNIT is 16 destination-aligned bytes of one piece (its first and
// last unit may be partial).  pre[s] = units in front of slot s (pre[64 * VCF_SLOTS] = all of them); a lane takes units lane,
// lane + 64, ... and finds each one's slot by binary search.  Two units per lane are in flight (their loads are issued together).
// Round 4 took the long records one after the other, every copy a dependent load -> table -> store chain of the whole wave:
// latency-bound at 0.2 TB/s of text on the SV mix (DU / IV of 50-500 bases: 85 % of the bytes).
struct VcfUnit { unsigned long long dst; const uint8_t *src; uint32_t lo, hi, mode; uint4 w; };      // bytes [lo, hi) of the unit at dst

__device__ __forceinline__ void vcf_unit_load(VcfUnit &u, uint32_t idx, uint32_t n_units, const VcfPiece *pcs, const uint32_t *pre) {
    u.hi = 0;
    if (idx >= n_units) return;
    uint32_t s = 0;
#pragma unroll
    for (int step = 32 * VCF_SLOTS; step; step >>= 1) if (pre[s + step] <= idx) s += step;
    const VcfPiece pc = pcs[s];
    const unsigned long long a0 = pc.dst & ~15ull, at = a0 + 16ull * (idx - pre[s]);
    const unsigned long long lo = at > pc.dst ? at : pc.dst, hi = at + 16 < pc.dst + pc.len ? at + 16 : pc.dst + pc.len;
    u.dst = at; u.lo = (uint32_t)(lo - at); u.hi = (uint32_t)(hi - at); u.mode = pc.mode;
    const unsigned long long k = lo - pc.dst;                            // piece offset of the unit's first byte
    u.src = pc.mode == 2 ? pc.src - k : pc.src + k;                      // source of that byte
    if (u.hi - u.lo == 16) {                                             // a whole unit: one unaligned 16-byte load
        if (pc.mode == 2) __builtin_memcpy(&u.w, u.src - 15, 16);
        else __builtin_memcpy(&u.w, u.src, 16);
    }
}

__device__ __forceinline__ void vcf_unit_store(const VcfUnit &u, char *__restrict__ text, const uint8_t *lut) {
    if (u.hi == 0) return;
    if (u.hi - u.lo == 16) {
        const uint4 o = u.mode == 0 ? u.w : u.mode == 1 ? lut16(u.w, lut + 768) : lut16_rev(u.w, lut + 1024);
        *reinterpret_cast<uint4 *>(text + u.dst) = o;
    } else {
        for (uint32_t q = u.lo; q < u.hi; q++) {
            const uint32_t i = q - u.lo;
            const uint8_t c = u.mode == 0 ? u.src[i] : u.mode == 1 ? lut[768 + u.src[i]] : lut[1024 + *(u.src - i)];
            text[u.dst + q] = (char)c;
        }
    }
}

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
// A wave takes 64 consecutive records, a LANE formats its own record's line (round 1 gave every line a wave: ~15 instructions of
// 64 lanes for 25 bytes, 135 GB/s).  Length pass: nothing but the record, one base per SNP and the ends of an inversion are read --
// every length follows from (type, pos, stop).  Write pass: the lines of a wave's records are adjacent in the text.  Where all of
// them are short, the lanes format into LDS and the wave stores the stretch 16 aligned bytes per lane; else a lane writes its
// line's short fields to the text itself and the long REF / ALT copies of all 64 records go through vcf_unit_*.
constexpr uint32_t VCF_WAVE_LDS = 64 * VCF_SLOTS * sizeof(VcfPiece) + (64 * VCF_SLOTS + 4) * sizeof(uint32_t);   // piece list + unit prefix

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
    constexpr uint32_t WB = !WRITE ? 16 : ALL_SNP ? VCF_STAGE : (VCF_WAVE_LDS > VCF_STAGE ? VCF_WAVE_LDS : VCF_STAGE);
    __shared__ __attribute__((aligned(16))) char wbuf[WRITE ? TX_WAVES : 1][WB];
    for (int i = threadIdx.x; i < 1280 / 4; i += TX_THREADS)
        reinterpret_cast<uint32_t *>(lut)[i] = reinterpret_cast<const uint32_t *>(lut_g)[i];
    __syncthreads();
    const uint32_t lane_id = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base_rec = (blockIdx.x * TX_WAVES + wave) * 64;
    if (base_rec >= n_rec) return;
    const uint32_t mine = base_rec + lane_id;
    const bool valid = mine < n_rec;
    msim_record my{};
    if (valid) my = recs[mine];
    const bool is_snp = ALL_SNP || my.type == MSIM_SN;
    if (!WRITE) {                                                        // ---- length pass
        if (!valid) return;
        if (is_snp) {                                                    // mutator.py:334-341
            const unsigned long long start = (unsigned long long)my.pos + 1;
            const uint8_t x = in[my.pos];
            const uint8_t ref = lut[768 + x], alt = lut[(uint32_t)my.aux * 256 + x];      // ti / tv column of conv(x)
            len_io[mine] = ref == alt ? 0u : name_len + (uint32_t)ndigits(start) + 19u;   // vcf_writer.py:123: REF == ALT suppressed
            ra[mine] = (uint16_t)((uint32_t)ref | ((uint32_t)alt << 8));
        } else if (!ALL_SNP) {
            CSink s;
            s.n = 0;
            format_record<true>(s, my, pool, in, L, name, name_len, lut);
            len_io[mine] = (uint32_t)s.n;
        }
        return;
    }
    // ---- write pass
    char *stage = wbuf[WRITE ? wave : 0];
    bool is_short = is_snp;
    if (!ALL_SNP && valid && !is_snp) {
        const unsigned long long hi = (unsigned long long)my.stop + 1 < L ? (unsigned long long)my.stop + 1 : L;
        const unsigned long long span = my.type == MSIM_TLI ? (hi > my.extra ? hi - my.extra : 0)
                                                            : (unsigned long long)my.stop - my.pos + 1;
        is_short = span <= VCF_LANE_SPAN;
    }
    unsigned long long my_off = 0;
    uint32_t my_len = 0;
    if (valid) { my_off = off[mine]; my_len = len_io[mine]; }
    const uint32_t last = min(63u, n_rec - 1 - base_rec);
    const unsigned long long start0 = __shfl(my_off, 0, 64);
    const unsigned long long end = __shfl(my_off + my_len, (int)last, 64);
    const uint32_t stretch = (uint32_t)min(end - start0, (unsigned long long)(2 * VCF_STAGE));
    const uint32_t phase = (uint32_t)(start0 & 15);
    const bool staged = __ballot(valid && !is_short) == 0ull && phase + stretch <= VCF_STAGE && (ALL_SNP || name_len <= VCF_INLINE);
    if (staged) {                                                        // (wave-uniform) every line is short: through LDS
        if (valid && my_len) {
            char *p = &stage[phase + (uint32_t)(my_off - start0)];
            if (is_snp) {
                const uint32_t r2 = ra[mine];
                const unsigned long long start = (unsigned long long)my.pos + 1;
                snp_line(p, name, name_len, start, ndigits(start), (uint8_t)r2, (uint8_t)(r2 >> 8));
            } else if (!ALL_SNP) {
                PSink s;
                s.text = p; s.n = 0; s.slots = nullptr; s.np = VCF_SLOTS;   // (no piece can arise: every copy is <= VCF_INLINE)
                format_record<false>(s, my, pool, in, L, name, name_len, lut);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (flat stores into LDS complete out of order with ds reads: wait for both)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        char *g0 = text + (start0 - phase);                              // 16-byte aligned (the text buffer is)
        const uint32_t lim = phase + stretch;
        for (uint32_t k = lane_id * 16; k < lim; k += 64 * 16) {
            if (k >= phase && k + 16 <= lim) {
                *reinterpret_cast<uint4 *>(g0 + k) = *reinterpret_cast<const uint4 *>(&stage[k]);
            } else {
                const uint32_t lo = max(k, phase), hi = min(k + 16, lim);
                for (uint32_t q = lo; q < hi; q++) g0[q] = stage[q];
            }
        }
        return;
    }
    if (ALL_SNP) {                                                       // (a stretch that does not fit the stage: long names)
        if (valid && my_len) {
            const uint32_t r2 = ra[mine];
            const unsigned long long start = (unsigned long long)my.pos + 1;
            snp_line(text + my_off, name, name_len, start, ndigits(start), (uint8_t)r2, (uint8_t)(r2 >> 8));
        }
        return;
    }
    VcfPiece *pcs = reinterpret_cast<VcfPiece *>(stage);
    uint32_t *pre = reinterpret_cast<uint32_t *>(stage + 64 * VCF_SLOTS * sizeof(VcfPiece));
    PSink s;
    s.text = text; s.n = my_off; s.slots = pcs + lane_id * VCF_SLOTS; s.np = 0;
    if (valid && my_len) {
        if (is_snp) {
            const uint32_t r2 = ra[mine];
            const unsigned long long start = (unsigned long long)my.pos + 1;
            snp_line(text + my_off, name, name_len, start, ndigits(start), (uint8_t)r2, (uint8_t)(r2 >> 8));
        } else {
            format_record<false>(s, my, pool, in, L, name, name_len, lut);
        }
    }
    // units of this lane's pieces, their prefix over the wave
    uint32_t cnt[VCF_SLOTS], tot = 0;
#pragma unroll
    for (int q = 0; q < VCF_SLOTS; q++) {
        cnt[q] = 0;
        if ((uint32_t)q < s.np) {
            const VcfPiece pc = s.slots[q];
            cnt[q] = (uint32_t)((((pc.dst + pc.len + 15) & ~15ull) - (pc.dst & ~15ull)) >> 4);
        }
        tot += cnt[q];
    }
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane_id >= (uint32_t)o) incl += t;
    }
    const uint32_t n_units = (uint32_t)__shfl((int)incl, 63, 64);
    if (n_units == 0) return;
    uint32_t run = incl - tot;
#pragma unroll
    for (int q = 0; q < VCF_SLOTS; q++) { pre[lane_id * VCF_SLOTS + q] = run; run += cnt[q]; }
    if (lane_id == 63) pre[64 * VCF_SLOTS] = n_units;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (uint32_t u0 = 0; u0 < n_units; u0 += 128) {
        VcfUnit a, b;
        vcf_unit_load(a, u0 + lane_id, n_units, pcs, pre);
        vcf_unit_load(b, u0 + 64 + lane_id, n_units, pcs, pre);
        vcf_unit_store(a, text, lut);
        vcf_unit_store(b, text, lut);
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
