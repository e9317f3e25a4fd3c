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
#include <cstdlib>
#include <cstring>

#include "ctx.h"

namespace msim {

uint8_t *ctx_lut(Ctx *c);             // msim_api.hip: 1280-byte translation table (layout in apply.hip)

namespace {

constexpr int TX_THREADS = 256;
constexpr int TX_WAVES = TX_THREADS / 64;

__device__ __forceinline__ int ndigits(unsigned long long v) {
    if ((v >> 32) == 0) {                                                // (every position of a contig below 4 GiB)
        const uint32_t x = (uint32_t)v;
        return 1 + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) + (x >= 10000000u) +
               (x >= 100000000u) + (x >= 1000000000u);
    }
    int k = 10;
    unsigned long long p = 10000000000ull;
    while (v >= p && k < 20) { p *= 10; k++; }
    return k;
}

// Two sinks with one interface, each used by ONE lane for its own record.
//   CSink  counts (the length pass).
//   (vcf_line_gaps measures a line's GAPS: a REF / ALT copy longer than VCF_INLINE bytes is not formatted by its lane -- its
//          16-byte-aligned interior (in text coordinates) becomes a PIECE that the whole wave copies -- and is missing from
//          the wave's LDS stage, which holds everything else of the wave's lines back to back ("compact" coordinates: text
//          offset minus the gaps in front of it; gaps are multiples of 16, so alignment survives).)
//   SSink  writes: short fields, inline copies and the edges of the pieces (the < 16 bytes in front of and behind a piece's
//          aligned interior) into the stage; pieces into the wave's LDS list; gs[u] += units of the gap that opens in front of
//          stage unit u.
constexpr uint32_t VCF_INLINE = 25;           // copies up to this length are made by the formatting lane
constexpr uint32_t VCF_STAGE = 4096;          // LDS per wave: the compact text of (a group of) its 64 records
constexpr uint32_t VCF_UNITS = VCF_STAGE / 16;
constexpr int VCF_SLOTS = 3;                  // pieces a record can have: REF, ALT (twice for a duplication)

// One REF / ALT copy of a line: its first byte's text offset (relative to the stage's text origin, which is 16-byte aligned), its
// length, and source | stage position of its first byte << 48 | mode << 62.
struct VcfPiece { uint32_t n_rel, len; unsigned long long src_ws_mode; };

// The 16-byte-aligned interior [ia, ib) of a copy of `len` bytes at text offset n -- the part that goes source -> text without
// touching the stage -- if it is worth having (copies up to VCF_INLINE bytes stay whole in the stage).
__device__ __forceinline__ bool vcf_piece_of(unsigned long long n, unsigned long long len, unsigned long long &ia, unsigned long long &ib) {
    if (len <= VCF_INLINE) return false;
    ia = (n + 15) & ~15ull;
    ib = (n + len) & ~15ull;
    return ib > ia;
}

struct CSink {
    unsigned long long n;
    __device__ __forceinline__ void chars(uint32_t, uint32_t k) { n += k; }
    __device__ __forceinline__ void chars64(unsigned long long, uint32_t k) { n += k; }
    __device__ __forceinline__ void put(char) { n++; }
    template <int N>
    __device__ __forceinline__ void lit(const char (&)[N]) { n += N - 1; }
    __device__ __forceinline__ void num(unsigned long long v) { n += (unsigned)ndigits(v); }
    __device__ __forceinline__ void name(const uint8_t *, uint32_t len) { n += len; }
    template <class D> __device__ __forceinline__ void begin(const D &) {}
    __device__ __forceinline__ void bulk(int, const uint8_t *, unsigned long long len, int, const uint8_t *) { n += len; }
};

struct SSink {
    char *stage;                          // the wave's compact stage (LDS)
    uint32_t w;                           // next byte in it
    unsigned long long n, origin;         // absolute text offset of the next byte; text offset of the stage's first unit
    VcfPiece *slots;                      // this lane's VCF_SLOTS entries of the wave's piece list
    uint32_t *gs;
    uint32_t np;
    __device__ __forceinline__ void put(char c) { stage[w++] = c; n++; }
    __device__ __forceinline__ void chars(uint32_t v, uint32_t k) {
        for (uint32_t i = 0; i < k; i++) { stage[w + i] = (char)v; v >>= 8; }
        w += k; n += k;
    }
    __device__ __forceinline__ void chars64(unsigned long long v, uint32_t k) {
        for (uint32_t i = 0; i < k; i++) { stage[w + i] = (char)v; v >>= 8; }
        w += k; n += k;
    }
    // literals go out eight, four, two characters per LDS store (gfx950 takes unaligned ds_write_b64 / b32 / b16): a byte per
    // store kept one register per character of "\t.\t.\tSVTYPE=" ... alive -- a hundred registers for the formatter
    template <int N>
    __device__ __forceinline__ void lit(const char (&s)[N]) {
        constexpr int M = N - 1;
        char *q = stage + w;
        int i = 0;
#pragma unroll
        for (; i + 8 <= M; i += 8) {
            unsigned long long v = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) v |= (unsigned long long)(uint8_t)s[i + j] << (8 * j);
            __builtin_memcpy(q + i, &v, 8);
        }
        if (i + 4 <= M) {
            const uint32_t v = (uint32_t)(uint8_t)s[i] | ((uint32_t)(uint8_t)s[i + 1] << 8) | ((uint32_t)(uint8_t)s[i + 2] << 16) |
                               ((uint32_t)(uint8_t)s[i + 3] << 24);
            __builtin_memcpy(q + i, &v, 4);
            i += 4;
        }
        if (i + 2 <= M) {
            const uint16_t v = (uint16_t)((uint32_t)(uint8_t)s[i] | ((uint32_t)(uint8_t)s[i + 1] << 8));
            __builtin_memcpy(q + i, &v, 2);
            i += 2;
        }
        if (i < M) q[i] = s[i];
        w += M; n += M;
    }
    __device__ __forceinline__ void num(unsigned long long v) {         // (v < 5e9: positions of a contig below 4 GiB, + 1)
        const int k = ndigits(v);
        uint32_t x = v >= 4000000000ull ? (uint32_t)(v - 4000000000ull) : (uint32_t)v;    // no 64-bit division (20 registers of it)
        const int low = v >= 4000000000ull ? 9 : k;
        for (int q = k - 1; q >= k - low; q--) { stage[w + q] = (char)('0' + x % 10u); x /= 10u; }
        if (low != k) stage[w] = '4';
        w += (unsigned)k; n += (unsigned)k;
    }
    __device__ __forceinline__ void name(const uint8_t *src, uint32_t len) {          // (src: the workgroup's LDS copy)
        for (uint32_t i = 0; i < len; i++) stage[w + i] = (char)src[i];
        w += len; n += len;
    }
    template <class D> __device__ __forceinline__ void begin(const D &) {}
    // mode 0 raw, 1 conv(x), 2 comp(conv(x)) read BACKWARDS from src (src points at the LAST source byte).
    // The lane copies nothing itself: it notes the copy as a piece, leaves room in the stage for what of it does not belong to
    // the aligned interior, and opens the gap.  (A byte-by-byte loop here was a chain of dependent global loads.)
    __device__ __forceinline__ void bulk(int, const uint8_t *__restrict__ src, unsigned long long len, int mode, const uint8_t *) {
        if (len == 0) return;
        unsigned long long ia, ib;                                       // (the wave checked: len, n - origin < 2^32, src < 2^48 -- vcf_fits)
        const bool interior = vcf_piece_of(n, len, ia, ib);
        VcfPiece pc;
        pc.n_rel = (uint32_t)(n - origin);
        pc.len = (uint32_t)len;
        pc.src_ws_mode = (unsigned long long)(uintptr_t)src | ((unsigned long long)w << 48) | ((unsigned long long)mode << 62);
        slots[np++] = pc;
        if (interior) {
            atomicAdd(&gs[(w + (uint32_t)(ia - n)) >> 4], (uint32_t)((ib - ia) >> 4));   // (a unit boundary: compact keeps text's alignment)
            w += (uint32_t)(len - (ib - ia));
        } else {
            w += (uint32_t)len;
        }
        n += len;
    }
};

// ... and the plain one for waves the stage scheme cannot describe (a wave whose 64 lines span 4 GiB of text and more -- a
// duplication of 1.4 Gb has such a line --, source addresses beyond 2^48): every lane writes its own line byte by byte.
struct DSink {
    char *text;
    unsigned long long n;
    __device__ __forceinline__ void put(char c) { text[n++] = c; }
    __device__ __forceinline__ void chars(uint32_t v, uint32_t k) { for (uint32_t i = 0; i < k; i++) { text[n++] = (char)v; v >>= 8; } }
    __device__ __forceinline__ void chars64(unsigned long long v, uint32_t k) { for (uint32_t i = 0; i < k; i++) { text[n++] = (char)v; v >>= 8; } }
    template <int N>
    __device__ __forceinline__ void lit(const char (&s)[N]) {
#pragma unroll
        for (int i = 0; i < N - 1; i++) text[n + i] = s[i];
        n += N - 1;
    }
    __device__ __forceinline__ void num(unsigned long long v) {
        const int k = ndigits(v);
        for (int q = k - 1; q >= 0; q--) { text[n + q] = (char)('0' + (int)(v % 10)); v /= 10; }
        n += (unsigned)k;
    }
    __device__ __forceinline__ void name(const uint8_t *src, uint32_t len) { for (uint32_t i = 0; i < len; i++) text[n++] = (char)src[i]; }
    template <class D> __device__ __forceinline__ void begin(const D &) {}
    __device__ __forceinline__ void bulk(int, const uint8_t *__restrict__ src, unsigned long long len, int mode, const uint8_t *lut) {
#pragma unroll 1
        for (unsigned long long k = 0; k < len; k++) {
            const uint8_t x = mode == 2 ? *(src - k) : src[k];
            text[n + k] = (char)(mode == 0 ? x : mode == 1 ? lut[768 + x] : lut[1024 + x]);
        }
        n += len;
    }
};

// One record's line (mutator.py:334-421 builds the record, vcf_writer.py:44-52,118-126 the line) as a DESCRIPTION that every
// type shares, so that the 64 lanes of a wave -- records of six types -- run ONE formatter instead of six one after the other:
//   name \t POS \t . \t  pre  copy0  mid  copy1  copy2  \t.\t.\t INFO \tGT\t1\n
// pre / mid: up to three / two literal characters (the anchor base of an insertion, the tab between REF and ALT, a deletion's ALT);
// copy: (source, length, mode) -- mode 0 raw, 1 conv(x), 2 comp(conv(x)) read BACKWARDS from the source (the LAST source byte).
//   SN   pre "R\tA"                                   INFO .
//   IN   pre "r\tr" copy0 ins            (pos 0: pre "r\t" copy0 ins mid "r")
//   TLI  like IN, copy0 = the linked span (reversed: mode 2)
//   DE / TL   copy0 REF, mid "\ta"
//   IV   copy0 REF, mid "\t", copy1 = REF reverse-complemented
//   DU   copy0 REF (raw), mid "\t", copy1 = copy2 = REF
struct VcfLine {
    unsigned long long start, end, svlen;            // POS; END / SVLEN of INFO
    const uint8_t *src[3];
    unsigned long long len[3];
    uint32_t mode;                                   // 2 bits per copy
    uint32_t pre, mid;                               // characters, first one in the low byte
    uint32_t n_pre, n_mid, svtype;                   // svtype: 0 none (SNP), 1 INS, 2 DEL, 3 INV, 4 DUP, 5 INS:ME, 6 DEL:ME
    bool none;                                       // REF == ALT: no line (vcf_writer.py:123)
};

// CHECK: evaluate the suppression rules that need bases -- the length pass does, the write pass skips records of length 0.
// snp_ra: REF | ALT << 8 of an SNP record where the caller has them already (the write pass: from the length pass), else 0.
template <bool CHECK>
__device__ __forceinline__ VcfLine vcf_describe(const msim_record &r, const uint8_t *__restrict__ pool, const uint8_t *__restrict__ in,
                                                unsigned long long L, const uint8_t *lut, uint32_t snp_ra) {
    VcfLine d;
    const unsigned long long pos = r.pos, stop = r.stop;
    d.start = pos + 1; d.end = 0; d.svlen = 0;
    d.src[0] = d.src[1] = d.src[2] = in;
    d.len[0] = d.len[1] = d.len[2] = 0;
    d.mode = 0; d.pre = 0; d.mid = 0; d.n_pre = 0; d.n_mid = 0; d.svtype = 0; d.none = false;
    switch (r.type) {
        case MSIM_SN: {                                                  // mutator.py:334-341
            uint32_t ref = snp_ra & 255, alt = snp_ra >> 8;
            if (CHECK) {
                const uint8_t x = in[pos];
                ref = lut[768 + x];
                alt = lut[(uint32_t)r.aux * 256 + x];                    // ti / tv column of conv(x)
            }
            d.none = ref == alt;
            d.pre = ref | ((uint32_t)'\t' << 8) | (alt << 16); d.n_pre = 3;
            break;
        }
        case MSIM_IN: {                                                  // mutator.py:343-358
            const uint32_t ref = lut[768 + in[pos > 0 ? pos - 1 : 0]];
            d.src[0] = pool + r.extra; d.len[0] = stop + 1 - pos;
            d.svtype = 1; d.svlen = d.len[0];
            if (pos > 0) { d.start = pos; d.pre = ref | ((uint32_t)'\t' << 8) | (ref << 16); d.n_pre = 3; }
            else { d.start = 1; d.pre = ref | ((uint32_t)'\t' << 8); d.n_pre = 2; d.mid = ref; d.n_mid = 1; }
            d.end = d.start;
            break;
        }
        case MSIM_TLI: {                                                 // mutator.py:401-421
            const unsigned long long src = r.extra;
            const unsigned long long hi = stop + 1 < L ? stop + 1 : L;
            const unsigned long long ilen = hi > src ? hi - src : 0;
            const bool rev = r.aux & 1, after = r.aux & 2;
            d.none = ilen == 0;                                          // REF == ALT: suppressed
            const uint32_t ref = lut[768 + in[after ? pos - 1 : pos]];
            d.start = after ? pos : pos + 1;
            if (after) { d.pre = ref | ((uint32_t)'\t' << 8) | (ref << 16); d.n_pre = 3; }
            else { d.pre = ref | ((uint32_t)'\t' << 8); d.n_pre = 2; d.mid = ref; d.n_mid = 1; }
            d.src[0] = rev ? in + hi - 1 : in + src; d.len[0] = ilen; d.mode = rev ? 2u : 1u;
            d.svtype = 5; d.end = d.start; d.svlen = ilen;
            break;
        }
        case MSIM_TL:
        case MSIM_DE: {                                                  // mutator.py:360-377
            unsigned long long end = stop + 1, lo = pos - 1;
            d.start = pos;
            if (pos == 0) { d.start = 1; end = stop + 2; lo = 0; }
            const unsigned long long hi = end < L ? end : L;             // slice clamps at len(sequence)
            d.src[0] = in + lo; d.len[0] = hi - lo; d.mode = 1u;
            d.mid = (uint32_t)'\t' | ((uint32_t)lut[768 + in[pos > 0 ? lo : hi - 1]] << 8); d.n_mid = 2;     // REF[0] / REF[-1]
            d.svtype = r.type == MSIM_DE ? 2 : 6; d.end = end; d.svlen = stop - pos + 1;
            break;
        }
        case MSIM_IV: {                                                  // mutator.py:379-387
            const unsigned long long len = stop - pos + 1;
            if (CHECK) {                                                 // REF == ALT (a reverse-complement palindrome): suppressed
                bool diff = false;                                       // (three of four random pairs differ: ~1.3 steps)
                for (unsigned long long q = 0; q < len && !diff; q++)
                    diff = lut[768 + in[pos + q]] != lut[1024 + in[stop - q]];
                d.none = !diff;
            }
            d.src[0] = in + pos; d.len[0] = len; d.src[1] = in + stop; d.len[1] = len; d.mode = 1u | (2u << 2);
            d.mid = (uint32_t)'\t'; d.n_mid = 1;
            d.svtype = 3; d.end = stop + 1; d.svlen = 0;
            break;
        }
        case MSIM_DU: {                                                  // mutator.py:389-399 (REF not converted)
            const unsigned long long len = stop - pos + 1;
            d.src[0] = d.src[1] = d.src[2] = in + pos; d.len[0] = d.len[1] = d.len[2] = len;
            d.mid = (uint32_t)'\t'; d.n_mid = 1;
            d.svtype = 4; d.end = pos + len; d.svlen = len;
            break;
        }
        default: d.none = true; break;
    }
    return d;
}

// the six SVTYPE names, eight bytes each (first character in the low byte) and their lengths
__device__ __forceinline__ unsigned long long vcf_svname(uint32_t t, uint32_t &n) {
    constexpr unsigned long long INS = 0x534e49ull, DEL = 0x4c4544ull, INV = 0x564e49ull, DUP = 0x505544ull,
                                 INSME = 0x454d3a534e49ull, DELME = 0x454d3a4c4544ull;
    n = t >= 5 ? 6u : 3u;
    return t == 1 ? INS : t == 2 ? DEL : t == 3 ? INV : t == 4 ? DUP : t == 5 ? INSME : DELME;
}

template <class S>
__device__ __forceinline__ void vcf_emit(S &s, const VcfLine &d, const uint8_t *name, uint32_t name_len, const uint8_t *lut) {
    s.begin(d);
    s.name(name, name_len);
    s.put('\t');
    s.num(d.start);
    s.lit("\t.\t");
    s.chars(d.pre, d.n_pre);
    s.bulk(0, d.src[0], d.len[0], (int)(d.mode & 3), lut);
    s.chars(d.mid, d.n_mid);
    s.bulk(1, d.src[1], d.len[1], (int)((d.mode >> 2) & 3), lut);
    s.bulk(2, d.src[2], d.len[2], (int)((d.mode >> 4) & 3), lut);
    s.lit("\t.\t.\t");
    if (d.svtype) {
        s.lit("SVTYPE=");
        uint32_t k;
        const unsigned long long nm = vcf_svname(d.svtype, k);
        s.chars64(nm, k);
        s.lit(";END=");
        s.num(d.end);
        s.lit(";SVLEN=");
        s.num(d.svlen);
    } else {
        s.put('.');
    }
    s.lit("\tGT\t1\n");
}

// Which waves the stage scheme cannot describe (wave-uniform): 64 lines spanning 4 GiB of text, source addresses beyond 2^48 (the
// pieces pack them), or the test hook.  k_vcf_lines<true, false> leaves them alone, k_vcf_plain writes them.
__device__ __forceinline__ bool vcf_wave_is_plain(unsigned long long first, unsigned long long end, const uint8_t *in, unsigned long long L,
                                                  const uint8_t *pool, uint32_t plain) {
    const unsigned long long top = (unsigned long long)(uintptr_t)(in + L) | (unsigned long long)(uintptr_t)(pool + (1ull << 32));
    return end - first >= (1ull << 32) - 16 || (top >> 48) != 0 || plain != 0;    // (plain: the test hook, or a name beyond VCF_MAX_NAME)
}
constexpr uint32_t VCF_MAX_NAME = 1024;       // one line's compact text (name + < 300 bytes) must fit the stage

// the gaps of a line written at text offset `off` (what SSink::bulk will open)
__device__ __forceinline__ unsigned long long vcf_line_gaps(const VcfLine &d, unsigned long long off, uint32_t name_len) {
    unsigned long long n = off + name_len + 1 + (unsigned)ndigits(d.start) + 3 + d.n_pre, gaps = 0, ia, ib;
    if (vcf_piece_of(n, d.len[0], ia, ib)) gaps += ib - ia;
    n += d.len[0] + d.n_mid;
    if (vcf_piece_of(n, d.len[1], ia, ib)) gaps += ib - ia;
    n += d.len[1];
    if (vcf_piece_of(n, d.len[2], ia, ib)) gaps += ib - ia;
    return gaps;
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
// The copies of a wave's records, made by the whole wave: a UNIT is what of a copy falls into one aligned 16-byte unit of the
// text.  A whole unit inside a copy's interior goes source -> text (one unaligned 16-byte load, one aligned store); the others --
// a short copy's units, the edges in front of and behind an interior -- go byte by byte into the stage, where the lane left
// room for them.  pre[s] = units in front of slot s (pre[64 * VCF_SLOTS] = all of them); a lane takes units lane, lane + 64, ...
// and finds each one's slot by binary search; two units per lane are in flight (their loads are issued together).
// Round 4 took the long records one after the other, every copy a dependent load -> table -> store chain of the whole wave:
// latency-bound at 0.2 TB/s of text on the SV mix (DU / IV of 50-500 bases: 85 % of the bytes).
struct VcfUnit { unsigned long long dst; uint32_t mode, cnt, spos; uint4 w; };       // cnt: 0 nothing, 16 a whole interior unit (dst),
                                                                                      // else cnt bytes at stage position spos
__device__ __forceinline__ uint32_t vcf_piece_units(const VcfPiece &pc) {
    return ((pc.n_rel + pc.len - 1) >> 4) - (pc.n_rel >> 4) + 1;                     // (len >= 1)
}

__device__ __forceinline__ void vcf_unit_load(VcfUnit &u, uint32_t idx, uint32_t n_units, const VcfPiece *pcs, const uint32_t *pre,
                                              unsigned long long origin) {
    u.cnt = 0;
    if (idx >= n_units) return;
    uint32_t s = 0;
#pragma unroll
    for (int step = 128; step; step >>= 1) if (s + step <= 64 * VCF_SLOTS && pre[s + step] <= idx) s += step;
    const VcfPiece pc = pcs[s];
    const uint32_t tu = (pc.n_rel >> 4) + (idx - pre[s]);                // text unit, counted from the origin
    const uint32_t lo = max(pc.n_rel, tu << 4), hi = min(pc.n_rel + pc.len, (tu + 1) << 4);      // (a wave's stretch stays below 4 GiB)
    const uint32_t k0 = lo - pc.n_rel;                                   // copy offset of the unit's first byte
    u.mode = (uint32_t)(pc.src_ws_mode >> 62);
    const uint8_t *src = reinterpret_cast<const uint8_t *>((uintptr_t)(pc.src_ws_mode & ((1ull << 48) - 1)));
    const uint32_t ws = (uint32_t)(pc.src_ws_mode >> 48) & 4095u;
    if (u.mode == 2) __builtin_memcpy(&u.w, src - k0 - 15, 16);          // one unaligned 16-byte load (slack on both sides of the
    else __builtin_memcpy(&u.w, src + k0, 16);                           //  buffers -- ctx.h: PAD -- keeps a partial unit's inside)
    u.cnt = hi - lo;
    const unsigned long long n = origin + pc.n_rel;
    unsigned long long ia, ib;
    const bool interior = vcf_piece_of(n, pc.len, ia, ib);
    if (u.cnt == 16 && interior && origin + lo >= ia && origin + hi <= ib) {
        u.dst = origin + lo;
    } else {                                                             // stage position: the copy's offset, minus its gap behind it
        u.cnt |= 32u;
        u.spos = ws + k0 - (interior && origin + lo >= ib ? (uint32_t)(ib - ia) : 0u);
    }
}

__device__ __forceinline__ void vcf_unit_store(const VcfUnit &u, char *__restrict__ text, char *stage, const uint8_t *lut) {
    if (u.cnt == 0) return;
    uint4 v = u.w;
    if (u.mode == 2) v = uint4{__builtin_bswap32(v.w), __builtin_bswap32(v.z), __builtin_bswap32(v.y), __builtin_bswap32(v.x)};   // read backwards
    if (u.mode) v = lut16(v, lut + (u.mode == 1 ? 768 : 1024));          // conv / comp(conv)
    if (u.cnt == 16) {
        *reinterpret_cast<uint4 *>(text + u.dst) = v;
    } else {
        const uint32_t c = u.cnt & 31u;
        unsigned long long lo = (unsigned long long)v.x | ((unsigned long long)v.y << 32), hi = (unsigned long long)v.z | ((unsigned long long)v.w << 32);
        for (uint32_t i = 0; i < c; i++) {
            stage[u.spos + i] = (char)lo;
            lo = (lo >> 8) | (hi << 56);
            hi >>= 8;
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
// types' formatter.
// A wave takes 64 consecutive records, a LANE formats its own record's line (round 1 gave every line a wave: ~15 instructions of
// 64 lanes for 25 bytes, 135 GB/s).
// Length pass: nothing but the record, one base per SNP and the ends of an inversion are read -- every length follows from
// (type, pos, stop).
// Write pass: the lines of a wave's records are adjacent in the text, so the wave builds them in LDS and stores 16 aligned bytes
// per lane -- minus the long REF / ALT copies (a duplication of 275 bases is a line of 885 bytes; 64 such lines do not fit any
// stage): their aligned interiors are gaps of the stage, copied source -> text by the whole wave (vcf_unit_*), and the flush
// shifts every stage unit by the gaps in front of it.  No byte of text is stored other than as part of an aligned 16-byte store
// (the first and the last unit of a wave's stretch excepted).  Records whose compact text does not fit the stage together go
// in groups (consecutive lanes).  Round 4 let a lane store its line byte by byte unless ALL 64 lines were short, and took long
// records one after the other, every copy a dependent load -> table -> store chain: 0.2 TB/s of text on the SV mix.
constexpr uint32_t VCF_O_GS = VCF_STAGE;                                                // u32[VCF_UNITS]: gap units in front of a stage unit
constexpr uint32_t VCF_O_PCS = VCF_O_GS + VCF_UNITS * 4;                               // VcfPiece[64 * VCF_SLOTS]
constexpr uint32_t VCF_O_PRE = VCF_O_PCS + 64 * VCF_SLOTS * sizeof(VcfPiece);          // u32[64 * VCF_SLOTS + 1] (+ pad)
constexpr uint32_t VCF_WAVE_LDS = VCF_O_PRE + (64 * VCF_SLOTS + 4) * 4;

template <bool WRITE, bool ALL_SNP>
__global__ __launch_bounds__(TX_THREADS) void k_vcf_lines(const msim_record *__restrict__ recs, uint32_t n_rec,
                                                          const uint8_t *__restrict__ pool,
                                                          const uint8_t *__restrict__ in, unsigned long long L,
                                                          const uint8_t *__restrict__ name, uint32_t name_len,
                                                          const uint8_t *__restrict__ lut_g,
                                                          uint32_t *__restrict__ len_io, uint16_t *__restrict__ ra,
                                                          const unsigned long long *__restrict__ off,
                                                          char *__restrict__ text, uint32_t plain) {
    __shared__ uint8_t lut[1280];
    __shared__ uint8_t lname[64];                                        // the sequence name (every line starts with it)
    constexpr uint32_t WB = !WRITE ? 16 : ALL_SNP ? VCF_STAGE : VCF_WAVE_LDS;
    __shared__ __attribute__((aligned(16))) char wbuf[WRITE ? TX_WAVES : 1][WB];
    for (int i = threadIdx.x; i < 1280 / 4; i += TX_THREADS)
        reinterpret_cast<uint32_t *>(lut)[i] = reinterpret_cast<const uint32_t *>(lut_g)[i];
    if (threadIdx.x < 64 && threadIdx.x < name_len) lname[threadIdx.x] = name[threadIdx.x];
    __syncthreads();
    const uint8_t *nm = name_len <= 64 ? lname : name;
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
            const VcfLine d = vcf_describe<true>(my, pool, in, L, lut, 0);
            CSink s;
            s.n = 0;
            if (!d.none) vcf_emit(s, d, name, name_len, lut);
            len_io[mine] = (uint32_t)s.n;
        }
        return;
    }
    // ---- write pass
    char *stage = wbuf[WRITE ? wave : 0];
    unsigned long long my_off = 0;
    uint32_t my_len = 0;
    if (valid) { my_off = off[mine]; my_len = len_io[mine]; }
    const uint32_t last = min(63u, n_rec - 1 - base_rec);
    if (ALL_SNP) {
        const unsigned long long start0 = __shfl(my_off, 0, 64);
        const unsigned long long end = __shfl(my_off + my_len, (int)last, 64);
        const uint32_t stretch = (uint32_t)min(end - start0, (unsigned long long)(2 * VCF_STAGE));
        const uint32_t phase = (uint32_t)(start0 & 15);
        const bool staged = phase + stretch <= VCF_STAGE;                 // (wave-uniform; else: long names)
        if (valid && my_len) {
            const uint32_t r2 = ra[mine];
            const unsigned long long start = (unsigned long long)my.pos + 1;
            if (staged) snp_line(&stage[phase + (uint32_t)(my_off - start0)], name, name_len, start, ndigits(start), (uint8_t)r2, (uint8_t)(r2 >> 8));
            else snp_line(text + my_off, name, name_len, start, ndigits(start), (uint8_t)r2, (uint8_t)(r2 >> 8));
        }
        if (!staged) return;
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
    } else {
        uint32_t *gs = reinterpret_cast<uint32_t *>(stage + VCF_O_GS);
        VcfPiece *pcs = reinterpret_cast<VcfPiece *>(stage + VCF_O_PCS);
        uint32_t *pre = reinterpret_cast<uint32_t *>(stage + VCF_O_PRE);
        auto wave_sync = []() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        // this lane's line, and its gaps (text coordinates are needed: a piece's interior is aligned in the text)
        const bool live = valid && my_len != 0;
        VcfLine d = vcf_describe<false>(my, pool, in, L, lut, live && is_snp ? (uint32_t)ra[mine] : 0u);
        if (vcf_wave_is_plain(__shfl(my_off, 0, 64), __shfl(my_off + my_len, (int)last, 64), in, L, pool, plain)) return;   // k_vcf_plain's
        const unsigned long long my_gap = live ? vcf_line_gaps(d, my_off, name_len) : 0ull;
        // compact size of the lines up to and including this lane's
        unsigned long long cp = (unsigned long long)my_len - my_gap;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(cp, o, 64);
            if (lane_id >= (uint32_t)o) cp += t;
        }
        const unsigned long long cp_excl = cp - ((unsigned long long)my_len - my_gap);
        for (uint32_t lo = 0; lo <= last;) {                             // groups of consecutive lanes that fit the stage together
            const unsigned long long off_lo = __shfl(my_off, (int)lo, 64), cp_lo = __shfl(cp_excl, (int)lo, 64);
            const unsigned long long origin = off_lo & ~15ull;
            const uint32_t phase = (uint32_t)(off_lo & 15);
            const unsigned long long fits = __ballot(lane_id >= lo && lane_id <= last && cp - cp_lo + phase <= VCF_STAGE);
            const unsigned long long rest = ~(fits >> lo);               // first lane from lo on that does not fit
            uint32_t hi = lo + (rest ? (uint32_t)__builtin_ctzll(rest) : 64u - lo);
            hi = min(hi, last + 1);
            if (hi == lo) hi = lo + 1;                                   // (cannot happen: one line's compact text is a few hundred bytes)
            const uint32_t csize = (uint32_t)(__shfl(cp, (int)(hi - 1), 64) - cp_lo) + phase;      // compact bytes incl. the phase
            const uint32_t units = (csize + 15) >> 4;
            reinterpret_cast<uint4 *>(gs)[lane_id] = uint4{0, 0, 0, 0};  // (VCF_UNITS == 256 == 64 lanes x 4)
            wave_sync();
            SSink s;
            s.stage = stage; s.gs = gs; s.slots = pcs + lane_id * VCF_SLOTS; s.np = 0; s.origin = origin;
            const bool in_group = lane_id >= lo && lane_id < hi;
            if (in_group && live) {
                s.w = phase + (uint32_t)(cp_excl - cp_lo);
                s.n = my_off;
                vcf_emit(s, d, nm, name_len, lut);
            }
            // ---- the pieces' interiors: source -> text
            uint32_t cnt[VCF_SLOTS], tot = 0;
#pragma unroll
            for (int q = 0; q < VCF_SLOTS; q++) {
                cnt[q] = (uint32_t)q < s.np ? vcf_piece_units(s.slots[q]) : 0u;
                tot += cnt[q];
            }
            uint32_t incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
                if (lane_id >= (uint32_t)o) incl += t;
            }
            const uint32_t n_units = (uint32_t)__shfl((int)incl, 63, 64);
            uint32_t run = incl - tot;
#pragma unroll
            for (int q = 0; q < VCF_SLOTS; q++) { pre[lane_id * VCF_SLOTS + q] = run; run += cnt[q]; }
            if (lane_id == 63) pre[64 * VCF_SLOTS] = n_units;
            wave_sync();
            for (uint32_t u0 = 0; u0 < n_units; u0 += 128) {
                VcfUnit a, b;
                vcf_unit_load(a, u0 + lane_id, n_units, pcs, pre, origin);
                vcf_unit_load(b, u0 + 64 + lane_id, n_units, pcs, pre, origin);
                vcf_unit_store(a, text, stage, lut);
                vcf_unit_store(b, text, stage, lut);
            }
            // ---- the stage: unit x goes to text unit x + (gap units in front of it)
            {
                uint4 v = reinterpret_cast<const uint4 *>(gs)[lane_id];
                v.y += v.x; v.z += v.y; v.w += v.z;
                uint32_t inc2 = v.w;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_up((int)inc2, o, 64);
                    if (lane_id >= (uint32_t)o) inc2 += t;
                }
                const uint32_t before = inc2 - v.w;
                v.x += before; v.y += before; v.z += before; v.w += before;
                reinterpret_cast<uint4 *>(gs)[lane_id] = v;
            }
            wave_sync();
            for (uint32_t x = lane_id; x < units; x += 64) {
                char *g0 = text + origin + 16ull * ((unsigned long long)x + gs[x]);
                const uint32_t b0 = x == 0 ? phase : 0u, b1 = x + 1 == units ? csize - 16 * x : 16u;
                if (b0 == 0 && b1 == 16) *reinterpret_cast<uint4 *>(g0) = *reinterpret_cast<const uint4 *>(&stage[16 * x]);
                else for (uint32_t q = b0; q < b1; q++) g0[q] = stage[16 * x + q];
            }
            wave_sync();                                                 // (the next group reuses the stage)
            lo = hi;
        }
    }
}

// The waves k_vcf_lines<true, false> left alone: every lane writes its own line byte by byte.
__global__ __launch_bounds__(TX_THREADS) void k_vcf_plain(const msim_record *__restrict__ recs, uint32_t n_rec,
                                                          const uint8_t *__restrict__ pool, const uint8_t *__restrict__ in,
                                                          unsigned long long L, const uint8_t *__restrict__ name, uint32_t name_len,
                                                          const uint8_t *__restrict__ lut_g, const uint32_t *__restrict__ len_io,
                                                          const uint16_t *__restrict__ ra, const unsigned long long *__restrict__ off,
                                                          char *__restrict__ text, uint32_t plain) {
    __shared__ uint8_t lut[1280];
    for (int i = threadIdx.x; i < 1280 / 4; i += TX_THREADS)
        reinterpret_cast<uint32_t *>(lut)[i] = reinterpret_cast<const uint32_t *>(lut_g)[i];
    __syncthreads();
    const uint32_t lane_id = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base_rec = (blockIdx.x * TX_WAVES + wave) * 64;
    if (base_rec >= n_rec) return;
    const uint32_t mine = base_rec + lane_id;
    const bool valid = mine < n_rec;
    msim_record my{};
    unsigned long long my_off = 0;
    uint32_t my_len = 0;
    if (valid) { my = recs[mine]; my_off = off[mine]; my_len = len_io[mine]; }
    const uint32_t last = min(63u, n_rec - 1 - base_rec);
    if (!vcf_wave_is_plain(__shfl(my_off, 0, 64), __shfl(my_off + my_len, (int)last, 64), in, L, pool, plain)) return;
    if (!valid || my_len == 0) return;
    const VcfLine d = vcf_describe<false>(my, pool, in, L, lut, my.type == MSIM_SN ? (uint32_t)ra[mine] : 0u);
    DSink s;
    s.text = text; s.n = my_off;
    vcf_emit(s, d, name, name_len, lut);
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
                   d_name, (uint32_t)name_len, ctx_lut(c), d_len, d_ra, (const unsigned long long *)nullptr, (char *)nullptr, 0u);
    hipLaunchKernelGGL(k_len_reduce, dim3(nb), dim3(TX_THREADS), 0, st, d_len, n, d_sums);
    hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, st, d_sums, nb, c->h_mail);
    hipLaunchKernelGGL(k_len_offsets, dim3(nb), dim3(TX_THREADS), 0, st, d_len, n, d_sums, d_off);
    MSIM_HIP(c, hipGetLastError());
    MSIM_HIP(c, wait_stream(st));
    const uint64_t total = *c->h_mail;
    rc = dev_reserve(c, (void **)buf, cap, total + 64);
    if (rc) return rc;
    if (total) {
        MSIM_VCF_LINES(true, grid, dim3(TX_THREADS), 0, st, g.d_recs, n, pool, in, (unsigned long long)g.len,
                       d_name, (uint32_t)name_len, ctx_lut(c), d_len, d_ra, d_off, reinterpret_cast<char *>(*buf),
                       (getenv("MSIM_DBG_VCF_PLAIN") || name_len > VCF_MAX_NAME) ? 1u : 0u);      // (the plain path: test hook, huge names)
#undef MSIM_VCF_LINES
        // waves the stage scheme cannot describe (vcf_wave_is_plain): only a text of 4 GiB and more can hold one
        const uint32_t plain = (getenv("MSIM_DBG_VCF_PLAIN") || name_len > VCF_MAX_NAME) ? 1u : 0u;
        const bool far = (((uintptr_t)(in + g.len) | (uintptr_t)(pool + (1ull << 32))) >> 48) != 0;
        if (!g.all_snp && (plain || far || total >= (1ull << 32) - 16))
            hipLaunchKernelGGL(k_vcf_plain, grid, dim3(TX_THREADS), 0, st, g.d_recs, n, pool, in, (unsigned long long)g.len, d_name,
                               (uint32_t)name_len, ctx_lut(c), d_len, d_ra, d_off, reinterpret_cast<char *>(*buf), plain);
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
    MSIM_HIP(c, wait_stream(c->stream));
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
    if (e == hipSuccess) e = wait_stream(c->stream);
    (void)hipFree(d_tab);
    if (e != hipSuccess) return hip_fail(c, e, "splice");
    return MSIM_OK;
}

}  // namespace msim
