// Counter-based sampler (MSIM_RNG_FAST, "--rng fast"): the arithmetic every draw is made of, host + device.
//
// NOT stream-compatible with the reference: the reference draws from two sequential MT19937
// streams, here every draw is a pure function of (key, contig ordinal, what is drawn, index) through Philox4x32-10
// (Salmon et al., SC'11), so nothing chains.  The CONSTRUCTION is the reference's:
//   util.py:93-109   k = int(len * rate) start positions per range = a uniform k-subset of range(start, stop - (k-1) d),
//                    moved up by rank * d
//   mutator.py:160-174   a type per candidate from the range's chances
//   mutator.py:228-265   a length per candidate, uniform in [min, max]; IV near the contig end dropped, DU / DE clamped
//   mutator.py:184-213   the boundary pass (blocked ranges), reset per range
//   mutator.py:428-471   transition with probability p_ti, else one of two transversions; insert bases uniform in ATGC
// so the distributions are the reference's; the numbers are not.
//
// How a uniform k-subset is drawn without any sequential pass (Sanders et al., "Efficient Parallel Random Sampling", 2018):
// the value range is cut into leaves of 2^lgB values; how many of the k points fall into each leaf is a multivariate
// hypergeometric vector, drawn by a binary splitting tree (a node with K points over N values sends
// Hypergeometric(N, N_left, K) of them to the left) whose nodes draw from their own counters; a leaf then draws its m
// distinct values by rejection into a bitmap.  Every leaf of every range of every contig is independent work.
//
// Everything here is integer arithmetic or IEEE double +, -, *, / with contraction OFF and a log / sqrt built from
// those alone: the same (key, settings) give the same mutations on any device and in the numpy restatement of
// tests/fast_twin.py, bit for bit.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MSIM_FHD __host__ __device__
#else
#define MSIM_FHD
#endif
#define MSIM_FP_STRICT _Pragma("clang fp contract(off)")

namespace msim {
namespace fastrng {

// ---- counters ---------------------------------------------------------------------------------------------------------
// philox counter = (x, y, contig ordinal, tag | hi << 8); one call = 128 random bits
constexpr uint32_t TAG_SPLIT = 16;     // x = heap index of the tree node, y = attempt, hi = drawing range
constexpr uint32_t TAG_POS = 17;       // x = draw index >> 1, y = leaf (numbered through the contig), hi = 0
constexpr uint32_t TAG_CAND = 18;      // x = candidate ordinal: low 64 bits -> type (53 bits), high 64 -> length, or the SNP's outcome
constexpr uint32_t TAG_INS = 19;       // x = 64-base chunk of the insert, y = candidate ordinal
constexpr int LG_LEAF_MIN = 10, LG_LEAF_MAX = 16;
constexpr uint32_t LEAF_TARGET = 224;  // a leaf should hold about this many points or more (up to twice as many): 96 -> 224 in round 5 (A/B on one box, interleaved: c4 -5 %, c4sv -4 %, c3 -1 %, c2 +-0; 320: c3 +4 %)

struct U4 { uint32_t x, y, z, w; };

MSIM_FHD inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
struct Key { uint32_t k0, k1, seq; };
MSIM_FHD inline U4 draw4(const Key &k, uint32_t x, uint32_t y, uint32_t tag, uint32_t hi = 0) {
    U4 c; c.x = x; c.y = y; c.z = k.seq; c.w = tag | (hi << 8);
    return philox4x32_10(c, k.k0, k.k1);
}
MSIM_FHD inline uint64_t lo64(const U4 &v) { return ((uint64_t)v.y << 32) | v.x; }
MSIM_FHD inline uint64_t hi64(const U4 &v) { return ((uint64_t)v.w << 32) | v.z; }

MSIM_FHD inline uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
// 64 random bits scaled to [0, n): bias below n / 2^64
MSIM_FHD inline uint64_t below(uint64_t r64, uint64_t n) { return mulhi64(r64, n); }

// ---- IEEE-only double helpers ------------------------------------------------------------------------------------------
MSIM_FHD inline uint64_t d_bits(double x) { union { double d; uint64_t u; } v; v.d = x; return v.u; }
MSIM_FHD inline double d_from(uint64_t u) { union { double d; uint64_t u; } v; v.u = u; return v.d; }
// (u52 + 1/2) / 2^52: strictly inside (0, 1), exact
MSIM_FHD inline double uni52(uint64_t r64) {
    MSIM_FP_STRICT
    return ((double)(r64 >> 12) + 0.5) * (1.0 / 4503599627370496.0);
}
// natural logarithm of a positive normal double: x = 2^e f, f in [sqrt(1/2), sqrt(2)), log f = 2 atanh((f-1)/(f+1)) by
// its series to s^22 (|s| < 0.1716: the first omitted term is below 2^-58 of the result)
MSIM_FHD inline double d_log(double x) {
    MSIM_FP_STRICT
    const uint64_t b = d_bits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double f = d_from((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (f > 1.4142135623730951) { f = f * 0.5; e += 1; }
    const double s = (f - 1.0) / (f + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    const double t = (2.0 * s) * p;
    return (double)e * 0.6931471805599453 + t;
}
// the series of d_log on s = t / (2 + t): log(1 + t) = 2 atanh(s).  One division; |s| < 0.1716 for t in (-0.29, 0.41),
// outside (-0.25, 0.4) the rounding of 1 + t no longer matters and d_log takes over
MSIM_FHD inline double d_log1p(double t) {
    MSIM_FP_STRICT
    if (!(t > -0.25 && t < 0.4)) return d_log(1.0 + t);
    const double s = t / (2.0 + t);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (2.0 * s) * p;
}
// an upper bound of sqrt(x) within 3e-4 of it, x a positive normal double: four Newton steps from 2^floor(e/2) (every
// Newton iterate after the first lies above the root).  The hat of the ratio-of-uniforms sampler only needs a bound.
MSIM_FHD inline double d_sqrt_up(double x) {
    MSIM_FP_STRICT
    const uint64_t b = d_bits(x);
    const int e = (int)((b >> 52) & 0x7ff) - 1023;
    double y = d_from((uint64_t)(1023 + (e >> 1)) << 52);
#pragma unroll
    for (int i = 0; i < 4; i++) y = 0.5 * (y + x / y);
    return y;
}
// Stirling correction of lgamma(y): 1/(12 y) - 1/(360 y^3) + 1/(1260 y^5); y >= 32: the next term is below 2e-14
MSIM_FHD inline double stirling_corr_inv(double iy) {
    MSIM_FP_STRICT
    const double iy2 = iy * iy;
    return iy * (1.0 / 12.0 - iy2 * (1.0 / 360.0 - iy2 * (1.0 / 1260.0)));
}
// log(x!)
MSIM_FHD inline double log_factorial(uint64_t x) {
    MSIM_FP_STRICT
    if (x < 32) {
        double p = 1.0;
        for (uint64_t i = 2; i <= x; i++) p = p * (double)i;
        return d_log(p);
    }
    const double y = (double)(x + 1);
    return (y - 0.5) * d_log(y) - y + 0.9189385332046727 + stirling_corr_inv(1.0 / y);
}
// log((a + d)!) - log(a!) for a FIXED a and many d (a + d >= 0), without the cancellation of two huge logarithms:
// what depends on a alone is prepared once
struct LfBase { uint64_t a; double iy0, ly0m1, c0, lf; };           // 1 / (a + 1), log(a + 1) - 1, stirling_corr(a + 1), log(a!)
MSIM_FHD inline LfBase lf_base(uint64_t a) {
    MSIM_FP_STRICT
    LfBase b;
    b.a = a;
    const double y0 = (double)(a + 1);
    b.iy0 = 1.0 / y0;
    b.ly0m1 = d_log(y0) - 1.0;
    b.c0 = stirling_corr_inv(b.iy0);
    b.lf = a + 1 >= 32 ? 0.0 : log_factorial(a);
    return b;
}
MSIM_FHD inline double log_factorial_diff(const LfBase &b, int64_t d) {
    MSIM_FP_STRICT
    if (d == 0) return 0.0;
    const uint64_t a1 = (uint64_t)((int64_t)b.a + d);
    if (b.a + 1 >= 32 && a1 + 1 >= 32) {
        const double y1 = (double)(a1 + 1), dd = (double)d;
        return (y1 - 0.5) * d_log1p(dd * b.iy0) + dd * b.ly0m1 + (stirling_corr_inv(1.0 / y1) - b.c0);
    }
    return log_factorial(a1) - (b.a + 1 >= 32 ? log_factorial(b.a) : b.lf);
}

// ---- Hypergeometric(good, bad, sample): how many of `sample` items drawn without replacement from good + bad are good ----
// small samples: the urn itself; otherwise Stadlober's ratio-of-uniforms algorithm HRUA (Stadlober 1990, "The ratio of
// uniforms approach for generating discrete random variates", J. Comput. Appl. Math. 31; the variant with the mode-centred
// table mountain NumPy also uses).  Draws come from node counters (x = node, y = attempt): attempt `att` is a pure function
// of (node, att), the result is the outcome of the FIRST accepted attempt -- so a kernel may evaluate several attempts of
// one node on neighbouring lanes at once (fast_kernels.h: hyp_group) and still land on the sequential answer.
struct HypPrep {
    int kind;                  // 0: z is the answer; 1: urn (m <= 10); 2: HRUA
    uint64_t z;
    uint64_t good, bad, sample, N, m, mingb, maxgb, mode;
    double a, h, bnd;
    LfBase t0, t1, t2, t3;     // mode, mingb - mode, m - mode, maxgb - m + mode
};
MSIM_FHD inline HypPrep hyp_prepare(uint64_t good, uint64_t bad, uint64_t sample) {
    MSIM_FP_STRICT
    HypPrep P;
    P.kind = 0; P.z = 0;
    P.good = good; P.bad = bad; P.sample = sample;
    const uint64_t N = good + bad;
    P.N = N;
    P.m = P.mingb = P.maxgb = P.mode = 0;
    P.a = P.h = P.bnd = 0.0;
    P.t0 = P.t1 = P.t2 = P.t3 = LfBase{0, 0.0, 0.0, 0.0, 0.0};
    if (sample == 0 || good == 0) return P;
    if (bad == 0) { P.z = sample; return P; }
    if (sample >= N) { P.z = good; return P; }
    const uint64_t m = sample < N - sample ? sample : N - sample;
    P.m = m;
    if (m <= 10) { P.kind = 1; return P; }
    P.kind = 2;
    const uint64_t mingb = good < bad ? good : bad, maxgb = good < bad ? bad : good;
    P.mingb = mingb; P.maxgb = maxgb;
    const double p = (double)mingb / (double)N, q = 1.0 - p;
    P.a = (double)m * p + 0.5;
    const double var = (double)(N - m) * (double)m * p * q / (double)(N - 1);
    const double c = d_sqrt_up(var + 0.5);
    P.h = 1.7155277699214135 * c + 0.8989161620588988;     // 2 sqrt(2/e) c + 3 - 2 sqrt(3/e)
    P.mode = (m + 1) * (mingb + 1) / (N + 2);
    const double lim_a = (double)((m < mingb ? m : mingb) + 1);
    double lim_b = P.a + 16.0 * c;
    lim_b = (double)(uint64_t)lim_b;                       // floor of a positive number
    P.bnd = lim_a < lim_b ? lim_a : lim_b;
    P.t0 = lf_base(P.mode);
    P.t1 = lf_base(mingb - P.mode);
    P.t2 = lf_base(m - P.mode);
    P.t3 = lf_base(maxgb - m + P.mode);
    return P;
}
// one HRUA attempt: true = accepted, Z = how many of the m items are of the rarer kind
MSIM_FHD inline bool hyp_attempt(const HypPrep &P, const Key &key, uint32_t node, uint32_t range, uint32_t att, uint64_t &Z) {
    MSIM_FP_STRICT
    const U4 v = draw4(key, node, att, TAG_SPLIT, range);
    const double U = uni52(lo64(v)), V = uni52(hi64(v));
    const double X = P.a + P.h * (V - 0.5) / U;
    if (X < 0.0 || X >= P.bnd) return false;
    Z = (uint64_t)X;
    const int64_t dz = (int64_t)Z - (int64_t)P.mode;
    const double T = -(log_factorial_diff(P.t0, dz) + log_factorial_diff(P.t1, -dz) + log_factorial_diff(P.t2, -dz) +
                       log_factorial_diff(P.t3, dz));
    if (U * (4.0 - U) - 3.0 <= T) return true;
    if (U * (U - T) >= 1.0) return false;
    return 2.0 * d_log(U) <= T;
}
MSIM_FHD inline uint64_t hyp_urn(const HypPrep &P, const Key &key, uint32_t node, uint32_t range) {   // the m items one by one
    uint64_t g = P.good, n = P.N, cnt = 0;
    U4 v{};
    for (uint64_t i = 0; i < P.m; i++) {
        if (!(i & 1)) v = draw4(key, node, (uint32_t)(i >> 1), TAG_SPLIT, range);
        const uint64_t r = (i & 1) ? hi64(v) : lo64(v);
        if (below(r, n) < g) { g--; cnt++; }
        n--;
    }
    return cnt;                                            // good items among the m drawn
}
// kind 1: z = hyp_urn's count; kind 2: z = the accepted attempt's Z
MSIM_FHD inline uint64_t hyp_finish(const HypPrep &P, uint64_t z) {
    if (P.kind == 0) return P.z;
    if (P.kind == 2) z = P.good > P.bad ? P.m - z : z;     // Z counted the rarer kind
    return P.m < P.sample ? P.good - z : z;                // the m items were the ones left OUT
}
constexpr uint32_t HYP_MAX_ATTEMPTS = 4096;                // (an acceptance rate above 1/2 per attempt: never reached)
MSIM_FHD inline uint64_t hypergeometric(uint64_t good, uint64_t bad, uint64_t sample, const Key &key, uint32_t node, uint32_t range,
                                       uint32_t *attempts_out = nullptr) {
    const HypPrep P = hyp_prepare(good, bad, sample);
    uint64_t z = 0;
    uint32_t att = 0;
    if (P.kind == 1) z = hyp_urn(P, key, node, range);
    else if (P.kind == 2) {
        z = P.mode;
        for (; att < HYP_MAX_ATTEMPTS; att++)
            if (hyp_attempt(P, key, node, range, att, z)) break;
        if (att >= HYP_MAX_ATTEMPTS) z = P.mode;
    }
    if (attempts_out) *attempts_out = att;
    return hyp_finish(P, z);
}

// ---- leaves ----------------------------------------------------------------------------------------------------------
// leaf size of a range with k points over n values: the smallest 2^e, e in [10, 16], holding LEAF_TARGET points on average
MSIM_FHD inline uint32_t leaf_lg(uint64_t n, uint64_t k) {
    uint32_t e = LG_LEAF_MIN;
    while (e < (uint32_t)LG_LEAF_MAX && (k << e) < (uint64_t)LEAF_TARGET * n) e++;
    return e;
}
// value of draw j of leaf `leaf` (len values): two draws per counter
MSIM_FHD inline uint32_t leaf_draw(const Key &key, uint32_t leaf, uint32_t j, uint32_t len) {
    const U4 v = draw4(key, j >> 1, leaf, TAG_POS);
    return (uint32_t)below((j & 1) ? hi64(v) : lo64(v), len);
}

// ---- candidates ------------------------------------------------------------------------------------------------------
// one MutationSettings object (rmt.py:79-163) as the draws need it
struct Settings {
    uint64_t thr[8];           // msim_range.cdf_thr
    uint32_t min_len[8];       // indexed by MSIM_* id
    uint32_t width[8];         // max_len - min_len + 1
    uint32_t max_len[8];
    uint8_t type[8];           // msim_range.types
    uint32_t n_types;
    uint32_t rsv;
};
// candidate meta byte: bits 0-2 type, bits 3-4 SNP outcome (0 transition, 1 / 2 transversion column), then flags
constexpr uint8_t CAND_DROPPED = 0x40;   // IV too close to the contig end (mutator.py:240-243): no record, blocks nothing
constexpr uint8_t CAND_KEEP = 0x80;
constexpr uint8_t CAND_VISIT = 0x20;
constexpr int CAND_AUX_SHIFT = 3;
struct Cand { uint32_t stop; uint32_t bend; uint8_t meta; };   // Mutation.stop, end of the blocked range it opens, type | aux | flags

// SNP outcome from 64 random bits: transition iff the 53-bit sample is below ti_lim (mutator.py:436-438: p <= p_ti), else one
// of the two transversion columns by the lowest bit (mutator.py:449-455)
MSIM_FHD inline uint8_t snp_from(uint64_t r64, uint64_t ti_lim) { return (r64 >> 11) < ti_lim ? (uint8_t)0 : (uint8_t)(1 + (r64 & 1u)); }

MSIM_FHD inline uint32_t sat_add32(uint32_t a, uint64_t b) {
    const uint64_t s = (uint64_t)a + b;
    return s > 0xffffffffull ? 0xffffffffu : (uint32_t)s;
}
// type, stop and blocked end of candidate `ord` at position pos (mutator.py:160-174, 184-265); block1[t] = block[t] + 1
MSIM_FHD inline Cand cand_draw(const Key &key, uint32_t ord, uint32_t pos, uint64_t L, const Settings &s,
                              const uint32_t *block1, uint32_t clip, uint64_t ti_lim) {
    const U4 v = draw4(key, ord, 0, TAG_CAND);
    const uint64_t u53 = lo64(v) >> 11;
    uint32_t idx = 0;
    for (uint32_t j = 0; j < 8; j++) idx += (j < s.n_types && s.thr[j] <= u53) ? 1u : 0u;
    if (idx >= s.n_types) idx = s.n_types - 1;
    const uint32_t t = s.type[idx];
    Cand c;
    c.meta = (uint8_t)t;
    uint64_t stop = pos;
    if (t == 1) c.meta |= (uint8_t)(snp_from(hi64(v), ti_lim) << CAND_AUX_SHIFT);      // an SNP draws no length: its outcome instead
    if (t != 1) {                                          // randint(start + min - 1, start + max - 1)
        stop = (uint64_t)pos + s.min_len[t] - 1 + below(hi64(v), s.width[t]);
        if (t == 5) {                                      // IV: dropped when it cannot fit (mutator.py:240-243)
            if ((uint64_t)pos + s.max_len[t] >= L - 1) { c.meta |= CAND_DROPPED; stop = pos; }
        } else if (t != 2) {                               // DU / DE clamp to the contig end (mutator.py:253-262)
            if (stop > L - 1) stop = L - 1;
        }
    }
    c.stop = stop > 0xffffffffull ? 0xffffffffu : (uint32_t)stop;
    uint32_t e;
    if (c.meta & CAND_DROPPED) e = pos + 1;                // takes part in nothing
    else if (t == 1 || t == 2) e = sat_add32(pos, block1[t]);          // range(start, start + 1 + block)  mutator.py:204-206
    else e = sat_add32(c.stop, block1[t]);                             // range(start, stop + 1 + block)   mutator.py:208-209
    c.bend = e < clip ? e : clip;                          // the blocked range is reset per range (mutator.py:184)
    if (c.bend <= pos) c.bend = pos + 1;
    return c;
}
// SNP outcome of candidate `ord` where every candidate is an SNP (the type draw is not looked at): the same bits cand_draw uses
MSIM_FHD inline uint8_t snp_outcome(const Key &key, uint32_t ord, uint64_t ti_lim) {
    return snp_from(hi64(draw4(key, ord, 0, TAG_CAND)), ti_lim);
}
// base j of the insert of candidate `ord` (mutator.py:465-471): 64 bases per counter, 2 bits each
MSIM_FHD inline uint8_t insert_base_of(const U4 &chunk, uint32_t j) {
    const uint32_t w = (j >> 4) == 0 ? chunk.x : (j >> 4) == 1 ? chunk.y : (j >> 4) == 2 ? chunk.z : chunk.w;
    return (uint8_t)"ATGC"[(w >> (2 * (j & 15))) & 3u];
}

}  // namespace fastrng
}  // namespace msim
