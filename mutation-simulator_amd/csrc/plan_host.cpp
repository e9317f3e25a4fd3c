// Host planner: the exact, sequential walk of the reference's two MT19937 streams for one contig.
//
// This is the general PLAN engine: it covers every stream structure the reference can produce
// (many ranges per contig, CPython's pool-path sample, the boundary pass whose randint draws chain
// through the kept/dropped decisions -- SURVEY.md 7.3 H2).  The GPU sampler (plan_gpu.hip) takes
// over where the structure parallelises exactly.  Nothing here touches genome bytes: RNG use is
// independent of sequence content, which is what makes the PLAN / APPLY split possible.
//
// Follows (behaviour, not text): mutator.py:144-265 (__get_mutations, __get_mut_positions,
// __get_stop_position), util.py:94-109 (sample_with_minimum_distance), CPython Lib/random.py
// (sample, _randbelow_with_getrandbits, randint, random), NumPy legacy RandomState.choice, and the
// draw order of mutator.py:318-471 (__mutate_sequence, __get_snp, __get_insert).
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>

#include "ctx.h"

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define MSIM_X86_HOST 1
#endif

namespace msim {

namespace {

// _randbelow() over a word window, in bulk: append to buf the next `need` ACCEPTED draws (word >> sh < n) starting at
// word w and return the index one past the word that held the last of them (SIZE_MAX: the window ran out).  Exactly
// what `need` calls of _randbelow_with_getrandbits consume.  The AVX-512 clone filters 16 words per step
// (shift, compare, compress-store); the chain only meets the accepted values afterwards.
inline size_t collect_accepted_scalar(const uint32_t *words, size_t w, size_t n_words, int sh, uint32_t n, size_t need,
                                      uint32_t *buf) {
    size_t got = 0;
    while (got < need) {
        if (w >= n_words) return SIZE_MAX;
        const uint32_t v = words[w++] >> sh;
        buf[got] = v;
        got += v < n;
    }
    return w;
}
#ifdef MSIM_X86_HOST
__attribute__((target("avx512f,bmi2,popcnt"))) size_t collect_accepted_avx512(const uint32_t *words, size_t w, size_t n_words,
                                                                               int sh, uint32_t n, size_t need, uint32_t *buf) {
    size_t got = 0;
    const __m512i nv = _mm512_set1_epi32((int)n);
    const __m128i shv = _mm_cvtsi32_si128(sh);
    while (got < need && w + 16 <= n_words) {
        const __m512i v = _mm512_srl_epi32(_mm512_loadu_si512(words + w), shv);
        const __mmask16 m = _mm512_cmplt_epu32_mask(v, nv);
        const unsigned c = (unsigned)__builtin_popcount(m);
        // (compress in the register + a full 16-lane store: the masked compress-STORE is microcoded on Zen 4; buf has
        //  16 entries of slack behind `need`)
        if (got + c < need) {
            _mm512_storeu_si512(buf + got, _mm512_maskz_compress_epi32(m, v));
            got += c;
            w += 16;
        } else {                                         // the need-th accepted draw is in this block: cut there
            const unsigned r = (unsigned)(need - got);   // 1 <= r <= c
            const unsigned pos = (unsigned)__builtin_ctz(_pdep_u32(1u << (r - 1), (unsigned)m));
            const __mmask16 m2 = (__mmask16)(m & ((2u << pos) - 1u));
            _mm512_storeu_si512(buf + got, _mm512_maskz_compress_epi32(m2, v));
            return w + pos + 1;
        }
    }
    while (got < need) {                                 // tail of the window
        if (w >= n_words) return SIZE_MAX;
        const uint32_t v = words[w++] >> sh;
        buf[got] = v;
        got += v < n;
    }
    return w;
}
#endif
// f(wi) for every non-zero 64-bit word of B[0 .. nw), ascending (B may be modified by f at wi only)
template <class F>
inline void nonzero_words_scalar(const uint64_t *B, size_t nw, F &f) {
    for (size_t wi = 0; wi < nw; wi++)
        if (B[wi]) f(wi);
}
#ifdef MSIM_X86_HOST
template <class F>
__attribute__((target("avx512f"))) void nonzero_words_avx512(const uint64_t *B, size_t nw, F &f) {
    size_t wi = 0;
    for (; wi + 8 <= nw; wi += 8) {
        const __m512i v = _mm512_loadu_si512(B + wi);
        unsigned m = (unsigned)_mm512_test_epi64_mask(v, v);
        while (m) {
            f(wi + (size_t)__builtin_ctz(m));
            m &= m - 1;
        }
    }
    for (; wi < nw; wi++)
        if (B[wi]) f(wi);
}
#endif
// indices of the non-zero 64-bit words of B[0 .. nw), ascending, into idx (room for nw + 16 entries); returns their number.
// Branch-free per 8 words where AVX-512 is there: test -> 8-bit mask -> compress-store of the lane indices.
inline size_t nonzero_index_scalar(const uint64_t *B, size_t nw, uint32_t *idx) {
    size_t c = 0;
    for (size_t wi = 0; wi < nw; wi++) { idx[c] = (uint32_t)wi; c += B[wi] != 0; }
    return c;
}
#ifdef MSIM_X86_HOST
__attribute__((target("avx512f,avx512vl,popcnt"))) size_t nonzero_index_avx512(const uint64_t *B, size_t nw, uint32_t *idx) {
    size_t c = 0, wi = 0;
    __m256i lane = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
    const __m256i step = _mm256_set1_epi32(8);
    for (; wi + 8 <= nw; wi += 8) {
        const __m512i v = _mm512_loadu_si512(B + wi);
        const __mmask8 m = _mm512_test_epi64_mask(v, v);
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(idx + c), _mm256_maskz_compress_epi32(m, lane));   // (compress in the
        c += (size_t)__builtin_popcount((unsigned)m);                    //  register: the store form is microcoded on Zen 4)
        lane = _mm256_add_epi32(lane, step);
    }
    for (; wi < nw; wi++) { idx[c] = (uint32_t)wi; c += B[wi] != 0; }
    return c;
}
#endif
inline size_t nonzero_index(const uint64_t *B, size_t nw, uint32_t *idx) {
#ifdef MSIM_X86_HOST
    static const bool wide = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
    if (wide) return nonzero_index_avx512(B, nw, idx);
#endif
    return nonzero_index_scalar(B, nw, idx);
}

template <class F>
inline void for_each_nonzero_word(const uint64_t *B, size_t nw, F &f) {
#ifdef MSIM_X86_HOST
    static const bool wide = __builtin_cpu_supports("avx512f");
    if (wide) { nonzero_words_avx512(B, nw, f); return; }
#endif
    nonzero_words_scalar(B, nw, f);
}

inline size_t collect_accepted(const uint32_t *words, size_t w, size_t n_words, int sh, uint32_t n, size_t need, uint32_t *buf) {
#ifdef MSIM_X86_HOST
    static const bool wide = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("bmi2");
    if (wide) return collect_accepted_avx512(words, w, n_words, sh, n, need, buf);
#endif
    return collect_accepted_scalar(words, w, n_words, sh, n, need, buf);
}

struct Cand { int64_t pos; int32_t type; int64_t stop; int64_t src = 0; bool rev = false, linked = false; };

inline uint64_t randbelow(HostMT &g, uint64_t n) {      // n < 2^32 on this path
    if (!n) return 0;
    const int sh = 32 - bit_length64(n);
    uint64_t r = g.next() >> sh;
    while (r >= n) r = g.next() >> sh;
    return r;
}
// getrandbits(k), 32 < k <= 64 (CPython _randommodule.c: the low word first, the LAST word shifted down to the bits that are
// left) and _randbelow over it: ranges of 2^32 positions and more
inline uint64_t getrandbits_wide(HostMT &g, int k) {
    const uint64_t lo = g.next();
    const uint64_t hi = g.next() >> (64 - k);
    return lo | (hi << 32);
}
inline uint64_t randbelow_wide(HostMT &g, uint64_t n) { // 2^32 <= n < 2^64
    const int k = bit_length64(n);
    uint64_t r = getrandbits_wide(g, k);
    while (r >= n) r = getrandbits_wide(g, k);
    return r;
}
inline int64_t randint(HostMT &g, int64_t a, int64_t b) { return a + (int64_t)randbelow(g, (uint64_t)(b - a + 1)); }

// sample(range(n), k) as a SORTED list of the selected values.  Only the set matters downstream
// (util.py:104-109 sorts), but the number of words consumed must be exact.
int sample_sorted(Ctx *c, HostMT &g, int64_t n, int64_t k, int64_t setsize, std::vector<int64_t> &out) {
    out.clear();
    if (n < 0) n = 0;                                   // len(range(a, b)) with b < a
    if (k < 0 || k > n)
        return fail(c, MSIM_ERR_VALUE, "Sample larger than population or is negative");
    if (k == 0) return MSIM_OK;
    out.reserve((size_t)k);
    if ((uint64_t)n >= (1ull << 32)) {
        // 2^32 positions and more (no contig of this build is that long -- records hold 32-bit positions -- but the host sampler
        // is also the IT pass's breakpoint sampler, msim_sample_min_distance): CPython's set path over two-word getrandbits.
        // n > setsize always here (setsize <= 21 + 4^ceil(log4(3 k)) with k < 2^31 allocatable entries).
        if (n <= setsize) return fail(c, MSIM_ERR_UNSUPPORTED, "pool-path sample over 2^32 or more positions");
        std::vector<uint64_t> seen, fresh, merged;
        int64_t need = k;
        while (need > 0) {
            fresh.clear();
            for (int64_t i = 0; i < need; i++) fresh.push_back(randbelow_wide(g, (uint64_t)n));   // each consumed whatever it is
            std::sort(fresh.begin(), fresh.end());
            fresh.erase(std::unique(fresh.begin(), fresh.end()), fresh.end());
            merged.clear();
            std::set_union(seen.begin(), seen.end(), fresh.begin(), fresh.end(), std::back_inserter(merged));
            need = k - (int64_t)merged.size();
            seen.swap(merged);
        }
        for (uint64_t v : seen) out.push_back((int64_t)v);
        return MSIM_OK;
    }
    if (n <= setsize) {                                 // pool path: partial Fisher-Yates
        std::vector<int64_t> pool((size_t)n);
        for (int64_t i = 0; i < n; i++) pool[(size_t)i] = i;
        for (int64_t i = 0; i < k; i++) {
            int64_t j = (int64_t)randbelow(g, (uint64_t)(n - i));
            out.push_back(pool[(size_t)j]);
            pool[(size_t)j] = pool[(size_t)(n - i - 1)];
        }
        std::sort(out.begin(), out.end());
        return MSIM_OK;
    }
    // set path: first k distinct accepted draws.  Dense enough -> bitmap (scan gives sorted order
    // for free); sparse -> sort + unique rounds.
    const int sh = 32 - bit_length64((uint64_t)n);
    if ((uint64_t)n <= 512ull * (uint64_t)k + (1u << 20)) {
        std::vector<uint64_t> bits(((size_t)n + 63) / 64, 0);
        int64_t got = 0;
        // The bitmap is far larger than the caches, so the test-and-set is a random DRAM access.  Draw the
        // accepted values in batches and prefetch their words first.  A batch never exceeds k - got: that
        // many more accepted draws are consumed in any case (each adds at most one distinct value), so the
        // stream position stays exact.
        uint32_t batch[64];
        while (got < k) {
            const int want = (int)std::min<int64_t>(64, k - got);
            int nb = 0;
            while (nb < want) {
                const uint64_t v = g.next() >> sh;
                if (v < (uint64_t)n) {
                    batch[nb++] = (uint32_t)v;
                    __builtin_prefetch(&bits[v >> 6], 1, 0);
                }
            }
            for (int i = 0; i < nb; i++) {
                const uint32_t v = batch[i];
                uint64_t &w = bits[v >> 6];
                const uint64_t m = 1ull << (v & 63);
                if (!(w & m)) { w |= m; got++; }
            }
        }
        for (size_t wi = 0; wi < bits.size(); wi++) {
            uint64_t w = bits[wi];
            while (w) { out.push_back((int64_t)(wi * 64 + (size_t)__builtin_ctzll(w))); w &= w - 1; }
        }
        return MSIM_OK;
    }
    std::vector<int64_t> seen;                          // sorted distinct values so far
    std::vector<int64_t> fresh;
    int64_t need = k;
    while (need > 0) {
        fresh.clear();
        // every one of the next `need` accepted draws lies before the cut, duplicate or not
        while ((int64_t)fresh.size() < need) {
            uint64_t v = g.next() >> sh;
            if (v < (uint64_t)n) fresh.push_back((int64_t)v);
        }
        std::sort(fresh.begin(), fresh.end());
        fresh.erase(std::unique(fresh.begin(), fresh.end()), fresh.end());
        std::vector<int64_t> merged;
        merged.reserve(seen.size() + fresh.size());
        std::set_union(seen.begin(), seen.end(), fresh.begin(), fresh.end(), std::back_inserter(merged));
        need = k - (int64_t)merged.size();
        seen.swap(merged);
    }
    out.swap(seen);
    return MSIM_OK;
}

}  // namespace

// util.py:93-109 sample_with_minimum_distance(start, stop, k, d) on the context's CPython stream: sample(range(start, stop -
// (k - 1) * d), k), sorted, the r-th smallest moved up by r * d.  (The IT pass draws its breakpoints this way,
// it_mutator.py:96-119: a quarter of a million per pair of human chromosomes at rate 0.001.)
int sample_min_distance_host(Ctx *c, int64_t start, int64_t stop, int64_t k, int64_t d, int64_t setsize, int64_t *out) {
    std::vector<int64_t> s;
    const int rc = sample_sorted(c, c->py, (stop - (k - 1) * d) - start, k, setsize, s);
    if (rc) return rc;
    for (int64_t r = 0; r < k; r++) out[r] = start + s[(size_t)r] + d * r;
    return MSIM_OK;
}

int chain_boundary_host(Ctx *c, const msim_range &r, uint64_t L, const uint32_t *pos, const uint8_t *type, size_t n,
                        const uint32_t *words, size_t n_words, uint32_t *stop, size_t *consumed, size_t *kept,
                        long long *len_delta) {
    const auto t0 = std::chrono::steady_clock::now();
    const msim_params &P = c->params;
    const int64_t last = (int64_t)L - 1;
    // per type: getrandbits shift + width of randint(pos+min-1, pos+max-1), offset, extent rules
    int sh[8] = {0};
    uint64_t width[8] = {0};
    int64_t add[8] = {0}, blk1[8] = {0};
    for (int t = 1; t <= 7; t++) {
        const int64_t w = r.max_len[t] - r.min_len[t] + 1;
        if (w < 0 || w >= (1ll << 32)) return fail(c, MSIM_ERR_UNSUPPORTED, "randint range of 2^32 or more values");
        width[t] = (uint64_t)w;
        sh[t] = 32 - bit_length64((uint64_t)w);
        add[t] = r.min_len[t] - 1;
        blk1[t] = 1 + P.block[t];
    }
    const int64_t iv_reach = r.max_len[MSIM_IV];
    size_t w = 0, nk = 0;
    // output length change per kept mutation: IN +len, DE -len, DU +len, IV 0   (mutator.py:343-399)
    int64_t dsign[8] = {0};
    dsign[MSIM_IN] = 1; dsign[MSIM_DE] = -1; dsign[MSIM_DU] = 1;
    int64_t delta = 0;
    int64_t blk_hi = 0;                                              // last_mut_range = range(0)
    for (size_t j = 0; j < n; j++) {
        const int64_t p = pos[j];
        if (p < blk_hi) { stop[j] = CHAIN_DROPPED; continue; }       // mutator.py:190-191
        const int t = type[j];
        if (t != MSIM_IN && t != MSIM_DE && t != MSIM_DU && t != MSIM_IV)
            return fail(c, MSIM_ERR_ARG, "chain_boundary_host: type outside IN/DE/DU/IV");
        if (t == MSIM_IV && p + iv_reach >= last) { stop[j] = CHAIN_DROPPED; continue; }   // mutator.py:240-245
        uint64_t v = 0;
        if (width[t]) {                                              // _randbelow_with_getrandbits
            do {
                if (w >= n_words) return fail(c, MSIM_ERR_HIP, "boundary chain: word window overflowed its margin");
                v = words[w++] >> sh[t];
            } while (v >= width[t]);
        }
        int64_t s = p + add[t] + (int64_t)v;
        if ((t == MSIM_DU || t == MSIM_DE) && s > last) s = last;    // mutator.py:253-264
        if (s < 0 || s >= (int64_t)CHAIN_DROPPED) return fail(c, MSIM_ERR_UNSUPPORTED, "mutation extent beyond 2^32");
        stop[j] = (uint32_t)s;
        blk_hi = (t == MSIM_IN ? p : s) + blk1[t];
        delta += dsign[t] * (s - p + 1);
        nk++;
    }
    *consumed = w;
    *kept = nk;
    *len_delta = delta;
    c->t.plan_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSIM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// The same walk over "next accepted draw" tables.  In chain_boundary_host two branches mispredict all the
// time with BASELINE's SV mix (a quarter of the candidates are blocked, a fifth of the words are rejected by
// randint's retry loop), which is most of its ~4.5 ns per candidate.  The device can take the retry loop
// out of the chain: for every word position w of the window and every distinct (shift, width) pair of the
// range's types it finds the first accepted draw at or after w in parallel (plan_kernels.h:
// k_accept_tables) and hands over  T[w][class] = (words consumed) << 24 | value.  The walk is then one
// table load per candidate with a branch-free keep / drop: the loop-carried chain is  w -> T[w] -> w  and
// blk_hi -> dropped -> blk_hi, about 8 cycles.
// append the classes of r's drawable IN/DE/DU/IV/TL lengths to g (at most 8); cls_of[t] = class of type t
static bool chain_classes_add(const msim_range &r, ChainClasses &g, uint8_t cls_of[8]) {
    for (int t = 0; t < 8; t++) cls_of[t] = 0;
    for (int t : {MSIM_IN, MSIM_DE, MSIM_DU, MSIM_IV, MSIM_TL}) {
        bool drawn = false;                                  // can the type draw of this range produce t at all?
        for (int j = 0; j < r.n_types && j < 8; j++) {
            const uint64_t lo = j ? r.cdf_thr[j - 1] : 0;
            if (r.types[j] == t && r.cdf_thr[j] > lo && lo < (1ull << 53)) drawn = true;
        }
        if (!drawn) continue;                                // its length bounds do not matter (class 0 is never looked up)
        const int64_t w = r.max_len[t] - r.min_len[t] + 1;
        if (w < 1 || w >= (1ll << 24)) return false;                 // value field of a table entry: 24 bits (23 beyond four classes: below)
        if (r.min_len[t] < 1 || r.max_len[t] >= (1ll << 30)) return false;   // keeps 0 <= stop < 2^32 - 1 without a check
        const uint32_t sh = (uint32_t)(32 - bit_length64((uint64_t)w));
        uint32_t k = 0;
        while (k < g.n && !(g.sh[k] == sh && g.width[k] == (uint32_t)w)) k++;
        if (k == g.n) {
            if (g.n >= 8) return false;
            g.sh[k] = sh; g.width[k] = (uint32_t)w; g.n++;
        }
        cls_of[t] = (uint8_t)k;
    }
    if (g.n > 4)                                                     // 8 slots per position: increments up to 63 << 3 take a 9th bit
        for (uint32_t k = 0; k < g.n; k++)
            if (g.width[k] >= (1u << chain_value_bits(3))) return false;
    return true;
}

bool chain_classes(const msim_range &r, ChainClasses &cc) {
    cc = ChainClasses{};
    if (!chain_classes_add(r, cc, cc.cls_of)) return false;
    if (!cc.n) { cc.n = 1; cc.sh[0] = 31; cc.width[0] = 1; }     // nothing but SNPs drawn: an unused placeholder class
    return true;
}

// test support (msim_dbg_chain_boundary_tables): the tables the device kernel builds, from tempered words
void accept_tables_host(const ChainClasses &cc, const uint32_t *words, size_t n_words, uint32_t *T) {
    const uint32_t lg = chain_lg_rows(cc);
    for (uint32_t k = 0; k < cc.n; k++) {
        uint32_t nd = 0, nv = 0;
        T[(n_words << lg) + k] = 0;
        for (size_t i = n_words; i-- > 0;) {
            const uint32_t v = words[i] >> cc.sh[k];
            if (v < cc.width[k]) { nd = 1; nv = v; }
            else if (nd && nd < CHAIN_TABLE_REACH) nd++;
            else nd = 0;
            T[(i << lg) + k] = (nd << lg) << chain_value_bits(lg) | (nd ? nv : 0u);
        }
    }
}

int ChainWalk::init(Ctx *c, const msim_range &r, uint64_t L, const ChainClasses &cc, size_t nw) {
    const msim_params &P = c->params;
    const int64_t last = (int64_t)L - 1;
    if (L >= (1ull << 31)) return fail(c, MSIM_ERR_UNSUPPORTED, "table walk: contig of 2^31 or more bases");
    *this = ChainWalk{};
    n_words = nw;
    lg_rows = chain_lg_rows(cc);
    vbits = chain_value_bits(lg_rows);
    for (int t = 0; t < 8; t++) {                          // ids outside the boundary pass: see types_ok
        const bool chain_type = t == MSIM_IN || t == MSIM_DE || t == MSIM_DU || t == MSIM_IV || t == MSIM_TL;
        const int64_t blk1 = 1 + P.block[t];
        add[t] = chain_type ? r.min_len[t] - 1 : 0;
        clamp[t] = (t == MSIM_DU || t == MSIM_DE || t == MSIM_TL) ? last : (int64_t)1 << 40;   // mutator.py:253-264
        drop_from[t] = t == MSIM_IV ? last - r.max_len[MSIM_IV] : INT64_MAX;       // mutator.py:240-245
        // the next blocked end: min(p + next_add + value, next_cap); an insert blocks from its position, whatever it draws
        in_mask[t] = t == MSIM_IN ? 0 : -1;
        next_add[t] = (t == MSIM_IN ? 0 : add[t]) + blk1;
        next_cap[t] = clamp[t] + blk1;
        row[t] = chain_type ? cc.cls_of[t] : 0;
        draw_mask[t] = -1;
    }
    // TLI (run_tl only): no branch of __get_stop_position draws for it, its stop stays 0 and the blocked range it leaves is
    // range(start, 1 + block) (mutator.py:207) -- an ABSOLUTE end: clamp the stop to 0, cap the next blocked end at 1 + block
    // (p + next_add is never below it), and mask the table entry out of the slot increment
    clamp[MSIM_TLI] = 0;
    next_add[MSIM_TLI] = 1 + P.block[MSIM_TLI];
    next_cap[MSIM_TLI] = 1 + P.block[MSIM_TLI];
    draw_mask[MSIM_TLI] = 0;
    return MSIM_OK;
}

bool ChainWalk::types_ok(const uint8_t *type, size_t n, bool with_tl) {
    uint32_t bad_type = 0;
    const uint32_t span = (uint32_t)((with_tl ? MSIM_TLI : MSIM_IV) - MSIM_IN);
    for (size_t i = 0; i < n; i++) bad_type |= (uint32_t)(uint8_t)(type[i] - MSIM_IN) > span;
    static_assert(MSIM_IN == 2 && MSIM_DE == 3 && MSIM_DU == 4 && MSIM_IV == 5 && MSIM_TL == 6 && MSIM_TLI == 7,
                  "boundary types are ids 2..5, translocations 6 and 7");
    return !bad_type;
}

// One candidate: the loop-carried state is (slot, hi); everything else hangs off the candidate index.  On the chain:
// table load -> value -> blocked end -> select, and table load -> increment -> select; both selects hang on one compare
// (x86: a cmp and two cmov -- the compiler turns the portable form into a branch that mispredicts every fourth time).
#define MSIM_CHAIN_STEP()                                                                                      \
    do {                                                                                                       \
        const int64_t p = pos[jj];                                                                             \
        const int t = type[jj] & 7;                                                                            \
        const int64_t pe = p >= drop_from[t] ? -1 : p;                 /* mutator.py:240-245: never kept */    \
        const int64_t a0 = p + next_add[t];                                                                    \
        const int64_t cap = (next_cap[t] & in_mask[t]) | (a0 & ~in_mask[t]);                                   \
        const uint32_t e = (T + row[t])[s_at];                         /* the one load of the chain */         \
        const int64_t d0 = e >> vb;                                    /* slot increment as tabulated */        \
        const int64_t dm = MSIM_CHAIN_TL ? draw_mask[t] : -1;          /* TLI draws nothing */                  \
        const int64_t d = d0 & dm, v = (int64_t)(e & vm) & dm;         /* d: slot increment */                 \
        int64_t nb = a0 + v;                                                                                   \
        nb = nb > cap ? cap : nb;                                                                              \
        const int64_t s_next = s_at + d;                                                                       \
        int64_t s = p + add[t] + v;                                                                            \
        s = s > clamp[t] ? clamp[t] : s;                                                                       \
        int64_t km = -(int64_t)(pe >= hi);                             /* all ones: kept (mutator.py:190-191) */ \
        asm("" : "+r"(km));                                                                                    \
        stop[jj] = (uint32_t)s | (uint32_t)~km;                        /* CHAIN_DROPPED = all ones */          \
        bd |= (d0 - 1) & km & dm;                                      /* kept with d == 0: no accepted draw in reach */ \
        MSIM_CHAIN_SELECT();                                                                                   \
    } while (0)
#if defined(__x86_64__)
#define MSIM_CHAIN_SELECT()                                                                                    \
    asm("cmp %[hi], %[pe]\n\tcmovge %[nb], %[hi]\n\tcmovge %[sn], %[at]"                                       \
        : [hi] "+r"(hi), [at] "+r"(s_at) : [pe] "r"(pe), [nb] "r"(nb), [sn] "r"(s_next) : "cc")
#else
#define MSIM_CHAIN_SELECT() do { hi = (nb & km) | (hi & ~km); s_at = (s_next & km) | (s_at & ~km); } while (0)
#endif

#define MSIM_CHAIN_TL 0
void ChainWalk::run(const uint32_t *pos, const uint8_t *type, size_t n, const uint32_t *T, size_t w_lim, uint32_t *stop) {
    size_t jj = j;
    int64_t s_at = (int64_t)ws, hi = blk_hi, bd = bad;
    const int64_t s_lim = (int64_t)(w_lim << lg_rows);
    const uint32_t vb = vbits, vm = (1u << vbits) - 1u;
    const int64_t RUN = 32, RUN_SLOTS = RUN * ((int64_t)64 << lg_rows);   // a candidate advances at most 63 << lg_rows slots
    while (jj < n && s_at < s_lim) {
        if (jj + RUN <= n && s_at + RUN_SLOTS <= s_lim) {
            for (const size_t je = jj + RUN; jj < je; jj++) MSIM_CHAIN_STEP();
        } else {
            MSIM_CHAIN_STEP();
            jj++;
        }
    }
    j = jj; ws = (size_t)s_at; blk_hi = hi; bad = bd;
}
#undef MSIM_CHAIN_TL
#define MSIM_CHAIN_TL 1
// the same walk for contigs whose ranges draw translocations: TL behaves like DE, TLI draws nothing (draw_mask)
void ChainWalk::run_tl(const uint32_t *pos, const uint8_t *type, size_t n, const uint32_t *T, size_t w_lim, uint32_t *stop) {
    size_t jj = j;
    int64_t s_at = (int64_t)ws, hi = blk_hi, bd = bad;
    const int64_t s_lim = (int64_t)(w_lim << lg_rows);
    const uint32_t vb = vbits, vm = (1u << vbits) - 1u;
    const int64_t RUN = 32, RUN_SLOTS = RUN * ((int64_t)64 << lg_rows);   // a candidate advances at most 63 << lg_rows slots
    while (jj < n && s_at < s_lim) {
        if (jj + RUN <= n && s_at + RUN_SLOTS <= s_lim) {
            for (const size_t je = jj + RUN; jj < je; jj++) MSIM_CHAIN_STEP();
        } else {
            MSIM_CHAIN_STEP();
            jj++;
        }
    }
    j = jj; ws = (size_t)s_at; blk_hi = hi; bad = bd;
}
#undef MSIM_CHAIN_TL
#undef MSIM_CHAIN_STEP
#undef MSIM_CHAIN_SELECT

int ChainWalk::finish(Ctx *c, size_t n, size_t *consumed) const {
    if (bad < 0 || j < n) return fail(c, MSIM_ERR_HIP, "boundary chain: word window overflowed its margin");
    *consumed = ws >> lg_rows;
    return MSIM_OK;
}

int chain_boundary_tables(Ctx *c, const msim_range &r, uint64_t L, const uint32_t *pos, const uint8_t *type, size_t n,
                          const ChainClasses &cc, const uint32_t *T, size_t n_words, uint32_t *stop, size_t *consumed,
                          size_t *kept, long long *len_delta) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!ChainWalk::types_ok(type, n)) return fail(c, MSIM_ERR_ARG, "chain_boundary_tables: type outside IN/DE/DU/IV");
    ChainWalk cw;
    int rc = cw.init(c, r, L, cc, n_words);
    if (rc) return rc;
    cw.run(pos, type, n, T, n_words + 1, stop);          // entry n_words is the end-of-window sentinel
    if ((rc = cw.finish(c, n, consumed))) return rc;
    size_t nk = 0;
    long long delta = 0;                                   // IN +len, DE -len, DU +len, IV 0   (mutator.py:343-399)
    for (size_t i = 0; i < n; i++) {
        if (stop[i] == CHAIN_DROPPED) continue;
        const long long len = (long long)stop[i] - pos[i] + 1;
        delta += type[i] == MSIM_DE ? -len : (type[i] == MSIM_IV ? 0 : len);
        nk++;
    }
    *kept = nk;
    *len_delta = delta;
    c->t.plan_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSIM_OK;
}

// The stream cuts of the same samples, and nothing else: where in the word window does each drawing range start?
// random.sample's set path consumes words until k DISTINCT accepted draws have been seen, so the cut after a range
// needs its duplicates -- a bitmap insert per accepted draw -- but not the sorted sample: the device recomputes the
// accepted draws of [cut[i], cut[i+1]) itself, ORs them into a contig-wide bitmap and expands that into records
// (plan_kernels.h: k_interval_bits, k_walk_expand).  The host leaves out the extraction, which was half of its
// time per position, and hands 4 bytes per range to the device instead of 4 per position.
// Pool-path ranges (n <= setsize: partial Fisher-Yates, util.py / CPython random.sample) are sampled here as before;
// their positions (start + value, any order) go to pool_pos.
int cut_ranges_host(Ctx *c, const msim_range *ranges, int n_ranges, int64_t d, const uint32_t *words, size_t n_words,
                    uint32_t *cut, uint32_t *pool_pos, size_t *n_pool_pos, size_t *consumed, const WordFeed *feed) {
    const auto t0 = std::chrono::steady_clock::now();
    size_t w = 0, at = 0, np = 0;
    size_t avail = feed ? 0 : n_words;                               // words [0, avail) have arrived
    static thread_local std::vector<uint64_t> bits;                 // all zero between ranges
    static thread_local std::vector<uint32_t> accbuf;               // accepted draws of a range (all rounds)
    std::vector<uint32_t> pool;
    int rc_feed = MSIM_OK;
    auto overflow = [&]() {
        std::fill(bits.begin(), bits.end(), 0);
        return rc_feed ? rc_feed : fail(c, MSIM_ERR_HIP, "host sampler: word window overflowed its margin");
    };
    auto more = [&]() {                                              // false: nothing more will come
        if (!feed || avail >= n_words) return false;
        rc_feed = feed->more(feed->user, &avail);
        return rc_feed == MSIM_OK;
    };
    for (int ri = 0; ri < n_ranges; ri++) {
        const msim_range &r = ranges[ri];
        const int64_t k = r.k;
        if (k == 0) continue;
        const int64_t n = (r.stop - (k - 1) * d) - r.start;          // util.py:104
        if (k < 0 || k > n) return fail(c, MSIM_ERR_VALUE, "Sample larger than population or is negative");
        if (n >= (1ll << 32)) return fail(c, MSIM_ERR_UNSUPPORTED, "sampling range of 2^32 or more positions");
        if (w >= (1ull << 32)) return overflow();
        cut[at++] = (uint32_t)w;
        const uint32_t base = (uint32_t)r.start;
        if (n <= r.setsize) {                                        // pool path: partial Fisher-Yates
            pool.resize((size_t)n);
            for (int64_t i = 0; i < n; i++) pool[(size_t)i] = (uint32_t)i;
            for (int64_t i = 0; i < k; i++) {
                const uint64_t m = (uint64_t)(n - i);
                const int sh = 32 - bit_length64(m);
                uint64_t v;
                do {
                    while (w >= avail) if (!more()) return overflow();
                    v = words[w++] >> sh;
                } while (v >= m);
                pool_pos[np++] = base + pool[(size_t)v];
                pool[(size_t)v] = pool[(size_t)(n - i - 1)];
            }
            continue;
        }
        // set path.  Rounds (the rule of the device sampler's tail): the next k - got accepted draws are consumed in any
        // case -- each adds at most one distinct value -- so they are collected in bulk (vectorised filter) and only
        // then meet the bitmap; the stream position stays exact.
        const int sh = 32 - bit_length64((uint64_t)n);
        const size_t nw = ((size_t)n + 63) / 64;
        if (bits.size() < nw + 1) bits.resize(nw + 1, 0);
        uint64_t *B = bits.data();
        const uint32_t nn = (uint32_t)n;
        const bool far = nw > 4096;                                  // bitmap beyond L1: prefetch ahead of the inserts
        const bool sweep = nw <= 32768;                              // clear by memset (256 KB); larger ones draw by draw
        int64_t got = 0;
        size_t held = 0;                                             // accepted draws kept for the clearing pass
        while (got < k) {
            const size_t need = (size_t)(k - got);
            if (accbuf.size() < held + need + 32) accbuf.resize(held + need + need / 2 + 1024);
            uint32_t *buf = accbuf.data() + held;
            size_t w2;
            while ((w2 = collect_accepted(words, w, avail, sh, nn, need, buf)) == SIZE_MAX)
                if (!more()) return overflow();
            w = w2;
            if (far) {
                constexpr size_t AHEAD = 64;
                for (size_t i = 0; i < std::min(AHEAD, need); i++) __builtin_prefetch(&B[buf[i] >> 6], 1, 3);
                for (size_t i = 0; i < need; i++) {
                    if (i + AHEAD < need) __builtin_prefetch(&B[buf[i + AHEAD] >> 6], 1, 3);
                    const uint32_t v = buf[i];
                    const uint64_t m = 1ull << (v & 63);
                    const uint64_t x = B[v >> 6];
                    got += (int64_t)!(x & m);
                    B[v >> 6] = x | m;
                }
            } else {
                for (size_t i = 0; i < need; i++) {
                    const uint32_t v = buf[i];
                    const uint64_t m = 1ull << (v & 63);
                    const uint64_t x = B[v >> 6];
                    got += (int64_t)!(x & m);
                    B[v >> 6] = x | m;
                }
            }
            if (!sweep) held += need;
        }
        if (sweep) memset(B, 0, nw * 8);
        else for (size_t i = 0; i < held; i++) B[accbuf[i] >> 6] = 0;
    }
    if (w >= (1ull << 32)) return overflow();
    cut[at] = (uint32_t)w;
    *n_pool_pos = np;
    *consumed = w;
    c->t.plan_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSIM_OK;
}

// random.sample() of ONE drawing range (util.py:94-109) over a window of tempered words: the sorted positions
// start + value + d * rank go to pos_out[0 .. k), *w_io advances by exactly the words CPython consumes.  `more`
// is asked for further words when the window (words [0, *avail)) runs out; it returns false when none will come.
// Returns 0, 1 (window exhausted) or -1 / -2 (the reference's ValueError / an unsupported size).
#ifdef MSIM_SAMPLE_PROF
static unsigned long long g_sp[8];
#define SP_T0() unsigned long long sp_t = __builtin_ia32_rdtsc()
#define SP_LAP(i) do { const unsigned long long n_ = __builtin_ia32_rdtsc(); g_sp[i] += n_ - sp_t; sp_t = n_; } while (0)
#else
#define SP_T0() do {} while (0)
#define SP_LAP(i) do {} while (0)
#endif
struct NoCutHook { void operator()(size_t) const {} };
template <class More, class CutHook = NoCutHook>
static inline int sample_one_range(const msim_range &r, int64_t d, const uint32_t *words, size_t *w_io, size_t *avail,
                                   More &&more, uint32_t *pos_out, CutHook &&cut_known = CutHook()) {
    static thread_local std::vector<uint64_t> bits;                 // all zero between ranges (cleared while it is scanned)
    static thread_local std::vector<uint32_t> accbuf;               // accepted draws of one round
    static thread_local std::vector<uint32_t> pool, picked;
    size_t w = *w_io, at = 0;
    SP_T0();
    const int64_t k = r.k;
    const int64_t n = (r.stop - (k - 1) * d) - r.start;              // util.py:104
    if (k < 0 || k > n) return -1;
    if (n >= (1ll << 32)) return -2;
    const uint32_t base = (uint32_t)r.start, dd = (uint32_t)d;
    auto fail_clean = [&]() { std::fill(bits.begin(), bits.end(), 0); return 1; };   // keep the scratch bitmap clean
    if (n <= r.setsize) {                                            // pool path: partial Fisher-Yates
        pool.resize((size_t)n);
        for (int64_t i = 0; i < n; i++) pool[(size_t)i] = (uint32_t)i;
        picked.clear();
        for (int64_t i = 0; i < k; i++) {
            const uint64_t m = (uint64_t)(n - i);
            const int sh = 32 - bit_length64(m);
            uint64_t v;
            do {
                while (w >= *avail) if (!more()) return 1;
                v = words[w++] >> sh;
            } while (v >= m);
            picked.push_back(pool[(size_t)v]);
            pool[(size_t)v] = pool[(size_t)(n - i - 1)];
        }
        cut_known(w);
        std::sort(picked.begin(), picked.end());
        for (int64_t i = 0; i < k; i++) pos_out[at++] = base + picked[(size_t)i] + dd * (uint32_t)i;
        *w_io = w;
        SP_LAP(0);
        return 0;
    }
    const int sh = 32 - bit_length64((uint64_t)n);
    const size_t nw = ((size_t)n + 63) / 64;
    if (bits.size() < nw + 1) bits.resize(nw + 1, 0);
    uint64_t *B = bits.data();
    int64_t got = 0;
    {
        // The data-dependent branches of the obvious loop (draw rejected? bitmap word empty?) mispredict about every
        // second time, so the loop is split: accepted draws are collected in bulk without branches, inserted without
        // branches (prefetched ahead when the bitmap is beyond L1), and the extraction runs over the list of non-empty
        // words (found eight at a time), writing three slots unconditionally per word.
        const uint32_t nn = (uint32_t)n;
        // Rounds (the rule of the device sampler's tail): the next k - got accepted draws are consumed in any
        // case -- each adds at most one distinct value -- so they are collected in bulk (vectorised filter) and
        // only then meet the bitmap; the stream position stays exact.
        if (accbuf.size() < (size_t)k + 32) accbuf.resize((size_t)k + (size_t)k / 4 + 64);
        uint32_t *buf = accbuf.data();
        const bool far = nw > 4096;                                  // bitmap beyond L1: prefetch ahead of the inserts
        SP_LAP(1);
        while (got < k) {
            const size_t need = (size_t)(k - got);
            size_t w2;
            while ((w2 = collect_accepted(words, w, *avail, sh, nn, need, buf)) == SIZE_MAX)
                if (!more()) return fail_clean();
            w = w2;
            SP_LAP(2);
            if (far) {
                constexpr size_t AHEAD = 64;
                for (size_t i = 0; i < std::min(AHEAD, need); i++) __builtin_prefetch(&B[buf[i] >> 6], 1, 3);
                for (size_t i = 0; i < need; i++) {
                    if (i + AHEAD < need) __builtin_prefetch(&B[buf[i + AHEAD] >> 6], 1, 3);
                    const uint32_t v = buf[i];
                    const uint64_t m = 1ull << (v & 63);
                    const uint64_t x = B[v >> 6];
                    got += (int64_t)!(x & m);
                    B[v >> 6] = x | m;
                }
            } else {
                for (size_t i = 0; i < need; i++) {
                    const uint32_t v = buf[i];
                    const size_t wi = v >> 6;
                    const uint64_t m = 1ull << (v & 63);
                    const uint64_t x = B[wi];
                    got += (int64_t)!(x & m);
                    B[wi] = x | m;
                }
            }
            SP_LAP(3);
        }
        cut_known(w);                                                // (the sample's end in the stream is fixed from here on)
        // extraction in two passes, so that no branch depends on the bitmap's content: the non-empty words' indices first
        // (8 words per step), then a loop of known length over them -- three slots written unconditionally per word
        // (slots beyond its popcount are overwritten by the next word; words with more than three bits are rare)
        static thread_local std::vector<uint32_t> nzidx;
        if (nzidx.size() < nw + 16) nzidx.resize(nw + nw / 2 + 64);
        uint32_t *idx = nzidx.data();
        const size_t nz = nonzero_index(B, nw, idx);
        SP_LAP(5);
        // two slots per word unconditionally (a word holds one or two positions 97 % of the time at the reference's
        // rates), the rest in a rarely taken branch; d * rank is carried along instead of multiplied
        uint32_t drank = 0;
        const size_t safe = (size_t)k >= 2 ? (size_t)k - 2 : 0;
        for (size_t j = 0; j < nz; j++) {
            if (far) __builtin_prefetch(&B[idx[j + 16]], 0, 3);       // (idx has slack behind nz)
            const size_t wi = idx[j];
            uint64_t x = B[wi];
            const uint32_t cn = (uint32_t)__builtin_popcountll(x);
            const uint32_t p0 = base + (uint32_t)(wi * 64) + drank;
            if (__builtin_expect(cn <= 2 && at <= safe, 1)) {
                const uint64_t x1 = x & (x - 1), top = 1ull << 63;
                pos_out[at] = p0 + (uint32_t)__builtin_ctzll(x);
                pos_out[at + 1] = p0 + dd + (uint32_t)__builtin_ctzll(x1 | top);
            } else {
                uint32_t q = 0;
                while (x) { pos_out[at + q] = p0 + dd * q + (uint32_t)__builtin_ctzll(x); q++; x &= x - 1; }
            }
            at += cn;
            drank += dd * cn;
        }
        memset(B, 0, nw * 8);                                        // (cheaper than a store per word inside the loop)
        *w_io = w;
        SP_LAP(4);
        return 0;
    }
}

static int sample_one_error(Ctx *c, int code, const char *who) {
    if (code == -1) return fail(c, MSIM_ERR_VALUE, "Sample larger than population or is negative");
    if (code == -2) return fail(c, MSIM_ERR_UNSUPPORTED, "sampling range of 2^32 or more positions");
    return fail(c, MSIM_ERR_HIP, std::string(who) + ": word window overflowed its margin");
}

int sample_ranges_host(Ctx *c, const msim_range *ranges, int n_ranges, int64_t d, const uint32_t *words,
                       size_t n_words, uint32_t *pos_out, size_t *consumed) {
    const auto t0 = std::chrono::steady_clock::now();
    size_t w = 0, at = 0, avail = n_words;
    auto none = []() { return false; };
    for (int ri = 0; ri < n_ranges; ri++) {
        const msim_range &r = ranges[ri];
        if (r.k == 0) continue;
        const int code = sample_one_range(r, d, words, &w, &avail, none, pos_out + at);
        if (code) return sample_one_error(c, code, "host sampler");
        at += (size_t)r.k;
    }
    *consumed = w;
    c->t.plan_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSIM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// General host-chain engine (ctx.h).  Per drawing range, in stream order: sample() -> sorted positions; the
// candidates on the chain (types known by ordinal) walk the boundary pass over the accept tables; the stream
// position after the walk is where the next range's sample starts.
static bool type_drawable(const msim_range &r, int j) {
    const uint64_t lo = j ? r.cdf_thr[j - 1] : 0;
    return r.cdf_thr[j] > lo && lo < (1ull << 53);
}

bool multimix_prepare(const Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges, MixSets &ms) {
    ms = MixSets{};
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);
    if (L >= (1ull << 31)) return false;                              // ChainWalk's clamp arithmetic
    for (int t = 1; t <= 7; t++)
        if (P.block[t] >= (1ll << 31)) return false;
    ms.sn_chained = P.block[MSIM_SN] != d;
    std::unordered_map<std::string, uint32_t> seen;
    int64_t prev_stop = -1;
    double pool_bound = 0;
    constexpr size_t KEY_OFF = offsetof(msim_range, n_types);
    for (int i = 0; i < n_ranges; i++) {
        const msim_range &r = ranges[i];
        if (r.k == 0) continue;                                       // draws nothing (mutator.py:163-164)
        const int64_t n = (r.stop - (r.k - 1) * d) - r.start;
        if (r.k < 0 || n < r.k || n >= (1ll << 32)) return false;     // ValueError / multi-word getrandbits: host planner
        if (r.start <= prev_stop || r.start < 0 || r.stop > (int64_t)L) return false;   // overlapping / unsorted / outside
        prev_stop = r.stop;
        if (r.n_types < 1 || r.n_types > 8) return false;
        const std::string key(reinterpret_cast<const char *>(&r) + KEY_OFF, sizeof(msim_range) - KEY_OFF);
        auto it = seen.find(key);
        uint32_t id;
        if (it == seen.end()) {
            ChainClasses own{};
            for (int j = 0; j < r.n_types; j++) {
                if (!type_drawable(r, j)) continue;
                const int t = r.types[j];
                if (t < MSIM_SN || t > MSIM_TLI) return false;
                if (t == MSIM_TL || t == MSIM_TLI) ms.has_tl = true;   // translocations: linked after the last range (below)
            }
            if (!chain_classes_add(r, ms.gcc, own.cls_of)) return false;
            id = (uint32_t)ms.rep.size();
            ms.rep.push_back(i);
            ms.cc.push_back(own);
            seen.emplace(key, id);
        } else id = it->second;
        ms.set_of.push_back(id);
        ms.K += (uint64_t)r.k;
        ms.n_draw++;
        for (int j = 0; j < r.n_types; j++)
            if (r.types[j] == MSIM_IN && type_drawable(r, j)) pool_bound += (double)r.k * (double)r.max_len[MSIM_IN];
    }
    if (ms.K == 0 || ms.K >= (1ull << 31) || pool_bound >= 4.0e9) return false;
    if (!ms.gcc.n) { ms.gcc.n = 1; ms.gcc.sh[0] = 31; ms.gcc.width[0] = 1; }   // SNPs only: an unused placeholder class
    for (auto &cc : ms.cc) {                                          // every set looks the union's classes up
        cc.n = ms.gcc.n;
        for (uint32_t k = 0; k < 8; k++) { cc.sh[k] = ms.gcc.sh[k]; cc.width[k] = ms.gcc.width[k]; }
    }
    return true;
}

namespace {
// A Python list under `del lst[i]` by index, without moving its tail: Fenwick tree over "still there" flags.
// select(i) = position of the i-th remaining element; erase(pos) removes it.  (__fix_tl_amount, mutator.py:286-302)
struct IndexedList {
    std::vector<int32_t> tree;
    size_t n = 0, alive = 0, top = 1;
    explicit IndexedList(size_t size) : tree(size + 1, 0), n(size), alive(size) {
        for (size_t i = 1; i <= n; i++) { tree[i] += 1; const size_t j = i + (i & (~i + 1)); if (j <= n) tree[j] += tree[i]; }
        while (top * 2 <= n) top *= 2;
    }
    size_t select(size_t i) const {                                   // 0-based rank among the remaining -> 0-based position
        size_t pos = 0, rem = i + 1;
        for (size_t step = top; step; step >>= 1)
            if (pos + step <= n && (size_t)tree[pos + step] < rem) { pos += step; rem -= (size_t)tree[pos]; }
        return pos;
    }
    void erase(size_t pos) { for (size_t i = pos + 1; i <= n; i += i & (~i + 1)) tree[i] -= 1; alive--; }
};
}  // namespace

// The kept TL spans and TLI sites among the chain's candidates: their numbers, then their indices in chain order
// (tls / tlis: room for 16 entries more than counted).
void count_translocations(const uint8_t *type, const uint32_t *stop, size_t n, size_t *n_tl, size_t *n_tli) {
    size_t a = 0, b = 0;
    for (size_t q = 0; q < n; q++) {
        const unsigned kept = stop[q] != CHAIN_DROPPED, t = type[q] & 7u;
        a += kept & (t == (unsigned)MSIM_TL);
        b += kept & (t == (unsigned)MSIM_TLI);
    }
    *n_tl = a;
    *n_tli = b;
}
static void collect_translocations_scalar(const uint8_t *type, const uint32_t *stop, size_t q, size_t n, uint32_t *tls, size_t a,
                                          uint32_t *tlis, size_t b) {
    for (; q < n; q++) {
        const unsigned kept = stop[q] != CHAIN_DROPPED, t = type[q] & 7u;
        tls[a] = (uint32_t)q;
        a += kept & (t == (unsigned)MSIM_TL);
        tlis[b] = (uint32_t)q;
        b += kept & (t == (unsigned)MSIM_TLI);
    }
}
#ifdef MSIM_X86_HOST
__attribute__((target("avx512f,avx512bw,avx512vl,popcnt"))) static void collect_translocations_avx512(
    const uint8_t *type, const uint32_t *stop, size_t n, uint32_t *tls, uint32_t *tlis) {
    size_t a = 0, b = 0, q = 0;
    __m512i lane = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const __m512i step = _mm512_set1_epi32(16), dropped = _mm512_set1_epi32((int)CHAIN_DROPPED), seven = _mm512_set1_epi32(7);
    const __m512i v_tl = _mm512_set1_epi32(MSIM_TL), v_tli = _mm512_set1_epi32(MSIM_TLI);
    for (; q + 16 <= n; q += 16) {
        const __m512i t = _mm512_and_si512(_mm512_cvtepu8_epi32(_mm_loadu_si128(reinterpret_cast<const __m128i *>(type + q))), seven);
        const __mmask16 kept = _mm512_cmpneq_epu32_mask(_mm512_loadu_si512(stop + q), dropped);
        const __mmask16 m_tl = _mm512_mask_cmpeq_epu32_mask(kept, t, v_tl), m_tli = _mm512_mask_cmpeq_epu32_mask(kept, t, v_tli);
        _mm512_storeu_si512(tls + a, _mm512_maskz_compress_epi32(m_tl, lane));      // (compress in the register: see collect_accepted)
        _mm512_storeu_si512(tlis + b, _mm512_maskz_compress_epi32(m_tli, lane));
        a += (size_t)__builtin_popcount((unsigned)m_tl);
        b += (size_t)__builtin_popcount((unsigned)m_tli);
        lane = _mm512_add_epi32(lane, step);
    }
    collect_translocations_scalar(type, stop, q, n, tls, a, tlis, b);
}
#endif
static void collect_translocations(const uint8_t *type, const uint32_t *stop, size_t n, uint32_t *tls, uint32_t *tlis) {
#ifdef MSIM_X86_HOST
    static const bool wide = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    if (wide) { collect_translocations_avx512(type, stop, n, tls, tlis); return; }
#endif
    collect_translocations_scalar(type, stop, 0, n, tls, 0, tlis, 0);
}

// __link_tls (mutator.py:130-131, 267-316) over the chain's candidates of one contig, in order: the kept TL spans and TLI
// sites; __fix_tl_amount deletes random surplus entries (they stay in ch_stop -- they blocked -- and get CHAIN_TOMBSTONE),
// shuffle(tls), one coin per pair.  src: the tempered CPython-stream words that follow -- words[w .. avail) are there,
// more() makes avail grow (false: the window is exhausted).
template <class More>
struct LinkWords { const uint32_t *words; size_t w; const size_t &avail; More more; };
template <class Src>
static bool link_translocations_impl(const uint32_t *ch_pos, const uint8_t *ch_type, uint32_t *ch_stop, uint32_t *ch_extra,
                                     uint8_t *ch_aux, size_t n_ch, Src &src) {
    auto next_word = [&](uint32_t &out) -> bool {
        while (src.w >= src.avail) if (!src.more()) return false;
        out = src.words[src.w++];
        return true;
    };
    static const bool prof = getenv("MSIM_WALK_PROF") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    double t_ph[4] = {0, 0, 0, 0};
    auto lap = [&](int ph) { if (prof) { const auto n = std::chrono::steady_clock::now(); t_ph[ph] += std::chrono::duration<double, std::micro>(n - tp).count(); tp = n; } };
    size_t n_tl = 0, n_tli = 0;
    count_translocations(ch_type, ch_stop, n_ch, &n_tl, &n_tli);
    std::vector<uint32_t> tls(n_tl + 16), tlis(n_tli + 16);           // (16: the collectors store whole vectors)
    collect_translocations(ch_type, ch_stop, n_ch, tls.data(), tlis.data());
    tls.resize(n_tl);
    tlis.resize(n_tli);
    auto randbelow_w = [&](uint64_t n, uint64_t &out) -> bool {      // Lib/random.py _randbelow_with_getrandbits
        if (!n) { out = 0; return true; }
        const int sh = 32 - bit_length64(n);
        uint64_t v;
        do {
            uint32_t word;
            if (!next_word(word)) return false;
            v = word >> sh;
        } while (v >= n);
        out = v;
        return true;
    };
    if (tls.empty()) {
        for (uint32_t q : tlis) { ch_extra[q] = ch_pos[q]; ch_stop[q] = 0; }   // `if tls:` is false: start = pos, stop = 0 stay
        return true;
    }
    lap(0);
    if (tls.size() != tlis.size()) {
        std::vector<uint32_t> &longer = tls.size() > tlis.size() ? tls : tlis;
        const size_t want = std::min(tls.size(), tlis.size());
        IndexedList lst(longer.size());
        while (lst.alive > want) {
            uint64_t idx;
            if (!randbelow_w(lst.alive, idx)) return false;              // randint(0, len - 1)
            const size_t at = lst.select((size_t)idx);
            ch_aux[longer[at]] |= CHAIN_TOMBSTONE;                          // del muts[...]: no record -- but it blocked
            lst.erase(at);
            longer[at] = 0xffffffffu;
        }
        longer.erase(std::remove(longer.begin(), longer.end(), 0xffffffffu), longer.end());
    }
    lap(1);
    for (size_t i = tls.size(); i > 1;) {                             // random.shuffle(tls): for i = n-1 .. 1: swap(i, randbelow(i + 1))
        uint32_t js[32];                                              // (the draws of a batch first: their targets are random cache lines)
        const size_t nb = std::min<size_t>(32, i - 1);
        for (size_t b = 0; b < nb; b++) {
            uint64_t j;
            if (!randbelow_w((uint64_t)(i - b), j)) return false;
            js[b] = (uint32_t)j;
            __builtin_prefetch(&tls[(size_t)j], 1);
        }
        for (size_t b = 0; b < nb; b++) std::swap(tls[i - 1 - b], tls[js[b]]);
        i -= nb;
    }
    lap(2);
    // __transloc_invert draws its coin first and looks at the length afterwards: randint(0, 1) = getrandbits(2) until < 2 -- a word
    // is accepted when its top bit is clear, the coin is the bit below.  All coins first, branch-free.
    std::vector<uint8_t> coins(tls.size() + 1);
    for (size_t got = 0; got < tls.size();) {
        while (src.w >= src.avail) if (!src.more()) return false;
        const size_t e = src.avail;
        size_t w = src.w;
        while (w < e && got < tls.size()) {
            const uint32_t word = src.words[w++];
            coins[got] = (uint8_t)((word >> 30) & 1u);
            got += (word >> 31) ^ 1u;
        }
        src.w = w;
    }
    for (size_t i = 0; i < tls.size(); i++) {
        if (i + 24 < tls.size()) { __builtin_prefetch(ch_pos + tls[i + 24]); __builtin_prefetch(ch_stop + tls[i + 24]); }
        const uint32_t tl = tls[i], tli = tlis[i];
        const unsigned coin = coins[i];
        const int64_t tlen = (int64_t)ch_stop[tl] + 1 - (int64_t)ch_pos[tl];
        const bool rev = !(coin == 0 || tlen < 2);
        ch_extra[tli] = ch_pos[tl];                                   // Mutation(TLI, tl_pos, muts[tl_pos].stop, rev, tli_pos)
        ch_stop[tli] = ch_stop[tl];
        ch_aux[tli] = (uint8_t)((ch_aux[tli] & CHAIN_TOMBSTONE) | (rev ? 1 : 0) | (ch_pos[tli] > 0 ? 2 : 0));
    }
    lap(3);
    if (prof) fprintf(stderr, "link: %zu pairs | collect %.0f fix %.0f shuffle %.0f pair %.0f us\n", tls.size(), t_ph[0], t_ph[1], t_ph[2], t_ph[3]);
    return true;
}

int link_translocations(Ctx *c, const uint32_t *words, size_t n_words, const uint32_t *ch_pos, const uint8_t *ch_type,
                        uint32_t *ch_stop, uint32_t *ch_extra, uint8_t *ch_aux, size_t n_ch, size_t *consumed) {
    const auto t0 = std::chrono::steady_clock::now();
    auto none = []() { return false; };
    const size_t avail = n_words;
    LinkWords<decltype(none)> src{words, 0, avail, none};
    memset(ch_extra, 0, n_ch * sizeof(uint32_t));
    memset(ch_aux, 0, n_ch);
    if (!link_translocations_impl(ch_pos, ch_type, ch_stop, ch_extra, ch_aux, n_ch, src))
        return fail(c, MSIM_ERR_HIP, "translocation linking: word window overflowed its margin");
    *consumed = src.w;
    c->t.plan_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSIM_OK;
}

int multimix_walk_host(Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges, int64_t d, const MixSets &ms,
                       const uint32_t *words, const uint32_t *T, size_t n_words, const uint32_t *ch_rank,
                       const uint8_t *ch_type, size_t n_ch, uint32_t *cand_pos, uint32_t *ch_stop, uint32_t *visit_from,
                       size_t *consumed, const WordFeed *feed, uint32_t *ch_extra, uint8_t *ch_aux) {
    const auto t0 = std::chrono::steady_clock::now();
    const msim_params &P = c->params;
    const bool generic = ms.sn_chained;                              // the plain loop: SNPs are on the chain as well
    if (ms.has_tl && (!ch_extra || !ch_aux)) return fail(c, MSIM_ERR_ARG, "host chain: translocations need ch_extra / ch_aux");
    if (generic) {
        uint32_t bad = 0;
        const uint32_t lo = ms.sn_chained ? MSIM_SN : MSIM_IN, hi_t = ms.has_tl ? MSIM_TLI : MSIM_IV;
        for (size_t i = 0; i < n_ch; i++) bad |= (uint32_t)(uint8_t)(ch_type[i] - lo) > hi_t - lo;
        if (bad) return fail(c, MSIM_ERR_HIP, "host chain: candidate type outside what its ranges draw");
    } else if (!ChainWalk::types_ok(ch_type, n_ch, ms.has_tl)) return fail(c, MSIM_ERR_HIP, "host chain: candidate type outside what its ranges draw");
    std::vector<ChainWalk> proto(ms.rep.size());
    for (size_t s = 0; s < proto.size(); s++) {
        const int rc = proto[s].init(c, ranges[ms.rep[s]], L, ms.cc[s], n_words);
        if (rc) return rc;
    }
    const uint32_t lg = chain_lg_rows(ms.gcc);
    size_t w = 0, avail = feed ? 0 : n_words;                        // words [0, avail) have arrived; so has T[.. avail)
    int rc_feed = MSIM_OK;
    auto more = [&]() {                                              // false: nothing more will come
        if (!feed || avail >= n_words) return false;
        rc_feed = feed->more(feed->user, &avail);
        return rc_feed == MSIM_OK;
    };
    auto overflow = [&]() { return rc_feed ? rc_feed : fail(c, MSIM_ERR_HIP, "host chain: word window overflowed its margin"); };
    auto t_lim = [&]() { return avail + (avail >= n_words ? 1u : 0u); };   // entry n_words is the end-of-window sentinel
    static thread_local std::vector<uint32_t> pbuf;
    std::vector<uint32_t> ch_pos;                                    // translocations: every chain candidate's position, and
    std::vector<size_t> span;                                        //   each range's slice of the chain, for the passes behind
    if (ms.has_tl) { ch_pos.resize(n_ch); span.reserve((size_t)ms.n_draw + 1); }
    size_t qa = 0, di = 0;
    uint64_t base = 0;
    uint32_t vf = 0;                                                 // consumed_to + 1 of __mutate_sequence's walk
    int64_t bad_acc = 0;
    static const bool prof = getenv("MSIM_WALK_PROF") != nullptr;
    double t_sample = 0, t_gather = 0, t_chain = 0;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](double &acc) { if (prof) { const auto n = std::chrono::steady_clock::now(); acc += std::chrono::duration<double, std::micro>(n - tp).count(); tp = n; } };
    for (int ri = 0; ri < n_ranges; ri++) {
        const msim_range &r = ranges[ri];
        if (r.k == 0) continue;
        const uint64_t k = (uint64_t)r.k;
        lap(t_chain);
        const int code = sample_one_range(r, d, words, &w, &avail, more, cand_pos + base, [&](size_t w_end) {
            // the boundary pass reads the tables from here on, in a burst of a few hundred bytes that the DMA engine wrote
            // moments ago (nothing of it is in a cache) -- fetch it while the sample is being extracted
            const char *t = reinterpret_cast<const char *>(T + (w_end << lg));
            for (int q = 0; q < 8; q++) __builtin_prefetch(t + 64 * q, 0, 3);
        });
        if (code) return rc_feed ? rc_feed : sample_one_error(c, code, "host chain");
        lap(t_sample);
        size_t qb = qa;
        while (qb < n_ch && ch_rank[qb] < base + k) qb++;
        const size_t m = qb - qa;
        if (pbuf.size() < m + 8) pbuf.resize(m + m / 2 + 64);
        uint32_t *pb = pbuf.data();
        for (size_t x = 0; x < m; x++) pb[x] = cand_pos[ch_rank[qa + x]];
        lap(t_gather);
        ChainWalk &pw = proto[ms.set_of[di]];
        if (ms.has_tl) {
            span.push_back(qa);
            for (size_t x = 0; x < m; x++) { ch_pos[qa + x] = pb[x]; ch_aux[qa + x] = 0; ch_extra[qa + x] = 0; }
        }
        if (m && !generic) {
            ChainWalk &cw = pw;                                      // (the set's walker: only its position is per range)
            cw.j = 0; cw.blk_hi = 0; cw.bad = 0;                     // last_mut_range = range(0), per range (mutator.py:184)
            cw.ws = w << lg;
            for (;;) {
                if (ms.has_tl) cw.run_tl(pb, ch_type + qa, m, T, t_lim(), ch_stop + qa);
                else cw.run(pb, ch_type + qa, m, T, t_lim(), ch_stop + qa);
                if (cw.j >= m) break;
                if (!more()) return overflow();
            }
            bad_acc |= cw.bad;
            w = cw.ws >> lg;
        } else if (m) {                                              // SNPs block too (mutator.py:204-206): every candidate chains
            int64_t hi = 0;                                          // last_mut_range = range(0), per range (mutator.py:184)
            for (size_t x = 0; x < m; x++) {
                const int64_t p = pb[x];
                const int t = ch_type[qa + x] & 7;
                if (p < hi) { ch_stop[qa + x] = CHAIN_DROPPED; continue; }              // mutator.py:190-191
                if (t == MSIM_SN) { ch_stop[qa + x] = (uint32_t)p; hi = p + 1 + P.block[MSIM_SN]; continue; }
                if (t == MSIM_TLI) {                                  // no branch of __get_stop_position: stop stays 0, so the
                    ch_stop[qa + x] = 0;                              // blocked range is range(start, 1 + block) (mutator.py:207)
                    hi = 1 + P.block[MSIM_TLI];
                    continue;
                }
                if (p >= pw.drop_from[t]) { ch_stop[qa + x] = CHAIN_DROPPED; continue; }   // mutator.py:240-245
                while (w >= t_lim()) if (!more()) return overflow();
                const uint32_t e = T[(w << lg) + pw.row[t]];
                const uint32_t inc = e >> pw.vbits;
                if (!inc) return overflow();                         // no accepted draw in reach: the window ends here
                int64_t s = p + pw.add[t] + (int64_t)(e & ((1u << pw.vbits) - 1u));
                s = s > pw.clamp[t] ? pw.clamp[t] : s;
                ch_stop[qa + x] = (uint32_t)s;
                hi = (t == MSIM_IN ? p : s) + 1 + P.block[t];
                w += inc >> lg;
            }
        }
        visit_from[di] = vf;
        for (size_t x = m; !ms.has_tl && x-- > 0;) {                 // the range's last kept DE / DU / IV: does it get visited?
            const int t = ch_type[qa + x] & 7;
            if (ch_stop[qa + x] == CHAIN_DROPPED || (t != MSIM_DE && t != MSIM_DU && t != MSIM_IV)) continue;
            if (pb[x] >= vf) vf = ch_stop[qa + x] + 1;               // pos = muts[pos].stop (mutator.py:376,386,398)
            break;
        }
        qa = qb;
        base += k;
        di++;
    }
    if (bad_acc < 0) return overflow();
    if (ms.has_tl) {
        // ---- __link_tls (mutator.py:130-131, 267-316), behind the last range: the kept TL spans and TLI sites of the whole
        // contig, in range order; __fix_tl_amount deletes random surplus entries, shuffle(tls), one coin per pair
        span.push_back(n_ch);
        auto step = [&]() -> bool { return more(); };                 // (raises `avail`, which the source sees by reference)
        LinkWords<decltype(step)> src{words, w, avail, step};
        const bool linked = link_translocations_impl(ch_pos.data(), ch_type, ch_stop, ch_extra, ch_aux, n_ch, src);
        w = src.w;
        if (!linked) return overflow();
        // the visit filter, now that it is known which TL spans are records at all
        uint32_t v2 = 0;
        for (size_t r2 = 0; r2 + 1 < span.size(); r2++) {
            visit_from[r2] = v2;
            for (size_t q = span[r2 + 1]; q-- > span[r2];) {
                const int t = ch_type[q] & 7;
                if (ch_stop[q] == CHAIN_DROPPED || (ch_aux[q] & CHAIN_TOMBSTONE) ||
                    (t != MSIM_DE && t != MSIM_DU && t != MSIM_IV && t != MSIM_TL)) continue;
                if (ch_pos[q] >= v2) v2 = ch_stop[q] + 1;
                break;
            }
        }
    }
    lap(t_chain);
    if (prof) fprintf(stderr, "walk: K %llu chain %zu ranges %zu | sample %.0f gather %.0f chain+visit %.0f us\n", (unsigned long long)ms.K, n_ch, di, t_sample, t_gather, t_chain);
#ifdef MSIM_SAMPLE_PROF
    if (prof) { fprintf(stderr, "  sample cycles: pool %llu setup %llu collect %llu insert %llu nzidx %llu extract %llu\n", g_sp[0], g_sp[1], g_sp[2], g_sp[3], g_sp[5], g_sp[4]); memset(g_sp, 0, sizeof g_sp); }
#endif
    *consumed = w;
    c->t.plan_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSIM_OK;
}

// Test support: the engine end to end on the host.  What the device does in plan_contig_gpu_multimix is restated
// sequentially here (types by ordinal, accept tables, clipped running maximum for the SNP filter, visit filter,
// records) -- same algorithm, no kernels -- so the CPU tier can compare it with plan_contig_host.
int multimix_plan_emulated(Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges, HostPlan &out) {
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);
    MixSets ms;
    if (!multimix_prepare(c, L, ranges, n_ranges, ms)) return fail(c, MSIM_ERR_UNSUPPORTED, "outside the host-chain engine");
    const size_t K = (size_t)ms.K;
    std::vector<uint8_t> ty(K);
    std::vector<uint32_t> rng_of(K), clip, ch_rank;
    std::vector<uint8_t> ch_type;
    {
        size_t j = 0;
        uint32_t di = 0;
        for (int ri = 0; ri < n_ranges; ri++) {
            const msim_range &r = ranges[ri];
            if (r.k == 0) continue;
            clip.push_back((uint32_t)(r.stop + 1));
            for (int64_t i = 0; i < r.k; i++, j++) {
                const uint64_t m = c->np.next53();                   // numpy.random.choice: mutator.py:170-174
                int idx = 0;
                while (idx < r.n_types && r.cdf_thr[idx] <= m) idx++;
                if (idx >= r.n_types) idx = r.n_types - 1;
                ty[j] = (uint8_t)r.types[idx];
                rng_of[j] = di;
                if (ms.sn_chained || ty[j] != MSIM_SN) { ch_rank.push_back((uint32_t)j); ch_type.push_back(ty[j]); }
            }
            di++;
        }
    }
    std::vector<uint32_t> cand_pos(K + 8), ch_stop(ch_rank.size() + 8), visit_from(ms.n_draw + 1), ch_extra(ch_rank.size() + 8);
    std::vector<uint8_t> ch_aux(ch_rank.size() + 8);
    size_t consumed = 0;
    const uint32_t lg = chain_lg_rows(ms.gcc);
    for (size_t W = 4 * K + 65536;; W *= 4) {                        // a window that proves too short is simply retried
        HostMT clone = c->py;
        std::vector<uint32_t> words(W), T((W + 1) << lg);
        for (size_t i = 0; i < W; i++) words[i] = clone.next();
        accept_tables_host(ms.gcc, words.data(), W, T.data());
        const int rc = multimix_walk_host(c, L, ranges, n_ranges, d, ms, words.data(), T.data(), W, ch_rank.data(),
                                          ch_type.data(), ch_rank.size(), cand_pos.data(), ch_stop.data(), visit_from.data(),
                                          &consumed, nullptr, ch_extra.data(), ch_aux.data());
        if (rc == MSIM_OK) break;
        if (rc != MSIM_ERR_HIP || W > (1ull << 31)) return rc;
    }
    for (size_t i = 0; i < consumed; i++) (void)c->py.next();
    std::vector<uint32_t> stop(K, CHAIN_DROPPED), extra(K, 0);
    std::vector<uint8_t> aux(K, 0);
    for (size_t q = 0; q < ch_rank.size(); q++) { stop[ch_rank[q]] = ch_stop[q]; extra[ch_rank[q]] = ch_extra[q]; aux[ch_rank[q]] = ch_aux[q]; }
    out.recs.clear();
    out.pool.clear();
    static const uint8_t ATGC[4] = {'A', 'T', 'G', 'C'};
    uint32_t run = 0;                                                // running maximum of the CLIPPED blocked ends
    size_t kept = 0;
    for (size_t j = 0; j < K; j++) {
        const uint32_t p = cand_pos[j], di = rng_of[j];
        const int t = ty[j];
        bool keep;
        if (t == MSIM_SN) keep = ms.sn_chained ? stop[j] != CHAIN_DROPPED : p >= run;
        else {
            keep = stop[j] != CHAIN_DROPPED;
            if (keep) {                                              // (a TLI's blocked range is range(start, 1 + block): absolute)
                const uint64_t e = t == MSIM_TLI ? 1 + (uint64_t)P.block[t]
                                                 : (uint64_t)(t == MSIM_IN ? p : stop[j]) + 1 + (uint64_t)P.block[t];
                run = std::max(run, (uint32_t)std::min<uint64_t>(e, clip[di]));
            }
            if (aux[j] & CHAIN_TOMBSTONE) keep = false;              // deleted by __fix_tl_amount -- after it had blocked
        }
        if (!keep) continue;
        kept++;
        if (p < visit_from[di]) continue;                            // inside a span an earlier range's DE/DU/IV consumed
        msim_record rec{};
        rec.pos = p;
        rec.type = (uint8_t)t;
        rec.stop = t == MSIM_SN ? p : stop[j];
        if (t == MSIM_SN) {
            const uint64_t u = c->py.next53();
            rec.aux = (u < P.ti_lim) ? 0 : (uint8_t)(1 + randbelow(c->py, 2));
        } else if (t == MSIM_IN) {
            const uint32_t len = rec.stop + 1 - p;
            rec.extra = (uint32_t)out.pool.size();
            for (uint32_t i = 0; i < len; i++) out.pool.push_back(ATGC[c->np.next() & 3u]);
        } else if (t == MSIM_TLI) {
            rec.extra = extra[j];
            rec.aux = aux[j] & 3;
        }
        out.recs.push_back(rec);
    }
    out.empty = kept == 0;
    return MSIM_OK;
}

int plan_contig_host(Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges, HostPlan &out) {
    const auto t0 = std::chrono::steady_clock::now();
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);       // min(mut_block.values())
    const int64_t chrom_len = (int64_t)L;

    std::vector<Cand> all;                                           // `muts` of mutator.py:112
    std::vector<int64_t> pos;
    std::vector<int64_t> tls, tlis;                                  // mutator.py:113-114
    bool sorted_concat = true;
    for (int ri = 0; ri < n_ranges; ri++) {
        const msim_range &r = ranges[ri];
        if (r.n_types < 1 || r.n_types > 8) return fail(c, MSIM_ERR_ARG, "msim_range.n_types out of range");
        const int64_t k = r.k;
        const int64_t n = (r.stop - (k - 1) * d) - r.start;          // util.py:104
        int rc = sample_sorted(c, c->py, n, k, r.setsize, pos);
        if (rc) return rc;
        if (k == 0) continue;                                        // mutator.py:163-164
        // types: numpy.random.choice(keys, p=chances, size=k)      mutator.py:170-174
        const size_t base = all.size();
        all.resize(base + (size_t)k);
        for (int64_t i = 0; i < k; i++) {
            const uint64_t m = c->np.next53();
            int idx = 0;
            while (idx < r.n_types && r.cdf_thr[idx] <= m) idx++;
            if (idx >= r.n_types) idx = r.n_types - 1;               // unreachable: cdf[-1] == 1.0 > u
            Cand cnd;
            cnd.pos = r.start + pos[(size_t)i] + d * i;
            cnd.type = r.types[idx];
            cnd.stop = 0;
            all[base + (size_t)i] = cnd;
        }
        // boundary pass                                             mutator.py:184-213
        size_t w = base;
        int64_t blk_lo = 0, blk_hi = 0;                              // last_mut_range = range(0)
        for (size_t i = base; i < base + (size_t)k; i++) {
            Cand m = all[i];
            if (m.pos >= blk_lo && m.pos < blk_hi) continue;
            const int t = m.type;
            if (t == MSIM_SN) {
                m.stop = m.pos;
            } else if (t == MSIM_IV) {                               // mutator.py:240-248
                if (m.pos + r.max_len[MSIM_IV] >= chrom_len - 1) continue;
                m.stop = randint(c->py, m.pos + r.min_len[t] - 1, m.pos + r.max_len[t] - 1);
            } else if (t == MSIM_IN) {
                m.stop = randint(c->py, m.pos + r.min_len[t] - 1, m.pos + r.max_len[t] - 1);
            } else if (t == MSIM_DU || t == MSIM_DE || t == MSIM_TL) {   // mutator.py:253-264
                m.stop = randint(c->py, m.pos + r.min_len[t] - 1, m.pos + r.max_len[t] - 1);
                if (m.stop > chrom_len - 1) m.stop = chrom_len - 1;
            } else if (t != MSIM_TLI) {
                return fail(c, MSIM_ERR_ARG, "unknown mutation type in msim_range.types");
            }   // TLI: no branch of __get_stop_position matches, stop stays 0 (mutator.py:238-265)
            blk_lo = m.pos;
            blk_hi = ((t == MSIM_SN || t == MSIM_IN) ? m.pos : m.stop) + 1 + P.block[t];
            if (t == MSIM_TL) tls.push_back(m.pos);                  // mutator.py:210-213
            if (t == MSIM_TLI) tlis.push_back(m.pos);
            all[w++] = m;
        }
        all.resize(w);
        if (base > 0 && w > base && all[base].pos <= all[base - 1].pos) sorted_concat = false;
    }
    if (!sorted_concat) {
        // overlapping RMT ranges: dict.update() semantics -- same position, later range wins
        std::vector<size_t> order(all.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return all[a].pos < all[b].pos; });
        std::vector<Cand> merged;
        merged.reserve(all.size());
        for (size_t i = 0; i < order.size(); i++) {
            if (i + 1 < order.size() && all[order[i + 1]].pos == all[order[i]].pos) continue;
            merged.push_back(all[order[i]]);
        }
        all.swap(merged);
    }
    out.empty = all.empty();

    // Translocations: pair every TL (excised span) with a TLI (insert site)   mutator.py:130-131, 267-316
    if (!tls.empty()) {
        auto find = [&](int64_t p) -> Cand * {
            auto it = std::lower_bound(all.begin(), all.end(), p, [](const Cand &a, int64_t v) { return a.pos < v; });
            return (it != all.end() && it->pos == p) ? &*it : nullptr;
        };
        // __fix_tl_amount: drop random surplus entries (a dropped entry leaves `muts`: type 0 = tombstone)
        while (tls.size() < tlis.size()) {
            const size_t idx = (size_t)randint(c->py, 0, (int64_t)tlis.size() - 1);
            if (Cand *q = find(tlis[idx])) q->type = 0;
            tlis.erase(tlis.begin() + (long)idx);
        }
        while (tls.size() > tlis.size()) {
            const size_t idx = (size_t)randint(c->py, 0, (int64_t)tls.size() - 1);
            if (Cand *q = find(tls[idx])) q->type = 0;
            tls.erase(tls.begin() + (long)idx);
        }
        for (size_t i = tls.size(); i-- > 1;) {                      // random.shuffle(tls)
            const size_t j = (size_t)randbelow(c->py, (uint64_t)i + 1);
            std::swap(tls[i], tls[j]);
        }
        for (size_t i = 0; i < tls.size(); i++) {
            Cand *tl = find(tls[i]), *tli = find(tlis[i]);
            if (!tl || !tli) continue;                               // overwritten by an overlapping range
            const int64_t tlen = tl->stop + 1 - tl->pos;
            const bool rev = !(randint(c->py, 0, 1) == 0 || tlen < 2);   // __transloc_invert (draws first)
            // muts[tli_pos] = Mutation(TLI, tl_pos, muts[tl_pos].stop, rev, tli_pos)
            tli->type = MSIM_TLI;
            tli->src = tl->pos;
            tli->stop = tl->stop;
            tli->rev = rev;
            tli->linked = true;
        }
        all.erase(std::remove_if(all.begin(), all.end(), [](const Cand &a) { return a.type == 0; }), all.end());
    }

    // Walk in position order like __mutate_sequence (mutator.py:332-425): entries inside a span an
    // earlier DE/IV/DU consumed are never visited; visited SNPs draw from the CPython stream,
    // visited inserts from the NumPy stream.
    out.recs.clear();
    out.pool.clear();
    out.recs.reserve(all.size());
    static const uint8_t ATGC[4] = {'A', 'T', 'G', 'C'};
    int64_t consumed_to = -1;
    for (const Cand &m : all) {
        if (m.pos <= consumed_to) continue;
        if (m.pos >= chrom_len) continue;                            // `while pos < len(sequence)`
        if ((uint64_t)m.stop >= (1ull << 32))
            return fail(c, MSIM_ERR_UNSUPPORTED, "mutation extent beyond 2^32");
        msim_record rec{};
        rec.pos = (uint32_t)m.pos;
        rec.stop = (uint32_t)m.stop;
        rec.type = (uint8_t)m.type;
        if (m.type == MSIM_SN) {                                     // mutator.py:436-441, :455
            const uint64_t u = c->py.next53();
            rec.aux = (u < P.ti_lim) ? 0 : (uint8_t)(1 + randbelow(c->py, 2));
        } else if (m.type == MSIM_IN) {                              // mutator.py:344, :471
            const int64_t len = m.stop + 1 - m.pos;
            if (out.pool.size() + (uint64_t)len >= (1ull << 32))
                return fail(c, MSIM_ERR_UNSUPPORTED, "insert pool of 4 GiB or more on one contig");
            rec.extra = (uint32_t)out.pool.size();
            const size_t at = out.pool.size();
            out.pool.resize(at + (size_t)len);
            uint8_t *dst = out.pool.data() + at;
            for (int64_t i = 0; i < len; i++) dst[i] = ATGC[c->np.next() & 3u];
        } else if (m.type == MSIM_TLI) {                             // mutator.py:401-421, draws nothing
            // Mutation.start / stop of the linked TL span; an unlinked TLI keeps start = pos, stop = 0
            const int64_t src = m.linked ? m.src : m.pos;
            rec.extra = (uint32_t)src;
            rec.aux = (uint8_t)((m.rev ? 1 : 0) | ((m.linked && m.pos > 0) ? 2 : 0));   // bit 1: trans_insert_pos > 0
        } else {
            consumed_to = m.stop;                                    // pos = muts[pos].stop  (DE, TL, IV, DU)
        }
        out.recs.push_back(rec);
    }
    c->t.plan_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSIM_OK;
}

}  // namespace msim

// ---- settings -> msim_range tables (include/msim.h: msim_build_ranges) ---------------------------------------------------
// The float expressions are the reference's (mutator.py:160-174, 225) evaluated with the same IEEE operations: a double
// multiply and a truncation for k, numpy's cumsum (sequential adds) and one division per threshold, an exact ceil of
// cdf * 2^53 (a multiplication by a power of two is exact).  tests/test_cabi_host.py holds it against
// mutator.plan_descriptors on a million random ranges and on the settings goldens.
extern "C" int msim_build_ranges(const msim_settings_desc *sets, int n_sets, const int64_t *start, const int64_t *stop,
                                 const int32_t *set_id, int64_t n, msim_range *out) {
    if (n < 0 || n_sets < 0 || (n && (!sets || !start || !stop || !set_id || !out))) return MSIM_ERR_ARG;
    std::vector<msim_range> tmpl((size_t)n_sets);
    for (int s = 0; s < n_sets; s++) {
        const msim_settings_desc &d = sets[s];
        if (d.n_types < 0 || d.n_types > 8) return MSIM_ERR_ARG;
        msim_range &t = tmpl[(size_t)s];
        memset(&t, 0, sizeof t);
        t.n_types = d.n_types;
        double cdf[8], run = 0.0;
        for (int j = 0; j < d.n_types; j++) { run += d.chances[j]; cdf[j] = run; }           // numpy.cumsum
        for (int j = 0; j < d.n_types; j++) {
            t.types[j] = d.types[j];
            const double c = cdf[j] / cdf[d.n_types - 1];                                     // cdf /= cdf[-1]
            double x = c * 9007199254740992.0;                                                // exact
            if (!(x > 0.0)) t.cdf_thr[j] = 0;                                                 // (0, negative or NaN: nothing below it)
            else { x = std::ceil(x); t.cdf_thr[j] = x >= 18446744073709551616.0 ? ~0ull : (uint64_t)x; }
        }
        for (int q = 0; q < 8; q++) { t.min_len[q] = d.min_len[q]; t.max_len[q] = d.max_len[q]; }
    }
    for (int64_t i = 0; i < n; i++) {
        if (set_id[i] < 0 || set_id[i] >= n_sets) return MSIM_ERR_ARG;
        msim_range &r = out[i];
        r = tmpl[(size_t)set_id[i]];
        r.start = start[i];
        r.stop = stop[i];
        const double span = (double)((stop[i] - start[i]) + 1);                                // (exact below 2^53)
        const double kf = span * sets[set_id[i]].rate_sum;
        const int64_t k = kf >= 9.2e18 ? INT64_MAX : kf <= -9.2e18 ? INT64_MIN : (int64_t)kf; // int(): towards zero
        r.k = k;
        int64_t setsize = 21;
        if (k > 5) {
            const unsigned __int128 x = (unsigned __int128)3 * (uint64_t)k - 1;                // smallest m with 4^m > 3 k
            const uint64_t hi = (uint64_t)(x >> 64), lo = (uint64_t)x;
            const int bits = hi ? 128 - __builtin_clzll(hi) : 64 - __builtin_clzll(lo);         // (x >= 17)
            const int m = (bits + 1) / 2;
            setsize = m >= 31 ? INT64_MAX : 21 + ((int64_t)1 << (2 * m));
        }
        r.setsize = setsize;
    }
    return MSIM_OK;
}

