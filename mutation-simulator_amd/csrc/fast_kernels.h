// Kernels of the counter-based PLAN engine (MSIM_RNG_FAST; fast_math.h has the construction and the arithmetic).
// gfx950 (MI355X) only; included by plan_fast.hip alone.
//
// Per contig, nothing sequential and nothing on the host:
//   k_fsplit        one workgroup per drawing range: the splitting tree -> points per leaf, candidate ordinal of every leaf
//   k_fleaf<SV>     one WAVE per leaf: m distinct values by rejection into an LDS bitmap, bitmap -> sorted positions
//                   (pos = start + value + d * rank, util.py:104-109).  SNP-only settings: the 16-byte records leave this
//                   kernel finished (SNP outcome included).  Otherwise: type, stop and blocked end per candidate.
//   boundary pass   (mutator.py:184-213) = the orbit of candidate 0 under next(i) = first j with pos[j] >= bend[i]:
//   k_forbit_local    per block of 2048 candidates, pointer doubling in LDS: for every ENTRY e of the block the blocked end
//                     the block hands on; a block whose answer is the same for every entry it can be entered at is
//                     independent of its predecessors (nearly all are: orbits merge within a few candidates)
//   k_forbit_resolve  entry of every block: the predecessor's answer; only runs of dependent blocks are walked
//   k_forbit_mark     the orbit of the entry, marked by doubling -> keep flags (+ per-block counts when final)
//                   contigs with several ranges AND consuming types run the three kernels a second time for the visit
//                   filter of __mutate_sequence (mutator.py:376,386,398): the visited records are the orbit under
//                   next(i) = first j with pos[j] > stop[i] over the kept DE / DU / IV records.
//   k_fscan         block counts -> offsets, totals -> the contig's DynSizes (records, mutated length, insert pool)
//   k_femit         kept candidates -> record table + output offsets + insert pool + SNP outcomes
// APPLY reads the sizes from DynSizes on the device (apply.hip): the host never waits for a count.
#pragma once
#include "ctx.h"
#include "fast_math.h"

namespace msim {
namespace {

using namespace fastrng;

struct FRange {            // one drawing range (k > 0)
    uint32_t start;        // first position
    uint32_t n;            // values of the underlying sample: (stop - (k - 1) d) - start
    uint32_t k;
    uint32_t cand_base;    // ordinal of its first candidate
    uint32_t leaf_base;    // number of its first leaf
    uint32_t lgB;          // a leaf holds 2^lgB values
    uint32_t clip;         // stop + 1: blocked ends are clipped here (the blocked range is reset per range)
    uint32_t set;          // index into the Settings table
};
struct LeafDesc { uint32_t m, cand0, range, t; };          // points, ordinal of the first, its range, leaf number inside the range
struct Block1 { uint32_t v[8]; };                          // block[t] + 1, saturated

enum : uint32_t {
    FF_LEAF_MISMATCH = 1u,     // a leaf's bitmap does not hold the points the tree gave it (internal error)
    FF_POOL_OVERFLOW = 2u,     // insert pool beyond its 16-sigma allocation
    FF_OUT_OVERFLOW = 4u,      // mutated contig beyond its 16-sigma allocation, or >= 4 GiB
    FF_SPLIT_GAVE_UP = 8u,     // 4096 rejections in a row in a hypergeometric draw (never)
    FF_LEAF_GAVE_UP = 16u,
};

// ------------------------------------------------------------------------------------------------ splitting tree
__global__ __launch_bounds__(256) void k_fsplit(const FRange *__restrict__ ranges, Key key, uint32_t *__restrict__ leaf_m,
                                                LeafDesc *__restrict__ leaves, uint32_t *__restrict__ flags) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry;
    const uint32_t r = blockIdx.x;
    const FRange R = ranges[r];
    const uint32_t lgB = R.lgB;
    const uint32_t T = (uint32_t)(((uint64_t)R.n + (1ull << lgB) - 1) >> lgB);
    uint32_t lgP = 0;
    while ((1u << lgP) < T) lgP++;
    uint32_t *m = leaf_m + R.leaf_base;
    if (threadIdx.x == 0) m[0] = R.k;
    __syncthreads();
    for (uint32_t lev = 0; lev < lgP; lev++) {
        const uint32_t S = 1u << (lgP - lev), half = S >> 1;
        for (uint32_t i = threadIdx.x; i < (1u << lev); i += 256) {
            const uint32_t a = i * S, mid = a + half;
            if (mid >= T) continue;                        // everything of this node lies left of the middle
            const uint32_t K = m[a];
            const uint64_t va = (uint64_t)a << lgB, vm = (uint64_t)mid << lgB;
            const uint64_t vb = min((uint64_t)(a + S) << lgB, (uint64_t)R.n);
            uint32_t att = 0;
            const uint32_t kl = (uint32_t)hypergeometric(vm - va, vb - vm, K, key, (1u << lev) + i, r, &att);
            if (att >= 4096) atomicOr(flags, (uint32_t)FF_SPLIT_GAVE_UP);
            m[a] = kl;
            m[mid] = K - kl;
        }
        __syncthreads();
    }
    // candidate ordinal of every leaf
    if (threadIdx.x == 0) carry = R.cand_base;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < T; base += 256) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = t < T ? m[t] : 0;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t x = __shfl_up(incl, o, 64);
            if (lane >= o) incl += x;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (int w = 0; w < 4; w++) { if (w < wave) pre += wsum[w]; tot += wsum[w]; }
        const uint32_t c0 = carry;
        if (t < T) {
            LeafDesc D;
            D.m = v; D.cand0 = c0 + pre + incl - v; D.range = r; D.t = t;
            leaves[R.leaf_base + t] = D;
        }
        __syncthreads();
        if (threadIdx.x == 0) carry = c0 + tot;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ leaves
// one wave per leaf; dynamic LDS: 4 bitmaps of 2^lgBmax bits
template <bool SV>
__global__ __launch_bounds__(256) void k_fleaf(const FRange *__restrict__ ranges, const LeafDesc *__restrict__ leaves,
                                               uint32_t n_leaves, uint32_t bm_words, Key key, uint32_t d, uint64_t L,
                                               const Settings *__restrict__ sets, Block1 block1, unsigned long long ti_lim,
                                               msim_record *__restrict__ recs, uint32_t *__restrict__ cand_pos,
                                               uint32_t *__restrict__ cand_stop, uint32_t *__restrict__ cand_bend,
                                               uint8_t *__restrict__ cand_meta, uint32_t *__restrict__ flags) {
    extern __shared__ uint32_t lds_bm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x * 4 + wave;
    if (g >= n_leaves) return;                             // (no workgroup barrier below: waves are on their own)
    uint32_t *bm = lds_bm + (size_t)wave * bm_words;
    const LeafDesc D = leaves[g];
    const FRange R = ranges[D.range];
    const uint32_t v0 = D.t << R.lgB;
    const uint32_t len = min(1u << R.lgB, R.n - v0);
    const uint32_t m = D.m;
    if (m == 0) return;
    const uint32_t words32 = (len + 31) >> 5;
    for (uint32_t w = lane; w < words32 + 1; w += 64) bm[w] = 0;       // (+1: the expansion reads 64-bit words)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the first `need` distinct values of the leaf's draw sequence: a uniform subset (of the complement when the leaf is
    // more than half full)
    const bool inv = 2 * m > len;
    const uint32_t need = inv ? len - m : m;
    uint32_t have = 0, done = 0;
    while (have < need) {
        const uint32_t cnt = min(need - have, 64u);
        bool fresh = false;
        if ((uint32_t)lane < cnt) {
            const uint32_t v = leaf_draw(key, g, done + lane, len);
            const uint32_t bit = 1u << (v & 31);
            fresh = !(atomicOr(&bm[v >> 5], bit) & bit);
        }
        have += (uint32_t)__popcll(__ballot(fresh));
        done += cnt;
        if (done > 64u * need + 65536u) {                  // (a geometric tail that long does not happen)
            if (lane == 0) atomicOr(flags, (uint32_t)FF_LEAF_GAVE_UP);
            return;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // bitmap -> sorted values: lane l owns 64-bit words [l W, (l + 1) W)
    const uint32_t words64 = (len + 63) >> 6;
    const uint32_t W = (words64 + 63) >> 6;
    auto word_at = [&](uint32_t wi) -> unsigned long long {
        if (wi >= words64) return 0ull;
        unsigned long long w = (unsigned long long)bm[2 * wi] | ((unsigned long long)bm[2 * wi + 1] << 32);
        if (inv) w = ~w;
        const uint32_t valid = len - wi * 64;
        if (valid < 64) w &= (1ull << valid) - 1ull;
        return w;
    };
    uint32_t cnt = 0;
    for (uint32_t q = 0; q < W; q++) cnt += (uint32_t)__popcll(word_at(lane * W + q));
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(incl, o, 64);
        if (lane >= o) incl += x;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    if (total != m) {
        if (lane == 0) atomicOr(flags, (uint32_t)FF_LEAF_MISMATCH);
        return;
    }
    uint32_t ord = D.cand0 + incl - cnt;
    const uint32_t pos0 = R.start + v0 - d * R.cand_base;  // pos = start + v0 + v + d * (ord - cand_base)   (mod 2^32 throughout)
    const Settings *S = SV ? sets + R.set : nullptr;
    for (uint32_t q = 0; q < W; q++) {
        const uint32_t wi = lane * W + q;
        unsigned long long w = word_at(wi);
        while (w) {
            const uint32_t v = wi * 64 + (uint32_t)__builtin_ctzll(w);
            w &= w - 1;
            const uint32_t pos = pos0 + v + d * ord;
            if (SV) {
                const Cand c = cand_draw(key, ord, pos, L, *S, block1.v, R.clip);
                cand_pos[ord] = pos;
                cand_stop[ord] = c.stop;
                cand_bend[ord] = c.bend;
                cand_meta[ord] = c.meta;
            } else {
                msim_record rec;
                rec.pos = pos; rec.stop = pos; rec.extra = 0; rec.type = MSIM_SN;
                rec.aux = snp_outcome(key, ord, ti_lim); rec.rsv = 0;
                recs[ord] = rec;
            }
            ord++;
        }
    }
}

// ------------------------------------------------------------------------------------------------ the orbit passes
constexpr int OB_THREADS = 256, OB_ITEMS = 8, OB_BLOCK = OB_THREADS * OB_ITEMS;   // 2048 candidates per workgroup
constexpr int OB_ROUNDS = 11;                                                       // 2^11 = OB_BLOCK

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint32_t x) {   // first i with a[i] >= x
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ bool type_consumes(uint32_t t) { return t == MSIM_DE || t == MSIM_DU || t == MSIM_IV; }
__device__ __forceinline__ long long cand_delta(uint32_t t, uint32_t pos, uint32_t stop) {          // mutator.py:343-399
    const long long len = (long long)stop - (long long)pos + 1;
    return t == MSIM_IN ? len : t == MSIM_DE ? -len : t == MSIM_DU ? len : 0ll;
}

// For every entry e of the block (the first candidate a predecessor's blocked end lets through) the blocked end the block
// hands to its successor: bend of the last candidate of e's orbit inside the block.
__global__ __launch_bounds__(OB_THREADS) void k_forbit_local(const uint32_t *__restrict__ cand_pos,
                                                             const uint32_t *__restrict__ cand_bend, uint32_t K, uint32_t maxspan,
                                                             uint32_t *__restrict__ blk_out, uint32_t *__restrict__ blk_S,
                                                             uint32_t *__restrict__ blk_indep) {
    __shared__ uint32_t pos[OB_BLOCK], E[OB_BLOCK];
    __shared__ uint16_t pa[OB_BLOCK], pb[OB_BLOCK];
    __shared__ uint32_t bad;
    const uint32_t base = blockIdx.x * OB_BLOCK, cnt = min((uint32_t)OB_BLOCK, K - base);
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        const uint32_t i = threadIdx.x + q * OB_THREADS;
        pos[i] = i < cnt ? cand_pos[base + i] : 0xffffffffu;
        E[i] = i < cnt ? cand_bend[base + i] : 0xffffffffu;
    }
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        const uint32_t i = threadIdx.x + q * OB_THREADS;
        if (i < cnt) {
            const uint32_t nx = lower_bound_u32(pos, cnt, E[i]);       // > i: bend > pos
            pa[i] = (uint16_t)(nx < cnt ? nx : i);                      // the last of its orbit points at itself
        }
    }
    __syncthreads();
    uint16_t *a = pa, *b = pb;
    for (int r = 0; r < OB_ROUNDS; r++) {
#pragma unroll
        for (int q = 0; q < OB_ITEMS; q++) {
            const uint32_t i = threadIdx.x + q * OB_THREADS;
            if (i < cnt) b[i] = a[a[i]];
        }
        __syncthreads();
        uint16_t *t = a; a = b; b = t;
    }
    const uint32_t S = E[a[0]];
    const uint32_t far = pos[0] + maxspan < pos[0] ? 0xffffffffu : pos[0] + maxspan;
    const uint32_t e_hi = lower_bound_u32(pos, cnt, far);               // entries 0 .. e_hi can occur; cnt = passed through
    if (threadIdx.x == 0 && e_hi >= cnt) bad = 1;
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        const uint32_t i = threadIdx.x + q * OB_THREADS;
        if (i < cnt) {
            const uint32_t oe = E[a[i]];
            blk_out[base + i] = oe;
            if (i <= e_hi && oe != S) bad = 1;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { blk_S[blockIdx.x] = S; blk_indep[blockIdx.x] = bad ? 0u : 1u; }
}

// blocked end every block is entered with.  One workgroup.
__global__ __launch_bounds__(1024) void k_forbit_resolve(const uint32_t *__restrict__ cand_pos, const uint32_t *__restrict__ blk_out,
                                                         const uint32_t *__restrict__ blk_S, const uint32_t *__restrict__ blk_indep,
                                                         uint32_t K, uint32_t nb, uint32_t *__restrict__ blk_in) {
    for (uint32_t b = threadIdx.x; b < nb; b += 1024) {
        const bool known = b == 0 || blk_indep[b - 1];
        if (!known) continue;                              // (set by the walk of the run it belongs to)
        uint32_t x = b ? blk_S[b - 1] : 0u;
        blk_in[b] = x;
        for (uint32_t cur = b; cur + 1 < nb && !blk_indep[cur]; cur++) {      // a run of dependent blocks: one after the other
            const uint32_t base = cur * OB_BLOCK, cnt = min((uint32_t)OB_BLOCK, K - base);
            const uint32_t e = lower_bound_u32(cand_pos + base, cnt, x);
            if (e < cnt) x = blk_out[base + e];
            blk_in[cur + 1] = x;
        }
    }
}

// The orbit of the block's entry, marked by doubling.
//   VISIT = false: boundary pass.  kept = on the orbit and not dropped -> CAND_KEEP; !FINAL: bend := the span the record
//                  consumes (stop + 1 for a kept DE / DU / IV, else pos + 1), the visit pass's input.
//   VISIT = true : visit pass.  record = CAND_KEEP and on the orbit -> CAND_VISIT.
//   FINAL: CAND_VISIT marks the records; per-block counts (records, insert bases, length change) for k_fscan.
template <bool VISIT, bool FINAL>
__global__ __launch_bounds__(OB_THREADS) void k_forbit_mark(const uint32_t *__restrict__ cand_pos, uint32_t *__restrict__ cand_bend,
                                                            const uint32_t *__restrict__ cand_stop, uint8_t *__restrict__ cand_meta,
                                                            uint32_t K, const uint32_t *__restrict__ blk_in,
                                                            uint32_t *__restrict__ blk_nrec, uint32_t *__restrict__ blk_pool,
                                                            long long *__restrict__ blk_delta, uint32_t *__restrict__ kept_any) {
    __shared__ uint32_t pos[OB_BLOCK], E[OB_BLOCK];
    __shared__ uint16_t pa[OB_BLOCK + 1], pb[OB_BLOCK + 1];
    __shared__ uint8_t mark[OB_BLOCK];
    __shared__ uint32_t red_n[OB_THREADS / 64], red_p[OB_THREADS / 64];
    __shared__ long long red_d[OB_THREADS / 64];
    const uint32_t base = blockIdx.x * OB_BLOCK, cnt = min((uint32_t)OB_BLOCK, K - base);
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        const uint32_t i = threadIdx.x + q * OB_THREADS;
        pos[i] = i < cnt ? cand_pos[base + i] : 0xffffffffu;
        E[i] = i < cnt ? cand_bend[base + i] : 0xffffffffu;
        mark[i] = 0;
    }
    __syncthreads();
    const uint32_t entry = lower_bound_u32(pos, cnt, blk_in[blockIdx.x]);
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        const uint32_t i = threadIdx.x + q * OB_THREADS;
        if (i < cnt) pa[i] = (uint16_t)lower_bound_u32(pos, cnt, E[i]);           // cnt: leaves the block
    }
    if (threadIdx.x == 0) { pa[cnt] = (uint16_t)cnt; pb[cnt] = (uint16_t)cnt; if (entry < cnt) mark[entry] = 1; }
    __syncthreads();
    uint16_t *a = pa, *b = pb;
    for (int r = 0; r < OB_ROUNDS; r++) {
#pragma unroll
        for (int q = 0; q < OB_ITEMS; q++) {
            const uint32_t i = threadIdx.x + q * OB_THREADS;
            if (i < cnt) {
                const uint32_t nx = a[i];
                if (mark[i] && nx < cnt) mark[nx] = 1;
                b[i] = a[nx];                                                       // a[cnt] = cnt
            }
        }
        __syncthreads();
        uint16_t *t = a; a = b; b = t;
    }
    uint32_t n_rec = 0, n_pool = 0;
    long long delta = 0;
    bool any = false;
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        const uint32_t i = threadIdx.x + q * OB_THREADS;
        if (i >= cnt) continue;
        uint8_t meta = cand_meta[base + i];
        const uint32_t t = meta & 7u;
        bool rec;
        if (!VISIT) {
            const bool keep = mark[i] && !(meta & CAND_DROPPED);
            if (keep) meta |= CAND_KEEP;
            any = any || keep;
            rec = keep;
            if (!FINAL) {
                const uint32_t stop = cand_stop[base + i];
                cand_bend[base + i] = (keep && type_consumes(t)) ? (stop == 0xffffffffu ? stop : stop + 1) : pos[i] + 1;
            }
        } else {
            rec = (meta & CAND_KEEP) && mark[i];
        }
        if (FINAL) {
            if (rec) {
                meta |= CAND_VISIT;
                const uint32_t stop = cand_stop[base + i];
                n_rec++;
                if (t == MSIM_IN) n_pool += stop - pos[i] + 1;
                delta += cand_delta(t, pos[i], stop);
            }
        }
        cand_meta[base + i] = meta;
    }
    if (!VISIT && any) *kept_any = 1u;                     // (mutator.py:125-129: the warning looks at muts before the rewrite)
    if (FINAL) {
        for (int o = 32; o > 0; o >>= 1) {
            n_rec += __shfl_down(n_rec, o, 64);
            n_pool += __shfl_down(n_pool, o, 64);
            delta += __shfl_down(delta, o, 64);
        }
        if ((threadIdx.x & 63) == 0) { red_n[threadIdx.x >> 6] = n_rec; red_p[threadIdx.x >> 6] = n_pool; red_d[threadIdx.x >> 6] = delta; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sn = 0, sp = 0;
            long long sd = 0;
            for (int w = 0; w < OB_THREADS / 64; w++) { sn += red_n[w]; sp += red_p[w]; sd += red_d[w]; }
            blk_nrec[blockIdx.x] = sn; blk_pool[blockIdx.x] = sp; blk_delta[blockIdx.x] = sd;
        }
    }
}

// ------------------------------------------------------------------------------------------------ counts -> offsets, sizes
struct DynSizes { uint32_t n_rec, out_len, pool_len, flags; };        // what APPLY and the collecting host read (apply.hip)

// exclusive scans of the three block arrays in place; totals -> DynSizes.  One workgroup.
__global__ __launch_bounds__(1024) void k_fscan(uint32_t *__restrict__ blk_nrec, uint32_t *__restrict__ blk_pool,
                                                long long *__restrict__ blk_delta, uint32_t nb, uint64_t L, uint64_t out_cap,
                                                uint64_t pool_cap, const uint32_t *__restrict__ kept_any, const uint32_t *__restrict__ flags,
                                                DynSizes *__restrict__ dyn) {
    __shared__ unsigned long long wsum[3][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long carry[3] = {0, 0, 0};
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        unsigned long long v[3] = {0, 0, 0}, incl[3];
        if (i < nb) { v[0] = blk_nrec[i]; v[1] = blk_pool[i]; v[2] = (unsigned long long)blk_delta[i]; }
        for (int j = 0; j < 3; j++) {
            incl[j] = v[j];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long x = __shfl_up(incl[j], o, 64);
                if (lane >= o) incl[j] += x;
            }
            if (lane == 63) wsum[j][wave] = incl[j];
        }
        __syncthreads();
        for (int j = 0; j < 3; j++) {
            unsigned long long pre = 0, tot = 0;
            for (int w = 0; w < 16; w++) { if (w < wave) pre += wsum[j][w]; tot += wsum[j][w]; }
            incl[j] = carry[j] + pre + incl[j] - v[j];
            carry[j] += tot;
        }
        if (i < nb) { blk_nrec[i] = (uint32_t)incl[0]; blk_pool[i] = (uint32_t)incl[1]; blk_delta[i] = (long long)incl[2]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const long long out_len = (long long)L + (long long)carry[2];
        uint32_t f = *flags;
        if (carry[1] > pool_cap || carry[1] >= (1ull << 32)) f |= FF_POOL_OVERFLOW;
        if (out_len < 0 || (unsigned long long)out_len > out_cap || (unsigned long long)out_len >= (1ull << 32)) f |= FF_OUT_OVERFLOW;
        DynSizes s;
        s.n_rec = (f & (FF_POOL_OVERFLOW | FF_OUT_OVERFLOW)) ? 0u : (uint32_t)carry[0];   // (an overflowing plan applies nothing)
        s.out_len = (f & (FF_POOL_OVERFLOW | FF_OUT_OVERFLOW)) ? (uint32_t)L : (uint32_t)out_len;
        s.pool_len = (uint32_t)carry[1];
        s.flags = f | (*kept_any ? 0x100u : 0u);
        *dyn = s;
    }
}

// ------------------------------------------------------------------------------------------------ records
// kept candidates -> record table (position order) + output offsets (apply.hip needs no scan) + insert pool + SNP outcomes
__global__ __launch_bounds__(OB_THREADS) void k_femit(const uint32_t *__restrict__ cand_pos, const uint32_t *__restrict__ cand_stop,
                                                      const uint8_t *__restrict__ cand_meta, uint32_t K,
                                                      const uint32_t *__restrict__ off_nrec, const uint32_t *__restrict__ off_pool,
                                                      const long long *__restrict__ off_delta, const DynSizes *__restrict__ dyn,
                                                      Key key, unsigned long long ti_lim, msim_record *__restrict__ recs,
                                                      uint32_t *__restrict__ rec_off, uint8_t *__restrict__ pool) {
    __shared__ uint32_t wn[OB_THREADS / 64], wp[OB_THREADS / 64];
    __shared__ long long wd[OB_THREADS / 64];
    if (dyn->flags & (FF_POOL_OVERFLOW | FF_OUT_OVERFLOW)) return;       // nothing may be written beyond an allocation
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i0 = blockIdx.x * OB_BLOCK + threadIdx.x * OB_ITEMS;
    uint32_t pos[OB_ITEMS], stop[OB_ITEMS];
    uint8_t meta[OB_ITEMS];
    uint32_t nk = 0, np = 0;
    long long nd = 0;
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        meta[q] = i0 + q < K ? cand_meta[i0 + q] : (uint8_t)0;
        pos[q] = 0; stop[q] = 0;
        if (meta[q] & CAND_VISIT) {
            pos[q] = cand_pos[i0 + q];
            stop[q] = cand_stop[i0 + q];
            const uint32_t t = meta[q] & 7u;
            nk++;
            if (t == MSIM_IN) np += stop[q] - pos[q] + 1;
            nd += cand_delta(t, pos[q], stop[q]);
        }
    }
    uint32_t in = nk, ip = np;
    long long id = nd;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t xn = __shfl_up(in, o, 64), xp = __shfl_up(ip, o, 64);
        const long long xd = __shfl_up(id, o, 64);
        if (lane >= o) { in += xn; ip += xp; id += xd; }
    }
    if (lane == 63) { wn[wave] = in; wp[wave] = ip; wd[wave] = id; }
    __syncthreads();
    uint32_t r = off_nrec[blockIdx.x] + in - nk, p = off_pool[blockIdx.x] + ip - np;
    long long shift = off_delta[blockIdx.x] + id - nd;
    for (int w = 0; w < wave; w++) { r += wn[w]; p += wp[w]; shift += wd[w]; }
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        if (!(meta[q] & CAND_VISIT)) continue;
        const uint32_t t = meta[q] & 7u, ord = i0 + q;
        msim_record rec;
        rec.pos = pos[q]; rec.stop = stop[q]; rec.extra = 0; rec.type = (uint8_t)t; rec.aux = 0; rec.rsv = 0;
        if (t == MSIM_SN) rec.aux = snp_outcome(key, ord, ti_lim);
        if (t == MSIM_IN) {
            rec.extra = p;
            const uint32_t len = stop[q] - pos[q] + 1;
            for (uint32_t c0 = 0; c0 < len; c0 += 64) {                 // 64 bases per counter
                const U4 ch = draw4(key, c0 >> 6, ord, TAG_INS);
                const uint32_t nb = min(64u, len - c0);
                for (uint32_t j = 0; j < nb; j++) pool[p + c0 + j] = insert_base_of(ch, j);
            }
            p += len;
        }
        rec_off[r] = (uint32_t)((long long)pos[q] + shift);
        shift += cand_delta(t, pos[q], stop[q]);
        recs[r++] = rec;
    }
}

}  // namespace
}  // namespace msim
