// Kernels of the counter-based PLAN engine (MSIM_RNG_FAST; fast_math.h has the construction and the arithmetic).
// gfx950 (MI355X) only; included by plan_fast.hip alone.
//
// Per contig, nothing sequential and nothing on the host:
//   k_fsplit_top / k_fsplit_sub   the splitting tree -> points per leaf, candidate ordinal of every leaf
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
    uint32_t sub_base;     // number of its first subtree (64 leaves each)
    uint32_t slot;         // the contig of the batch it belongs to
};
// One contig of a batch.  Every kernel below runs over ALL contigs of a batch at once (a genome = one launch per stage:
// per-contig launches are latency-bound -- a 125 Mb contig is one round of workgroups -- and 24 of them on 8 streams still
// left the device half idle).  Counters of the generator are per contig (ordinals, leaves, ranges restart at 0 with every
// contig, `seq` tells the contigs apart), storage is concatenated: *_off says where the contig's part starts.
struct FSlot {
    uint64_t L;                // contig length
    uint64_t out_cap, pool_cap;
    msim_record *recs;         // the contig's own tables (ctx.h: Contig)
    uint32_t *rec_off;
    uint8_t *pool;
    struct DynSizes *dyn;
    uint32_t seq;              // contig ordinal of the generator
    uint32_t K;                // candidates
    uint32_t cand_off;         // first candidate in the candidate arrays (a multiple of OB_BLOCK)
    uint32_t blk_off, nb;      // its blocks in the block arrays
    uint32_t range_off, leaf_off, sub_off;
};
// A leaf, self-contained (k_fsplit_sub writes it, k_fleaf reads nothing else but its contig's slot and settings: the chain
// leaf -> range -> slot -> settings was four dependent round trips at the head of every wave)
struct LeafDesc {
    uint32_t m, cand0;         // points; the contig's ordinal of the first one
    uint32_t leaf_id;          // the generator's leaf number (per contig)
    uint32_t len;              // values it spans (2^lgB, the range's last leaf fewer)
    uint32_t pos0;             // position of value 0 of the leaf's candidate 0: pos = pos0 + value + d * ordinal  (mod 2^32)
    uint32_t clip, set, slot;  // the range's stop + 1, its settings, its contig
};
struct Key2 { uint32_t k0, k1; };                          // the generator's key; the contig ordinal comes from the slot
struct Block1 { uint32_t v[8]; };                          // block[t] + 1, saturated

enum : uint32_t {
    FF_LEAF_MISMATCH = 1u,     // a leaf's bitmap does not hold the points the tree gave it (internal error)
    FF_POOL_OVERFLOW = 2u,     // insert pool beyond its 16-sigma allocation
    FF_OUT_OVERFLOW = 4u,      // mutated contig beyond its 16-sigma allocation, or >= 4 GiB
    FF_SPLIT_GAVE_UP = 8u,     // 4096 rejections in a row in a hypergeometric draw (never)
    FF_LEAF_GAVE_UP = 16u,
    FF_KEPT_ANY = 0x100u,      // (DynSizes only) some candidate survived the boundary pass
    FF_NEED_ORBIT = 0x200u,    // (DynSizes only) the block-local pass handed over: the plan must be replayed with the orbit kernels
};

// a record as ONE 16-byte store (field by field the compiler stores pos / stop / extra, type, aux and rsv separately: four stores)
__device__ __forceinline__ void store_record(msim_record *dst, uint32_t pos, uint32_t stop, uint32_t extra, uint32_t type, uint32_t aux) {
    uint4 v;
    v.x = pos; v.y = stop; v.z = extra; v.w = type | (aux << 8);
    *reinterpret_cast<uint4 *>(dst) = v;                   // (tables are 16-byte aligned: hipMalloc + 16 n)
}
// ------------------------------------------------------------------------------------------------ splitting tree
// A node's draw costs a few microseconds of dependent double-precision work, so the tree is cut at subtrees of 64 leaves:
//   k_fsplit_top   one workgroup per range with more than 64 leaves: the levels above the subtrees (level l has 2^l nodes),
//                  then the candidate ordinal of every subtree
//   k_fsplit_sub   one WAVE per subtree (every range has at least one): its six levels in LDS, lane j = node j of the
//                  level, then the candidate ordinal of every leaf
// (the first version walked all levels in ONE workgroup per range: 445 us per 240 Mb contig, 84 % of the engine's kernel time --
//  the two bottom levels alone are 12 000 draws on four waves.)
constexpr int OB_THREADS = 256, OB_ITEMS = 8, OB_BLOCK = OB_THREADS * OB_ITEMS;   // candidates per workgroup of the boundary pass
constexpr int OB_BLOCK_LG = 11, OB_ROUNDS = 11;                                     // 2^11 = OB_BLOCK
constexpr uint32_t LG_SUB = 6;                             // leaves per subtree: 2^6
struct SubDesc { uint32_t range, s; };                     // subtree s of its range (leaves [64 s, 64 s + 64))

// Hypergeometric(good, bad, sample) of node `node`, evaluated by a GROUP of G = 2^lgG neighbouring lanes (G <= 64, groups
// aligned): the lanes of a group hold the same arguments, lane j of it tries attempts j, G + j, ... of the ratio-of-uniforms
// loop and the group takes the outcome of the lowest accepted attempt -- what the sequential loop of fast_math.h returns,
// in a third of its rounds (a wave otherwise waits for its unluckiest lane: ~3 attempts at 64 lanes).  Every lane of the
// wave must call it (active = false: no node).
__device__ __forceinline__ uint32_t hyp_group(bool active, uint64_t good, uint64_t bad, uint64_t sample, const Key &key, uint32_t node,
                                              uint32_t range, uint32_t lgG, uint32_t *flags) {
    const uint32_t lane = threadIdx.x & 63, G = 1u << lgG, grp0 = lane & ~(G - 1), sub = lane & (G - 1);
    HypPrep P;
    P.kind = 0; P.z = 0;
    if (active) P = hyp_prepare(good, bad, sample);
    uint64_t z = 0;
    if (active && P.kind == 1) z = hyp_urn(P, key, node, range);
    bool pending = active && P.kind == 2;
    uint32_t base = 0;
    while (__ballot(pending)) {
        bool acc = false;
        uint64_t Zc = 0;
        if (pending) acc = hyp_attempt(P, key, node, range, base + sub, Zc);
        const unsigned long long mask = __ballot(acc);
        const unsigned long long gm = G == 64 ? mask : (mask >> grp0) & ((1ull << G) - 1ull);
        const int win = gm ? __builtin_ctzll(gm) : 0;
        const uint32_t zw = (uint32_t)__shfl((int)(uint32_t)Zc, (int)(grp0 + win), 64);      // (counts are below 2^31)
        if (pending && gm) { z = zw; pending = false; }
        base += G;
        if (base >= HYP_MAX_ATTEMPTS) {
            if (pending) { atomicOr(flags, (uint32_t)FF_SPLIT_GAVE_UP); z = P.mode; pending = false; }
        }
    }
    return active ? (uint32_t)hyp_finish(P, z) : 0u;
}

__global__ __launch_bounds__(1024) void k_fsplit_top(const FRange *__restrict__ ranges, const uint32_t *__restrict__ big,
                                                     const FSlot *__restrict__ slots, Key2 key2,
                                                     uint32_t *__restrict__ sub_k, uint32_t *__restrict__ sub_c0,
                                                     uint32_t *__restrict__ flags) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const uint32_t r_tab = big[blockIdx.x];
    const FRange R = ranges[r_tab];
    const FSlot &SL = slots[R.slot];
    const Key key{key2.k0, key2.k1, SL.seq};
    const uint32_t r = r_tab - SL.range_off;               // the generator counts ranges per contig
    sub_k += SL.sub_off;
    sub_c0 += SL.sub_off;
    const uint32_t lgB = R.lgB;
    const uint32_t T = (uint32_t)(((uint64_t)R.n + (1ull << lgB) - 1) >> lgB);
    uint32_t lgP = 0;
    while ((1u << lgP) < T) lgP++;
    const uint32_t lgTop = lgP - LG_SUB, n_sub = (T + 63) >> LG_SUB;
    uint32_t *m = sub_k + R.sub_base;                       // K of the node that starts at subtree a >> 6
    if (threadIdx.x == 0) m[0] = R.k;
    __syncthreads();
    for (uint32_t lev = 0; lev < lgTop; lev++) {
        const uint32_t S = 1u << (lgP - lev), half = S >> 1;
        const uint32_t lgG = lev >= 10 ? 0u : min(6u, 10u - lev);       // lanes per node: the 1024 threads over the level's nodes
        const uint32_t per_pass = 1024u >> lgG;
        for (uint32_t i0 = 0; i0 < (1u << lev); i0 += per_pass) {        // (uniform trip count: hyp_group needs whole waves)
            const uint32_t i = i0 + (threadIdx.x >> lgG);
            const uint32_t a = i * S, mid = a + half;
            const bool split = i < (1u << lev) && mid < T;              // else: everything of this node lies left of the middle
            const uint32_t K = split ? m[a >> LG_SUB] : 0;
            const uint64_t va = (uint64_t)a << lgB, vm = (uint64_t)mid << lgB;
            const uint64_t vb = min((uint64_t)(a + S) << lgB, (uint64_t)R.n);
            const uint32_t kl = hyp_group(split, vm - va, split ? vb - vm : 0, K, key, (1u << lev) + i, r, lgG, flags);
            if (split && (threadIdx.x & ((1u << lgG) - 1)) == 0) {
                m[a >> LG_SUB] = kl;
                m[mid >> LG_SUB] = K - kl;
            }
        }
        __syncthreads();
    }
    // candidate ordinal of every subtree
    if (threadIdx.x == 0) carry = R.cand_base;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n_sub; base += 1024) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = t < n_sub ? m[t] : 0;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t x = __shfl_up(incl, o, 64);
            if (lane >= o) incl += x;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (int w = 0; w < 16; w++) { if (w < wave) pre += wsum[w]; tot += wsum[w]; }
        const uint32_t c0 = carry;
        if (t < n_sub) sub_c0[R.sub_base + t] = c0 + pre + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c0 + tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_fsplit_sub(const FRange *__restrict__ ranges, const SubDesc *__restrict__ subs, uint32_t n_subs,
                                                    const FSlot *__restrict__ slots, Key2 key2,
                                                    const uint32_t *__restrict__ sub_k, const uint32_t *__restrict__ sub_c0,
                                                    LeafDesc *__restrict__ leaves, uint32_t *__restrict__ flags,
                                                    uint32_t *__restrict__ zero_a, uint32_t n_zero_a, uint32_t *__restrict__ zero_b,
                                                    uint32_t n_zero_b, uint32_t d) {
    __shared__ uint32_t lds_m[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (what the later kernels of this batch accumulate into starts at zero: the per-block maxima, the kept-any / hand-over words)
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < n_zero_a; q += gridDim.x * 256) zero_a[q] = 0;
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < n_zero_b; q += gridDim.x * 256) zero_b[q] = 0;
    const uint32_t g = blockIdx.x * 4 + wave;
    if (g >= n_subs) return;                               // (no workgroup barrier below)
    const SubDesc D0 = subs[g];
    const FRange R = ranges[D0.range];
    const FSlot &SL = slots[R.slot];
    const Key key{key2.k0, key2.k1, SL.seq};
    SubDesc D = D0;
    D.range = D0.range - SL.range_off;                     // the generator counts ranges per contig
    sub_k += SL.sub_off;
    sub_c0 += SL.sub_off;
    leaves += SL.leaf_off;
    const uint32_t lgB = R.lgB;
    const uint32_t T = (uint32_t)(((uint64_t)R.n + (1ull << lgB) - 1) >> lgB);
    uint32_t lgP = 0;
    while ((1u << lgP) < T) lgP++;
    const bool has_top = lgP > LG_SUB;
    const uint32_t lgTop = has_top ? lgP - LG_SUB : 0;
    const uint32_t a0 = D.s << LG_SUB;                     // first leaf of the subtree
    uint32_t *m = lds_m[wave];
    m[lane] = 0;
    if (lane == 0) m[0] = has_top ? sub_k[R.sub_base + D.s] : R.k;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t lev = lgTop; lev < lgP; lev++) {
        const uint32_t S = 1u << (lgP - lev), half = S >> 1;
        const uint32_t lgN = lev - lgTop;                  // 2^lgN nodes of this subtree at this level, 64 >> lgN lanes each
        const uint32_t lgG = LG_SUB - lgN;
        const uint32_t i = (D.s << lgN) + ((uint32_t)lane >> lgG);
        const uint32_t a = i * S, mid = a + half;
        const bool split = mid < T;
        const uint32_t K = split ? m[a - a0] : 0;
        const uint64_t va = (uint64_t)a << lgB, vm = (uint64_t)mid << lgB;
        const uint64_t vb = min((uint64_t)(a + S) << lgB, (uint64_t)R.n);
        const uint32_t kl = hyp_group(split, vm - va, split ? vb - vm : 0, K, key, (1u << lev) + i, D.range, lgG, flags);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (split && ((uint32_t)lane & ((1u << lgG) - 1)) == 0) { m[a - a0] = kl; m[mid - a0] = K - kl; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t t = a0 + lane;
    const uint32_t v = t < T ? m[lane] : 0;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(incl, o, 64);
        if (lane >= o) incl += x;
    }
    if (t < T) {
        LeafDesc L;
        L.m = v; L.cand0 = (has_top ? sub_c0[R.sub_base + D.s] : R.cand_base) + incl - v;
        L.leaf_id = R.leaf_base + t;
        L.len = min(1u << lgB, R.n - (t << lgB));
        L.pos0 = R.start + (t << lgB) - d * R.cand_base;
        L.clip = R.clip; L.set = R.set; L.slot = R.slot;
        leaves[R.leaf_base + t] = L;
    }
}

// ------------------------------------------------------------------------------------------------ leaves
// one wave per leaf; dynamic LDS per wave: a bitmap of 2^lgBmax bits, then the list of the leaf's values
constexpr uint32_t LEAF_LIST = 512;                        // values a leaf can stage for the balanced pass (a leaf holds 224-448; fuller
                                                           // ones -- 3 sigma and more above the densest mean -- take the unbalanced
                                                           // loop), 16 bits each (a value lies below 2^16: LG_LEAF_MAX): 1 KB beside
                                                           // the 4 KB bitmap of a 2^15-value leaf -> eight workgroups per CU
template <bool SV>
__global__ __launch_bounds__(256) void k_fleaf(const FRange *__restrict__ ranges, const LeafDesc *__restrict__ leaves,
                                               uint32_t n_leaves, uint32_t bm_words, const FSlot *__restrict__ slots, Key2 key2,
                                               uint32_t d, const Settings *__restrict__ sets, Block1 block1, unsigned long long ti_lim,
                                               uint32_t *__restrict__ cand_pos,
                                               uint32_t *__restrict__ cand_stop, uint32_t *__restrict__ cand_bend,
                                               uint8_t *__restrict__ cand_meta, uint32_t *__restrict__ blk_max,
                                               uint32_t *__restrict__ flags) {
    static_assert(OB_BLOCK_LG == 11, "blk_max is indexed by candidate ordinal >> 11");
    extern __shared__ uint32_t lds_bm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x * 4 + wave;
    if (g >= n_leaves) return;                             // (no workgroup barrier below: waves are on their own)
    uint32_t *bm = lds_bm + (size_t)wave * (bm_words + LEAF_LIST / 2);
    uint16_t *list = reinterpret_cast<uint16_t *>(bm + bm_words);
    const LeafDesc D = leaves[g];
    const FSlot &SL = slots[D.slot];
    const Key key{key2.k0, key2.k1, SL.seq};
    const uint32_t leaf_id = D.leaf_id;                    // the generator counts leaves per contig
    const uint64_t L = SL.L;
    msim_record *recs = SL.recs;
    if (SV) {                                              // storage of the batch is concatenated, ordinals are per contig
        cand_pos += SL.cand_off; cand_stop += SL.cand_off; cand_bend += SL.cand_off; cand_meta += SL.cand_off;
        blk_max += SL.blk_off;
    }
    const uint32_t len = D.len;
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
        // a counter yields two draws (fast_math.h: leaf_draw): lane l takes draws base + 2 l and base + 2 l + 1 of the sequence
        const uint32_t base = done & ~1u, cnt = min(need - have, 128u - (done & 1u));
        const uint32_t ja = base + 2 * lane, jb = ja + 1;
        const bool va = ja >= done && ja < done + cnt, vb = jb >= done && jb < done + cnt;
        bool fa = false, fb = false;
        if (va || vb) {
            const U4 r = draw4(key, ja >> 1, leaf_id, TAG_POS);
            if (va) {
                const uint32_t v = (uint32_t)below(lo64(r), len), bit = 1u << (v & 31);
                fa = !(atomicOr(&bm[v >> 5], bit) & bit);
            }
            if (vb) {
                const uint32_t v = (uint32_t)below(hi64(r), len), bit = 1u << (v & 31);
                fb = !(atomicOr(&bm[v >> 5], bit) & bit);
            }
        }
        have += (uint32_t)__popcll(__ballot(fa)) + (uint32_t)__popcll(__ballot(fb));
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
    const uint32_t pos0 = D.pos0;                          // pos = start + v0 + v + d * (ord - cand_base)   (mod 2^32 throughout)
    const Settings *S = SV ? sets + D.set : nullptr;
    // what one candidate is: its draws (one counter: type + length, or the SNP's outcome) and its stores
    auto emit = [&](uint32_t v, uint32_t ord) -> uint32_t {
        const uint32_t pos = pos0 + v + d * ord;
        if (SV) {
            const Cand c = cand_draw(key, ord, pos, L, *S, block1.v, D.clip, ti_lim);
            cand_pos[ord] = pos;
            cand_stop[ord] = c.stop;
            cand_bend[ord] = c.bend;
            cand_meta[ord] = c.meta;
            return c.bend;
        } else {
            store_record(recs + ord, pos, pos, 0u, MSIM_SN, snp_outcome(key, ord, ti_lim));
            return 0u;
        }
    };
    if (m <= LEAF_LIST) {
        // A lane's words hold anything from none to a dozen of the leaf's values: the values go to an LDS list first (a few
        // instructions each), and the expensive part -- a Philox call and the stores per candidate -- runs balanced over the
        // lanes, neighbouring lanes writing neighbouring candidates (first version: 7-8 rounds per wave instead of 2-3).
        uint32_t lr = incl - cnt;
        for (uint32_t q = 0; q < W; q++) {
            const uint32_t wi = lane * W + q;
            unsigned long long w = word_at(wi);
            while (w) {
                list[lr++] = (uint16_t)(wi * 64 + (uint32_t)__builtin_ctzll(w));
                w &= w - 1;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j0 = 0; j0 < m; j0 += 64) {
            const uint32_t j = j0 + lane;
            uint32_t e = 0;
            if (j < m) e = emit(list[j], D.cand0 + j);
            if (SV) {                                      // per-block maxima of the blocked ends (k_fkeep's far field): the 64
                const uint32_t ord = D.cand0 + j;          // candidates of a round lie in at most two blocks
                const uint32_t b_lo = (D.cand0 + j0) >> 11;
                uint32_t e_lo = (j < m && (ord >> 11) == b_lo) ? e : 0u, e_hi = (j < m && (ord >> 11) != b_lo) ? e : 0u;
                for (int o = 32; o > 0; o >>= 1) {
                    e_lo = max(e_lo, (uint32_t)__shfl_down((int)e_lo, o, 64));
                    e_hi = max(e_hi, (uint32_t)__shfl_down((int)e_hi, o, 64));
                }
                if (lane == 0) {
                    if (e_lo) atomicMax(&blk_max[b_lo], e_lo);
                    if (e_hi) atomicMax(&blk_max[b_lo + 1], e_hi);
                }
            }
        }
    } else {
        uint32_t ord = D.cand0 + incl - cnt;
        for (uint32_t q = 0; q < W; q++) {
            const uint32_t wi = lane * W + q;
            unsigned long long w = word_at(wi);
            while (w) {
                const uint32_t e = emit(wi * 64 + (uint32_t)__builtin_ctzll(w), ord);
                if (SV) atomicMax(&blk_max[ord >> 11], e);
                ord++;
                w &= w - 1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ the orbit passes

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
                                                             uint32_t *__restrict__ blk_indep, const uint32_t *__restrict__ run_if) {
    if (run_if && !*run_if) return;                        // (k_fkeep did the pass)
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
                                                         uint32_t K, uint32_t nb, uint32_t *__restrict__ blk_in,
                                                         const uint32_t *__restrict__ run_if) {
    if (run_if && !*run_if) return;
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
//   VISIT = false: boundary pass.  kept = on the orbit and not dropped -> CAND_KEEP; !FINAL: cand_end2 := the span the record
//                  consumes (stop + 1 for a kept DE / DU / IV, else pos + 1), the visit pass's input.
//   VISIT = true : visit pass.  record = CAND_KEEP and on the orbit -> CAND_VISIT.
//   FINAL: CAND_VISIT marks the records; per-block counts (records, insert bases, length change) for k_fscan.
template <bool VISIT, bool FINAL>
__global__ __launch_bounds__(OB_THREADS) void k_forbit_mark(const uint32_t *__restrict__ cand_pos, const uint32_t *__restrict__ cand_bend,
                                                            uint32_t *__restrict__ cand_end2,
                                                            const uint32_t *__restrict__ cand_stop, uint8_t *__restrict__ cand_meta,
                                                            uint32_t K, const uint32_t *__restrict__ blk_in,
                                                            uint32_t *__restrict__ blk_nrec, uint32_t *__restrict__ blk_pool,
                                                            long long *__restrict__ blk_delta, uint32_t *__restrict__ kept_any,
                                                            const uint32_t *__restrict__ run_if) {
    if (run_if && !*run_if) return;
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
            meta &= (uint8_t)~(CAND_KEEP | CAND_VISIT);    // (a k_fkeep pass that gave up may have set some)
            const bool keep = mark[i] && !(meta & CAND_DROPPED);
            if (keep) meta |= CAND_KEEP;
            any = any || keep;
            rec = keep;
            if (!FINAL) {
                const uint32_t stop = cand_stop[base + i];
                cand_end2[base + i] = (keep && type_consumes(t)) ? (stop == 0xffffffffu ? stop : stop + 1) : pos[i] + 1;
            }
        } else {
            meta &= (uint8_t)~CAND_VISIT;
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

// ------------------------------------------------------------------------------------------------ the same passes, block-local
// What the orbit kernels above compute in three launches with 11 doubling rounds, for the shape real settings have: blocked
// ranges that reach a few candidates far.  A candidate is FREE when no earlier candidate's blocked end reaches it (running
// maximum of the ends <= its position): whatever happens before, it is examined unblocked.  Between two free candidates lies a
// CLUSTER of contested ones whose fate depends on the walk from the free candidate in front of them -- a few steps, done by
// that candidate's lane over LDS.  A block (2048 candidates) needs nothing from its predecessors but the state at its first
// candidate, which it re-derives itself: it loads the block before it too, the maximum of the ends over everything earlier
// comes from the per-block maxima (blk_max: written by whoever produced the ends), so the running maximum is EXACT over both
// blocks and the walk restarts at the last free candidate of the previous block (the ANCHOR).  No free candidate among the
// 2048 before a block (every one of them inside somebody's blocked range: megabase deletions at a high rate) -> *fallback is
// raised, nothing of the plan is used and it is replayed with the orbit kernels (plan_fast.hip).
template <bool VISIT, bool FINAL>
__global__ __launch_bounds__(OB_THREADS) void k_fkeep(const uint32_t *__restrict__ cand_pos, const uint32_t *__restrict__ cand_end,
                                                      const uint32_t *__restrict__ blk_max, uint32_t *__restrict__ cand_end2,
                                                      uint32_t *__restrict__ blk_max2, const uint32_t *__restrict__ cand_stop,
                                                      uint8_t *__restrict__ cand_meta, uint32_t K,
                                                      uint32_t *__restrict__ blk_nrec, uint32_t *__restrict__ blk_pool,
                                                      long long *__restrict__ blk_delta, uint32_t *__restrict__ kept_all,
                                                      const FSlot *__restrict__ slots, const uint32_t *__restrict__ blk_slot) {
    // (of the previous block only its last KEEP_HALO candidates sit in LDS -- the walk that enters this block starts at the
    //  previous block's last free candidate, a few candidates from its end; the running maxima over the whole previous block
    //  come from registers.  22 KB instead of 34: seven workgroups per CU instead of four.  An anchor further back than the
    //  halo is treated like no anchor at all: the orbit kernels replay the plan.)
    constexpr int KEEP_HALO = 512;
    __shared__ uint32_t pos_h[KEEP_HALO + OB_BLOCK + 1], E_h[KEEP_HALO + OB_BLOCK];
    __shared__ uint8_t onorb[OB_BLOCK];
    __shared__ uint32_t wmax[2][OB_THREADS / 64], wred[OB_THREADS / 64];
    __shared__ int32_t wanchor[OB_THREADS / 64];
    __shared__ uint32_t red_n[OB_THREADS / 64], red_p[OB_THREADS / 64];
    __shared__ long long red_d[OB_THREADS / 64];
    const uint32_t slot = blk_slot[blockIdx.x];             // which contig of the batch (one load: a search through the slot table
    const FSlot &SL = slots[slot];                          //  was five dependent round trips at the head of every workgroup)
    K = SL.K;
    cand_pos += SL.cand_off; cand_end += SL.cand_off; cand_stop += SL.cand_off; cand_meta += SL.cand_off;
    if (cand_end2) cand_end2 += SL.cand_off;
    blk_max += SL.blk_off;
    uint32_t *kept_any = kept_all + 4 * slot, *fallback = kept_any + (VISIT ? 2 : 1);
    const uint32_t b = blockIdx.x - SL.blk_off, base = b * OB_BLOCK, cnt = min((uint32_t)OB_BLOCK, K - base);
    const bool prev = b > 0;                                // (the previous block is always a full one)
    uint32_t *pos = pos_h + KEEP_HALO, *E = E_h + KEEP_HALO;   // in-block index i; the previous block's tail at i - 2048 >= -KEEP_HALO
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i0 = threadIdx.x * OB_ITEMS;             // a thread owns 8 consecutive candidates of each block
    // ---- loads; maximum of the ends over everything before the previous block
    uint32_t far = 0;
    if (b >= 2) for (uint32_t q = threadIdx.x; q + 1 < b; q += OB_THREADS) far = max(far, blk_max[q]);
    uint32_t my_p[OB_ITEMS], my_e[OB_ITEMS], pv_p[OB_ITEMS], pv_e[OB_ITEMS];
    // (a thread's 8 consecutive words as two 16-byte loads: eight dword loads per array touch every cache line eight times)
    auto load8 = [](const uint32_t *src, uint32_t *dst) {
        const uint4 a = *reinterpret_cast<const uint4 *>(src), b = *reinterpret_cast<const uint4 *>(src + 4);
        dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w; dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
    };
    const bool full = cnt == (uint32_t)OB_BLOCK;            // (storage of a contig starts at a multiple of 2048: aligned)
    if (full) { load8(cand_pos + base + i0, my_p); load8(cand_end + base + i0, my_e); }
    if (prev) { load8(cand_pos + base - OB_BLOCK + i0, pv_p); load8(cand_end + base - OB_BLOCK + i0, pv_e); }
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        const uint32_t i = i0 + q;
        if (!full) {
            my_p[q] = i < cnt ? cand_pos[base + i] : 0xffffffffu;
            my_e[q] = i < cnt ? cand_end[base + i] : 0u;
        }
        if (!prev) { pv_p[q] = 0u; pv_e[q] = 0u; }
        pos[i] = my_p[q]; E[i] = my_e[q];
        if (i >= (uint32_t)(OB_BLOCK - KEEP_HALO)) { pos_h[i - (OB_BLOCK - KEEP_HALO)] = pv_p[q]; E_h[i - (OB_BLOCK - KEEP_HALO)] = pv_e[q]; }
    }
    for (int o = 32; o > 0; o >>= 1) far = max(far, (uint32_t)__shfl_down((int)far, o, 64));
    uint32_t rp = 0, rm = 0;                               // this thread's maxima of the ends: previous block, this block
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) { rp = max(rp, pv_e[q]); rm = max(rm, my_e[q]); }
    uint32_t ip = rp, im = rm;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t xp = __shfl_up(ip, o, 64), xm = __shfl_up(im, o, 64);
        if (lane >= o) { ip = max(ip, xp); im = max(im, xm); }
    }
    uint32_t exp_ = __shfl_up(ip, 1, 64), exm = __shfl_up(im, 1, 64);
    if (lane == 0) { exp_ = 0; exm = 0; wred[wave] = far; }
    if (lane == 63) { wmax[0][wave] = ip; wmax[1][wave] = im; }
    if (threadIdx.x == 0) pos[cnt] = 0xffffffffu;
    __syncthreads();
    uint32_t far_all = 0, prev_all = 0;
    for (int w = 0; w < OB_THREADS / 64; w++) { far_all = max(far_all, wred[w]); prev_all = max(prev_all, wmax[0][w]); }
    // running maximum in front of this thread's first candidate of the previous block / of this block
    uint32_t pm_p = max(far_all, exp_), pm_m = max(max(far_all, prev_all), exm);
    for (int w = 0; w < wave; w++) { pm_p = max(pm_p, wmax[0][w]); pm_m = max(pm_m, wmax[1][w]); }
    // ---- anchor: the last free candidate of the previous block
    int32_t anchor = -1;
    if (prev) {
#pragma unroll
        for (int q = 0; q < OB_ITEMS; q++) {
            if (pm_p <= pv_p[q]) anchor = (int32_t)(i0 + q);
            pm_p = max(pm_p, pv_e[q]);
        }
        for (int o = 32; o > 0; o >>= 1) anchor = max(anchor, __shfl_down(anchor, o, 64));
        if (lane == 0) wanchor[wave] = anchor;
    }
    bool free_[OB_ITEMS];
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        free_[q] = i0 + q < cnt && pm_m <= my_p[q];
        onorb[i0 + q] = free_[q] ? 1 : 0;
        pm_m = max(pm_m, my_e[q]);
    }
    __syncthreads();
    if (prev) {
        anchor = -1;
        for (int w = 0; w < OB_THREADS / 64; w++) anchor = max(anchor, wanchor[w]);
        if (anchor < OB_BLOCK - KEEP_HALO) {               // (uniform; none at all: -1) the orbit kernels take over
            if (threadIdx.x == 0) atomicOr(fallback, 1u);
            return;
        }
    }
    // ---- the walks: one lane per cluster
    auto walk = [&](int32_t j, uint32_t cur, uint32_t M) {  // from candidate j (in-block index, negative: previous block) on
        while (j < (int32_t)cnt && (j < 0 || M > pos[j])) {
            if (pos[j] >= cur) {
                if (j >= 0) onorb[j] = 1;
                cur = E[j];
            }
            M = max(M, E[j]);
            j++;
        }
    };
    if (threadIdx.x == 0 && prev) {                         // the cluster that ENTERS the block
        const int32_t f = anchor - OB_BLOCK;
        walk(f + 1, E[f], E[f]);
    }
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++)
        if (free_[q] && i0 + q + 1 < cnt && my_e[q] > pos[i0 + q + 1]) walk((int32_t)(i0 + q + 1), my_e[q], my_e[q]);
    __syncthreads();
    // ---- flags, the visit pass's input, counts
    uint32_t n_rec = 0, n_pool = 0, max2 = 0;
    long long delta = 0;
    bool any = false;
    uint32_t stops[OB_ITEMS];
    uint8_t metas[OB_ITEMS];
    if (full) {
        load8(cand_stop + base + i0, stops);
        const uint2 m8 = *reinterpret_cast<const uint2 *>(cand_meta + base + i0);
#pragma unroll
        for (int q = 0; q < OB_ITEMS; q++) metas[q] = (uint8_t)((q < 4 ? m8.x : m8.y) >> (8 * (q & 3)));
    } else {
#pragma unroll
        for (int q = 0; q < OB_ITEMS; q++) {
            stops[q] = i0 + q < cnt ? cand_stop[base + i0 + q] : 0u;
            metas[q] = i0 + q < cnt ? cand_meta[base + i0 + q] : (uint8_t)0;
        }
    }
    uint32_t e2s[OB_ITEMS];
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        const uint32_t i = i0 + q;
        e2s[q] = 0;
        if (i >= cnt) continue;
        uint8_t meta = metas[q];
        const uint32_t t = meta & 7u;
        const bool on = onorb[i] != 0;
        bool rec;
        if (!VISIT) {
            meta &= (uint8_t)~(CAND_KEEP | CAND_VISIT);
            const bool keep = on && !(meta & CAND_DROPPED);
            if (keep) meta |= CAND_KEEP;
            any = any || keep;
            rec = keep;
            if (!FINAL) {
                const uint32_t stop = stops[q];
                const uint32_t e2 = (keep && type_consumes(t)) ? (stop == 0xffffffffu ? stop : stop + 1) : my_p[q] + 1;
                e2s[q] = e2;
                max2 = max(max2, e2);
            }
        } else {
            meta &= (uint8_t)~CAND_VISIT;
            rec = (meta & CAND_KEEP) && on;
        }
        if (FINAL && rec) {
            meta |= CAND_VISIT;
            const uint32_t stop = stops[q];
            n_rec++;
            if (t == MSIM_IN) n_pool += stop - my_p[q] + 1;
            delta += cand_delta(t, my_p[q], stop);
        }
        metas[q] = meta;
    }
    if (full) {
        uint2 m8;
        m8.x = metas[0] | (metas[1] << 8) | (metas[2] << 16) | ((uint32_t)metas[3] << 24);
        m8.y = metas[4] | (metas[5] << 8) | (metas[6] << 16) | ((uint32_t)metas[7] << 24);
        *reinterpret_cast<uint2 *>(cand_meta + base + i0) = m8;
        if (!VISIT && !FINAL) {
            uint4 a, b2;
            a.x = e2s[0]; a.y = e2s[1]; a.z = e2s[2]; a.w = e2s[3]; b2.x = e2s[4]; b2.y = e2s[5]; b2.z = e2s[6]; b2.w = e2s[7];
            *reinterpret_cast<uint4 *>(cand_end2 + base + i0) = a;
            *reinterpret_cast<uint4 *>(cand_end2 + base + i0 + 4) = b2;
        }
    } else {
#pragma unroll
        for (int q = 0; q < OB_ITEMS; q++) {
            if (i0 + q >= cnt) continue;
            cand_meta[base + i0 + q] = metas[q];
            if (!VISIT && !FINAL) cand_end2[base + i0 + q] = e2s[q];
        }
    }
    if (!VISIT && any) *kept_any = 1u;
    if (FINAL) {
        for (int o = 32; o > 0; o >>= 1) {
            n_rec += __shfl_down(n_rec, o, 64);
            n_pool += __shfl_down(n_pool, o, 64);
            delta += __shfl_down(delta, o, 64);
        }
        if (lane == 0) { red_n[wave] = n_rec; red_p[wave] = n_pool; red_d[wave] = delta; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sn = 0, sp = 0;
            long long sd = 0;
            for (int w = 0; w < OB_THREADS / 64; w++) { sn += red_n[w]; sp += red_p[w]; sd += red_d[w]; }
            blk_nrec[blockIdx.x] = sn; blk_pool[blockIdx.x] = sp; blk_delta[blockIdx.x] = sd;
        }
    } else if (!VISIT) {                                    // the visit pass's per-block maxima
        for (int o = 32; o > 0; o >>= 1) max2 = max(max2, (uint32_t)__shfl_down((int)max2, o, 64));
        if (lane == 0) red_n[wave] = max2;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t m = 0;
            for (int w = 0; w < OB_THREADS / 64; w++) m = max(m, red_n[w]);
            blk_max2[blockIdx.x] = m;
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
        s.flags = f | (kept_any[0] ? (uint32_t)FF_KEPT_ANY : 0u);
        if (kept_any[1] | kept_any[2]) { s.n_rec = 0; s.out_len = (uint32_t)L; s.pool_len = 0; s.flags = FF_NEED_ORBIT; }
        *dyn = s;
    }
}

// ------------------------------------------------------------------------------------------------ records
// kept candidates -> record table (position order) + output offsets (apply.hip needs no scan) + insert pool + SNP outcomes.
// SCAN = true (up to 4096 blocks): no k_fscan in front -- every workgroup sums the raw counts of the blocks before it (a few KB
// from L2, cheaper than a dependent single-workgroup launch) and the LAST workgroup publishes the contig's DynSizes.
// SCAN = false: the block arrays hold offsets (k_fscan ran).
template <bool SCAN>
__global__ __launch_bounds__(OB_THREADS) void k_femit(const uint32_t *__restrict__ cand_pos, const uint32_t *__restrict__ cand_stop,
                                                      const uint8_t *__restrict__ cand_meta,
                                                      const uint32_t *__restrict__ blk_nrec, const uint32_t *__restrict__ blk_pool,
                                                      const long long *__restrict__ blk_delta, const uint32_t *__restrict__ kept_all,
                                                      uint32_t *__restrict__ flags, const FSlot *__restrict__ slots,
                                                      const uint32_t *__restrict__ blk_slot, Key2 key2) {
    __shared__ uint32_t wn[OB_THREADS / 64], wp[OB_THREADS / 64];
    __shared__ long long wd[OB_THREADS / 64];
    __shared__ uint32_t base_n, base_p;
    __shared__ long long base_d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t slot = blk_slot[blockIdx.x];             // which contig of the batch (see k_fkeep)
    const FSlot &SL = slots[slot];
    const Key key{key2.k0, key2.k1, SL.seq};
    const uint32_t K = SL.K, nb = SL.nb, blk = blockIdx.x - SL.blk_off;
    const uint64_t L = SL.L, out_cap = SL.out_cap, pool_cap = SL.pool_cap;
    const uint32_t *kept_any = kept_all + 4 * slot;
    DynSizes *dyn = SL.dyn;
    msim_record *recs = SL.recs;
    uint32_t *rec_off = SL.rec_off;
    uint8_t *pool = SL.pool;
    cand_pos += SL.cand_off; cand_stop += SL.cand_off; cand_meta += SL.cand_off;
    blk_nrec += SL.blk_off; blk_pool += SL.blk_off; blk_delta += SL.blk_off;
    if (SCAN) {
        if (kept_any[1] | kept_any[2]) {                   // k_fkeep handed the pass over: this plan is replayed (plan_fast.hip)
            if (blk == nb - 1 && threadIdx.x == 0) {
                DynSizes s; s.n_rec = 0; s.out_len = (uint32_t)L; s.pool_len = 0; s.flags = FF_NEED_ORBIT;
                *dyn = s;
            }
            return;
        }
        uint32_t sn = 0, sp = 0;
        long long sd = 0;
        for (uint32_t b2 = threadIdx.x; b2 < blk; b2 += OB_THREADS) { sn += blk_nrec[b2]; sp += blk_pool[b2]; sd += blk_delta[b2]; }
        for (int o = 32; o > 0; o >>= 1) { sn += __shfl_down(sn, o, 64); sp += __shfl_down(sp, o, 64); sd += __shfl_down(sd, o, 64); }
        if (lane == 0) { wn[wave] = sn; wp[wave] = sp; wd[wave] = sd; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tn = 0, tp = 0;
            long long td = 0;
            for (int w = 0; w < OB_THREADS / 64; w++) { tn += wn[w]; tp += wp[w]; td += wd[w]; }
            base_n = tn; base_p = tp; base_d = td;
            if (blk == nb - 1) {                    // totals = offsets of a block behind the last
                const unsigned long long pool_tot = (unsigned long long)tp + blk_pool[nb - 1];
                const long long out_len = (long long)L + td + blk_delta[nb - 1];
                uint32_t f = 0;
                if (pool_tot > pool_cap || pool_tot >= (1ull << 32)) f |= FF_POOL_OVERFLOW;
                if (out_len < 0 || (unsigned long long)out_len > out_cap || (unsigned long long)out_len >= (1ull << 32)) f |= FF_OUT_OVERFLOW;
                if (f) atomicOr(flags, f);
                DynSizes s;
                s.n_rec = f ? 0u : tn + blk_nrec[nb - 1];   // (an overflowing plan applies nothing)
                s.out_len = f ? (uint32_t)L : (uint32_t)out_len;
                s.pool_len = (uint32_t)pool_tot;
                s.flags = f | (kept_any[0] ? FF_KEPT_ANY : 0u);
                *dyn = s;
            }
        }
        __syncthreads();
    } else {
        if (dyn->flags & (FF_POOL_OVERFLOW | FF_OUT_OVERFLOW | FF_NEED_ORBIT)) return;
        if (threadIdx.x == 0) { base_n = blk_nrec[blk]; base_p = blk_pool[blk]; base_d = blk_delta[blk]; }
        __syncthreads();
    }
    const uint32_t i0 = blk * OB_BLOCK + threadIdx.x * OB_ITEMS;
    uint32_t pos[OB_ITEMS], stop[OB_ITEMS];
    uint8_t meta[OB_ITEMS];
    if (i0 + OB_ITEMS <= K) {                              // 8 consecutive candidates: 8 + 32 + 32 bytes in three / five loads
        const uint2 m8 = *reinterpret_cast<const uint2 *>(cand_meta + i0);
        const uint4 p0 = *reinterpret_cast<const uint4 *>(cand_pos + i0), p1 = *reinterpret_cast<const uint4 *>(cand_pos + i0 + 4);
        const uint4 s0 = *reinterpret_cast<const uint4 *>(cand_stop + i0), s1 = *reinterpret_cast<const uint4 *>(cand_stop + i0 + 4);
        const uint32_t pp[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w}, ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int q = 0; q < OB_ITEMS; q++) {
            meta[q] = (uint8_t)((q < 4 ? m8.x : m8.y) >> (8 * (q & 3)));
            pos[q] = pp[q]; stop[q] = ss[q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < OB_ITEMS; q++) {
            const bool in = i0 + q < K;
            meta[q] = in ? cand_meta[i0 + q] : (uint8_t)0;
            pos[q] = in ? cand_pos[i0 + q] : 0u;
            stop[q] = in ? cand_stop[i0 + q] : 0u;
        }
    }
    uint32_t nk = 0, np = 0;
    long long nd = 0;
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        if (meta[q] & CAND_VISIT) {
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
    __syncthreads();                                       // (wn / wp / wd are reused)
    if (lane == 63) { wn[wave] = in; wp[wave] = ip; wd[wave] = id; }
    __syncthreads();
    uint32_t r = base_n + in - nk, p = base_p + ip - np;
    long long shift = base_d + id - nd;
    for (int w = 0; w < wave; w++) { r += wn[w]; p += wp[w]; shift += wd[w]; }
    const uint32_t p_first = p;                            // (pool offset of this lane's first insertion)
    // Insert bases (mutator.py:465-471) are NOT written from this loop: an insertion is one candidate in eight, so a lane-per-
    // candidate loop would run its Philox call and a dozen stores in every one of its eight rounds for a handful of lanes
    // (measured: 286 of the kernel's 426 us).  The workgroup's insertions go to an LDS list and are filled one lane each.
    // (half a block's worth of list -- 12 KB, eight workgroups per CU; a block with more insertions than that, i.e. settings that
    //  draw mostly insertions, fills the rest from the emitting lanes themselves)
    constexpr uint32_t INS_LIST = OB_BLOCK / 2;
    __shared__ uint32_t ins_ord[INS_LIST], ins_at[INS_LIST], ins_len[INS_LIST];
    __shared__ uint32_t n_ins;
    auto fill_insert = [&](uint32_t ord, uint32_t at, uint32_t len) {
        uint8_t *dst0 = pool + at;
        for (uint32_t c0 = 0; c0 < len; c0 += 64) {                       // 64 bases per counter, 16 per word, stored 4 at a time
            const U4 ch = draw4(key, c0 >> 6, ord, TAG_INS);
            const uint32_t nbases = min(64u, len - c0);
            uint8_t *dst = dst0 + c0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const uint32_t word = w == 0 ? ch.x : w == 1 ? ch.y : w == 2 ? ch.z : ch.w;
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const uint32_t j = 16u * w + 4u * g4;
                    if (j >= nbases) break;
                    const uint32_t bits = (word >> (8 * g4)) & 0xffu;                        // four 2-bit codes
                    const uint32_t codes = (bits & 3u) | ((bits & 12u) << 6) | ((bits & 48u) << 12) | ((bits & 192u) << 18);
                    const uint32_t four = __builtin_amdgcn_perm(0u, 0x43475441u, codes);     // "ATGC"[code] per byte
                    if (j + 4 <= nbases) {
                        typedef uint32_t u32_a1 __attribute__((aligned(1)));
                        *reinterpret_cast<u32_a1 *>(dst + j) = four;
                    } else {
                        for (uint32_t x = j; x < nbases; x++) dst[x] = (uint8_t)(four >> (8 * (x - j)));
                    }
                }
            }
        }
    };
    if (threadIdx.x == 0) n_ins = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < OB_ITEMS; q++) {
        if (!(meta[q] & CAND_VISIT)) continue;
        const uint32_t t = meta[q] & 7u, ord = i0 + q;
        uint32_t extra = 0;
        const uint32_t aux = t == MSIM_SN ? (meta[q] >> CAND_AUX_SHIFT) & 3u : 0u;     // (drawn with the candidate: k_fleaf)
        if (t == MSIM_IN) {
            extra = p;
            const uint32_t len = stop[q] - pos[q] + 1;
            if ((unsigned long long)p + len <= pool_cap) {                // (an overflowing plan: flagged by the last workgroup)
                const uint32_t k = atomicAdd(&n_ins, 1u);
                if (k < INS_LIST) { ins_ord[k] = ord; ins_at[k] = p; ins_len[k] = len; }
            }
            p += len;
        }
        rec_off[r] = (uint32_t)((long long)pos[q] + shift);
        shift += cand_delta(t, pos[q], stop[q]);
        store_record(recs + r, pos[q], stop[q], extra, t, aux);
        r++;
    }
    __syncthreads();
    if (n_ins <= INS_LIST) {
        for (uint32_t k = threadIdx.x; k < n_ins; k += OB_THREADS) fill_insert(ins_ord[k], ins_at[k], ins_len[k]);
    } else {                                               // (the list overflowed: every lane fills its own, the list is ignored)
        uint32_t at = p_first;
#pragma unroll 1
        for (int q = 0; q < OB_ITEMS; q++) {
            if (!(meta[q] & CAND_VISIT) || (meta[q] & 7u) != MSIM_IN) continue;
            const uint32_t len = stop[q] - pos[q] + 1;
            if ((unsigned long long)at + len <= pool_cap) fill_insert(i0 + q, at, len);
            at += len;
        }
    }
}

}  // namespace
}  // namespace msim
