// GPU sampler (PLAN on the device): reproduces the reference's CPython MT19937 stream bit for bit
// for the stream structures that parallelise exactly.                       gfx950 (MI355X) only.
//
// What the reference does per contig (mutator.py:144-226, util.py:94-109, mutator.py:428-463):
//   random.sample(range(n), k)  -> first k DISTINCT accepted draws of  word >> (32 - bits) < n
//   sorted, + d * rank          -> candidate positions
//   numpy.random.choice(types)  -> a type per candidate (2 NumPy words each)
//   boundary pass               -> keep / drop, randint() for lengths
//   per kept SNP, in position order: uniform(0,1) [2 words], transversion: randbelow(2) [>= 1 word]
//
// All of it is a deterministic function of the word stream, so it can be evaluated out of order
// as long as the *stream positions* come out identical:
//   1. MT19937 words for thousands of 159 744-word chunks at once (jump-ahead polynomials,
//      tools/gen_mt_jump.py; k_mt_jump / k_mt_generate)
//   2. ordered compaction of the accepted draws (k_accept_count / k_scan_u32 / k_accept_scatter)
//   3. first-occurrence de-duplication with a bitmap in HBM: the first k accepted draws go in with
//      atomicOr; each duplicate found means exactly one more accepted draw is consumed, so the tail
//      converges in a few rounds (k_bitmap_insert / k_sample_tail) and yields the exact cut
//   4. the bitmap IS the sorted sample: popcount-rank + expand (k_bitmap_count / k_bitmap_expand)
//      -- no sort; position = start + value + d * rank, written straight into the record table
//   5. the SNP draws are a 3-state transducer over the following words (expect word 1 / expect
//      word 2 / inside randbelow(2)); its state maps compose associatively, so a scan over
//      per-thread maps gives every thread its true start state and record index
//      (k_snp_reduce / k_snp_scan / k_snp_emit)
//
// Eligible: ranges whose sample takes CPython's set path, whose type draw is deterministic SN and
// whose SNP block equals the sampling distance (nothing is ever blocked, no randint is drawn) --
// i.e. SNP-only ARGS runs such as BASELINE config 2.  Everything else goes through plan_host.cpp.
#include <algorithm>
#include <cmath>

#include "ctx.h"
#include "mt_jump_table.h"
#include "plan_gpu.h"

namespace msim {

namespace {

constexpr int GEN_STEP = MT_N - MT_M;                    // 227 words are independent per step
constexpr int JUMP_Z = MT_POLY_DEG + MT_N;               // raw words a jump convolves: 20561
constexpr int ACC_THREADS = 256;
constexpr int ACC_ITEMS = 8;
constexpr int ACC_BLOCK = ACC_THREADS * ACC_ITEMS;       // 2048 stream words per workgroup
constexpr int SNP_THREADS = 256;
constexpr int SNP_ITEMS = 16;
constexpr int SNP_BLOCK = SNP_THREADS * SNP_ITEMS;       // 4096 stream words per workgroup
constexpr int BM_THREADS = 256;

enum : uint32_t { FLAG_SAMPLE_OVERFLOW = 1u, FLAG_SNP_OVERFLOW = 2u };

// device-resident bookkeeping of one plan call
struct PlanState {
    unsigned long long pos;        // index into the raw word array of the next unconsumed word
    uint32_t flags;
    uint32_t dups;                 // duplicates found by the first-k insert
    uint32_t accepted_used;        // accepted draws consumed by the last sample
    uint32_t rsv;
};

// ------------------------------------------------------------------ 1. MT19937 in bulk
// state' = g(A) state : z = state followed by 19 937 more raw words, out[m] = XOR_{i in g} z[i+m].
// One jump is spread over JUMP_SPLIT workgroups (64 outputs each) so the early cascade levels, which
// have few source states, still fill the chip; inside a workgroup the four waves take every fourth
// polynomial limb (wave-uniform bit scan on the scalar unit, four LDS reads in flight per lane).
constexpr int JUMP_OUT = 64;
constexpr int JUMP_SPLIT = (MT_N + JUMP_OUT - 1) / JUMP_OUT;

constexpr int JUMP_THREADS = 1024;
constexpr int JUMP_WAVES = JUMP_THREADS / 64;

__global__ __launch_bounds__(JUMP_THREADS) void k_mt_jump(uint32_t *__restrict__ states, uint32_t n_src,
                                                          const uint32_t *__restrict__ poly) {
    __shared__ uint32_t z[JUMP_Z + 3];
    __shared__ uint32_t g[MT_POLY_WORDS];
    __shared__ uint32_t red[JUMP_THREADS];
    const uint32_t src = blockIdx.x;
    const int m0 = blockIdx.y * JUMP_OUT;
    const uint32_t *s = states + (size_t)src * MT_N;
    uint32_t *dst = states + (size_t)(src + n_src) * MT_N;
    for (int i = threadIdx.x; i < MT_N; i += JUMP_THREADS) z[i] = s[i];
    for (int i = threadIdx.x; i < MT_POLY_WORDS; i += JUMP_THREADS) g[i] = poly[i];
    __syncthreads();
    for (int base = 0; base < JUMP_Z - MT_N; base += GEN_STEP) {
        const int t = base + (int)threadIdx.x;
        if ((int)threadIdx.x < GEN_STEP && t + MT_N < JUMP_Z) z[t + MT_N] = mt_twist(z[t], z[t + 1], z[t + MT_M]);
        __syncthreads();
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = m0 + lane;
    uint32_t acc = 0;
    if (m < MT_N) {
        const uint32_t *zl = z + m;
        for (int j = wave; j < MT_POLY_WORDS; j += JUMP_WAVES) {
            uint32_t bits = __builtin_amdgcn_readfirstlane(g[j]);
            const uint32_t *zj = zl + j * 32;
            while (bits) {
                uint32_t v0, v1 = 0, v2 = 0, v3 = 0;
                v0 = zj[__builtin_ctz(bits)]; bits &= bits - 1;
                if (bits) { v1 = zj[__builtin_ctz(bits)]; bits &= bits - 1; }
                if (bits) { v2 = zj[__builtin_ctz(bits)]; bits &= bits - 1; }
                if (bits) { v3 = zj[__builtin_ctz(bits)]; bits &= bits - 1; }
                acc ^= (v0 ^ v1) ^ (v2 ^ v3);
            }
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < JUMP_OUT && m0 + (int)threadIdx.x < MT_N) {
        uint32_t r = 0;
#pragma unroll
        for (int w = 0; w < JUMP_WAVES; w++) r ^= red[threadIdx.x + 64 * w];
        dst[m0 + threadIdx.x] = r;
    }
}

// chunk j: raw words x[624 + j*S .. 624 + (j+1)*S) from state_j (the 624 words before the chunk).
// One wave per chunk: 227 words are mutually independent per step, a step's reads never touch the
// slots it overwrites, so the only ordering needed is "this step's writes before the next step's
// reads" -- a single-wave workgroup barrier.
__global__ __launch_bounds__(64) void k_mt_generate(const uint32_t *__restrict__ states,
                                                    uint32_t *__restrict__ raw, uint32_t first_chunk) {
    __shared__ uint32_t ring[1024];
    const uint32_t j = first_chunk + blockIdx.x;
    const uint32_t *s = states + (size_t)j * MT_N;
    uint32_t *out = raw + MT_N + (size_t)j * MT_CHUNK_WORDS;
    for (int i = threadIdx.x; i < MT_N; i += 64) ring[i] = s[i];
    __syncthreads();
    // word t of the chunk is sequence index 624 + t relative to the state: needs t, t+1, t+397
    for (int base = 0; base < MT_CHUNK_WORDS; base += GEN_STEP) {
        uint32_t v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int o = (int)threadIdx.x + 64 * q;
            const int t = base + o;
            v[q] = 0;
            if (o < GEN_STEP && t < MT_CHUNK_WORDS)
                v[q] = mt_twist(ring[t & 1023], ring[(t + 1) & 1023], ring[(t + MT_M) & 1023]);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int o = (int)threadIdx.x + 64 * q;
            const int t = base + o;
            if (o < GEN_STEP && t < MT_CHUNK_WORDS) {
                ring[(t + MT_N) & 1023] = v[q];
                out[t] = v[q];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ generic u32 exclusive scan
// in place over a[0..n), total to a[n]; single workgroup (n up to a few million is fine)
__global__ __launch_bounds__(1024) void k_scan_u32(uint32_t *__restrict__ a, uint32_t n) {
    __shared__ uint32_t buf[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? a[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const uint32_t t = threadIdx.x >= (unsigned)o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        const uint32_t incl = buf[threadIdx.x], c = carry;
        if (i < n) a[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) a[n] = carry;
}

// ------------------------------------------------------------------ 2. accepted draws, in order
__device__ __forceinline__ bool accepted(const uint32_t *__restrict__ raw, unsigned long long p,
                                         uint32_t shift, uint32_t n, uint32_t &v) {
    v = mt_temper(raw[p]) >> shift;                      // getrandbits(bits)
    return v < n;                                        // _randbelow: retry while r >= n
}

__global__ __launch_bounds__(ACC_THREADS) void k_accept_count(const uint32_t *__restrict__ raw,
                                                              const PlanState *ps, uint32_t W,
                                                              uint32_t shift, uint32_t n,
                                                              uint32_t *__restrict__ block_cnt,
                                                              PlanState *ps_rw) {
    __shared__ uint32_t red[ACC_THREADS / 64];
    if (blockIdx.x == 0 && threadIdx.x == 0) { ps_rw->dups = 0; ps_rw->accepted_used = 0; }   // new range
    const unsigned long long p0 = ps->pos;
    const uint32_t i0 = blockIdx.x * ACC_BLOCK + threadIdx.x * ACC_ITEMS;
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < ACC_ITEMS; q++) {
        uint32_t v;
        if (i0 + q < W && accepted(raw, p0 + i0 + q, shift, n, v)) c++;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(ACC_THREADS) void k_accept_scatter(const uint32_t *__restrict__ raw,
                                                                const PlanState *__restrict__ ps, uint32_t W,
                                                                uint32_t shift, uint32_t n,
                                                                const uint32_t *__restrict__ block_off,
                                                                uint32_t *__restrict__ acc) {
    __shared__ uint32_t part[ACC_THREADS];
    const unsigned long long p0 = ps->pos;
    const uint32_t i0 = blockIdx.x * ACC_BLOCK + threadIdx.x * ACC_ITEMS;
    uint32_t vals[ACC_ITEMS];
    uint32_t mask = 0, c = 0;
#pragma unroll
    for (int q = 0; q < ACC_ITEMS; q++) {
        uint32_t v = 0;
        const bool ok = i0 + q < W && accepted(raw, p0 + i0 + q, shift, n, v);
        vals[q] = v;
        if (ok) { mask |= 1u << q; c++; }
    }
    part[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < ACC_THREADS; o <<= 1) {
        const uint32_t t = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t w = block_off[blockIdx.x] + part[threadIdx.x] - c;
#pragma unroll
    for (int q = 0; q < ACC_ITEMS; q++)
        if (mask & (1u << q)) acc[w++] = vals[q];
}

// ------------------------------------------------------------------ 3. first-occurrence de-dup
__global__ __launch_bounds__(256) void k_bitmap_insert(const uint32_t *__restrict__ acc, uint32_t count,
                                                       uint32_t *__restrict__ bitmap, PlanState *__restrict__ ps) {
    uint32_t d = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
        const uint32_t v = acc[i];
        const uint32_t bit = 1u << (v & 31);
        const uint32_t old = atomicOr(&bitmap[v >> 5], bit);
        if (old & bit) d++;
    }
    for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
    if ((threadIdx.x & 63) == 0 && d) atomicAdd(&ps->dups, d);
}

// Tail rounds + exact cut.  One workgroup.  `total_acc` = accepted draws available in the window.
__global__ __launch_bounds__(1024) void k_sample_tail(const uint32_t *__restrict__ raw, const uint32_t *__restrict__ acc,
                                                      const uint32_t *__restrict__ block_off, uint32_t n_blocks,
                                                      uint32_t W, uint32_t shift, uint32_t n, uint32_t k,
                                                      uint32_t *__restrict__ bitmap, PlanState *__restrict__ ps) {
    __shared__ uint32_t red[16];
    __shared__ uint32_t s_need, s_pos, s_blk;
    const uint32_t total_acc = block_off[n_blocks];
    if (threadIdx.x == 0) { s_pos = k; s_need = ps->dups; }
    __syncthreads();
    if (total_acc < k) {                                  // window too small even for the first k
        if (threadIdx.x == 0) atomicOr(&ps->flags, FLAG_SAMPLE_OVERFLOW);
        return;
    }
    while (true) {
        const uint32_t need = s_need, pos = s_pos;
        if (need == 0) break;
        if (pos + need > total_acc) {
            if (threadIdx.x == 0) atomicOr(&ps->flags, FLAG_SAMPLE_OVERFLOW);
            return;
        }
        uint32_t d = 0;
        for (uint32_t i = threadIdx.x; i < need; i += 1024) {
            const uint32_t v = acc[pos + i];
            const uint32_t bit = 1u << (v & 31);
            const uint32_t old = atomicOr(&bitmap[v >> 5], bit);
            if (old & bit) d++;
        }
        for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
        __syncthreads();                                  // everyone has read s_need / s_pos
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int w = 0; w < 16; w++) t += red[w];
            s_pos = pos + need;
            s_need = t;
        }
        __syncthreads();
    }
    // A accepted draws were consumed; the stream cut is one past the word holding the A-th of them
    const uint32_t A = s_pos;
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = n_blocks;                   // last block with block_off[b] < A
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (block_off[mid] < A) lo = mid; else hi = mid;
        }
        s_blk = lo;
    }
    __syncthreads();
    const uint32_t b = s_blk;
    const uint32_t want = A - block_off[b];               // 1-based rank inside block b
    const unsigned long long p0 = ps->pos;
    // 1024 threads x 2 words cover the block's 2048 words in stream order
    uint32_t f[2], v;
    const uint32_t i0 = b * ACC_BLOCK + threadIdx.x * 2;
    f[0] = (i0 < W && accepted(raw, p0 + i0, shift, n, v)) ? 1u : 0u;
    f[1] = (i0 + 1 < W && accepted(raw, p0 + i0 + 1, shift, n, v)) ? 1u : 0u;
    const uint32_t mine = f[0] + f[1];
    // inclusive scan over the workgroup: ballot-free, via shuffles + LDS
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if ((threadIdx.x & 63) >= (unsigned)o) incl += t;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wave_off = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wave_off += red[w];
    incl += wave_off;
    const uint32_t excl = incl - mine;
    if (excl < want && want <= incl) {
        const uint32_t idx = (f[0] && excl + 1 == want) ? i0 : i0 + 1;
        ps->pos = p0 + idx + 1;
        ps->accepted_used = A;
    }
}

// ------------------------------------------------------------------ 4. bitmap -> sorted positions
__global__ __launch_bounds__(BM_THREADS) void k_bitmap_count(const uint64_t *__restrict__ bm, uint32_t n_words,
                                                             uint32_t *__restrict__ block_cnt) {
    __shared__ uint32_t red[BM_THREADS / 64];
    const uint32_t i = blockIdx.x * BM_THREADS + threadIdx.x;
    uint32_t c = i < n_words ? (uint32_t)__popcll(bm[i]) : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// record i of the range: pos = start + value + d * rank (util.py:104-109), type SN, stop = pos
__global__ __launch_bounds__(BM_THREADS) void k_bitmap_expand(const uint64_t *__restrict__ bm, uint32_t n_words,
                                                              const uint32_t *__restrict__ block_off,
                                                              uint32_t start, uint32_t d,
                                                              msim_record *__restrict__ recs) {
    __shared__ uint32_t part[BM_THREADS];
    const uint32_t i = blockIdx.x * BM_THREADS + threadIdx.x;
    uint64_t w = i < n_words ? bm[i] : 0;
    const uint32_t c = (uint32_t)__popcll(w);
    part[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < BM_THREADS; o <<= 1) {
        const uint32_t t = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t rank = block_off[blockIdx.x] + part[threadIdx.x] - c;
    while (w) {
        const uint32_t bit = (uint32_t)__builtin_ctzll(w);
        w &= w - 1;
        const uint32_t pos = start + (i * 64 + bit) + d * rank;
        msim_record r;
        r.pos = pos; r.stop = pos; r.extra = 0; r.type = MSIM_SN; r.aux = 0; r.rsv = 0;
        recs[rank] = r;
        rank++;
    }
}

// ------------------------------------------------------------------ 5. SNP ti/tv transducer
// states: 0 expect 1st uniform word, 1 expect 2nd (decides ti / tv), 2 inside randbelow(2)
struct SnpMap { uint32_t c[3]; uint32_t e; };            // per start state: emitted count, end state (2 bits each)

__device__ __forceinline__ SnpMap snp_compose(const SnpMap &f, const SnpMap &g) {   // f first, then g
    SnpMap r;
    r.e = 0;
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const uint32_t mid = (f.e >> (2 * s)) & 3;
        r.c[s] = f.c[s] + g.c[mid];
        r.e |= ((g.e >> (2 * mid)) & 3) << (2 * s);
    }
    return r;
}
__device__ __forceinline__ SnpMap snp_identity() { SnpMap r; r.c[0] = r.c[1] = r.c[2] = 0; r.e = 0 | (1 << 2) | (2 << 4); return r; }

// one step; returns new state, sets emit/aux
__device__ __forceinline__ uint32_t snp_step(uint32_t s, uint32_t prev, uint32_t cur, unsigned long long ti_lim,
                                             bool &emit, uint32_t &aux) {
    emit = false;
    if (s == 0) return 1;
    if (s == 1) {
        const unsigned long long m = ((unsigned long long)(prev >> 5) << 26) | (cur >> 6);
        if (m < ti_lim) { emit = true; aux = 0; return 0; }          // p <= p_ti: transition
        return 2;
    }
    if ((cur >> 31) == 0) { emit = true; aux = 1 + ((cur >> 30) & 1); return 0; }   // getrandbits(2) < 2
    return 2;
}

__device__ __forceinline__ SnpMap snp_thread_map(const uint32_t *u, uint32_t prev, int cnt, unsigned long long ti_lim) {
    SnpMap r;
    r.e = 0;
#pragma unroll
    for (int s0 = 0; s0 < 3; s0++) {
        uint32_t s = s0, c = 0, p = prev;
        for (int q = 0; q < cnt; q++) {
            bool emit; uint32_t aux;
            s = snp_step(s, p, u[q], ti_lim, emit, aux);
            c += emit ? 1u : 0u;
            p = u[q];
        }
        r.c[s0] = c;
        r.e |= s << (2 * s0);
    }
    return r;
}

// exclusive scan of per-thread maps inside the workgroup; returns this thread's prefix map and
// leaves the workgroup aggregate in total
__device__ __forceinline__ SnpMap snp_block_scan(const SnpMap &mine, SnpMap *lds, SnpMap &total) {
    lds[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < SNP_THREADS; o <<= 1) {
        SnpMap t = snp_identity();
        const bool on = threadIdx.x >= (unsigned)o;
        if (on) t = lds[threadIdx.x - o];
        __syncthreads();
        if (on) lds[threadIdx.x] = snp_compose(t, lds[threadIdx.x]);
        __syncthreads();
    }
    total = lds[SNP_THREADS - 1];
    SnpMap ex = snp_identity();
    if (threadIdx.x > 0) ex = lds[threadIdx.x - 1];
    __syncthreads();
    return ex;
}

__device__ __forceinline__ int snp_load(const uint32_t *__restrict__ raw, unsigned long long p0, uint32_t i0,
                                        uint32_t W, uint32_t *u, uint32_t &prev) {
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < SNP_ITEMS; q++) {
        u[q] = 0;
        if (i0 + q < W) { u[q] = mt_temper(raw[p0 + i0 + q]); cnt = q + 1; }
    }
    prev = i0 > 0 ? mt_temper(raw[p0 + i0 - 1]) : 0;
    return cnt;
}

__global__ __launch_bounds__(SNP_THREADS) void k_snp_reduce(const uint32_t *__restrict__ raw,
                                                            const PlanState *__restrict__ ps, uint32_t W,
                                                            unsigned long long ti_lim, SnpMap *__restrict__ block_maps) {
    __shared__ SnpMap lds[SNP_THREADS];
    const unsigned long long p0 = ps->pos;
    const uint32_t i0 = blockIdx.x * SNP_BLOCK + threadIdx.x * SNP_ITEMS;
    uint32_t u[SNP_ITEMS], prev;
    const int cnt = snp_load(raw, p0, i0, W, u, prev);
    const SnpMap mine = snp_thread_map(u, prev, cnt, ti_lim);
    SnpMap total;
    (void)snp_block_scan(mine, lds, total);
    if (threadIdx.x == 0) block_maps[blockIdx.x] = total;
}

// sequential-in-chunks scan of the workgroup maps; afterwards block_maps[b] = (state, count) at the
// start of block b when the stream starts in state 0: c[0] = count, e = state
__global__ __launch_bounds__(1024) void k_snp_scan(SnpMap *__restrict__ block_maps, uint32_t nb) {
    __shared__ SnpMap buf[1024];
    __shared__ uint32_t c_state, c_count;
    if (threadIdx.x == 0) { c_state = 0; c_count = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        SnpMap v = snp_identity();
        if (i < nb) v = block_maps[i];
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            SnpMap t = snp_identity();
            const bool on = threadIdx.x >= (unsigned)o;
            if (on) t = buf[threadIdx.x - o];
            __syncthreads();
            if (on) buf[threadIdx.x] = snp_compose(t, buf[threadIdx.x]);
            __syncthreads();
        }
        SnpMap ex = snp_identity();
        if (threadIdx.x > 0) ex = buf[threadIdx.x - 1];
        const uint32_t s0 = c_state, n0 = c_count;
        const SnpMap last = buf[1023];
        __syncthreads();
        if (i < nb) {
            SnpMap r;
            r.c[0] = n0 + ex.c[s0];
            r.c[1] = r.c[2] = 0;
            r.e = (ex.e >> (2 * s0)) & 3;
            block_maps[i] = r;
        }
        if (threadIdx.x == 0) {
            c_count = n0 + last.c[s0];
            c_state = (last.e >> (2 * s0)) & 3;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {                               // totals after the whole window
        SnpMap r;
        r.c[0] = c_count; r.c[1] = r.c[2] = 0; r.e = c_state;
        block_maps[nb] = r;
    }
}

__global__ __launch_bounds__(SNP_THREADS) void k_snp_emit(const uint32_t *__restrict__ raw, PlanState *__restrict__ ps,
                                                          unsigned long long p0, uint32_t W, unsigned long long ti_lim,
                                                          const SnpMap *__restrict__ block_maps,
                                                          msim_record *__restrict__ recs, uint32_t K) {
    __shared__ SnpMap lds[SNP_THREADS];
    const uint32_t i0 = blockIdx.x * SNP_BLOCK + threadIdx.x * SNP_ITEMS;
    uint32_t u[SNP_ITEMS], prev;
    const int cnt = snp_load(raw, p0, i0, W, u, prev);
    const SnpMap mine = snp_thread_map(u, prev, cnt, ti_lim);
    SnpMap total;
    const SnpMap ex = snp_block_scan(mine, lds, total);
    const uint32_t bs = block_maps[blockIdx.x].e, bc = block_maps[blockIdx.x].c[0];
    uint32_t s = (ex.e >> (2 * bs)) & 3;
    uint32_t idx = bc + ex.c[bs];
    uint32_t p = prev;
    for (int q = 0; q < cnt; q++) {
        bool emit; uint32_t aux;
        s = snp_step(s, p, u[q], ti_lim, emit, aux);
        p = u[q];
        if (emit) {
            if (idx < K) recs[idx].aux = (uint8_t)aux;
            idx++;
            if (idx == K) ps->pos = p0 + i0 + q + 1;      // the K-th SNP completed on this word
        }
    }
}

// single-lane epilogues: flag an undersized SNP window, then publish the bookkeeping block to the
// pinned host mailbox (a plain store over PCIe -- no blit-kernel D2H copy on the critical path)
__global__ void k_snp_check_publish(const SnpMap *__restrict__ block_maps, uint32_t nb, uint32_t K,
                                    PlanState *__restrict__ ps, PlanState *__restrict__ mailbox) {
    if (K && block_maps[nb].c[0] < K) ps->flags |= FLAG_SNP_OVERFLOW;
    *mailbox = *ps;
    __threadfence_system();
}
__global__ void k_publish(const PlanState *__restrict__ ps, PlanState *__restrict__ mailbox) {
    *mailbox = *ps;
    __threadfence_system();
}

__global__ void k_state_init(PlanState *ps, unsigned long long pos) {
    ps->pos = pos; ps->flags = 0; ps->dups = 0; ps->accepted_used = 0; ps->rsv = 0;
}

}  // namespace

// ====================================================================== host orchestration
struct GpuStream {
    uint32_t *d_raw = nullptr;          // x[0 .. cap)
    uint64_t cap = 0;
    uint32_t *d_states = nullptr;       // chunk start states
    uint32_t states_cap = 0;            // allocated states
    uint32_t n_states = 0;              // valid states (power of two once the cascade ran)
    uint32_t n_chunks = 0;              // chunks generated
    uint64_t pos = 0;                   // next unconsumed index into x (exact, host copy)
    uint64_t last_session_words = 0;    // words the previous (re)seeded session went through: sizing hint
    bool live = false;                  // device copy is the authoritative stream
};

struct GpuPlan {
    GpuStream s[2];
    uint32_t *d_poly = nullptr;
    PlanState *d_ps = nullptr;
    PlanState *h_mail = nullptr;        // pinned, device-visible mailbox
    // scratch
    uint32_t *d_acc = nullptr; size_t acc_cap = 0;
    uint32_t *d_cnt = nullptr; size_t cnt_cap = 0;
    uint32_t *d_bitmap = nullptr; size_t bm_cap = 0;     // bytes
    SnpMap *d_maps = nullptr; size_t maps_cap = 0;
    uint64_t reserve_words[2] = {0, 0};
};

static int grow(Ctx *c, void **p, size_t *cap, size_t want_bytes) {
    if (*cap >= want_bytes) return MSIM_OK;
    if (*p) MSIM_HIP(c, hipFree(*p));
    *p = nullptr; *cap = 0;
    const size_t sz = want_bytes + want_bytes / 4 + 4096;
    MSIM_HIP(c, hipMalloc(p, sz));
    *cap = sz;
    return MSIM_OK;
}

GpuPlan *gpu_plan_create() { return new GpuPlan(); }

void gpu_plan_destroy(GpuPlan *g) {
    if (!g) return;
    for (auto &s : g->s) { if (s.d_raw) (void)hipFree(s.d_raw); if (s.d_states) (void)hipFree(s.d_states); }
    if (g->d_poly) (void)hipFree(g->d_poly);
    if (g->d_ps) (void)hipFree(g->d_ps);
    if (g->h_mail) (void)hipHostFree(g->h_mail);
    if (g->d_acc) (void)hipFree(g->d_acc);
    if (g->d_cnt) (void)hipFree(g->d_cnt);
    if (g->d_bitmap) (void)hipFree(g->d_bitmap);
    if (g->d_maps) (void)hipFree(g->d_maps);
    delete g;
}

void gpu_plan_invalidate(GpuPlan *g) {
    for (auto &s : g->s) {
        if (s.live) s.last_session_words = std::max<uint64_t>(s.pos, MT_N + (uint64_t)s.n_chunks * MT_CHUNK_WORDS);
        s.live = false;
    }
}
void gpu_plan_reserve(GpuPlan *g, uint64_t py_words, uint64_t np_words) { g->reserve_words[0] = py_words; g->reserve_words[1] = np_words; }

// make x[0 .. upto) available on the device for stream `si`
static int ensure_words(Ctx *c, GpuPlan *g, int si, uint64_t upto) {
    GpuStream &s = g->s[si];
    const uint64_t have = MT_N + (uint64_t)s.n_chunks * MT_CHUNK_WORDS;
    if (upto <= have) return MSIM_OK;
    // batch the extension: an explicit hint, or what the previous session on this context needed
    upto = std::max<uint64_t>(upto, s.pos + g->reserve_words[si]);
    if (si == 0) upto = std::max<uint64_t>(upto, std::min<uint64_t>(s.last_session_words, upto * 64));
    const uint32_t need_chunks = (uint32_t)((upto - MT_N + MT_CHUNK_WORDS - 1) / MT_CHUNK_WORDS);
    uint32_t want_states = 1;
    int levels = 0;
    while (want_states < need_chunks) { want_states <<= 1; levels++; }
    if (levels > MT_JUMP_LEVELS) return fail(c, MSIM_ERR_UNSUPPORTED, "random stream longer than the jump table covers");
    if (!g->d_poly) {
        MSIM_HIP(c, hipMalloc(&g->d_poly, sizeof(MT_JUMP_POLY)));
        MSIM_HIP(c, hipMemcpyAsync(g->d_poly, MT_JUMP_POLY, sizeof(MT_JUMP_POLY), hipMemcpyHostToDevice, c->stream));
    }
    if (want_states > s.states_cap) {
        uint32_t *ns = nullptr;
        MSIM_HIP(c, hipMalloc(&ns, (size_t)want_states * MT_N * sizeof(uint32_t)));
        if (s.d_states) {
            MSIM_HIP(c, hipMemcpyAsync(ns, s.d_states, (size_t)s.n_states * MT_N * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
            MSIM_HIP(c, hipStreamSynchronize(c->stream));
            MSIM_HIP(c, hipFree(s.d_states));
        }
        s.d_states = ns;
        s.states_cap = want_states;
    }
    const uint64_t want_cap = MT_N + (uint64_t)need_chunks * MT_CHUNK_WORDS;
    if (want_cap > s.cap) {
        uint32_t *nr = nullptr;
        MSIM_HIP(c, hipMalloc(&nr, want_cap * sizeof(uint32_t)));
        if (s.d_raw) {
            MSIM_HIP(c, hipMemcpyAsync(nr, s.d_raw, have * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
            MSIM_HIP(c, hipStreamSynchronize(c->stream));
            MSIM_HIP(c, hipFree(s.d_raw));
        }
        s.d_raw = nr;
        s.cap = want_cap;
    }
    // cascade: level r turns states [0, 2^r) into [2^r, 2^(r+1))
    while (s.n_states < want_states) {
        int r = 0;
        while ((1u << r) < s.n_states) r++;
        hipLaunchKernelGGL(k_mt_jump, dim3(s.n_states, JUMP_SPLIT), dim3(JUMP_THREADS), 0, c->stream, s.d_states, s.n_states,
                           g->d_poly + (size_t)r * MT_POLY_WORDS);
        MSIM_HIP(c, hipGetLastError());
        s.n_states <<= 1;
    }
    if (need_chunks > s.n_chunks) {
        hipLaunchKernelGGL(k_mt_generate, dim3(need_chunks - s.n_chunks), dim3(64), 0, c->stream, s.d_states,
                           s.d_raw, s.n_chunks);
        MSIM_HIP(c, hipGetLastError());
        s.n_chunks = need_chunks;
    }
    return MSIM_OK;
}

// host generator -> device stream
static int stream_to_device(Ctx *c, GpuPlan *g, int si) {
    GpuStream &s = g->s[si];
    HostMT &h = si ? c->np : c->py;
    if (s.live) return MSIM_OK;
    if (!s.d_states) {
        MSIM_HIP(c, hipMalloc(&s.d_states, (size_t)MT_N * sizeof(uint32_t)));
        s.states_cap = 1;
    }
    if (!s.d_raw) {
        MSIM_HIP(c, hipMalloc(&s.d_raw, (size_t)MT_N * sizeof(uint32_t)));
        s.cap = MT_N;
    }
    MSIM_HIP(c, hipMemcpyAsync(s.d_states, h.mt, sizeof h.mt, hipMemcpyHostToDevice, c->stream));
    MSIM_HIP(c, hipMemcpyAsync(s.d_raw, h.mt, sizeof h.mt, hipMemcpyHostToDevice, c->stream));
    MSIM_HIP(c, hipStreamSynchronize(c->stream));       // h.mt may change right after
    s.n_states = 1;
    s.n_chunks = 0;
    s.pos = (uint64_t)h.idx;
    s.live = true;
    return MSIM_OK;
}

// device stream -> host generator (any 624-word window ending at pos is a valid state)
int gpu_plan_sync_to_host(Ctx *c, GpuPlan *g) {
    for (int si = 0; si < 2; si++) {
        GpuStream &s = g->s[si];
        if (!s.live) continue;
        HostMT &h = si ? c->np : c->py;
        const uint64_t consumed = s.pos;                  // index of the next word
        int rc = ensure_words(c, g, si, consumed);
        if (rc) return rc;
        if (consumed >= MT_N) {
            MSIM_HIP(c, hipMemcpyAsync(h.mt, s.d_raw + (consumed - MT_N), sizeof h.mt, hipMemcpyDeviceToHost, c->stream));
            h.idx = MT_N;
        } else {
            MSIM_HIP(c, hipMemcpyAsync(h.mt, s.d_raw, sizeof h.mt, hipMemcpyDeviceToHost, c->stream));
            h.idx = (int)consumed;
        }
        MSIM_HIP(c, hipStreamSynchronize(c->stream));
        s.last_session_words = std::max<uint64_t>(s.pos, MT_N + (uint64_t)s.n_chunks * MT_CHUNK_WORDS);
        s.live = false;
    }
    return MSIM_OK;
}

bool gpu_plan_eligible(const Ctx *c, const msim_range *ranges, int n_ranges) {
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);
    if (P.block[MSIM_SN] != d) return false;              // an SNP could block its successor
    for (int i = 0; i < n_ranges; i++) {
        const msim_range &r = ranges[i];
        if (r.k == 0) continue;                           // draws nothing (mutator.py:163-164)
        if (r.k < 4096) return false;                     // tiny ranges: the sequential host walk is faster
        const int64_t n = (r.stop - (r.k - 1) * d) - r.start;
        if (r.k < 0 || n < r.k) return false;             // ValueError: let the host planner raise it
        if (n <= r.setsize || n >= (1ll << 32)) return false;   // pool path / multi-word getrandbits
        if (r.start < 0 || r.stop >= (1ll << 32)) return false;
        // type draw must be deterministic SN: every threshold 0 or >= 2^53
        int zeros = 0;
        for (int j = 0; j < r.n_types; j++) {
            if (r.cdf_thr[j] == 0) zeros++;
            else if (r.cdf_thr[j] < (1ull << 53)) return false;
        }
        if (zeros >= r.n_types || r.types[zeros] != MSIM_SN) return false;
    }
    return true;                                          // (a contig that draws nothing is trivially fine)
}

int plan_contig_gpu(Ctx *c, GpuPlan *g, Contig &ct, const msim_range *ranges, int n_ranges) {
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);
    int rc;
    if ((rc = stream_to_device(c, g, 0))) return rc;
    if ((rc = stream_to_device(c, g, 1))) return rc;
    if (!g->d_ps) MSIM_HIP(c, hipMalloc(&g->d_ps, sizeof(PlanState)));
    if (!g->h_mail) MSIM_HIP(c, hipHostMalloc(&g->h_mail, sizeof(PlanState), hipHostMallocMapped));
    uint64_t K = 0;
    for (int i = 0; i < n_ranges; i++) K += (uint64_t)ranges[i].k;
    if (K >= (1ull << 31)) return fail(c, MSIM_ERR_UNSUPPORTED, "more than 2^31 mutations on one contig");
    ct.n_rec = K;
    ct.pool_len = 0;
    ct.plan_empty = K == 0;
    ct.all_snp = true;                                    // every record is an SNP: output offset == position
    if ((rc = dev_reserve(c, (void **)&ct.d_recs, &ct.cap_recs, std::max<uint64_t>(K, 1) * sizeof(msim_record)))) return rc;
    if ((rc = dev_reserve(c, (void **)&ct.d_pool, &ct.cap_pool, PAD))) return rc;
    const double p_tv = 1.0 - std::min(1.0, (double)P.ti_lim / 9007199254740992.0);

    GpuStream &py = g->s[0];
    const uint64_t pos_start = py.pos;
    for (int attempt = 0; attempt < 6; attempt++) {
        const double slack = (double)(1u << attempt);
        MSIM_HIP(c, hipEventRecord(c->ev0, c->stream));
        hipLaunchKernelGGL(k_state_init, dim3(1), dim3(1), 0, c->stream, g->d_ps, (unsigned long long)pos_start);
        uint64_t pos_hi = pos_start;                      // upper bound of the device position
        uint64_t rec_base = 0;
        for (int i = 0; i < n_ranges; i++) {
            const msim_range &r = ranges[i];
            if (r.k == 0) continue;
            const uint32_t k = (uint32_t)r.k;
            const uint64_t n = (uint64_t)((r.stop - (r.k - 1) * d) - r.start);
            const int bits = bit_length64(n);
            const double p_acc = (double)n / (double)(1ull << bits);
            // accepted draws needed ~ -n ln(1 - k/n) (coupon collector), words = that / p_acc
            const double need_acc = -(double)n * std::log1p(-(double)k / (double)n);
            const double target = need_acc + slack * (8.0 * std::sqrt(need_acc) + 2048.0);
            const double wd = target / p_acc + slack * (8.0 * std::sqrt(target) / p_acc + 4096.0);
            if (wd >= 4.0e9) return fail(c, MSIM_ERR_UNSUPPORTED, "sample window beyond 2^32 words");
            const uint32_t W = (uint32_t)wd;
            if ((rc = ensure_words(c, g, 0, pos_hi + W + 1))) return rc;
            const uint32_t nb = (W + ACC_BLOCK - 1) / ACC_BLOCK;
            if ((rc = grow(c, (void **)&g->d_cnt, &g->cnt_cap, (size_t)(std::max<uint64_t>(nb, (n + 63) / 64 / BM_THREADS + 1) + 2) * sizeof(uint32_t)))) return rc;
            if ((rc = grow(c, (void **)&g->d_acc, &g->acc_cap, (size_t)W * sizeof(uint32_t)))) return rc;
            const size_t bm_words64 = (size_t)((n + 63) / 64);
            if ((rc = grow(c, (void **)&g->d_bitmap, &g->bm_cap, bm_words64 * 8))) return rc;
            MSIM_HIP(c, hipMemsetAsync(g->d_bitmap, 0, bm_words64 * 8, c->stream));
            hipLaunchKernelGGL(k_accept_count, dim3(nb), dim3(ACC_THREADS), 0, c->stream, py.d_raw, g->d_ps, W,
                               (uint32_t)(32 - bits), (uint32_t)n, g->d_cnt, g->d_ps);
            hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, c->stream, g->d_cnt, nb);
            hipLaunchKernelGGL(k_accept_scatter, dim3(nb), dim3(ACC_THREADS), 0, c->stream, py.d_raw, g->d_ps, W,
                               (uint32_t)(32 - bits), (uint32_t)n, g->d_cnt, g->d_acc);
            hipLaunchKernelGGL(k_bitmap_insert, dim3(std::min<uint32_t>((k + 255) / 256, 256 * 8)), dim3(256), 0, c->stream,
                               g->d_acc, k, g->d_bitmap, g->d_ps);
            hipLaunchKernelGGL(k_sample_tail, dim3(1), dim3(1024), 0, c->stream, py.d_raw, g->d_acc, g->d_cnt, nb, W,
                               (uint32_t)(32 - bits), (uint32_t)n, k, g->d_bitmap, g->d_ps);
            // sorted positions straight into the record table
            const uint32_t bmw = (uint32_t)bm_words64;
            const uint32_t bnb = (bmw + BM_THREADS - 1) / BM_THREADS;
            hipLaunchKernelGGL(k_bitmap_count, dim3(bnb), dim3(BM_THREADS), 0, c->stream,
                               reinterpret_cast<const uint64_t *>(g->d_bitmap), bmw, g->d_cnt);
            hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, c->stream, g->d_cnt, bnb);
            hipLaunchKernelGGL(k_bitmap_expand, dim3(bnb), dim3(BM_THREADS), 0, c->stream,
                               reinterpret_cast<const uint64_t *>(g->d_bitmap), bmw, g->d_cnt, (uint32_t)r.start,
                               (uint32_t)d, ct.d_recs + rec_base);
            MSIM_HIP(c, hipGetLastError());
            pos_hi += W;
            rec_base += k;
        }
        // SNP draws of the whole contig, in position order
        uint32_t W2 = 0, nb2 = 0;
        if (K) {
            const double w2 = (double)K * (2.0 + 2.0 * p_tv) + slack * (8.0 * std::sqrt(4.0 * (double)K) + 8192.0);
            if (w2 >= 4.0e9) return fail(c, MSIM_ERR_UNSUPPORTED, "SNP draw window beyond 2^32 words");
            W2 = (uint32_t)w2;
            if ((rc = ensure_words(c, g, 0, pos_hi + W2 + 1))) return rc;
            nb2 = (W2 + SNP_BLOCK - 1) / SNP_BLOCK;
            if ((rc = grow(c, (void **)&g->d_maps, &g->maps_cap, (size_t)(nb2 + 1) * sizeof(SnpMap)))) return rc;
            hipLaunchKernelGGL(k_snp_reduce, dim3(nb2), dim3(SNP_THREADS), 0, c->stream, py.d_raw, g->d_ps, W2,
                               (unsigned long long)P.ti_lim, g->d_maps);
            hipLaunchKernelGGL(k_snp_scan, dim3(1), dim3(1024), 0, c->stream, g->d_maps, nb2);
        }
        hipLaunchKernelGGL(k_snp_check_publish, dim3(1), dim3(1), 0, c->stream, g->d_maps, nb2, (uint32_t)K, g->d_ps,
                           g->h_mail);
        MSIM_HIP(c, hipGetLastError());
        // the emit pass needs the position the sample phases ended at as a plain argument
        MSIM_HIP(c, hipStreamSynchronize(c->stream));
        PlanState h = *g->h_mail;
        if (!(h.flags & (FLAG_SAMPLE_OVERFLOW | FLAG_SNP_OVERFLOW)) && K) {
            hipLaunchKernelGGL(k_snp_emit, dim3(nb2), dim3(SNP_THREADS), 0, c->stream, py.d_raw, g->d_ps,
                               (unsigned long long)h.pos, W2, (unsigned long long)P.ti_lim, g->d_maps, ct.d_recs,
                               (uint32_t)K);
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, c->stream, g->d_ps, g->h_mail);
            MSIM_HIP(c, hipGetLastError());
        }
        MSIM_HIP(c, hipEventRecord(c->ev1, c->stream));
        MSIM_HIP(c, hipStreamSynchronize(c->stream));
        h = *g->h_mail;
        float ms = 0;
        MSIM_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->t.plan_gpu_ms += ms;
        if (h.flags & (FLAG_SAMPLE_OVERFLOW | FLAG_SNP_OVERFLOW)) continue;   // rare: widen the windows and redo
        c->t.py_words += h.pos - pos_start;
        c->t.np_words += 2 * K;
        py.pos = h.pos;
        g->s[1].pos += 2 * K;                             // numpy.random.choice(size=k): 2 words per candidate
        ct.planned = true;
        return MSIM_OK;
    }
    return fail(c, MSIM_ERR_HIP, "GPU sampler: stream window overflow persisted after 6 attempts");
}

}  // namespace msim
