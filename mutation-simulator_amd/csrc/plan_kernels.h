// Device kernels of the PLAN engines (included once, by plan_gpu.hip).            gfx950 (MI355X) only.
//
//   1. MT19937 in bulk: jump-ahead cascade (k_mt_extend / k_mt_jump) and chunk generation (k_mt_generate)
//   2. accepted draws of random.sample() in stream order (k_accept_* / k_bin_scatter)
//   3. first-occurrence de-duplication and the exact stream cut (k_bin_dedupe / k_sample_tail)
//   4. bitmap -> sorted positions (k_bitmap_count / k_bitmap_expand[_cand])
//   5. the SNP ti/tv transducer (k_snp_reduce / k_snp_scan_cut / k_snp_emit)
//   6. SV mixes: type draw, non-SNP compaction, keep flags, record compaction, insert pool
//      host-sampled contigs: word window, records from positions
// See plan_gpu.hip for what each engine enqueues and DESIGN.md section 3 for the reasoning.
#pragma once
#include <algorithm>

#include "ctx.h"
#include "mt_jump_table.h"

namespace msim {

namespace {

constexpr int GEN_STEP = MT_N - MT_M;                    // 227 words are independent per step
constexpr int JUMP_Z = MT_POLY_DEG + MT_N;               // raw words a jump convolves: 20561
constexpr int ACC_THREADS = 256;
constexpr int ACC_ITEMS = 8;
constexpr int ACC_BLOCK = ACC_THREADS * ACC_ITEMS;       // 2048 stream words per workgroup
constexpr int SNP_THREADS = 256;


constexpr int BM_THREADS = 256;

enum : uint32_t { FLAG_SAMPLE_OVERFLOW = 1u, FLAG_SNP_OVERFLOW = 2u };

// device-resident bookkeeping of one plan call
struct PlanState {
    unsigned long long pos;        // index into the raw word array of the next unconsumed word
    unsigned long long snp_base;   // pos at which the current contig's SNP draws start (= end of its samples)
    uint32_t flags;
    uint32_t dups;                 // duplicates found by the first-k insert
    uint32_t accepted_used;        // accepted draws consumed by the last sample
    uint32_t n_nsn;                // SV mixes: non-SNP candidates of the current range
    uint32_t n_rec, n_sn;          // SV mixes: kept mutations / kept SNPs of the current contig
    uint32_t pool_len;             // SV mixes: insert bases of the current contig
    long long len_delta;           // SV mixes: output length - input length of the current contig (kept IN/DU +len, DE -len)
    uint32_t ahead_margin_used;    // anchored windows: the largest |exact start - expected start| of a sample planned ahead, in permille
    uint32_t rsv;                  //   of the deviation the host allowed for (8 sigma + 256 words; k_ahead_fringe)
};

// ------------------------------------------------------------------ 1. MT19937 in bulk
// Jump-ahead: state' = g(A) state.  With z = the source state followed by 19 937 more raw words,
// out[m] = XOR_{i in g} z[i + m], m = 0..623 -- a GF(2) correlation of the 19 937-bit polynomial with z:
// 624 x ~9 970 word XORs per jump, ~1000 jumps per 3 Gb genome.
//
// The first version read one LDS word per (set bit, output) and spent ~16 instructions on each; it cost 1.75 ms of
// GPU time per genome and, worse, its 88 KB / 1024-thread workgroups blocked the chip for the first 2 ms of a step
// (rocprofv3 timeline, round 2).  This version is bound by the XORs themselves: a lane owns JUMP_OPL consecutive
// outputs and, per 32-bit polynomial limb, loads the window z[32 j + 10 l .. + 41] into REGISTERS once
// (21 ds_read_b64); every set bit b of the limb (wave-uniform: scalar branch) then costs JUMP_OPL register XORs
// acc[q] ^= W[b + q] -- one VALU instruction per (set bit, output) and 1/8 of the LDS traffic.  One workgroup per
// jump (z is staged once, not once per output split); its 16 waves take every 16th limb and are XOR-reduced
// through LDS.
constexpr int JUMP_THREADS = 1024;
constexpr int JUMP_WAVES = JUMP_THREADS / 64;
constexpr int JUMP_OPL = 10;                             // outputs per lane: 63 lanes x 10 cover the 624 outputs
constexpr int JUMP_WIN2 = (32 + JUMP_OPL) / 2;           // window of a limb, in 8-byte reads: 42 words >= 32 + 10 - 1
constexpr int JUMP_OUT_PAD = 64 * JUMP_OPL;              // 640
constexpr int JUMP_ZL = 32 * (MT_POLY_WORDS - 1) + JUMP_OUT_PAD + 2 * JUMP_WIN2 - JUMP_OPL;   // highest index read + 1
static_assert(MT_POLY_WORDS % JUMP_WAVES == 0, "limbs are dealt to the waves in whole rounds");
static_assert(MT_POLY_WORDS / JUMP_WAVES <= 64, "a wave keeps its limbs one per lane");

// A jump needs z = the source state followed by 19 937 more raw words.  That extension is a sequential
// recurrence (227 independent words per step, 88 steps); it is computed ONCE per source state (k_mt_extend,
// same scheme as chunk generation) into a z buffer which every jump from that source reads.
constexpr int JUMP_ZP = (JUMP_Z + 63) & ~63;             // z row pitch in words
constexpr int GEN_THREADS = 256;                         // one word per lane and step (227 of them active)

__global__ __launch_bounds__(GEN_THREADS) void k_mt_extend(const uint32_t *__restrict__ states, uint32_t *__restrict__ zbuf) {
    __shared__ uint32_t ring[1024];
    const uint32_t j = blockIdx.x;
    const uint32_t *s = states + (size_t)j * MT_N;
    uint32_t *out = zbuf + (size_t)j * JUMP_ZP;
    for (int i = threadIdx.x; i < MT_N; i += GEN_THREADS) { const uint32_t v = s[i]; ring[i] = v; out[i] = v; }
    __syncthreads();
    // word t beyond the state is sequence index 624 + t: needs t, t+1, t+397 (ring of 1024 >= 624 + 227 live words).
    // A step's writes [t+624] never touch the slots it reads, so one barrier per step orders everything.
    for (int base = 0; base < JUMP_Z - MT_N; base += GEN_STEP) {
        const int t = base + (int)threadIdx.x;
        if ((int)threadIdx.x < GEN_STEP && t + MT_N < JUMP_Z) {
            const uint32_t v = mt_twist(ring[t & 1023], ring[(t + 1) & 1023], ring[(t + MT_M) & 1023]);
            ring[(t + MT_N) & 1023] = v;
            out[t + MT_N] = v;
        }
        __syncthreads();
    }
}

// One cascade level: every source state j < n_src is advanced by mult * n_src chunks for mult = m_first,
// m_first + 1, ... (blockIdx.x / n_src selects the multiplier and with it the polynomial), giving the states of
// chunks j + mult * n_src.  All multipliers of a level are independent, hence one launch per level.
__global__ __launch_bounds__(JUMP_THREADS) void k_mt_jump(uint32_t *__restrict__ states, const uint32_t *__restrict__ zbuf,
                                                          uint32_t n_src, const uint32_t *__restrict__ poly_level,
                                                          uint32_t m_first) {
    __shared__ __attribute__((aligned(16))) uint32_t z[JUMP_ZL];
    __shared__ uint32_t red[JUMP_WAVES][JUMP_OUT_PAD];
    const uint32_t src = blockIdx.x % n_src;
    const uint32_t mult = m_first + blockIdx.x / n_src;
    const uint32_t *poly = poly_level + (size_t)(mult - 1) * MT_POLY_WORDS;
    const uint32_t *zs = zbuf + (size_t)src * JUMP_ZP;
    uint32_t *dst = states + ((size_t)src + (size_t)mult * n_src) * MT_N;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // this wave's limbs, one per lane: limb index wave + 16 * lane
    const uint32_t my_limb = lane < MT_POLY_WORDS / JUMP_WAVES ? poly[wave + JUMP_WAVES * lane] : 0u;
    for (int i = threadIdx.x; i < JUMP_ZL; i += JUMP_THREADS) z[i] = i < JUMP_Z ? zs[i] : 0u;
    __syncthreads();
    uint32_t acc[JUMP_OPL];
#pragma unroll
    for (int q = 0; q < JUMP_OPL; q++) acc[q] = 0;
    for (int it = 0; it < MT_POLY_WORDS / JUMP_WAVES; it++) {
        const uint32_t bits = __builtin_amdgcn_readlane(my_limb, it);        // wave-uniform
        if (bits == 0) continue;
        const int j = wave + JUMP_WAVES * it;
        const uint2 *wp = reinterpret_cast<const uint2 *>(z + 32 * j + JUMP_OPL * lane);   // 8-byte aligned
        uint32_t W[2 * JUMP_WIN2];
#pragma unroll
        for (int r = 0; r < JUMP_WIN2; r++) { const uint2 t = wp[r]; W[2 * r] = t.x; W[2 * r + 1] = t.y; }
#pragma unroll
        for (int b = 0; b < 32; b++) {
            if (bits & (1u << b)) {                                           // scalar branch
#pragma unroll
                for (int q = 0; q < JUMP_OPL; q++) acc[q] ^= W[b + q];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < JUMP_OPL; q++) red[wave][JUMP_OPL * lane + q] = acc[q];
    __syncthreads();
    if (threadIdx.x < MT_N) {
        uint32_t r = 0;
#pragma unroll
        for (int w = 0; w < JUMP_WAVES; w++) r ^= red[w][threadIdx.x];
        dst[threadIdx.x] = r;
    }
}

// chunk j: raw words x[624 + j*S .. 624 + (j+1)*S) from state_j (the 624 words before the chunk).
// One workgroup per chunk, one word per lane and step: 227 words are mutually independent per step and a
// step's reads never touch the slots it overwrites, so the only ordering needed is "this step's writes before
// the next step's reads" -- one barrier.  (The first version gave a chunk to a single wave, four words per
// lane and step: 312 us per batch whatever its size, the latency that decided when a step's first contig could
// start.  Four waves take a quarter of that.)
__global__ __launch_bounds__(GEN_THREADS) void k_mt_generate(const uint32_t *__restrict__ states,
                                                             uint32_t *__restrict__ raw, uint32_t first_chunk) {
    __shared__ uint32_t ring[1024];
    const uint32_t j = first_chunk + blockIdx.x;
    const uint32_t *s = states + (size_t)j * MT_N;
    uint32_t *out = raw + MT_N + (size_t)j * MT_CHUNK_WORDS;
    for (int i = threadIdx.x; i < MT_N; i += GEN_THREADS) ring[i] = s[i];
    __syncthreads();
    // word t of the chunk is sequence index 624 + t relative to the state: needs t, t+1, t+397
    for (int base = 0; base < MT_CHUNK_WORDS; base += GEN_STEP) {
        const int t = base + (int)threadIdx.x;
        if ((int)threadIdx.x < GEN_STEP && t < MT_CHUNK_WORDS) {
            const uint32_t v = mt_twist(ring[t & 1023], ring[(t + 1) & 1023], ring[(t + MT_M) & 1023]);
            ring[(t + MT_N) & 1023] = v;
            out[t] = v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ generic u32 exclusive scan
// in place over a[0..n), total to a[n]; single workgroup (these scans sit on the stream-position critical
// path, so: wave scan by shuffles, two barriers per round; a round covers 4096 items, or 16384 when more than
// one round of 4096 would be needed -- the loads of a round are independent, so the longer round costs one
// memory latency, not four)
template <int ITEMS, int WAVES = 16>
__device__ __forceinline__ void scan_rounds(uint32_t *__restrict__ a, uint32_t n, uint32_t *wsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += 64 * WAVES * ITEMS) {
        const uint32_t i0 = base + threadIdx.x * ITEMS;
        uint32_t v[ITEMS];
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < ITEMS; q++) { v[q] = i0 + q < n ? a[i0 + q] : 0; mine += v[q]; }
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t pre = carry, total = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            const uint32_t t = wsum[w];
            if (w < wave) pre += t;
            total += t;
        }
        uint32_t run = pre + incl - mine;
#pragma unroll
        for (int q = 0; q < ITEMS; q++) {
            if (i0 + q < n) a[i0 + q] = run;
            run += v[q];
        }
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) a[n] = carry;
}

__global__ __launch_bounds__(1024) void k_scan_u32(uint32_t *__restrict__ a, uint32_t n) {
    __shared__ uint32_t wsum[16];
    if (n <= 4096) scan_rounds<4>(a, n, wsum);
    else scan_rounds<16>(a, n, wsum);
}
// The same with four waves: a workgroup of 16 needs half a CU's wave slots at once, and next to a rewrite kernel that fills
// every slot it waits until four of that kernel's workgroups on ONE CU have ended with no refill in between
__global__ __launch_bounds__(256) void k_scan_u32_w4(uint32_t *__restrict__ a, uint32_t n) {
    __shared__ uint32_t wsum[4];
    scan_rounds<16, 4>(a, n, wsum);
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
                                                              PlanState *ps_rw, uint32_t *__restrict__ cursors,
                                                              uint32_t n_cursors) {
    __shared__ uint32_t red[ACC_THREADS / 64];
    if (blockIdx.x == 0 && threadIdx.x == 0) { ps_rw->dups = 0; ps_rw->accepted_used = 0; }   // new range
    // bin cursors of the scatter that follows (saves a fill-buffer dispatch on the critical path)
    for (uint32_t i = blockIdx.x * ACC_THREADS + threadIdx.x; i < n_cursors; i += gridDim.x * ACC_THREADS) cursors[i] = 0;
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
    // every lane issues its four atomics back to back (latency-bound otherwise), then counts
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x;
    uint32_t old[4], bit[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t i = base + 256 * q;
        old[q] = 0; bit[q] = 0;
        if (i < count) {
            const uint32_t v = acc[i];
            bit[q] = 1u << (v & 31);
            old[q] = atomicOr(&bitmap[v >> 5], bit[q]);
        }
    }
    uint32_t d = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) d += (old[q] & bit[q]) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
    if ((threadIdx.x & 63) == 0 && d) atomicAdd(&ps->dups, d);
}

// ---- locality-friendly de-dup: bins of 2^20 values, each de-duplicated in a 128 KB LDS bitmap ----
// inclusive prefix sum over the workgroup: shuffles inside a wave, one barrier across the waves (wsum: one word per wave; the
// caller puts a barrier between two uses of the same wsum)
__device__ __forceinline__ uint32_t wg_scan_incl(uint32_t v, uint32_t *wsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    for (int w = 0; w < wave; w++) incl += wsum[w];
    return incl;
}

constexpr int BIN_SHIFT = 20;
constexpr int BIN_VALUES = 1 << BIN_SHIFT;
constexpr int BIN_WORDS = BIN_VALUES / 32;               // 32768 x u32 = 128 KB
constexpr int MAX_BINS = 1024;                           // n <= 2^30; larger ranges use the global-atomic path
constexpr int BIN_SUBS = 16;                             // sub-cursors per bin: workgroup b fills sub-list b % 16,
                                                         // so a cursor sees 1/16 of the same-address atomics
constexpr int SPL_ITEMS = 32;
constexpr int SPL_BLOCK = ACC_THREADS * SPL_ITEMS;       // 8192 stream words per workgroup (= 4 count blocks)
constexpr int SPL_LDS = SPL_BLOCK + SPL_BLOCK / 32;

// Ordered accept (as k_accept_scatter) but the first k accepted draws go to their value bin (order
// inside a bin is irrelevant: only the SET of the first k matters), later ones to the ordered tail
// list.  A workgroup stages its 8192 words through row-padded LDS (coalesced loads, conflict-free
// per-lane runs), counting-sorts its accepted values by bin in LDS, reserves bin space with one
// atomicAdd per non-empty bin and writes every bin's run out contiguously.
__global__ __launch_bounds__(ACC_THREADS) void k_bin_scatter(const uint32_t *__restrict__ raw, const PlanState *ps,
                                                             uint32_t W, uint32_t shift, uint32_t n, uint32_t k,
                                                             const uint32_t *__restrict__ block_off, uint32_t n_bins,
                                                             uint32_t bin_cap, uint32_t *__restrict__ cursors,
                                                             uint32_t *__restrict__ bins, uint32_t *__restrict__ tail,
                                                             PlanState *ps_rw, uint32_t raw_counts) {
    // raw_counts: block_off holds k_accept_count's COUNTS, not their prefix sums -- every workgroup adds up the counts in front
    // of its first block itself (a few loads per lane), which takes the scan kernel and its launch off the chain
    __shared__ uint32_t stage[SPL_LDS];
    __shared__ uint32_t wsum[ACC_THREADS / 64], wsum2[ACC_THREADS / 64], wsum3[ACC_THREADS / 64];
    __shared__ uint32_t s_total;
    __shared__ uint32_t lhist[MAX_BINS];
    __shared__ uint32_t lbase[MAX_BINS + 1];
    __shared__ uint32_t gbase[MAX_BINS];
    for (uint32_t b = threadIdx.x; b < n_bins; b += ACC_THREADS) lhist[b] = 0;
    const unsigned long long p0 = ps->pos;
    const uint32_t base = blockIdx.x * SPL_BLOCK;
    {
        uint32_t w[SPL_ITEMS];
#pragma unroll
        for (int r = 0; r < SPL_ITEMS; r++) {
            const uint32_t idx = r * ACC_THREADS + threadIdx.x;
            w[r] = base + idx < W ? raw[p0 + base + idx] : 0u;
        }
#pragma unroll
        for (int r = 0; r < SPL_ITEMS; r++) {
            const uint32_t idx = r * ACC_THREADS + threadIdx.x;
            uint32_t v = 0xffffffffu;                    // rejected / out of window
            if (base + idx < W) { v = mt_temper(w[r]) >> shift; if (v >= n) v = 0xffffffffu; }
            stage[idx + (idx >> 5)] = v;
        }
    }
    __syncthreads();
    uint32_t vals[SPL_ITEMS];
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < SPL_ITEMS; q++) {
        vals[q] = stage[threadIdx.x * 33 + q];
        c += vals[q] != 0xffffffffu ? 1u : 0u;
    }
    uint32_t before;                                      // accepted draws in front of this workgroup's words
    if (raw_counts) {
        uint32_t t = 0;
        for (uint32_t i = threadIdx.x; i < blockIdx.x * (SPL_BLOCK / ACC_BLOCK); i += ACC_THREADS) t += block_off[i];
        const uint32_t incl = wg_scan_incl(t, wsum3);
        if (threadIdx.x == ACC_THREADS - 1) s_total = incl;
        __syncthreads();
        before = s_total;
    } else {
        before = block_off[blockIdx.x * (SPL_BLOCK / ACC_BLOCK)];
    }
    uint32_t a = before + wg_scan_incl(c, wsum) - c;      // accepted index
    uint32_t slot[SPL_ITEMS];
#pragma unroll
    for (int q = 0; q < SPL_ITEMS; q++) {
        slot[q] = 0xffffffffu;
        if (vals[q] != 0xffffffffu) {
            if (a < k) slot[q] = atomicAdd(&lhist[vals[q] >> BIN_SHIFT], 1u);
            else tail[a - k] = vals[q];
            a++;
        }
    }
    __syncthreads();
    // exclusive scan of the local histogram (<= 1024 bins: 4 per lane) + global space reservation
    {
        uint32_t h[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = threadIdx.x * 4 + q;
            h[q] = b < n_bins ? lhist[b] : 0;
            sum += h[q];
        }
        const uint32_t incl_h = wg_scan_incl(sum, wsum2);
        uint32_t run = incl_h - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = threadIdx.x * 4 + q;
            if (b < n_bins) {
                lbase[b] = run;
                gbase[b] = h[q] ? atomicAdd(&cursors[b * BIN_SUBS + (blockIdx.x & (BIN_SUBS - 1))], h[q]) : 0;
            }
            run += h[q];
        }
        if (threadIdx.x == ACC_THREADS - 1) lbase[MAX_BINS] = incl_h;   // total placed
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SPL_ITEMS; q++)
        if (slot[q] != 0xffffffffu) stage[lbase[vals[q] >> BIN_SHIFT] + slot[q]] = vals[q];
    __syncthreads();
    const uint32_t total = lbase[MAX_BINS];
    bool over = false;
    for (uint32_t i = threadIdx.x; i < total; i += ACC_THREADS) {
        const uint32_t v = stage[i];
        const uint32_t b = v >> BIN_SHIFT;
        const uint32_t at = gbase[b] + (i - lbase[b]);
        if (at < bin_cap) bins[((size_t)b * BIN_SUBS + (blockIdx.x & (BIN_SUBS - 1))) * bin_cap + at] = v;
        else over = true;
    }
    if (over) atomicOr(&ps_rw->flags, FLAG_SAMPLE_OVERFLOW);
}

// One workgroup per bin: LDS bitmap, LDS atomics, duplicate count, then the bitmap slice is written
// out with coalesced 16-B stores (so the global bitmap needs no memset).
__global__ __launch_bounds__(512) void k_bin_dedupe(const uint32_t *__restrict__ bins, const uint32_t *__restrict__ cursors,
                                                    uint32_t bin_cap, uint32_t *__restrict__ bitmap,
                                                    PlanState *__restrict__ ps) {
    __shared__ __attribute__((aligned(16))) uint32_t lbm[BIN_WORDS];
    __shared__ uint32_t red[8];
    const uint32_t b = blockIdx.x;
    uint4 *l4 = reinterpret_cast<uint4 *>(lbm);
    for (int i = threadIdx.x; i < BIN_WORDS / 4; i += 512) l4[i] = uint4{0, 0, 0, 0};
    __syncthreads();
    uint32_t d = 0;
    for (int sub = 0; sub < BIN_SUBS; sub++) {
        const uint32_t cnt = min(cursors[b * BIN_SUBS + sub], bin_cap);
        const uint32_t *mine = bins + ((size_t)b * BIN_SUBS + sub) * bin_cap;
        for (uint32_t i = threadIdx.x; i < cnt; i += 512) {
            const uint32_t v = mine[i] & (BIN_VALUES - 1);
            const uint32_t bit = 1u << (v & 31);
            const uint32_t old = atomicOr(&lbm[v >> 5], bit);
            d += (old & bit) ? 1u : 0u;
        }
    }
    for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 8; w++) t += red[w];
        if (t) atomicAdd(&ps->dups, t);
    }
    uint4 *g4 = reinterpret_cast<uint4 *>(bitmap + (size_t)b * BIN_WORDS);
    for (int i = threadIdx.x; i < BIN_WORDS / 4; i += 512) g4[i] = l4[i];
}

// ---- anchored windows (plan_gpu.hip: enqueue_sample_ahead).  Where a contig's sample STARTS is known exactly only when the
// chain of the contig before it has ended -- but the host knows it to within a few standard deviations of the words consumed
// since the last exact position: s in [lo, H].  So the heavy part of a sample (count, scatter, de-dup) runs OFF the chain on a
// window anchored at H, for the first k_core = K - a_max accepted draws from there (a_max bounds the accepted draws in [lo, H):
// whatever s turns out to be, those k_core draws are consumed), and the accepted draws of [lo, H) are compacted beside it.
// On the chain, once s is there: the accepted draws of [s, H) (h_acc of them) and the first need0 = K - D_core - h_acc draws of
// the ordered tail list are all consumed whatever their order (each adds at most one new value), so they go into the bitmap in
// parallel (k_ahead_fringe); the d1 values among them that were there already are what k_sample_tail's rounds still have to find.
struct SpecHdr { uint32_t need0, d1, rsv0, rsv1; };
constexpr int AHEAD_MAX_HEAD_BLOCKS = 1024;              // [lo, H) of at most 2 M words
// Off the chain, launch 1 of 3 (then k_bin_scatter with raw counts, k_bin_dedupe): blocks [0, nb) count the accepted draws of the
// core window's 2048-word blocks (k_accept_count, the window's origin as an argument); blocks [nb, nb + nbh) compact the accepted
// draws of the head interval's blocks, in order, each into its own 2048-entry segment.  Block 0 also resets the bookkeeping.
__global__ __launch_bounds__(ACC_THREADS) void k_ahead_count(const uint32_t *__restrict__ raw, PlanState *ps_core, PlanState *ps_head,
                                                             SpecHdr *hdr, unsigned long long H, unsigned long long lo, uint32_t W,
                                                             uint32_t F, uint32_t nb, uint32_t shift, uint32_t n,
                                                             uint32_t *__restrict__ block_cnt, uint32_t *__restrict__ cursors,
                                                             uint32_t n_cursors, uint32_t *__restrict__ hcnt,
                                                             uint32_t *__restrict__ hacc) {
    __shared__ uint32_t red[ACC_THREADS / 64];
    __shared__ uint32_t part[ACC_THREADS];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ps_core->pos = H; ps_core->snp_base = H; ps_core->flags = 0; ps_core->dups = 0; ps_core->accepted_used = 0;
        ps_head->pos = lo; ps_head->snp_base = lo; ps_head->flags = 0; ps_head->dups = 0; ps_head->accepted_used = 0;
        hdr->need0 = 0; hdr->d1 = 0;
    }
    if (blockIdx.x < nb) {
        for (uint32_t i = blockIdx.x * ACC_THREADS + threadIdx.x; i < n_cursors; i += nb * ACC_THREADS) cursors[i] = 0;
        const uint32_t i0 = blockIdx.x * ACC_BLOCK + threadIdx.x * ACC_ITEMS;
        uint32_t c = 0;
#pragma unroll
        for (int q = 0; q < ACC_ITEMS; q++) {
            uint32_t v;
            if (i0 + q < W && accepted(raw, H + i0 + q, shift, n, v)) c++;
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) block_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        return;
    }
    const uint32_t hb = blockIdx.x - nb;
    const uint32_t i0 = hb * ACC_BLOCK + threadIdx.x * ACC_ITEMS;
    uint32_t vals[ACC_ITEMS];
    uint32_t mask = 0, c = 0;
#pragma unroll
    for (int q = 0; q < ACC_ITEMS; q++) {
        uint32_t v = 0;
        const bool ok = i0 + q < F && accepted(raw, lo + i0 + q, shift, n, v);
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
    uint32_t w = hb * ACC_BLOCK + part[threadIdx.x] - c;
#pragma unroll
    for (int q = 0; q < ACC_ITEMS; q++)
        if (mask & (1u << q)) hacc[w++] = vals[q];
    if (threadIdx.x == ACC_THREADS - 1) hcnt[hb] = part[threadIdx.x];
}

constexpr int FRINGE_ITEMS = 4;
__global__ __launch_bounds__(ACC_THREADS) void k_ahead_fringe(const uint32_t *__restrict__ raw, PlanState *__restrict__ ps,
                                                              const PlanState *__restrict__ ps_core,
                                                              const PlanState *__restrict__ ps_head, SpecHdr *__restrict__ hdr,
                                                              unsigned long long lo, uint32_t F,
                                                              const uint32_t *__restrict__ hcnt, uint32_t nbh,
                                                              const uint32_t *__restrict__ hacc, const uint32_t *__restrict__ tail,
                                                              uint32_t K, uint32_t k_core, uint32_t shift, uint32_t n,
                                                              uint32_t *__restrict__ bitmap, unsigned long long expect,
                                                              uint32_t soft_half) {
    // expect / soft_half: the start the host expected and the half-width 8 sigma + 256 it allowed around it BEFORE the hard
    // bounds clipped the interval to [lo, lo + F] (a start at a hard bound is certain, not lucky): telemetry only
    __shared__ uint32_t red[ACC_THREADS / 64];
    __shared__ uint32_t hoff[AHEAD_MAX_HEAD_BLOCKS + 1];   // accepted draws of [s, H) in front of head block b
    __shared__ uint32_t wsum[ACC_THREADS / 64];
    const unsigned long long s = ps->pos;
    if (s < lo || s > lo + F) {                            // (uniform) the start fell outside the interval the host planned for
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&ps->flags, FLAG_SAMPLE_OVERFLOW);
        return;
    }
    const uint32_t rel = (uint32_t)(s - lo);
    if (blockIdx.x == 0 && threadIdx.x == 0 && soft_half) {   // telemetry: how much of the allowed deviation this start used
        const unsigned long long dev = s > expect ? s - expect : expect - s;
        atomicMax(&ps->ahead_margin_used, (uint32_t)min(1000000ull, 1000ull * dev / soft_half));
    }
    const uint32_t blk = min(rel / ACC_BLOCK, nbh);         // head block that holds s (nbh: s == H on a block border)
    uint32_t h_acc = 0, skip = 0;
    if (nbh) {
        // accepted draws of [block start, s): the entries of block blk's segment that do not belong to the sample
        {
            const uint32_t i0 = blk * ACC_BLOCK + threadIdx.x * ACC_ITEMS;
            uint32_t c = 0;
#pragma unroll
            for (int q = 0; q < ACC_ITEMS; q++) {
                uint32_t v;
                if (i0 + q < rel && accepted(raw, lo + i0 + q, shift, n, v)) c++;
            }
            for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
            __syncthreads();
            skip = red[0] + red[1] + red[2] + red[3];
        }
        // offsets of the blocks from blk on (four consecutive blocks per lane)
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = threadIdx.x * 4 + q;
            v[q] = (b >= blk && b < nbh) ? hcnt[b] - (b == blk ? skip : 0u) : 0u;
            sum += v[q];
        }
        uint32_t run = wg_scan_incl(sum, wsum) - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = threadIdx.x * 4 + q;
            if (b <= AHEAD_MAX_HEAD_BLOCKS) hoff[b] = run;
            run += v[q];
        }
        if (threadIdx.x == ACC_THREADS - 1) hoff[AHEAD_MAX_HEAD_BLOCKS] = run;
        __syncthreads();
        h_acc = hoff[AHEAD_MAX_HEAD_BLOCKS];
    }
    const uint32_t items = K - k_core + ps_core->dups;     // = h_acc + need0
    if (h_acc > items) {                                   // (uniform) more accepted draws in front of the anchor than a_max
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&ps->flags, FLAG_SAMPLE_OVERFLOW);
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr->need0 = items - h_acc;
        const uint32_t fl = ps_core->flags | ps_head->flags;
        if (fl) atomicOr(&ps->flags, fl);
    }
    uint32_t d = 0;
    const uint32_t stride = gridDim.x * ACC_THREADS;
    for (uint32_t base = blockIdx.x * ACC_THREADS + threadIdx.x; base < items; base += FRINGE_ITEMS * stride) {
        uint32_t old[FRINGE_ITEMS], bit[FRINGE_ITEMS];
#pragma unroll
        for (int q = 0; q < FRINGE_ITEMS; q++) {           // the atomics back to back, then their answers
            const uint32_t i = base + q * stride;
            old[q] = 0; bit[q] = 0;
            if (i < items) {
                uint32_t v;
                if (i < h_acc) {                           // head entry i: last block b >= blk with hoff[b] <= i
                    uint32_t lo_b = blk, hi_b = nbh;
                    while (hi_b - lo_b > 1) {
                        const uint32_t mid = (lo_b + hi_b) >> 1;
                        if (hoff[mid] <= i) lo_b = mid; else hi_b = mid;
                    }
                    v = hacc[lo_b * ACC_BLOCK + (i - hoff[lo_b]) + (lo_b == blk ? skip : 0u)];
                } else {
                    v = tail[i - h_acc];
                }
                if (v < n) {                               // (a list entry nobody wrote: the tail pass reports the short window)
                    bit[q] = 1u << (v & 31);
                    old[q] = atomicOr(&bitmap[v >> 5], bit[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < FRINGE_ITEMS; q++) d += (old[q] & bit[q]) ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
    if ((threadIdx.x & 63) == 0 && d) atomicAdd(&hdr->d1, d);
}

// Tail rounds + exact cut.  One workgroup.  `total_acc` = accepted draws available in the window.
// `acc` holds the accepted draws from accepted-index `acc_first` on (0: the whole list, k: tail only).
constexpr int TAIL_LDS_OFFS = 8192;
__global__ __launch_bounds__(1024) void k_sample_tail(const uint32_t *__restrict__ raw, const uint32_t *__restrict__ acc,
                                                      uint32_t acc_first,
                                                      const uint32_t *__restrict__ block_off, uint32_t n_blocks,
                                                      uint32_t W, uint32_t shift, uint32_t n, uint32_t k,
                                                      uint32_t *__restrict__ bitmap, PlanState *__restrict__ ps,
                                                      uint32_t raw_counts = 0, const PlanState *__restrict__ ps_win = nullptr,
                                                      const SpecHdr *__restrict__ hdr = nullptr,
                                                      unsigned long long pos_limit = ~0ull) {
    // raw_counts (n_blocks <= TAIL_LDS_OFFS): block_off holds k_accept_count's counts; their prefix sums are made here, in LDS
    // ps_win / hdr (anchored windows, see k_ahead_fringe): the window starts at ps_win->pos, not at the chain's position; `k`
    // is the core's share of the sample, the fringe pass has consumed hdr->need0 more accepted draws and left hdr->d1 to find
    __shared__ uint32_t red[16];
    __shared__ uint32_t s_need, s_pos, s_blk, s_total;
    __shared__ uint32_t loff[TAIL_LDS_OFFS + 1];
    const bool offs_in_lds = n_blocks <= TAIL_LDS_OFFS;
    uint32_t total_acc;
    if (raw_counts) {                                      // eight consecutive counts per lane, one workgroup scan
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = threadIdx.x * 8 + q;
            v[q] = i < n_blocks ? block_off[i] : 0;
            sum += v[q];
        }
        uint32_t run = wg_scan_incl(sum, red) - sum;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = threadIdx.x * 8 + q;
            if (i <= n_blocks) loff[i] = run;              // (entry n_blocks: the total)
            run += v[q];
        }
        if (threadIdx.x == 1023) s_total = run;
        __syncthreads();
        total_acc = s_total;
    } else {
        if (offs_in_lds)
            for (uint32_t i = threadIdx.x; i <= n_blocks; i += 1024) loff[i] = block_off[i];
        total_acc = block_off[n_blocks];
    }
    if (threadIdx.x == 0) { s_pos = hdr ? k + hdr->need0 : k; s_need = hdr ? hdr->d1 : ps->dups; }
    __syncthreads();
    if (total_acc < s_pos) {                              // window too small even for the first k
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
            const uint32_t v = acc[pos - acc_first + i];
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
            if ((offs_in_lds ? loff[mid] : block_off[mid]) < A) lo = mid; else hi = mid;
        }
        s_blk = lo;
    }
    __syncthreads();
    const uint32_t b = s_blk;
    const uint32_t want = A - (offs_in_lds ? loff[b] : block_off[b]);   // 1-based rank inside block b
    const unsigned long long p0 = ps_win ? ps_win->pos : ps->pos;
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
        unsigned long long cut = p0 + idx + 1;
        if (cut > pos_limit) { cut = pos_limit; atomicOr(&ps->flags, FLAG_SAMPLE_OVERFLOW); }   // (beyond what the host made sure exists)
        ps->pos = cut;
        ps->snp_base = cut;
        ps->accepted_used = A;
    }
}

// ------------------------------------------------------------------ 4. bitmap -> sorted positions
__device__ __forceinline__ void bitmap_count_body(const uint64_t *__restrict__ bm, uint32_t n_words,
                                                  uint32_t *__restrict__ block_cnt, uint32_t blk, uint32_t *red) {
    const uint32_t i = blk * BM_THREADS + threadIdx.x;
    uint32_t c = i < n_words ? (uint32_t)__popcll(bm[i]) : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blk] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(BM_THREADS) void k_bitmap_count(const uint64_t *__restrict__ bm, uint32_t n_words,
                                                             uint32_t *__restrict__ block_cnt) {
    __shared__ uint32_t red[BM_THREADS / 64];
    bitmap_count_body(bm, n_words, block_cnt, blockIdx.x, red);
}

// ---- the emission of several contigs in ONE launch per stage (SNP sampler, one drawing range per contig).  A contig's emission
// is a train of five small kernels (count, scan, outcomes, expansion, tile index) in front of its rewrite kernel; per contig they
// cost as much in launch gaps as in work, and that train -- not the chain -- bounded a c2 step.  The contigs whose chains have
// completed wait for each other in groups of EMIT_G and go through each stage together; jobs travel as kernel arguments.
constexpr int EMIT_G = 8;                                // jobs a launch can carry (the group size is chosen at run time)
struct SnpMap;
struct EmitJob {
    const uint64_t *bm; uint32_t *cnt2; msim_record *recs; uint8_t *aux8;
    const unsigned long long *base; const SnpMap *win_maps;
    uint32_t bmw, bnb, start, K, W2, nb2, blk0, eblk0;       // blk0 / eblk0: the job's first block in the bitmap / the emit grids
    // the three-launch train (k_snp_emit_count_b, k_bitmap_expand_tiles_b): cblk0 = the job's first SUPER-block in the count
    // grid; first / n_tiles / err: the contig's APPLY tile index and KeyError word, written by the expansion (first == nullptr:
    // the contig is not applied with its group)
    int32_t *first; unsigned long long *err;
    uint32_t cblk0, n_tiles;
};
struct EmitJobs { EmitJob j[EMIT_G]; uint32_t n, d, total_blk, total_eblk, total_cblk, tile_shift; };   // (tile = 1 << tile_shift bytes)
__device__ __forceinline__ uint32_t emit_job_of(const EmitJobs &J, uint32_t blk, bool emit_grid) {
    uint32_t k = 0;
    for (uint32_t q = 1; q < J.n; q++) if ((emit_grid ? J.j[q].eblk0 : J.j[q].blk0) <= blk) k = q;
    return k;
}
__global__ __launch_bounds__(BM_THREADS) void k_bitmap_count_b(EmitJobs J) {
    __shared__ uint32_t red[BM_THREADS / 64];
    const EmitJob &T = J.j[emit_job_of(J, blockIdx.x, false)];
    bitmap_count_body(T.bm, T.bmw, T.cnt2, blockIdx.x - T.blk0, red);
}
__global__ __launch_bounds__(256) void k_scan_u32_b(EmitJobs J) {           // one workgroup per job
    __shared__ uint32_t wsum[4];
    scan_rounds<16, 4>(J.j[blockIdx.x].cnt2, J.j[blockIdx.x].bnb, wsum);
}

// record i of the range: pos = start + value + d * rank (util.py:104-109), type SN, stop = pos
// aux8 (one-range contigs): the SNP outcomes by rank, left there by k_snp_emit_abs -- the records are then written once,
// complete, instead of being patched a byte each by the emit pass (a read-modify-write of every record's cache line)
__device__ __forceinline__ void bitmap_expand_body(const uint64_t *__restrict__ bm, uint32_t n_words,
                                                   const uint32_t *__restrict__ block_off, uint32_t start, uint32_t d,
                                                   msim_record *__restrict__ recs, const uint8_t *__restrict__ aux8,
                                                   uint32_t blk, uint32_t *wsum) {
    const uint32_t i = blk * BM_THREADS + threadIdx.x;
    uint64_t w = i < n_words ? bm[i] : 0;
    const uint32_t c = (uint32_t)__popcll(w);
    uint32_t incl = c;                                    // inclusive prefix: shuffles inside the wave, one barrier across waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if ((threadIdx.x & 63) >= (unsigned)o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    for (uint32_t q = 0; q < (threadIdx.x >> 6); q++) incl += wsum[q];
    uint32_t rank = block_off[blk] + incl - c;
    while (w) {
        const uint32_t bit = (uint32_t)__builtin_ctzll(w);
        w &= w - 1;
        const uint32_t pos = start + (i * 64 + bit) + d * rank;
        msim_record r;
        r.pos = pos; r.stop = pos; r.extra = 0; r.type = MSIM_SN; r.aux = aux8 ? aux8[rank] : (uint8_t)0; r.rsv = 0;
        recs[rank] = r;
        rank++;
    }
}
__global__ __launch_bounds__(BM_THREADS) void k_bitmap_expand(const uint64_t *__restrict__ bm, uint32_t n_words,
                                                              const uint32_t *__restrict__ block_off,
                                                              uint32_t start, uint32_t d,
                                                              msim_record *__restrict__ recs,
                                                              const uint8_t *__restrict__ aux8 = nullptr) {
    __shared__ uint32_t wsum[BM_THREADS / 64];
    bitmap_expand_body(bm, n_words, block_off, start, d, recs, aux8, blockIdx.x, wsum);
}
__global__ __launch_bounds__(BM_THREADS) void k_bitmap_expand_b(EmitJobs J) {
    __shared__ uint32_t wsum[BM_THREADS / 64];
    const EmitJob &T = J.j[emit_job_of(J, blockIdx.x, false)];
    bitmap_expand_body(T.bm, T.bmw, T.cnt2, T.start, J.d, T.recs, T.aux8, blockIdx.x - T.blk0, wsum);
}

// ---- the emission train in THREE launches (round 6; count, scan, outcomes, expansion, tile index, rewrite before):
//   k_snp_emit_count_b      the SNP outcomes by rank (snp_emit_body) and, in the same grid, the bitmap's popcounts: a workgroup per
//                           SUPER-block of 16 expansion blocks writes their 16 counts and the super-block's total -- exclusive
//                           owners, no atomics, nothing to zero
//   k_bitmap_expand_tiles_b the expansion; a workgroup makes its own rank base from the two levels (the totals of the super-blocks in
//                           front of its own + the counts of the blocks in front of it inside: <= 16 loads per lane for n = 2^30 --
//                           the scan launch is gone) and writes the APPLY tile index while it is there (below)
//   k_rewrite_snp_b         (apply.hip)
constexpr int EMIT_SUPER = 16;                            // expansion blocks per super-block (4096 bitmap words)
__device__ __forceinline__ void bitmap_count_super_body(const uint64_t *__restrict__ bm, uint32_t n_words,
                                                        uint32_t *__restrict__ block_cnt, uint32_t *__restrict__ super_cnt,
                                                        uint32_t sb, uint32_t *red) {
    // wave w takes blocks 4q + w of the super-block (q = 0 .. 3): four 64-word loads per lane and block, a wave reduction each
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t tot = 0;
#pragma unroll
    for (int q = 0; q < EMIT_SUPER / 4; q++) {
        const uint32_t blk = sb * EMIT_SUPER + q * 4 + wave;
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < BM_THREADS / 64; r++) {
            const uint32_t i = blk * BM_THREADS + r * 64 + lane;
            c += i < n_words ? (uint32_t)__popcll(bm[i]) : 0u;
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if (lane == 0) { block_cnt[blk] = c; tot += c; }    // (the count array holds whole super-blocks: plan_gpu.hip sizes it so)
    }
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) super_cnt[sb] = red[0] + red[1] + red[2] + red[3];
}

// first[t] = (number of records at or in front of position t * tile) - 1 for t = 0 .. n_tiles -- what k_tile_index finds by a binary
// search per tile over the finished table (apply.hip), here from the expansion's own registers: the positions a bitmap word's bits
// can take are the interval [start + 64 i + d rank_before(i), start + 64 (i + 1) + d rank_before(i + 1)), the intervals of
// consecutive words tile the axis, so every tile border lies in exactly one lane's interval (the first word's reaches down to 0, the
// last word's up to the last border), and that lane counts how many of its own bits lie at or in front of the border.
__device__ __forceinline__ void expand_tile_borders(uint64_t w, uint32_t i, uint32_t n_words, uint32_t start, uint32_t d,
                                                    uint32_t rank, uint32_t c, uint32_t tile_shift, uint32_t n_tiles,
                                                    int32_t *__restrict__ first) {
    const unsigned long long p_lo = i == 0 ? 0ull : (unsigned long long)start + 64ull * i + (unsigned long long)d * rank;
    const unsigned long long p_hi = i + 1 == n_words ? ~0ull
                                                     : (unsigned long long)start + 64ull * (i + 1) + (unsigned long long)d * (rank + c);
    unsigned long long t = (p_lo + ((1ull << tile_shift) - 1)) >> tile_shift;
    for (; t <= n_tiles && (t << tile_shift) < p_hi; t++) {
        const unsigned long long B = t << tile_shift;
        uint32_t below = 0, r = rank;
        uint64_t x = w;
        while (x) {
            const uint32_t bit = (uint32_t)__builtin_ctzll(x);
            if ((unsigned long long)start + 64ull * i + bit + (unsigned long long)d * r > B) break;
            x &= x - 1;
            below++; r++;
        }
        first[t] = (int32_t)(rank + below) - 1;
    }
}
__global__ __launch_bounds__(BM_THREADS) void k_bitmap_expand_tiles_b(EmitJobs J) {
    __shared__ uint32_t wsum[BM_THREADS / 64];
    __shared__ uint32_t bsum[BM_THREADS / 64];
    const EmitJob &T = J.j[emit_job_of(J, blockIdx.x, false)];
    const uint32_t blk = blockIdx.x - T.blk0;
    const uint32_t sb = blk / EMIT_SUPER, n_super = (T.bnb + EMIT_SUPER - 1) / EMIT_SUPER;
    const uint32_t *super_cnt = T.cnt2 + (size_t)n_super * EMIT_SUPER;
    // the block's rank base: super-block totals in front of its super-block + block counts in front of it inside
    uint32_t v = 0;
    for (uint32_t j = threadIdx.x; j < sb; j += BM_THREADS) v += super_cnt[j];
    if (threadIdx.x < blk % EMIT_SUPER) v += T.cnt2[sb * EMIT_SUPER + threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) bsum[threadIdx.x >> 6] = v;
    const uint32_t i = blk * BM_THREADS + threadIdx.x;
    uint64_t w = i < T.bmw ? T.bm[i] : 0;
    const uint32_t c = (uint32_t)__popcll(w);
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if ((threadIdx.x & 63) >= (unsigned)o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    for (uint32_t q = 0; q < (threadIdx.x >> 6); q++) incl += wsum[q];
    uint32_t rank = bsum[0] + bsum[1] + bsum[2] + bsum[3] + incl - c;
    if (T.first) {
        if (blk == 0 && threadIdx.x == 0) *T.err = ~0ull;  // the contig's KeyError word: none so far (the rewrite kernel follows)
        if (i < T.bmw) expand_tile_borders(w, i, T.bmw, T.start, J.d, rank, c, J.tile_shift, T.n_tiles, T.first);
    }
    while (w) {
        const uint32_t bit = (uint32_t)__builtin_ctzll(w);
        w &= w - 1;
        const uint32_t pos = T.start + (i * 64 + bit) + J.d * rank;
        msim_record r;
        r.pos = pos; r.stop = pos; r.extra = 0; r.type = MSIM_SN; r.aux = T.aux8[rank]; r.rsv = 0;
        T.recs[rank] = r;
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

// A lane owns 32 consecutive stream words, reduced to three bitmasks:
//   A bit i : words (i-1, i) as a uniform(0,1) sample decide "transition"  (p <= p_ti, mutator.py:438)
//   B bit i : word i ends a randbelow(2) loop (getrandbits(2) < 2, i.e. top bit clear)
//   T bit i : the transversion column word i would pick (bit 30)
// With them the transducer walks SNP by SNP (ctz over the masks) instead of word by word.
struct SnpBits { uint32_t A, B, T; int S, E; };          // words [S, E) of the lane's 32 are live

// Branch-free, bit-sliced transducer over one lane's <= 32 words.  Bit k (k = 0..2) of s0/s1/s2 says
// whether the simulation that STARTED in state k is currently in state 0/1/2, so all three start
// states advance together with a fixed sequence of bitwise ops per word (no divergence):
//   word i:  emit = (s1 & a_i) | (s2 & b_i);  s0' = emit;  s1' = s0;  s2' = (s1 & ~a_i) | (s2 & ~b_i)
// Emit counts are packed byte counters (<= 32 each).
__device__ __forceinline__ SnpMap snp_lane_map(const SnpBits &m) {
    uint32_t s0 = 1u, s1 = 2u, s2 = 4u, cnt = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const uint32_t live = (i >= m.S && i < m.E) ? 7u : 0u;   // words outside the window leave the state alone
        const uint32_t a = (0u - ((m.A >> i) & 1u)) & live;
        const uint32_t b = (0u - ((m.B >> i) & 1u)) & live;
        const uint32_t emit = (s1 & a) | (s2 & b);
        const uint32_t n2 = (s1 & ~a & live) | (s2 & ~b & live) | (s2 & ~live);
        const uint32_t n1 = (s0 & live) | (s1 & ~live);
        const uint32_t n0 = emit | (s0 & ~live);
        cnt += (emit | (emit << 7) | (emit << 14)) & 0x00010101u;
        s0 = n0; s1 = n1; s2 = n2;
    }
    SnpMap r;
    r.c[0] = cnt & 0xff; r.c[1] = (cnt >> 8) & 0xff; r.c[2] = (cnt >> 16) & 0xff;
    r.e = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t e = ((s0 >> k) & 1u) ? 0u : ((s1 >> k) & 1u) ? 1u : 2u;
        r.e |= e << (2 * k);
    }
    return r;
}

// One simulation (true start state st): emit mask (bit i: an SNP completes on word i) and the mask of
// emits that came out of the randbelow(2) loop (transversions).  Returns the end state.
__device__ __forceinline__ uint32_t snp_lane_emits(uint32_t st, const SnpBits &m, uint32_t &emits, uint32_t &from2) {
    uint32_t s0 = st == 0, s1 = st == 1, s2 = st == 2;
    emits = 0; from2 = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const uint32_t live = (i >= m.S && i < m.E) ? 1u : 0u;
        const uint32_t a = (m.A >> i) & live, b = (m.B >> i) & live;
        const uint32_t e1 = s1 & a, e2 = s2 & b;
        const uint32_t n2 = (s1 & ~a & live) | (s2 & ~b & live) | (s2 & ~live & 1u);
        const uint32_t n1 = (s0 & live) | (s1 & ~live & 1u);
        const uint32_t n0 = e1 | e2 | (s0 & ~live & 1u);
        emits |= (e1 | e2) << i;
        from2 |= e2 << i;
        s0 = n0; s1 = n1; s2 = n2;
    }
    return s0 ? 0u : s1 ? 1u : 2u;
}

constexpr int SNP_ITEMS2 = 32;
constexpr int SNP_BLOCK2 = SNP_THREADS * SNP_ITEMS2;     // 8192 stream words per workgroup
constexpr int SNP_LDS_WORDS = SNP_BLOCK2 + SNP_BLOCK2 / 32 + 1;

// Coalesced load of the workgroup's 8192 words (tempered) into LDS, row-padded (stride 33) so a
// lane's 32 consecutive words are conflict-free; then each lane builds its masks.
__device__ __forceinline__ SnpBits snp_stage(const uint32_t *__restrict__ raw, unsigned long long p0, uint32_t base,
                                             uint32_t W, unsigned long long ti_lim, uint32_t *sw, uint32_t start_off = 0) {
    uint32_t w[SNP_ITEMS2];
#pragma unroll
    for (int r = 0; r < SNP_ITEMS2; r++) {               // all 32 loads in flight before the first use
        const uint32_t idx = r * SNP_THREADS + threadIdx.x;
        w[r] = base + idx < W ? raw[p0 + base + idx] : 0u;
    }
#pragma unroll
    for (int r = 0; r < SNP_ITEMS2; r++) {
        const uint32_t idx = r * SNP_THREADS + threadIdx.x;
        sw[idx + (idx >> 5)] = base + idx < W ? mt_temper(w[r]) : 0u;
    }
    if (threadIdx.x == 0) sw[SNP_LDS_WORDS - 1] = base > 0 ? mt_temper(raw[p0 + base - 1]) : 0;
    __syncthreads();
    SnpBits m;
    m.A = m.B = m.T = 0;
    const uint32_t first = base + threadIdx.x * SNP_ITEMS2;
    m.E = first >= W ? 0 : (int)std::min<uint32_t>(SNP_ITEMS2, W - first);
    {   // words of the block before `start_off` (block-relative) are not part of the window
        const uint32_t lane0 = threadIdx.x * SNP_ITEMS2;
        m.S = start_off <= lane0 ? 0 : (int)std::min<uint32_t>(SNP_ITEMS2, start_off - lane0);
    }
    uint32_t prev = threadIdx.x ? sw[(threadIdx.x - 1) * 33 + 31] : sw[SNP_LDS_WORDS - 1];
    const uint32_t *mine = sw + threadIdx.x * 33;
#pragma unroll
    for (int i = 0; i < SNP_ITEMS2; i++) {
        const uint32_t w = mine[i];
        const unsigned long long u = ((unsigned long long)(prev >> 5) << 26) | (w >> 6);
        m.A |= (u < ti_lim ? 1u : 0u) << i;
        m.B |= ((w >> 31) ^ 1u) << i;
        m.T |= ((w >> 30) & 1u) << i;
        prev = w;
    }
    return m;
}

// What k_snp_maps_abs leaves behind per lane of every absolute block, besides the block's map: the lane's three masks and its
// own map (counts <= 32 and end states, packed).  Masks and maps depend on the words and on ti_lim only, so the kernels that
// follow -- k_snp_scan_cut_abs ON the chain, k_snp_emit_abs beside it -- load 16 bytes per lane instead of staging the lane's
// 32 words through LDS, tempering them and rebuilding the masks (round 4: the emit pass was the heaviest thing running beside
// the chain; without it a c2 step took 4.45 ms instead of 4.98).  Only where a window starts or ends inside a lane is the
// lane's map recomputed from its masks (bit operations, no loads).
struct SnpLane { uint32_t A, B, T, pm; };
__device__ __forceinline__ uint32_t snp_pack_map(const SnpMap &m) {
    return (m.c[0] & 63u) | ((m.c[1] & 63u) << 6) | ((m.c[2] & 63u) << 12) | ((m.e & 63u) << 18);
}
__device__ __forceinline__ SnpMap snp_unpack_map(uint32_t pm) {
    SnpMap r;
    r.c[0] = pm & 63u; r.c[1] = (pm >> 6) & 63u; r.c[2] = (pm >> 12) & 63u; r.e = (pm >> 18) & 63u;
    return r;
}
// the lane's masks with the live range of a window: block words [start_off, ..) and absolute words below W
__device__ __forceinline__ SnpBits snp_bits_of(const SnpLane &L, uint32_t base, uint32_t W, uint32_t start_off) {
    SnpBits m;
    m.A = L.A; m.B = L.B; m.T = L.T;
    const uint32_t first = base + threadIdx.x * SNP_ITEMS2;
    m.E = first >= W ? 0 : (int)min((uint32_t)SNP_ITEMS2, W - first);
    const uint32_t lane0 = threadIdx.x * SNP_ITEMS2;
    m.S = start_off <= lane0 ? 0 : (int)min((uint32_t)SNP_ITEMS2, start_off - lane0);
    return m;
}
__device__ __forceinline__ SnpMap snp_lane_map_of(const SnpLane &L, const SnpBits &m) {
    return (m.S == 0 && m.E == SNP_ITEMS2) ? snp_unpack_map(L.pm) : snp_lane_map(m);
}

// exclusive prefix of the lanes' maps across the workgroup (shuffles inside a wave, LDS across waves)
__device__ __forceinline__ SnpMap snp_shfl_up(const SnpMap &v, int o) {
    SnpMap r;
    r.c[0] = __shfl_up(v.c[0], o, 64); r.c[1] = __shfl_up(v.c[1], o, 64);
    r.c[2] = __shfl_up(v.c[2], o, 64); r.e = __shfl_up(v.e, o, 64);
    return r;
}
__device__ __forceinline__ SnpMap snp_block_scan2(const SnpMap &mine, SnpMap *wave_tot, SnpMap &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    SnpMap incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const SnpMap t = snp_shfl_up(incl, o);
        if (lane >= o) incl = snp_compose(t, incl);
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    SnpMap pre = snp_identity();
    for (int w = 0; w < wave; w++) pre = snp_compose(pre, wave_tot[w]);
    total = wave_tot[0];
    for (int w = 1; w < SNP_THREADS / 64; w++) total = snp_compose(total, wave_tot[w]);
    SnpMap ex = snp_shfl_up(incl, 1);
    if (lane == 0) ex = snp_identity();
    __syncthreads();
    return snp_compose(pre, ex);
}

// The map of a block depends on the stream words and on ti_lim only -- not on where a contig's SNP draws start.
// So the maps of ALL blocks of 8192 words, aligned to ABSOLUTE stream positions, are computed right behind chunk
// generation on the generation stream (one launch per batch, off every chain; the event the plan stream waits for
// anyway covers them).  On the chain only k_snp_scan_cut_abs is left: it maps the partial first block itself (the words
// before the start position are masked out), scans the precomputed maps and re-walks the block in which the K-th SNP
// completes.  Round 1 computed the maps per contig, relative to the start position, ON the chain (k_snp_reduce: 44 us
// of ~200 per contig).
__global__ __launch_bounds__(SNP_THREADS) void k_snp_maps_abs(const uint32_t *__restrict__ raw, uint32_t n_words,
                                                              unsigned long long ti_lim, uint32_t first_block,
                                                              SnpMap *__restrict__ abs_maps, SnpLane *__restrict__ lanes) {
    __shared__ uint32_t sw[SNP_LDS_WORDS];
    __shared__ SnpMap wave_tot[SNP_THREADS / 64];
    const uint32_t b = first_block + blockIdx.x;
    const SnpBits m = snp_stage(raw, 0ull, b * SNP_BLOCK2, n_words, ti_lim, sw);
    const SnpMap lm = snp_lane_map(m);
    lanes[(size_t)b * SNP_THREADS + threadIdx.x] = SnpLane{m.A, m.B, m.T, snp_pack_map(lm)};   // (only complete blocks are mapped)
    SnpMap total;
    (void)snp_block_scan2(lm, wave_tot, total);
    if (threadIdx.x == 0) abs_maps[b] = total;
}

// ONE workgroup, on the stream-position critical path.  Window = absolute blocks b0 .. b0 + nb - 1 where b0 holds
// the start position (its words before the start are masked).  Afterwards win_maps[j] = (state, count) at the start
// of window block j when the stream starts in state 0 (c[0] = count, e = state) -- what k_snp_emit_abs needs -- and
// the block in which the K-th SNP completes has been re-walked: the word on which it completes + 1 is the new
// stream position.  Also saves the start position for the emit pass.
__global__ __launch_bounds__(SNP_THREADS) void k_snp_scan_cut_abs(const SnpLane *__restrict__ lanes, PlanState *__restrict__ ps,
                                                                  uint32_t W,
                                                                  const SnpMap *__restrict__ abs_maps,
                                                                  SnpMap *__restrict__ win_maps, uint32_t nb_max, uint32_t K,
                                                                  unsigned long long *__restrict__ base_out,
                                                                  unsigned long long pos_limit = ~0ull) {
    __shared__ SnpMap wave_tot[SNP_THREADS / 64];
    __shared__ SnpMap s_first;
    __shared__ uint32_t s_blk, s_bs, s_bc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long p0 = ps->snp_base;
    const uint32_t b0 = (uint32_t)(p0 / SNP_BLOCK2), off = (uint32_t)(p0 % SNP_BLOCK2);
    const uint32_t nb = min(nb_max, (off + W + SNP_BLOCK2 - 1) / SNP_BLOCK2);
    const uint32_t w_end = (uint32_t)min<unsigned long long>(p0 + W, 0xffffffffull);       // words beyond the window are not live
    if (threadIdx.x == 0) { s_blk = 0xffffffffu; *base_out = p0; }
    {   // the partial first block, mapped here
        const SnpLane L0 = lanes[(size_t)b0 * SNP_THREADS + threadIdx.x];
        const SnpBits m0 = snp_bits_of(L0, b0 * SNP_BLOCK2, w_end, off);
        SnpMap total;
        (void)snp_block_scan2(snp_lane_map_of(L0, m0), wave_tot, total);
        if (threadIdx.x == 0) s_first = total;
    }
    __syncthreads();
    uint32_t c_state = 0, c_count = 0;                    // carried across chunks, identical in every lane
    for (uint32_t base = 0; base < nb; base += 4 * SNP_THREADS) {
        const uint32_t i0 = base + threadIdx.x * 4;
        SnpMap v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = i0 + q >= nb ? snp_identity() : (i0 + q == 0 ? s_first : abs_maps[b0 + i0 + q]);
        SnpMap incl = snp_compose(snp_compose(v[0], v[1]), snp_compose(v[2], v[3]));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const SnpMap t = snp_shfl_up(incl, o);
            if (lane >= o) incl = snp_compose(t, incl);
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        SnpMap run = snp_identity();                      // all blocks of this chunk before i0
        for (int w = 0; w < wave; w++) run = snp_compose(run, wave_tot[w]);
        SnpMap total = wave_tot[0];
        for (int w = 1; w < SNP_THREADS / 64; w++) total = snp_compose(total, wave_tot[w]);
        SnpMap ex = snp_shfl_up(incl, 1);
        if (lane == 0) ex = snp_identity();
        run = snp_compose(run, ex);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const SnpMap nxt = snp_compose(run, v[q]);
            if (i0 + q < nb) {
                SnpMap r;
                r.c[0] = c_count + run.c[c_state];
                r.c[1] = r.c[2] = 0;
                r.e = (run.e >> (2 * c_state)) & 3;
                win_maps[i0 + q] = r;
                // the block in which the K-th SNP completes: count-before < K <= count-after
                const uint32_t after = c_count + nxt.c[c_state];
                if (r.c[0] < K && K <= after) { s_blk = i0 + q; s_bs = r.e; s_bc = r.c[0]; }
            }
            run = nxt;
        }
        const uint32_t n_count = c_count + total.c[c_state], n_state = (total.e >> (2 * c_state)) & 3;
        __syncthreads();
        c_count = n_count;
        c_state = n_state;
    }
    if (threadIdx.x == 0) {                               // totals after the whole window
        win_maps[nb_max].c[0] = c_count;
        win_maps[nb_max].c[1] = nb;
        win_maps[nb_max].e = c_state;
        if (c_count < K) ps->flags |= FLAG_SNP_OVERFLOW;
    }
    if (c_count < K) return;                              // uniform
    const uint32_t b = s_blk, bs = s_bs, bc = s_bc;
    const uint32_t wbase = (b0 + b) * SNP_BLOCK2;         // absolute
    const SnpLane Lb = lanes[(size_t)(b0 + b) * SNP_THREADS + threadIdx.x];
    const SnpBits m = snp_bits_of(Lb, wbase, w_end, b == 0 ? off : 0u);
    SnpMap tot2;
    const SnpMap ex = snp_block_scan2(snp_lane_map_of(Lb, m), wave_tot, tot2);
    const uint32_t st = (ex.e >> (2 * bs)) & 3;
    const uint32_t idx = bc + ex.c[bs];                   // SNPs completed before this lane's words
    uint32_t emits, from2;
    (void)snp_lane_emits(st, m, emits, from2);
    const uint32_t mine = (uint32_t)__popc(emits);
    if (idx < K && K <= idx + mine) {                     // the K-th SNP completes in this lane: on which word?
        uint32_t e = emits;
        for (uint32_t q = idx + 1; q < K; q++) e &= e - 1; // drop the first K - idx - 1 emits
        const unsigned long long w0 = (unsigned long long)wbase + (unsigned long long)threadIdx.x * SNP_ITEMS2;
        const unsigned long long cut = w0 + (unsigned long long)__builtin_ctz(e) + 1;
        ps->pos = cut;
        // pos_limit: the host's bound of where these draws end -- what it lays the next stage's window out from (and makes sure
        // exists).  Beyond it the next stage would read words nobody generated: an overflow, never a silent short read.
        if (cut > pos_limit) atomicOr(&ps->flags, FLAG_SNP_OVERFLOW);
    }
}

// aux of every SNP record (off the critical path; runs on the emit stream).  Window block j = absolute block b0 + j.
__device__ __forceinline__ void snp_emit_body(const SnpLane *__restrict__ lanes, const unsigned long long *__restrict__ base_in,
                                              uint32_t W, const SnpMap *__restrict__ win_maps, uint32_t nb_max,
                                              msim_record *__restrict__ recs, uint32_t K, const uint32_t *__restrict__ sn_index,
                                              uint8_t *__restrict__ aux8, uint32_t blk, SnpMap *wave_tot) {
    if (blk >= win_maps[nb_max].c[1]) return;             // beyond the window (uniform)
    const uint32_t bc = win_maps[blk].c[0];
    if (bc >= K) return;                                  // window slack beyond the last SNP (uniform)
    const unsigned long long p0 = *base_in;
    const uint32_t b0 = (uint32_t)(p0 / SNP_BLOCK2), off = (uint32_t)(p0 % SNP_BLOCK2);
    const uint32_t w_end = (uint32_t)min<unsigned long long>(p0 + W, 0xffffffffull);
    const SnpLane Lm = lanes[(size_t)(b0 + blk) * SNP_THREADS + threadIdx.x];
    const SnpBits m = snp_bits_of(Lm, (b0 + blk) * SNP_BLOCK2, w_end, blk == 0 ? off : 0u);
    SnpMap total;
    const SnpMap ex = snp_block_scan2(snp_lane_map_of(Lm, m), wave_tot, total);
    const uint32_t bs = win_maps[blk].e;
    const uint32_t st = (ex.e >> (2 * bs)) & 3;
    uint32_t idx = bc + ex.c[bs];
    uint32_t emits, from2;
    (void)snp_lane_emits(st, m, emits, from2);
    while (emits && idx < K) {                            // aux: 0 transition, 1/2 transversion column (bit 30)
        const uint32_t i = (uint32_t)__builtin_ctz(emits);
        emits &= emits - 1;
        const uint8_t val = (uint8_t)(((from2 >> i) & 1u) ? 1u + ((m.T >> i) & 1u) : 0u);
        if (aux8) aux8[idx] = val;                        // (compact, by rank: k_bitmap_expand folds it into the record it writes)
        else recs[sn_index ? sn_index[idx] : idx].aux = val;
        idx++;
    }
}
__global__ __launch_bounds__(SNP_THREADS) void k_snp_emit_abs(const SnpLane *__restrict__ lanes,
                                                              const unsigned long long *__restrict__ base_in, uint32_t W,
                                                              const SnpMap *__restrict__ win_maps, uint32_t nb_max,
                                                              msim_record *__restrict__ recs, uint32_t K,
                                                              const uint32_t *__restrict__ sn_index,
                                                              uint8_t *__restrict__ aux8 = nullptr) {
    __shared__ SnpMap wave_tot[SNP_THREADS / 64];
    snp_emit_body(lanes, base_in, W, win_maps, nb_max, recs, K, sn_index, aux8, blockIdx.x, wave_tot);
}
__global__ __launch_bounds__(SNP_THREADS) void k_snp_emit_abs_b(const SnpLane *__restrict__ lanes, EmitJobs J) {
    __shared__ SnpMap wave_tot[SNP_THREADS / 64];
    const EmitJob &T = J.j[emit_job_of(J, blockIdx.x, true)];
    snp_emit_body(lanes, T.base, T.W2, T.win_maps, T.nb2, T.recs, T.K, nullptr, T.aux8, blockIdx.x - T.eblk0, wave_tot);
}
// ... and with the bitmap's popcounts in the same grid: blocks [0, total_eblk) are k_snp_emit_abs_b's, blocks behind them count one
// super-block each (two independent jobs of the emission train in ONE launch; SNP_THREADS == BM_THREADS)
static_assert(SNP_THREADS == BM_THREADS, "k_snp_emit_count_b runs both bodies with one block size");
__global__ __launch_bounds__(SNP_THREADS) void k_snp_emit_count_b(const SnpLane *__restrict__ lanes, EmitJobs J) {
    __shared__ SnpMap wave_tot[SNP_THREADS / 64];
    __shared__ uint32_t red[BM_THREADS / 64];
    if (blockIdx.x < J.total_eblk) {
        const EmitJob &T = J.j[emit_job_of(J, blockIdx.x, true)];
        snp_emit_body(lanes, T.base, T.W2, T.win_maps, T.nb2, T.recs, T.K, nullptr, T.aux8, blockIdx.x - T.eblk0, wave_tot);
        return;
    }
    const uint32_t cb = blockIdx.x - J.total_eblk;
    uint32_t k = 0;
    for (uint32_t q = 1; q < J.n; q++) if (J.j[q].cblk0 <= cb) k = q;
    const EmitJob &T = J.j[k];
    const uint32_t n_super = (T.bnb + EMIT_SUPER - 1) / EMIT_SUPER;
    bitmap_count_super_body(T.bm, T.bmw, T.cnt2, T.cnt2 + (size_t)n_super * EMIT_SUPER, cb - T.cblk0, red);
}

// ------------------------------------------------------------------ 6. SV mixes: candidates, types, filter
// A range whose type draw is not deterministic (insertions, deletions, duplications, inversions beside
// SNPs) still samples its positions as above; what changes is everything after the bitmap:
//   a. every candidate gets its type from the NumPy stream (2 words each, no rejection: parallel)
//   b. the NON-SNP candidates are compacted and handed to the host, which runs the boundary pass over
//      them -- the one stage that is a true sequential chain (plan_host.cpp: chain_boundary_host)
//   c. back on the device: an SNP is kept iff it lies outside the blocked range of the last kept
//      non-SNP before it (running maximum of the blocked-range ends), kept candidates are compacted
//      into the record table, insert bases come from the NumPy stream by prefix sum of the insert
//      lengths, and the kept SNPs' draws run through the transducer of section 5.
constexpr int CB_THREADS = 256;
constexpr int CB_ITEMS = 8;
constexpr int CB_BLOCK = CB_THREADS * CB_ITEMS;          // 2048 candidates per workgroup
constexpr uint8_t KEEP_BIT = 0x80;

struct TypeTable { unsigned long long thr[8]; uint32_t n; uint8_t type[8]; };   // msim_range.cdf_thr / .types
struct BlockTable { uint32_t p1[8]; };                   // block[t] + 1 (saturated), indexed by MSIM_* id

// exclusive prefix over the workgroup (sum / max); wsum: one word per wave
__device__ __forceinline__ uint32_t block_scan_add(uint32_t v, uint32_t *wsum, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t pre = 0;
    total = 0;
    for (int w = 0; w < CB_THREADS / 64; w++) {
        if (w < wave) pre += wsum[w];
        total += wsum[w];
    }
    __syncthreads();
    return pre + incl - v;
}
__device__ __forceinline__ uint32_t block_scan_max(uint32_t v, uint32_t *wsum, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl = max(incl, t);
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t pre = 0;
    total = 0;
    for (int w = 0; w < CB_THREADS / 64; w++) {
        if (w < wave) pre = max(pre, wsum[w]);
        total = max(total, wsum[w]);
    }
    uint32_t ex = __shfl_up(incl, 1, 64);
    if (lane == 0) ex = 0;
    __syncthreads();
    return max(pre, ex);
}

__device__ __forceinline__ long long block_scan_add64(long long v, long long *wsum, long long &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long long pre = 0;
    total = 0;
    for (int w = 0; w < CB_THREADS / 64; w++) {
        if (w < wave) pre += wsum[w];
        total += wsum[w];
    }
    __syncthreads();
    return pre + incl - v;
}

// bitmap -> candidate i of the range: pos = start + value + d * rank (util.py:104-109), type from
// numpy.random.choice(p=...) = searchsorted(cdf, u, 'right') on the 53-bit sample of words 2i, 2i+1
// (mutator.py:170-174)
__global__ __launch_bounds__(BM_THREADS) void k_bitmap_expand_cand(const uint64_t *__restrict__ bm, uint32_t n_words,
                                                                   const uint32_t *__restrict__ block_off,
                                                                   uint32_t start, uint32_t d,
                                                                   const uint32_t *__restrict__ np_raw,
                                                                   unsigned long long np_base, TypeTable tt,
                                                                   uint32_t *__restrict__ cand_pos,
                                                                   uint8_t *__restrict__ cand_type) {
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
        const uint32_t a = mt_temper(np_raw[np_base + 2ull * rank]);
        const uint32_t b = mt_temper(np_raw[np_base + 2ull * rank + 1]);
        const unsigned long long m = ((unsigned long long)(a >> 5) << 26) | (b >> 6);
        uint32_t idx = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) idx += ((uint32_t)j < tt.n && tt.thr[j] <= m) ? 1u : 0u;
        if (idx >= tt.n) idx = tt.n - 1;                 // unreachable: cdf[-1] == 1.0 > u
        cand_pos[rank] = start + (i * 64 + bit) + d * rank;
        cand_type[rank] = tt.type[idx];
        rank++;
    }
}

// (all != 0: every candidate counts -- contigs whose SNPs are on the boundary chain too, SNP block > sampling distance)
__global__ __launch_bounds__(CB_THREADS) void k_nsn_count(const uint8_t *__restrict__ cand_type, uint32_t k,
                                                          uint32_t *__restrict__ cnt, uint32_t all) {
    __shared__ uint32_t wsum[CB_THREADS / 64];
    const uint32_t i0 = blockIdx.x * CB_BLOCK + threadIdx.x * CB_ITEMS;
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < CB_ITEMS; q++)
        if (i0 + q < k && (all || cand_type[i0 + q] != MSIM_SN)) c++;
    uint32_t total;
    (void)block_scan_add(c, wsum, total);
    if (threadIdx.x == 0) cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(CB_THREADS) void k_nsn_scatter(const uint32_t *__restrict__ cand_pos,
                                                            const uint8_t *__restrict__ cand_type, uint32_t k,
                                                            const uint32_t *__restrict__ off, uint32_t n_blocks,
                                                            uint32_t *__restrict__ nsn_pos, uint8_t *__restrict__ nsn_type,
                                                            uint32_t *__restrict__ nsn_rank, PlanState *__restrict__ ps,
                                                            uint32_t all) {
    __shared__ uint32_t wsum[CB_THREADS / 64];
    const uint32_t i0 = blockIdx.x * CB_BLOCK + threadIdx.x * CB_ITEMS;
    uint8_t t[CB_ITEMS];
    bool on[CB_ITEMS];
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < CB_ITEMS; q++) {
        t[q] = i0 + q < k ? cand_type[i0 + q] : (uint8_t)MSIM_SN;
        on[q] = i0 + q < k && (all || t[q] != MSIM_SN);
        c += on[q] ? 1u : 0u;
    }
    uint32_t total;
    uint32_t j = off[blockIdx.x] + block_scan_add(c, wsum, total);
#pragma unroll
    for (int q = 0; q < CB_ITEMS; q++) {
        if (on[q]) {
            if (cand_pos) nsn_pos[j] = cand_pos[i0 + q];             // (the host-chain engine has no positions yet)
            nsn_type[j] = t[q];
            nsn_rank[j] = i0 + q;
            j++;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ps->n_nsn = off[n_blocks];
}

// tempered words of the window the boundary chain may consume (D2H staging)
__global__ __launch_bounds__(256) void k_temper_window(const uint32_t *__restrict__ raw, unsigned long long p0,
                                                       uint32_t n, uint32_t *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = mt_temper(raw[p0 + i]);
}

// "Next accepted draw" tables for the host walk (ctx.h: ChainWalk).  One thread per (word position, class):
// randint's retry loop, started at every position of the window at once.  A thread reads 1 / acceptance <= 2
// words on average, all from L2 (the window was generated moments ago).  Layout: position-major, the classes of
// a position side by side (1 << lg_rows slots), so that a prefix of the window is a contiguous block -- it is
// copied to the host in pieces while the host already walks the first ones.
__global__ __launch_bounds__(256) void k_accept_tables(const uint32_t *__restrict__ raw, unsigned long long p0, uint32_t n,
                                                       ChainClasses cc, uint32_t lg_rows, uint32_t *__restrict__ T) {
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = slot >> lg_rows, k = slot & ((1u << lg_rows) - 1);
    if (i > n || k >= cc.n) return;                        // entry n: the end-of-window sentinel
    const uint32_t sh = cc.sh[k], width = cc.width[k];
    uint32_t e = 0;
    const uint32_t end = min(n, i + CHAIN_TABLE_REACH);
    for (uint32_t q = i; q < end; q++) {
        const uint32_t v = mt_temper(raw[p0 + q]) >> sh;
        if (v < width) { e = ((q - i + 1) << lg_rows) << chain_value_bits(lg_rows) | v; break; }
    }
    T[slot] = e;
}

// the same from the CURRENT device position on (the host-chain engine: no round trip to learn it)
__global__ __launch_bounds__(256) void k_accept_tables_ps(const uint32_t *__restrict__ raw, const PlanState *__restrict__ ps,
                                                          uint32_t n, ChainClasses cc, uint32_t lg_rows, uint32_t *__restrict__ T) {
    const unsigned long long p0 = ps->pos;
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = slot >> lg_rows, k = slot & ((1u << lg_rows) - 1);
    if (i > n || k >= cc.n) return;
    const uint32_t sh = cc.sh[k], width = cc.width[k];
    uint32_t e = 0;
    const uint32_t end = min(n, i + CHAIN_TABLE_REACH);
    for (uint32_t q = i; q < end; q++) {
        const uint32_t v = mt_temper(raw[p0 + q]) >> sh;
        if (v < width) { e = ((q - i + 1) << lg_rows) << chain_value_bits(lg_rows) | v; break; }
    }
    T[slot] = e;
}

// ---- host-chain engine (plan_gpu.hip: plan_contig_gpu_multimix): several drawing ranges with their own settings
struct MixRangeDev { uint32_t rec_base, clip, set_id, rsv; };   // per drawing range: first candidate ordinal, stop + 1, settings
__device__ __forceinline__ uint32_t mix_range_of(const MixRangeDev *__restrict__ rt, uint32_t n_draw, uint32_t ordinal) {
    uint32_t lo = 0, hi = n_draw;                              // last range with rec_base <= ordinal
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rt[mid].rec_base <= ordinal) lo = mid; else hi = mid;
    }
    return lo;
}

// The type of candidate j is a function of j alone: NumPy words 2j, 2j+1 against the thresholds of the range that owns
// ordinal j (mutator.py:170-174 per range; the NumPy stream never rejects, so ordinals map to words directly) --
// known before any position exists, which is what lets the host walk sample and boundary pass in one go.
__global__ __launch_bounds__(CB_THREADS) void k_types_multi(const uint32_t *__restrict__ np_raw, unsigned long long np_base,
                                                            uint32_t K, const MixRangeDev *__restrict__ rt, uint32_t n_draw,
                                                            const TypeTable *__restrict__ sets, uint8_t *__restrict__ cand_type) {
    const uint32_t i0 = blockIdx.x * CB_BLOCK + threadIdx.x * CB_ITEMS;
    if (i0 >= K) return;
    uint32_t r = mix_range_of(rt, n_draw, i0);
    TypeTable tt = sets[rt[r].set_id];
    uint32_t next = r + 1 < n_draw ? rt[r + 1].rec_base : 0xffffffffu;
#pragma unroll
    for (int q = 0; q < CB_ITEMS; q++) {
        const uint32_t j = i0 + q;
        if (j >= K) break;
        while (j >= next) {
            r++;
            tt = sets[rt[r].set_id];
            next = r + 1 < n_draw ? rt[r + 1].rec_base : 0xffffffffu;
        }
        const uint32_t a = mt_temper(np_raw[np_base + 2ull * j]);
        const uint32_t b = mt_temper(np_raw[np_base + 2ull * j + 1]);
        const unsigned long long m = ((unsigned long long)(a >> 5) << 26) | (b >> 6);
        uint32_t idx = 0;
#pragma unroll
        for (int x = 0; x < 8; x++) idx += ((uint32_t)x < tt.n && tt.thr[x] <= m) ? 1u : 0u;
        if (idx >= tt.n) idx = tt.n - 1;                 // unreachable: cdf[-1] == 1.0 > u
        cand_type[j] = tt.type[idx];
    }
}

__global__ __launch_bounds__(256) void k_stop_scatter(const uint32_t *__restrict__ nsn_rank,
                                                      const uint32_t *__restrict__ nsn_stop, uint32_t n_nsn,
                                                      uint32_t *__restrict__ cand_stop) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < n_nsn) cand_stop[nsn_rank[j]] = nsn_stop[j];
}
// ... and what __link_tls decided for the chain's candidates (ctx.h: ch_extra / ch_aux), contigs with translocations
__global__ __launch_bounds__(256) void k_link_scatter(const uint32_t *__restrict__ nsn_rank,
                                                      const uint32_t *__restrict__ nsn_extra,
                                                      const uint8_t *__restrict__ nsn_aux, uint32_t n_nsn,
                                                      uint32_t *__restrict__ cand_extra, uint8_t *__restrict__ cand_aux) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < n_nsn) { cand_extra[nsn_rank[j]] = nsn_extra[j]; cand_aux[nsn_rank[j]] = nsn_aux[j]; }
}
// output length change of one kept record (apply.hip: rec_lengths): IN / DU + len, DE / TL - len, IV 0, TLI + the copied span
__device__ __forceinline__ long long record_delta(uint8_t type, uint32_t pos, uint32_t stop, uint32_t extra) {
    const long long len = (long long)stop - (long long)pos + 1;
    switch (type) {
        case MSIM_IN: case MSIM_DU: return len;
        case MSIM_DE: case MSIM_TL: return -len;
        case MSIM_TLI: return stop + 1 > extra ? (long long)(stop + 1 - extra) : 0ll;
        default: return 0ll;
    }
}

// end (exclusive) of the blocked range a kept non-SNP opens: [pos, stop + block] resp. [pos, pos + block]
// for an insertion (mutator.py:204-209); 0 for everything else
__device__ __forceinline__ uint32_t blocked_end(uint32_t pos, uint8_t type, uint32_t stop, const BlockTable &bt) {
    if (type == MSIM_SN || stop == CHAIN_DROPPED) return 0;
    if (type == MSIM_TLI) return bt.p1[MSIM_TLI];          // its stop is 0 in the boundary pass: range(start, 1 + block) (mutator.py:207)
    const unsigned long long e = (unsigned long long)(type == MSIM_IN ? pos : stop) + bt.p1[type & 7];
    return e > 0xffffffffull ? 0xffffffffu : (uint32_t)e;
}

__global__ __launch_bounds__(CB_THREADS) void k_blk_reduce(const uint32_t *__restrict__ cand_pos,
                                                           const uint8_t *__restrict__ cand_type,
                                                           const uint32_t *__restrict__ cand_stop, uint32_t k,
                                                           BlockTable bt, uint32_t *__restrict__ bmax,
                                                           const MixRangeDev *__restrict__ rt, uint32_t n_draw) {
    __shared__ uint32_t wsum[CB_THREADS / 64];
    const uint32_t i0 = blockIdx.x * CB_BLOCK + threadIdx.x * CB_ITEMS;
    uint32_t m = 0;
    // Several drawing ranges: the blocked range is reset per range (mutator.py:184).  Clipping every blocked end to its
    // range's stop + 1 (<= the next range's start) makes the contig-wide running maximum equal the per-range one.
    uint32_t r = (rt && i0 < k) ? mix_range_of(rt, n_draw, i0) : 0;
#pragma unroll
    for (int q = 0; q < CB_ITEMS; q++) {
        if (i0 + q < k) {
            uint32_t clip = 0xffffffffu;
            if (rt) {
                while (r + 1 < n_draw && rt[r + 1].rec_base <= i0 + q) r++;
                clip = rt[r].clip;
            }
            const uint8_t t = cand_type[i0 + q];
            if (t != MSIM_SN) m = max(m, min(clip, blocked_end(cand_pos[i0 + q], t, cand_stop[i0 + q], bt)));
        }
    }
    uint32_t total;
    (void)block_scan_max(m, wsum, total);
    if (threadIdx.x == 0) bmax[blockIdx.x] = total;
}

// exclusive running maximum of a[0..n) in place; single workgroup
__global__ __launch_bounds__(1024) void k_scan_max_u32(uint32_t *__restrict__ a, uint32_t n) {
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
            buf[threadIdx.x] = max(buf[threadIdx.x], t);
            __syncthreads();
        }
        const uint32_t ex = threadIdx.x ? buf[threadIdx.x - 1] : 0;
        const uint32_t c = carry, last = buf[1023];
        if (i < n) a[i] = max(c, ex);
        __syncthreads();
        if (threadIdx.x == 0) carry = max(c, last);
        __syncthreads();
    }
}

// keep flags (KEEP_BIT in cand_type) + per-workgroup counts of kept mutations, kept SNPs, insert bases.
// rt != nullptr (host-chain engine): blocked ends clipped per range (see k_blk_reduce); a candidate below its range's
// visit_from lies inside a DE / DU / IV span of an EARLIER range and is never visited by __mutate_sequence
// (mutator.py:376,386,398) -- it took part in its own range's boundary pass, but is no record and draws nothing;
// sn_chained: the SNPs went through the host chain as well (cand_stop holds their verdict).
__global__ __launch_bounds__(CB_THREADS) void k_keep_flags(const uint32_t *__restrict__ cand_pos,
                                                           uint8_t *__restrict__ cand_type,
                                                           const uint32_t *__restrict__ cand_stop, uint32_t k,
                                                           BlockTable bt, const uint32_t *__restrict__ bmax,
                                                           uint32_t *__restrict__ cnt_keep, uint32_t *__restrict__ cnt_sn,
                                                           uint32_t *__restrict__ cnt_ins, long long *__restrict__ blk_delta,
                                                           const MixRangeDev *__restrict__ rt, uint32_t n_draw,
                                                           const uint32_t *__restrict__ visit_from, uint32_t sn_chained,
                                                           const uint32_t *__restrict__ cand_extra,
                                                           const uint8_t *__restrict__ cand_aux) {
    __shared__ uint32_t wsum[CB_THREADS / 64];
    const uint32_t i0 = blockIdx.x * CB_BLOCK + threadIdx.x * CB_ITEMS;
    uint32_t pos[CB_ITEMS], stop[CB_ITEMS], before[CB_ITEMS];
    uint8_t t[CB_ITEMS];
    bool vis[CB_ITEMS];
    uint32_t run = 0;
    uint32_t r = (rt && i0 < k) ? mix_range_of(rt, n_draw, i0) : 0;
#pragma unroll
    for (int q = 0; q < CB_ITEMS; q++) {
        const bool in = i0 + q < k;
        pos[q] = in ? cand_pos[i0 + q] : 0;
        t[q] = in ? cand_type[i0 + q] : (uint8_t)0;
        stop[q] = (in && (t[q] != MSIM_SN || sn_chained)) ? cand_stop[i0 + q] : CHAIN_DROPPED;
        before[q] = run;
        uint32_t clip = 0xffffffffu;
        vis[q] = true;
        if (rt && in) {
            while (r + 1 < n_draw && rt[r + 1].rec_base <= i0 + q) r++;
            clip = rt[r].clip;
            vis[q] = pos[q] >= visit_from[r];
        }
        if (in) run = max(run, min(clip, blocked_end(pos[q], t[q], stop[q], bt)));
    }
    uint32_t total;
    const uint32_t pre = max(bmax[blockIdx.x], block_scan_max(run, wsum, total));
    uint32_t nk = 0, ns = 0, ni = 0;
    long long delta = 0;
#pragma unroll
    for (int q = 0; q < CB_ITEMS; q++) {
        if (i0 + q >= k) continue;
        bool keep;
        if (t[q] == MSIM_SN) {                                           // mutator.py:190-196
            keep = (sn_chained ? stop[q] != CHAIN_DROPPED : pos[q] >= max(pre, before[q])) && vis[q];
            ns += keep ? 1u : 0u;
        } else {
            // (translocations: an entry __fix_tl_amount deleted took part in the boundary pass -- its stop blocked above --
            //  but is no record)
            const bool tomb = cand_aux && (cand_aux[i0 + q] & CHAIN_TOMBSTONE);
            keep = stop[q] != CHAIN_DROPPED && vis[q] && !tomb;
            if (keep && t[q] == MSIM_IN) ni += stop[q] - pos[q] + 1;
            if (keep) delta += record_delta(t[q], pos[q], stop[q], cand_extra ? cand_extra[i0 + q] : 0u);   // mutator.py:343-421
        }
        nk += keep ? 1u : 0u;
        if (keep) cand_type[i0 + q] = t[q] | KEEP_BIT;
    }
    // the workgroup's length change: scanned with the counts (k_scan4) -- its prefix gives every record's output
    // offset at emission (k_emit_records), its total the contig's length delta
    __shared__ long long wdelta[CB_THREADS / 64];
    for (int o = 32; o > 0; o >>= 1) delta += __shfl_down(delta, o, 64);
    if ((threadIdx.x & 63) == 0) wdelta[threadIdx.x >> 6] = delta;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < CB_THREADS / 64; w++) t += wdelta[w];
        blk_delta[blockIdx.x] = t;
    }
    uint32_t tk, ts, ti;
    (void)block_scan_add(nk, wsum, tk);
    (void)block_scan_add(ns, wsum, ts);
    (void)block_scan_add(ni, wsum, ti);
    if (threadIdx.x == 0) { cnt_keep[blockIdx.x] = tk; cnt_sn[blockIdx.x] = ts; cnt_ins[blockIdx.x] = ti; }
}

// four exclusive scans in one launch (blockIdx.x selects the array), totals to a[n]; publishes them.  Workgroups 0..2:
// kept mutations / kept SNPs / insert bases; workgroup 3: the 64-bit length deltas.  A wave scans by shuffles, the 16
// wave totals go through LDS: two barriers per 1024 items.
template <class T>
__device__ __forceinline__ T scan_chunks(T *__restrict__ a, uint32_t n, T *wsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T carry = 0;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const T v = i < n ? a[i] : (T)0;
        T incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const T t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        T pre = 0, tot = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) pre += wsum[w];
            tot += wsum[w];
        }
        if (i < n) a[i] = carry + pre + incl - v;
        carry += tot;
        __syncthreads();
    }
    return carry;
}
__global__ __launch_bounds__(1024) void k_scan4(uint32_t *__restrict__ a0, uint32_t *__restrict__ a1, uint32_t *__restrict__ a2,
                                                long long *__restrict__ a3, uint32_t n, PlanState *__restrict__ ps) {
    __shared__ long long wsum64[16];
    if (blockIdx.x == 3) {
        const long long total = scan_chunks<long long>(a3, n, wsum64);
        if (threadIdx.x == 0) { a3[n] = total; ps->len_delta = total; }
        return;
    }
    uint32_t *a = blockIdx.x == 0 ? a0 : blockIdx.x == 1 ? a1 : a2;
    const uint32_t total = scan_chunks<uint32_t>(a, n, reinterpret_cast<uint32_t *>(wsum64));
    if (threadIdx.x == 0) {
        a[n] = total;
        if (blockIdx.x == 0) ps->n_rec = total; else if (blockIdx.x == 1) ps->n_sn = total; else ps->pool_len = total;
    }
}

// kept candidates -> record table (position order), SNP ordinal -> record index, insert pool offsets
__global__ __launch_bounds__(CB_THREADS) void k_emit_records(const uint32_t *__restrict__ cand_pos,
                                                             const uint8_t *__restrict__ cand_type,
                                                             const uint32_t *__restrict__ cand_stop, uint32_t k,
                                                             const uint32_t *__restrict__ off_keep,
                                                             const uint32_t *__restrict__ off_sn,
                                                             const uint32_t *__restrict__ off_ins,
                                                             const long long *__restrict__ off_delta,
                                                             msim_record *__restrict__ recs,
                                                             uint32_t *__restrict__ sn_index, uint32_t *__restrict__ rec_off,
                                                             const uint32_t *__restrict__ cand_extra,
                                                             const uint8_t *__restrict__ cand_aux) {
    __shared__ uint32_t wsum[CB_THREADS / 64];
    __shared__ long long wsum64[CB_THREADS / 64];
    const uint32_t i0 = blockIdx.x * CB_BLOCK + threadIdx.x * CB_ITEMS;
    uint8_t t[CB_ITEMS];
    uint32_t pos[CB_ITEMS], stop[CB_ITEMS], ext[CB_ITEMS];
    uint32_t nk = 0, ns = 0, ni = 0;
    long long nd = 0;
#pragma unroll
    for (int q = 0; q < CB_ITEMS; q++) {
        t[q] = i0 + q < k ? cand_type[i0 + q] : (uint8_t)0;
        pos[q] = 0; stop[q] = 0; ext[q] = 0;
        if (t[q] & KEEP_BIT) {
            pos[q] = cand_pos[i0 + q];
            const uint8_t ty = t[q] & 7;
            stop[q] = ty == MSIM_SN ? pos[q] : cand_stop[i0 + q];
            if (ty == MSIM_TLI && cand_extra) ext[q] = cand_extra[i0 + q];
            nk++;
            if (ty == MSIM_SN) ns++;
            if (ty == MSIM_IN) ni += stop[q] - pos[q] + 1;
            nd += record_delta(ty, pos[q], stop[q], ext[q]);             // mutator.py:343-421
        }
    }
    uint32_t tot;
    long long tot64;
    uint32_t r = off_keep[blockIdx.x] + block_scan_add(nk, wsum, tot);
    uint32_t s = off_sn[blockIdx.x] + block_scan_add(ns, wsum, tot);
    uint32_t p = off_ins[blockIdx.x] + block_scan_add(ni, wsum, tot);
    long long shift = off_delta[blockIdx.x] + block_scan_add64(nd, wsum64, tot64);   // length change of every record before
#pragma unroll
    for (int q = 0; q < CB_ITEMS; q++) {
        if (!(t[q] & KEEP_BIT)) continue;
        const uint8_t ty = t[q] & 7;
        msim_record rec;
        rec.pos = pos[q]; rec.stop = stop[q]; rec.extra = 0; rec.type = ty; rec.aux = 0; rec.rsv = 0;
        if (ty == MSIM_SN) sn_index[s++] = r;
        if (ty == MSIM_IN) { rec.extra = p; p += stop[q] - pos[q] + 1; }
        if (ty == MSIM_TLI) { rec.extra = ext[q]; rec.aux = cand_aux ? (uint8_t)(cand_aux[i0 + q] & 3) : (uint8_t)0; }   // linked span + flags
        // the record's offset in the mutated stream, the table APPLY would otherwise scan for (apply.hip: k_offsets)
        rec_off[r] = (uint32_t)((long long)pos[q] + shift);
        shift += record_delta(ty, pos[q], stop[q], ext[q]);
        recs[r++] = rec;
    }
}

// insert bases: "ATGC"[word & 3], one NumPy-stream word per base, in position order (mutator.py:465-471)
__global__ __launch_bounds__(256) void k_pool_fill(const uint32_t *__restrict__ np_raw, unsigned long long np_base,
                                                   uint32_t pool_len, uint8_t *__restrict__ pool) {
    const uint32_t g = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (g >= pool_len) return;
    uint32_t x = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t w = g + q < pool_len ? mt_temper(np_raw[np_base + g + q]) : 0u;
        x |= (uint32_t)("ATGC"[w & 3u]) << (8 * q);
    }
    *reinterpret_cast<uint32_t *>(pool + g) = x;         // the pool buffer is padded: whole dwords are in bounds
}

__global__ void k_set_pos(PlanState *ps, unsigned long long pos) { ps->pos = pos; ps->snp_base = pos; ps->len_delta = 0; }

// ---- host-sampled contigs (many small ranges): the device still owns the streams
// tempered words from the CURRENT device position on (the host needs no round trip to learn it)
__global__ __launch_bounds__(256) void k_temper_window_ps(const uint32_t *__restrict__ raw, const PlanState *__restrict__ ps,
                                                          uint32_t n, uint32_t *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = mt_temper(raw[ps->pos + i]);
}

// ------------------------------------------------------------------ 7. many small sampled ranges
// An RMT file in the style of the reference's examples leaves thousands of drawing ranges per contig, a few hundred
// SNPs each.  Every random.sample() starts at the word where the previous one stopped and the stops are data
// dependent, so the ranges form a chain; the host finds the cuts (plan_host.cpp: cut_ranges_host) and the device
// does everything per position, in bulk: pos = start + value + d * rank-inside-the-range (util.py:104-109).
// (Round 2 also carried a single-workgroup device walk of that chain, k_sample_walk: bit-exact, but a link of the
// chain is ~12 dependent LDS / barrier phases -- 2.4 us per range against 0.5 us on one host core -- removed in round 3.)
struct WalkRange { uint32_t start, k, n, rec_base, pool; };    // pool = 1: n <= setsize (pool path)

// Host-cut contigs (ctx.h: cut_ranges_host): the host found where each range's sample starts in the word window;
// every word of [0, consumed) looks up its range (cuts ascend strictly: a drawing range consumes at least one word),
// repeats the acceptance test and ORs start + value into the contig-wide bitmap -- duplicates collapse by themselves.
// Words of pool-path ranges are skipped (their positions come from the host as a list, k_list_to_bits).
__global__ __launch_bounds__(256) void k_interval_bits(const uint32_t *__restrict__ raw, const unsigned long long *__restrict__ p0_slot,
                                                       uint32_t consumed, const uint32_t *__restrict__ cut,
                                                       const WalkRange *__restrict__ ranges, uint32_t n_draw,
                                                       uint32_t *__restrict__ bits) {
    __shared__ uint32_t span[2];
    const uint32_t first = blockIdx.x * 256, i = first + threadIdx.x;
    if (threadIdx.x < 2) {                                 // ranges of the workgroup's first and last word
        const uint32_t key = threadIdx.x ? min(first + 255u, consumed - 1) : first;
        uint32_t lo = 0, hi = n_draw;                      // last r with cut[r] <= key
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cut[mid] <= key) lo = mid; else hi = mid;
        }
        span[threadIdx.x] = lo;
    }
    __syncthreads();
    if (i >= consumed) return;
    uint32_t lo = span[0], hi = span[1] + 1;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cut[mid] <= i) lo = mid; else hi = mid;
    }
    const WalkRange R = ranges[lo];
    if (R.pool) return;
    const uint32_t v = mt_temper(raw[*p0_slot + i]) >> __clz(R.n);                 // 32 - bit_length(n)
    if (v < R.n) {
        const uint32_t b = R.start + v;
        atomicOr(&bits[b >> 5], 1u << (b & 31));
    }
}

// the position the window started at stays available to kernels that run beside the chain
__global__ void k_advance_pos_save(PlanState *ps, unsigned long long words, unsigned long long *p0_slot) {
    *p0_slot = ps->pos;
    ps->pos += words;
    ps->snp_base = ps->pos;
}

// unordered list of sampled positions -> bits of the contig-wide bitmap (bulk, off the chain)
__global__ __launch_bounds__(256) void k_list_to_bits(const uint32_t *__restrict__ list, uint32_t n, uint32_t *__restrict__ bits) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = list[i];
    atomicOr(&bits[b >> 5], 1u << (b & 31));
}

// Contig-wide bitmap of (range start + value) -> records, in bulk: global rank by popcount prefix (k_bitmap_count +
// k_scan_u32 as for the SNP sampler), range of a bit by binary search in the range table, rank inside the range =
// global rank - rec_base of the range.  Leaves the bitmap zeroed for the next contig that uses this scratch set.
__global__ __launch_bounds__(BM_THREADS) void k_walk_expand(uint64_t *__restrict__ bm, uint32_t n_words,
                                                            const uint32_t *__restrict__ block_off,
                                                            const WalkRange *__restrict__ ranges, uint32_t n_ranges,
                                                            uint32_t d, msim_record *__restrict__ recs) {
    __shared__ uint32_t part[BM_THREADS];
    const uint32_t i = blockIdx.x * BM_THREADS + threadIdx.x;
    uint64_t w = i < n_words ? bm[i] : 0;
    if (w) bm[i] = 0;
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
    uint32_t r = 0;
    bool have = false;
    while (w) {
        const uint32_t bit = i * 64 + (uint32_t)__builtin_ctzll(w);
        w &= w - 1;
        if (!have || (r + 1 < n_ranges && ranges[r + 1].start <= bit)) {            // last range with start <= bit
            uint32_t lo = 0, hi = n_ranges;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ranges[mid].start <= bit) lo = mid; else hi = mid;
            }
            r = lo;
            have = true;
        }
        const uint32_t pos = bit + d * (rank - ranges[r].rec_base);
        msim_record rec;
        rec.pos = pos; rec.stop = pos; rec.extra = 0; rec.type = MSIM_SN; rec.aux = 0; rec.rsv = 0;
        recs[rank] = rec;
        rank++;
    }
}

// (the counter-based engine of MSIM_RNG_FAST lives in fast_math.h / fast_kernels.h / plan_fast.hip)

// single lane: the bookkeeping block goes to the pinned host mailbox (plain stores over PCIe)
__global__ void k_publish(const PlanState *__restrict__ ps, PlanState *__restrict__ mailbox) {
    *mailbox = *ps;
    __threadfence_system();
}

// the same for a host that polls instead of synchronising the stream: the sequence word is written last
__global__ void k_publish_seq(const PlanState *__restrict__ ps, PlanState *__restrict__ mailbox,
                              uint32_t *__restrict__ seq_word, uint32_t seq) {
    *mailbox = *ps;
    __threadfence_system();
    __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_raise_flag(PlanState *ps, uint32_t flag) { ps->flags |= flag; }

__global__ void k_state_init(PlanState *ps, unsigned long long pos) {
    ps->pos = pos; ps->snp_base = pos; ps->flags = 0; ps->dups = 0; ps->accepted_used = 0;
    ps->n_nsn = ps->n_rec = ps->n_sn = ps->pool_len = 0;
    ps->len_delta = 0;
    ps->ahead_margin_used = 0; ps->rsv = 0;
}

}  // namespace

}  // namespace msim
