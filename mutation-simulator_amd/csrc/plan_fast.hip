// PLAN with the counter-based generator (MSIM_RNG_FAST, "--rng fast"): host orchestration.  gfx950 (MI355X) only.
//
// fast_math.h: the construction (the reference's, mutator.py:144-265 / util.py:93-109) and its arithmetic;
// fast_kernels.h: the kernels.  Nothing here is stream-compatible with the reference and nothing chains: a contig is
// 2 launches (SNP-only settings) or 7-10 (types beyond SNPs) on one of four streams, its APPLY follows on the emit stream
// behind an event, and the host never waits for a count -- the kernels read record counts and mutated lengths from device
// memory (Contig::d_dyn), the host collects them at the next synchronising call.
#include <algorithm>
#include <chrono>
#include <climits>
#include <cstddef>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "ctx.h"
#include "fast_kernels.h"

namespace msim {

namespace {

constexpr int F_SETS = 3;                                  // batches in flight (one stream + one scratch set each)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;                                        // elements
};
template <class T>
struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
};

struct FastSet {
    DevBuf<uint8_t> tab;                                   // FRange | Settings | SubDesc | big-range list: ONE copy per contig
    PinBuf<uint8_t> h_tab;                                 // its pinned source
    DevBuf<LeafDesc> leaves;
    DevBuf<uint32_t> sub_k, sub_c0;                        // per subtree of 64 leaves: its points, its first candidate
    DevBuf<uint32_t> cand_pos, cand_stop, cand_bend, cand_end2, blk_out, off_dummy;
    DevBuf<uint32_t> kept;                                 // per contig of the batch: [0] kept any, [1] / [2] pass 1 / 2 handed over
    DevBuf<uint8_t> cand_meta;
    DevBuf<uint32_t> blk_u32;                              // S | indep | in | nrec | pool  (5 arrays of nb + 1)
    DevBuf<long long> blk_delta;
    hipEvent_t done = nullptr;
    bool pending = false;
};

struct Prep {
    std::vector<FRange> fr;
    std::vector<Settings> sets;
    std::vector<SubDesc> subs;
    std::vector<uint32_t> big;
    uint64_t K = 0;
    uint64_t n_leaves = 0;
    uint32_t lgB_max = LG_LEAF_MIN;
    bool snp_only = true;                                  // every candidate is an SNP that blocks nothing: records straight from the leaves
    bool all_sn = true;                                    // every candidate is an SNP (no length change)
    bool need_visit = false;                               // several ranges and a consuming type: the visit filter can drop records
    uint32_t maxspan = 1, maxspan_visit = 1;
    uint64_t pool_cap = 0, out_cap = 0, n_struct_est = 0;
    Block1 block1{};
    int64_t d = 1;
};

struct Replay { uint64_t key = 0; uint32_t seq = 0; Prep P; };
struct Pending { int contig = -1; bool apply = false; uint64_t key = 0; uint32_t seq = 0; Prep P; };   // planned, not enqueued yet

}  // namespace

struct HostProf {                                          // MSIM_FAST_PROF=1: where the host's time per contig goes (stderr at collection)
    bool on = getenv("MSIM_FAST_PROF") != nullptr;
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t n = 0;
    std::chrono::steady_clock::time_point last;
    void start() { if (on) last = std::chrono::steady_clock::now(); }
    void lap(int i) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        t[i] += std::chrono::duration<double, std::micro>(now - last).count();
        last = now;
    }
    void report() {
        if (!on || !n) return;
        fprintf(stderr, "fast PLAN host us per contig (%llu contigs): prepare %.1f, wait-set %.1f, tables+grow %.1f, copy %.1f, split %.1f, leaf %.1f, "
                        "keep+emit %.1f, events %.1f\n", (unsigned long long)n, t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[4] / n, t[5] / n, t[6] / n, t[7] / n);
        for (double &x : t) x = 0;
        n = 0;
    }
};

struct FastPlan {
    HostProf prof;                                         // (per context: two contexts may plan from two threads)
    hipStream_t lane[F_SETS] = {};
    FastSet set[F_SETS];
    uint32_t *d_flags = nullptr;                           // sticky FF_* flags of everything since the last collection
    DynSizes *d_dyn = nullptr;                             // one per contig of the context
    DynSizes *h_dyn = nullptr;                             // pinned copy (collection)
    hipEvent_t t0 = nullptr, t1[F_SETS] = {};
    bool pending = false;                                  // work enqueued since the last collection
    std::vector<int> sized;                                // contigs whose sizes are still on the device only
    std::map<int, Replay> replay;                          // per contig planned with device-side counts: its tables (see enqueue_batch)
    uint64_t replays = 0;                                  // plans that had to be replayed with the orbit kernels
    std::vector<Pending> queue;                            // msim_plan_contig calls whose device work has not been enqueued
    // A run of plans between two synchronising calls is a cycle (a genome in bench.py, a contig in the CLI).  Its first half
    // goes to the device as soon as it is queued -- judged by the previous cycle's size -- so that the device works while the
    // host still prepares the tables of the second half (35 k ranges: 0.8 ms of host work in front of a 2.5 ms step otherwise).
    // Where the tables themselves are the host's work (an RMT file: thousands of ranges per contig, 20-40 us of `prepare` each),
    // a first, smaller part goes out before that: once 100 us of preparation are queued (`queued_host_us`) -- the device then
    // starts on the three largest contigs while the host is a fifth into the genome, not half.
    uint64_t queued_K = 0, cycle_K = 0, last_cycle_K = 0;
    double queued_host_us = 0;
    bool early_done = false, half_done = false;
    // The batches of a cycle take the lanes in turn, from lane 0.  (Measured and not kept, round 4: the later batches on
    // high-priority lanes, and/or their kernels gated behind the PLAN kernels of the batch before -- so that the first batch plans
    // alone and the second one's short kernels do not starve behind the first one's rewrite.  c3 2.58 ms vs 2.44 without: the
    // PLAN kernels are latency-bound and two batches planning side by side fill each other's bubbles.)
    uint32_t cycle_batches = 0;
    uint32_t *h_flags = nullptr;                           // pinned: [lane] flags word, then [lane][MAX_CONTIGS] DynSizes (collection)
    DynSizes *h_dyn_lane[F_SETS] = {};
    std::vector<uint8_t> lane_of;                          // per contig: the lane that planned it last
};

namespace {

template <class T>
int dev_grow(Ctx *c, FastPlan *f, DevBuf<T> &b, size_t want) {
    if (b.cap >= want) return MSIM_OK;
    // a buffer is replaced: nothing that may still use the old one can be in flight
    for (auto st : f->lane) if (st) MSIM_HIP(c, wait_stream(st));
    if (b.p) MSIM_HIP(c, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    const size_t n = want + want / 4 + 1024;
    MSIM_HIP(c, hipMalloc(&b.p, n * sizeof(T)));
    b.cap = n;
    return MSIM_OK;
}

bool type_drawable(const msim_range &r, int j) {
    const uint64_t lo = j ? r.cdf_thr[j - 1] : 0;
    return r.cdf_thr[j] > lo && lo < (1ull << 53);
}

// msim_range tables -> what the kernels read.  Returns MSIM_OK, MSIM_ERR_VALUE (the reference's ValueError) or
// MSIM_ERR_UNSUPPORTED with the reason in c->err.
int prepare(Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges, Prep &P) {
    const msim_params &mp = c->params;
    int64_t d = mp.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, mp.block[t]);
    if (d < 0 || d >= (1ll << 31)) return fail(c, MSIM_ERR_UNSUPPORTED, "fast RNG mode: block sizes beyond 2^31");
    P.d = d;
    for (int t = 0; t < 8; t++) {
        const int64_t b = t ? mp.block[t] : 0;
        P.block1.v[t] = b < 0 ? 1u : (b >= 0xfffffffell ? 0xffffffffu : (uint32_t)(b + 1));
    }
    if (L >= (1ull << 32)) return fail(c, MSIM_ERR_UNSUPPORTED, "contig of 4 GiB or more");
    const bool sn_blocks = mp.block[MSIM_SN] != d;         // an SNP then blocks successors (mutator.py:204-206)
    int64_t prev_stop = -1;
    double pool_mean = 0, pool_var = 0, grow_mean = 0, grow_var = 0, max_piece = 0, struct_est = 0;
    bool consuming = false;
    uint32_t n_draw = 0;
    // A table of tens of thousands of ranges repeats a few MutationSettings (rmt.py:79-163: the std line between two gene blocks,
    // the hot and the cold spots): what a range's settings come to -- its Settings entry, the checks, the terms of the size
    // estimates -- is worked out when they differ from the drawing range before, and replayed (same additions, same order) when
    // they do not.  45 -> 14 ns per range: the tables of c4sv's 35 k ranges were 0.8 ms of host work in front of a 2.7 ms step.
    struct Term { double p; double hi; uint8_t t; };
    struct Derived {
        const msim_range *of = nullptr;
        uint32_t set = 0;
        bool sn_only = true;
        int n_terms = 0;
        Term term[8];
    } D;
    constexpr size_t SET_OFF = offsetof(msim_range, n_types), SET_BYTES = sizeof(msim_range) - SET_OFF;
    P.fr.reserve((size_t)n_ranges);
    P.subs.reserve((size_t)n_ranges + 64);
    for (int i = 0; i < n_ranges; i++) {
        const msim_range &r = ranges[i];
        if (r.k == 0) continue;
        const int64_t n = (r.stop - (r.k - 1) * d) - r.start;
        // random.sample(range(start, stop - (k - 1) d), k): "Sample larger than population or is negative" (util.py:104)
        if (r.k < 0 || n < r.k) return fail(c, MSIM_ERR_VALUE, "Sample larger than population or is negative");
        if (r.start < 0 || r.stop >= (1ll << 32) || n >= (1ll << 32))
            return fail(c, MSIM_ERR_UNSUPPORTED, "fast RNG mode: range beyond 2^32 positions");
        if (r.start <= prev_stop)
            return fail(c, MSIM_ERR_UNSUPPORTED, "fast RNG mode: overlapping or unsorted ranges (the reference's dict semantics) are not covered");
        if ((uint64_t)r.stop >= L) return fail(c, MSIM_ERR_UNSUPPORTED, "fast RNG mode: range beyond the contig end");
        prev_stop = r.stop;
        if (!D.of || memcmp(reinterpret_cast<const char *>(D.of) + SET_OFF, reinterpret_cast<const char *>(&r) + SET_OFF, SET_BYTES)) {
            if (r.n_types < 1 || r.n_types > 8) return fail(c, MSIM_ERR_ARG, "range with no or more than 8 mutation types");
            Settings s;
            memset(&s, 0, sizeof s);
            s.n_types = (uint32_t)r.n_types;
            D.sn_only = true;
            D.n_terms = 0;
            for (int j = 0; j < r.n_types; j++) {
                s.thr[j] = r.cdf_thr[j];
                s.type[j] = (uint8_t)r.types[j];
                if (!type_drawable(r, j)) continue;
                const int t = r.types[j];
                const double p = (double)(std::min<uint64_t>(r.cdf_thr[j], 1ull << 53) - (j ? r.cdf_thr[j - 1] : 0)) / 9007199254740992.0;
                if (t == MSIM_SN) continue;
                D.sn_only = false;
                if (t == MSIM_TL || t == MSIM_TLI)
                    return fail(c, MSIM_ERR_UNSUPPORTED, "fast RNG mode: translocations (-tl) are not covered; use --rng compat");
                if (t != MSIM_IN && t != MSIM_DE && t != MSIM_DU && t != MSIM_IV) return fail(c, MSIM_ERR_ARG, "unknown mutation type");
                const int64_t lo = r.min_len[t], hi = r.max_len[t];
                if (lo < 1 || hi < lo || hi >= (1ll << 31)) return fail(c, MSIM_ERR_UNSUPPORTED, "fast RNG mode: mutation lengths outside 1 .. 2^31");
                D.term[D.n_terms++] = Term{p, (double)hi, (uint8_t)t};
                if (t == MSIM_IN || t == MSIM_DU) max_piece = std::max(max_piece, (double)hi);
                if (t == MSIM_DE || t == MSIM_DU || t == MSIM_IV) {
                    consuming = true;
                    P.maxspan_visit = std::max<uint64_t>(P.maxspan_visit, (uint64_t)hi);
                    P.maxspan = (uint32_t)std::min<uint64_t>(0xffffffffull, std::max<uint64_t>(P.maxspan, (uint64_t)hi - 1 + P.block1.v[t]));
                } else {
                    P.maxspan = std::max(P.maxspan, P.block1.v[t]);
                }
            }
            P.maxspan = std::max(P.maxspan, P.block1.v[MSIM_SN]);
            for (int t = 0; t < 8; t++) {
                const int64_t lo = std::max<int64_t>(r.min_len[t], 0), hi = std::max<int64_t>(r.max_len[t], lo);
                s.min_len[t] = (uint32_t)std::min<int64_t>(lo, 0x7fffffff);
                s.max_len[t] = (uint32_t)std::min<int64_t>(hi, 0x7fffffff);
                s.width[t] = s.max_len[t] - s.min_len[t] + 1;
            }
            if (!D.sn_only) { P.all_sn = false; P.snp_only = false; }
            if (sn_blocks) P.snp_only = false;
            uint32_t set = 0;
            for (; set < P.sets.size(); set++) if (!memcmp(&P.sets[set], &s, sizeof s)) break;
            if (set == P.sets.size()) P.sets.push_back(s);
            D.set = set;
            D.of = &r;
        }
        const double kd = (double)r.k;
        for (int q = 0; q < D.n_terms; q++) {
            const Term &tm = D.term[q];
            struct_est += kd * tm.p;
            if (tm.t == MSIM_IN) { pool_mean += kd * tm.p * tm.hi; pool_var += kd * tm.p * tm.hi * tm.hi; }
            if (tm.t == MSIM_IN || tm.t == MSIM_DU) { grow_mean += kd * tm.p * tm.hi; grow_var += kd * tm.p * tm.hi * tm.hi; }
        }
        FRange f;
        f.start = (uint32_t)r.start;
        f.n = (uint32_t)n;
        f.k = (uint32_t)r.k;
        f.cand_base = (uint32_t)P.K;
        f.leaf_base = (uint32_t)P.n_leaves;
        f.lgB = leaf_lg((uint64_t)n, (uint64_t)r.k);
        f.clip = (uint32_t)std::min<int64_t>(r.stop + 1, 0xffffffffll);
        f.set = D.set;
        f.sub_base = (uint32_t)P.subs.size();
        f.slot = 0;
        const uint64_t T = ((uint64_t)n + (1ull << f.lgB) - 1) >> f.lgB;
        if (T > (1u << LG_SUB)) P.big.push_back((uint32_t)P.fr.size());
        for (uint64_t sidx = 0; sidx < (T + 63) >> LG_SUB; sidx++) P.subs.push_back(SubDesc{(uint32_t)P.fr.size(), (uint32_t)sidx});
        P.fr.push_back(f);
        P.lgB_max = std::max(P.lgB_max, f.lgB);
        P.K += (uint64_t)r.k;
        P.n_leaves += T;
        n_draw++;
        if (P.K >= (1ull << 31) || P.n_leaves >= (1ull << 31)) return fail(c, MSIM_ERR_UNSUPPORTED, "fast RNG mode: 2^31 or more candidates on one contig");
    }
    if (n_draw >= (1u << 24)) return fail(c, MSIM_ERR_UNSUPPORTED, "fast RNG mode: 2^24 or more drawing ranges on one contig");
    P.need_visit = n_draw > 1 && consuming;
    P.n_struct_est = (uint64_t)struct_est;
    const double pool_cap = pool_mean + 16.0 * std::sqrt(pool_var) + max_piece + 64.0;
    const double out_cap = (double)L + grow_mean + 16.0 * std::sqrt(grow_var) + max_piece + 64.0;
    if (!P.all_sn && (pool_cap >= 4.0e9 || out_cap >= 4294967295.0))
        return fail(c, MSIM_ERR_UNSUPPORTED, "mutated contig (or its insert pool) of 4 GiB or more");
    P.pool_cap = P.all_sn ? 0 : (uint64_t)pool_cap;
    P.out_cap = P.all_sn ? L : (uint64_t)out_cap;
    return MSIM_OK;
}

int ensure_plan(Ctx *c) {
    if (c->fast) return MSIM_OK;
    FastPlan *f = new FastPlan();
    c->fast = f;
    for (auto &st : f->lane) MSIM_HIP(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto &s : f->set) {
        MSIM_HIP(c, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    MSIM_HIP(c, hipMalloc(&f->d_flags, 64));
    MSIM_HIP(c, hipMemset(f->d_flags, 0, 64));
    MSIM_HIP(c, hipMalloc(&f->d_dyn, sizeof(DynSizes) * MAX_CONTIGS));
    MSIM_HIP(c, hipHostMalloc(&f->h_dyn, sizeof(DynSizes) * MAX_CONTIGS * F_SETS + 256, hipHostMallocDefault));
    for (int i = 0; i < F_SETS; i++) f->h_dyn_lane[i] = f->h_dyn + (size_t)i * MAX_CONTIGS;
    f->h_flags = reinterpret_cast<uint32_t *>(f->h_dyn + (size_t)F_SETS * MAX_CONTIGS);
    MSIM_HIP(c, hipEventCreate(&f->t0));
    for (auto &e : f->t1) MSIM_HIP(c, hipEventCreate(&e));
    return MSIM_OK;
}

}  // namespace

void fast_plan_destroy(Ctx *c) {
    FastPlan *f = c->fast;
    if (!f) return;
    for (auto st : f->lane) if (st) { (void)wait_stream(st); (void)hipStreamDestroy(st); }
    for (auto &s : f->set) {
        void *bufs[] = {s.tab.p, s.leaves.p, s.sub_k.p, s.sub_c0.p, s.cand_pos.p, s.cand_stop.p, s.cand_bend.p, s.cand_end2.p, s.blk_out.p, s.off_dummy.p, s.kept.p,
                        s.cand_meta.p, s.blk_u32.p, s.blk_delta.p};
        for (void *p : bufs) if (p) (void)hipFree(p);
        if (s.h_tab.p) (void)hipHostFree(s.h_tab.p);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (f->d_flags) (void)hipFree(f->d_flags);
    if (f->d_dyn) (void)hipFree(f->d_dyn);
    if (f->h_dyn) (void)hipHostFree(f->h_dyn);
    if (f->t0) (void)hipEventDestroy(f->t0);
    for (auto e : f->t1) if (e) (void)hipEventDestroy(e);
    delete f;
    c->fast = nullptr;
}

int fast_plan_check(Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges) {
    Prep P;
    return prepare(c, L, ranges, n_ranges, P);
}


static int enqueue_batch(Ctx *c, std::vector<Pending> &items, bool orbit_only);

bool fast_lane_joined_at_collect(const Ctx *c, hipStream_t s) {
    const FastPlan *f = c->fast;
    if (!f || !f->pending) return false;
    for (int i = 0; i < F_SETS; i++) if (f->lane[i] == s) return f->set[i].pending;
    return false;
}

int fast_plan_collect(Ctx *c, int (*behind)(Ctx *), int *behind_state) {
    FastPlan *f = c->fast;
    if (behind_state) *behind_state = 0;
    if (f && !f->queue.empty()) {
        const int rc = fast_plan_flush(c);
        if (rc) return rc;
    }
    if (!f || !f->pending) return MSIM_OK;
    if (f->cycle_K) f->last_cycle_K = f->cycle_K;          // a cycle ends here (see FastPlan::queued_K)
    f->cycle_K = 0;
    f->queued_host_us = 0;
    f->early_done = f->half_done = false;
    c->fast->prof.report();
    for (int round = 0; f->pending && round < 3; round++) {
        f->pending = false;
        f->cycle_batches = 0;
        // Flags and sizes come back by asynchronous copies into pinned memory behind each lane's work (a blocking hipMemcpy
        // costs ~30 us at every step boundary): a contig's sizes are read from the copy of the lane that planned it.
        std::vector<int> sized, redo;
        sized.swap(f->sized);
        int lo_idx[F_SETS], hi_idx[F_SETS];
        for (int i = 0; i < F_SETS; i++) { lo_idx[i] = INT_MAX; hi_idx[i] = -1; }
        for (int idx : sized) {
            if ((size_t)idx >= f->lane_of.size()) continue;
            const int li = f->lane_of[(size_t)idx];
            lo_idx[li] = std::min(lo_idx[li], idx); hi_idx[li] = std::max(hi_idx[li], idx);
        }
        for (int i = 0; i < F_SETS; i++) {
            if (!f->set[i].pending) { f->h_flags[i] = 0; continue; }
            if (hi_idx[i] >= 0)
                MSIM_HIP(c, hipMemcpyAsync(f->h_dyn_lane[i] + lo_idx[i], f->d_dyn + lo_idx[i], sizeof(DynSizes) * (size_t)(hi_idx[i] - lo_idx[i] + 1),
                                           hipMemcpyDeviceToHost, f->lane[i]));
            MSIM_HIP(c, hipMemcpyAsync(f->h_flags + i, f->d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, f->lane[i]));
            MSIM_HIP(c, hipEventRecord(f->t1[i], f->lane[i]));
        }
        // one host wait: the emit stream joins the lanes (their t1 events), takes the caller's copies, and is waited for --
        // a hipStreamSynchronize per lane and another round trip for the caller's copies were ~100 us of every step boundary
        for (int i = 0; i < F_SETS; i++)
            if (f->set[i].pending) MSIM_HIP(c, hipStreamWaitEvent(c->emit_stream, f->t1[i], 0));
        if (behind && round == 0) {
            const int brc = behind(c);
            if (brc) return brc;
            if (behind_state) *behind_state = 1;
        }
        MSIM_HIP(c, wait_stream(c->emit_stream));
        for (int i = 0; i < F_SETS; i++)                   // (done by now: the emit stream waited for them)
            if (f->set[i].pending) MSIM_HIP(c, wait_event(f->t1[i]));
        float ms = 0;
        for (int i = 0; i < F_SETS; i++) {                 // (first batch's start to the end of the lane that finished last)
            if (!f->set[i].pending) continue;
            float m = 0;
            MSIM_HIP(c, hipEventElapsedTime(&m, f->t0, f->t1[i]));
            ms = std::max(ms, m);
        }
        c->t.plan_gpu_ms += ms;
        uint32_t flags = 0;
        for (int i = 0; i < F_SETS; i++) flags |= f->h_flags[i];
        for (auto &s : f->set) s.pending = false;
        {
            for (int idx : sized) {
                if ((size_t)idx >= c->contigs.size() || (size_t)idx >= f->lane_of.size()) continue;
                Contig &g = c->contigs[(size_t)idx];
                if (!g.sizes_pending) continue;
                const DynSizes s = f->h_dyn_lane[f->lane_of[(size_t)idx]][idx];
                if (s.flags & FF_NEED_ORBIT) { redo.push_back(idx); continue; }
                flags |= s.flags & 0xffu;
                g.sizes_pending = false;
                g.d_dyn = nullptr;                         // host-known from here on (an APPLY in flight keeps its copy of the pointer)
                g.n_rec = s.n_rec;
                g.pool_len = s.pool_len;
                g.plan_empty = !(s.flags & FF_KEPT_ANY);
                if (!g.all_snp) {
                    g.known_delta = (long long)s.out_len - (long long)g.len;
                    g.delta_known = true;
                    if (g.applied) g.out_len = s.out_len;
                }
                f->replay.erase(idx);
            }
        }
        if (flags) {
            MSIM_HIP(c, hipMemset(f->d_flags, 0, 64));
            if (flags & (FF_POOL_OVERFLOW | FF_OUT_OVERFLOW))
                return fail(c, MSIM_ERR_HIP, "fast RNG sampler: the mutated length or the insert pool overflowed its 16-sigma allocation (results discarded)");
            return fail(c, MSIM_ERR_HIP, "fast RNG sampler: internal error (flags " + std::to_string(flags) + "; results discarded)");
        }
        // Contigs whose block-local boundary pass handed over (blocked ranges reaching over more than 64 candidates: long
        // deletions at a high rate): planned again with the orbit kernels, and applied again where they already were -- the
        // first plan left an empty record table, so the APPLY that ran was a plain copy.
        for (int idx : redo) {
            auto it = f->replay.find(idx);
            if (it == f->replay.end()) return fail(c, MSIM_ERR_HIP, "fast RNG sampler: internal error (no tables to replay a plan from)");
            Replay rp = std::move(it->second);
            f->replay.erase(it);
            Contig &g = c->contigs[(size_t)idx];
            const bool was_applied = g.applied;
            std::vector<Pending> one(1);
            one[0].contig = idx; one[0].key = rp.key; one[0].seq = rp.seq; one[0].P = std::move(rp.P);
            int rc = enqueue_batch(c, one, true);
            if (rc) return rc;
            f->replays++;
            if (behind_state && *behind_state) *behind_state = 2;
            if (was_applied) {
                g.apply_pending = false;                   // (its first APPLY has completed: everything was synchronised above)
                g.dyn_applied = false;
                if ((rc = apply_contig_device(c, g))) return rc;
            }
        }
    }
    return MSIM_OK;
}

// Enqueue PLAN of a batch of contigs: one launch per stage over all of them (fast_kernels.h: FSlot).
// orbit_only (batches of one): the boundary pass / visit filter by the orbit kernels alone -- the replay of a plan whose
// block-local pass handed over, contigs of more than 4096 blocks, and the MSIM_FAST_FORCE_ORBIT test hook.
static int enqueue_batch(Ctx *c, std::vector<Pending> &items, bool orbit_only) {
    int rc;
    if (items.empty()) return MSIM_OK;
    if ((rc = ensure_plan(c))) return rc;
    FastPlan *f = c->fast;
    const uint32_t li = f->cycle_batches++ % F_SETS;
    FastSet &S = f->set[li];
    hipStream_t st = f->lane[li];
    c->fast->prof.start();
    if (S.pending) {                                       // its last user (a few batches ago) may still be in flight on this stream
        MSIM_HIP(c, wait_stream(st));
        S.pending = false;
    }
    c->fast->prof.lap(1);
    if (!f->pending) MSIM_HIP(c, hipEventRecord(f->t0, st));
    f->pending = true;
    const uint64_t key64 = items[0].key;
    const Key2 key2{(uint32_t)key64, (uint32_t)(key64 >> 32)};
    // ---- the batch's tables: ranges, settings, subtrees, big ranges and slots, concatenated
    const uint32_t n_slots = (uint32_t)items.size();
    size_t n_draw = 0, n_sets = 0, n_subs = 0, n_big = 0;
    uint64_t n_leaves = 0, cand_total = 0, nb_total = 0;
    bool snp_only = true, any_all_sn = false, need_visit = false;
    uint32_t lgB_max = LG_LEAF_MIN;
    for (auto &it : items) {
        const Prep &P = it.P;
        n_draw += P.fr.size(); n_sets += P.sets.size(); n_subs += P.subs.size(); n_big += P.big.size();
        n_leaves += P.n_leaves;
        const uint64_t nb = (P.K + OB_BLOCK - 1) / OB_BLOCK;
        cand_total += nb * OB_BLOCK;
        nb_total += nb;
        snp_only = snp_only && P.snp_only;
        any_all_sn = any_all_sn || P.all_sn;
        need_visit = need_visit || P.need_visit;
        lgB_max = std::max(lgB_max, P.lgB_max);
    }
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t off_sets = up16(n_draw * sizeof(FRange)), off_subs = up16(off_sets + n_sets * sizeof(Settings)),
                 off_big = up16(off_subs + n_subs * sizeof(SubDesc)), off_slots = up16(off_big + n_big * sizeof(uint32_t)),
                 off_bslot = up16(off_slots + (size_t)n_slots * sizeof(FSlot));
    const size_t tab_bytes = up16(off_bslot + (snp_only ? 0 : (size_t)nb_total * sizeof(uint32_t)));
    if (S.h_tab.cap < tab_bytes) {
        if (S.h_tab.p) MSIM_HIP(c, hipHostFree(S.h_tab.p));
        S.h_tab.p = nullptr; S.h_tab.cap = 0;
        MSIM_HIP(c, hipHostMalloc(&S.h_tab.p, tab_bytes * 2 + 4096, hipHostMallocDefault));
        S.h_tab.cap = tab_bytes * 2 + 4096;
    }
    if ((rc = dev_grow(c, f, S.tab, tab_bytes))) return rc;
    if ((rc = dev_grow(c, f, S.leaves, n_leaves))) return rc;
    if ((rc = dev_grow(c, f, S.sub_k, n_subs + 1))) return rc;
    if ((rc = dev_grow(c, f, S.sub_c0, n_subs + 1))) return rc;
    if ((rc = dev_grow(c, f, S.kept, (size_t)4 * n_slots))) return rc;
    if (!snp_only) {
        if ((rc = dev_grow(c, f, S.cand_pos, cand_total))) return rc;
        if ((rc = dev_grow(c, f, S.cand_stop, cand_total))) return rc;
        if ((rc = dev_grow(c, f, S.cand_bend, cand_total))) return rc;
        if ((need_visit || orbit_only) && (rc = dev_grow(c, f, S.cand_end2, cand_total))) return rc;
        if (orbit_only && (rc = dev_grow(c, f, S.blk_out, cand_total))) return rc;
        if (any_all_sn && (rc = dev_grow(c, f, S.off_dummy, cand_total))) return rc;  // (SNP-only tables: offsets nobody reads)
        if ((rc = dev_grow(c, f, S.cand_meta, cand_total + 16))) return rc;
        if ((rc = dev_grow(c, f, S.blk_u32, (size_t)7 * (nb_total + 1)))) return rc;
        if ((rc = dev_grow(c, f, S.blk_delta, nb_total + 1))) return rc;
    }
    FRange *h_fr = reinterpret_cast<FRange *>(S.h_tab.p);
    Settings *h_sets = reinterpret_cast<Settings *>(S.h_tab.p + off_sets);
    SubDesc *h_subs = reinterpret_cast<SubDesc *>(S.h_tab.p + off_subs);
    uint32_t *h_big = reinterpret_cast<uint32_t *>(S.h_tab.p + off_big);
    FSlot *h_slots = reinterpret_cast<FSlot *>(S.h_tab.p + off_slots);
    uint32_t *h_bslot = reinterpret_cast<uint32_t *>(S.h_tab.p + off_bslot);
    {
        uint32_t r0 = 0, s0 = 0, sub0 = 0, big0 = 0, leaf0 = 0, cand0 = 0, blk0 = 0;
        for (uint32_t si = 0; si < n_slots; si++) {
            Pending &it = items[si];
            const Prep &P = it.P;
            Contig &ct = c->contigs[(size_t)it.contig];
            {   // the record table (and what APPLY reads beside it) may still be in use by an earlier APPLY of this contig
                const size_t want = (size_t)P.K * sizeof(msim_record);
                const size_t want_pool = (size_t)P.pool_cap + 2 * PAD, want_off = P.all_sn ? 0 : (size_t)P.K * sizeof(uint32_t);
                if (ct.cap_recs < want || ct.cap_pool < want_pool || ct.cap_off < want_off) {
                    MSIM_HIP(c, wait_stream(c->stream));
                    MSIM_HIP(c, wait_stream(c->emit_stream));
                    if (ct.apply_stream) MSIM_HIP(c, wait_stream(ct.apply_stream));
                } else if (ct.apply_pending && ct.ea2) {
                    MSIM_HIP(c, hipStreamWaitEvent(st, ct.ea2, 0));
                }
                if ((rc = dev_reserve(c, (void **)&ct.d_recs, &ct.cap_recs, want))) return rc;
                if ((rc = dev_reserve(c, (void **)&ct.d_pool, &ct.cap_pool, want_pool))) return rc;
                if (want_off && (rc = dev_reserve(c, (void **)&ct.d_off, &ct.cap_off, want_off))) return rc;
            }
            const uint32_t nb = (uint32_t)((P.K + OB_BLOCK - 1) / OB_BLOCK);
            for (size_t q = 0; q < P.fr.size(); q++) {
                FRange fr = P.fr[q];
                fr.slot = si;
                fr.set += s0;
                h_fr[r0 + q] = fr;
            }
            memcpy(h_sets + s0, P.sets.data(), P.sets.size() * sizeof(Settings));
            for (size_t q = 0; q < P.subs.size(); q++) h_subs[sub0 + q] = SubDesc{P.subs[q].range + r0, P.subs[q].s};
            for (size_t q = 0; q < P.big.size(); q++) h_big[big0 + q] = P.big[q] + r0;
            FSlot sl;
            memset(&sl, 0, sizeof sl);
            sl.L = ct.len; sl.out_cap = P.out_cap; sl.pool_cap = P.pool_cap;
            sl.recs = ct.d_recs;
            sl.rec_off = P.all_sn ? (snp_only ? nullptr : S.off_dummy.p + cand0) : ct.d_off;
            sl.pool = ct.d_pool + PAD;
            sl.dyn = f->d_dyn + ct.index;
            sl.seq = it.seq; sl.K = (uint32_t)P.K; sl.cand_off = cand0; sl.blk_off = blk0; sl.nb = nb;
            sl.range_off = r0; sl.leaf_off = leaf0; sl.sub_off = sub0;
            h_slots[si] = sl;
            if (!snp_only) for (uint32_t q = 0; q < nb; q++) h_bslot[blk0 + q] = si;
            r0 += (uint32_t)P.fr.size(); s0 += (uint32_t)P.sets.size(); sub0 += (uint32_t)P.subs.size(); big0 += (uint32_t)P.big.size();
            leaf0 += (uint32_t)P.n_leaves; cand0 += nb * OB_BLOCK; blk0 += nb;
        }
    }
    c->fast->prof.lap(2);
    MSIM_HIP(c, hipMemcpyAsync(S.tab.p, S.h_tab.p, tab_bytes, hipMemcpyHostToDevice, st));
    c->fast->prof.lap(3);
    const FRange *d_ranges = reinterpret_cast<const FRange *>(S.tab.p);
    const Settings *d_sets = reinterpret_cast<const Settings *>(S.tab.p + off_sets);
    const SubDesc *d_subs = reinterpret_cast<const SubDesc *>(S.tab.p + off_subs);
    const uint32_t *d_big = reinterpret_cast<const uint32_t *>(S.tab.p + off_big);
    const FSlot *d_slots = reinterpret_cast<const FSlot *>(S.tab.p + off_slots);
    const uint32_t *d_bslot = reinterpret_cast<const uint32_t *>(S.tab.p + off_bslot);
    const uint32_t nbt = (uint32_t)nb_total;
    uint32_t *blk_S = S.blk_u32.p, *blk_indep = blk_S + (nbt + 1), *blk_in = blk_indep + (nbt + 1), *blk_nrec = blk_in + (nbt + 1),
             *blk_pool = blk_nrec + (nbt + 1), *blk_max1 = blk_pool + (nbt + 1), *blk_max2 = blk_max1 + (nbt + 1);
    // ---- positions
    if (n_big)
        hipLaunchKernelGGL(k_fsplit_top, dim3((uint32_t)n_big), dim3(1024), 0, st, d_ranges, d_big, d_slots, key2, S.sub_k.p, S.sub_c0.p,
                           f->d_flags);
    hipLaunchKernelGGL(k_fsplit_sub, dim3(((uint32_t)n_subs + 3) / 4), dim3(256), 0, st, d_ranges, d_subs, (uint32_t)n_subs, d_slots, key2,
                       S.sub_k.p, S.sub_c0.p, S.leaves.p, f->d_flags, snp_only ? (uint32_t *)nullptr : blk_max1,
                       snp_only ? 0u : 2 * (nbt + 1), S.kept.p, 4 * n_slots, (uint32_t)items[0].P.d);
    c->fast->prof.lap(4);
    const uint32_t bm_words = (1u << lgB_max) / 32 + 2;
    const size_t lds = (size_t)4 * (bm_words + LEAF_LIST / 2) * sizeof(uint32_t);
    const uint32_t leaf_blocks = ((uint32_t)n_leaves + 3) / 4;
    const uint32_t d = (uint32_t)items[0].P.d;
    const Block1 block1 = items[0].P.block1;
    const unsigned long long ti_lim = (unsigned long long)c->params.ti_lim;
    if (snp_only) {
        hipLaunchKernelGGL(k_fleaf<false>, dim3(leaf_blocks), dim3(256), lds, st, d_ranges, S.leaves.p, (uint32_t)n_leaves, bm_words, d_slots,
                           key2, d, (const Settings *)nullptr, block1, ti_lim, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                           (uint8_t *)nullptr, (uint32_t *)nullptr, f->d_flags);
        MSIM_HIP(c, hipGetLastError());
        c->fast->prof.lap(5);
    } else {
        hipLaunchKernelGGL(k_fleaf<true>, dim3(leaf_blocks), dim3(256), lds, st, d_ranges, S.leaves.p, (uint32_t)n_leaves, bm_words, d_slots,
                           key2, d, d_sets, block1, ti_lim, S.cand_pos.p, S.cand_stop.p, S.cand_bend.p, S.cand_meta.p, blk_max1, f->d_flags);
        c->fast->prof.lap(5);
        // One pass = k_fkeep (block-local: free candidates + short cluster walks).  Where everything in front of a block is
        // inside somebody's blocked range it raises the contig's hand-over word and that plan is replayed with the three orbit
        // kernels when the sizes are collected (fast_plan_collect): nothing of the common path pays for them.
        const Prep &P0 = items[0].P;
        auto pass = [&](bool visit, bool final, const uint32_t *end_in, uint32_t *end2, uint32_t maxspan) {
#define MSIM_FK(V, F) hipLaunchKernelGGL((k_fkeep<V, F>), dim3(nbt), dim3(OB_THREADS), 0, st, S.cand_pos.p, end_in, \
                                         (const uint32_t *)(visit ? blk_max2 : blk_max1), end2, blk_max2, S.cand_stop.p, \
                                         S.cand_meta.p, 0u, blk_nrec, blk_pool, S.blk_delta.p, S.kept.p, d_slots, d_bslot)
#define MSIM_FM(V, F) hipLaunchKernelGGL((k_forbit_mark<V, F>), dim3(nbt), dim3(OB_THREADS), 0, st, S.cand_pos.p, end_in, end2, S.cand_stop.p, \
                                         S.cand_meta.p, (uint32_t)P0.K, blk_in, blk_nrec, blk_pool, S.blk_delta.p, S.kept.p, \
                                         (const uint32_t *)nullptr)
            if (!orbit_only) {
                if (!visit && final) MSIM_FK(false, true); else if (!visit) MSIM_FK(false, false); else MSIM_FK(true, true);
            } else {                                       // (one contig: offsets are zero)
                hipLaunchKernelGGL(k_forbit_local, dim3(nbt), dim3(OB_THREADS), 0, st, S.cand_pos.p, end_in, (uint32_t)P0.K, maxspan,
                                   S.blk_out.p, blk_S, blk_indep, (const uint32_t *)nullptr);
                hipLaunchKernelGGL(k_forbit_resolve, dim3(1), dim3(1024), 0, st, S.cand_pos.p, S.blk_out.p, blk_S, blk_indep, (uint32_t)P0.K,
                                   nbt, blk_in, (const uint32_t *)nullptr);
                if (!visit && final) MSIM_FM(false, true); else if (!visit) MSIM_FM(false, false); else MSIM_FM(true, true);
            }
#undef MSIM_FK
#undef MSIM_FM
        };
        // ---- boundary pass (mutator.py:184-213); contigs of one range have no visit filter to fear, but in a batch with one that
        //      has they run it too: it changes nothing for them
        uint32_t maxspan = 1, maxspan_visit = 1;
        for (auto &it : items) { maxspan = std::max(maxspan, it.P.maxspan); maxspan_visit = std::max(maxspan_visit, it.P.maxspan_visit); }
        if (!need_visit) {
            pass(false, true, S.cand_bend.p, nullptr, maxspan);
        } else {
            pass(false, false, S.cand_bend.p, S.cand_end2.p, maxspan);
            // ---- visit filter (mutator.py:376,386,398): the same pass over what the kept records consume
            pass(true, true, S.cand_end2.p, nullptr, maxspan_visit);
        }
        if (!orbit_only || nbt <= 4096) {
            hipLaunchKernelGGL(k_femit<true>, dim3(nbt), dim3(OB_THREADS), 0, st, S.cand_pos.p, S.cand_stop.p, S.cand_meta.p, blk_nrec, blk_pool,
                               S.blk_delta.p, S.kept.p, f->d_flags, d_slots, d_bslot, key2);
        } else {
            hipLaunchKernelGGL(k_fscan, dim3(1), dim3(1024), 0, st, blk_nrec, blk_pool, S.blk_delta.p, nbt, c->contigs[(size_t)items[0].contig].len,
                               P0.out_cap, P0.pool_cap, S.kept.p, f->d_flags, f->d_dyn + items[0].contig);
            hipLaunchKernelGGL(k_femit<false>, dim3(nbt), dim3(OB_THREADS), 0, st, S.cand_pos.p, S.cand_stop.p, S.cand_meta.p, blk_nrec, blk_pool,
                               S.blk_delta.p, S.kept.p, f->d_flags, d_slots, d_bslot, key2);
        }
        MSIM_HIP(c, hipGetLastError());
    }
    c->fast->prof.lap(6);
    if (f->lane_of.size() < c->contigs.size()) f->lane_of.resize(c->contigs.size(), 0);
    for (auto &it : items) f->lane_of[(size_t)it.contig] = (uint8_t)li;
    // ---- per contig: what the host knows now; its APPLY follows on the same stream (apply.hip: Contig::apply_stream) -- no event,
    // no cross-stream wait; whoever reads the records from another stream (text, fetches) drains first
    for (auto &it : items) {
        Contig &ct = c->contigs[(size_t)it.contig];
        ct.apply_stream = st;
        if (!it.P.snp_only) {
            ct.d_dyn = reinterpret_cast<const uint32_t *>(f->d_dyn + ct.index);
            ct.sizes_pending = true;
            ct.off_ready = !it.P.all_sn;
            f->sized.push_back(ct.index);
            if (!orbit_only) {                             // what a replay needs, should the block-local pass hand over
                Replay &rp = f->replay[ct.index];
                rp.key = it.key; rp.seq = it.seq;
                rp.P = std::move(it.P);
            }
        }
    }
    S.pending = true;
    c->fast->prof.lap(7);
    c->fast->prof.n += n_slots;
    return MSIM_OK;
}

// msim_plan_contig of a fast context: the contig's tables are prepared (what the reference refuses is refused here, now) and the
// contig joins the queue; the device work of everything queued goes out in ONE batch at the next call that needs a result
// (fast_plan_flush: every entry point but msim_plan_contig / msim_apply_contig).
int plan_contig_fast(Ctx *c, Contig &ct, const msim_range *ranges, int n_ranges, uint64_t key64, uint32_t seq) {
    int rc;
    if ((rc = ensure_plan(c))) return rc;
    FastPlan *f = c->fast;
    for (const Pending &q : f->queue)
        if (q.contig == ct.index) { if ((rc = fast_plan_flush(c))) return rc; break; }      // planned again before anybody looked
    static const bool no_early = getenv("MSIM_FAST_NO_EARLY_FLUSH") != nullptr;
    // (halves: thirds and quarters measured slower on every bench shape -- more batches, more fixed latency)
    static const bool no_head = getenv("MSIM_FAST_NO_HEAD_FLUSH") != nullptr;
    const bool half = !f->half_done && f->cycle_K >= std::max<uint64_t>(1u << 22, f->last_cycle_K / 2);
    // (100 us and a sixth of the last cycle: 60 us / an eighth, 40 / a twelfth, 150 / a fifth, 100 / a quarter measured the same or slower)
    const bool head = !no_head && !f->early_done && f->queued_host_us >= 100.0 && f->queued_K >= std::max<uint64_t>(1u << 21, f->last_cycle_K / 6);
    if (!no_early && f->queue.size() >= 3 && (half || head)) {
        f->early_done = true;                              // (the contigs queued so far have had their msim_apply_contig)
        if (half) f->half_done = true;
        if ((rc = fast_plan_flush(c))) return rc;
    }
    Pending it;
    c->fast->prof.start();
    const auto tp0 = std::chrono::steady_clock::now();
    rc = prepare(c, ct.len, ranges, n_ranges, it.P);
    f->queued_host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp0).count();
    c->fast->prof.lap(0);
    if (rc) return rc;
    const Prep &P = it.P;
    ct.n_rec = P.snp_only ? P.K : 0;
    ct.n_rec_cap = P.K;
    ct.pool_len = 0;
    ct.plan_empty = P.K == 0;
    ct.all_snp = P.all_sn;
    ct.d_dyn = nullptr;
    ct.sizes_pending = P.K != 0 && !P.snp_only;
    ct.out_cap_len = P.out_cap;
    ct.n_struct_est = P.n_struct_est;
    ct.planned = true;
    if (!P.K) return MSIM_OK;
    it.contig = ct.index; it.key = key64; it.seq = seq; it.apply = false;
    f->queued_K += P.K;
    f->cycle_K += P.K;
    f->queue.push_back(std::move(it));
    return MSIM_OK;
}

bool fast_plan_queued(Ctx *c, int contig, bool mark_apply) {
    FastPlan *f = c->fast;
    if (!f) return false;
    for (Pending &q : f->queue)
        if (q.contig == contig) { if (mark_apply) q.apply = true; return true; }
    return false;
}

int fast_plan_flush(Ctx *c) {
    FastPlan *f = c->fast;
    if (!f || f->queue.empty()) return MSIM_OK;
    std::vector<Pending> queue;
    queue.swap(f->queue);
    const bool force_orbit = getenv("MSIM_FAST_FORCE_ORBIT") != nullptr;
    int rc = MSIM_OK;
    std::vector<Pending> batch;
    uint64_t cand = 0, leaves = 0;
    auto go = [&](bool orbit) {
        if (batch.empty()) return MSIM_OK;
        std::vector<int> applies;
        for (const Pending &q : batch) if (q.apply) applies.push_back(q.contig);
        int r = enqueue_batch(c, batch, orbit);
        batch.clear();
        cand = leaves = 0;
        if (!r && !applies.empty()) r = apply_batch_device(c, applies, true);
        return r;
    };
    // A genome goes out in a few batches rather than one: the APPLYs of a batch (HBM-bound) then run beside the PLAN kernels of
    // the next one (latency- and issue-bound) on another stream.  MSIM_FAST_BATCHES overrides the number (1 = everything at once).
    uint64_t total = 0;
    for (const Pending &q : queue) total += q.P.K;
    f->queued_K = 0;
    f->queued_host_us = 0;
    static const int env_batches = getenv("MSIM_FAST_BATCHES") ? std::max(1, atoi(getenv("MSIM_FAST_BATCHES"))) : 2;
    const int want_batches = f->early_done ? 1 : env_batches;   // (a cycle whose first half went out early: each flush is one batch)
    const uint64_t cut = queue.size() >= 6 && total >= (1u << 22) ? total / (uint64_t)want_batches + 1 : ~0ull;
    uint64_t in_batch = 0;
    for (Pending &q : queue) {
        const uint64_t nb = (q.P.K + OB_BLOCK - 1) / OB_BLOCK;
        const bool solo = !q.P.snp_only && (force_orbit || nb > 4096);      // (k_fkeep / k_femit<true> read all block words in front of a block)
        if (solo || cand + nb * OB_BLOCK >= (1ull << 31) || leaves + q.P.n_leaves >= (1ull << 31) || batch.size() >= 4096 ||
            (!batch.empty() && batch[0].key != q.key) || (!batch.empty() && in_batch + q.P.K / 2 > cut)) {
            if ((rc = go(false))) return rc;
            in_batch = 0;
        }
        in_batch += q.P.K;
        cand += nb * OB_BLOCK;
        leaves += q.P.n_leaves;
        batch.push_back(std::move(q));
        if (solo && (rc = go(true))) return rc;
    }
    return go(false);
}

// ======================================================================================== host restatement (test support)
// The engine, sequentially: the same draws (fast_math.h), but the boundary pass and the visit filter written the way the
// reference writes them (mutator.py:184-213: a blocked range that is reset per range; :318-421: the rewrite's walk) rather
// than as orbits -- so that the CPU tier holds the kernels' formulation against the plain one (tests/test_fast_host.py).
int fast_plan_emulated(Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges, uint64_t key64, uint32_t seq, HostPlan &out) {
    Prep P;
    int rc = prepare(c, L, ranges, n_ranges, P);
    if (rc) return rc;
    const Key key{(uint32_t)key64, (uint32_t)(key64 >> 32), seq};
    out.recs.clear();
    out.pool.clear();
    out.empty = true;
    struct Kept { uint32_t pos, stop, ord; uint8_t type, aux; };
    std::vector<Kept> kept;
    for (size_t ri = 0; ri < P.fr.size(); ri++) {
        const FRange &R = P.fr[ri];
        const uint32_t lgB = R.lgB, B = 1u << lgB;
        const uint32_t T = (uint32_t)(((uint64_t)R.n + B - 1) >> lgB);
        uint32_t lgP = 0;
        while ((1u << lgP) < T) lgP++;
        std::vector<uint32_t> m((size_t)1 << lgP, 0);
        m[0] = R.k;
        for (uint32_t lev = 0; lev < lgP; lev++) {
            const uint32_t S = 1u << (lgP - lev), half = S >> 1;
            for (uint32_t i = 0; i < (1u << lev); i++) {
                const uint32_t a = i * S, mid = a + half;
                if (mid >= T) continue;
                const uint32_t Kn = m[a];
                const uint64_t va = (uint64_t)a << lgB, vm = (uint64_t)mid << lgB, vb = std::min((uint64_t)(a + S) << lgB, (uint64_t)R.n);
                const uint32_t kl = (uint32_t)hypergeometric(vm - va, vb - vm, Kn, key, (1u << lev) + i, (uint32_t)ri);
                m[a] = kl;
                m[mid] = Kn - kl;
            }
        }
        uint32_t ord = R.cand_base;
        int64_t blocked_end = 0;                                         // last_mut_range = range(0)   mutator.py:184
        std::vector<uint8_t> seen;
        for (uint32_t t = 0; t < T; t++) {
            const uint32_t v0 = t << lgB, len = std::min(B, R.n - v0), mt = m[t];
            if (!mt) continue;
            const bool inv = 2 * mt > len;
            const uint32_t need = inv ? len - mt : mt;
            seen.assign(len, 0);
            for (uint32_t have = 0, j = 0; have < need; j++) {
                const uint32_t v = leaf_draw(key, R.leaf_base + t, j, len);
                if (!seen[v]) { seen[v] = 1; have++; }
            }
            for (uint32_t v = 0; v < len; v++) {
                if ((seen[v] != 0) == inv) continue;
                const uint32_t pos = R.start + v0 + v + (uint32_t)P.d * (ord - R.cand_base);
                const Cand cd = cand_draw(key, ord, pos, L, P.sets[R.set], P.block1.v, R.clip, c->params.ti_lim);
                const uint32_t o = ord++;
                if ((int64_t)pos < blocked_end) continue;                // mutator.py:190-192
                if (cd.meta & CAND_DROPPED) continue;                    // mutator.py:199-201
                const uint8_t ty = cd.meta & 7;
                blocked_end = (ty == MSIM_SN || ty == MSIM_IN) ? (int64_t)pos + P.block1.v[ty] : (int64_t)cd.stop + P.block1.v[ty];
                kept.push_back({pos, cd.stop, o, ty, (uint8_t)((cd.meta >> CAND_AUX_SHIFT) & 3)});
            }
        }
    }
    out.empty = kept.empty();
    int64_t cover = -1;                                                  // last base an earlier visited record consumed
    for (const Kept &q : kept) {
        if ((int64_t)q.pos <= cover) continue;                           // never visited   mutator.py:376,386,398
        msim_record rec;
        rec.pos = q.pos; rec.stop = q.stop; rec.extra = 0; rec.type = q.type; rec.aux = 0; rec.rsv = 0;
        if (q.type == MSIM_SN) rec.aux = q.aux;
        if (q.type == MSIM_IN) {
            rec.extra = (uint32_t)out.pool.size();
            const uint32_t len = q.stop - q.pos + 1;
            for (uint32_t c0 = 0; c0 < len; c0 += 64) {
                const U4 ch = draw4(key, c0 >> 6, q.ord, TAG_INS);
                for (uint32_t j = 0; j < std::min(64u, len - c0); j++) out.pool.push_back(insert_base_of(ch, j));
            }
        }
        if (q.type == MSIM_DE || q.type == MSIM_DU || q.type == MSIM_IV) cover = q.stop;
        out.recs.push_back(rec);
    }
    return MSIM_OK;
}

}  // namespace msim

// ---- test support: the arithmetic of fast_math.h from the host (tests/test_fast_host.py holds it against the numpy restatement
// and against the exact hypergeometric law) -- no context, no GPU
extern "C" {

// out[i] = Hypergeometric(good, bad, sample) drawn from node counter node0 + i of drawing range `range`
int msim_dbg_fast_hypergeom(uint64_t good, uint64_t bad, uint64_t sample, uint64_t key, uint32_t seq, uint32_t node0, uint32_t range,
                            uint64_t n, uint64_t *out) {
    if (!out || sample > good + bad) return MSIM_ERR_ARG;
    const msim::fastrng::Key k{(uint32_t)key, (uint32_t)(key >> 32), seq};
    for (uint64_t i = 0; i < n; i++) out[i] = msim::fastrng::hypergeometric(good, bad, sample, k, node0 + (uint32_t)i, range);
    return MSIM_OK;
}
// plans of this context that were replayed with the orbit kernels (the block-local boundary pass handed over)
int msim_dbg_fast_replays(msim_ctx *p, uint64_t *n) {
    msim::Ctx *c = reinterpret_cast<msim::Ctx *>(p);
    if (!c || !n) return MSIM_ERR_ARG;
    *n = c->fast ? c->fast->replays : 0;
    return MSIM_OK;
}
// op 0: d_log(x)  1: d_sqrt(x)  2: log_factorial((uint64) x)  3: d_log1p(x)  4: log_factorial_diff((uint64) x, (int64) y[i])
int msim_dbg_fast_math(int op, const double *x, const double *y, uint64_t n, double *out) {
    if (!x || !out) return MSIM_ERR_ARG;
    for (uint64_t i = 0; i < n; i++) {
        switch (op) {
            case 0: out[i] = msim::fastrng::d_log(x[i]); break;
            case 1: out[i] = msim::fastrng::d_sqrt_up(x[i]); break;
            case 2: out[i] = msim::fastrng::log_factorial((uint64_t)x[i]); break;
            case 3: out[i] = msim::fastrng::d_log1p(x[i]); break;
            case 4: if (!y) return MSIM_ERR_ARG; out[i] = msim::fastrng::log_factorial_diff(msim::fastrng::lf_base((uint64_t)x[i]), (int64_t)y[i]); break;
            default: return MSIM_ERR_ARG;
        }
    }
    return MSIM_OK;
}

}

