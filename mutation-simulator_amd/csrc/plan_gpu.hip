// PLAN on the device: host orchestration of the three engines that reproduce the reference's two MT19937
// streams bit for bit (kernels: plan_kernels.h).                                    gfx950 (MI355X) only.
//
// What the reference does per contig (mutator.py:144-226, util.py:94-109, mutator.py:428-463):
//   random.sample(range(n), k)  -> first k DISTINCT accepted draws of  word >> (32 - bits) < n
//   sorted, + d * rank          -> candidate positions
//   numpy.random.choice(types)  -> a type per candidate (2 NumPy words each)
//   boundary pass               -> keep / drop, randint() for lengths
//   per kept SNP, in position order: uniform(0,1) [2 words], transversion: randbelow(2) [>= 1 word]
//
// All of it is a deterministic function of the word streams, so it can be evaluated out of order as long as
// the *stream positions* come out identical.  The device owns both streams (jump-ahead cascade + chunk
// generation on side streams, ensure_words) and does all per-record work; what differs between the engines is
// how much of the decision logic is a sequential chain that the host has to walk (over device-generated words):
//   plan_contig_gpu            SNP sampler: nothing (fully asynchronous)            -- BASELINE config 2
//   plan_contig_gpu_mixed      SV mix: the boundary chain over non-SNP candidates   -- BASELINE configs 3, 5
//   plan_contig_gpu_hostsample many small deterministic-SNP ranges: the stream cuts   -- BASELINE config 4
// Everything else (translocations, overlapping ranges, ValueError cases) goes through plan_host.cpp.
// DESIGN.md section 3 has the reasoning and the measurements.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <sys/mman.h>

#include "ctx.h"
#include "mt_jump_table.h"
#include "plan_gpu.h"
#include "plan_kernels.h"

namespace msim {

// ====================================================================== host orchestration
struct GpuStream {
    uint32_t *d_raw = nullptr;          // x[0 .. cap)
    uint64_t cap = 0;
    uint32_t *d_states = nullptr;       // chunk start states
    uint32_t states_cap = 0;            // allocated states
    uint32_t n_states = 0;              // valid states = n_src * (m_done + 1)
    uint32_t lvl = 0, n_src = 1, m_done = 0;   // cascade position: level, states at its start, multipliers done
    SnpMap *d_maps = nullptr;           // CPython stream only: SNP transducer maps of the absolute 8192-word blocks
    SnpLane *d_lanes = nullptr;         //   ... and per lane of every block its masks + map (plan_kernels.h: SnpLane), maps_cap blocks
    size_t maps_cap = 0;                //   (entries)
    uint32_t mapped_blocks = 0;         //   blocks [0, mapped_blocks) are mapped (enqueued on the generation stream)
    unsigned long long maps_ti_lim = 0; //   with this transition threshold
    uint32_t *d_z = nullptr;            // extensions of the current level's source states (k_mt_extend)
    size_t z_cap = 0;                   // in source states
    int z_lvl = -1;                     // level whose sources d_z holds
    uint32_t n_chunks = 0;              // chunks generated
    uint64_t pos = 0;                   // next unconsumed index into x (exact, host copy)
    uint64_t last_session_words = 0;    // words the previous (re)seeded session asked for (its windows' ends): sizing hint
    uint64_t max_upto = 0;              // ... of this session so far
    bool live = false;                  // device copy is the authoritative stream
    // generation runs on its own HIP stream; batch b covers chunks [.., ready_hi[b]) and signals ready_ev[b]
    std::vector<uint32_t> ready_hi;
    std::vector<hipEvent_t> words_ev;   // ... and words_ev[b] as soon as its words are there (the SNP maps of the batch follow)
    uint32_t waited_words = 0;          // the plan stream already waited for the WORDS of chunks below this
    // A session's first two cascade levels and its first generation batch run ON the plan stream (nothing else is there yet and
    // the first contig's chain waits for exactly them): every hand-over between streams costs 50-80 us of idle queue, three of
    // them sat in front of the first chain kernel.  casc_ev: behind the last such launch, for the jump stream to wait on.
    hipEvent_t casc_ev = nullptr;
    bool casc_pending = false;
    std::vector<hipEvent_t> ready_ev;
    uint32_t waited_chunks = 0;         // the plan stream already waited for chunks below this
};

constexpr int N_SETS = 32;               // scratch sets: one per in-flight contig, so the plan stream does not have to
                                         // wait for the emit stream (a cross-stream wait costs tens of us of idle
                                         // queue even when it is about to be satisfied); 288 GB of HBM make this free

struct SampleSet {                       // scratch of one sampled range
    uint32_t *acc = nullptr; size_t acc_cap = 0;          // accepted draws, stream order
    uint32_t *cnt = nullptr; size_t cnt_cap = 0;          // per-workgroup accept counts / offsets
    uint32_t *bitmap = nullptr; size_t bm_cap = 0;        // n-bit de-dup bitmap (bytes)
    uint32_t *cnt2 = nullptr; size_t cnt2_cap = 0;        // per-workgroup popcounts / ranks
    uint32_t *bins = nullptr; size_t bins_cap = 0;        // first-k draws grouped by value bin
    uint32_t *cursors = nullptr; size_t cursors_cap = 0;  // per-bin fill counters
    // anchored windows (enqueue_sample_ahead): bookkeeping of the core window and of the head interval, the accepted draws of
    // the head interval with their per-block offsets, and the event behind the set's off-chain work
    uint8_t *ahead = nullptr;                             // PlanState core | PlanState head | SpecHdr
    uint32_t *hcnt = nullptr; size_t hcnt_cap = 0;
    uint32_t *hacc = nullptr; size_t hacc_cap = 0;
    hipEvent_t prep_done = nullptr;
    hipEvent_t chain_done = nullptr;                      // behind the set's last kernel on the plan stream (anchored windows)
    uint8_t last_user = 0;                                // since the last finish: 1 a sample on the chain, 2 an anchored window
    hipEvent_t emit_done = nullptr;
    hipEvent_t wait_ev = nullptr;                         // what marks its last emit: its own emit_done or its group's event
    bool pending = false;
};
struct SnpSet {                          // scratch of one contig's SNP draws
    SnpMap *maps = nullptr; size_t cap = 0;
    uint8_t *aux8 = nullptr; size_t aux_cap = 0;          // one-range contigs: the outcomes by rank (folded into the records by k_bitmap_expand)
    unsigned long long *base = nullptr;                   // device word: stream position the draws start at
    hipEvent_t emit_done = nullptr;
    hipEvent_t wait_ev = nullptr;                         // (as SampleSet::wait_ev)
    bool pending = false;
};

struct SnpDefer { SnpSet *T; uint32_t W2, nb2; };           // what a deferred emit pass needs (enqueue_snp_stage)
// A contig whose chain is enqueued and whose emission waits for its group (plan_kernels.h: EmitJobs)
struct EmitItem { int contig; SampleSet *S; SnpSet *T; uint32_t bmw, bnb, start, K, W2, nb2; msim_record *recs; bool apply; };

struct MixedSet {                        // scratch of one SV-mix range (section 6)
    uint32_t *cand_pos = nullptr; size_t cap_pos = 0;     // candidates in position order
    uint8_t *cand_type = nullptr; size_t cap_type = 0;    // MSIM_* id | KEEP_BIT
    uint32_t *cand_stop = nullptr; size_t cap_stop = 0;   // non-SNP candidates: stop from the host chain
    uint32_t *nsn_pos = nullptr; size_t cap_npos = 0;     // compacted non-SNP candidates
    uint8_t *nsn_type = nullptr; size_t cap_ntype = 0;
    uint32_t *nsn_rank = nullptr; size_t cap_nrank = 0;   // their candidate index
    uint32_t *nsn_stop = nullptr; size_t cap_nstop = 0;
    uint32_t *sn_index = nullptr; size_t cap_snidx = 0;   // kept-SNP ordinal -> record index
    uint32_t *cnt = nullptr; size_t cap_cnt = 0;          // five per-workgroup counter arrays
    uint32_t *words = nullptr; size_t cap_words = 0;      // tempered word window for the host chain
    unsigned long long *p0_slot = nullptr;                // host-cut contigs: stream position their window started at
    WalkRange *walk_d = nullptr; size_t cap_walk_d = 0;   // device-walked / host-cut contigs: range table
    WalkRange *walk_h = nullptr; size_t cap_walk_h = 0;   //   its pinned staging (one per set: the copy is asynchronous)
    uint32_t *wbits = nullptr; size_t cap_wbits = 0;      //   contig-wide bitmap of sampled positions (zero between uses)
    bool wbits_dirty = false;                             //   a failed pass may have left bits behind
    uint32_t *wcnt = nullptr; size_t cap_wcnt = 0;        //   per-workgroup popcounts / ranks of its expansion
    uint32_t *tables = nullptr; size_t cap_tables = 0;    // host-chain engine: accept tables over the word window
    uint32_t *cand_extra = nullptr; size_t cap_cextra = 0;   //   translocations: linked span start per candidate,
    uint8_t *cand_aux = nullptr; size_t cap_caux = 0;        //   flags (reversed / insert_pos > 0 / tombstone),
    uint32_t *nsn_extra = nullptr; size_t cap_nextra = 0;    //   and their staging in chain order
    uint8_t *nsn_aux = nullptr; size_t cap_naux = 0;
    uint8_t *mm_d = nullptr; size_t cap_mm_d = 0;         //   range table | settings' type tables | visit_from
    uint8_t *mm_h = nullptr; size_t cap_mm_h = 0;         //   its pinned staging (one per set: the copies are asynchronous)
    hipEvent_t emit_done = nullptr;
    bool pending = false;
};

struct GpuPlan {
    GpuStream s[2];
    MixedSet mixed[N_SETS];
    uint32_t mixed_unit = 0;
    uint32_t *h_words = nullptr; size_t cap_h_words = 0;  // pinned host staging of the boundary chain
    uint32_t *h_npos = nullptr; size_t cap_h_npos = 0;
    uint8_t *h_ntype = nullptr; size_t cap_h_ntype = 0;
    uint32_t *h_nstop = nullptr; size_t cap_h_nstop = 0;
    uint32_t *h_win = nullptr; size_t cap_h_win = 0;      // host-chain engine: the tempered word window (h_words holds the tables)
    uint32_t *h_nrank = nullptr; size_t cap_h_nrank = 0;  //   candidate ordinals of the chain
    uint32_t *h_nextra = nullptr; size_t cap_h_nextra = 0;   //   translocations: what __link_tls decided, chain order
    uint8_t *h_naux = nullptr; size_t cap_h_naux = 0;
    MixSets mm_sets;                                       //   what gpu_plan_multimix_eligible derived for the contig being planned
    const msim_range *mm_for = nullptr; int mm_n = 0;
    uint32_t *h_seed = nullptr;         // pinned staging of the two host generator states (reseed without a stream sync)
    hipEvent_t seed_ev[2] = {nullptr, nullptr};   //   recorded behind the copies that read a slot
    hipStream_t gen_stream = nullptr;   // chunk generation (latency-bound, ~300 us per batch)
    hipStream_t jump_stream = nullptr;  // jump cascade: never waits for a generation batch
    std::vector<hipEvent_t> ev_pool;
    uint32_t *d_poly = nullptr;
    PlanState *d_ps = nullptr;
    PlanState *h_mail = nullptr;        // pinned, device-visible mailbox
    // handover without a stream synchronisation (SV mix): sequence number of the mailbox (never 0), written last
    uint32_t *h_sig = nullptr;
    uint32_t epoch = 0;
    hipStream_t copy_stream = nullptr;  // candidates D2H beside the plan stream
    hipEvent_t ev_cand = nullptr, ev_piece[3] = {}, ev_cpiece[3] = {};
    SampleSet sample[N_SETS];
    SnpSet snp[N_SETS];
    uint32_t unit = 0, snp_unit = 0;    // rotation counters
    hipEvent_t chain_ev[2 * N_SETS] = {};
    uint32_t chain_i = 0;
    // ONE event per emission group, behind its rewrite launch, instead of two per contig between the expansion and the rewrite:
    // an event record is a packet on the queue (~6-10 us of it each) -- four of them and the APPLY's two stood between a pair's
    // expansion and its rewrite (70 us in the trace of a step whose emission train is the bound)
    hipEvent_t grp_ev[2 * N_SETS] = {};
    uint32_t grp_i = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;   // chain-time span since the last finish
    bool ps_valid = false;              // device PlanState carries the current session's position
    bool unverified = false;            // work enqueued since the last finish (flags / exact position unknown)
    std::vector<EmitItem> emit_items;   // SNP sampler: contigs waiting for their emission group (gpu_emit_flush)
    uint32_t emit_d = 1;                //   ... and the sampling distance they share
    uint64_t verified_pos = 0;          // exact position at the last finish
    uint64_t reserve_words[2] = {0, 0};
    // per context, read when it is created (tests run several group sizes / stream spans in one process):
    int emit_group = 0;                 //   contigs per emission group (MSIM_EMIT_GROUP; 0 = by the context's role: see plan_contig_gpu)
    int emit_train = 0;                 //   launches of an emission group's train: 3, 6, or 0 = by the context's role (gpu_emit_flush; MSIM_EMIT_TRAIN)
    uint32_t max_chunks = MT_JUMP_MAX_CHUNKS;   //   chunks one (re)seeded session may span (MSIM_DBG_JUMP_MAX_CHUNKS lowers it)
    uint32_t rebases = 0;               //   sessions ended because the next contig would not fit into the span
    // anchored windows: the CPython stream's position as the host knows it between two exact readings -- mean and variance of
    // the words consumed since (the sampler's rejections and duplicates, the SNP draws' transversion loops), a hard lower bound
    hipStream_t prep_stream = nullptr;  //   the off-chain part of the samples: the stream of the sample being enqueued,
    hipStream_t prep_streams[8] = {};   //   one of these in turn (contigs' off-chain parts are independent of each other)
    // THREE streams at NORMAL priority + ONE in the low-priority pool (round 6; four at the chain's priority in round 5).  Measured
    // on one box, c2 3 Gb, ms per step of a rank that owns 0 / 3 / 12 of 24 contigs (tools/compat_steps.py,
    // profiles/r06_prep_streams.txt):
    //   4 at the chain's priority 2.27 / 2.46 / 3.45   3 there 2.24 / 2.46 / 3.62   2 there 2.11 / 2.93 / 4.03   1 there 2.00 / 4.35 / 5.3
    //   4 normal 2.13 / 2.77 / 3.52   3 normal 2.10 / 2.29 / 3.30   2 normal 2.08 / 2.61 / 3.59   1 normal 1.98 / 3.8 / 4.6
    // and for a rank that owns everything (profiles/r06_prep_stream_pools.txt): 3 normal 3.60-3.70, **3 normal + 1 low 3.43-3.62**,
    // 2 normal + 1 low 3.77-3.82, 3 normal + 1 at the chain's priority 3.67-3.69, 3 normal + 2 low 4.0.
    // Stream layouts are an EMPIRICAL matter on this runtime (up to four hardware queues per priority, which outlive the streams
    // that asked for them), and a layout can slow every LATER pass of the process that has a host chain by ~6 us per kernel launch
    // (the SV-mix engine's plan spans 7.0 -> 10.0 ms per 3 Gb step, c3 30 -> 35 ms): round 5's four streams at the chain's priority
    // did (its "regression" of c3 from 94 to 84 Gbases/s in the driver's line, which measures c3 behind chain-only c2 steps), so do
    // three there, a fourth stream in the low-priority pool beside this one's three (copy stream), GPU_MAX_HW_QUEUES >= 6.  Most
    // of it reads as "nine queues in the process is one too many" (this context: plan | emission, three side streams, the SV-mix
    // engine's copy stream sharing one of them | generation, jump cascade, the fourth side stream = eight) -- round 5's layout
    // does not fit a pure count, and an isolated probe of launch latency against the number of streams shows nothing
    // (tools/hw_queue_probe.py, profiles/r06_hw_queue_probe.txt): the effect needs the real kernels.  NOTES section 10 has every A/B.
    int n_prep = 4, n_prep_low = 1, prep_i = 0, prep_prio = 0;
    double ahead_sigma = 8.0;
    uint32_t prep_waited[8] = {};       //   ... already waited for the words of chunks below this (per stream)
    // Who plans ahead.  Round 5: only a context that has been asked for msim_plan_chain in this pass or the last one (a rank that
    // owns every contig was bound by its emission + APPLY train either way: 4.1 ms on the chain, 4.2-4.7 ahead).  Round 6: with the
    // train in three launches, one event per group and groups of four (gpu_emit_flush) everybody: 3.66 ms against 4.03-4.12
    // (profiles/r06_emission_train.txt).
    int ahead = 2;                      //   0 never (MSIM_NO_AHEAD), 1 sharded ranks only (round 5's default), 2 always
    uint32_t pass_chain_only = 0;       //   msim_plan_chain calls in this pass
    bool sharded_rank = false;          //   ... there were some in this pass or in the one before
    bool est_ok = false;
    double est_e = 0, est_v = 0;
    uint64_t est_lo = 0;
};

static int grow(Ctx *c, void **p, size_t *cap, size_t want_bytes, bool *grew) {
    if (*cap >= want_bytes) return MSIM_OK;
    if (!*grew) {                         // a buffer is about to be replaced: nothing may be in flight
        if (c->gpu) for (auto st : c->gpu->prep_streams) if (st) MSIM_HIP(c, wait_stream(st));
        MSIM_HIP(c, wait_stream(c->stream));
        MSIM_HIP(c, wait_stream(c->emit_stream));
        *grew = true;
    }
    if (*p) MSIM_HIP(c, hipFree(*p));
    *p = nullptr; *cap = 0;
    const size_t sz = want_bytes + want_bytes / 4 + 4096;
    MSIM_HIP(c, hipMalloc(p, sz));
    *cap = sz;
    return MSIM_OK;
}

GpuPlan *gpu_plan_create() {
    GpuPlan *g = new GpuPlan();
    if (const char *e = getenv("MSIM_EMIT_GROUP")) g->emit_group = std::min(EMIT_G, std::max(0, atoi(e)));
    if (const char *e = getenv("MSIM_DBG_JUMP_MAX_CHUNKS"))
        g->max_chunks = (uint32_t)std::min<long>(MT_JUMP_MAX_CHUNKS, std::max<long>(2, atol(e)));
    if (const char *e = getenv("MSIM_EMIT_TRAIN")) g->emit_train = atoi(e) == 3 ? 3 : atoi(e) == 6 ? 6 : 0;
    if (getenv("MSIM_NO_AHEAD")) g->ahead = 0;
    else if (const char *e = getenv("MSIM_AHEAD")) g->ahead = std::min(2, std::max(0, atoi(e)));
    if (const char *e = getenv("MSIM_PREP_STREAMS")) { g->n_prep = std::min(8, std::max(1, atoi(e))); g->n_prep_low = 0; }
    if (const char *e = getenv("MSIM_PREP_LOW")) g->n_prep_low = std::min(g->n_prep, std::max(0, atoi(e)));
    if (const char *e = getenv("MSIM_PREP_PRIO")) g->prep_prio = atoi(e);
    if (const char *e = getenv("MSIM_AHEAD_SIGMA")) g->ahead_sigma = std::min(16.0, std::max(0.0, atof(e)));
    return g;
}

// Page-locked staging buffers of the engines with a host chain (grow_host): plain 2 MB-aligned memory on huge pages,
// registered with the runtime.  Measured on the bench box per GiB: hipHostMalloc 182 ms + 90 ms to free; aligned_alloc +
// MADV_HUGEPAGE + hipHostRegister 43 ms, unregister + free 37 ms -- and copies are just as asynchronous and as fast
// (57 GB/s; into plain touched memory the "async" copy blocks the caller).  A 1 Gb contig needs 1.7 GB of them at once.
// (mapped, not malloc'ed: a freed block must leave the address space -- glibc would hand a heap block to the next caller
//  while the runtime may still know the range as registered.  The size sits in a header page in front of the block.)
constexpr size_t STAGE_HUGE = (size_t)2 << 20;
static void *host_stage_alloc(size_t sz, hipError_t *err) {
    *err = hipSuccess;
    const size_t total = sz + 2 * STAGE_HUGE;              // room to align the block and for the header in front of it
    void *raw = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (raw == MAP_FAILED) return nullptr;
    uint8_t *q = reinterpret_cast<uint8_t *>(((uintptr_t)raw + STAGE_HUGE + STAGE_HUGE - 1) & ~(uintptr_t)(STAGE_HUGE - 1));
    size_t *hdr = reinterpret_cast<size_t *>(q - 4096);
    hdr[0] = (size_t)(uintptr_t)raw;
    hdr[1] = total;
#ifdef MADV_HUGEPAGE
    (void)madvise(q, sz, MADV_HUGEPAGE);
#endif
    *err = hipHostRegister(q, sz, hipHostRegisterDefault);
    if (*err != hipSuccess) { munmap(raw, total); return nullptr; }
    return q;
}
static void host_stage_free(void *p) {
    if (!p) return;
    (void)hipDeviceSynchronize();                          // (what hipHostFree does by itself: no copy may still use the block)
    (void)hipHostUnregister(p);
    const size_t *hdr = reinterpret_cast<const size_t *>(static_cast<uint8_t *>(p) - 4096);
    munmap(reinterpret_cast<void *>((uintptr_t)hdr[0]), hdr[1]);
}

void gpu_plan_destroy(GpuPlan *g) {
    if (!g) return;
    if (g->jump_stream) { (void)wait_stream(g->jump_stream); (void)hipStreamDestroy(g->jump_stream); }
    if (g->gen_stream) { (void)wait_stream(g->gen_stream); (void)hipStreamDestroy(g->gen_stream); }
    for (auto st : g->prep_streams) if (st) { (void)wait_stream(st); (void)hipStreamDestroy(st); }
    for (auto &s : g->s) {
        if (s.d_raw) (void)hipFree(s.d_raw);
        if (s.d_states) (void)hipFree(s.d_states);
        if (s.d_z) (void)hipFree(s.d_z);
        if (s.d_maps) (void)hipFree(s.d_maps);
        if (s.d_lanes) (void)hipFree(s.d_lanes);
        for (auto e : s.ready_ev) (void)hipEventDestroy(e);
        for (auto e : s.words_ev) if (e) (void)hipEventDestroy(e);
        if (s.casc_ev) (void)hipEventDestroy(s.casc_ev);
    }
    for (auto e : g->ev_pool) (void)hipEventDestroy(e);
    for (auto &t : g->sample) {
        if (t.acc) (void)hipFree(t.acc);
        if (t.cnt) (void)hipFree(t.cnt);
        if (t.bitmap) (void)hipFree(t.bitmap);
        if (t.cnt2) (void)hipFree(t.cnt2);
        if (t.bins) (void)hipFree(t.bins);
        if (t.cursors) (void)hipFree(t.cursors);
        if (t.ahead) (void)hipFree(t.ahead);
        if (t.hcnt) (void)hipFree(t.hcnt);
        if (t.hacc) (void)hipFree(t.hacc);
        if (t.prep_done) (void)hipEventDestroy(t.prep_done);
        if (t.chain_done) (void)hipEventDestroy(t.chain_done);
        if (t.emit_done) (void)hipEventDestroy(t.emit_done);
    }
    for (auto &t : g->snp) {
        if (t.maps) (void)hipFree(t.maps);
        if (t.aux8) (void)hipFree(t.aux8);
        if (t.base) (void)hipFree(t.base);
        if (t.emit_done) (void)hipEventDestroy(t.emit_done);
    }
    for (auto &t : g->mixed) {
        void *bufs[] = {t.cand_pos, t.cand_type, t.cand_stop, t.nsn_pos, t.nsn_type, t.nsn_rank, t.nsn_stop, t.sn_index,
                        t.cnt, t.words, t.walk_d, t.wbits, t.wcnt, t.p0_slot, t.tables, t.mm_d, t.cand_extra, t.cand_aux,
                        t.nsn_extra, t.nsn_aux};
        for (void *b : bufs) if (b) (void)hipFree(b);
        host_stage_free(t.walk_h);
        host_stage_free(t.mm_h);
        if (t.emit_done) (void)hipEventDestroy(t.emit_done);
    }
    void *hb[] = {g->h_words, g->h_npos, g->h_ntype, g->h_nstop, g->h_win, g->h_nrank, g->h_nextra, g->h_naux};
    for (void *b : hb) host_stage_free(b);
    for (auto e : g->chain_ev) if (e) (void)hipEventDestroy(e);
    for (auto e : g->grp_ev) if (e) (void)hipEventDestroy(e);
    if (g->t0) (void)hipEventDestroy(g->t0);
    if (g->t1) (void)hipEventDestroy(g->t1);
    if (g->h_seed) (void)hipHostFree(g->h_seed);
    for (auto e : g->seed_ev) if (e) (void)hipEventDestroy(e);
    if (g->d_poly) (void)hipFree(g->d_poly);
    if (g->d_ps) (void)hipFree(g->d_ps);
    if (g->h_mail) (void)hipHostFree(g->h_mail);
    if (g->h_sig) (void)hipHostFree(g->h_sig);
    if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
    if (g->ev_cand) (void)hipEventDestroy(g->ev_cand);
    for (auto e : g->ev_cpiece) if (e) (void)hipEventDestroy(e);
    for (auto e : g->ev_piece) if (e) (void)hipEventDestroy(e);
    delete g;
}

void gpu_plan_invalidate(GpuPlan *g) {
    g->unit = g->snp_unit = g->mixed_unit = 0;            // a new pass: contig i meets scratch set i again (sizes fit)
    g->sharded_rank = g->pass_chain_only > 0;
    g->pass_chain_only = 0;
    for (auto &s : g->s) {
        if (s.live) s.last_session_words = std::max<uint64_t>(s.pos, s.max_upto);
        s.live = false;
    }
}
void gpu_plan_reserve(GpuPlan *g, uint64_t py_words, uint64_t np_words) { g->reserve_words[0] = py_words; g->reserve_words[1] = np_words; }

// Words one (re)seeded session can address: the jump polynomials (mt_jump_table.h) reach MT_JUMP_MAX_CHUNKS chunks from its origin.
static inline uint64_t span_words(const GpuPlan *g) { return (uint64_t)MT_N + (uint64_t)g->max_chunks * MT_CHUNK_WORDS; }

// The reference's generators have no length limit (mutator.py:105-142 draws from the two global streams for as long as the genome
// lasts; util.py:104), the jump tables have: 8192 chunks = 1.31 G words from a session's origin (-sn 0.01: ~20 Gb of genome).
// Before a contig is handed to a device engine: will its windows fit into what is left of the span?  If not, the session is
// RE-BASED -- everything in flight is finished, the 624-word window at each stream's exact position goes to the host generators
// (gpu_plan_sync_to_host: any such window is a valid state) and the engine's stream_to_device makes it the origin of a new
// session; a synchronisation and a fresh cascade every 1.3 G words.  *fits = false: not even an empty span holds this contig
// (hundreds of millions of candidates on one contig) -- AUTO plans it on the host, which has no such limit.
// The need is an upper bound built from the engines' own window formulas (enqueue_sample_chain, enqueue_snp_stage, window_of,
// mixed_link_translocations), summed as if every stage started at the end of the previous one's window.
int gpu_plan_make_room(Ctx *c, GpuPlan *g, const msim_range *ranges, int n_ranges, bool *fits) {
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);
    double py = 0, np = 0, K = 0, K_chain = 0, pool_mean = 0, pool_var = 0, e_small = 0, var_small = 0;
    bool has_tl = false, bad = false;
    int n_big = 0;
    for (int i = 0; i < n_ranges; i++) {
        const msim_range &r = ranges[i];
        if (r.k <= 0) continue;
        const double k = (double)r.k;
        const double n = (double)((r.stop - (r.k - 1) * d) - r.start);
        K += k;
        if (n < k) bad = true;                                           // ValueError: whoever plans it raises it
        else if (n <= (double)r.setsize) { e_small += 2.0 * k; var_small += 2.0 * k; }    // pool path: < 2 words per draw
        else if (n >= 4294967296.0) bad = true;                          // (multi-word getrandbits: no device engine takes it)
        else {                                                           // set path: the first k distinct accepted draws
            const double p_acc = n / (double)(1ull << bit_length64((uint64_t)n));
            const double need = k >= n ? 64.0 * k : -n * std::log1p(-k / n);             // coupon collector
            if (r.k >= 4096) {                                           // a window of its own (enqueue_sample_chain)
                const double target = need + 16.0 * std::sqrt(need) + 4096.0;
                py += target / p_acc + 16.0 * std::sqrt(target) / p_acc + 8192.0;
                n_big++;
            } else {                                                     // small ranges share one window (host-cut / host-chain engines)
                e_small += need / p_acc;
                var_small += need * (1.0 - p_acc) / (p_acc * p_acc) + 4.0 * (need - k) / (p_acc * p_acc) + need / p_acc;
            }
        }
        bool chain = P.block[MSIM_SN] != d;                              // an SNP that blocks rides the chain too
        for (int j = 0; j < r.n_types; j++) {
            const uint64_t lo = j ? r.cdf_thr[j - 1] : 0;
            if (!(r.cdf_thr[j] > lo && lo < (1ull << 53))) continue;
            const int t = r.types[j];
            if (t == MSIM_SN) continue;
            chain = true;
            if (t == MSIM_TL || t == MSIM_TLI) has_tl = true;
            if (t == MSIM_IN) {
                const double q = (double)(std::min<uint64_t>(r.cdf_thr[j], 1ull << 53) - lo) / 9007199254740992.0;
                const double mx = (double)std::max<int64_t>(r.max_len[t], 1);
                pool_mean += k * q * mx;                                 // (every insert at its longest: a bound, not an estimate)
                pool_var += k * q * mx * mx;
            }
        }
        if (chain) K_chain += k;
    }
    py += e_small + 16.0 * std::sqrt(var_small);
    if (K_chain > 0) py += 2.0 * K_chain + 32.0 * std::sqrt(K_chain) + 4096.0;              // randint windows (acceptance >= 1/2)
    if (has_tl) py += 6.0 * K_chain + 16.0 * std::sqrt(25.0 * K_chain + 1.0) + 4096.0;      // __link_tls
    py += 4.0 * K + 16.0 * std::sqrt(4.0 * K) + 16384.0 + 3.0 * SNP_BLOCK2;                 // SNP draws (<= 4 words each, expected)
    py += 2.0 * 65536.0 + 8192.0 * n_big;
    np = 2.0 * K + pool_mean + 16.0 * std::sqrt(pool_var) + 4096.0;
    const double span = (double)(span_words(g) - MT_N);
    *fits = true;
    if (bad) return MSIM_OK;                                             // (the engines' eligibility refuses these themselves)
    *fits = std::isfinite(py) && py < span && np < span;
    if (!*fits) return MSIM_OK;
    const GpuStream &s0 = g->s[0], &s1 = g->s[1];
    const bool live = s0.live || s1.live;
    const double at0 = s0.live ? (double)s0.pos : (double)MT_N, at1 = s1.live ? (double)s1.pos : (double)MT_N;
    if (live && (at0 + py > (double)span_words(g) || at1 + np > (double)span_words(g))) {
        const int rc = gpu_plan_sync_to_host(c, g);          // both streams: exact positions -> host states; the next engine reseeds
        if (rc) return rc;
        g->rebases++;
        c->t.stream_rebases++;
    }
    return MSIM_OK;
}

// Side streams of the stream machinery: the jump cascade and chunk generation run beside the plan chain.
static int ensure_side_streams(Ctx *c, GpuPlan *g) {
    if (g->gen_stream) return MSIM_OK;
    // distinct priorities -> distinct hardware queues (streams of one priority may share a queue and
    // then serialise): the cascade must not queue behind a generation batch
    int lo = 0, hi = 0;
    MSIM_HIP(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
    (void)hi;
    // bulk side work at the LOWEST priority: its workgroups (123 KB of LDS each for a jump) must not crowd out
    // the chain kernels of the first contigs, which already wait for nothing but free CU resources
    MSIM_HIP(c, hipStreamCreateWithPriority(&g->gen_stream, hipStreamNonBlocking, lo));
    MSIM_HIP(c, hipStreamCreateWithPriority(&g->jump_stream, hipStreamNonBlocking, lo));
    return MSIM_OK;
}

// Make x[0 .. upto) available to kernels on the PLAN stream.  Jumps and chunk generation are
// enqueued on the generation stream, level by level (after cascade level r the states of chunks
// < 2^(r+1) exist and those chunks can be generated), each batch followed by an event; the plan
// stream only waits for the batch that covers what it is about to read.  With the sizing hint the
// whole session is enqueued at the first call, so the cascade overlaps the planning of the first
// contigs instead of preceding it.
// maps = false: the caller reads the words only (a sample's chain) -- it need not wait for the batch's SNP maps, which the first
// batch of a step delivers ~110 us behind its words
static int ensure_words(Ctx *c, GpuPlan *g, int si, uint64_t upto, bool maps = true) {
    GpuStream &s = g->s[si];
    {
        int rc0 = ensure_side_streams(c, g);
        if (rc0) return rc0;
    }
    // (the hint for the next session on this context is what this one ASKED for, not what it generated: a first session extends
    //  by doubling, and repeating its 943 chunks where 690 are read cost a third round of jumps and a third more words to
    //  generate and to map, every step: c2 3.89-3.95 -> 3.75-3.87 ms)
    s.max_upto = std::max(s.max_upto, upto);
    const uint64_t have = MT_N + (uint64_t)s.n_chunks * MT_CHUNK_WORDS;
    if (upto > have) {
        // batch the extension: an explicit hint, or what the previous session on this context needed
        // A generation batch costs ~300 us of latency whatever its size, so never extend piecemeal: take
        // what the previous session on this context went through, else at least double the stream.
        const uint64_t span = span_words(g);               // what the jump tables reach from this session's origin
        // (gpu_plan_make_room re-bases the session between contigs so that a plan never gets here; should a window estimate
        //  ever fall short, the caller's recovery for overflowed windows applies: mutator.py re-plans through the host planner)
        if (upto > span) return fail(c, MSIM_ERR_HIP, "GPU sampler: a stream window overflowed its session's jump-table span (results discarded)");
        uint64_t want = std::max<uint64_t>(upto, s.pos + g->reserve_words[si]);
        want = std::max<uint64_t>(want, std::min<uint64_t>(s.last_session_words, want * 64));
        if (s.n_chunks) want = std::max<uint64_t>(want, 2 * have);
        want = std::min<uint64_t>(want, span);             // hints never reach beyond the span
        const uint32_t need_chunks = (uint32_t)((want - MT_N + MT_CHUNK_WORDS - 1) / MT_CHUNK_WORDS);
        // states the cascade will hold once it covers need_chunks: whole levels, then part of one
        uint32_t want_states = 1;
        for (int l = 0; l < MT_JUMP_LEVELS && want_states < need_chunks; l++) {
            const uint32_t R = (uint32_t)MT_JUMP_RADIX[l];
            if ((uint64_t)want_states * R >= need_chunks) { want_states *= (need_chunks + want_states - 1) / want_states; break; }
            want_states *= R;
        }
        if (!g->d_poly) {
            MSIM_HIP(c, hipMalloc(&g->d_poly, sizeof(MT_JUMP_POLY)));
            MSIM_HIP(c, hipMemcpy(g->d_poly, MT_JUMP_POLY, sizeof(MT_JUMP_POLY), hipMemcpyHostToDevice));
        }
        const uint64_t want_cap = MT_N + (uint64_t)need_chunks * MT_CHUNK_WORDS;
        if (want_states > s.states_cap || want_cap > s.cap) {      // grow (rare): quiesce every stream first
            MSIM_HIP(c, wait_stream(g->jump_stream));
            MSIM_HIP(c, wait_stream(g->gen_stream));
            for (auto st : g->prep_streams) if (st) MSIM_HIP(c, wait_stream(st));
            MSIM_HIP(c, wait_stream(c->stream));
            MSIM_HIP(c, wait_stream(c->emit_stream));     // record emission reads the word arrays too
            if (want_states > s.states_cap) {
                uint32_t *ns = nullptr;
                MSIM_HIP(c, hipMalloc(&ns, (size_t)want_states * MT_N * sizeof(uint32_t)));
                if (s.d_states) {
                    MSIM_HIP(c, hipMemcpy(ns, s.d_states, (size_t)s.n_states * MT_N * sizeof(uint32_t), hipMemcpyDeviceToDevice));
                    MSIM_HIP(c, hipFree(s.d_states));
                }
                s.d_states = ns;
                s.states_cap = want_states;
            }
            if (si == 0 && want_cap / SNP_BLOCK2 + 2 > s.maps_cap) {
                SnpMap *nm = nullptr;
                SnpLane *nl = nullptr;
                const size_t cap = (size_t)(want_cap / SNP_BLOCK2 + 2) * 5 / 4;
                MSIM_HIP(c, hipMalloc(&nm, cap * sizeof(SnpMap)));
                MSIM_HIP(c, hipMalloc(&nl, cap * SNP_THREADS * sizeof(SnpLane)));
                if (s.d_maps) {
                    MSIM_HIP(c, hipMemcpy(nm, s.d_maps, (size_t)s.mapped_blocks * sizeof(SnpMap), hipMemcpyDeviceToDevice));
                    MSIM_HIP(c, hipMemcpy(nl, s.d_lanes, (size_t)s.mapped_blocks * SNP_THREADS * sizeof(SnpLane), hipMemcpyDeviceToDevice));
                    MSIM_HIP(c, hipFree(s.d_maps));
                    MSIM_HIP(c, hipFree(s.d_lanes));
                }
                s.d_maps = nm;
                s.d_lanes = nl;
                s.maps_cap = cap;
            }
            if (want_cap > s.cap) {
                uint32_t *nr = nullptr;
                MSIM_HIP(c, hipMalloc(&nr, want_cap * sizeof(uint32_t)));
                if (s.d_raw) {
                    MSIM_HIP(c, hipMemcpy(nr, s.d_raw, have * sizeof(uint32_t), hipMemcpyDeviceToDevice));
                    MSIM_HIP(c, hipFree(s.d_raw));
                }
                s.d_raw = nr;
                s.cap = want_cap;
            }
        }
        static const bool lead_on_plan = getenv("MSIM_NO_LEAD_ON_PLAN") == nullptr;
        auto join_plan_cascade = [&]() -> int {             // the jump stream continues behind what the plan stream did of the cascade
            if (s.casc_pending) {
                MSIM_HIP(c, hipStreamWaitEvent(g->jump_stream, s.casc_ev, 0));
                s.casc_pending = false;
            }
            return MSIM_OK;
        };
        auto take_event = [&](hipEvent_t &ev) -> int {
            if (!g->ev_pool.empty()) { ev = g->ev_pool.back(); g->ev_pool.pop_back(); }
            else MSIM_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            return MSIM_OK;
        };
        while (s.n_chunks < need_chunks) {
            const uint32_t hi = std::min<uint32_t>(need_chunks, s.n_states);
            // A generation batch costs ~110-140 us of latency whatever its size and batches queue in order; a cascade level costs an
            // extension (50 us) and its jumps, 256 of them at a time (one 88 KB workgroup per CU): ~100 us per round.  Radices
            // 256 x 4 x 8 (round 5; 16 x 16 x 16 x 2 before): the 689 jumps of a 3 Gb -sn 0.01 session are three rounds in two
            // launches, both on the plan stream, and ONE generation batch + ONE map pass follow -- the first chain kernel starts
            // later than it did (0.52 ms against 0.44) but nothing of the cascade runs beside the first contigs' chains any more,
            // which used to stretch them to 2-3 x their time: c2 4.04-4.16 -> 3.92-3.95 ms per step, a chain-only rank 2.60 -> 2.43.
            // MSIM_GEN_FIRST_LEVELS=1: the first batch behind the first level already (256 chunks, 41 M words: the first four
            // contigs), the second level beside their chains -- 3.96-4.01.
            static const uint32_t first_levels = getenv("MSIM_GEN_FIRST_LEVELS") ? (uint32_t)std::max(1, atoi(getenv("MSIM_GEN_FIRST_LEVELS"))) : 2u;
            const uint32_t first_states = first_levels >= 2 ? (uint32_t)(MT_JUMP_RADIX[0] * MT_JUMP_RADIX[1]) : (uint32_t)MT_JUMP_RADIX[0];
            if (hi > s.n_chunks && hi >= std::min<uint32_t>(need_chunks, first_states)) {
                // states [n_chunks, hi) are complete: generate those chunks (the session's first batch on the plan stream itself,
                // right behind the cascade levels that ran there; later ones on the generation stream behind the jump stream)
                const bool first = lead_on_plan && s.n_chunks == 0 && s.casc_pending;
                hipStream_t gs = first ? c->stream : g->gen_stream;
                int rc = MSIM_OK;
                if (!first) {
                    if ((rc = join_plan_cascade())) return rc;
                    hipEvent_t st_ev;
                    rc = take_event(st_ev);
                    if (rc) return rc;
                    MSIM_HIP(c, hipEventRecord(st_ev, g->jump_stream));
                    MSIM_HIP(c, hipStreamWaitEvent(g->gen_stream, st_ev, 0));
                    s.ready_ev.push_back(st_ev);            // recycled with the batch events at the next reseed
                    s.ready_hi.push_back(0);
                    s.words_ev.push_back(nullptr);
                }
                hipLaunchKernelGGL(k_mt_generate, dim3(hi - s.n_chunks), dim3(GEN_THREADS), 0, gs, s.d_states,
                                   s.d_raw, s.n_chunks);
                MSIM_HIP(c, hipGetLastError());
                hipEvent_t wev = nullptr;
                if (si == 0 || first) {
                    rc = take_event(wev);
                    if (rc) return rc;
                    MSIM_HIP(c, hipEventRecord(wev, gs));
                    if (first) {                            // the map pass (and whatever follows there) stays on the generation stream
                        MSIM_HIP(c, hipStreamWaitEvent(g->gen_stream, wev, 0));
                        s.waited_words = hi;                // (the plan stream made these words itself)
                    }
                }
                if (si == 0) {                              // SNP transducer maps of every block the batch completed
                    const uint32_t n_words = (uint32_t)(MT_N + (uint64_t)hi * MT_CHUNK_WORDS);
                    const uint32_t complete = n_words / SNP_BLOCK2;
                    if (complete > s.mapped_blocks) {
                        hipLaunchKernelGGL(k_snp_maps_abs, dim3(complete - s.mapped_blocks), dim3(SNP_THREADS), 0, g->gen_stream,
                                           s.d_raw, n_words, (unsigned long long)c->params.ti_lim, s.mapped_blocks, s.d_maps, s.d_lanes);
                        MSIM_HIP(c, hipGetLastError());
                        s.mapped_blocks = complete;
                    }
                }
                hipEvent_t ev;
                rc = take_event(ev);
                if (rc) return rc;
                MSIM_HIP(c, hipEventRecord(ev, g->gen_stream));
                s.ready_hi.push_back(hi);
                s.ready_ev.push_back(ev);
                s.words_ev.push_back(wev);
                s.n_chunks = hi;
            }
            if (s.n_chunks < need_chunks) {                // extend the cascade (mixed radix, one launch per level)
                if (s.m_done == (uint32_t)MT_JUMP_RADIX[s.lvl] - 1) {       // level complete: the next one starts
                    s.lvl++;
                    s.n_src = s.n_states;
                    s.m_done = 0;
                }
                const uint32_t R = (uint32_t)MT_JUMP_RADIX[s.lvl];
                const uint32_t m_need = std::min<uint32_t>(R - 1, (need_chunks + s.n_src - 1) / s.n_src - 1);
                uint32_t level_base = 0;
                for (uint32_t l = 0; l < s.lvl; l++) level_base += (uint32_t)MT_JUMP_RADIX[l] - 1;
                const bool on_plan = lead_on_plan && s.lvl <= 1 && s.n_chunks == 0;
                hipStream_t js = on_plan ? c->stream : g->jump_stream;
                if (!on_plan) { const int rcj = join_plan_cascade(); if (rcj) return rcj; }
                if (s.z_lvl != (int)s.lvl) {               // extend this level's source states once
                    if (s.z_cap < s.n_src) {
                        MSIM_HIP(c, wait_stream(g->jump_stream));   // the old buffer may still be read
                        MSIM_HIP(c, wait_stream(c->stream));
                        if (s.d_z) MSIM_HIP(c, hipFree(s.d_z));
                        s.d_z = nullptr; s.z_cap = 0;
                        MSIM_HIP(c, hipMalloc(&s.d_z, (size_t)s.n_src * JUMP_ZP * sizeof(uint32_t)));
                        s.z_cap = s.n_src;
                    }
                    hipLaunchKernelGGL(k_mt_extend, dim3(s.n_src), dim3(GEN_THREADS), 0, js, s.d_states, s.d_z);
                    MSIM_HIP(c, hipGetLastError());
                    s.z_lvl = (int)s.lvl;
                }
                hipLaunchKernelGGL(k_mt_jump, dim3(s.n_src * (m_need - s.m_done)), dim3(JUMP_THREADS), 0,
                                   js, s.d_states, s.d_z, s.n_src,
                                   g->d_poly + (size_t)level_base * MT_POLY_WORDS, s.m_done + 1);
                MSIM_HIP(c, hipGetLastError());
                if (on_plan) {
                    if (!s.casc_ev) MSIM_HIP(c, hipEventCreateWithFlags(&s.casc_ev, hipEventDisableTiming));
                    MSIM_HIP(c, hipEventRecord(s.casc_ev, c->stream));
                    s.casc_pending = true;
                }
                s.m_done = m_need;
                s.n_states = s.n_src * (s.m_done + 1);
            }
        }
    }
    // the plan stream waits for the batch that covers `upto`
    if (upto > MT_N) {
        const uint32_t need = (uint32_t)((upto - MT_N + MT_CHUNK_WORDS - 1) / MT_CHUNK_WORDS);
        if (need > (maps ? s.waited_chunks : s.waited_words)) {
            for (size_t b = 0; b < s.ready_hi.size(); b++) {
                if (s.ready_hi[b] >= need) {
                    const bool words_only = !maps && s.words_ev[b];
                    MSIM_HIP(c, hipStreamWaitEvent(c->stream, words_only ? s.words_ev[b] : s.ready_ev[b], 0));
                    s.waited_words = std::max(s.waited_words, s.ready_hi[b]);
                    if (!words_only) s.waited_chunks = s.ready_hi[b];
                    break;
                }
            }
        }
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
    {
        int rc0 = ensure_side_streams(c, g);
        if (rc0) return rc0;
    }
    MSIM_HIP(c, wait_stream(g->jump_stream));     // nothing of the old session in flight
    MSIM_HIP(c, wait_stream(g->gen_stream));
    for (auto st : g->prep_streams) if (st) MSIM_HIP(c, wait_stream(st));
    // the state goes through a pinned staging slot (one per stream), so the copies need no host synchronisation:
    // h.mt may change right after; the slot is rewritten only once the copies that read it are known to be done,
    // and the side streams are ordered behind the copies by the same event
    if (!g->h_seed) MSIM_HIP(c, hipHostMalloc(&g->h_seed, 2 * sizeof h.mt, hipHostMallocDefault));
    if (!g->seed_ev[si]) MSIM_HIP(c, hipEventCreateWithFlags(&g->seed_ev[si], hipEventDisableTiming));
    else MSIM_HIP(c, wait_event(g->seed_ev[si]));
    uint32_t *slot = g->h_seed + (size_t)si * MT_N;
    memcpy(slot, h.mt, sizeof h.mt);
    MSIM_HIP(c, hipMemcpyAsync(s.d_states, slot, sizeof h.mt, hipMemcpyHostToDevice, c->stream));
    MSIM_HIP(c, hipMemcpyAsync(s.d_raw, slot, sizeof h.mt, hipMemcpyHostToDevice, c->stream));
    MSIM_HIP(c, hipEventRecord(g->seed_ev[si], c->stream));
    MSIM_HIP(c, hipStreamWaitEvent(g->jump_stream, g->seed_ev[si], 0));
    MSIM_HIP(c, hipStreamWaitEvent(g->gen_stream, g->seed_ev[si], 0));
    if (si == 0) for (auto st : g->prep_streams) if (st) MSIM_HIP(c, hipStreamWaitEvent(st, g->seed_ev[si], 0));
    for (auto e : s.ready_ev) g->ev_pool.push_back(e);
    s.ready_ev.clear();
    s.ready_hi.clear();
    for (auto e : s.words_ev) if (e) g->ev_pool.push_back(e);
    s.words_ev.clear();
    s.waited_chunks = 0;
    s.waited_words = 0;
    if (si == 0) for (auto &w : g->prep_waited) w = 0;
    s.casc_pending = false;
    s.n_states = 1;
    s.lvl = 0; s.n_src = 1; s.m_done = 0;
    s.z_lvl = -1;
    s.n_chunks = 0;
    s.mapped_blocks = 0;
    s.maps_ti_lim = c->params.ti_lim;
    s.pos = (uint64_t)h.idx;
    s.max_upto = 0;
    s.live = true;
    if (si == 0) { g->ps_valid = false; g->verified_pos = s.pos; }
    return MSIM_OK;
}

// device stream -> host generator (any 624-word window ending at pos is a valid state)
int gpu_plan_sync_to_host(Ctx *c, GpuPlan *g) {
    int frc = gpu_plan_finish(c, g);
    if (frc) return frc;
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
        MSIM_HIP(c, wait_stream(c->stream));
        h.state_changed();
        s.last_session_words = std::max<uint64_t>(s.pos, s.max_upto);
        s.live = false;
    }
    return MSIM_OK;
}

bool gpu_plan_eligible(const Ctx *c, const msim_range *ranges, int n_ranges) {
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);
    if (P.block[MSIM_SN] != d) return false;              // an SNP could block its successor
    int64_t prev_stop = -1;
    for (int i = 0; i < n_ranges; i++) {
        const msim_range &r = ranges[i];
        if (r.k == 0) continue;                           // draws nothing (mutator.py:163-164)
        if (r.k < 4096) return false;                     // tiny ranges: the sequential host walk is faster
        // overlapping or unsorted drawing ranges: the reference merges the per-range dicts with update()
        // (mutator.py:121, later range wins) -- only the host planner reproduces that
        if (r.start <= prev_stop) return false;
        prev_stop = r.stop;
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

// Everything enqueued so far has completed: collect the sticky flags and the exact stream position.
int gpu_plan_finish(Ctx *c, GpuPlan *g) {
    {
        const int rc = gpu_emit_flush(c);                  // (a group that is still waiting goes out now)
        if (rc) return rc;
    }
    if (!g->unverified) return MSIM_OK;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, c->stream, g->d_ps, g->h_mail);
    MSIM_HIP(c, hipGetLastError());
    MSIM_HIP(c, hipEventRecord(g->t1, c->stream));
    MSIM_HIP(c, wait_stream(c->stream));
    MSIM_HIP(c, wait_stream(c->emit_stream));
    float ms = 0;
    MSIM_HIP(c, hipEventElapsedTime(&ms, g->t0, g->t1));
    c->t.plan_gpu_ms += ms;
    g->unverified = false;
    for (auto &t : g->sample) { t.pending = false; t.last_user = 0; }
    for (auto &t : g->snp) t.pending = false;
    for (auto &t : g->mixed) t.pending = false;
    const PlanState h = *g->h_mail;
    if (h.flags & (FLAG_SAMPLE_OVERFLOW | FLAG_SNP_OVERFLOW)) {
        g->s[0].live = g->s[1].live = false;
        for (auto &t : g->mixed) t.wbits_dirty = true;
        return fail(c, MSIM_ERR_HIP, "GPU sampler: a stream window overflowed its margin (16 sigma; 8 for the start of a sample planned ahead of the chain; results discarded)");
    }
    c->t.py_words += h.pos - g->verified_pos;
    c->t.snp_ahead_margin_permille = std::max<uint64_t>(c->t.snp_ahead_margin_permille, h.ahead_margin_used);
    g->verified_pos = h.pos;
    g->s[0].pos = h.pos;
    return MSIM_OK;
}

// test support (MSIM_DBG_FORCE_OVERFLOW): what a 16-sigma window overflow leaves behind
// A device engine gave up in the middle of a contig (a host wait met its deadline, a HIP call failed): whatever the device
// streams' session holds is no longer a position the host knows.  The session ends here -- the next plan starts from the host
// generators, which the caller sets (msim_seed / msim_set_mt_state: mutator.py restores them before it re-plans).
void gpu_plan_abandon(GpuPlan *g) {
    g->s[0].live = g->s[1].live = false;
    g->unverified = false;
    g->ps_valid = false;
    g->est_ok = false;
}

int gpu_plan_force_overflow(Ctx *c, GpuPlan *g) {
    if (!g->d_ps) return MSIM_OK;
    hipLaunchKernelGGL(k_raise_flag, dim3(1), dim3(1), 0, c->stream, g->d_ps, (uint32_t)FLAG_SAMPLE_OVERFLOW);
    MSIM_HIP(c, hipGetLastError());
    g->unverified = true;
    return MSIM_OK;
}

static hipEvent_t next_chain_event(GpuPlan *g) {
    hipEvent_t &e = g->chain_ev[g->chain_i++ % (2 * N_SETS)];
    if (!e) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    return e;
}

// A scratch set may still be read by the emit work of its previous user.  Skip the cross-stream wait when that
// work is known to have finished.
static int wait_if_pending(Ctx *c, bool &pending, hipEvent_t ev) {
    if (!pending) return MSIM_OK;
    if (hipEventQuery(ev) == hipSuccess) { pending = false; return MSIM_OK; }
    (void)hipGetLastError();                               // hipErrorNotReady is not an error
    MSIM_HIP(c, hipStreamWaitEvent(c->stream, ev, 0));
    return MSIM_OK;
}

// Enqueue the chain of one sampled range on the plan stream: where does random.sample() end (exact stream
// cut in the device PlanState) and which values did it draw (the set, as a bitmap in S.bitmap).
struct SampleLaunch { SampleSet *S; uint32_t W, bmw, bnb; };
// raw_other / ps_other: a word buffer and a bookkeeping block of the caller's instead of the CPython stream's (the fast RNG
// mode samples every contig from words of its own, position 0: nothing to generate, nothing chained).
// e_limit: the host's (tighter) bound of where the sample ends -- the next stage's window is laid out from min(pos_hi + W,
// e_limit), so a cut beyond that raises the overflow flag instead of letting the next stage read words nobody made.
static int enqueue_sample_chain(Ctx *c, GpuPlan *g, const msim_range &r, int64_t d, uint64_t pos_hi, bool &grew,
                                SampleLaunch &out, const uint32_t *raw_other = nullptr, PlanState *ps_other = nullptr,
                                uint64_t e_limit = ~0ull) {
    const uint32_t *raw = raw_other ? raw_other : g->s[0].d_raw;
    PlanState *ps = ps_other ? ps_other : g->d_ps;
    int rc;
    const uint32_t k = (uint32_t)r.k;
    const uint64_t n = (uint64_t)((r.stop - (r.k - 1) * d) - r.start);
    const int bits = bit_length64(n);
    const double p_acc = (double)n / (double)(1ull << bits);
    // accepted draws needed ~ -n ln(1 - k/n) (coupon collector); window = that / p_acc, 16-sigma margins
    const double need_acc = -(double)n * std::log1p(-(double)k / (double)n);
    const double target = need_acc + 16.0 * std::sqrt(need_acc) + 4096.0;
    const double wd = target / p_acc + 16.0 * std::sqrt(target) / p_acc + 8192.0;
    if (wd >= 4.0e9) return fail(c, MSIM_ERR_UNSUPPORTED, "sample window beyond 2^32 words");
    const uint32_t W = (uint32_t)wd;
    SampleSet &S = g->sample[g->unit++ % N_SETS];
    S.last_user = 1;
    if ((rc = wait_if_pending(c, S.pending, S.wait_ev))) return rc;            // its last emit may still read it
    const uint32_t nb = (W + ACC_BLOCK - 1) / ACC_BLOCK;
    const size_t bm_words64 = (size_t)((n + 63) / 64);
    const uint32_t bmw = (uint32_t)bm_words64;
    const uint32_t bnb = (bmw + BM_THREADS - 1) / BM_THREADS;
    if ((rc = grow(c, (void **)&S.cnt, &S.cnt_cap, (size_t)(nb + 2) * sizeof(uint32_t), &grew))) return rc;
    if ((rc = grow(c, (void **)&S.cnt2, &S.cnt2_cap, (size_t)(bnb + EMIT_SUPER + bnb / EMIT_SUPER + 4) * sizeof(uint32_t), &grew))) return rc;   // (+ whole super-blocks and their totals: k_snp_emit_count_b)
    if ((rc = grow(c, (void **)&S.acc, &S.acc_cap, (size_t)W * sizeof(uint32_t), &grew))) return rc;
    if ((rc = grow(c, (void **)&S.bitmap, &S.bm_cap, bm_words64 * 8, &grew))) return rc;
    if (!S.emit_done) MSIM_HIP(c, hipEventCreateWithFlags(&S.emit_done, hipEventDisableTiming));
    if (!raw_other) {
        if ((rc = ensure_words(c, g, 0, pos_hi + W + 1, false))) return rc;
        raw = g->s[0].d_raw;                               // (may have been reallocated)
    }
    const uint64_t pos_limit = (raw_other || e_limit == ~0ull) ? ~0ull : std::min<uint64_t>(pos_hi + W, e_limit);
    // ---- chain (plan stream): where does this sample end?
    const uint32_t n_bins = (uint32_t)((n + BIN_VALUES - 1) >> BIN_SHIFT);
    const bool binned = n_bins <= (uint32_t)MAX_BINS;
    uint32_t bin_cap = 0;
    if (binned) {

        // expected k/n_bins per bin (the last bin is partial), 16-sigma + slack capacity
        // sub-list s of a bin is filled by the scatter workgroups with index == s (mod 16).  Only the
        // workgroups up to the one holding the k-th accepted draw contribute (the window has slack
        // behind it), each at most 8192 * p_acc draws, of which the bin's share is min(1, 2^20 / n).
        const uint32_t nbk = (uint32_t)((double)k / p_acc / SPL_BLOCK) + 3;
        const double mean = (double)((nbk + BIN_SUBS - 1) / BIN_SUBS) * SPL_BLOCK * p_acc *
                            std::min(1.0, (double)BIN_VALUES / (double)n);
        bin_cap = (uint32_t)std::min<double>((double)k + 16.0, 1.25 * mean + 16.0 * std::sqrt(mean) + 512.0);
        if ((rc = grow(c, (void **)&S.bins, &S.bins_cap, (size_t)n_bins * BIN_SUBS * bin_cap * sizeof(uint32_t), &grew))) return rc;
        if ((rc = grow(c, (void **)&S.cursors, &S.cursors_cap, (size_t)MAX_BINS * BIN_SUBS * sizeof(uint32_t), &grew))) return rc;
        if ((rc = grow(c, (void **)&S.bitmap, &S.bm_cap, (size_t)n_bins * BIN_WORDS * 4, &grew))) return rc;
    }
    hipLaunchKernelGGL(k_accept_count, dim3(nb), dim3(ACC_THREADS), 0, c->stream, raw, ps, W,
                       (uint32_t)(32 - bits), (uint32_t)n, S.cnt, ps, binned ? S.cursors : nullptr,
                       binned ? n_bins * BIN_SUBS : 0u);
    // the counts' prefix sums: made by their two consumers themselves where the tail kernel can hold them in LDS (every real
    // contig) -- a launch less on the chain; else by a scan of their own
    static const bool no_fold_scan = getenv("MSIM_NO_SCAN_FOLD") != nullptr;
    const uint32_t raw_counts = (binned && nb < (uint32_t)TAIL_LDS_OFFS && !no_fold_scan) ? 1u : 0u;
    if (!raw_counts) hipLaunchKernelGGL(k_scan_u32_w4, dim3(1), dim3(256), 0, c->stream, S.cnt, nb);   // (four waves: see the kernel)
    if (binned) {
        hipLaunchKernelGGL(k_bin_scatter, dim3((W + SPL_BLOCK - 1) / SPL_BLOCK), dim3(ACC_THREADS), 0, c->stream, raw, ps, W,
                           (uint32_t)(32 - bits), (uint32_t)n, k, S.cnt, n_bins, bin_cap, S.cursors, S.bins, S.acc,
                           ps, raw_counts);
        hipLaunchKernelGGL(k_bin_dedupe, dim3(n_bins), dim3(512), 0, c->stream, S.bins, S.cursors, bin_cap, S.bitmap,
                           ps);
        hipLaunchKernelGGL(k_sample_tail, dim3(1), dim3(1024), 0, c->stream, raw, S.acc, k, S.cnt, nb, W,
                           (uint32_t)(32 - bits), (uint32_t)n, k, S.bitmap, ps, raw_counts, (const PlanState *)nullptr,
                           (const SpecHdr *)nullptr, (unsigned long long)pos_limit);
    } else {
        MSIM_HIP(c, hipMemsetAsync(S.bitmap, 0, bm_words64 * 8, c->stream));
        hipLaunchKernelGGL(k_accept_scatter, dim3(nb), dim3(ACC_THREADS), 0, c->stream, raw, ps, W,
                           (uint32_t)(32 - bits), (uint32_t)n, S.cnt, S.acc);
        hipLaunchKernelGGL(k_bitmap_insert, dim3((k + 1023) / 1024), dim3(256), 0, c->stream, S.acc, k, S.bitmap, ps);
        hipLaunchKernelGGL(k_sample_tail, dim3(1), dim3(1024), 0, c->stream, raw, S.acc, 0u, S.cnt, nb, W,
                           (uint32_t)(32 - bits), (uint32_t)n, k, S.bitmap, ps, 0u, (const PlanState *)nullptr,
                           (const SpecHdr *)nullptr, (unsigned long long)pos_limit);
    }
    MSIM_HIP(c, hipGetLastError());
    out.S = &S; out.W = W; out.bmw = bmw; out.bnb = bnb;
    return MSIM_OK;
}

// ---- anchored windows: the same sample with its heavy part OFF the chain (plan_kernels.h: k_ahead_fringe has the argument).
// The host knows the sample's start to within [lo, H]; count, scatter and de-dup of the first k_core accepted draws from H and
// the compaction of the accepted draws of [lo, H) run on the prep stream as soon as the words exist; the chain keeps the
// fringe pass (needs the exact start), the tail rounds and the cut.  e_limit: the cut may not lie beyond it (what the caller
// will have made sure exists for the stage that follows).
static int prep_wait_words(Ctx *c, GpuPlan *g, uint64_t upto) {
    GpuStream &s = g->s[0];
    if (upto <= MT_N) return MSIM_OK;
    const uint32_t need = (uint32_t)((upto - MT_N + MT_CHUNK_WORDS - 1) / MT_CHUNK_WORDS);
    uint32_t &waited = g->prep_waited[g->prep_i];
    if (need <= waited) return MSIM_OK;
    for (size_t b = 0; b < s.ready_hi.size(); b++) {
        if (s.ready_hi[b] >= need) {
            MSIM_HIP(c, hipStreamWaitEvent(g->prep_stream, s.words_ev[b] ? s.words_ev[b] : s.ready_ev[b], 0));
            waited = s.ready_hi[b];
            break;
        }
    }
    return MSIM_OK;
}

// grouped: the contig's emission goes out with a group (gpu_emit_flush) -- the group's event, behind its rewrite launch, then also
// covers the chain kernels enqueued here (the emit stream waited for the plan stream's position at the flush), so no event of the
// set's own is recorded behind them: one packet less per contig on the chain's queue
static int enqueue_sample_ahead(Ctx *c, GpuPlan *g, const msim_range &r, int64_t d, uint64_t lo, uint64_t H, uint64_t e_limit,
                                bool &grew, SampleLaunch &out, bool &took, bool grouped, uint64_t expect, uint32_t soft_half) {
    int rc;
    const uint32_t K = (uint32_t)r.k;
    const uint64_t n = (uint64_t)((r.stop - (r.k - 1) * d) - r.start);
    const int bits = bit_length64(n);
    const double p_acc = (double)n / (double)(1ull << bits);
    const double need_acc = -(double)n * std::log1p(-(double)K / (double)n);
    const double target = need_acc + 16.0 * std::sqrt(need_acc) + 4096.0;
    const double wd = target / p_acc + 16.0 * std::sqrt(target) / p_acc + 8192.0;
    if (wd >= 4.0e9) return fail(c, MSIM_ERR_UNSUPPORTED, "sample window beyond 2^32 words");
    const uint32_t W = (uint32_t)wd;                       // (sized for all K draws from H: the core needs fewer)
    const uint32_t F = (uint32_t)(H - lo);
    took = (W + ACC_BLOCK - 1) / ACC_BLOCK < (uint32_t)TAIL_LDS_OFFS && (F + ACC_BLOCK - 1) / ACC_BLOCK <= (uint32_t)AHEAD_MAX_HEAD_BLOCKS;
    if (!took) return MSIM_OK;                             // (counts beyond what the chain's kernels hold in LDS: the caller's other path)
    const uint32_t a_max = F ? (uint32_t)std::min<double>((double)F, (double)F * p_acc + 16.0 * std::sqrt((double)F * p_acc * (1.0 - p_acc)) + 64.0) : 0u;
    const uint32_t k_core = K - a_max;
    g->prep_i = (g->prep_i + 1) % g->n_prep;
    if (!g->prep_streams[g->prep_i]) {
        int plo = 0, phi = 0;
        MSIM_HIP(c, hipDeviceGetStreamPriorityRange(&plo, &phi));
        const int prio = g->prep_i >= g->n_prep - g->n_prep_low ? plo : g->prep_prio > 0 ? plo : g->prep_prio < 0 ? phi : 0;
        MSIM_HIP(c, hipStreamCreateWithPriority(&g->prep_streams[g->prep_i], hipStreamNonBlocking, prio));
        if (g->seed_ev[0]) MSIM_HIP(c, hipStreamWaitEvent(g->prep_streams[g->prep_i], g->seed_ev[0], 0));   // (the session's first 624 words are a copy)
    }
    g->prep_stream = g->prep_streams[g->prep_i];
    hipStream_t ps_ = g->prep_stream;
    SampleSet &S = g->sample[g->unit++ % N_SETS];
    const uint32_t nb = (W + ACC_BLOCK - 1) / ACC_BLOCK;
    const uint32_t nbh = (F + ACC_BLOCK - 1) / ACC_BLOCK;
    const size_t bm_words64 = (size_t)((n + 63) / 64);
    const uint32_t bmw = (uint32_t)bm_words64;
    const uint32_t bnb = (bmw + BM_THREADS - 1) / BM_THREADS;
    const uint32_t n_bins = (uint32_t)((n + BIN_VALUES - 1) >> BIN_SHIFT);
    const uint32_t nbk = (uint32_t)((double)K / p_acc / SPL_BLOCK) + 3;
    const double mean = (double)((nbk + BIN_SUBS - 1) / BIN_SUBS) * SPL_BLOCK * p_acc * std::min(1.0, (double)BIN_VALUES / (double)n);
    const uint32_t bin_cap = (uint32_t)std::min<double>((double)K + 16.0, 1.25 * mean + 16.0 * std::sqrt(mean) + 512.0);
    if ((rc = grow(c, (void **)&S.cnt, &S.cnt_cap, (size_t)(nb + 2) * sizeof(uint32_t), &grew))) return rc;
    if ((rc = grow(c, (void **)&S.cnt2, &S.cnt2_cap, (size_t)(bnb + EMIT_SUPER + bnb / EMIT_SUPER + 4) * sizeof(uint32_t), &grew))) return rc;   // (+ whole super-blocks and their totals: k_snp_emit_count_b)
    if ((rc = grow(c, (void **)&S.acc, &S.acc_cap, (size_t)W * sizeof(uint32_t), &grew))) return rc;
    if ((rc = grow(c, (void **)&S.bins, &S.bins_cap, (size_t)n_bins * BIN_SUBS * bin_cap * sizeof(uint32_t), &grew))) return rc;
    if ((rc = grow(c, (void **)&S.cursors, &S.cursors_cap, (size_t)MAX_BINS * BIN_SUBS * sizeof(uint32_t), &grew))) return rc;
    if ((rc = grow(c, (void **)&S.bitmap, &S.bm_cap, std::max(bm_words64 * 8, (size_t)n_bins * BIN_WORDS * 4), &grew))) return rc;
    if ((rc = grow(c, (void **)&S.hcnt, &S.hcnt_cap, (size_t)(nbh + 2) * sizeof(uint32_t), &grew))) return rc;
    if ((rc = grow(c, (void **)&S.hacc, &S.hacc_cap, ((size_t)nbh * ACC_BLOCK + 64) * sizeof(uint32_t), &grew))) return rc;
    if (!S.ahead) MSIM_HIP(c, hipMalloc(&S.ahead, 2 * sizeof(PlanState) + sizeof(SpecHdr)));
    if (!S.emit_done) MSIM_HIP(c, hipEventCreateWithFlags(&S.emit_done, hipEventDisableTiming));
    if (!S.prep_done) MSIM_HIP(c, hipEventCreateWithFlags(&S.prep_done, hipEventDisableTiming));
    PlanState *ps_core = reinterpret_cast<PlanState *>(S.ahead), *ps_head = ps_core + 1;
    SpecHdr *hdr = reinterpret_cast<SpecHdr *>(ps_head + 1);
    if ((rc = ensure_words(c, g, 0, H + W + 1, false))) return rc;
    const uint32_t *raw = g->s[0].d_raw;
    if ((rc = prep_wait_words(c, g, H + W + 1))) return rc;
    if (S.pending) {                                       // the set's last emission may still read it
        if (hipEventQuery(S.wait_ev) == hipSuccess) S.pending = false;
        else { (void)hipGetLastError(); MSIM_HIP(c, hipStreamWaitEvent(ps_, S.wait_ev, 0)); }
    }
    // ... and so may the chain of the contig that had the set N_SETS contigs ago (the host may be that far ahead of the device):
    // an anchored window left an event behind its last chain kernel; behind a sample on the chain, whatever the plan stream holds
    if (!S.chain_done) MSIM_HIP(c, hipEventCreateWithFlags(&S.chain_done, hipEventDisableTiming));
    if (S.last_user == 1) MSIM_HIP(c, hipEventRecord(S.chain_done, c->stream));
    if ((S.last_user == 1 || S.last_user == 2) && hipEventQuery(S.chain_done) != hipSuccess) {
        (void)hipGetLastError();
        MSIM_HIP(c, hipStreamWaitEvent(ps_, S.chain_done, 0));
    }
    S.last_user = grouped ? 3 : 2;                         // (3: its group's event -- S.wait_ev, checked above -- stands for chain_done)
    // ---- off the chain
    const uint32_t shift = (uint32_t)(32 - bits);
    hipLaunchKernelGGL(k_ahead_count, dim3(nb + nbh), dim3(ACC_THREADS), 0, ps_, raw, ps_core, ps_head, hdr, (unsigned long long)H,
                       (unsigned long long)lo, W, F, nb, shift, (uint32_t)n, S.cnt, S.cursors, n_bins * BIN_SUBS, S.hcnt, S.hacc);
    hipLaunchKernelGGL(k_bin_scatter, dim3((W + SPL_BLOCK - 1) / SPL_BLOCK), dim3(ACC_THREADS), 0, ps_, raw, ps_core, W, shift,
                       (uint32_t)n, k_core, S.cnt, n_bins, bin_cap, S.cursors, S.bins, S.acc, ps_core, 1u);
    hipLaunchKernelGGL(k_bin_dedupe, dim3(n_bins), dim3(512), 0, ps_, S.bins, S.cursors, bin_cap, S.bitmap, ps_core);
    MSIM_HIP(c, hipGetLastError());
    MSIM_HIP(c, hipEventRecord(S.prep_done, ps_));
    // ---- on the chain: fringe (exact start), tail rounds, cut
    MSIM_HIP(c, hipStreamWaitEvent(c->stream, S.prep_done, 0));
    const double dups_est = (double)K * (double)K / (2.0 * (double)n);
    const uint32_t items_est = a_max + (uint32_t)(dups_est + 16.0 * std::sqrt(dups_est) + 64.0);
    const uint32_t G = std::max(1u, std::min(256u, (items_est + ACC_THREADS * FRINGE_ITEMS - 1) / (ACC_THREADS * FRINGE_ITEMS)));
    hipLaunchKernelGGL(k_ahead_fringe, dim3(G), dim3(ACC_THREADS), 0, c->stream, raw, g->d_ps, ps_core, ps_head, hdr,
                       (unsigned long long)lo, F, S.hcnt, nbh, S.hacc, S.acc, K, k_core, shift, (uint32_t)n, S.bitmap,
                       (unsigned long long)expect, soft_half);
    hipLaunchKernelGGL(k_sample_tail, dim3(1), dim3(1024), 0, c->stream, raw, S.acc, k_core, S.cnt, nb, W, shift, (uint32_t)n,
                       k_core, S.bitmap, g->d_ps, 1u, ps_core, hdr, (unsigned long long)e_limit);
    MSIM_HIP(c, hipGetLastError());
    if (!grouped) MSIM_HIP(c, hipEventRecord(S.chain_done, c->stream));
    out.S = &S; out.W = W; out.bmw = bmw; out.bnb = bnb;
    c->t.snp_samples_ahead++;
    static const bool log_ahead = getenv("MSIM_DBG_AHEAD_LOG") != nullptr;
    if (log_ahead) fprintf(stderr, "msim: sample ahead of the chain: K %u, start in [%llu, %llu], core %u draws\n", K, (unsigned long long)lo, (unsigned long long)H, k_core);
    return MSIM_OK;
}

// SNP draws of a contig's K (kept) SNPs in position order, starting at the device position (ps->snp_base): the chain
// part (k_snp_scan_cut_abs: exact end of the draws) on the plan stream, the aux bytes (k_snp_emit_abs) on the emit
// stream.  pos_hi: upper bound of the start position on entry, of the end position on return.
// aux8_out (SNP sampler, one drawing range): the outcomes go to a compact byte array by rank instead of into the records; the
// caller launches the bitmap expansion behind this stage and hands it that array (*aux8_out).
// defer (with aux8_out): the emit pass is not launched here at all -- the caller queues it for the contig's emission group.
static int enqueue_snp_stage(Ctx *c, GpuPlan *g, Contig &ct, uint64_t K, const uint32_t *sn_index, uint64_t &pos_hi, bool &grew,
                             uint8_t **aux8_out = nullptr, SnpDefer *defer = nullptr, uint64_t e_limit = ~0ull) {
    const msim_params &P = c->params;
    GpuStream &py = g->s[0];
    int rc;
    const double p_tv = 1.0 - std::min(1.0, (double)P.ti_lim / 9007199254740992.0);
    const double w2 = (double)K * (2.0 + 2.0 * p_tv) + 16.0 * std::sqrt(4.0 * (double)K) + 16384.0;
    if (w2 >= 4.0e9) return fail(c, MSIM_ERR_UNSUPPORTED, "SNP draw window beyond 2^32 words");
    const uint32_t W2 = (uint32_t)w2;
    const uint32_t nb2 = W2 / SNP_BLOCK2 + 2;              // window blocks incl. the partial first one
    SnpSet &T = g->snp[g->snp_unit++ % N_SETS];
    if ((rc = wait_if_pending(c, T.pending, T.wait_ev))) return rc;
    if ((rc = grow(c, (void **)&T.maps, &T.cap, (size_t)(nb2 + 1) * sizeof(SnpMap), &grew))) return rc;
    if (aux8_out) {
        if ((rc = grow(c, (void **)&T.aux8, &T.aux_cap, (size_t)K + 64, &grew))) return rc;
        *aux8_out = T.aux8;
    }
    if (!T.base) MSIM_HIP(c, hipMalloc(&T.base, 64));
    if (!T.emit_done) MSIM_HIP(c, hipEventCreateWithFlags(&T.emit_done, hipEventDisableTiming));
    // words AND absolute maps up to the end of the last window block
    if ((rc = ensure_words(c, g, 0, ((pos_hi + W2) / SNP_BLOCK2 + 2) * SNP_BLOCK2))) return rc;
    if (py.maps_ti_lim != P.ti_lim) {                      // titv changed inside a session: the maps are stale
        if (!py.ready_ev.empty()) MSIM_HIP(c, hipStreamWaitEvent(c->stream, py.ready_ev.back(), 0));
        if (py.mapped_blocks)
            hipLaunchKernelGGL(k_snp_maps_abs, dim3(py.mapped_blocks), dim3(SNP_THREADS), 0, c->stream, py.d_raw,
                               (uint32_t)(MT_N + (uint64_t)py.n_chunks * MT_CHUNK_WORDS), (unsigned long long)P.ti_lim, 0u, py.d_maps, py.d_lanes);
        py.maps_ti_lim = P.ti_lim;
    }
    hipLaunchKernelGGL(k_snp_scan_cut_abs, dim3(1), dim3(SNP_THREADS), 0, c->stream, py.d_lanes, g->d_ps, W2,
                       py.d_maps, T.maps, nb2, (uint32_t)K, T.base,
                       (unsigned long long)(e_limit == ~0ull ? ~0ull : std::min<uint64_t>(pos_hi + W2, e_limit)));
    MSIM_HIP(c, hipGetLastError());
    if (defer) { defer->T = &T; defer->W2 = W2; defer->nb2 = nb2; }
    if (!c->chain_only && !defer) {                       // (chain only: where the draws END is all that is wanted)
        hipEvent_t ce = next_chain_event(g);
        MSIM_HIP(c, hipEventRecord(ce, c->stream));
        MSIM_HIP(c, hipStreamWaitEvent(c->emit_stream, ce, 0));
        hipLaunchKernelGGL(k_snp_emit_abs, dim3(nb2), dim3(SNP_THREADS), 0, c->emit_stream, py.d_lanes, T.base, W2,
                           T.maps, nb2, ct.d_recs, (uint32_t)K, sn_index, aux8_out ? T.aux8 : (uint8_t *)nullptr);
        MSIM_HIP(c, hipGetLastError());
        if (!aux8_out) {                                  // (else: the caller records it behind the expansion that reads aux8)
            MSIM_HIP(c, hipEventRecord(T.emit_done, c->emit_stream));
            T.wait_ev = T.emit_done;
            T.pending = true;
        }
    }
    pos_hi += W2;
    return MSIM_OK;
}

// The emission group: count, scan, SNP outcomes and expansion of every queued contig in one launch per stage on the emission
// stream, behind the chain of the last one; then the APPLYs msim_apply_contig marked (one tile-index launch for all of them,
// apply.hip: apply_batch_device), the rewrite kernels back to back.
int gpu_emit_flush(Ctx *c) {
    GpuPlan *g = c->gpu;
    if (!g || g->emit_items.empty()) return MSIM_OK;
    std::vector<EmitItem> items;
    items.swap(g->emit_items);
    EmitJobs J;
    memset(&J, 0, sizeof J);
    J.n = (uint32_t)items.size();
    J.d = g->emit_d;
    uint32_t blk = 0, eblk = 0;
    for (uint32_t i = 0; i < J.n; i++) {
        const EmitItem &it = items[i];
        EmitJob &T = J.j[i];
        T.bm = reinterpret_cast<const uint64_t *>(it.S->bitmap); T.cnt2 = it.S->cnt2; T.recs = it.recs; T.aux8 = it.T->aux8;
        T.base = it.T->base; T.win_maps = it.T->maps;
        T.bmw = it.bmw; T.bnb = it.bnb; T.start = it.start; T.K = it.K; T.W2 = it.W2; T.nb2 = it.nb2; T.blk0 = blk; T.eblk0 = eblk;
        blk += it.bnb; eblk += it.nb2;
    }
    J.total_blk = blk; J.total_eblk = eblk;
    hipStream_t es = c->emit_stream;
    // The train in three launches (MSIM_EMIT_TRAIN=6: the six of rounds 4-5: count, scan, outcomes, expansion, tile index,
    // rewrite): outcomes + popcounts in one grid; the expansion makes its rank base from two levels of counts and writes the
    // APPLY tile index of the contigs that are applied with their group; the rewrite.
    // Which one: measured on one box, c2 3 Gb, ms per step (profiles/r06_emission_train.txt).  A rank that owns everything and
    // samples on the chain is bound by that chain, and the denser train slows the chain's kernels more than the six small
    // launches with their gaps did: 3.88 (six) against 4.07-4.14 (three).  Where the samples' heavy kernels run off the chain
    // (anchored windows: ranks of a sharded step, MSIM_AHEAD=2) the train is the bound: a rank owning 12 of 24 contigs 3.35-3.47
    // (six) -> 3.18-3.21 (three); owning 3: 2.38-2.40 / 2.34-2.41.
    const bool ahead_in_use = g->ahead == 2 || (g->ahead == 1 && g->sharded_rank);
    const bool train3 = g->emit_train == 3 || (g->emit_train == 0 && ahead_in_use);
    std::vector<int> applies;
    if (train3) {
        uint32_t cblk = 0;
        for (uint32_t i = 0; i < J.n; i++) {
            const EmitItem &it = items[i];
            EmitJob &T = J.j[i];
            T.cblk0 = cblk;
            cblk += (it.bnb + EMIT_SUPER - 1) / EMIT_SUPER;
            T.first = nullptr; T.err = nullptr; T.n_tiles = 0;
            if (it.apply && (size_t)it.contig < c->contigs.size()) {
                Contig &ct = c->contigs[(size_t)it.contig];
                ct.apply_stream = es;                      // (apply_batch_device takes contigs with a stream of their own)
                uint32_t shift = 0;
                const int rc = apply_prepare_tile_index(c, ct, es, &T.first, &T.n_tiles, &shift, &T.err);
                if (rc) return rc;
                J.tile_shift = shift;
                if (!T.n_tiles) T.first = nullptr;
                applies.push_back(it.contig);
            }
        }
        J.total_cblk = cblk;
    }
    hipEvent_t ce = next_chain_event(g);
    MSIM_HIP(c, hipEventRecord(ce, c->stream));
    MSIM_HIP(c, hipStreamWaitEvent(es, ce, 0));
    if (train3) {
        hipLaunchKernelGGL(k_snp_emit_count_b, dim3(eblk + J.total_cblk), dim3(SNP_THREADS), 0, es, g->s[0].d_lanes, J);
        hipLaunchKernelGGL(k_bitmap_expand_tiles_b, dim3(blk), dim3(BM_THREADS), 0, es, J);
    } else {
        hipLaunchKernelGGL(k_bitmap_count_b, dim3(blk), dim3(BM_THREADS), 0, es, J);
        hipLaunchKernelGGL(k_scan_u32_b, dim3(J.n), dim3(256), 0, es, J);
        hipLaunchKernelGGL(k_snp_emit_abs_b, dim3(eblk), dim3(SNP_THREADS), 0, es, g->s[0].d_lanes, J);
        hipLaunchKernelGGL(k_bitmap_expand_b, dim3(blk), dim3(BM_THREADS), 0, es, J);
    }
    MSIM_HIP(c, hipGetLastError());
    for (const EmitItem &it : items) {
        if (!train3 && it.apply && (size_t)it.contig < c->contigs.size()) {
            c->contigs[(size_t)it.contig].apply_stream = es;      // (apply_batch_device takes contigs with a stream of their own)
            applies.push_back(it.contig);
        }
    }
    const int arc = applies.empty() ? MSIM_OK : apply_batch_device(c, applies, true);
    // the group's event: behind the rewrite (a scratch set meets its next user a step later -- what it waits for may as well
    // include the APPLY)
    hipEvent_t &ge = g->grp_ev[g->grp_i++ % (2 * N_SETS)];
    if (!ge) MSIM_HIP(c, hipEventCreateWithFlags(&ge, hipEventDisableTiming));
    MSIM_HIP(c, hipEventRecord(ge, es));
    for (const EmitItem &it : items) {
        it.S->wait_ev = ge; it.S->pending = true;
        it.T->wait_ev = ge; it.T->pending = true;
    }
    return arc;
}

bool gpu_emit_pending(Ctx *c, int contig, bool mark_apply) {
    GpuPlan *g = c->gpu;
    if (!g) return false;
    for (EmitItem &it : g->emit_items)
        if (it.contig == contig) { if (mark_apply) it.apply = true; return true; }
    return false;
}

StreamMoments sample_words_moments(uint64_t n, uint64_t k) {
    const double n_d = (double)n, k_d = (double)k;
    const double p_acc = n_d / (double)(1ull << bit_length64(n));
    const double l1p = std::log1p(-k_d / n_d);
    const double EA = -n_d * l1p, VA = std::max(0.0, n_d * (k_d / (n_d - k_d) + l1p));
    return StreamMoments{EA / p_acc, EA * (1.0 - p_acc) / (p_acc * p_acc) + VA / (p_acc * p_acc)};
}
StreamMoments snp_words_moments(uint64_t K, unsigned long long ti_lim) {
    const double p_tv = 1.0 - std::min(1.0, (double)ti_lim / 9007199254740992.0);
    return StreamMoments{(double)K * (2.0 + 2.0 * p_tv), (double)K * (2.0 * p_tv + 4.0 * p_tv * (1.0 - p_tv))};
}

// Asynchronous: enqueues the contig's chain (stream positions) on the plan stream and its emit work
// (records) on the emit stream; nothing is waited for.  Flags and the exact position are collected
// by gpu_plan_finish at the next synchronising call.
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
    bool grew = false;
    if (!c->chain_only) {   // the record table may still be read by an earlier apply of this contig
        const size_t want = std::max<uint64_t>(K, 1) * sizeof(msim_record);
        if (ct.cap_recs < want || ct.cap_pool < 2 * PAD) {
            MSIM_HIP(c, wait_stream(c->stream));
            MSIM_HIP(c, wait_stream(c->emit_stream));
        }
        if ((rc = dev_reserve(c, (void **)&ct.d_recs, &ct.cap_recs, want))) return rc;
        if ((rc = dev_reserve(c, (void **)&ct.d_pool, &ct.cap_pool, 2 * PAD))) return rc;
    }
    ct.n_rec = K;
    ct.pool_len = 0;
    ct.plan_empty = K == 0;
    ct.all_snp = true;                                    // every record is an SNP: output offset == position

    GpuStream &py = g->s[0];
    if (!g->t0) { MSIM_HIP(c, hipEventCreate(&g->t0)); MSIM_HIP(c, hipEventCreate(&g->t1)); }
    if (!g->unverified) MSIM_HIP(c, hipEventRecord(g->t0, c->stream));
    if (!g->ps_valid) {
        hipLaunchKernelGGL(k_state_init, dim3(1), dim3(1), 0, c->stream, g->d_ps, (unsigned long long)py.pos);
        g->ps_valid = true;
    }
    uint64_t pos_hi = py.pos;                             // upper bound of the device position
    uint64_t rec_base = 0;
    if (c->chain_only) { g->pass_chain_only++; g->sharded_rank = true; }
    if (!g->unverified) {                                 // py.pos is exact: the estimate of the position starts here
        g->est_ok = true;
        g->est_e = (double)py.pos; g->est_v = 0.0; g->est_lo = py.pos;
    }
    // mean + 16 sigma of the position (est_*), never beyond the sum of the windows
    auto est_hi = [&](uint64_t bound) -> uint64_t {
        if (!g->est_ok) return bound;
        const double m = g->est_v > 0.0 ? 16.0 * std::sqrt(1.1 * g->est_v) + 256.0 : 0.0;
        return std::min<uint64_t>(bound, (uint64_t)std::ceil(g->est_e + m));
    };
    // One drawing range (ARGS mode): the SNP outcomes are folded into the records by the expansion, which then runs behind
    // the SNP stage.  Several ranges keep the order expansion -> outcomes patched in (their bitmaps rotate through N_SETS
    // scratch sets and cannot all wait for the contig's one SNP stage).
    int n_draw = 0;
    for (int i = 0; i < n_ranges; i++) n_draw += ranges[i].k != 0;
    static const bool no_fold = getenv("MSIM_NO_AUX_FOLD") != nullptr;
    const bool fold_aux = n_draw == 1 && !c->chain_only && !no_fold;
    // ... and such contigs go through the emission stages in groups of EMIT_G (gpu_emit_flush), one launch per stage
    static const bool no_group = getenv("MSIM_NO_EMIT_GROUP") != nullptr;
    const bool grouped = fold_aux && !no_group;
    // A full group goes out when the NEXT contig is planned, in front of its chain: by then msim_apply_contig has marked the
    // group's last contig too, so all of its APPLYs share the group's tile-index launch.
    // (pairs.  Measured with the calling thread on the GPU's NUMA node, where runs repeat within 0.05 ms: pairs 4.10-4.21 ms per
    //  c2 step with the rewrite launches at 0.72-0.74 of the HBM peak, threes 4.22-4.31 at 0.75-0.77, fours 4.29-4.32 at 0.78,
    //  sixes 4.36-4.38 at 0.79 -- a larger group is a better rewrite launch and a longer stretch in which the chain's kernels
    //  crawl beside it, plus more left to do behind the last chain; groups cut by their bases instead of their count, 250-400 Mb
    //  with at most 4-6 contigs, were no better than threes; one per group = the ungrouped 4.6-5.1)
    // (round 6: with the samples' heavy kernels off the chain -- anchored windows, now every context's default -- the train is the
    //  bound and larger groups pay: fours 3.67-3.70 ms, threes 3.69-3.82, pairs 3.76-3.81 on a box where pairs with every sample
    //  on the chain take 3.90-4.00; profiles/r06_emission_train.txt)
    const bool ahead_role = g->ahead == 2 || (g->ahead == 1 && g->sharded_rank);
    const int group = g->emit_group ? g->emit_group : (ahead_role ? 4 : 2);
    if (!g->emit_items.empty() && (!grouped || g->emit_d != (uint32_t)d || g->emit_items.size() >= (size_t)group))
        if ((rc = gpu_emit_flush(c))) return rc;
    struct { SampleSet *S; uint32_t bmw, bnb, start; } late = {nullptr, 0, 0, 0};
    for (int i = 0; i < n_ranges; i++) {
        const msim_range &r = ranges[i];
        if (r.k == 0) continue;
        const uint32_t k = (uint32_t)r.k;
        SampleLaunch sl;
        // moments of the words this sample consumes: A accepted draws until k distinct values (duplicates: a coupon collector's
        // first k), each after a geometric number of rejected words (_randbelow)
        const double n_d = (double)((r.stop - (r.k - 1) * d) - r.start);
        const StreamMoments ms = sample_words_moments((uint64_t)n_d, (uint64_t)k);
        const double e_samp = ms.e, v_samp = ms.v;
        bool ahead = false;
        uint64_t lo = 0, H = 0, expect = 0;
        uint32_t soft_half = 0;
        if ((g->ahead == 2 || (g->ahead == 1 && g->sharded_rank)) && (grouped || (c->chain_only && n_draw == 1)) && g->est_ok && (uint64_t)n_d <= ((uint64_t)MAX_BINS << BIN_SHIFT) && 4.0 * (double)k <= n_d) {
            // the interval costs what it holds (its accepted draws go through the fringe pass on the chain): g->ahead_sigma
            // (8) standard deviations instead of the windows' 16 -- a start outside it is reported like an overflowed window
            const double m = g->est_v > 0.0 ? g->ahead_sigma * std::sqrt(1.1 * g->est_v) + 256.0 : 0.0;
            expect = (uint64_t)std::max(0.0, g->est_e);
            soft_half = (uint32_t)std::min(4.0e9, m);
            lo = std::max<uint64_t>(g->est_lo, (uint64_t)std::max(0.0, std::floor(g->est_e - m)));
            H = std::max<uint64_t>(lo, std::min<uint64_t>(pos_hi, (uint64_t)std::ceil(g->est_e + m)));
            // (a sample smaller than the uncertainty of its start stays on the chain; so does one whose start is known exactly --
            //  behind a synchronisation the chain is empty and nothing is gained by making it wait for the prep stream)
            static const bool ahead_first = getenv("MSIM_AHEAD_FIRST") != nullptr;
            ahead = 2 * (H - lo) <= (uint64_t)k && (g->est_v > 0.0 || ahead_first);
        }
        if (g->est_ok) { g->est_e += e_samp; g->est_v += v_samp; g->est_lo += k; }
        if (ahead) {
            // the sample ends at or in front of e_lim (and inside its window): what the SNP stage's windows are laid out from
            const uint64_t e_lim = est_hi(~0ull);
            if ((rc = enqueue_sample_ahead(c, g, r, d, lo, H, e_lim, grew, sl, ahead, grouped, expect, soft_half))) return rc;
            if (ahead) pos_hi = std::min<uint64_t>(H + sl.W, e_lim) - sl.W;      // (+ W below)
        }
        if (!ahead && (rc = enqueue_sample_chain(c, g, r, d, pos_hi, grew, sl, nullptr, nullptr, est_hi(~0ull)))) return rc;
        SampleSet &S = *sl.S;
        const uint32_t W = sl.W, bmw = sl.bmw, bnb = sl.bnb;
        if (grouped) {
            late = {&S, bmw, bnb, (uint32_t)r.start};
        } else if (!c->chain_only) {
            hipEvent_t ce = next_chain_event(g);
            MSIM_HIP(c, hipEventRecord(ce, c->stream));
            // ---- emit (emit stream): the bitmap is the sorted sample -> records
            MSIM_HIP(c, hipStreamWaitEvent(c->emit_stream, ce, 0));
            hipLaunchKernelGGL(k_bitmap_count, dim3(bnb), dim3(BM_THREADS), 0, c->emit_stream,
                               reinterpret_cast<const uint64_t *>(S.bitmap), bmw, S.cnt2);
            hipLaunchKernelGGL(k_scan_u32_w4, dim3(1), dim3(256), 0, c->emit_stream, S.cnt2, bnb);
            if (fold_aux) {                               // the expansion follows the SNP stage (below): it writes complete records
                late = {&S, bmw, bnb, (uint32_t)r.start};
            } else {
                hipLaunchKernelGGL(k_bitmap_expand, dim3(bnb), dim3(BM_THREADS), 0, c->emit_stream,
                                   reinterpret_cast<const uint64_t *>(S.bitmap), bmw, S.cnt2, (uint32_t)r.start, (uint32_t)d,
                                   ct.d_recs + rec_base, (const uint8_t *)nullptr);
                MSIM_HIP(c, hipGetLastError());
                MSIM_HIP(c, hipEventRecord(S.emit_done, c->emit_stream));
                S.wait_ev = S.emit_done;
                S.pending = true;
            }
        }
        pos_hi += W;
        if (!ahead) pos_hi = est_hi(pos_hi);
        rec_base += k;
    }
    if (K) {                                          // SNP draws in position order (chain: scan + cut; aux off the chain)
        uint8_t *aux8 = nullptr;
        SnpDefer df{nullptr, 0, 0};
        if (g->est_ok) {                              // a uniform() = 2 words, a transversion's randbelow(2) = a geometric(1/2) loop
            const StreamMoments mv = snp_words_moments(K, P.ti_lim);
            g->est_e += mv.e;
            g->est_v += mv.v;
            g->est_lo += 2 * K;
        }
        // (est_hi(~0): the bound the position is tightened to behind this stage -- the kernel flags a cut beyond it)
        if ((rc = enqueue_snp_stage(c, g, ct, K, nullptr, pos_hi, grew, late.S ? &aux8 : nullptr, grouped ? &df : nullptr, est_hi(~0ull)))) return rc;
        pos_hi = est_hi(pos_hi);
        if (grouped) {
            g->emit_d = (uint32_t)d;
            g->emit_items.push_back(EmitItem{ct.index, late.S, df.T, late.bmw, late.bnb, late.start, (uint32_t)K, df.W2, df.nb2,
                                             ct.d_recs, false});
            if (g->emit_items.size() >= (size_t)EMIT_G && (rc = gpu_emit_flush(c))) return rc;   // (the launch's job table is full)
        } else if (late.S) {
            SampleSet &S = *late.S;
            SnpSet &T = g->snp[(g->snp_unit - 1) % N_SETS];        // (the set enqueue_snp_stage just took)
            hipLaunchKernelGGL(k_bitmap_expand, dim3(late.bnb), dim3(BM_THREADS), 0, c->emit_stream,
                               reinterpret_cast<const uint64_t *>(S.bitmap), late.bmw, S.cnt2, late.start, (uint32_t)d,
                               ct.d_recs, (const uint8_t *)aux8);
            MSIM_HIP(c, hipGetLastError());
            MSIM_HIP(c, hipEventRecord(S.emit_done, c->emit_stream));
            S.wait_ev = S.emit_done;
            S.pending = true;
            MSIM_HIP(c, hipEventRecord(T.emit_done, c->emit_stream));
            T.wait_ev = T.emit_done;
            T.pending = true;
        }
    }
    py.pos = pos_hi;                                       // bound until gpu_plan_finish reads the exact value
    g->s[1].pos += 2 * K;                                  // numpy.random.choice(size=k): 2 words per candidate
    c->t.np_words += 2 * K;
    g->unverified = true;
    ct.planned = true;
    return MSIM_OK;
}

// ====================================================================== SV mixes (section 6 kernels)
static int grow_host(Ctx *c, void **p, size_t *cap, size_t want_bytes) {
    if (*cap >= want_bytes) return MSIM_OK;
    host_stage_free(*p);
    *p = nullptr; *cap = 0;
    const size_t sz = (want_bytes + want_bytes / 4 + STAGE_HUGE) & ~(STAGE_HUGE - 1);
    hipError_t e;
    void *q = host_stage_alloc(sz, &e);
    if (!q) return e != hipSuccess ? hip_fail(c, e, "hipHostRegister(host staging buffer)") : fail(c, MSIM_ERR_NOMEM, "host staging buffer");
    *p = q;
    *cap = sz;
    return MSIM_OK;
}

// Which non-SNP types can the type draw of this range produce?  (thresholds are cumulative: entry j is
// drawn iff thr[j-1] < thr[j], with thr[-1] = 0, and only below 2^53)
static bool range_type_drawable(const msim_range &r, int j) {
    const uint64_t lo = j ? r.cdf_thr[j - 1] : 0;
    return r.cdf_thr[j] > lo && lo < (1ull << 53);
}

static bool range_draws_translocations(const msim_range &r) {
    for (int j = 0; j < r.n_types; j++)
        if ((r.types[j] == MSIM_TL || r.types[j] == MSIM_TLI) && range_type_drawable(r, j)) return true;
    return false;
}

// SV mix on one large set-path range (ARGS mode: one range per contig): SNPs plus any of IN/DE/DU/IV/TL/TLI.
bool gpu_plan_mixed_eligible(const Ctx *c, const msim_range *ranges, int n_ranges) {
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);
    if (P.block[MSIM_SN] != d) return false;              // an SNP could block its successor: the chain needs all candidates
    const msim_range *one = nullptr;
    for (int i = 0; i < n_ranges; i++) {
        if (ranges[i].k == 0) continue;
        if (one) return false;                            // several drawing ranges: host planner
        one = &ranges[i];
    }
    if (!one) return false;
    const msim_range &r = *one;
    if (r.k < 4096 || r.k >= (1ll << 31)) return false;
    const int64_t n = (r.stop - (r.k - 1) * d) - r.start;
    if (n < r.k || n <= r.setsize || n >= (1ll << 32)) return false;
    if (r.start < 0 || r.stop >= (1ll << 32)) return false;
    if (r.n_types < 1 || r.n_types > 8) return false;
    for (int j = 0; j < r.n_types; j++) {
        if (!range_type_drawable(r, j)) continue;
        const int t = r.types[j];
        if (t == MSIM_SN) continue;
        if (t == MSIM_TLI) continue;                      // an insertion site: no length draw
        if (t != MSIM_IN && t != MSIM_DE && t != MSIM_DU && t != MSIM_IV && t != MSIM_TL) return false;
        const int64_t w = r.max_len[t] - r.min_len[t] + 1;
        if (r.min_len[t] < 1 || w < 1 || w >= (1ll << 32)) return false;
        if (t == MSIM_IN && (double)r.k * (double)r.max_len[t] >= 4.0e9) return false;     // insert pool offsets are 32-bit
    }
    for (int t = 1; t <= 7; t++)
        if (P.block[t] >= (1ll << 32)) return false;
    if (range_draws_translocations(r)) {                  // their walk exists over accept tables only (ChainWalk::run_tl)
        ChainClasses cc;
        if (!chain_classes(r, cc)) return false;
    }
    return true;
}

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

// Wait for a word in pinned host memory that a kernel on `s` raises to `want`.  The stream is queried now and then so
// that a failed launch or a device fault ends the wait with an error instead of a hang; a stream that neither drains nor
// raises the word ends it at the deadline of every host wait (ctx.h: MSIM_WAIT_TIMEOUT_S).
static int spin_until(Ctx *c, const uint32_t *word, uint32_t want, hipStream_t s) {
    std::chrono::steady_clock::time_point t0{};
    double limit = -1;
    for (uint64_t it = 1;; it++) {
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == want) return MSIM_OK;
        cpu_relax();
        if ((it & 0x3ffff) == 0) {
            const hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) {
                if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == want) return MSIM_OK;
                return fail(c, MSIM_ERR_HIP, "plan stream drained without raising its signal");
            }
            if (e != hipErrorNotReady) return hip_fail(c, e, "hipStreamQuery(plan stream)");
            const auto now = std::chrono::steady_clock::now();
            if (limit < 0) { t0 = now; limit = wait_limit_seconds(); }
            else if (limit > 0 && std::chrono::duration<double>(now - t0).count() > limit)
                return hip_fail(c, MSIM_WAIT_TIMED_OUT, "spin_until(mailbox signal of the plan stream)");
        }
    }
}

// (an event the host chain is about to need: polled without a pause -- it is microseconds away; `what` names it at the deadline)
static int spin_event(Ctx *c, hipEvent_t ev, const char *what = "spin_event(copy piece of the host chain)") {
    for (uint32_t it = 1;; it++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return MSIM_OK;
        if (e != hipErrorNotReady) return hip_fail(c, e, "hipEventQuery");
        cpu_relax();
        if ((it & 0xfff) == 0) {                           // ~ every few ms: hand over to the bounded wait
            const hipError_t w = wait_event(ev);
            if (w == hipSuccess) return MSIM_OK;
            return hip_fail(c, w, what);
        }
    }
}

static int ensure_signals(Ctx *c, GpuPlan *g) {
    if (!g->h_sig) {
        MSIM_HIP(c, hipHostMalloc(&g->h_sig, 4096, hipHostMallocMapped));
        memset(g->h_sig, 0, 4096);
    }
    if (!g->copy_stream) {
        {   // The candidates' copy stream: normal priority (MSIM_COPY_PRIO=1 / -1: the low-priority pool / the plan stream's), where
            // it shares a hardware queue with an (idle) side stream.  In the low pool it had a queue of its own and c3 behind c2
            // passes gained 2 % (30.5 -> 29.8 ms) -- until the fourth side stream went there: both in the low pool cost c3 5 ms per
            // step (GpuPlan::n_prep has the story).
            int lo = 0, hi = 0;
            MSIM_HIP(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
            const char *e = getenv("MSIM_COPY_PRIO");
            const int prio = !e ? 0 : atoi(e) > 0 ? lo : atoi(e) < 0 ? hi : 0;
            MSIM_HIP(c, hipStreamCreateWithPriority(&g->copy_stream, hipStreamNonBlocking, prio));
        }
        MSIM_HIP(c, hipEventCreateWithFlags(&g->ev_cand, hipEventDisableTiming));
        for (auto &e : g->ev_piece) MSIM_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : g->ev_cpiece) MSIM_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return MSIM_OK;
}

// Exact stream position + counts from the mailbox, chain time accounted -- without draining the stream: the mailbox
// kernel writes a sequence word last and the host polls it (a stream synchronisation costs 25-35 us of wake-up
// latency per call, twice per contig)
// behind_mailbox: device work the caller can already enqueue behind the mailbox kernel -- the host would otherwise sit in the
// poll and only then start launching it (20-35 us of launches + their latency, per contig)
template <class F>
static int mixed_poll(Ctx *c, GpuPlan *g, PlanState &h, F &&behind_mailbox) {
    int rc = ensure_signals(c, g);
    if (rc) return rc;
    if (++g->epoch == 0) g->epoch = 1;
    MSIM_HIP(c, hipEventRecord(g->t1, c->stream));
    hipLaunchKernelGGL(k_publish_seq, dim3(1), dim3(1), 0, c->stream, g->d_ps, g->h_mail, g->h_sig, g->epoch);
    MSIM_HIP(c, hipGetLastError());
    if ((rc = behind_mailbox())) return rc;
    if ((rc = spin_until(c, g->h_sig, g->epoch, c->stream))) return rc;
    float ms = 0;
    hipError_t e = hipEventElapsedTime(&ms, g->t0, g->t1);
    if (e == hipErrorNotReady) { MSIM_HIP(c, wait_event(g->t1)); e = hipEventElapsedTime(&ms, g->t0, g->t1); }
    MSIM_HIP(c, e);
    c->t.plan_gpu_ms += ms;
    h = *g->h_mail;
    if (h.flags & (FLAG_SAMPLE_OVERFLOW | FLAG_SNP_OVERFLOW)) {
        g->s[0].live = g->s[1].live = false;
        g->unverified = false;
        return fail(c, MSIM_ERR_HIP, "GPU sampler: a stream window overflowed its margin (16 sigma; 8 for the start of a sample planned ahead of the chain; results discarded)");
    }
    c->t.py_words += h.pos - g->verified_pos;
    g->verified_pos = h.pos;
    g->s[0].pos = h.pos;
    MSIM_HIP(c, hipEventRecord(g->t0, c->stream));        // the next span starts here
    return MSIM_OK;
}

static int mixed_poll(Ctx *c, GpuPlan *g, PlanState &h) {
    return mixed_poll(c, g, h, []() { return MSIM_OK; });
}

// The plan stream is about to idle while the host walks a chain: close the timed GPU span (t1 must have been
// recorded before the synchronisation that just returned) and reopen it when device work is enqueued again.
static int span_close(Ctx *c, GpuPlan *g) {
    float ms = 0;
    MSIM_HIP(c, hipEventElapsedTime(&ms, g->t0, g->t1));
    c->t.plan_gpu_ms += ms;
    return MSIM_OK;
}

// The previous contig's deferred APPLY goes out when this contig's host chain starts -- and behind the last of the copies
// that chain waits for, so that the rewrite kernel really runs on an idle device: launched right away it shared the CUs
// with k_accept_tables and the D2H copies of the window (c3: 1.54 -> 1.92 ms of rewrite kernel per step).
static int flush_deferred_apply_behind(Ctx *c, hipEvent_t last_copy) {
    if (c->deferred_apply >= 0) MSIM_HIP(c, hipStreamWaitEvent(c->emit_stream, last_copy, 0));
    return flush_deferred_apply(c, true);                  // (groups: an incomplete one stays for its successors)
}

// M.cnt: five u32 counter arrays of nbk + 2 entries, then (8-byte aligned) one 64-bit array of the same length
static inline size_t mixed_cnt_words(uint32_t nbk) { return ((size_t)5 * (nbk + 2) + 1) & ~(size_t)1; }
static inline size_t mixed_cnt_bytes(uint32_t nbk) { return mixed_cnt_words(nbk) * 4 + (size_t)(nbk + 2) * 8; }

// Steps 3 and 4 of an SV-mix plan, shared by the single-range engine and the host-chain engine: the stops of the chain's
// candidates are in M.cand_stop -> keep flags and counts -> records, insert pool, SNP draws.  p_s: where the SNP draws of
// __mutate_sequence start in the CPython stream.  rt / visit_from / sn_chained: see k_keep_flags (nullptr for one range).
static int mixed_emit(Ctx *c, GpuPlan *g, Contig &ct, MixedSet &M, uint32_t k, uint64_t p_s, const MixRangeDev *rt,
                      uint32_t n_draw, const uint32_t *visit_from, bool sn_chained, bool &grew, bool has_tl = false) {
    const uint32_t *c_extra = has_tl ? M.cand_extra : nullptr;
    const uint8_t *c_aux = has_tl ? M.cand_aux : nullptr;
    const msim_params &P = c->params;
    GpuStream &py = g->s[0], &np = g->s[1];
    int rc;
    const uint32_t nbk = (k + CB_BLOCK - 1) / CB_BLOCK;
    uint32_t *bmax = M.cnt + (nbk + 2), *cnt_keep = M.cnt + 2 * (nbk + 2), *cnt_sn = M.cnt + 3 * (nbk + 2),
             *cnt_ins = M.cnt + 4 * (nbk + 2);
    long long *blk_delta = reinterpret_cast<long long *>(M.cnt + mixed_cnt_words(nbk));   // (8-byte aligned: see mixed_cnt_bytes)
    hipLaunchKernelGGL(k_set_pos, dim3(1), dim3(1), 0, c->stream, g->d_ps, (unsigned long long)p_s);

    // ---- 3. keep flags, counts
    BlockTable bt{};
    for (int t = 1; t <= 7; t++) bt.p1[t] = (uint32_t)std::min<int64_t>(P.block[t] + 1, 0xffffffffll);
    hipLaunchKernelGGL(k_blk_reduce, dim3(nbk), dim3(CB_THREADS), 0, c->stream, M.cand_pos, M.cand_type, M.cand_stop, k, bt, bmax,
                       rt, n_draw);
    hipLaunchKernelGGL(k_scan_max_u32, dim3(1), dim3(1024), 0, c->stream, bmax, nbk);
    hipLaunchKernelGGL(k_keep_flags, dim3(nbk), dim3(CB_THREADS), 0, c->stream, M.cand_pos, M.cand_type, M.cand_stop, k, bt,
                       bmax, cnt_keep, cnt_sn, cnt_ins, blk_delta, rt, n_draw, visit_from, sn_chained ? 1u : 0u, c_extra, c_aux);
    hipLaunchKernelGGL(k_scan4, dim3(4), dim3(1024), 0, c->stream, cnt_keep, cnt_sn, cnt_ins, blk_delta, nbk, g->d_ps);
    MSIM_HIP(c, hipGetLastError());
    PlanState h;
    if ((rc = mixed_poll(c, g, h))) return rc;
    const uint32_t n_rec = h.n_rec, n_sn = h.n_sn, pool_len = h.pool_len;

    // ---- 4. records, insert pool, SNP draws
    if (!c->chain_only) {   // the record table / pool may still be read by an earlier apply of this contig
        const size_t want = std::max<uint64_t>(n_rec, 1) * sizeof(msim_record);
        if (ct.cap_recs < want || ct.cap_pool < pool_len + 2 * PAD) {
            MSIM_HIP(c, wait_stream(c->stream));
            MSIM_HIP(c, wait_stream(c->emit_stream));
        }
        if ((rc = dev_reserve(c, (void **)&ct.d_recs, &ct.cap_recs, want))) return rc;
        if ((rc = dev_reserve(c, (void **)&ct.d_pool, &ct.cap_pool, pool_len + 2 * PAD))) return rc;
        if ((rc = dev_reserve(c, (void **)&ct.d_off, &ct.cap_off, std::max<uint64_t>(n_rec, 1) * sizeof(uint32_t)))) return rc;
    }
    ct.n_rec = n_rec;
    ct.pool_len = pool_len;
    ct.plan_empty = n_rec == 0;
    ct.all_snp = h.n_rec == h.n_sn;
    ct.delta_known = true;
    ct.known_delta = h.len_delta;
    ct.off_ready = !c->chain_only;                         // k_emit_records leaves every record's output offset in d_off
    if (pool_len && (rc = ensure_words(c, g, 1, np.pos + pool_len + 1))) return rc;
    uint64_t pos_hi = p_s;
    hipEvent_t ce = next_chain_event(g);
    MSIM_HIP(c, hipEventRecord(ce, c->stream));
    MSIM_HIP(c, hipStreamWaitEvent(c->emit_stream, ce, 0));
    if (!c->chain_only) {
        hipLaunchKernelGGL(k_emit_records, dim3(nbk), dim3(CB_THREADS), 0, c->emit_stream, M.cand_pos, M.cand_type, M.cand_stop, k,
                           cnt_keep, cnt_sn, cnt_ins, blk_delta, ct.d_recs, M.sn_index, ct.d_off, c_extra, c_aux);
        if (pool_len)
            hipLaunchKernelGGL(k_pool_fill, dim3((pool_len / 4 + 256) / 256), dim3(256), 0, c->emit_stream, np.d_raw,
                               (unsigned long long)np.pos, pool_len, ct.d_pool + PAD);
        MSIM_HIP(c, hipGetLastError());
    }
    np.pos += pool_len;
    c->t.np_words += pool_len;
    if (n_sn) {                                          // SNP draws in position order (chain: scan + cut; aux off the chain)
        if ((rc = enqueue_snp_stage(c, g, ct, n_sn, M.sn_index, pos_hi, grew))) return rc;
    }
    MSIM_HIP(c, hipEventRecord(M.emit_done, c->emit_stream));
    M.pending = true;
    py.pos = pos_hi;                                       // bound until the next sync reads the exact value
    g->unverified = true;
    g->est_ok = false;                                    // (the SNP sampler's estimate of the position ends here)
    ct.planned = true;
    return MSIM_OK;
}

// __link_tls of the SV-mix engine (mutator.py:130-131 -- after the boundary pass, before __mutate_sequence): its draws follow
// the boundary pass's in the CPython stream, so once the walk knows where that ended, a 16-sigma window of tempered words
// from there comes over and plan_host.cpp links on it.  consumed grows by the words linking drew.
static int mixed_link_translocations(Ctx *c, GpuPlan *g, MixedSet &M, uint32_t n_nsn, uint64_t p_b, size_t &consumed, bool &grew) {
    size_t n_tl = 0, n_tli = 0;
    count_translocations(g->h_ntype, g->h_nstop, n_nsn, &n_tl, &n_tli);
    uint32_t Wl = 0;
    int rc;
    if (n_tl) {                                            // (no TL: `if tls:` is false and nothing is drawn)
        // |n_tl - n_tli| deletions, min - 1 shuffle swaps and one coin per pair (randint(0, 1) takes two bits and rejects half
        // of them): every draw accepts a word with probability >= 1/2 -- at most 2 words expected, variance 2
        const double mn = (double)std::min(n_tl, n_tli), draws = (double)(std::max(n_tl, n_tli) - std::min(n_tl, n_tli)) + 2.0 * mn;
        const double wl = 2.0 * draws + 16.0 * std::sqrt(2.0 * draws + 1.0) + 4096.0;
        if (wl >= 4.0e9) return fail(c, MSIM_ERR_UNSUPPORTED, "linking window beyond 2^32 words");
        Wl = (uint32_t)wl;
        const uint64_t p_l = p_b + consumed;
        if ((rc = ensure_words(c, g, 0, p_l + Wl + 1))) return rc;
        if ((rc = grow(c, (void **)&M.words, &M.cap_words, (size_t)Wl * 4 + 64, &grew))) return rc;
        if (g->cap_h_win < (size_t)Wl * 4 && (rc = grow_host(c, (void **)&g->h_win, &g->cap_h_win, (size_t)Wl * 4))) return rc;
        hipLaunchKernelGGL(k_temper_window, dim3((Wl + 255) / 256), dim3(256), 0, c->stream, g->s[0].d_raw, (unsigned long long)p_l, Wl,
                           M.words);
        MSIM_HIP(c, hipGetLastError());
        MSIM_HIP(c, hipMemcpyAsync(g->h_win, M.words, (size_t)Wl * 4, hipMemcpyDeviceToHost, c->stream));
        MSIM_HIP(c, wait_stream(c->stream));
    }
    size_t used = 0;
    if ((rc = link_translocations(c, g->h_win, Wl, g->h_npos, g->h_ntype, g->h_nstop, g->h_nextra, g->h_naux, n_nsn, &used))) return rc;
    consumed += used;
    return MSIM_OK;
}

int plan_contig_gpu_mixed(Ctx *c, GpuPlan *g, Contig &ct, const msim_range *ranges, int n_ranges) {
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);
    const msim_range *rp = nullptr;
    for (int i = 0; i < n_ranges; i++) if (ranges[i].k) rp = &ranges[i];
    const msim_range &r = *rp;
    const uint32_t k = (uint32_t)r.k;
    int rc;
    if ((rc = stream_to_device(c, g, 0))) return rc;
    if ((rc = stream_to_device(c, g, 1))) return rc;
    if (!g->d_ps) MSIM_HIP(c, hipMalloc(&g->d_ps, sizeof(PlanState)));
    if (!g->h_mail) MSIM_HIP(c, hipHostMalloc(&g->h_mail, sizeof(PlanState), hipHostMallocMapped));
    GpuStream &py = g->s[0], &np = g->s[1];
    if (!g->t0) { MSIM_HIP(c, hipEventCreate(&g->t0)); MSIM_HIP(c, hipEventCreate(&g->t1)); }
    if (!g->unverified) MSIM_HIP(c, hipEventRecord(g->t0, c->stream));
    if (!g->ps_valid) {
        hipLaunchKernelGGL(k_state_init, dim3(1), dim3(1), 0, c->stream, g->d_ps, (unsigned long long)py.pos);
        g->ps_valid = true;
    }
    g->unverified = true;
    g->est_ok = false;                                    // (the SNP sampler's estimate of the position ends here)
    bool grew = false;
    MixedSet &M = g->mixed[g->mixed_unit++ % N_SETS];
    if ((rc = wait_if_pending(c, M.pending, M.emit_done))) return rc;          // its last emit may still read it
    if (!M.emit_done) MSIM_HIP(c, hipEventCreateWithFlags(&M.emit_done, hipEventDisableTiming));
    const uint32_t nbk = (k + CB_BLOCK - 1) / CB_BLOCK;
    if ((rc = grow(c, (void **)&M.cand_pos, &M.cap_pos, (size_t)k * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.cand_type, &M.cap_type, (size_t)k + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.cand_stop, &M.cap_stop, (size_t)k * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.nsn_pos, &M.cap_npos, (size_t)k * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.nsn_type, &M.cap_ntype, (size_t)k + 64, &grew))) return rc;
    if (g->cap_h_npos < (size_t)k * 4 + 64 || g->cap_h_ntype < (size_t)k + 64 || g->cap_h_nstop < (size_t)k * 4 + 64) {
        MSIM_HIP(c, wait_stream(c->stream));     // an earlier contig's copies may still use the old blocks
        if (g->copy_stream) MSIM_HIP(c, wait_stream(g->copy_stream));
        if ((rc = grow_host(c, (void **)&g->h_npos, &g->cap_h_npos, (size_t)k * 4 + 64))) return rc;
        if ((rc = grow_host(c, (void **)&g->h_ntype, &g->cap_h_ntype, (size_t)k + 64))) return rc;
        if ((rc = grow_host(c, (void **)&g->h_nstop, &g->cap_h_nstop, (size_t)k * 4 + 64))) return rc;
    }
    if ((rc = grow(c, (void **)&M.nsn_rank, &M.cap_nrank, (size_t)k * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.nsn_stop, &M.cap_nstop, (size_t)k * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.sn_index, &M.cap_snidx, (size_t)k * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.cnt, &M.cap_cnt, mixed_cnt_bytes(nbk), &grew))) return rc;
    const bool has_tl = range_draws_translocations(r);
    if (has_tl) {                                          // what __link_tls decides, per candidate and in chain order
        if ((rc = grow(c, (void **)&M.cand_extra, &M.cap_cextra, (size_t)k * 4 + 64, &grew))) return rc;
        if ((rc = grow(c, (void **)&M.cand_aux, &M.cap_caux, (size_t)k + 64, &grew))) return rc;
        if ((rc = grow(c, (void **)&M.nsn_extra, &M.cap_nextra, (size_t)k * 4 + 64, &grew))) return rc;
        if ((rc = grow(c, (void **)&M.nsn_aux, &M.cap_naux, (size_t)k + 64, &grew))) return rc;
        if (g->cap_h_nextra < (size_t)k * 4 + 64 || g->cap_h_naux < (size_t)k + 64) {
            MSIM_HIP(c, wait_stream(c->stream));
            if (g->copy_stream) MSIM_HIP(c, wait_stream(g->copy_stream));
            if ((rc = grow_host(c, (void **)&g->h_nextra, &g->cap_h_nextra, (size_t)k * 4 + 64))) return rc;
            if ((rc = grow_host(c, (void **)&g->h_naux, &g->cap_h_naux, (size_t)k + 64))) return rc;
        }
    }
    uint32_t *cnt_nsn = M.cnt;                             // (the other four counter arrays: mixed_emit)

    // ---- 1. sample -> bitmap; bitmap -> candidates with types; non-SNP candidates compacted
    SampleLaunch sl;
    if ((rc = enqueue_sample_chain(c, g, r, d, py.pos, grew, sl))) return rc;
    SampleSet &S = *sl.S;
    const uint64_t np_base = np.pos;                       // exact: the NumPy stream never rejects
    if ((rc = ensure_words(c, g, 1, np_base + 2ull * k + 1))) return rc;
    TypeTable tt{};
    tt.n = (uint32_t)r.n_types;
    for (int j = 0; j < r.n_types; j++) { tt.thr[j] = r.cdf_thr[j]; tt.type[j] = (uint8_t)r.types[j]; }
    hipLaunchKernelGGL(k_bitmap_count, dim3(sl.bnb), dim3(BM_THREADS), 0, c->stream,
                       reinterpret_cast<const uint64_t *>(S.bitmap), sl.bmw, S.cnt2);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, c->stream, S.cnt2, sl.bnb);
    hipLaunchKernelGGL(k_bitmap_expand_cand, dim3(sl.bnb), dim3(BM_THREADS), 0, c->stream,
                       reinterpret_cast<const uint64_t *>(S.bitmap), sl.bmw, S.cnt2, (uint32_t)r.start, (uint32_t)d,
                       np.d_raw, (unsigned long long)np_base, tt, M.cand_pos, M.cand_type);
    hipLaunchKernelGGL(k_nsn_count, dim3(nbk), dim3(CB_THREADS), 0, c->stream, M.cand_type, k, cnt_nsn, 0u);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, c->stream, cnt_nsn, nbk);
    hipLaunchKernelGGL(k_nsn_scatter, dim3(nbk), dim3(CB_THREADS), 0, c->stream, M.cand_pos, M.cand_type, k, cnt_nsn, nbk,
                       M.nsn_pos, M.nsn_type, M.nsn_rank, g->d_ps, 0u);
    MSIM_HIP(c, hipGetLastError());
    S.pending = false;                                     // consumed on the plan stream itself
    // The host walks the non-SNP candidates.  Their copy starts at once, beside the plan stream and before their number
    // is known here: a 16-sigma bound of the type draw's binomial is copied (the rest, if ever, afterwards).
    if ((rc = ensure_signals(c, g))) return rc;
    double q_nsn = 0;
    for (int j = 0; j < r.n_types; j++) {
        const uint64_t lo = j ? r.cdf_thr[j - 1] : 0;
        if (r.types[j] != MSIM_SN && r.cdf_thr[j] > lo && lo < (1ull << 53))
            q_nsn += (double)(std::min<uint64_t>(r.cdf_thr[j], 1ull << 53) - lo) / 9007199254740992.0;
    }
    q_nsn = std::min(1.0, q_nsn);
    const uint32_t n_hi = (uint32_t)std::min<double>(k, q_nsn * k + 16.0 * std::sqrt(q_nsn * (1.0 - q_nsn) * k) + 64.0);
    const size_t ccut[4] = {0, (size_t)n_hi / 8, (size_t)n_hi / 2, (size_t)n_hi};
    {
        hipEvent_t se = next_chain_event(g);
        MSIM_HIP(c, hipEventRecord(se, c->stream));
        MSIM_HIP(c, hipStreamWaitEvent(g->copy_stream, se, 0));
        // in three pieces (1/8, 3/8, 1/2), like the tables: the walk starts on the first one -- with one copy of 3.75 MB (a big
        // contig of config 3) it was the candidates, not the table, that the walk waited for
        for (int q = 0; q < 3; q++) {
            const size_t a = ccut[q], b = ccut[q + 1];
            if (b > a) {
                MSIM_HIP(c, hipMemcpyAsync(g->h_npos + a, M.nsn_pos + a, (b - a) * 4, hipMemcpyDeviceToHost, g->copy_stream));
                MSIM_HIP(c, hipMemcpyAsync(g->h_ntype + a, M.nsn_type + a, b - a, hipMemcpyDeviceToHost, g->copy_stream));
            }
            MSIM_HIP(c, hipEventRecord(g->ev_cpiece[q], g->copy_stream));
        }
        MSIM_HIP(c, hipEventRecord(g->ev_cand, g->copy_stream));
    }
    // The accept tables of the boundary walk depend on where the sample ended (the device knows) and on the window's size --
    // which follows from the number of non-SNP candidates, known only after the poll: they are launched BEHIND the mailbox
    // kernel over a 16-sigma bound of it, so that they are under way while the host still polls.
    double acc_min = 1.0;                                  // least acceptance of randint among the types this range draws
    for (int j = 0; j < r.n_types; j++) {
        const int t = r.types[j];
        if (t == MSIM_SN || t == MSIM_TLI || !range_type_drawable(r, j)) continue;
        const int64_t w = r.max_len[t] - r.min_len[t] + 1;
        if (w >= 1 && w < (1ll << 32)) acc_min = std::min(acc_min, (double)w / (double)(1ull << bit_length64((uint64_t)w)));
    }
    auto window_of = [&](double n) { return n / acc_min + 16.0 * std::sqrt(n) / acc_min + 4096.0; };
    ChainClasses cc;
    const bool tables = chain_classes(r, cc);              // the host walks "next accepted draw" tables (ctx.h)
    const uint32_t lg = chain_lg_rows(cc);
    const bool early = tables && n_hi > 0 && window_of((double)n_hi) < 4.0e9;
    const uint32_t Wb_hi = early ? (uint32_t)window_of((double)n_hi) : 0u;
    size_t cut[4] = {0, 0, 0, 0};
    if (early) {
        const size_t bytes = ((size_t)(Wb_hi + 1) << lg) * 4;
        if (g->cap_h_words < bytes) {
            MSIM_HIP(c, wait_stream(c->stream));
            if ((rc = grow_host(c, (void **)&g->h_words, &g->cap_h_words, bytes))) return rc;
        }
        if ((rc = grow(c, (void **)&M.words, &M.cap_words, bytes, &grew))) return rc;
        if ((rc = ensure_words(c, g, 0, py.pos + sl.W + Wb_hi + 2))) return rc;      // (the sample ends below py.pos + W)
    }
    PlanState h;
    const auto tq0 = std::chrono::steady_clock::now();
    rc = mixed_poll(c, g, h, [&]() -> int {
        if (!early) return MSIM_OK;
        const size_t n_slots = (size_t)(Wb_hi + 1) << lg;
        hipLaunchKernelGGL(k_accept_tables_ps, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, c->stream, py.d_raw, g->d_ps, Wb_hi,
                           cc, lg, M.words);
        MSIM_HIP(c, hipGetLastError());
        // three pieces (1/8, 3/8, 1/2 of the positions): the walk starts on the first one while the others are in flight -- it
        // consumes the table at ~3 GB/s, the copies deliver 50 GB/s
        cut[1] = (size_t)(Wb_hi + 1) / 8; cut[2] = (size_t)(Wb_hi + 1) / 2; cut[3] = (size_t)Wb_hi + 1;
        for (int q = 0; q < 3; q++) {
            if (cut[q + 1] > cut[q])
                MSIM_HIP(c, hipMemcpyAsync(g->h_words + (cut[q] << lg), M.words + (cut[q] << lg), ((cut[q + 1] - cut[q]) << lg) * 4,
                                           hipMemcpyDeviceToHost, c->stream));
            MSIM_HIP(c, hipEventRecord(g->ev_piece[q], c->stream));
        }
        return MSIM_OK;
    });
    if (rc) return rc;
    const auto tq1 = std::chrono::steady_clock::now();
    const uint32_t n_nsn = h.n_nsn;
    const uint64_t p_b = h.pos;                            // the boundary pass draws from here
    np.pos = np_base + 2ull * k;
    c->t.np_words += 2ull * k;

    // ---- 2. the sequential chain over the non-SNP candidates, on the host
    size_t consumed = 0;
    const bool early_ok = early && n_nsn > 0 && n_nsn <= n_hi;    // (beyond 16 sigma: the window is built again, exactly, below)
    if (early && !early_ok) MSIM_HIP(c, wait_stream(c->stream));   // nobody reads that table: the blocks are free again
    if (n_nsn) {
        const double wb = window_of((double)n_nsn);
        if (wb >= 4.0e9) return fail(c, MSIM_ERR_UNSUPPORTED, "boundary window beyond 2^32 words");
        const uint32_t Wb = early_ok ? Wb_hi : (uint32_t)wb;
        const size_t words_bytes = tables ? ((size_t)(Wb + 1) << lg) * 4 : (size_t)Wb * 4;
        if (!early_ok) {
            if (g->cap_h_words < words_bytes) {
                MSIM_HIP(c, wait_stream(c->stream));
                if ((rc = grow_host(c, (void **)&g->h_words, &g->cap_h_words, words_bytes))) return rc;
            }
            if ((rc = grow(c, (void **)&M.words, &M.cap_words, words_bytes, &grew))) return rc;
            if ((rc = ensure_words(c, g, 0, p_b + Wb + 1))) return rc;
        }
        if (n_nsn > n_hi) {                                // beyond 16 sigma: the rest of the candidates
            MSIM_HIP(c, hipMemcpyAsync(g->h_npos + n_hi, M.nsn_pos + n_hi, (size_t)(n_nsn - n_hi) * 4, hipMemcpyDeviceToHost, g->copy_stream));
            MSIM_HIP(c, hipMemcpyAsync(g->h_ntype + n_hi, M.nsn_type + n_hi, (size_t)(n_nsn - n_hi), hipMemcpyDeviceToHost, g->copy_stream));
            MSIM_HIP(c, hipEventRecord(g->ev_cand, g->copy_stream));
        }
        if (tables) {
            if (!early_ok) {                               // (not launched behind the mailbox: now, with the exact window)
                const size_t n_slots = (size_t)(Wb + 1) << lg;
                hipLaunchKernelGGL(k_accept_tables, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, c->stream, py.d_raw,
                                   (unsigned long long)p_b, Wb, cc, lg, M.words);
                MSIM_HIP(c, hipGetLastError());
                cut[0] = 0; cut[1] = (size_t)(Wb + 1) / 8; cut[2] = (size_t)(Wb + 1) / 2; cut[3] = (size_t)Wb + 1;
                for (int q = 0; q < 3; q++) {
                    if (cut[q + 1] > cut[q])
                        MSIM_HIP(c, hipMemcpyAsync(g->h_words + (cut[q] << lg), M.words + (cut[q] << lg),
                                                   ((cut[q + 1] - cut[q]) << lg) * 4, hipMemcpyDeviceToHost, c->stream));
                    MSIM_HIP(c, hipEventRecord(g->ev_piece[q], c->stream));
                }
            }
            MSIM_HIP(c, hipEventRecord(g->t1, c->stream));
            if (c->deferred_apply >= 0) MSIM_HIP(c, hipStreamWaitEvent(c->emit_stream, g->ev_cand, 0));   // (... and behind the candidates' copies)
            if ((rc = flush_deferred_apply_behind(c, g->ev_piece[2]))) return rc;   // the previous contig's APPLY
            const auto w0 = std::chrono::steady_clock::now();
            ChainWalk cw;
            if ((rc = cw.init(c, r, ct.len, cc, Wb))) return rc;
            static const bool prof = getenv("MSIM_CHAIN_PROF") != nullptr;
            double t_wait = 0, t_run = 0;
            auto tp = std::chrono::steady_clock::now();
            auto lap = [&](double &acc) { const auto n = std::chrono::steady_clock::now(); acc += std::chrono::duration<double, std::micro>(n - tp).count(); tp = n; };
            // candidates and table both arrive in pieces; the walk runs over what is there and waits for whichever ran out
            int ci = 0, ti = 0;
            size_t avail_c = 0, avail_w = 0;
            while (!rc && cw.j < n_nsn) {
                const bool need_c = cw.j >= avail_c, need_w = (cw.ws >> cw.lg_rows) >= avail_w;
                if (!need_c && !need_w) break;             // (run stops for one of the two reasons only)
                if (need_c) {
                    const size_t had = avail_c;
                    if (ci < 3) { if ((rc = spin_event(c, g->ev_cpiece[ci], "spin_event(candidate piece, copy stream)"))) break; avail_c = std::min<size_t>(ccut[++ci], n_nsn); }
                    else if (avail_c < n_nsn) { if ((rc = spin_event(c, g->ev_cand, "spin_event(candidates of the host chain, copy stream)"))) break; avail_c = n_nsn; }   // (beyond 16 sigma)
                    else break;
                    if (!ChainWalk::types_ok(g->h_ntype + had, avail_c - had, has_tl)) {
                        rc = fail(c, MSIM_ERR_HIP, "boundary chain: candidate type outside the range's draw");
                        break;
                    }
                }
                if (need_w) {
                    if (ti >= 3) break;                    // the window is used up: finish() reports it
                    if ((rc = spin_event(c, g->ev_piece[ti], "spin_event(accept-table piece, plan stream)"))) break;
                    avail_w = cut[++ti];
                }
                lap(t_wait);
                if (has_tl) cw.run_tl(g->h_npos, g->h_ntype, avail_c, g->h_words, avail_w, g->h_nstop);
                else cw.run(g->h_npos, g->h_ntype, avail_c, g->h_words, avail_w, g->h_nstop);
                lap(t_run);
            }
            if (!rc) rc = spin_event(c, g->ev_cand, "spin_event(candidates of the host chain, copy stream)");       // (every copy into the pinned blocks has landed before they are reused)
            if (prof) fprintf(stderr, "chain: n_nsn %u Wb %u wait %.0f us run %.0f us (%.2f ns/cand) w %zu | poll %.0f us, enqueue %.0f us\n", n_nsn, Wb, t_wait, t_run, t_run * 1e3 / n_nsn, cw.ws >> cw.lg_rows,
                              std::chrono::duration<double, std::micro>(tq1 - tq0).count(), std::chrono::duration<double, std::micro>(w0 - tq1).count());
            if (!rc) rc = cw.finish(c, n_nsn, &consumed);
            c->t.host_walk_run_ms += t_run * 1e-3;
            c->t.host_walk_wait_ms += t_wait * 1e-3 + std::chrono::duration<double, std::milli>(tq1 - tq0).count();
            c->t.host_walk_candidates += n_nsn;
            c->t.plan_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
            if (!rc) {
                MSIM_HIP(c, wait_event(g->t1));   // long complete
                rc = span_close(c, g);
            }
        } else {
            hipLaunchKernelGGL(k_temper_window, dim3((Wb + 255) / 256), dim3(256), 0, c->stream, py.d_raw,
                               (unsigned long long)p_b, Wb, M.words);
            MSIM_HIP(c, hipGetLastError());
            MSIM_HIP(c, hipMemcpyAsync(g->h_words, M.words, words_bytes, hipMemcpyDeviceToHost, c->stream));
            MSIM_HIP(c, hipEventRecord(g->t1, c->stream));
            if ((rc = flush_deferred_apply(c, true))) return rc;
            MSIM_HIP(c, wait_stream(c->stream));
            MSIM_HIP(c, wait_stream(g->copy_stream));
            if ((rc = span_close(c, g))) return rc;
            size_t kept_host = 0;
            long long delta_host = 0;                      // both are summed on the device, where the stops end up
            rc = chain_boundary_host(c, r, ct.len, g->h_npos, g->h_ntype, n_nsn, g->h_words, Wb, g->h_nstop,
                                     &consumed, &kept_host, &delta_host);
        }
        if (!rc && has_tl) rc = mixed_link_translocations(c, g, M, n_nsn, p_b, consumed, grew);
        if (rc) { g->s[0].live = g->s[1].live = false; g->unverified = false; return rc; }
        MSIM_HIP(c, hipEventRecord(g->t0, c->stream));    // the host chain is not GPU time
        MSIM_HIP(c, hipMemcpyAsync(M.nsn_stop, g->h_nstop, (size_t)n_nsn * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_stop_scatter, dim3((n_nsn + 255) / 256), dim3(256), 0, c->stream, M.nsn_rank, M.nsn_stop, n_nsn,
                           M.cand_stop);
    }
    if (has_tl) {
        MSIM_HIP(c, hipMemsetAsync(M.cand_aux, 0, (size_t)k, c->stream));
        if (n_nsn) {
            MSIM_HIP(c, hipMemcpyAsync(M.nsn_extra, g->h_nextra, (size_t)n_nsn * 4, hipMemcpyHostToDevice, c->stream));
            MSIM_HIP(c, hipMemcpyAsync(M.nsn_aux, g->h_naux, (size_t)n_nsn, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_link_scatter, dim3((n_nsn + 255) / 256), dim3(256), 0, c->stream, M.nsn_rank, M.nsn_extra, M.nsn_aux,
                               n_nsn, M.cand_extra, M.cand_aux);
        }
        MSIM_HIP(c, hipGetLastError());
    }
    const uint64_t p_s = p_b + consumed;                   // the SNP draws of __mutate_sequence start here
    return mixed_emit(c, g, ct, M, k, p_s, nullptr, 0, nullptr, false, grew, has_tl);
}


// ====================================================================== host-cut contigs
// Deterministic-SNP ranges of any size and number (RMT gene-blocking files: thousands of small ranges per
// contig, pool-path hot spots): the stream cuts form a chain of thousands of tiny data-dependent samples,
// so the host finds them (plan_host.cpp: cut_ranges_host) -- over words the DEVICE generated, and nothing
// but the cuts: the device repeats the acceptance test over every range's interval, builds the sorted sample
// in a contig-wide bitmap and from it the records (k_interval_bits, k_walk_expand); the SNP transducer of
// section 5 and APPLY follow as for every other engine.
static bool range_is_deterministic_sn(const msim_range &r) {
    if (r.n_types < 1 || r.n_types > 8) return false;
    int zeros = 0;
    for (int j = 0; j < r.n_types; j++) {
        if (r.cdf_thr[j] == 0) zeros++;
        else if (r.cdf_thr[j] < (1ull << 53)) return false;
    }
    return zeros < r.n_types && r.types[zeros] == MSIM_SN;
}

bool gpu_plan_hostsample_eligible(const Ctx *c, const msim_range *ranges, int n_ranges) {
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);
    if (P.block[MSIM_SN] != d) return false;
    int64_t prev_stop = -1;
    uint64_t K = 0;
    for (int i = 0; i < n_ranges; i++) {
        const msim_range &r = ranges[i];
        if (r.k == 0) continue;
        const int64_t n = (r.stop - (r.k - 1) * d) - r.start;
        if (r.k < 0 || n < r.k || n >= (1ll << 32)) return false;        // ValueError / multi-word: host planner decides
        if (r.start <= prev_stop || r.stop >= (1ll << 32)) return false; // overlapping or unsorted ranges: dict semantics
        if (!range_is_deterministic_sn(r)) return false;
        prev_stop = r.stop;
        K += (uint64_t)r.k;
    }
    return K > 0 && K < (1ull << 31);
}

int plan_contig_gpu_hostsample(Ctx *c, GpuPlan *g, Contig &ct, const msim_range *ranges, int n_ranges) {
    static const bool prof = getenv("MSIM_CHAIN_PROF") != nullptr;
    static std::chrono::steady_clock::time_point last_exit;
    const auto tp0 = std::chrono::steady_clock::now();
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);
    int rc;
    if ((rc = stream_to_device(c, g, 0))) return rc;
    if ((rc = stream_to_device(c, g, 1))) return rc;
    if (!g->d_ps) MSIM_HIP(c, hipMalloc(&g->d_ps, sizeof(PlanState)));
    if (!g->h_mail) MSIM_HIP(c, hipHostMalloc(&g->h_mail, sizeof(PlanState), hipHostMallocMapped));
    GpuStream &py = g->s[0];
    if (!g->t0) { MSIM_HIP(c, hipEventCreate(&g->t0)); MSIM_HIP(c, hipEventCreate(&g->t1)); }
    if (!g->unverified) MSIM_HIP(c, hipEventRecord(g->t0, c->stream));
    if (!g->ps_valid) {
        hipLaunchKernelGGL(k_state_init, dim3(1), dim3(1), 0, c->stream, g->d_ps, (unsigned long long)py.pos);
        g->ps_valid = true;
    }
    g->unverified = true;
    g->est_ok = false;                                    // (the SNP sampler's estimate of the position ends here)
    // word window: expected consumption of every sample + 16 sigma of the total
    uint64_t K = 0, K_pool = 0;
    uint32_t n_draw = 0;
    double e_words = 0, var = 0;
    for (int i = 0; i < n_ranges; i++) {
        const msim_range &r = ranges[i];
        if (r.k == 0) continue;
        n_draw++;
        const double k = (double)r.k, n = (double)((r.stop - (r.k - 1) * d) - r.start);
        K += (uint64_t)r.k;
        if (n <= (double)r.setsize) { K_pool += (uint64_t)r.k; e_words += 2.0 * k; var += 2.0 * k; continue; }   // pool path: < 2 words per draw
        const double p_acc = n / (double)(1ull << bit_length64((uint64_t)n));
        const double need = k >= n ? 64.0 * k : -n * std::log1p(-k / n);                // coupon collector
        e_words += need / p_acc;
        var += need * (1.0 - p_acc) / (p_acc * p_acc) + 4.0 * (need - k) / (p_acc * p_acc) + need / p_acc;
    }
    const double wd = e_words + 16.0 * std::sqrt(var) + 65536.0;
    if (wd >= 4.0e9) return fail(c, MSIM_ERR_UNSUPPORTED, "sample window beyond 2^32 words");
    const uint32_t W = (uint32_t)wd;
    bool grew = false;
    MixedSet &M = g->mixed[g->mixed_unit++ % N_SETS];
    if ((rc = wait_if_pending(c, M.pending, M.emit_done))) return rc;
    if (M.pending) {                                       // the pinned range table of this set may still be in flight
        MSIM_HIP(c, wait_event(M.emit_done));
        M.pending = false;
    }
    if (!M.emit_done) MSIM_HIP(c, hipEventCreateWithFlags(&M.emit_done, hipEventDisableTiming));
    // what the host hands back: one cut per drawing range (+ the end) and the pool-path positions
    const size_t n_back = (size_t)n_draw + 1 + K_pool;
    if ((rc = grow(c, (void **)&M.words, &M.cap_words, (size_t)W * 4, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.cand_pos, &M.cap_pos, n_back * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.walk_d, &M.cap_walk_d, (size_t)n_draw * sizeof(WalkRange) + 64, &grew))) return rc;
    if ((rc = grow_host(c, (void **)&M.walk_h, &M.cap_walk_h, (size_t)n_draw * sizeof(WalkRange) + 64))) return rc;
    if (g->cap_h_words < (size_t)W * 4 || g->cap_h_npos < n_back * 4 + 64) {
        MSIM_HIP(c, wait_stream(c->stream));     // an earlier contig's copies may still use the old blocks
        if ((rc = grow_host(c, (void **)&g->h_words, &g->cap_h_words, (size_t)W * 4))) return rc;
        if ((rc = grow_host(c, (void **)&g->h_npos, &g->cap_h_npos, n_back * 4 + 64))) return rc;
    }
    if (!M.p0_slot) MSIM_HIP(c, hipMalloc(&M.p0_slot, 64));
    const uint32_t bmw = (uint32_t)((ct.len + 63) / 64);  // contig-wide bitmap, 64-bit words
    const uint32_t bnb = (bmw + BM_THREADS - 1) / BM_THREADS;
    {
        const size_t want = (size_t)bmw * 8 + 64;
        if (M.cap_wbits < want) {
            const size_t before = M.cap_wbits;
            if ((rc = grow(c, (void **)&M.wbits, &M.cap_wbits, want, &grew))) return rc;
            if (M.cap_wbits != before) MSIM_HIP(c, hipMemset(M.wbits, 0, M.cap_wbits));   // k_walk_expand leaves it zeroed
        }
        if (M.wbits_dirty) {
            MSIM_HIP(c, hipMemsetAsync(M.wbits, 0, M.cap_wbits, c->stream));
            M.wbits_dirty = false;
        }
    }
    if ((rc = grow(c, (void **)&M.wcnt, &M.cap_wcnt, (size_t)(bnb + 2) * sizeof(uint32_t), &grew))) return rc;
    if (!c->chain_only) {   // the record table may still be read by an earlier apply of this contig
        const size_t want = (size_t)K * sizeof(msim_record);
        if (ct.cap_recs < want || ct.cap_pool < 2 * PAD) {
            MSIM_HIP(c, wait_stream(c->stream));
            MSIM_HIP(c, wait_stream(c->emit_stream));
        }
        if ((rc = dev_reserve(c, (void **)&ct.d_recs, &ct.cap_recs, want))) return rc;
        if ((rc = dev_reserve(c, (void **)&ct.d_pool, &ct.cap_pool, 2 * PAD))) return rc;
    }
    {
        uint32_t at = 0, base = 0;
        for (int i = 0; i < n_ranges; i++) {
            const msim_range &r = ranges[i];
            if (r.k == 0) continue;
            const int64_t n = (r.stop - (r.k - 1) * d) - r.start;
            WalkRange w;
            w.start = (uint32_t)r.start; w.k = (uint32_t)r.k; w.n = (uint32_t)n; w.rec_base = base;
            w.pool = n <= r.setsize ? 1u : 0u;
            M.walk_h[at++] = w;
            base += (uint32_t)r.k;
        }
    }
    if ((rc = ensure_words(c, g, 0, py.pos + W + 1))) return rc;
    hipLaunchKernelGGL(k_temper_window_ps, dim3((W + 255) / 256), dim3(256), 0, c->stream, py.d_raw, g->d_ps, W, M.words);
    MSIM_HIP(c, hipGetLastError());
    // the window comes over in three pieces (1/8, 3/8, 1/2); the host starts on the first while the others are in flight
    if ((rc = ensure_signals(c, g))) return rc;
    struct Feed { Ctx *c; GpuPlan *g; size_t cut[4]; int next; double wait_us; } fd{c, g, {0, (size_t)W / 8, (size_t)W / 2, (size_t)W}, 0, 0.0};
    for (int q = 0; q < 3; q++) {
        if (fd.cut[q + 1] > fd.cut[q])
            MSIM_HIP(c, hipMemcpyAsync(g->h_words + fd.cut[q], M.words + fd.cut[q], (fd.cut[q + 1] - fd.cut[q]) * 4,
                                       hipMemcpyDeviceToHost, c->stream));
        MSIM_HIP(c, hipEventRecord(g->ev_piece[q], c->stream));
    }
    MSIM_HIP(c, hipEventRecord(g->t1, c->stream));
    MSIM_HIP(c, hipMemcpyAsync(M.walk_d, M.walk_h, (size_t)n_draw * sizeof(WalkRange), hipMemcpyHostToDevice, c->stream));
    const auto tp1 = std::chrono::steady_clock::now();
    const auto tp2 = tp1;
    // ---- the host finds the stream cuts (and samples the pool-path ranges); everything per position stays here
    size_t consumed = 0, n_pool_pos = 0;
    uint32_t *h_cut = g->h_npos, *h_pool = g->h_npos + n_draw + 1;
    WordFeed feed;
    feed.user = &fd;
    feed.more = [](void *u, size_t *avail) -> int {
        Feed &f = *static_cast<Feed *>(u);
        if (f.next >= 3) return MSIM_OK;
        const auto w0 = std::chrono::steady_clock::now();
        const int rc = spin_event(f.c, f.g->ev_piece[f.next], "spin_event(word-window piece, plan stream)");
        f.wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        if (!rc) *avail = f.cut[++f.next];
        return rc;
    };
    if ((rc = flush_deferred_apply_behind(c, g->ev_piece[2]))) return rc;   // the previous contig's APPLY
    {
        const auto cw0 = std::chrono::steady_clock::now();
        rc = cut_ranges_host(c, ranges, n_ranges, d, g->h_words, W, h_cut, h_pool, &n_pool_pos, &consumed, &feed);
        const double all_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - cw0).count();
        c->t.host_walk_wait_ms += fd.wait_us * 1e-3;
        c->t.host_walk_run_ms += std::max(0.0, all_ms - fd.wait_us * 1e-3);
        c->t.host_cut_words += consumed;
    }
    if (!rc) {
        MSIM_HIP(c, wait_event(g->t1));           // all pieces landed (usually long ago): the pinned window is free again
        rc = span_close(c, g);
    }
    if (rc) { g->s[0].live = g->s[1].live = false; g->unverified = false; M.wbits_dirty = true; return rc; }
    const auto tp3 = std::chrono::steady_clock::now();
    MSIM_HIP(c, hipEventRecord(g->t0, c->stream));        // the host chain is not GPU time
    MSIM_HIP(c, hipMemcpyAsync(M.cand_pos, g->h_npos, ((size_t)n_draw + 1 + n_pool_pos) * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_advance_pos_save, dim3(1), dim3(1), 0, c->stream, g->d_ps, (unsigned long long)consumed, M.p0_slot);
    MSIM_HIP(c, hipGetLastError());
    if (!c->chain_only) {   // records on the emit stream (ordered after any earlier APPLY that still reads this contig's table): the
        // words of every range -> contig-wide bitmap -> sorted sample -> records
        hipEvent_t ce0 = next_chain_event(g);
        MSIM_HIP(c, hipEventRecord(ce0, c->stream));
        MSIM_HIP(c, hipStreamWaitEvent(c->emit_stream, ce0, 0));
        hipLaunchKernelGGL(k_interval_bits, dim3(((uint32_t)consumed + 255) / 256), dim3(256), 0, c->emit_stream, py.d_raw,
                           M.p0_slot, (uint32_t)consumed, M.cand_pos, M.walk_d, n_draw, reinterpret_cast<uint32_t *>(M.wbits));
        if (n_pool_pos)
            hipLaunchKernelGGL(k_list_to_bits, dim3(((uint32_t)n_pool_pos + 255) / 256), dim3(256), 0, c->emit_stream,
                               M.cand_pos + n_draw + 1, (uint32_t)n_pool_pos, reinterpret_cast<uint32_t *>(M.wbits));
        hipLaunchKernelGGL(k_bitmap_count, dim3(bnb), dim3(BM_THREADS), 0, c->emit_stream,
                           reinterpret_cast<const uint64_t *>(M.wbits), bmw, M.wcnt);
        hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, c->emit_stream, M.wcnt, bnb);
        hipLaunchKernelGGL(k_walk_expand, dim3(bnb), dim3(BM_THREADS), 0, c->emit_stream,
                           reinterpret_cast<uint64_t *>(M.wbits), bmw, M.wcnt, M.walk_d, n_draw, (uint32_t)d, ct.d_recs);
        MSIM_HIP(c, hipGetLastError());
    }
    ct.n_rec = K;
    ct.pool_len = 0;
    ct.plan_empty = false;
    ct.all_snp = true;
    uint64_t pos_hi = py.pos + consumed;
    if (K) {                                          // SNP draws in position order (chain: scan + cut; aux off the chain)
        if ((rc = enqueue_snp_stage(c, g, ct, K, nullptr, pos_hi, grew))) return rc;
    }
    MSIM_HIP(c, hipEventRecord(M.emit_done, c->emit_stream));
    M.pending = true;
    py.pos = pos_hi;
    g->s[1].pos += 2 * K;                                  // numpy.random.choice(size=k) per drawing range
    c->t.np_words += 2 * K;
    ct.planned = true;
    if (prof) {
        const auto tp4 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "hostcut: n_draw %u K %llu W %u | outside %.0f setup %.0f wait %.0f cut %.0f post %.0f us\n", n_draw, (unsigned long long)K, W,
                us(last_exit, tp0), us(tp0, tp1), us(tp1, tp2), us(tp2, tp3), us(tp3, tp4));
        last_exit = tp4;
    }
    return MSIM_OK;
}


// ====================================================================== host-chain engine
// Contigs the three engines above decline but whose stream structure is still a plain chain: several drawing ranges with
// their own settings (RMT: gene blocks + an SV `std` line, hot / cold ranges), SV types on many small ranges, an SNP
// block above the sampling distance.  Per range the reference runs sample -> type draw -> boundary pass before it touches
// the next range (mutator.py:116-123,144-214), so the CPython stream interleaves samples and randint draws and the whole
// contig is ONE chain: the host walks it (plan_host.cpp: multimix_walk_host) over a window of tempered words, the accept
// tables of every randint class (k_accept_tables_ps) and the candidate TYPES -- a function of the candidate's ordinal
// alone (k_types_multi), hence known before any position is -- all produced on the device.  Back on the device: the
// SNP filter (blocked ends clipped per range), the visit filter across range borders, records, insert pool, SNP draws.
constexpr uint64_t MM_MIN_K = 1024;                      // below: the host planner is as fast as the round trips

bool gpu_plan_multimix_eligible(const Ctx *c, GpuPlan *g, uint64_t L, const msim_range *ranges, int n_ranges) {
    g->mm_for = nullptr;
    if (!multimix_prepare(c, L, ranges, n_ranges, g->mm_sets)) return false;
    if (g->mm_sets.K < MM_MIN_K) return false;
    g->mm_for = ranges; g->mm_n = n_ranges;
    return true;
}

int plan_contig_gpu_multimix(Ctx *c, GpuPlan *g, Contig &ct, const msim_range *ranges, int n_ranges) {
    static const bool prof = getenv("MSIM_CHAIN_PROF") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    const msim_params &P = c->params;
    int64_t d = P.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, P.block[t]);
    int rc;
    if (g->mm_for != ranges || g->mm_n != n_ranges) {
        if (!multimix_prepare(c, ct.len, ranges, n_ranges, g->mm_sets)) return fail(c, MSIM_ERR_UNSUPPORTED, "outside the host-chain engine");
    }
    g->mm_for = nullptr;
    const MixSets &ms = g->mm_sets;
    const uint32_t K = (uint32_t)ms.K, n_draw = ms.n_draw, n_sets = (uint32_t)ms.rep.size();
    const uint32_t lg = chain_lg_rows(ms.gcc);
    if ((rc = stream_to_device(c, g, 0))) return rc;
    if ((rc = stream_to_device(c, g, 1))) return rc;
    if (!g->d_ps) MSIM_HIP(c, hipMalloc(&g->d_ps, sizeof(PlanState)));
    if (!g->h_mail) MSIM_HIP(c, hipHostMalloc(&g->h_mail, sizeof(PlanState), hipHostMallocMapped));
    GpuStream &py = g->s[0], &np = g->s[1];
    if (!g->t0) { MSIM_HIP(c, hipEventCreate(&g->t0)); MSIM_HIP(c, hipEventCreate(&g->t1)); }
    if (!g->unverified) MSIM_HIP(c, hipEventRecord(g->t0, c->stream));
    if (!g->ps_valid) {
        hipLaunchKernelGGL(k_state_init, dim3(1), dim3(1), 0, c->stream, g->d_ps, (unsigned long long)py.pos);
        g->ps_valid = true;
    }
    g->unverified = true;
    g->est_ok = false;                                    // (the SNP sampler's estimate of the position ends here)
    // ---- sizes: the word window (every sample + every randint, 16 sigma of the total) and the chain length
    double e_words = 0, var = 0, e_ch = 0, var_ch = 0;
    std::vector<double> q_nsn(n_sets, 0.0), acc_min(n_sets, 1.0), q_tl(n_sets, 0.0);
    for (uint32_t s = 0; s < n_sets; s++) {
        const msim_range &r = ranges[ms.rep[s]];
        for (int j = 0; j < r.n_types; j++) {
            if (!range_type_drawable(r, j) || r.types[j] == MSIM_SN) continue;
            const uint64_t lo = j ? r.cdf_thr[j - 1] : 0;
            const double q = (double)(std::min<uint64_t>(r.cdf_thr[j], 1ull << 53) - lo) / 9007199254740992.0;
            q_nsn[s] += q;
            if (r.types[j] == MSIM_TL || r.types[j] == MSIM_TLI) q_tl[s] += q;
            if (r.types[j] == MSIM_TLI) continue;                     // (draws nothing in the boundary pass)
            const int64_t w = r.max_len[r.types[j]] - r.min_len[r.types[j]] + 1;
            acc_min[s] = std::min(acc_min[s], (double)w / (double)(1ull << bit_length64((uint64_t)w)));
        }
        q_nsn[s] = std::min(1.0, q_nsn[s]);
    }
    {
        uint32_t di = 0;
        for (int i = 0; i < n_ranges; i++) {
            const msim_range &r = ranges[i];
            if (r.k == 0) continue;
            const uint32_t s = ms.set_of[di++];
            const double k = (double)r.k, n = (double)((r.stop - (r.k - 1) * d) - r.start);
            if (n <= (double)r.setsize) { e_words += 2.0 * k; var += 2.0 * k; }           // pool path: < 2 words per draw
            else {
                const double p_acc = n / (double)(1ull << bit_length64((uint64_t)n));
                const double need = k >= n ? 64.0 * k : -n * std::log1p(-k / n);          // coupon collector
                e_words += need / p_acc;
                var += need * (1.0 - p_acc) / (p_acc * p_acc) + 4.0 * (need - k) / (p_acc * p_acc) + need / p_acc;
            }
            const double m = k * q_nsn[s];                                                 // randint draws: at most one per non-SNP
            e_ch += m; var_ch += m * (1.0 - q_nsn[s]);
            e_words += m / acc_min[s];
            var += m / (acc_min[s] * acc_min[s]);
            if (ms.has_tl) { e_words += 5.0 * k * q_tl[s]; var += 25.0 * k * q_tl[s]; }   // __link_tls: deletions, shuffle, one coin
                                                                                            // per pair, <= 2 words per draw on average
        }
    }
    const double wd = e_words + 16.0 * std::sqrt(var) + 65536.0;
    if (wd >= 1.0e9) return fail(c, MSIM_ERR_UNSUPPORTED, "host-chain window beyond 2^30 words");
    const uint32_t W = (uint32_t)wd;
    const uint32_t n_hi = ms.sn_chained ? K : (uint32_t)std::min<double>(K, e_ch + 16.0 * std::sqrt(var_ch) + 64.0);
    bool grew = false;
    MixedSet &M = g->mixed[g->mixed_unit++ % N_SETS];
    if ((rc = wait_if_pending(c, M.pending, M.emit_done))) return rc;
    if (M.pending) {                                       // the pinned tables of this set may still be in flight
        MSIM_HIP(c, wait_event(M.emit_done));
        M.pending = false;
    }
    if (!M.emit_done) MSIM_HIP(c, hipEventCreateWithFlags(&M.emit_done, hipEventDisableTiming));
    const uint32_t nbk = (K + CB_BLOCK - 1) / CB_BLOCK;
    const size_t n_slots = ((size_t)W + 1) << lg;
    if ((rc = grow(c, (void **)&M.cand_pos, &M.cap_pos, (size_t)K * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.cand_type, &M.cap_type, (size_t)K + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.cand_stop, &M.cap_stop, (size_t)K * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.nsn_type, &M.cap_ntype, (size_t)K + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.nsn_rank, &M.cap_nrank, (size_t)K * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.nsn_stop, &M.cap_nstop, (size_t)K * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.sn_index, &M.cap_snidx, (size_t)K * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.cnt, &M.cap_cnt, mixed_cnt_bytes(nbk), &grew))) return rc;
    if ((rc = grow(c, (void **)&M.words, &M.cap_words, (size_t)W * 4 + 64, &grew))) return rc;
    if ((rc = grow(c, (void **)&M.tables, &M.cap_tables, n_slots * 4 + 64, &grew))) return rc;
    if (ms.has_tl) {
        if ((rc = grow(c, (void **)&M.cand_extra, &M.cap_cextra, (size_t)K * 4 + 64, &grew))) return rc;
        if ((rc = grow(c, (void **)&M.cand_aux, &M.cap_caux, (size_t)K + 64, &grew))) return rc;
        if ((rc = grow(c, (void **)&M.nsn_extra, &M.cap_nextra, (size_t)K * 4 + 64, &grew))) return rc;
        if ((rc = grow(c, (void **)&M.nsn_aux, &M.cap_naux, (size_t)K + 64, &grew))) return rc;
    }
    // range table | type tables | visit_from, one device block and one pinned block per scratch set
    const size_t off_sets = ((size_t)n_draw * sizeof(MixRangeDev) + 63) & ~(size_t)63;
    const size_t off_visit = (off_sets + (size_t)n_sets * sizeof(TypeTable) + 63) & ~(size_t)63;
    const size_t mm_bytes = off_visit + (size_t)n_draw * 4 + 64;
    if ((rc = grow(c, (void **)&M.mm_d, &M.cap_mm_d, mm_bytes, &grew))) return rc;
    if ((rc = grow_host(c, (void **)&M.mm_h, &M.cap_mm_h, mm_bytes))) return rc;
    if (g->cap_h_words < n_slots * 4 || g->cap_h_win < (size_t)W * 4 || g->cap_h_npos < (size_t)K * 4 + 64 ||
        g->cap_h_ntype < (size_t)K + 64 || g->cap_h_nstop < (size_t)K * 4 + 64 || g->cap_h_nrank < (size_t)K * 4 + 64 ||
        (ms.has_tl && (g->cap_h_nextra < (size_t)K * 4 + 64 || g->cap_h_naux < (size_t)K + 64))) {
        MSIM_HIP(c, wait_stream(c->stream));     // an earlier contig's copies may still use the old blocks
        if (g->copy_stream) MSIM_HIP(c, wait_stream(g->copy_stream));
        if ((rc = grow_host(c, (void **)&g->h_words, &g->cap_h_words, n_slots * 4))) return rc;
        if ((rc = grow_host(c, (void **)&g->h_win, &g->cap_h_win, (size_t)W * 4))) return rc;
        if ((rc = grow_host(c, (void **)&g->h_npos, &g->cap_h_npos, (size_t)K * 4 + 64))) return rc;
        if ((rc = grow_host(c, (void **)&g->h_ntype, &g->cap_h_ntype, (size_t)K + 64))) return rc;
        if ((rc = grow_host(c, (void **)&g->h_nstop, &g->cap_h_nstop, (size_t)K * 4 + 64))) return rc;
        if ((rc = grow_host(c, (void **)&g->h_nrank, &g->cap_h_nrank, (size_t)K * 4 + 64))) return rc;
        if (ms.has_tl) {
            if ((rc = grow_host(c, (void **)&g->h_nextra, &g->cap_h_nextra, (size_t)K * 4 + 64))) return rc;
            if ((rc = grow_host(c, (void **)&g->h_naux, &g->cap_h_naux, (size_t)K + 64))) return rc;
        }
    }
    MixRangeDev *rt_h = reinterpret_cast<MixRangeDev *>(M.mm_h);
    TypeTable *sets_h = reinterpret_cast<TypeTable *>(M.mm_h + off_sets);
    uint32_t *visit_h = reinterpret_cast<uint32_t *>(M.mm_h + off_visit);
    const MixRangeDev *rt_d = reinterpret_cast<const MixRangeDev *>(M.mm_d);
    const TypeTable *sets_d = reinterpret_cast<const TypeTable *>(M.mm_d + off_sets);
    uint32_t *visit_d = reinterpret_cast<uint32_t *>(M.mm_d + off_visit);
    {
        uint32_t di = 0, base = 0;
        for (int i = 0; i < n_ranges; i++) {
            const msim_range &r = ranges[i];
            if (r.k == 0) continue;
            rt_h[di] = MixRangeDev{base, (uint32_t)(r.stop + 1), ms.set_of[di], 0u};
            base += (uint32_t)r.k;
            di++;
        }
        for (uint32_t s = 0; s < n_sets; s++) {
            const msim_range &r = ranges[ms.rep[s]];
            TypeTable tt{};
            tt.n = (uint32_t)r.n_types;
            for (int j = 0; j < r.n_types; j++) { tt.thr[j] = r.cdf_thr[j]; tt.type[j] = (uint8_t)r.types[j]; }
            sets_h[s] = tt;
        }
    }
    // ---- 1. device: word window, accept tables, candidate types, the chain's candidates
    const uint64_t np_base = np.pos;                       // exact: the NumPy stream never rejects
    if ((rc = ensure_words(c, g, 0, py.pos + W + 1))) return rc;
    if ((rc = ensure_words(c, g, 1, np_base + 2ull * K + 1))) return rc;
    if ((rc = ensure_signals(c, g))) return rc;
    MSIM_HIP(c, hipMemcpyAsync(M.mm_d, M.mm_h, off_visit, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_types_multi, dim3(nbk), dim3(CB_THREADS), 0, c->stream, np.d_raw, (unsigned long long)np_base, K, rt_d,
                       n_draw, sets_d, M.cand_type);
    hipLaunchKernelGGL(k_nsn_count, dim3(nbk), dim3(CB_THREADS), 0, c->stream, M.cand_type, K, M.cnt, ms.sn_chained ? 1u : 0u);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, c->stream, M.cnt, nbk);
    hipLaunchKernelGGL(k_nsn_scatter, dim3(nbk), dim3(CB_THREADS), 0, c->stream, (const uint32_t *)nullptr, M.cand_type, K, M.cnt, nbk,
                       (uint32_t *)nullptr, M.nsn_type, M.nsn_rank, g->d_ps, ms.sn_chained ? 1u : 0u);
    MSIM_HIP(c, hipGetLastError());
    {   // the chain's candidates go over beside the plan stream, a 16-sigma bound of their number first
        hipEvent_t se = next_chain_event(g);
        MSIM_HIP(c, hipEventRecord(se, c->stream));
        MSIM_HIP(c, hipStreamWaitEvent(g->copy_stream, se, 0));
        MSIM_HIP(c, hipMemcpyAsync(g->h_nrank, M.nsn_rank, (size_t)n_hi * 4, hipMemcpyDeviceToHost, g->copy_stream));
        MSIM_HIP(c, hipMemcpyAsync(g->h_ntype, M.nsn_type, (size_t)n_hi, hipMemcpyDeviceToHost, g->copy_stream));
        MSIM_HIP(c, hipEventRecord(g->ev_cand, g->copy_stream));
    }
    hipLaunchKernelGGL(k_temper_window_ps, dim3((W + 255) / 256), dim3(256), 0, c->stream, py.d_raw, g->d_ps, W, M.words);
    hipLaunchKernelGGL(k_accept_tables_ps, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, c->stream, py.d_raw, g->d_ps, W,
                       ms.gcc, lg, M.tables);
    MSIM_HIP(c, hipGetLastError());
    // window and tables come over in three pieces (1/8, 3/8, 1/2); the host starts on the first while the others are in flight.
    // Their copies go out behind the mailbox kernel, before the host polls: nothing of them depends on what the poll tells.
    struct Feed { Ctx *c; GpuPlan *g; size_t cut[4]; int next; double wait_us; } fd{c, g, {0, (size_t)W / 8, (size_t)W / 2, (size_t)W}, 0, 0.0};
    PlanState h;
    rc = mixed_poll(c, g, h, [&]() -> int {                // exact stream position + the chain's length
        for (int q = 0; q < 3; q++) {
            const size_t a = fd.cut[q], b = fd.cut[q + 1], tb = q == 2 ? (size_t)W + 1 : b;   // (+ the end-of-window sentinel)
            if (b > a) MSIM_HIP(c, hipMemcpyAsync(g->h_win + a, M.words + a, (b - a) * 4, hipMemcpyDeviceToHost, c->stream));
            if (tb > a) MSIM_HIP(c, hipMemcpyAsync(g->h_words + (a << lg), M.tables + (a << lg), ((tb - a) << lg) * 4, hipMemcpyDeviceToHost, c->stream));
            MSIM_HIP(c, hipEventRecord(g->ev_piece[q], c->stream));
        }
        return MSIM_OK;
    });
    if (rc) return rc;
    const uint32_t n_ch = h.n_nsn;
    const uint64_t p0 = h.pos;
    np.pos = np_base + 2ull * K;
    c->t.np_words += 2ull * K;
    if (n_ch > n_hi) {                                     // beyond 16 sigma: the rest of the chain's candidates
        MSIM_HIP(c, hipMemcpyAsync(g->h_nrank + n_hi, M.nsn_rank + n_hi, (size_t)(n_ch - n_hi) * 4, hipMemcpyDeviceToHost, g->copy_stream));
        MSIM_HIP(c, hipMemcpyAsync(g->h_ntype + n_hi, M.nsn_type + n_hi, (size_t)(n_ch - n_hi), hipMemcpyDeviceToHost, g->copy_stream));
        MSIM_HIP(c, hipEventRecord(g->ev_cand, g->copy_stream));
    }
    MSIM_HIP(c, hipEventRecord(g->t1, c->stream));
    if (c->deferred_apply >= 0) MSIM_HIP(c, hipStreamWaitEvent(c->emit_stream, g->ev_cand, 0));   // (behind the candidates' copies too)
    if ((rc = flush_deferred_apply_behind(c, g->ev_piece[2]))) return rc;   // the previous contig's APPLY
    const auto tp1 = std::chrono::steady_clock::now();
    // ---- 2. the chain, on the host
    size_t consumed = 0;
    WordFeed feed;
    feed.user = &fd;
    feed.more = [](void *u, size_t *avail) -> int {
        Feed &f = *static_cast<Feed *>(u);
        if (f.next >= 3) return MSIM_OK;
        const auto w0 = std::chrono::steady_clock::now();
        const int rc = spin_event(f.c, f.g->ev_piece[f.next], "spin_event(word-window piece, plan stream)");
        f.wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        if (!rc) *avail = f.cut[++f.next];
        return rc;
    };
    rc = spin_event(c, g->ev_cand, "spin_event(candidates of the host chain, copy stream)");
    const auto mw0 = std::chrono::steady_clock::now();
    c->t.host_walk_wait_ms += std::chrono::duration<double, std::milli>(mw0 - tp1).count();
    if (!rc) rc = multimix_walk_host(c, ct.len, ranges, n_ranges, d, ms, g->h_win, g->h_words, W, g->h_nrank, g->h_ntype, n_ch,
                                     g->h_npos, g->h_nstop, visit_h, &consumed, &feed, ms.has_tl ? g->h_nextra : nullptr,
                                     ms.has_tl ? g->h_naux : nullptr);
    {
        const double all_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - mw0).count();
        c->t.host_walk_wait_ms += fd.wait_us * 1e-3;
        c->t.host_walk_run_ms += std::max(0.0, all_ms - fd.wait_us * 1e-3);
        c->t.host_walk_candidates += K;
    }
    if (!rc) {
        MSIM_HIP(c, wait_event(g->t1));           // all pieces landed (usually long ago): the pinned blocks are free again
        rc = span_close(c, g);
    }
    if (rc) { g->s[0].live = g->s[1].live = false; g->unverified = false; return rc; }
    const auto tp2 = std::chrono::steady_clock::now();
    MSIM_HIP(c, hipEventRecord(g->t0, c->stream));        // the host chain is not GPU time
    // ---- 3. back on the device
    MSIM_HIP(c, hipMemcpyAsync(M.cand_pos, g->h_npos, (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
    if (n_ch) MSIM_HIP(c, hipMemcpyAsync(M.nsn_stop, g->h_nstop, (size_t)n_ch * 4, hipMemcpyHostToDevice, c->stream));
    MSIM_HIP(c, hipMemcpyAsync(visit_d, visit_h, (size_t)n_draw * 4, hipMemcpyHostToDevice, c->stream));
    if (n_ch) hipLaunchKernelGGL(k_stop_scatter, dim3((n_ch + 255) / 256), dim3(256), 0, c->stream, M.nsn_rank, M.nsn_stop, n_ch, M.cand_stop);
    if (ms.has_tl) {                                       // what __link_tls decided: linked spans, flags, tombstones
        MSIM_HIP(c, hipMemsetAsync(M.cand_aux, 0, (size_t)K, c->stream));
        if (n_ch) {
            MSIM_HIP(c, hipMemcpyAsync(M.nsn_extra, g->h_nextra, (size_t)n_ch * 4, hipMemcpyHostToDevice, c->stream));
            MSIM_HIP(c, hipMemcpyAsync(M.nsn_aux, g->h_naux, (size_t)n_ch, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_link_scatter, dim3((n_ch + 255) / 256), dim3(256), 0, c->stream, M.nsn_rank, M.nsn_extra, M.nsn_aux,
                               n_ch, M.cand_extra, M.cand_aux);
        }
    }
    MSIM_HIP(c, hipGetLastError());
    rc = mixed_emit(c, g, ct, M, K, p0 + consumed, rt_d, n_draw, visit_d, ms.sn_chained, grew, ms.has_tl);
    if (prof) {
        const auto tp3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "hostchain: n_draw %u sets %u K %u chain %u W %u lg %u | pre %.0f walk %.0f (%.2f ns/cand) post %.0f us\n", n_draw, n_sets, K,
                n_ch, W, lg, us(tp0, tp1), us(tp1, tp2), us(tp1, tp2) * 1e3 / K, us(tp2, tp3));
    }
    return rc;
}

// test hook: which of the context's streams still have work queued (hipStreamQuery; never blocks)
void gpu_plan_stream_status(Ctx *c, GpuPlan *g, int out[8]) {
    auto q = [](hipStream_t s) { if (!s) return -1; const hipError_t e = hipStreamQuery(s); (void)hipGetLastError(); return e == hipSuccess ? 0 : e == hipErrorNotReady ? 1 : 2; };
    out[0] = q(c->stream); out[1] = q(c->emit_stream); out[2] = q(g->gen_stream); out[3] = q(g->jump_stream);
    out[4] = (int)g->s[0].n_chunks; out[5] = (int)g->s[0].n_states; out[6] = (int)g->s[0].ready_ev.size(); out[7] = (int)g->s[0].waited_chunks;
}

}  // namespace msim
