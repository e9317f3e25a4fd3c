// Internal state of a libmsim context (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/msim.h"
#include "mt19937.h"

namespace msim {

struct Contig {
    uint64_t len = 0;
    uint8_t *d_in = nullptr;          // allocation; the bases (uint8, upper-cased) start at d_in + PAD
    // plan result
    bool planned = false;
    bool plan_empty = true;
    bool all_snp = false;             // record table holds SNPs only: no length change, offset == pos
    bool delta_known = false;         // the planner already knows out_len - len (SV mixes planned on the device):
    long long known_delta = 0;        //   APPLY needs no round trip for the output size
    bool off_ready = false;           // ... and already left every record's output offset in d_off (k_emit_records)
    // counter-based engine (plan_fast.hip), types beyond SNPs: how many candidates survive is decided on the device and
    // nobody waits for it -- n_rec / pool_len / out_len are filled in when a synchronising call collects d_dyn
    const uint32_t *d_dyn = nullptr;  // {records, mutated length, insert pool bytes, flags} on the device; null: host-known sizes
    bool sizes_pending = false;       // d_dyn not collected yet
    bool dyn_applied = false;         // the APPLY in flight was sized by d_dyn (its byte counts are added when collected)
    uint64_t n_rec_cap = 0;           // candidates: the record table's allocation
    uint64_t out_cap_len = 0;         // 16-sigma bound of the mutated length: d_out's allocation and the APPLY grid
    uint64_t n_struct_est = 0;        // expected non-SNP candidates (selects the rewrite kernel's window)
    hipStream_t apply_stream = nullptr;   // the stream its PLAN ran on: its APPLY follows there (null: the context's emit stream)
    int32_t *d_first = nullptr;       // tile index of its own (contigs on different streams cannot share the context's scratch)
    size_t cap_first = 0;
    bool tile_index_done = false;     // (transient) apply_batch_device has already launched this contig's tile index
    bool tile_index_by_plan = false;  // (one shot) the SNP sampler's expansion is writing d_first on the contig's stream
                                      // (apply_prepare_tile_index): apply_batch_device launches no tile-index job for it
    bool timing_shared = false;       // its rewrite ran inside another contig's batched launch: that contig's events time it
    uint64_t n_rec = 0, pool_len = 0;
    msim_record *d_recs = nullptr;    // sorted, visited-only records
    uint8_t *d_pool = nullptr;        // allocation; insert bases start at d_pool + PAD
    size_t cap_recs = 0, cap_pool = 0, cap_out = 0, cap_off = 0;   // bytes kept allocated across plans
    // apply result
    bool applied = false;
    uint64_t out_len = 0;
    uint8_t *d_out = nullptr;
    uint32_t *d_off = nullptr;        // output offset of every record
    uint8_t key_base = 0;             // KeyError report
    uint64_t key_pos = 0;
    bool key_error = false;
    bool key_reported = false;
    int index = 0;                    // position in Ctx::contigs (selects the error word)
    bool apply_pending = false;       // APPLY enqueued, result not yet collected
    bool defer_apply = false;         // planned by an engine with a host chain: its APPLY may wait for the next plan's chain
    hipEvent_t ea0 = nullptr, ea1 = nullptr, ea2 = nullptr;   // APPLY timing (emit stream)
    bool ea0_is_ea1 = false;          // nothing ran between the APPLY's start and its rewrite launch (tile index made by PLAN): ea0 was
                                      // not recorded -- one packet less on the queue -- and ea1 stands for it
    // host-only context (device_id -1): the record table stays here
    std::vector<msim_record> h_recs;
    std::vector<uint8_t> h_pool;
};

constexpr uint64_t PAD = 64;          // slack after every byte buffer so 16-B vector accesses stay in bounds

struct GpuPlan;
struct FastPlan;                      // plan_fast.hip: the counter-based PLAN engine's streams + scratch
struct Comm;                          // comm.cpp: RCCL communicator + receive buffers of the gather
struct Batch;                         // msim_api.hip: state of msim_batch_run
struct FileIo;                        // file_io.hip: the output channels (thread + stream + pinned ring per output file)

struct Ctx {
    int device = 0;
    bool host_only = false;           // msim_create(-1): PLAN + text rendering only, no GPU touched
    bool chain_only = false;          // msim_plan_chain in progress: engines advance the streams and emit nothing
    uint32_t flags = 0;
    hipStream_t stream = nullptr;         // plan chain, uploads
    hipStream_t emit_stream = nullptr;    // record emission + APPLY (overlaps the next contig's chain)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    std::string err;
    std::string devname;
    HostMT py, np;                    // stream states, host representation
    GpuPlan *gpu = nullptr;           // device representation of the streams + sampler scratch
    Comm *comm = nullptr;             // multi-GPU: set by msim_comm_init
    void *rw_collect = nullptr;       // apply.hip: where apply_contig_device hands rewrite launches over while apply_batch_device runs
    Batch *batch = nullptr;           // last batch of small contigs (host buffers)
    FileIo *file_io = nullptr;        // file_io.hip (made on first use)
    msim_params params{};
    bool have_params = false;
    std::vector<Contig> contigs;
    int deferred_apply = -1;              // the contig whose APPLY msim_apply_contig deferred last (msim_api.hip), -1: none
    int deferred_more[3] = {-1, -1, -1};  //   ... and those deferred before it, oldest first: host-chain contigs are applied in
    int n_deferred_more = 0;              //   GROUPS (one tile-index launch, one rewrite launch: a 60 us kernel's ramp, tail and
                                          //   the gap to the next launch are a tenth of it); only_groups leaves an incomplete one waiting
    uint64_t fast_key = 0x9E3779B97F4A7C15ull;   // MSIM_RNG_FAST: Philox key (msim_set_fast_key) ...
    uint32_t fast_seq = 0;                //   ... and the ordinal of the next contig planned or walked past
    FastPlan *fast = nullptr;             //   ... and the engine's streams + scratch (plan_fast.hip)
    msim_timing t{};
    // device scratch
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    unsigned long long *h_mail = nullptr;   // pinned, device-visible mailbox for small results
    unsigned long long *d_errs = nullptr;   // one KeyError word per contig
    unsigned long long *h_errs = nullptr;   // pinned copy of them (apply_finish)
    size_t cap_h_errs = 0;
    std::vector<int> pending_apply;
    // text on the device (text_gpu.hip): rendered VCF lines / framed FASTA body / raw FASTA body staging
    uint8_t *d_text = nullptr;
    size_t cap_text = 0;
    uint64_t text_len = 0;
    int text_contig = -1, text_kind = 0;  // what d_text currently holds: 1 VCF lines, 2 framed FASTA body
    uint32_t text_bpl = 0;
    uint8_t *d_text_scratch = nullptr;
    size_t cap_text_scratch = 0;
};

// roctx range (rocprofv3 --marker-trace shows PLAN / APPLY / text / gather per contig).  The marker library is
// dlopen()ed on first use (librocprofiler-sdk-roctx, else libroctx64); without it the ranges cost one branch.
struct TraceRange {
    explicit TraceRange(const char *name);
    ~TraceRange();
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
    bool on;
};

int fail(Ctx *c, int code, const std::string &msg);
int hip_fail(Ctx *c, hipError_t e, const char *what);

// Host waits for the device, with a deadline.  hipStreamSynchronize / hipEventSynchronize block for as long as the device
// takes -- for ever when a signal is lost -- and a caller blocked inside them cannot even be interrupted (the reference cannot
// hang: mutator.py:105-142 is a plain loop).  These poll instead (spinning for the first 2 ms: a poll also returns sooner than
// the blocking calls' wake-up, then sleeping 50 us between queries) and give up after MSIM_WAIT_TIMEOUT_S seconds (default
// 120, 0 = never; read when a wait leaves its spinning phase) with MSIM_WAIT_TIMED_OUT, which hip_fail words as
// "<call>(<stream or event>): no completion within N s" -- MSIM_ERR_HIP for the caller, the device work stays queued.
constexpr hipError_t MSIM_WAIT_TIMED_OUT = hipErrorLaunchTimeOut;
hipError_t wait_stream(hipStream_t s);
hipError_t wait_event(hipEvent_t ev);
double wait_limit_seconds();

#define MSIM_HIP(ctx, call)                                              \
    do {                                                                 \
        hipError_t e__ = (call);                                         \
        if (e__ != hipSuccess) return ::msim::hip_fail((ctx), e__, #call); \
    } while (0)

// msim_api.hip: enqueue the APPLY that msim_apply_contig deferred, if any (the engines with a host chain call it when
// their chain starts)
int flush_deferred_apply(Ctx *c, bool only_groups = false);
bool deferred_apply_holds(const Ctx *c, int contig);

// plan_host.cpp
struct HostPlan {
    std::vector<msim_record> recs;
    std::vector<uint8_t> pool;
    bool empty = true;
};
int plan_contig_host(Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges, HostPlan &out);
// The boundary pass (mutator.py:184-265) of ONE range restricted to its non-SNP candidates -- the only
// stage of an SV-mix plan that is a true sequential chain.  pos/type: the candidates in
// position order; words: tempered CPython-stream words from the current position on.  stop[j] receives
// Mutation.stop, or CHAIN_DROPPED for a candidate that is blocked / dropped.  Valid when
// block[SN] == min(block): an SNP then never blocks a successor, so SNPs cannot influence the chain.
constexpr uint32_t CHAIN_DROPPED = 0xffffffffu;
int sample_min_distance_host(Ctx *c, int64_t start, int64_t stop, int64_t k, int64_t d, int64_t setsize, int64_t *out);
int chain_boundary_host(Ctx *c, const msim_range &r, uint64_t L, const uint32_t *pos, const uint8_t *type, size_t n,
                        const uint32_t *words, size_t n_words, uint32_t *stop, size_t *consumed, size_t *kept,
                        long long *len_delta);

// The same walk over per-class "next accepted draw" tables built on the device (plan_kernels.h: k_accept_tables):
// a class is a distinct (getrandbits shift, randint width) pair among the range's IN/DE/DU/IV lengths, and
// T[(w << lg_rows) + class] = (words consumed from w up to and including the first accepted draw) << (24 + lg_rows)
// | value -- the top byte is the increment of the slot index w << lg_rows, so the walk never scales anything -- or 0
// where no draw is accepted within CHAIN_TABLE_REACH words / the end of the window; w = 0..n_words.
constexpr uint32_t CHAIN_TABLE_REACH = 63;                           // acceptance >= 1/2: 63 rejections in a row never happen
// Up to eight classes: five SV types with five different length widths on one contig take 8 slots per word position.  The
// slot increment of an entry is at most 63 << lg_rows: 8 bits above a 24-bit value up to four classes, 9 above 23 beyond.
struct ChainClasses { uint32_t n; uint32_t sh[8]; uint32_t width[8]; uint8_t cls_of[8]; };
inline uint32_t chain_lg_rows(const ChainClasses &cc) { return cc.n <= 1 ? 0u : (cc.n == 2 ? 1u : (cc.n <= 4 ? 2u : 3u)); }
constexpr uint32_t chain_value_bits(uint32_t lg_rows) { return lg_rows <= 2 ? 24u : 23u; }
bool chain_classes(const msim_range &r, ChainClasses &cc);          // false: some length does not fit the table entry
void accept_tables_host(const ChainClasses &cc, const uint32_t *words, size_t n_words, uint32_t *T);   // test support
// The walk is resumable, so that the host can start on the first piece of the table while the rest is still being
// copied: run() walks candidates [j, n) while the next position stays below w_lim (positions [0, w_lim) are valid).
// It writes stops only; the kept count and the length delta are summed where the stops are consumed (device:
// k_keep_flags; chain_boundary_tables: a pass of its own).  types_ok() is the argument check the loop leaves out.
struct ChainWalk {
    size_t j = 0, ws = 0, n_words = 0;                               // ws: next word position << lg_rows
    int64_t blk_hi = 0, bad = 0;                                     // last_mut_range = range(0)
    int64_t add[8], next_add[8], next_cap[8], clamp[8], drop_from[8], in_mask[8], draw_mask[8];
    size_t row[8];
    uint32_t lg_rows = 0, vbits = 24;                                // table entry = slot increment << vbits | value
    int init(Ctx *c, const msim_range &r, uint64_t L, const ChainClasses &cc, size_t n_words);
    static bool types_ok(const uint8_t *type, size_t n, bool with_tl = false);
    void run(const uint32_t *pos, const uint8_t *type, size_t n, const uint32_t *T, size_t w_lim, uint32_t *stop);
    void run_tl(const uint32_t *pos, const uint8_t *type, size_t n, const uint32_t *T, size_t w_lim, uint32_t *stop);   // + TL / TLI
    int finish(Ctx *c, size_t n, size_t *consumed) const;
};
int chain_boundary_tables(Ctx *c, const msim_range &r, uint64_t L, const uint32_t *pos, const uint8_t *type, size_t n,
                          const ChainClasses &cc, const uint32_t *T, size_t n_words, uint32_t *stop, size_t *consumed,
                          size_t *kept, long long *len_delta);

// The stream cuts of the same samples without the samples: cut[i] = first word of the i-th drawing range (k > 0),
// cut[number of drawing ranges] = words consumed; pool_pos: start + value of every pool-path draw (sum of k over the
// ranges with n <= setsize slots).  The device derives the set-path positions from the cuts (k_interval_bits).
// feed (optional): the window is still arriving -- more() blocks until further words are there and raises *avail.
struct WordFeed { int (*more)(void *user, size_t *avail); void *user; };
int cut_ranges_host(Ctx *c, const msim_range *ranges, int n_ranges, int64_t d, const uint32_t *words, size_t n_words,
                    uint32_t *cut, uint32_t *pool_pos, size_t *n_pool_pos, size_t *consumed, const WordFeed *feed = nullptr);

// random.sample() of every drawing range of a contig (util.py:94-109), reading tempered CPython-stream words
// from `words` instead of generating them: set path and pool path, exact word consumption.  Writes the
// candidate positions (start + value + d * rank, ascending per range) to pos_out.  For contigs with many
// small ranges (RMT mode), where a chain of thousands of data-dependent stream cuts leaves nothing to
// parallelise -- the device still generates the words and does all per-record work.
int sample_ranges_host(Ctx *c, const msim_range *ranges, int n_ranges, int64_t d, const uint32_t *words,
                       size_t n_words, uint32_t *pos_out, size_t *consumed);

// ---- general host-chain engine (plan_gpu.hip: plan_contig_gpu_multimix) ---------------------------------------
// Contigs whose drawing ranges differ in their settings (RMT files: gene blocks + an SV `std` line, hot / cold
// ranges with their own rates and lengths) or whose SNP block exceeds the sampling distance.  Per range the
// reference runs sample() -> type draw -> boundary pass (mutator.py:144-214) before it touches the next range, so
// the CPython stream interleaves samples and randint draws: one chain over all ranges, walked here over words,
// "next accepted draw" tables and candidate types the DEVICE produced.  Distinct MutationSettings are numbered
// (`sets`); the randint classes of all of them share one table (at most 4 classes).
struct MixSets {
    ChainClasses gcc{};                     // union of the (shift, width) classes of every set; cls_of unused
    std::vector<uint32_t> set_of;           // per DRAWING range (k > 0), in order: its set
    std::vector<int> rep;                   // per set: index into ranges[] of one range that uses it
    std::vector<ChainClasses> cc;           // per set: gcc with the set's own cls_of
    bool sn_chained = false;                // block[SN] != d: SNPs block their successors -> every candidate is on the chain
    bool has_tl = false;                    // some range draws TL / TLI: __link_tls runs behind the last range (host)
    uint64_t K = 0;                         // candidates of the contig
    uint32_t n_draw = 0;
};
// false: outside the engine (translocations, overlapping / unsorted ranges, ValueError cases, > 4 randint classes,
// lengths beyond a table entry, contig >= 2^31) -- the host planner decides
bool multimix_prepare(const Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges, MixSets &ms);
// The walk.  ch_rank / ch_type: candidate ordinal and type of every candidate ON THE CHAIN in ordinal order (the
// non-SNPs; all candidates when ms.sn_chained) -- types come from the NumPy stream by ordinal alone, so the device
// knows them before any position exists.  words: tempered CPython-stream words from the current position on;
// T: accept tables of ms.gcc over the same window (n_words + 1 entries << lg_rows).  Out: cand_pos[K] (every
// candidate, position order), ch_stop[n_ch] (Mutation.stop or CHAIN_DROPPED), visit_from[n_draw]: candidates of
// drawing range i below visit_from[i] lie inside a DE/DU/IV span of an EARLIER range (the blocked range is reset
// per range, mutator.py:184) and are never visited by __mutate_sequence (mutator.py:376,386,398).
// ms.has_tl (translocations): ch_extra / ch_aux (n_ch entries each) receive, per chain candidate, what __link_tls
// (mutator.py:267-316) decides behind the last range -- a linked TLI's source span start (its ch_stop becomes the TL's stop)
// and flags: bit 0 reversed copy, bit 1 trans_insert_pos > 0, bit 7 TOMBSTONE (deleted by __fix_tl_amount: it took part in
// the boundary pass -- its stop still blocks -- but is no record).
constexpr uint8_t CHAIN_TOMBSTONE = 0x80;
int multimix_walk_host(Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges, int64_t d, const MixSets &ms,
                       const uint32_t *words, const uint32_t *T, size_t n_words, const uint32_t *ch_rank,
                       const uint8_t *ch_type, size_t n_ch, uint32_t *cand_pos, uint32_t *ch_stop, uint32_t *visit_from,
                       size_t *consumed, const WordFeed *feed = nullptr, uint32_t *ch_extra = nullptr,
                       uint8_t *ch_aux = nullptr);
// __link_tls for one contig's chain candidates over a plain word window (the SV-mix engine: one range, the window fetched
// behind the boundary walk); *consumed = words drawn.  ch_extra / ch_aux are cleared first.
int link_translocations(Ctx *c, const uint32_t *words, size_t n_words, const uint32_t *ch_pos, const uint8_t *ch_type,
                        uint32_t *ch_stop, uint32_t *ch_extra, uint8_t *ch_aux, size_t n_ch, size_t *consumed);
void count_translocations(const uint8_t *type, const uint32_t *stop, size_t n, size_t *n_tl, size_t *n_tli);
// test support (msim_dbg_multimix_plan): the whole engine on the host -- the device's parts (types, tables, keep
// flags, records) restated sequentially -- so the CPU tier can hold the algorithm against plan_contig_host
int multimix_plan_emulated(Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges, HostPlan &out);

// plan_fast.hip: the counter-based PLAN engine (MSIM_RNG_FAST)
struct FastPlan;
void fast_plan_destroy(Ctx *c);
// MSIM_OK / MSIM_ERR_VALUE (the reference's ValueError: a sample larger than its population) / MSIM_ERR_UNSUPPORTED
int fast_plan_check(Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges);
int plan_contig_fast(Ctx *c, Contig &ct, const msim_range *ranges, int n_ranges, uint64_t key, uint32_t seq);
// plans queue up (plan_contig_fast) and go to the device as ONE batch: at the next entry point that needs a result
int fast_plan_flush(Ctx *c);
// is this contig's plan still queued?  mark_apply: its APPLY is enqueued right behind the batch (msim_apply_contig)
bool fast_plan_queued(Ctx *c, int contig, bool mark_apply);
// everything the engine enqueued has completed (the caller synchronised): sticky flags, sizes of the contigs planned with
// device-side counts.  Idempotent.
// behind (optional): device work of the caller's that belongs behind everything the lanes have in flight -- enqueued on the
// emit stream once that stream waits for the lanes, so that ONE host wait covers both (apply_finish: the KeyError words).
// *behind_state: 0 it did not run (nothing was pending), 1 it ran and what it copied is complete, 2 it ran, but a plan was
// replayed (and applied again) afterwards -- stale
int fast_plan_collect(Ctx *c, int (*behind)(Ctx *) = nullptr, int *behind_state = nullptr);
// is `s` a lane whose set is pending -- one the next fast_plan_collect makes the emit stream wait for?
bool fast_lane_joined_at_collect(const Ctx *c, hipStream_t s);
// test support (msim_dbg_fast_plan): the engine restated sequentially on the host over the same arithmetic (fast_math.h)
int fast_plan_emulated(Ctx *c, uint64_t L, const msim_range *ranges, int n_ranges, uint64_t key, uint32_t seq, HostPlan &out);

// text_gpu.hip
// (buf / cap: render there instead of into the context's text buffer -- an output channel's; text_len is then left alone)
int vcf_render_device(Ctx *c, Contig &g, const char *seq_name, uint64_t *bytes, uint8_t **buf = nullptr, size_t *cap = nullptr);
int fasta_frame_device(Ctx *c, Contig &g, uint32_t bpl, uint64_t *bytes, uint8_t **buf = nullptr, size_t *cap = nullptr);
int splice_device(Ctx *c, const Contig &a, const Contig *b, const uint32_t *seg_out, const uint32_t *seg_src, uint32_t n_seg,
                  Contig &dst);
int fasta_gather_device(Ctx *c, const uint8_t *body, uint64_t body_bytes, uint64_t n_bases, uint32_t lenc,
                        uint32_t lenb, uint8_t *d_dst);

// file_io.hip: device text -> output files on the channels' own threads (0 the Fasta, 1 the VCF)
int file_check(Ctx *c, int fd);
int file_text_buffer(Ctx *c, int ch, int *slot, uint8_t ***buf, size_t **cap);
int file_enqueue(Ctx *c, int ch, int slot, uint64_t n, int fd, uint64_t offset);
int file_enqueue_host(Ctx *c, int ch, const uint8_t *src, uint64_t n, int fd, uint64_t offset);
void file_channel_idle(Ctx *c, int ch);
int file_wait(Ctx *c);
void file_io_destroy(Ctx *c);
void device_host_cpus(int device, char *buf, size_t cap);   // cpulist of the device's NUMA node ("" if unknown)

// render.cpp
uint64_t render_vcf_unchecked(const msim_record *recs, uint64_t n_records, const uint8_t *insert_pool, const uint8_t *bases,
                              uint64_t len, const char *seq_name, char *out);

// comm.cpp
void comm_destroy(Ctx *c);

// apply.hip
int apply_contig_device(Ctx *c, Contig &g);
// contigs of one batch of the counter-based engine / one emission group of the SNP sampler: ONE tile-index launch for all of them;
// batch_rewrites: their rewrite kernels as one launch per kernel variant too (the counter-based engine: 1.84 -> 1.63 ms per c2
// step, 2.87 -> 2.55 for c3; the SNP sampler's groups of three: the same step time within the noise, 4.19-4.24 vs 4.23-4.39 ms,
// and the three-contig rewrite kernel runs at 0.76 of the HBM peak inside the pipeline instead of 0.67 -- fewer ramps and tails)
int apply_batch_device(Ctx *c, const std::vector<int> &ids, bool batch_rewrites);
int apply_finish(Ctx *c);             // collect results of asynchronous APPLYs (timing, KeyError words)
// SNP sampler (plan_gpu.hip: gpu_emit_flush): the expansion kernel writes an SNP-only contig's tile index itself -- room for it,
// its geometry (tile = 1 << *tile_shift output bytes, *n_tiles of them; n_tiles + 1 entries) and the contig's KeyError word
int apply_prepare_tile_index(Ctx *c, Contig &g, hipStream_t st, int32_t **first, uint32_t *n_tiles, uint32_t *tile_shift,
                             unsigned long long **err);
constexpr int MAX_CONTIGS = 1 << 16;
int synth_contig_device(Ctx *c, uint8_t *d_dst, uint64_t len, uint64_t seed);
int checksum_device(Ctx *c, const uint8_t *d_src, uint64_t len, uint64_t *sum);
int ensure_scratch(Ctx *c, size_t bytes);
// grow-only device buffer (contents are NOT preserved)
int dev_reserve(Ctx *c, void **p, size_t *cap, size_t want_bytes);

}  // namespace msim
