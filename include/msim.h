/*
 * msim.h -- C-ABI of libmsim.so: the MI355X-native mutation-injection path.
 *
 * The reference (mkpython3/Mutation-Simulator 3.0.2) is pure Python and has NO plugin / FFI /
 * native boundary of its own: the hot path sits behind the Python class `Mutator`
 * (mutation_simulator/mutator.py:73-479, called from __main__.py:80-85).  This header is the
 * boundary a maintainer would bind with ctypes to replace that class's internals; each entry
 * point cites the reference code whose work it takes over (paths relative to
 * /root/reference/mutation_simulator/).  INTEGRATION.md shows the ctypes stub.
 *
 * Contract: plain C, opaque handle, caller-owned host buffers, int return codes (0 = ok) with
 * msim_last_error() for the text; one context per process/GPU; like the reference (a single
 * thread driving two global RNGs) a context is NOT thread-safe.  No Python, torch or HIP types
 * cross this line.  Everything floating point on the path (rate sums -> k, chances -> cdf, titv ->
 * p_ti) is evaluated by the caller with the reference's own Python expressions and arrives here
 * as integers (see msim_range / msim_params), so the device work is integer/byte only.
 *
 * Two phases -- possible because RNG use never depends on genome content:
 *   PLAN   two MT19937 streams -> a position-sorted table of 16-byte mutation records per contig
 *   APPLY  records + uint8 genome in HBM -> mutated uint8 stream in HBM (HBM-bandwidth bound)
 */
#ifndef MSIM_H
#define MSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSIM_ABI_VERSION 8

/* ---- return codes ---------------------------------------------------------------------------- */
#define MSIM_OK               0
#define MSIM_ERR_ARG          1   /* bad argument / call order                                      */
#define MSIM_ERR_HIP          2   /* HIP runtime failure (no device, OOM, launch error)             */
#define MSIM_ERR_VALUE        3   /* reference raises ValueError("Sample larger than population or
                                     is negative") -- util.py:104 via random.sample                */
#define MSIM_ERR_KEY          4   /* reference raises KeyError(base): transversion of a base outside
                                     A,G,T,C,N -- mutator.py:449-455; see msim_key_error()         */
#define MSIM_ERR_UNSUPPORTED  5   /* valid for the reference, outside this build: contigs (or their mutated
                                     form) of 4 GiB and more, 2^31 mutations on one contig, an output file
                                     that cannot be written at an offset, MSIM_PLAN_GPU forced on a contig
                                     no device engine takes (README "Limits")                          */
#define MSIM_ERR_NOMEM        6
#define MSIM_ERR_IO           7   /* a write queued by the *_file entry points failed (msim_last_error)   */

/* ---- mutation types: numerically identical to mut_types.py:6-12 --------------------------------- */
#define MSIM_SN  1
#define MSIM_IN  2
#define MSIM_DE  3
#define MSIM_DU  4
#define MSIM_IV  5
#define MSIM_TL  6
#define MSIM_TLI 7

/* ---- msim_create flags --------------------------------------------------------------------- */
#define MSIM_PLAN_AUTO   0u      /* per contig: a device PLAN engine where one applies, else the host planner */
#define MSIM_PLAN_HOST   1u      /* force the sequential host planner (cross-check / debugging)               */
#define MSIM_PLAN_GPU    2u      /* force a device engine; MSIM_ERR_UNSUPPORTED where none can run             */
#define MSIM_RNG_FAST    4u      /* NOT stream-compatible with the reference: a counter-based generator (Philox4x32-10) replaces
                                    the two MT19937 streams, so that no draw depends on another one, PLAN has no sequential chain
                                    and the host does nothing (csrc/plan_fast.hip, fast_math.h).  The CONSTRUCTION is the
                                    reference's -- k = int(len * rate) candidates per range as a uniform k-subset with the minimum
                                    distance (util.py:93-109), a type per candidate from the range's chances, a uniform length,
                                    IV dropped / DU, DE clamped at the contig end, the boundary pass with its blocked ranges reset
                                    per range (mutator.py:144-265), the rewrite's visit rule, transition with probability p_ti,
                                    insert bases uniform in ATGC -- same distributions, different numbers.  SN / IN / DE / DU / IV
                                    on any number of sorted, non-overlapping ranges; translocations and overlapping RMT ranges:
                                    MSIM_ERR_UNSUPPORTED; a sample larger than its population: MSIM_ERR_VALUE like the reference.
                                    msim_seed / msim_set_mt_state are ignored, msim_set_fast_key seeds it.  msim_plan_contig only
                                    QUEUES the contig; everything queued goes to the device as one batch (one launch per stage over
                                    all contigs) at the next call that needs a result -- plan a whole genome, then ask.        */
/* Device PLAN engines: the device owns both MT19937 streams and does all per-record work;
 * SNP-only large ranges need nothing from the host, SV mixes hand the boundary chain over their non-SNP
 * candidates to the host, contigs with many small SNP ranges hand it the chain of sample() calls, contigs whose
 * ranges have their own SV settings (or whose SNPs block) hand it both -- always over words the device generated.
 * Results are bit-identical whichever engine runs; msim_timing.contigs_* tells which one did.                  */

typedef struct msim_ctx msim_ctx;

/* One mutation -- the reference's `Mutation` object (mutator.py:26-50) in 16 bytes.
 * Positions are 0-based inclusive like the reference's.  Tables are sorted by pos and hold only
 * the mutations __mutate_sequence actually visits (entries swallowed by an earlier DE/IV/DU span,
 * mutator.py:376/386/398, are dropped: they draw no random numbers and emit nothing).            */
typedef struct msim_record {
    uint32_t pos;     /* Mutation.start (dict key)                                               */
    uint32_t stop;    /* Mutation.stop; for IN: pos + insert_len - 1 (mutator.py:344)            */
    uint32_t extra;   /* IN: offset of the insert's bases in the contig's insert pool;
                         TLI: Mutation.start of the linked TL span (stop = its Mutation.stop)        */
    uint8_t  type;    /* MSIM_SN .. MSIM_TLI                                                     */
    uint8_t  aux;     /* SN: 0 = transition, 1/2 = transversion table column 0/1 (mutator.py:428-463);
                         TLI: bit 0 = trans_reverse, bit 1 = trans_insert_pos > 0 (mutator.py:281-284) */
    uint16_t rsv;
} msim_record;

/* One RangeDefinition that has mutations (rmt.py:166-189 + MutationSettings rmt.py:79-163),
 * reduced to the integers __get_mutations (mutator.py:144-214) needs.                            */
typedef struct msim_range {
    int64_t  start, stop;     /* 0-based inclusive                                                */
    int64_t  k;               /* int(((stop-start)+1) * sum(rates))           mutator.py:225       */
    int64_t  setsize;         /* CPython sample(): 21 + 4**ceil(log(3k,4)) if k>5 else 21          */
    int32_t  n_types;         /* len(mut_chances), dict order                 mutator.py:171-173   */
    int32_t  types[8];        /* MSIM_* ids in that order                                          */
    uint64_t cdf_thr[8];      /* ceil(cdf_j * 2^53), cdf = cumsum(p)/cumsum(p)[-1]: the type drawn
                                 for a 53-bit NumPy sample m is types[#{j: cdf_thr[j] <= m}]       */
    int64_t  min_len[8];      /* mut_lengs["min"/"max"], indexed by MSIM_* id                      */
    int64_t  max_len[8];
} msim_range;

/* SimulationSettings.mut_block / .titv (rmt.py:326-352) */
typedef struct msim_params {
    int64_t  block[8];        /* indexed by MSIM_* id; min over ids 1..7 is the sampling distance d */
    uint64_t ti_lim;          /* floor(p_ti * 2^53) + 1, p_ti = titv*(1/(titv+1)); 0 if p_ti is NaN:
                                 transition iff the 53-bit sample m < ti_lim  (p <= p_ti,
                                 mutator.py:436-438)                                               */
} msim_params;

typedef struct msim_timing {  /* milliseconds; device stages are HIP-event times on the ctx stream */
    double plan_host_ms;      /* host planner wall time (sequential RNG chain)                      */
    double plan_gpu_ms;       /* GPU sampler kernels                                                */
    double upload_ms;         /* record table H2D                                                   */
    double apply_ms;          /* APPLY kernels (scan + tile index + rewrite): the time during which */
    double apply_kernel_ms;   /* at least one was in flight; the rewrite kernel alone (roofline     */
                              /* kernel), likewise -- launches on different streams may overlap     */
    uint64_t apply_launches;  /* rewrite-kernel launches accumulated since msim_reset_stats         */
    uint64_t bytes_in, bytes_out, records;   /* algorithmic traffic of those launches               */
    uint64_t py_words, np_words;             /* MT19937 words consumed from each stream             */
    /* contigs planned per PLAN engine since msim_reset_stats: which engine a run went through   */
    uint64_t contigs_snp;        /* SNP sampler: everything on the device                                              */
    uint64_t contigs_svmix;      /* one large SV-mix range: the boundary chain on the host                             */
    uint64_t contigs_hostcut;    /* many deterministic-SNP ranges: the stream cuts on the host                         */
    uint64_t contigs_hostchain;  /* several ranges with their own settings / SNPs that block: samples + chain on the host */
    uint64_t contigs_host;       /* sequential host planner (translocations, overlapping ranges, tiny contigs)         */
    uint64_t contigs_batch;      /* contigs that went through msim_batch_run (host planner, one APPLY per batch)       */
    uint64_t contigs_fast;       /* MSIM_RNG_FAST contexts: contigs planned with the counter-based generator            */
    uint64_t stream_rebases;     /* device MT19937 sessions re-based: the jump tables span 8192 chunks (1.31 G words) from a
                                    session's origin; before a contig that would not fit, the state at the streams' exact
                                    positions becomes the next session's origin (no limit on a run's stream length)     */
    uint64_t snp_samples_ahead;  /* SNP sampler: samples -- of owned contigs and of contigs walked for their stream positions
                                    (msim_plan_chain) alike -- whose count / scatter / de-dup ran off the stream-position chain,
                                    on a window anchored at the host's bound of the start (every
                                    context's default since round 6, MSIM_AHEAD=1: sharded ranks only, MSIM_NO_AHEAD: never) */
    /* The host-sequential stages of the bit-compatible engines, apart from what surrounds them (round 6, ABI 8): a slower
       host core shows in host_walk_run_ms per item, a slower link or device in host_walk_wait_ms -- so that "slower box" and
       "slower code" can be told apart from one bench line.                                                              */
    double host_walk_run_ms;     /* the host walking: boundary chains (SV mix, host chain), stream cuts (host cut)        */
    double host_walk_wait_ms;    /* ... and waiting: for the mailbox, for the pieces of its word window / candidates      */
    uint64_t host_walk_candidates;  /* candidates those chains walked (SV mix: the non-SNP ones; host chain: all)        */
    uint64_t host_cut_words;     /* stream words the host-cut engine's cuts went through                                  */
    uint64_t snp_ahead_margin_permille;  /* anchored windows: the largest |exact start - expected start| of a sample planned
                                    ahead of the chain, in permille of the deviation the host allowed for (8 sigma + 256 words:
                                    1000 = the soft edge; beyond: MSIM_ERR_HIP, the caller re-plans) -- the maximum since
                                    msim_reset_stats / the session's start.  Telemetry of the moment model behind the windows */
} msim_timing;

/* ---- lifetime -------------------------------------------------------------------------------- */
int  msim_abi_version(void);
/* Initialise the HIP runtime for `device_id` (idempotent, thread-safe, no context): the first HIP call of a process costs
 * ~0.2 s; a caller with host work of its own to do first (reading and indexing the FASTA) runs this beside it.        */
int  msim_warm_up(int device_id);
/* The CPUs of the NUMA node the device hangs on, as a Linux cpulist ("64-127,192-255"; "" when the host has one node or sysfs
 * does not tell).  On a two-socket host a run whose host threads stay on the GPU's socket moves its bytes once across the
 * inter-socket links instead of twice (CLI end to end: 0.30 -> 0.25-0.27 s); libmsim pins its own output channels' threads
 * there (MSIM_IO_CPUS=none / a cpulist overrides), the caller decides about its own.  Initialises the HIP runtime.
 * It matters for the launching thread too: the queues' packets live in its node's memory -- a -sn 0.01 step over a 3 Gb
 * genome (~160 dependent launches) takes 4.2 ms from the GPU's socket and 4.4 ms from the other one.                   */
int  msim_device_host_cpus(int device_id, char *cpulist, int cap);
int  msim_create(int device_id, uint32_t flags, msim_ctx **out);     /* Mutator.__init__ mutator.py:79 */
void msim_destroy(msim_ctx *ctx);                                     /* Mutator.close    mutator.py:95 */
const char *msim_last_error(const msim_ctx *ctx);                     /* "" when none; ctx may be NULL  */
int  msim_device_name(const msim_ctx *ctx, char *dst, int cap);
int  msim_sync(msim_ctx *ctx);
/* Change the PLAN mode of a live context (0 = AUTO, MSIM_PLAN_HOST, MSIM_PLAN_GPU).  The host package uses it to
 * re-plan a contig through the sequential host planner when a device engine reports that a stream window
 * overflowed its margin -- 16 sigma; 8 for the interval in which a sample planned ahead of the stream-position chain
 * (msim_timing.snp_samples_ahead) expects its start -- (the streams are put back with msim_set_mt_state first). */
int  msim_set_plan_mode(msim_ctx *ctx, uint32_t mode);

/* ---- the two global RNGs the reference draws from ---------------------------------------------- */
/* random.seed(int) == init_by_array(little-endian 32-bit limbs of abs(seed));
 * numpy.random.seed(int) == init_genrand(seed).  (mutator.py:4,8; util.py:5)                      */
int msim_seed(msim_ctx *ctx, const uint32_t *py_key, int n_key, uint32_t np_seed);
/* Hand over / read back raw MT19937 states (random.getstate()[1], numpy.random.get_state()[1:3]) so a
 * caller that seeded the Python generators itself gets the reference's exact stream, and can put the
 * advanced state back afterwards.  stream: 0 = CPython `random`, 1 = numpy.random.                 */
int msim_set_mt_state(msim_ctx *ctx, int stream, const uint32_t mt[624], int pos);
int msim_get_mt_state(msim_ctx *ctx, int stream, uint32_t mt[624], int *pos);

/* Optional sizing hint for the GPU sampler: how many words of each stream the coming plan calls
 * will roughly consume, so stream chunks are generated in one batch.  Never changes results.      */
int msim_reserve_streams(msim_ctx *ctx, uint64_t py_words, uint64_t np_words);
/* MSIM_RNG_FAST contexts: the generator's key; also restarts the contig ordinal that every msim_plan_contig /
 * msim_plan_chain call advances (so ranks that skip contigs they do not own stay aligned with those that plan them). */
int msim_set_fast_key(msim_ctx *ctx, uint64_t key);

/* ---- genome in HBM ----------------------------------------------------------------------------- */
/* Upload one contig (upper-cased bases, what pyfaidx hands the reference: util.py:84-88).
 * Contigs are numbered in call order like fasta[i].                                                */
int msim_add_contig(msim_ctx *ctx, const uint8_t *bases_upper, uint64_t len, int *contig);
/* Synthesize i.i.d. uniform A/C/G/T directly in HBM, 32 bases per 64-bit hash:
 *   base(i) = "ACGT"[(mix64(seed + (i >> 5)) >> (2 * (i & 31))) & 3],   mix64 = the splitmix64 finaliser
 * (k_synth in csrc/apply.hip; host twin synth_host in tests/test_gpu_parity.py).  Benchmark input.          */
int msim_add_contig_synthetic(msim_ctx *ctx, uint64_t len, uint64_t seed, int *contig);
int msim_contig_length(msim_ctx *ctx, int contig, uint64_t *len);
int msim_read_contig(msim_ctx *ctx, int contig, uint64_t offset, uint64_t n, uint8_t *dst);
int msim_clear(msim_ctx *ctx);            /* drop all contigs and results, keep RNG states          */

/* ---- PLAN: Mutator.__get_mutations + the RNG draws of __mutate_sequence -------------------------- */
int msim_set_params(msim_ctx *ctx, const msim_params *params);
/* One iteration of mutate()'s contig loop up to the rewrite (mutator.py:111-131 + the SNP/insert
 * draws of :334-358 in position order).  Consumes both streams exactly as the reference does and
 * leaves the contig's record table (+ insert pool) in HBM.  Contigs must be planned in index order
 * -- the streams are chained across contigs.                                                       */
int msim_plan_contig(msim_ctx *ctx, int contig, const msim_range *ranges, int n_ranges);
/* The same iteration for a contig this process does NOT mutate (multi-GPU runs: a rank that does not own a contig):
 * both streams advance exactly as msim_plan_contig on a contig of `len` bases would leave them -- the chain across
 * contigs stays intact -- but no record table, insert pool or SNP outcome is produced and no contig is created.   */
int msim_plan_chain(msim_ctx *ctx, uint64_t len, const msim_range *ranges, int n_ranges);
/* 1 if the last plan of this contig left `muts` empty (warning at mutator.py:125-129).             */
int msim_plan_was_empty(msim_ctx *ctx, int contig, int *empty);

/* ---- settings -> msim_range tables, natively ------------------------------------------------------ */
/* One MutationSettings object (rmt.py:79-163) as the reference holds it: floats stay floats until here.  */
typedef struct msim_settings_desc {
    double   rate_sum;        /* sum(mut_rates.values()) -- Python's left-to-right float sum, evaluated by the caller (one number) */
    int32_t  n_types;         /* len(mut_chances), dict order                                       */
    int32_t  types[8];        /* MSIM_* ids in that order                                            */
    double   chances[8];      /* mut_chances.values()                                                */
    int64_t  min_len[8];      /* mut_lengs["min"/"max"], indexed by MSIM_* id                        */
    int64_t  max_len[8];
} msim_settings_desc;
/* The msim_range table of a contig from its ranges' (start, stop, settings index) triples: what mutator.py:157-174,225 and
 * CPython's sample() derive per range, with the same IEEE operations --
 *   k        = (int64) ((double)(stop - start + 1) * rate_sum)                  int(((stop - start) + 1) * sum(rates))
 *   setsize  = 21 + 4^ceil(log4(3 k)) for k > 5, else 21, in integers (3 k is never a power of 4; exact for k < 2^33)
 *   cdf_thr  = ceil(cdf_j * 2^53), cdf = cumsum(chances) / cumsum(chances)[-1]  numpy.random.choice(p=...)
 * An RMT file in the style of the reference's examples gives a genome 35 000 ranges over three settings objects: this is
 * their marshalling, off the Python interpreter.  Pure host code, no context.  MSIM_ERR_ARG: a settings index out of range.  */
int msim_build_ranges(const msim_settings_desc *sets, int n_sets, const int64_t *start, const int64_t *stop,
                      const int32_t *set_id, int64_t n, msim_range *out);

/* ---- APPLY: Mutator.__mutate_sequence (mutator.py:318-426) -------------------------------------- */
/* Execution model: msim_plan_contig (SNP sampler engine) and msim_apply_contig (device-planned tables) only
 * ENQUEUE work -- the chain that fixes stream positions on one HIP stream, record emission and the rewrite
 * kernel on another, overlapping the next contig's chain.  (The SV-mix and host-cut engines wait inside
 * msim_plan_contig for the device data their host chain reads -- the boundary walk, the stream cuts.)  Deferred outcomes (the reference's KeyError, an
 * internal window overflow) are reported by the next call that synchronises: msim_sync,
 * msim_result_sizes(out_len), msim_fetch_*, msim_result_checksum, msim_get_mt_state, msim_stats, the text calls.
 * Enqueued does not mean launched at once: contigs the SNP sampler planned from ONE drawing range (ARGS mode) gather in groups
 * (fours by default) whose emission stages and rewrite go to the device as one launch each; the APPLYs of contigs
 * planned by an engine with a host chain wait for the walks of the contigs behind them and go out three at a time --
 * when the next msim_plan_contig arrives or at any other entry point, whichever comes first (plan + apply a whole genome,
 * then ask: that is the fast order; asking after every contig is as correct and launches per contig).                  */
int msim_apply_contig(msim_ctx *ctx, int contig);
/* If apply hit the reference's KeyError: the offending (ambiguity-converted) base and position.
 * contig == -1: the first contig (in index order) that hit it.                                    */
int msim_key_error(msim_ctx *ctx, int contig, uint8_t *base, uint64_t *pos);

int msim_result_sizes(msim_ctx *ctx, int contig, uint64_t *out_len, uint64_t *n_records,
                      uint64_t *insert_pool_len);
int msim_fetch_sequence(msim_ctx *ctx, int contig, uint64_t offset, uint64_t n, uint8_t *dst);
int msim_fetch_records(msim_ctx *ctx, int contig, msim_record *dst, uint8_t *insert_pool_dst);
/* 64-bit FNV-style checksum of the mutated stream computed on the device (parity at full size).   */
int msim_result_checksum(msim_ctx *ctx, int contig, uint64_t *sum);
int msim_release_result(msim_ctx *ctx, int contig);
/* Device address of the mutated stream (valid until the contig is re-applied / released / cleared), for
 * communication layers that move results GPU-to-GPU (RCCL gather of sharded contigs).
 * Synchronises like msim_fetch_sequence.                                                            */
int msim_result_device_ptr(msim_ctx *ctx, int contig, uint64_t *device_address, uint64_t *len);

/* ---- VCF text (vcf_writer.py:118-126 + record construction mutator.py:334-399) ------------------ */
/* Render the record lines of one contig.  Stateless host helper: `bases` is the INPUT contig.
 * Call with out == NULL to get the size.  Lines whose REF == ALT are suppressed like the reference. */
int msim_render_vcf(const msim_record *recs, uint64_t n_records, const uint8_t *insert_pool,
                    const uint8_t *bases, uint64_t len, const char *seq_name,
                    char *out, uint64_t cap, uint64_t *needed);

/* ---- text on the device ------------------------------------------------ */
/* The same record lines as msim_render_vcf, rendered by HIP kernels from the record table, insert pool and
 * input contig already in HBM (mutator.py:334-421 + vcf_writer.py:44-52,118-126).  Two-call protocol:
 * out == NULL renders into a device buffer and reports the size in *needed; a following call for the same
 * contig with out != NULL (cap >= *needed) copies the text to the host.                               */
int msim_render_vcf_device(msim_ctx *ctx, int contig, const char *seq_name, char *out, uint64_t cap,
                           uint64_t *needed);
/* The mutated contig as FASTA body text: '\n' after every `bpl` bases, none after a partial last line --
 * what FastaWriter.write emits between two headers (fasta_writer.py:40-58).  *needed = L + L / bpl.
 * Same two-call protocol.                                                                             */
int msim_fetch_sequence_framed(msim_ctx *ctx, int contig, uint32_t bpl, uint8_t *out, uint64_t cap,
                               uint64_t *needed);
/* The same two texts written into an open output file, off the calling thread: bytes [offset, offset + *written) of `fd`
 * (a regular file; libmsim keeps its own duplicate of the descriptor) receive what FastaWriter / VcfWriter would write()
 * there (fasta_writer.py:40-58, vcf_writer.py:118-126).  The call renders on the device, reports the size and QUEUES the
 * transfer on the output channel of that file kind (file_io.hip: a thread, a HIP stream and a pinned ring per channel --
 * device -> ring -> pwrite in 8 MiB pieces); it returns while the bytes are on their way, so that the caller's next contig
 * (ingest, PLAN, APPLY) overlaps them.  msim_file_wait (also msim_sync, msim_destroy) returns once everything queued is in
 * its file -- call it before the files are read, truncated or closed for good -- and reports the first failure of a queued
 * transfer (MSIM_ERR_IO: no space ...; MSIM_ERR_HIP).  MSIM_ERR_UNSUPPORTED: `fd` is no regular file (a pipe ...): use the
 * buffer calls above and write().  At most two transfers per channel are in flight: a third call waits for the oldest.     */
int msim_render_vcf_device_file(msim_ctx *ctx, int contig, const char *seq_name, int fd, uint64_t offset, uint64_t *written);
int msim_fetch_sequence_framed_file(msim_ctx *ctx, int contig, uint32_t bpl, int fd, uint64_t offset, uint64_t *written);
int msim_file_wait(msim_ctx *ctx);
/* Ingest one FASTA record straight from file text: `body` = the bytes after the header line, n_bases bases
 * in lines of `lenc` bases every `lenb` bytes (the .fai columns; uniform line width is what pyfaidx
 * requires, util.py:77-91).  Line terminators are skipped and a-z upper-cased on the device
 * (sequence_always_upper=True, util.py:84-88).  Replaces msim_add_contig + host-side parsing.           */
int msim_add_contig_text(msim_ctx *ctx, const uint8_t *body, uint64_t body_bytes, uint64_t n_bases,
                         uint32_t lenc, uint32_t lenb, int *contig);

/* Page-locked host memory for the caller's ingest / egress buffers: any host pointer is accepted by the text calls above,
 * but copies to and from page-locked memory run at the link's speed without a staging pass (and without the page faults
 * of a freshly allocated buffer) -- the CLI double-buffers its contigs through a few of these (mutator.py).            */
int msim_host_alloc(msim_ctx *ctx, uint64_t bytes, void **ptr);
int msim_host_free(msim_ctx *ctx, void *ptr);

/* ---- interchromosomal translocations (the reference's second pass, it_mutator.py) --------------------------------- */
/* __write_with_bp (it_mutator.py:121-146) for one contig: a NEW contig whose mutated stream is contig `a` cut at bp_a[0..n_bp)
 * and contig `b` cut at bp_b[0..n_bp), the segments taken alternately -- a's first, b's second, a's third ... (segment i =
 * bases [bp[i-1], bp[i]) with bp[-1] = 0 and bp[n_bp] = the contig's length; 0-based, as pairwise() over [0] + bp + [len]
 * cuts them).  The segments are read from the contigs' INPUT bases (the IT pass reads the file the mutation pass wrote).
 * n_bp = 0: a copy of a (__write_chrom_full, it_mutator.py:148-156; b is ignored).  The result counts as applied:
 * msim_fetch_sequence(_framed), msim_result_sizes and msim_result_checksum read it; it has no records.              */
int msim_splice_contigs(msim_ctx *ctx, int a, int b, uint64_t n_bp, const uint64_t *bp_a, const uint64_t *bp_b, int *contig);
/* sample_with_minimum_distance(start, stop, k, d) (util.py:93-109) on the context's CPython stream: the breakpoints of one
 * contig (it_mutator.py:96-119 calls it with (1, len, k, 1)).  setsize: CPython's 21 + 4^ceil(log4(3k)) for k > 5, else 21
 * (the caller evaluates the float expression, as for msim_range).  out: k ascending positions.  Works on a host-only
 * context.  MSIM_ERR_VALUE: "Sample larger than population or is negative" -- nothing was drawn.                       */
int msim_sample_min_distance(msim_ctx *ctx, int64_t start, int64_t stop, int64_t k, int64_t d, int64_t setsize, int64_t *out);

/* ---- many small contigs in one pass ------------------------------------------------------------------------------- */
/* mutate()'s loop body (mutator.py:111-141) for a run of SMALL contigs at once -- assemblies with thousands of scaffolds:
 * per contig the file text of the record (as for msim_add_contig_text), its ranges and its name.  The RNG streams are
 * consumed contig by contig exactly as by msim_plan_contig; ONE APPLY runs over the concatenation; the host frames the
 * FASTA bodies (fasta_writer.py:40-58) and renders the VCF lines (vcf_writer.py:118-126).  Results stay in the context
 * until the next batch; the FASTA text is framed when it is asked for -- by msim_batch_fetch straight into the caller's
 * memory (a mapped span of the output file), by msim_batch_view into a buffer of the context -- so the deflines
 * (msim_batch_contig.header) must stay valid until then.  MSIM_ERR_KEY: msim_batch_key_contig tells which contig hit the
 * reference's KeyError.                                                                                                */
typedef struct msim_batch_contig {
    const uint8_t *body; uint64_t body_bytes; uint64_t n_bases; uint32_t lenc, lenb;   /* as msim_add_contig_text     */
    const msim_range *ranges; int32_t n_ranges;                                        /* as msim_plan_contig         */
    const char *name;                                                                  /* CHROM of its VCF lines      */
    const char *header;      /* NULL: the FASTA text holds the framed bodies only.  Else the defline without '>': the text
                                is the complete run of records as FastaWriter writes them (fasta_writer.py:31-38) --
                                '>' header '\n' body, with a '\n' before a header iff the body before it ended mid-line */
    uint32_t name_len, header_len;   /* 0: `name` / `header` are NUL-terminated strings.  Else their length in bytes: they may
                                then point straight into the FASTA file text (an assembly has 10^5 deflines; nobody has
                                to copy them) */
} msim_batch_contig;
int msim_batch_run(msim_ctx *ctx, const msim_batch_contig *contigs, int n);
/* per contig: bytes of its part of the FASTA text (header line included where given), bytes of its VCF lines, plan-was-empty flag (mutator.py:125-129), records */
int msim_batch_sizes(msim_ctx *ctx, int n, uint64_t *fasta_bytes, uint64_t *vcf_bytes, int32_t *empty, uint64_t *n_records);
/* the framed bodies / the VCF lines of all contigs of the batch, back to back in contig order (either may be NULL)       */
int msim_batch_fetch(msim_ctx *ctx, uint8_t *fasta_text, uint64_t fasta_cap, char *vcf_text, uint64_t vcf_cap);
/* The last batch's two texts queued for their output files (see msim_fetch_sequence_framed_file: same channels, same
 * msim_file_wait): the FASTA text is framed into libmsim's buffer by its host threads and written from there, the VCF text
 * as it was rendered; either descriptor may be -1 (not wanted).  The texts' sizes: msim_batch_view (fasta_text == NULL).
 * The next msim_batch_run waits, where it overwrites them, until the channel has let go of the buffers.                  */
int msim_batch_fetch_file(msim_ctx *ctx, int fasta_fd, uint64_t fasta_offset, int vcf_fd, uint64_t vcf_offset);
/* the same texts in place (valid until the next batch / destroy): no copy of a few hundred MB.  *last_line_bases: bases on
 * the partial last line of the FASTA text (what FastaWriter needs to know to continue after it)                           */
int msim_batch_view(msim_ctx *ctx, const uint8_t **fasta_text, uint64_t *fasta_bytes, const char **vcf_text,
                    uint64_t *vcf_bytes, uint64_t *last_line_bases);
int msim_batch_key_contig(msim_ctx *ctx, int *contig);

/* ---- FASTA index pass of the host loader (replaces pyfaidx's index, util.py:77-91) ---------------------------------- */
/* Pure host code, no context: per record of a FASTA text the facts pyfaidx's index holds -- where the defline text and the
 * body sit, bases, bases per line (lenc), bytes per line (lenb) -- and pyfaidx's consistency verdict.  The body is NOT
 * touched otherwise: it goes to the device as text (msim_add_contig_text).  Call with records = NULL to count.
 * MSIM_ERR_VALUE: sequence text before the first defline (pyfaidx: FastaIndexingError).                                */
#define MSIM_FASTA_HAS_BODY    1u   /* at least one line follows the defline                                              */
#define MSIM_FASTA_BAD_LINES   2u   /* a line before the last non-empty one differs from the first line's length, or the
                                       last non-empty line is longer ("Line length of fasta file is not consistent")       */
#define MSIM_FASTA_NONUNIFORM  4u   /* mixed "\n" / "\r\n" terminators inside the record: no fixed stride for the device  */
typedef struct msim_fasta_record {
    uint64_t h0, h1;       /* defline text [h0, h1): without '>' and without the line terminator                          */
    uint64_t b0, b1;       /* body text [b0, b1): from the first base to the end of the record's last line (exclusive)     */
    uint64_t n_bases;
    uint32_t lenc, lenb;   /* bases / bytes of the first body line (bytes include its terminator)                          */
    uint32_t flags, rsv;
} msim_fasta_record;
int msim_fasta_index(const uint8_t *text, uint64_t n, msim_fasta_record *records, uint64_t cap, uint64_t *n_records);

/* ---- multi-GPU: one process per GPU, contigs' APPLY sharded, gather over RCCL ------------------ */
/* mutate()'s contig loop (mutator.py:111-141) is the unit of sharding: PLAN is replayed on every rank (the two
 * MT19937 streams chain across contigs), each rank APPLYs the contigs it owns, and the one exchange step is the
 * gather of the mutated contigs to `root` -- grouped ncclSend / ncclRecv, every peer over its own xGMI link.
 * librccl is dlopen()ed by msim_comm_init; the 128-byte ncclUniqueId travels over the caller's control plane. */
#define MSIM_COMM_ID_BYTES 128
int msim_comm_unique_id(uint8_t id[MSIM_COMM_ID_BYTES]);                       /* on one rank; broadcast the bytes */
int msim_comm_init(msim_ctx *ctx, const uint8_t id[MSIM_COMM_ID_BYTES], int rank, int world);   /* collective   */
int msim_comm_destroy(msim_ctx *ctx);
/* Mutated length of a planned contig where PLAN alone fixes it (SNP-only tables, SV mixes planned on the device):
 * every rank then knows every contig's size without an exchange.  *known = 0: only the applying rank knows it. */
int msim_planned_out_len(msim_ctx *ctx, int contig, uint64_t *out_len, int *known);
/* Slot i = contig contig_ids[i] (this context's id), applied by rank owner[i].  A slot has three PARTS: 0 the mutated stream
 * (out_len[i] bytes), 1 its record table (n_records[i] records of 16 bytes: the binary VCF, what __mutate_sequence hands the
 * VcfWriter, mutator.py:334-421), 2 its insert pool (pool_len[i] bytes).  n_records / pool_len NULL: streams only.  Sizes of
 * contigs a rank does not own come over the caller's control plane (msim_result_sizes on the owner).  Synchronises, then moves
 * every part to `root`.  device_addrs[3 i + part] (optional) = where the part now lives on this rank: the contig's own buffer
 * (owner), a receive buffer of the context (root), 0 elsewhere / for an empty part.  Valid until the next gather / clear.   */
/* A failure of this call on ONE rank (a slot that rank has not applied, a size that disagrees with its result) is fatal
 * for the communicator: its peers have posted the matching transfers and wait for them.  Tear the communicator down
 * (msim_comm_destroy on every rank) -- agree on the slots over the control plane first, as gather.Communicator does.      */
int msim_gather_to_root(msim_ctx *ctx, int n, const int *contig_ids, const int *owner, const uint64_t *out_len,
                        const uint64_t *n_records, const uint64_t *pool_len, int root, uint64_t *device_addrs);
/* The transfers `rank` posts for that gather, without touching a GPU: ops[5k..] = kind (0 send, 1 recv, 2 already
 * local), slot, part, peer, bytes -- in posting order ((slot, part) order on both sides of every pair); capacity 3 n ops. */
int msim_gather_plan(int n, const int *owner, const uint64_t *out_len, const uint64_t *n_records, const uint64_t *pool_len,
                     int rank, int world, int root, int64_t *ops, int *n_ops);
/* bytes at a device address msim_gather_to_root reported -> host (the root reading a remote contig's records / pool to
 * render its VCF lines with msim_render_vcf).                                                                            */
int msim_gather_fetch(msim_ctx *ctx, uint64_t device_addr, uint64_t bytes, void *dst);

/* ---- stats -------------------------------------------------------------------------------------- */
int msim_stats(msim_ctx *ctx, msim_timing *out);
int msim_reset_stats(msim_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MSIM_H */
