// GPU sampler interface (plan_gpu.hip)
#pragma once
#include "ctx.h"

namespace msim {

struct GpuPlan;
GpuPlan *gpu_plan_create();
void gpu_plan_destroy(GpuPlan *g);
// host generator states changed (msim_seed / msim_set_mt_state / host planner ran): drop device streams
void gpu_plan_invalidate(GpuPlan *g);
// bring the host generators up to the device streams' positions (before the host planner runs or
// msim_get_mt_state answers)
int gpu_plan_sync_to_host(Ctx *c, GpuPlan *g);
// optional sizing hint: generate at least this many words ahead on the first extension
void gpu_plan_reserve(GpuPlan *g, uint64_t py_words, uint64_t np_words);
// wait for everything enqueued, collect flags + exact stream position (no-op if nothing is pending)
int gpu_plan_finish(Ctx *c, GpuPlan *g);
bool gpu_plan_eligible(const Ctx *c, const msim_range *ranges, int n_ranges);
int plan_contig_gpu(Ctx *c, GpuPlan *g, Contig &ct, const msim_range *ranges, int n_ranges);
// SV mixes (SNPs + IN/DE/DU/IV on one large range): sample, type draw, filter, records, insert pool and SNP
// draws on the device; only the boundary chain over the non-SNP candidates runs on the host
bool gpu_plan_mixed_eligible(const Ctx *c, const msim_range *ranges, int n_ranges);
int plan_contig_gpu_mixed(Ctx *c, GpuPlan *g, Contig &ct, const msim_range *ranges, int n_ranges);
// deterministic-SNP ranges of any size and number (RMT mode): the host walks the chain of samples over words the
// device generated; records, SNP draws and APPLY stay on the device
bool gpu_plan_hostsample_eligible(const Ctx *c, const msim_range *ranges, int n_ranges);
int plan_contig_gpu_hostsample(Ctx *c, GpuPlan *g, Contig &ct, const msim_range *ranges, int n_ranges);
// everything else that is still a plain chain: several ranges with their own settings, SV types on many small ranges,
// SNP block above the sampling distance -- the host walks samples and boundary passes in one go over device-made
// words, accept tables and candidate types
bool gpu_plan_multimix_eligible(const Ctx *c, GpuPlan *g, uint64_t L, const msim_range *ranges, int n_ranges);
int plan_contig_gpu_multimix(Ctx *c, GpuPlan *g, Contig &ct, const msim_range *ranges, int n_ranges);
// SNP sampler: contigs with one drawing range queue their emission (records, SNP outcomes) and go through its stages in groups;
// gpu_emit_flush sends what is queued (every entry point that needs a result does), gpu_emit_pending tells whether a contig is
// still queued (mark_apply: its APPLY follows its group's emission -- msim_apply_contig)
int gpu_emit_flush(Ctx *c);
bool gpu_emit_pending(Ctx *c, int contig, bool mark_apply);
int gpu_plan_force_overflow(Ctx *c, GpuPlan *g);          // test support
void gpu_plan_abandon(GpuPlan *g);                        // a device engine failed mid-contig: the session is over
// before a contig goes to a device engine: room for its windows in the current session's jump-table span, re-basing the session
// where there is not (*fits = false: no span holds it -- the host planner's)
int gpu_plan_make_room(Ctx *c, GpuPlan *g, const msim_range *ranges, int n_ranges, bool *fits);

void gpu_plan_stream_status(Ctx *c, GpuPlan *g, int out[8]);

// Mean and variance of the CPython stream's words one stage consumes -- what the SNP sampler's anchored windows lay out the
// interval of a sample's start from (plan_gpu.hip: plan_contig_gpu; checked against simulation in tests/test_ahead_moments.py).
//   sample: random.sample(range(n), k) by the set path draws until k DISTINCT values are there -- A accepted draws, a coupon
//     collector's first k: E A = sum n/(n-i) = -n ln(1 - k/n), Var A = sum (i/n)/(1-i/n)^2 = n (k/(n-k) + ln(1 - k/n)) --,
//     each after a geometric number of getrandbits(bits) words with success p = n / 2^bits (_randbelow's rejection):
//     E = E A / p, Var = E A (1-p)/p^2 + Var A / p^2
//   SNP draws of K SNPs (mutator.py:428-463): uniform() = 2 words; with probability p_tv a transversion's randint(0, 1) =
//     _randbelow(2) = getrandbits(2) until < 2, a geometric(1/2) loop: E = K (2 + 2 p_tv), Var = K (2 p_tv + 4 p_tv (1 - p_tv))
struct StreamMoments { double e, v; };
StreamMoments sample_words_moments(uint64_t n, uint64_t k);
StreamMoments snp_words_moments(uint64_t K, unsigned long long ti_lim);

}  // namespace msim
