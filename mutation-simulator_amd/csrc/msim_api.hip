// C-ABI entry points of libmsim.so (include/msim.h).  gfx950 (MI355X) only.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <system_error>
#include <thread>
#include <sys/mman.h>

#include <dlfcn.h>

#include "ctx.h"
#include "plan_gpu.h"

namespace msim {

static thread_local std::string g_create_error;

namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char *n : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"}) {
            void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            *reinterpret_cast<void **>(&push) = dlsym(h, "roctxRangePushA");
            *reinterpret_cast<void **>(&pop) = dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
Roctx &roctx() { static Roctx r; return r; }
}  // namespace

TraceRange::TraceRange(const char *name) : on(false) {
    Roctx &r = roctx();
    if (r.push) { (void)r.push(name); on = true; }
}
TraceRange::~TraceRange() { if (on) (void)roctx().pop(); }

int fail(Ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}
int hip_fail(Ctx *c, hipError_t e, const char *what) {
    if (e == MSIM_WAIT_TIMED_OUT) {
        char buf[160];
        snprintf(buf, sizeof buf, ": no completion within %g s (MSIM_WAIT_TIMEOUT_S); the work it waits for is still queued or the device hangs",
                 wait_limit_seconds());
        return fail(c, MSIM_ERR_HIP, std::string(what) + buf);
    }
    return fail(c, MSIM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

double wait_limit_seconds() {
    const char *e = getenv("MSIM_WAIT_TIMEOUT_S");
    if (!e || !*e) return 120.0;
    const double v = atof(e);
    return v > 0 ? v : 0.0;
}

template <class Q>
static hipError_t wait_poll(Q &&query) {
    hipError_t e = query();
    if (e != hipErrorNotReady) return e;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {                                             // spinning: what nearly every wait of a step ends in
        for (int i = 0; i < 16; i++) __builtin_ia32_pause();
        if ((e = query()) != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    const double limit = wait_limit_seconds();
    for (;;) {                                             // long waits (a file's last pieces, a first step's allocations): leave the core
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((e = query()) != hipErrorNotReady) return e;
        if (limit > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return MSIM_WAIT_TIMED_OUT;
    }
}
hipError_t wait_stream(hipStream_t s) { return wait_poll([s]() { return hipStreamQuery(s); }); }
hipError_t wait_event(hipEvent_t ev) { return wait_poll([ev]() { return hipEventQuery(ev); }); }

struct Batch;
static void batch_free(Ctx *c);

struct CtxDev {            // small device-side constants owned by the ctx
    uint8_t *d_lut = nullptr;
};
static CtxDev *dev_of(Ctx *c);
uint8_t *ctx_lut(Ctx *c) { return dev_of(c)->d_lut; }

// Everything enqueued has completed and its deferred results are collected: sampler flags + exact
// stream position, APPLY timings + KeyError words.  Returns the sampler's error, if any.
static int drain(Ctx *c) {
    if (c->host_only) return MSIM_OK;
    TraceRange tr("msim drain (collect deferred results)");
    int rc = c->gpu ? gpu_plan_finish(c, c->gpu) : MSIM_OK;
    int rc2 = apply_finish(c);                             // (collects the counter-based engine's flags and sizes too)
    if (rc) return rc;
    if (rc2) return rc2;
    MSIM_HIP(c, wait_stream(c->stream));
    MSIM_HIP(c, wait_stream(c->emit_stream));
    return MSIM_OK;
}
// The APPLYs msim_apply_contig deferred.  only_groups (the engines' flush point at the start of a host walk): an incomplete group
// keeps waiting for its successors -- a full one goes out together through apply_batch_device.
static int defer_group_size() {
    static const int n = getenv("MSIM_NO_DEFER_PAIRS") ? 1 : getenv("MSIM_DEFER_GROUP") ? std::min(4, std::max(1, atoi(getenv("MSIM_DEFER_GROUP")))) : 3;
    return n;
}
bool deferred_apply_holds(const Ctx *c, int contig) {
    if (c->deferred_apply == contig) return contig >= 0;
    for (int q = 0; q < c->n_deferred_more; q++) if (c->deferred_more[q] == contig) return true;
    return false;
}
int flush_deferred_apply(Ctx *c, bool only_groups) {
    if (c->deferred_apply < 0) return MSIM_OK;
    if (only_groups && c->n_deferred_more + 1 < defer_group_size()) return MSIM_OK;
    std::vector<int> ids;
    for (int q = 0; q < c->n_deferred_more; q++) ids.push_back(c->deferred_more[q]);
    ids.push_back(c->deferred_apply);
    c->deferred_apply = -1;
    c->n_deferred_more = 0;
    ids.erase(std::remove_if(ids.begin(), ids.end(), [&](int idx) { return idx < 0 || idx >= (int)c->contigs.size(); }), ids.end());
    if (ids.size() >= 2) {
        for (int idx : ids) c->contigs[(size_t)idx].apply_stream = c->emit_stream;   // (apply_batch_device: contigs with a stream of their own)
        return apply_batch_device(c, ids, true);
    }
    for (int idx : ids) {
        const int rc = apply_contig_device(c, c->contigs[(size_t)idx]);
        if (rc) return rc;
    }
    return MSIM_OK;
}
static int key_error_of(Ctx *c, Contig &g) {
    g.key_reported = true;
    return fail(c, MSIM_ERR_KEY, std::string("KeyError: '") + (char)g.key_base + "'");
}

struct CtxFull : Ctx { CtxDev dev; };
static CtxDev *dev_of(Ctx *c) { return &static_cast<CtxFull *>(c)->dev; }

// Translation tables of mutator.py:75-77 and the transversion dict of mutator.py:449-455.
static void build_lut(uint8_t *lut) {
    uint8_t conv[256], comp[256], ti[256];
    for (int i = 0; i < 256; i++) conv[i] = comp[i] = ti[i] = (uint8_t)i;
    const char *a = "KSYMWRBDHV-", *b = "GCCAAACAAAN";
    for (int i = 0; a[i]; i++) conv[(uint8_t)a[i]] = (uint8_t)b[i];
    a = "ACGTUMRWSYKVHDB"; b = "TGCAAKYWSRMBDHV";
    for (int i = 0; a[i]; i++) comp[(uint8_t)a[i]] = (uint8_t)b[i];
    a = "AGTC"; b = "GACT";
    for (int i = 0; a[i]; i++) ti[(uint8_t)a[i]] = (uint8_t)b[i];
    for (int x = 0; x < 256; x++) {
        const uint8_t r = conv[x];
        lut[x] = ti[r];
        const char *pair = nullptr;
        switch (r) {
            case 'A': pair = "TC"; break;
            case 'G': pair = "CT"; break;
            case 'T': pair = "GA"; break;
            case 'C': pair = "AG"; break;
            case 'N': pair = "NN"; break;
            default: break;
        }
        lut[256 + x] = pair ? (uint8_t)pair[0] : 0;      // 0 -> the reference raises KeyError(r)
        lut[512 + x] = pair ? (uint8_t)pair[1] : 0;
        lut[768 + x] = r;
        lut[1024 + x] = comp[r];
    }
}

// forget a contig's plan/apply results; device buffers stay allocated for the next plan of this contig
static void reset_contig(Contig &g) {   // (callers also drop the context's text cache: see text_kind)
    g.planned = g.applied = false;
    g.defer_apply = false;
    g.tile_index_by_plan = false;
    g.delta_known = false;
    g.off_ready = false;
    g.known_delta = 0;
    g.d_dyn = nullptr;
    g.sizes_pending = false;
    g.n_rec_cap = g.out_cap_len = g.n_struct_est = 0;
    g.n_rec = g.pool_len = g.out_len = 0;
    g.h_recs.clear(); g.h_recs.shrink_to_fit();
    g.h_pool.clear(); g.h_pool.shrink_to_fit();
}

static int free_contig(Ctx *c, Contig &g, bool keep_input) {
    if (g.d_recs) { MSIM_HIP(c, hipFree(g.d_recs)); g.d_recs = nullptr; }
    if (g.d_pool) { MSIM_HIP(c, hipFree(g.d_pool)); g.d_pool = nullptr; }
    if (g.d_out) { MSIM_HIP(c, hipFree(g.d_out)); g.d_out = nullptr; }
    if (g.d_off) { MSIM_HIP(c, hipFree(g.d_off)); g.d_off = nullptr; }
    if (g.d_first) { MSIM_HIP(c, hipFree(g.d_first)); g.d_first = nullptr; }
    g.cap_recs = g.cap_pool = g.cap_out = g.cap_off = g.cap_first = 0;
    if (g.ea0) { (void)hipEventDestroy(g.ea0); (void)hipEventDestroy(g.ea1); (void)hipEventDestroy(g.ea2); g.ea0 = g.ea1 = g.ea2 = nullptr; }
    g.apply_pending = false;
    g.key_error = false;
    reset_contig(g);
    if (!keep_input && g.d_in) { MSIM_HIP(c, hipFree(g.d_in)); g.d_in = nullptr; }
    return MSIM_OK;
}

static Contig *get_contig(Ctx *c, int id) {
    if (id < 0 || (size_t)id >= c->contigs.size()) { fail(c, MSIM_ERR_ARG, "no such contig"); return nullptr; }
    return &c->contigs[(size_t)id];
}

static int new_contig(Ctx *c, uint64_t len, Contig **out) {
    if (len >= (1ull << 32)) return fail(c, MSIM_ERR_UNSUPPORTED, "contig of 4 GiB or more (multi-word getrandbits)");
    if (c->contigs.size() >= (size_t)MAX_CONTIGS) return fail(c, MSIM_ERR_UNSUPPORTED, "more than 65536 contigs in one context");
    Contig g;
    g.len = len;
    g.index = (int)c->contigs.size();
    if (!c->host_only) {
        // PAD bytes of zero slack on BOTH sides: shifted 16-B reads may start before / end after the data
        MSIM_HIP(c, hipMalloc(&g.d_in, len + 2 * PAD));
        MSIM_HIP(c, hipMemsetAsync(g.d_in, 0, PAD, c->stream));
        MSIM_HIP(c, hipMemsetAsync(g.d_in + PAD + len, 0, PAD, c->stream));
    }
    c->contigs.push_back(g);
    *out = &c->contigs.back();
    return MSIM_OK;
}

}  // namespace msim

using namespace msim;

extern "C" {

int msim_abi_version(void) { return MSIM_ABI_VERSION; }

int msim_warm_up(int device_id) {
    if (device_id < 0) return MSIM_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id >= n) return MSIM_ERR_HIP;
    if (hipSetDevice(device_id) != hipSuccess) return MSIM_ERR_HIP;
    return hipFree(nullptr) == hipSuccess ? MSIM_OK : MSIM_ERR_HIP;     // forces the runtime + device context up
}

int msim_device_host_cpus(int device_id, char *cpulist, int cap) {
    if (!cpulist || cap <= 0) return MSIM_ERR_ARG;
    cpulist[0] = 0;
    if (device_id < 0) return MSIM_OK;
    char buf[4096];
    device_host_cpus(device_id, buf, sizeof buf);
    snprintf(cpulist, (size_t)cap, "%s", buf);
    return MSIM_OK;
}

int msim_create(int device_id, uint32_t flags, msim_ctx **out) {
    if (!out) return MSIM_ERR_ARG;
    *out = nullptr;
    if (device_id == -1) {
        // Host-only context: the sequential planner and the text renderer work, every entry point
        // that needs the GPU fails with MSIM_ERR_HIP.  Exists so the planner can be checked against
        // the oracle on machines without a GPU; it cannot produce a mutated sequence.
        CtxFull *c = new (std::nothrow) CtxFull();
        if (!c) return MSIM_ERR_NOMEM;
        c->device = -1;
        c->host_only = true;
        c->flags = flags;
        c->devname = "host-only (no GPU)";
        c->py.init_genrand(5489u);
        c->np.init_genrand(5489u);
        for (int i = 0; i < 8; i++) c->params.block[i] = 1;
        c->params.ti_lim = (1ull << 52) + 1;
        *out = reinterpret_cast<msim_ctx *>(static_cast<Ctx *>(c));
        return MSIM_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, MSIM_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, MSIM_ERR_ARG, "device_id out of range");
    CtxFull *c = new (std::nothrow) CtxFull();
    if (!c) return MSIM_ERR_NOMEM;
    c->device = device_id;
    c->flags = flags;
    c->py.init_genrand(5489u);
    c->np.init_genrand(5489u);
    for (int i = 0; i < 8; i++) c->params.block[i] = 1;
    c->params.ti_lim = (1ull << 52) + 1;
    auto bail = [&](hipError_t he, const char *what) {
        int rc = hip_fail(nullptr, he, what);
        delete c;
        return rc;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail(e, "hipSetDevice");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
    c->devname = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    {   // the plan chain is the critical path of a step: give its stream the highest priority (and with it
        // a hardware queue of its own), so its small kernels are dispatched ahead of the bulk work that
        // runs beside it (jump cascade, chunk generation, record emission, rewrite)
        int lo = 0, hi = 0;
        if ((e = hipDeviceGetStreamPriorityRange(&lo, &hi)) != hipSuccess) return bail(e, "hipDeviceGetStreamPriorityRange");
        if ((e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi)) != hipSuccess) return bail(e, "hipStreamCreate");
    }
    if ((e = hipStreamCreateWithFlags(&c->emit_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    hipEvent_t *evs[4] = {&c->ev0, &c->ev1, &c->ev2, &c->ev3};
    for (auto ev : evs)
        if ((e = hipEventCreate(ev)) != hipSuccess) return bail(e, "hipEventCreate");
    uint8_t lut[1280];
    build_lut(lut);
    if ((e = hipMalloc(&c->dev.d_lut, sizeof lut)) != hipSuccess) return bail(e, "hipMalloc(lut)");
    if ((e = hipMemcpy(c->dev.d_lut, lut, sizeof lut, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(lut)");
    if ((e = hipMalloc(&c->d_errs, (size_t)2 * MAX_CONTIGS * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(errs)");   // KeyError words + length-check words
    if ((e = hipHostMalloc(&c->h_mail, 64, hipHostMallocMapped)) != hipSuccess) return bail(e, "hipHostMalloc(mailbox)");
    c->gpu = gpu_plan_create();
    *out = reinterpret_cast<msim_ctx *>(static_cast<Ctx *>(c));
    return MSIM_OK;
}

static Ctx *C(msim_ctx *p) { return reinterpret_cast<Ctx *>(p); }
// Every entry point except msim_plan_contig first enqueues an APPLY that msim_apply_contig deferred (see there).
#define CTX_FLUSHED(c, p)                                   \
    Ctx *c = C(p);                                          \
    if (c) {                                                \
        int flush_rc_ = flush_deferred_apply(c);            \
        if (!flush_rc_ && c->fast) flush_rc_ = fast_plan_flush(c);   /* (fast contexts: the plans queued so far, as one batch) */ \
        if (!flush_rc_ && c->gpu) flush_rc_ = gpu_emit_flush(c);     /* (SNP sampler: the emission group that is still waiting) */ \
        if (flush_rc_) return flush_rc_;                    \
    }
#define NEED_GPU(c) do { if ((c)->host_only) return fail((c), MSIM_ERR_HIP, "host-only context: this call needs the GPU"); } while (0)

void msim_destroy(msim_ctx *p) {
    if (!p) return;
    CtxFull *c = static_cast<CtxFull *>(C(p));
    c->deferred_apply = -1;                                // nobody will ask for their results
    c->n_deferred_more = 0;
    if (c->host_only) { file_io_destroy(c); batch_free(c); delete c; return; }
    static const bool prof = getenv("MSIM_BATCH_PROF") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    double ph[6] = {0, 0, 0, 0, 0, 0};
    auto lap = [&](int i) { const auto n = std::chrono::steady_clock::now(); ph[i] += std::chrono::duration<double, std::milli>(n - tp).count(); tp = n; };
    (void)hipSetDevice(c->device);
    file_io_destroy(c);                                    // (finishes what is queued for the output files first)
    (void)wait_stream(c->stream);
    (void)wait_stream(c->emit_stream);
    lap(0);
    for (auto &g : c->contigs) (void)free_contig(c, g, false);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->dev.d_lut) (void)hipFree(c->dev.d_lut);
    if (c->d_errs) (void)hipFree(c->d_errs);
    if (c->h_errs) (void)hipHostFree(c->h_errs);
    if (c->d_text) (void)hipFree(c->d_text);
    if (c->d_text_scratch) (void)hipFree(c->d_text_scratch);
    lap(1);
    batch_free(c);
    lap(2);
    comm_destroy(c);
    fast_plan_destroy(c);
    gpu_plan_destroy(c->gpu);
    lap(3);
    if (c->h_mail) (void)hipHostFree(c->h_mail);
    hipEvent_t evs[4] = {c->ev0, c->ev1, c->ev2, c->ev3};
    for (auto ev : evs) if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->emit_stream) (void)hipStreamDestroy(c->emit_stream);
    lap(4);
    if (prof) fprintf(stderr, "msim_destroy: drain %.1f ms, contigs+scratch %.1f, batch buffers %.1f, plan engines %.1f, streams %.1f\n", ph[0], ph[1], ph[2], ph[3], ph[4]);
    delete c;
}

const char *msim_last_error(const msim_ctx *p) {
    if (!p) return g_create_error.c_str();
    return reinterpret_cast<const Ctx *>(p)->err.c_str();
}

int msim_device_name(const msim_ctx *p, char *dst, int cap) {
    if (!p || !dst || cap <= 0) return MSIM_ERR_ARG;
    snprintf(dst, (size_t)cap, "%s", reinterpret_cast<const Ctx *>(p)->devname.c_str());
    return MSIM_OK;
}

int msim_sync(msim_ctx *p) {
    CTX_FLUSHED(c, p)
    if (!c) return MSIM_ERR_ARG;
    if (c->host_only) return MSIM_OK;
    int rc = drain(c);
    if (rc) return rc;
    for (auto &g : c->contigs)                             // deferred KeyError of an asynchronous APPLY
        if (g.key_error && !g.key_reported) return key_error_of(c, g);
    return file_wait(c);                                   // ... and what was queued for the output files is there
}

int msim_seed(msim_ctx *p, const uint32_t *py_key, int n_key, uint32_t np_seed) {
    CTX_FLUSHED(c, p)
    if (!c || !py_key || n_key < 1) return MSIM_ERR_ARG;
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    c->py.init_by_array(py_key, n_key);
    c->py.words = 0;
    c->np.init_genrand(np_seed);
    c->np.words = 0;
    if (c->gpu) gpu_plan_invalidate(c->gpu);
    return MSIM_OK;
}

int msim_set_mt_state(msim_ctx *p, int stream, const uint32_t mt[624], int pos) {
    CTX_FLUSHED(c, p)
    if (!c || !mt || pos < 0 || pos > 624 || stream < 0 || stream > 1) return MSIM_ERR_ARG;
    if (c->gpu) {                      // keep the other stream's position before dropping the device copy
        int rc = gpu_plan_sync_to_host(c, c->gpu);
        if (rc) return rc;
    }
    HostMT &g = stream ? c->np : c->py;
    memcpy(g.mt, mt, sizeof g.mt);
    g.idx = pos;
    g.state_changed();
    if (c->gpu) gpu_plan_invalidate(c->gpu);
    return MSIM_OK;
}

int msim_get_mt_state(msim_ctx *p, int stream, uint32_t mt[624], int *pos) {
    CTX_FLUSHED(c, p)
    if (!c || !mt || !pos || stream < 0 || stream > 1) return MSIM_ERR_ARG;
    if (c->gpu) {
        int rc = gpu_plan_sync_to_host(c, c->gpu);
        if (rc) return rc;
    }
    HostMT &g = stream ? c->np : c->py;
    memcpy(mt, g.mt, sizeof g.mt);
    *pos = g.idx;
    return MSIM_OK;
}

int msim_reserve_streams(msim_ctx *p, uint64_t py_words, uint64_t np_words) {
    CTX_FLUSHED(c, p)
    if (!c) return MSIM_ERR_ARG;
    if (c->gpu) gpu_plan_reserve(c->gpu, py_words, np_words);
    return MSIM_OK;
}

int msim_add_contig(msim_ctx *p, const uint8_t *bases, uint64_t len, int *contig) {
    CTX_FLUSHED(c, p)
    if (!c || (!bases && len) || !contig) return MSIM_ERR_ARG;
    Contig *g;
    int rc = new_contig(c, len, &g);
    if (rc) return rc;
    if (!c->host_only) {
        if (len) MSIM_HIP(c, hipMemcpyAsync(g->d_in + PAD, bases, len, hipMemcpyHostToDevice, c->stream));
        MSIM_HIP(c, wait_stream(c->stream));
    }
    *contig = (int)c->contigs.size() - 1;
    return MSIM_OK;
}

int msim_add_contig_synthetic(msim_ctx *p, uint64_t len, uint64_t seed, int *contig) {
    CTX_FLUSHED(c, p)
    if (!c || !contig) return MSIM_ERR_ARG;
    NEED_GPU(c);
    Contig *g;
    int rc = new_contig(c, len, &g);
    if (rc) return rc;
    rc = synth_contig_device(c, g->d_in + PAD, len, seed);
    if (rc) return rc;
    MSIM_HIP(c, wait_stream(c->stream));
    *contig = (int)c->contigs.size() - 1;
    return MSIM_OK;
}

int msim_contig_length(msim_ctx *p, int contig, uint64_t *len) {
    CTX_FLUSHED(c, p)
    if (!c || !len) return MSIM_ERR_ARG;
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    *len = g->len;
    return MSIM_OK;
}

int msim_read_contig(msim_ctx *p, int contig, uint64_t offset, uint64_t n, uint8_t *dst) {
    CTX_FLUSHED(c, p)
    if (!c || (!dst && n)) return MSIM_ERR_ARG;
    NEED_GPU(c);
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (offset > g->len || n > g->len - offset) return fail(c, MSIM_ERR_ARG, "read beyond contig end");
    if (n) MSIM_HIP(c, hipMemcpyAsync(dst, g->d_in + PAD + offset, n, hipMemcpyDeviceToHost, c->stream));
    MSIM_HIP(c, wait_stream(c->stream));
    return MSIM_OK;
}

int msim_clear(msim_ctx *p) {
    CTX_FLUSHED(c, p)
    if (!c) return MSIM_ERR_ARG;
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    for (auto &g : c->contigs) {
        int rc = free_contig(c, g, false);
        if (rc) return rc;
    }
    c->contigs.clear();
    c->text_kind = 0;
    return MSIM_OK;
}

int msim_set_params(msim_ctx *p, const msim_params *params) {
    CTX_FLUSHED(c, p)
    if (!c || !params) return MSIM_ERR_ARG;
    for (int t = 1; t <= 7; t++)
        if (params->block[t] < 1) return fail(c, MSIM_ERR_ARG, "block values must be >= 1 (rmt.py:326-345)");
    c->params = *params;
    c->have_params = true;
    return MSIM_OK;
}

int msim_set_plan_mode(msim_ctx *p, uint32_t mode) {
    CTX_FLUSHED(c, p)
    if (!c || (mode & ~(MSIM_PLAN_HOST | MSIM_PLAN_GPU)) || mode == (MSIM_PLAN_HOST | MSIM_PLAN_GPU)) return MSIM_ERR_ARG;
    if ((mode & MSIM_PLAN_GPU) && c->host_only) return fail(c, MSIM_ERR_HIP, "host-only context: no device engine to force");
    c->flags = (c->flags & ~(uint32_t)(MSIM_PLAN_HOST | MSIM_PLAN_GPU)) | mode;
    return MSIM_OK;
}

// One iteration of mutate()'s contig loop up to the rewrite, for contig `g` (id `contig`; chain only: a length-only stand-in,
// id -1): picks the PLAN engine, keeps the streams chained.
static int plan_dispatch(Ctx *c, Contig *g, int contig, const msim_range *ranges, int n_ranges) {
    int rc = MSIM_OK;
    TraceRange tr(c->chain_only ? "msim PLAN contig (chain only)" : "msim PLAN contig");
    if (c->flags & MSIM_RNG_FAST) {                        // counter-based generator: nothing chains, nothing is stream-compatible
        const uint32_t seq = c->fast_seq++;
        // (a contig another rank owns: only its ordinal matters -- but what the reference refuses (ValueError) and what this
        //  mode does not cover is refused on every rank alike)
        if (c->chain_only) return fast_plan_check(c, g->len, ranges, n_ranges);
        if (c->host_only || !c->gpu) return fail(c, MSIM_ERR_HIP, "fast RNG mode needs the GPU");
        if ((rc = flush_deferred_apply(c))) return rc;
        if (g->apply_pending && (rc = apply_finish(c))) return rc;      // this contig's tables may still be read by its last APPLY
        reset_contig(*g);
        c->text_kind = 0;
        rc = plan_contig_fast(c, *g, ranges, n_ranges, c->fast_key, seq);
        if (!rc) c->t.contigs_fast++;
        return rc;
    }
    const bool dev = !c->host_only && c->gpu && !(c->flags & MSIM_PLAN_HOST);
    bool gpu_ok = dev && gpu_plan_eligible(c, ranges, n_ranges);
    bool mixed_ok = dev && !gpu_ok && gpu_plan_mixed_eligible(c, ranges, n_ranges);
    bool hs_ok = dev && !gpu_ok && !mixed_ok && gpu_plan_hostsample_eligible(c, ranges, n_ranges);
    bool mm_ok = dev && !gpu_ok && !mixed_ok && !hs_ok && gpu_plan_multimix_eligible(c, c->gpu, g->len, ranges, n_ranges);
    if (gpu_ok || mixed_ok || hs_ok || mm_ok) {            // the device streams' sessions span 1.31 G words: re-base where this contig
        bool fits = true;                                  // would not fit; a contig no span holds is the host planner's
        if ((rc = gpu_plan_make_room(c, c->gpu, ranges, n_ranges, &fits))) return rc;
        if (!fits) gpu_ok = mixed_ok = hs_ok = mm_ok = false;
    }
    // A deferred APPLY of the previous contig (msim_apply_contig) is enqueued when this plan's host chain starts --
    // by the engine itself -- so that it fills the device's idle time instead of competing with this contig's
    // latency-bound chain kernels.  Every other route enqueues it now.
    if (c->gpu && (!gpu_ok || gpu_emit_pending(c, contig, false)) && (rc = gpu_emit_flush(c))) return rc;   // (another engine / planned again)
    const bool host_chain = mixed_ok || hs_ok || mm_ok;
    const bool engine_flushes = host_chain && !deferred_apply_holds(c, contig);
    if (!engine_flushes && (rc = flush_deferred_apply(c))) return rc;
    reset_contig(*g);
    c->text_kind = 0;
    if ((c->flags & MSIM_PLAN_GPU) && !gpu_ok && !host_chain)
        return fail(c, MSIM_ERR_UNSUPPORTED, "GPU sampler not available for this stream structure");
    if (gpu_ok || host_chain) {
        if (gpu_ok) rc = plan_contig_gpu(c, c->gpu, *g, ranges, n_ranges);
        else {
            if (g->apply_pending && (mixed_ok || mm_ok)) {   // this contig's buffers may still be read by its last APPLY
                rc = apply_finish(c);
                if (rc) return rc;
            }
            if (mixed_ok) rc = plan_contig_gpu_mixed(c, c->gpu, *g, ranges, n_ranges);
            else if (hs_ok) rc = plan_contig_gpu_hostsample(c, c->gpu, *g, ranges, n_ranges);
            else rc = plan_contig_gpu_multimix(c, c->gpu, *g, ranges, n_ranges);
        }
        if (!rc && !c->chain_only)
            (gpu_ok ? c->t.contigs_snp : mixed_ok ? c->t.contigs_svmix : hs_ok ? c->t.contigs_hostcut : c->t.contigs_hostchain)++;
        // test hook: MSIM_DBG_FORCE_OVERFLOW=n raises the window-overflow flag behind the n-th device-planned contig of
        // the process (1-based), so that the callers' recovery (mutator.py: re-plan through the host planner) can be tested
        static const int force_at = getenv("MSIM_DBG_FORCE_OVERFLOW") ? atoi(getenv("MSIM_DBG_FORCE_OVERFLOW")) : 0;
        static int device_plans = 0;
        if (!rc && force_at && ++device_plans == force_at) rc = gpu_plan_force_overflow(c, c->gpu);
        if (rc == MSIM_ERR_HIP) gpu_plan_abandon(c->gpu);      // (a wait's deadline, a failed call: nothing of this session is known any more)
        const int frc = flush_deferred_apply(c, rc == MSIM_OK);   // (an engine that failed never reached its flush point; a single
                                                                   //  contig left waiting by design stays for its successor)
        g->defer_apply = !rc && host_chain && !c->chain_only;
        if (c->chain_only) g->planned = false;              // nothing to apply or fetch
        return rc ? rc : frc;
    }
    if (c->gpu) {                      // the host planner continues from wherever the device streams stand
        rc = gpu_plan_sync_to_host(c, c->gpu);
        if (rc) return rc;
    }
    if (!c->host_only) {               // this contig's buffers may still be read by an asynchronous APPLY
        rc = apply_finish(c);
        if (rc) return rc;
    }
    HostPlan hp;
    const uint64_t w_py = c->py.words, w_np = c->np.words;
    rc = plan_contig_host(c, g->len, ranges, n_ranges, hp);
    if (rc) return rc;
    c->t.py_words += c->py.words - w_py;
    c->t.np_words += c->np.words - w_np;
    if (c->chain_only) return MSIM_OK;                     // the streams have advanced: that is all
    c->t.contigs_host++;
    const auto t0 = std::chrono::steady_clock::now();
    g->n_rec = hp.recs.size();
    g->pool_len = hp.pool.size();
    g->plan_empty = hp.empty;
    g->all_snp = true;
    for (const msim_record &r : hp.recs)
        if (r.type != MSIM_SN) { g->all_snp = false; break; }
    if (c->host_only) {
        g->h_recs.swap(hp.recs);
        g->h_pool.swap(hp.pool);
        g->planned = true;
        return MSIM_OK;
    }
    if (g->n_rec) {
        rc = dev_reserve(c, (void **)&g->d_recs, &g->cap_recs, g->n_rec * sizeof(msim_record));
        if (rc) return rc;
        MSIM_HIP(c, hipMemcpyAsync(g->d_recs, hp.recs.data(), g->n_rec * sizeof(msim_record), hipMemcpyHostToDevice, c->stream));
    }
    rc = dev_reserve(c, (void **)&g->d_pool, &g->cap_pool, g->pool_len + 2 * PAD);
    if (rc) return rc;
    if (g->pool_len)
        MSIM_HIP(c, hipMemcpyAsync(g->d_pool + PAD, hp.pool.data(), g->pool_len, hipMemcpyHostToDevice, c->stream));
    MSIM_HIP(c, wait_stream(c->stream));
    c->t.upload_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    g->planned = true;
    return MSIM_OK;
}

int msim_plan_contig(msim_ctx *p, int contig, const msim_range *ranges, int n_ranges) {
    Ctx *c = C(p);
    if (!c || n_ranges < 0 || (n_ranges && !ranges)) return MSIM_ERR_ARG;
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (!c->have_params) return fail(c, MSIM_ERR_ARG, "msim_set_params has not been called");
    return plan_dispatch(c, g, contig, ranges, n_ranges);
}

int msim_plan_chain(msim_ctx *p, uint64_t len, const msim_range *ranges, int n_ranges) {
    Ctx *c = C(p);
    if (!c || n_ranges < 0 || (n_ranges && !ranges)) return MSIM_ERR_ARG;
    if (!c->have_params) return fail(c, MSIM_ERR_ARG, "msim_set_params has not been called");
    if (len >= (1ull << 32)) return fail(c, MSIM_ERR_UNSUPPORTED, "contig of 4 GiB or more (multi-word getrandbits)");
    Contig stand_in;                                       // a length, nothing else: no bases, no tables
    stand_in.len = len;
    c->chain_only = true;
    const int rc = plan_dispatch(c, &stand_in, -1, ranges, n_ranges);
    c->chain_only = false;
    return rc;
}

int msim_plan_was_empty(msim_ctx *p, int contig, int *empty) {
    CTX_FLUSHED(c, p)
    if (!c || !empty) return MSIM_ERR_ARG;
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (!g->planned) return fail(c, MSIM_ERR_ARG, "contig has not been planned");
    if (g->sizes_pending) {                                // the counter-based engine decides on the device what survives
        const int rc = drain(c);
        if (rc) return rc;
    }
    *empty = g->plan_empty ? 1 : 0;
    return MSIM_OK;
}

int msim_apply_contig(msim_ctx *p, int contig) {
    Ctx *c = C(p);
    if (!c) return MSIM_ERR_ARG;
    // fast contexts: a contig whose plan is still queued gets its APPLY enqueued right behind the batch (plan_fast.hip)
    if (c->fast && fast_plan_queued(c, contig, true)) return MSIM_OK;
    // SNP sampler: a contig whose emission still waits for its group is applied right behind that group (plan_gpu.hip)
    if (c->gpu && gpu_emit_pending(c, contig, true)) return MSIM_OK;
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    static const bool no_defer = getenv("MSIM_NO_DEFER") != nullptr;
    const bool defer = g->planned && g->defer_apply && !no_defer && !c->host_only;
    {   // (a contig that will wait itself pairs up with the one already waiting; anything else sends what waits first)
        int rc = (defer && c->n_deferred_more + 1 < defer_group_size() && !deferred_apply_holds(c, contig)) ? MSIM_OK : flush_deferred_apply(c);
        if (!rc && c->fast) rc = fast_plan_flush(c);
        if (rc) return rc;
    }
    NEED_GPU(c);
    if (!g->planned) return fail(c, MSIM_ERR_ARG, "msim_apply_contig before msim_plan_contig");
    TraceRange tr("msim APPLY contig");
    c->text_kind = 0;
    // Contigs of the engines with a host chain: the next contig's plan starts with latency-bound device kernels and
    // then leaves the device idle for a millisecond while the host walks -- that is where this APPLY belongs.  It is
    // enqueued by the next entry point, whichever it is (the next plan at the start of its host chain).
    if (defer) {
        if (c->deferred_apply >= 0) c->deferred_more[c->n_deferred_more++] = c->deferred_apply;   // (the group this one joins)
        c->deferred_apply = contig;
        return MSIM_OK;
    }
    return apply_contig_device(c, *g);
}

int msim_key_error(msim_ctx *p, int contig, uint8_t *base, uint64_t *pos) {
    CTX_FLUSHED(c, p)
    if (!c) return MSIM_ERR_ARG;
    Contig *g = nullptr;
    if (contig == -1) {                                    // first contig that hit it
        for (auto &q : c->contigs) if (q.key_error) { g = &q; break; }
        if (!g) return fail(c, MSIM_ERR_ARG, "no KeyError recorded");
    } else {
        g = get_contig(c, contig);
        if (!g) return MSIM_ERR_ARG;
    }
    if (!g->key_error) return fail(c, MSIM_ERR_ARG, "no KeyError recorded for this contig");
    if (base) *base = g->key_base;
    if (pos) *pos = g->key_pos;
    return MSIM_OK;
}

int msim_result_sizes(msim_ctx *p, int contig, uint64_t *out_len, uint64_t *n_records, uint64_t *pool_len) {
    CTX_FLUSHED(c, p)
    if (!c) return MSIM_ERR_ARG;
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (!g->planned) return fail(c, MSIM_ERR_ARG, "contig has not been planned");
    if (g->sizes_pending) {
        const int rc = drain(c);
        if (rc) return rc;
    }
    if (out_len) {
        if (!g->applied) return fail(c, MSIM_ERR_ARG, "contig has not been applied");
        int rc = drain(c);
        if (rc) return rc;
        if (g->key_error) return key_error_of(c, *g);
        *out_len = g->out_len;
    }
    if (n_records) *n_records = g->n_rec;
    if (pool_len) *pool_len = g->pool_len;
    return MSIM_OK;
}

int msim_fetch_sequence(msim_ctx *p, int contig, uint64_t offset, uint64_t n, uint8_t *dst) {
    CTX_FLUSHED(c, p)
    if (!c || (!dst && n)) return MSIM_ERR_ARG;
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (!g->applied) return fail(c, MSIM_ERR_ARG, "contig has not been applied");
    {
        int rc = drain(c);
        if (rc) return rc;
        if (g->key_error) return key_error_of(c, *g);
    }
    if (offset > g->out_len || n > g->out_len - offset) return fail(c, MSIM_ERR_ARG, "fetch beyond mutated contig end");
    if (n) MSIM_HIP(c, hipMemcpyAsync(dst, g->d_out + offset, n, hipMemcpyDeviceToHost, c->stream));
    MSIM_HIP(c, wait_stream(c->stream));
    return MSIM_OK;
}

int msim_fetch_records(msim_ctx *p, int contig, msim_record *dst, uint8_t *pool_dst) {
    CTX_FLUSHED(c, p)
    if (!c) return MSIM_ERR_ARG;
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (!g->planned) return fail(c, MSIM_ERR_ARG, "contig has not been planned");
    if (c->host_only) {
        if (dst && g->n_rec) memcpy(dst, g->h_recs.data(), g->n_rec * sizeof(msim_record));
        if (pool_dst && g->pool_len) memcpy(pool_dst, g->h_pool.data(), g->pool_len);
        return MSIM_OK;
    }
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    if (dst && g->n_rec)
        MSIM_HIP(c, hipMemcpyAsync(dst, g->d_recs, g->n_rec * sizeof(msim_record), hipMemcpyDeviceToHost, c->stream));
    if (pool_dst && g->pool_len)
        MSIM_HIP(c, hipMemcpyAsync(pool_dst, g->d_pool + PAD, g->pool_len, hipMemcpyDeviceToHost, c->stream));
    MSIM_HIP(c, wait_stream(c->stream));
    return MSIM_OK;
}

int msim_result_checksum(msim_ctx *p, int contig, uint64_t *sum) {
    CTX_FLUSHED(c, p)
    if (!c || !sum) return MSIM_ERR_ARG;
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (!g->applied) return fail(c, MSIM_ERR_ARG, "contig has not been applied");
    {
        int rc = drain(c);
        if (rc) return rc;
        if (g->key_error) return key_error_of(c, *g);
    }
    return checksum_device(c, g->d_out, g->out_len, sum);
}

int msim_result_device_ptr(msim_ctx *p, int contig, uint64_t *device_address, uint64_t *len) {
    CTX_FLUSHED(c, p)
    if (!c || !device_address || !len) return MSIM_ERR_ARG;
    NEED_GPU(c);
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (!g->applied) return fail(c, MSIM_ERR_ARG, "contig has not been applied");
    int rc = drain(c);
    if (rc) return rc;
    if (g->key_error) return key_error_of(c, *g);
    *device_address = (uint64_t)(uintptr_t)g->d_out;
    *len = g->out_len;
    return MSIM_OK;
}

int msim_planned_out_len(msim_ctx *p, int contig, uint64_t *out_len, int *known) {
    CTX_FLUSHED(c, p)
    if (!c || !out_len || !known) return MSIM_ERR_ARG;
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    *known = 0;
    *out_len = 0;
    if (!g->planned) return MSIM_OK;                       // (a contig this rank only walked the chain for: its owner knows)
    if (g->sizes_pending && !g->all_snp) return MSIM_OK;   // (decided on the device, not collected yet)
    if (g->all_snp || g->n_rec == 0) { *known = 1; *out_len = g->len; }
    else if (g->delta_known) { *known = 1; *out_len = (uint64_t)((long long)g->len + g->known_delta); }
    return MSIM_OK;
}

int msim_release_result(msim_ctx *p, int contig) {
    CTX_FLUSHED(c, p)
    if (!c) return MSIM_ERR_ARG;
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    return free_contig(c, *g, true);
}

// ---- text on the device -------------------------------------------------
static int text_copy_out(Ctx *c, uint8_t *out, uint64_t cap, uint64_t *needed) {
    *needed = c->text_len;
    if (!out) return MSIM_OK;
    if (cap < c->text_len) return fail(c, MSIM_ERR_ARG, "text buffer too small (see *needed)");
    if (c->text_len) MSIM_HIP(c, hipMemcpyAsync(out, c->d_text, c->text_len, hipMemcpyDeviceToHost, c->stream));
    MSIM_HIP(c, wait_stream(c->stream));
    return MSIM_OK;
}

// render (unless the preceding size call left the text in d_text)
static int vcf_text_ready(Ctx *c, int contig, const char *seq_name, bool reuse) {
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (!g->planned) return fail(c, MSIM_ERR_ARG, "contig has not been planned");
    if (reuse && c->text_kind == 1 && c->text_contig == contig) return MSIM_OK;
    int rc = drain(c);                                                   // record aux bytes come from the emit stream
    if (rc) return rc;
    c->text_kind = 0;
    uint64_t bytes = 0;
    rc = vcf_render_device(c, *g, seq_name, &bytes);
    if (rc) return rc;
    c->text_kind = 1;
    c->text_contig = contig;
    return MSIM_OK;
}

static int framed_text_ready(Ctx *c, int contig, uint32_t bpl, bool reuse) {
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (!g->applied) return fail(c, MSIM_ERR_ARG, "contig has not been applied");
    if (reuse && c->text_kind == 2 && c->text_contig == contig && c->text_bpl == bpl) return MSIM_OK;
    int rc = drain(c);
    if (rc) return rc;
    if (g->key_error) return key_error_of(c, *g);
    c->text_kind = 0;
    uint64_t bytes = 0;
    rc = fasta_frame_device(c, *g, bpl, &bytes);
    if (rc) return rc;
    c->text_kind = 2;
    c->text_contig = contig;
    c->text_bpl = bpl;
    return MSIM_OK;
}

int msim_render_vcf_device(msim_ctx *p, int contig, const char *seq_name, char *out, uint64_t cap, uint64_t *needed) {
    CTX_FLUSHED(c, p)
    if (!c || !seq_name || !needed) return MSIM_ERR_ARG;
    NEED_GPU(c);
    TraceRange tr("msim text: VCF lines");
    int rc = vcf_text_ready(c, contig, seq_name, out != nullptr);        // (a size call always renders)
    if (rc) return rc;
    return text_copy_out(c, reinterpret_cast<uint8_t *>(out), cap, needed);
}

int msim_fetch_sequence_framed(msim_ctx *p, int contig, uint32_t bpl, uint8_t *out, uint64_t cap, uint64_t *needed) {
    CTX_FLUSHED(c, p)
    if (!c || !needed || bpl == 0) return MSIM_ERR_ARG;
    NEED_GPU(c);
    TraceRange tr("msim text: framed FASTA");
    int rc = framed_text_ready(c, contig, bpl, out != nullptr);
    if (rc) return rc;
    return text_copy_out(c, out, cap, needed);
}

// The two texts queued for an output file (file_io.hip).  Rendering runs on the context's stream into a buffer of the output
// channel; the channel's thread copies and writes it while the caller goes on.
int msim_render_vcf_device_file(msim_ctx *p, int contig, const char *seq_name, int fd, uint64_t offset, uint64_t *written) {
    CTX_FLUSHED(c, p)
    if (!c || !seq_name || !written || fd < 0) return MSIM_ERR_ARG;
    NEED_GPU(c);
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (!g->planned) return fail(c, MSIM_ERR_ARG, "contig has not been planned");
    TraceRange tr("msim text: VCF lines -> file");
    int rc = file_check(c, fd);
    if (rc) return rc;
    rc = drain(c);                                                       // record aux bytes come from the emit stream
    if (rc) return rc;
    int slot;
    uint8_t **buf;
    size_t *cap;
    rc = file_text_buffer(c, 1, &slot, &buf, &cap);
    if (rc) return rc;
    uint64_t bytes = 0;
    rc = vcf_render_device(c, *g, seq_name, &bytes, buf, cap);
    if (rc) return rc;
    *written = bytes;
    return file_enqueue(c, 1, slot, bytes, fd, offset);
}

int msim_fetch_sequence_framed_file(msim_ctx *p, int contig, uint32_t bpl, int fd, uint64_t offset, uint64_t *written) {
    CTX_FLUSHED(c, p)
    if (!c || !written || bpl == 0 || fd < 0) return MSIM_ERR_ARG;
    NEED_GPU(c);
    Contig *g = get_contig(c, contig);
    if (!g) return MSIM_ERR_ARG;
    if (!g->applied) return fail(c, MSIM_ERR_ARG, "contig has not been applied");
    TraceRange tr("msim text: framed FASTA -> file");
    int rc = file_check(c, fd);
    if (rc) return rc;
    rc = drain(c);
    if (rc) return rc;
    if (g->key_error) return key_error_of(c, *g);
    int slot;
    uint8_t **buf;
    size_t *cap;
    rc = file_text_buffer(c, 0, &slot, &buf, &cap);
    if (rc) return rc;
    uint64_t bytes = 0;
    rc = fasta_frame_device(c, *g, bpl, &bytes, buf, cap);
    if (rc) return rc;
    *written = bytes;
    return file_enqueue(c, 0, slot, bytes, fd, offset);
}

int msim_file_wait(msim_ctx *p) {
    Ctx *c = C(p);
    if (!c) return MSIM_ERR_ARG;
    return file_wait(c);
}

int msim_add_contig_text(msim_ctx *p, const uint8_t *body, uint64_t body_bytes, uint64_t n_bases, uint32_t lenc,
                         uint32_t lenb, int *contig) {
    CTX_FLUSHED(c, p)
    if (!c || !contig || (!body && n_bases)) return MSIM_ERR_ARG;
    NEED_GPU(c);
    if (n_bases) {
        if (lenc == 0 || lenb < lenc) return fail(c, MSIM_ERR_ARG, "line width / line stride of the FASTA body are inconsistent");
        const uint64_t last = (n_bases - 1) / lenc * lenb + (n_bases - 1) % lenc;    // offset of the last base
        if (last >= body_bytes) return fail(c, MSIM_ERR_ARG, "FASTA body shorter than n_bases at this line width");
    }
    TraceRange tr("msim text: FASTA ingest");
    Contig *g;
    int rc = new_contig(c, n_bases, &g);
    if (rc) return rc;
    c->text_kind = 0;                                                    // the staging buffer is about to be reused
    rc = fasta_gather_device(c, body, body_bytes, n_bases, lenc, lenb, g->d_in + PAD);
    if (rc) return rc;
    *contig = (int)c->contigs.size() - 1;
    return MSIM_OK;
}

int msim_set_fast_key(msim_ctx *p, uint64_t key) {
    CTX_FLUSHED(c, p)
    if (!c) return MSIM_ERR_ARG;
    c->fast_key = key;
    c->fast_seq = 0;
    return MSIM_OK;
}

int msim_sample_min_distance(msim_ctx *p, int64_t start, int64_t stop, int64_t k, int64_t d, int64_t setsize, int64_t *out) {
    CTX_FLUSHED(c, p)
    if (!c || (!out && k > 0)) return MSIM_ERR_ARG;
    if (!c->host_only) {
        int rc = drain(c);
        if (rc) return rc;
        if (c->gpu && (rc = gpu_plan_sync_to_host(c, c->gpu))) return rc;      // continue from wherever the device streams stand
    }
    const uint64_t w0 = c->py.words;
    const int rc = sample_min_distance_host(c, start, stop, k, d, setsize, out);
    c->t.py_words += c->py.words - w0;
    return rc;
}

int msim_splice_contigs(msim_ctx *p, int a, int b, uint64_t n_bp, const uint64_t *bp_a, const uint64_t *bp_b, int *contig) {
    CTX_FLUSHED(c, p)
    if (!c || !contig || (n_bp && (!bp_a || !bp_b))) return MSIM_ERR_ARG;
    NEED_GPU(c);
    if (!get_contig(c, a) || (n_bp && !get_contig(c, b))) return MSIM_ERR_ARG;
    TraceRange tr("msim IT: splice two contigs");
    const uint64_t len_a = c->contigs[(size_t)a].len, len_b = n_bp ? c->contigs[(size_t)b].len : 0;
    if (n_bp >= (1ull << 31)) return fail(c, MSIM_ERR_UNSUPPORTED, "2^31 breakpoints or more");
    // segment i = [bp[i-1], bp[i]) of its contig, bp[-1] = 0, bp[n_bp] = the contig's length; even i from a, odd i from b
    const uint32_t n_seg = (uint32_t)n_bp + 1;
    std::vector<uint32_t> seg_out((size_t)n_seg + 1), seg_src(n_seg);
    uint64_t out = 0, pa = 0, pb = 0;
    for (uint32_t i = 0; i < n_seg; i++) {
        const uint64_t ea = i < n_bp ? bp_a[i] : len_a, eb = i < n_bp ? bp_b[i] : len_b;
        if (ea < pa || ea > len_a || (n_bp && (eb < pb || eb > len_b)))
            return fail(c, MSIM_ERR_ARG, "breakpoints must ascend and lie inside their contig");
        seg_out[i] = (uint32_t)out;
        seg_src[i] = (uint32_t)((i & 1u) ? pb : pa);
        out += (i & 1u) ? eb - pb : ea - pa;
        if (out >= (1ull << 32)) return fail(c, MSIM_ERR_UNSUPPORTED, "spliced contig of 4 GiB or more");
        pa = ea;
        pb = eb;
    }
    seg_out[n_seg] = (uint32_t)out;
    int rc = drain(c);
    if (rc) return rc;
    Contig *g;
    rc = new_contig(c, 0, &g);                             // (no input of its own: the result is its mutated stream)
    if (rc) return rc;
    const Contig &ga = c->contigs[(size_t)a];              // (after new_contig: the vector may have moved)
    const Contig *gb = n_bp ? &c->contigs[(size_t)b] : nullptr;
    c->text_kind = 0;
    rc = splice_device(c, ga, gb, seg_out.data(), seg_src.data(), n_seg, *g);
    if (rc) { (void)free_contig(c, *g, false); c->contigs.pop_back(); return rc; }
    g->planned = true;
    g->plan_empty = true;
    g->applied = true;
    *contig = (int)c->contigs.size() - 1;
    return MSIM_OK;
}

int msim_host_alloc(msim_ctx *p, uint64_t bytes, void **ptr) {
    Ctx *c = C(p);
    if (!c || !ptr) return MSIM_ERR_ARG;
    NEED_GPU(c);
    *ptr = nullptr;
    MSIM_HIP(c, hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return MSIM_OK;
}

int msim_host_free(msim_ctx *p, void *ptr) {
    Ctx *c = C(p);
    if (!c) return MSIM_ERR_ARG;
    NEED_GPU(c);
    if (ptr) MSIM_HIP(c, hipHostFree(ptr));
    return MSIM_OK;
}

// ---- test hooks: NOT part of the ABI (absent from include/msim.h).  They expose the two host-side chains of the
// device PLAN engines (plan_host.cpp) so that the CPU test tier can pin them against CPython's own random.sample /
// random.randint and run them under ASan/UBSan without a GPU (tests/test_cabi_host.py).
int msim_dbg_sample_ranges(msim_ctx *p, const msim_range *ranges, int n_ranges, const uint32_t *words, uint64_t n_words,
                           uint32_t *pos_out, uint64_t *consumed) {
    CTX_FLUSHED(c, p)
    if (!c || !consumed || (n_ranges && !ranges)) return MSIM_ERR_ARG;
    int64_t d = c->params.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, c->params.block[t]);
    size_t used = 0;
    const int rc = sample_ranges_host(c, ranges, n_ranges, d, words, (size_t)n_words, pos_out, &used);
    *consumed = used;
    return rc;
}

// cut_ranges_host: cut[] needs one slot per drawing range (k > 0) plus one, pool_pos the sum of k over pool-path ranges
int msim_dbg_cut_ranges(msim_ctx *p, const msim_range *ranges, int n_ranges, const uint32_t *words, uint64_t n_words,
                        uint32_t *cut, uint32_t *pool_pos, uint64_t *n_pool_pos, uint64_t *consumed) {
    CTX_FLUSHED(c, p)
    if (!c || !consumed || !n_pool_pos || !cut || (n_ranges && !ranges)) return MSIM_ERR_ARG;
    int64_t d = c->params.block[1];
    for (int t = 2; t <= 7; t++) d = std::min(d, c->params.block[t]);
    size_t used = 0, np = 0;
    const int rc = cut_ranges_host(c, ranges, n_ranges, d, words, (size_t)n_words, cut, pool_pos, &np, &used);
    *consumed = used; *n_pool_pos = np;
    return rc;
}

int msim_dbg_chain_boundary(msim_ctx *p, const msim_range *r, uint64_t L, const uint32_t *pos, const uint8_t *type,
                            uint64_t n, const uint32_t *words, uint64_t n_words, uint32_t *stop, uint64_t *consumed,
                            uint64_t *kept, int64_t *len_delta) {
    CTX_FLUSHED(c, p)
    if (!c || !r || !consumed || !kept || !len_delta) return MSIM_ERR_ARG;
    size_t used = 0, nk = 0;
    long long delta = 0;
    const int rc = chain_boundary_host(c, *r, L, pos, type, (size_t)n, words, (size_t)n_words, stop, &used, &nk, &delta);
    *consumed = used; *kept = nk; *len_delta = delta;
    return rc;
}

// The table walk (chain_boundary_tables) over tables built on the host from the same words: same arguments and results
// as msim_dbg_chain_boundary; MSIM_ERR_UNSUPPORTED when the range's lengths do not fit a table entry.
int msim_dbg_chain_boundary_tables(msim_ctx *p, const msim_range *r, uint64_t L, const uint32_t *pos, const uint8_t *type,
                                   uint64_t n, const uint32_t *words, uint64_t n_words, uint32_t *stop, uint64_t *consumed,
                                   uint64_t *kept, int64_t *len_delta) {
    CTX_FLUSHED(c, p)
    if (!c || !r || !consumed || !kept || !len_delta) return MSIM_ERR_ARG;
    ChainClasses cc;
    if (!chain_classes(*r, cc)) return MSIM_ERR_UNSUPPORTED;
    std::vector<uint32_t> T((size_t)(n_words + 1) << chain_lg_rows(cc));
    accept_tables_host(cc, words, (size_t)n_words, T.data());
    size_t used = 0, nk = 0;
    long long delta = 0;
    const int rc = chain_boundary_tables(c, *r, L, pos, type, (size_t)n, cc, T.data(), (size_t)n_words, stop, &used, &nk, &delta);
    *consumed = used; *kept = nk; *len_delta = delta;
    return rc;
}

// The general host-chain engine end to end on the host (multimix_plan_emulated): the record table and insert pool it
// arrives at, for comparison with msim_plan_contig's host planner on the same streams.  Call with recs == NULL for the sizes.
int msim_dbg_multimix_plan(msim_ctx *p, uint64_t L, const msim_range *ranges, int n_ranges, msim_record *recs, uint64_t cap_recs,
                           uint8_t *pool, uint64_t cap_pool, uint64_t *n_recs, uint64_t *pool_len, int *empty) {
    CTX_FLUSHED(c, p)
    if (!c || !n_recs || !pool_len || !empty || (n_ranges && !ranges)) return MSIM_ERR_ARG;
    if (!c->host_only) return fail(c, MSIM_ERR_ARG, "msim_dbg_multimix_plan needs a host-only context");
    HostPlan hp;
    const int rc = multimix_plan_emulated(c, L, ranges, n_ranges, hp);
    if (rc) return rc;
    *n_recs = hp.recs.size(); *pool_len = hp.pool.size(); *empty = hp.empty ? 1 : 0;
    if (recs) {
        if (cap_recs < hp.recs.size() || cap_pool < hp.pool.size()) return fail(c, MSIM_ERR_ARG, "buffers too small");
        if (!hp.recs.empty()) memcpy(recs, hp.recs.data(), hp.recs.size() * sizeof(msim_record));
        if (!hp.pool.empty() && pool) memcpy(pool, hp.pool.data(), hp.pool.size());
    }
    return MSIM_OK;
}

// The counter-based engine (MSIM_RNG_FAST) restated sequentially on the host (fast_plan_emulated): what msim_plan_contig of
// a fast context leaves for the same key, contig ordinal and ranges.  Works on a host-only context.  recs == NULL: sizes only.
int msim_dbg_fast_plan(msim_ctx *p, uint64_t L, const msim_range *ranges, int n_ranges, uint64_t key, uint32_t seq, msim_record *recs,
                       uint64_t cap_recs, uint8_t *pool, uint64_t cap_pool, uint64_t *n_recs, uint64_t *pool_len, int *empty) {
    CTX_FLUSHED(c, p)
    if (!c || !n_recs || !pool_len || !empty || (n_ranges && !ranges)) return MSIM_ERR_ARG;
    if (!c->have_params) return fail(c, MSIM_ERR_ARG, "msim_set_params has not been called");
    HostPlan hp;
    const int rc = fast_plan_emulated(c, L, ranges, n_ranges, key, seq, hp);
    if (rc) return rc;
    *n_recs = hp.recs.size(); *pool_len = hp.pool.size(); *empty = hp.empty ? 1 : 0;
    if (recs) {
        if (cap_recs < hp.recs.size() || cap_pool < hp.pool.size()) return fail(c, MSIM_ERR_ARG, "buffers too small");
        if (!hp.recs.empty()) memcpy(recs, hp.recs.data(), hp.recs.size() * sizeof(msim_record));
        if (!hp.pool.empty() && pool) memcpy(pool, hp.pool.data(), hp.pool.size());
    }
    return MSIM_OK;
}

// test support (tests/test_ahead_moments.py, CPU tier): the moments the anchored windows are laid out from; needs no context
int msim_dbg_stream_moments(uint64_t n, uint64_t k, uint64_t K, uint64_t ti_lim, double out[4]) {
    if (!out || !n || k >= n) return MSIM_ERR_ARG;
    const StreamMoments a = sample_words_moments(n, k), b = snp_words_moments(K, (unsigned long long)ti_lim);
    out[0] = a.e; out[1] = a.v; out[2] = b.e; out[3] = b.v;
    return MSIM_OK;
}

int msim_dbg_stream_status(msim_ctx *p, int out[8]) {
    CTX_FLUSHED(c, p)
    if (!c || !out || c->host_only || !c->gpu) return MSIM_ERR_ARG;
    gpu_plan_stream_status(c, c->gpu, out);
    return MSIM_OK;
}

// test support (tests/test_gpu_waits.py): `ms` milliseconds of a one-lane kernel that does nothing, on the plan stream
// (which = 0) or the emit stream (1) -- bounded by the device's 100 MHz wall clock, so the queue always drains -- for the
// deadline of the host waits (ctx.h: wait_stream / wait_event, MSIM_WAIT_TIMEOUT_S) to be met by a wait that is really kept waiting
__global__ void k_dbg_stall(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}
int msim_dbg_stall(msim_ctx *p, int which, uint32_t ms) {
    Ctx *c = C(p);
    if (!c || c->host_only || which < 0 || which > 1 || ms > 10000) return MSIM_ERR_ARG;
    hipLaunchKernelGGL(k_dbg_stall, dim3(1), dim3(1), 0, which ? c->emit_stream : c->stream, (unsigned long long)ms * 100000ull);
    MSIM_HIP(c, hipGetLastError());
    return MSIM_OK;
}

// test support (tools/hw_queue_probe.py): n more streams at a priority (-1 the plan stream's, 0 normal, 1 the lowest), each used
// once so that the runtime really gives it a hardware queue, kept until the process ends -- and the time of `launches` dependent
// one-lane launches on the context's plan stream (ns per launch).  What it shows: how a process's NUMBER of hardware queues
// changes the latency of every launch.
int msim_dbg_queue_probe(msim_ctx *p, int n_more, int prio, int launches, int idle_us, double *ns_per_launch) {
    Ctx *c = C(p);
    if (!c || c->host_only || n_more < 0 || n_more > 64 || launches < 1 || !ns_per_launch) return MSIM_ERR_ARG;
    int lo = 0, hi = 0;
    MSIM_HIP(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
    for (int i = 0; i < n_more; i++) {
        hipStream_t s = nullptr;                               // (deliberately never destroyed)
        MSIM_HIP(c, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio < 0 ? hi : prio > 0 ? lo : 0));
        hipLaunchKernelGGL(k_dbg_stall, dim3(1), dim3(1), 0, s, 0ull);
        MSIM_HIP(c, wait_stream(s));
    }
    for (int w = 0; w < 64; w++) hipLaunchKernelGGL(k_dbg_stall, dim3(1), dim3(1), 0, c->stream, 0ull);
    MSIM_HIP(c, wait_stream(c->stream));
    if (idle_us > 0) {
        // a BURST of four dependent launches after the queue has idled for idle_us (what a host-chain engine does between two
        // walks): launch -> completion of the burst, averaged over `launches` bursts
        double sum = 0;
        for (int i = 0; i < launches; i++) {
            std::this_thread::sleep_for(std::chrono::microseconds(idle_us));
            const auto b0 = std::chrono::steady_clock::now();
            for (int q = 0; q < 4; q++) hipLaunchKernelGGL(k_dbg_stall, dim3(1), dim3(1), 0, c->stream, 0ull);
            MSIM_HIP(c, wait_stream(c->stream));
            sum += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - b0).count();
        }
        *ns_per_launch = sum / launches;
        return MSIM_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < launches; i++) hipLaunchKernelGGL(k_dbg_stall, dim3(1), dim3(1), 0, c->stream, 0ull);
    MSIM_HIP(c, hipGetLastError());
    MSIM_HIP(c, wait_stream(c->stream));
    *ns_per_launch = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / launches;
    return MSIM_OK;
}

int msim_stats(msim_ctx *p, msim_timing *out) {
    CTX_FLUSHED(c, p)
    if (!c || !out) return MSIM_ERR_ARG;
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    *out = c->t;
    return MSIM_OK;
}

int msim_reset_stats(msim_ctx *p) {
    CTX_FLUSHED(c, p)
    if (!c) return MSIM_ERR_ARG;
    {
        int rc = drain(c);
        if (rc) return rc;
    }
    c->t = msim_timing{};
    return MSIM_OK;
}

}  // extern "C"

// ---- batch of small contigs ----------------------------------------------------------------------------------------
// An assembly with thousands of scaffolds pays a fixed ~0.35 ms of HIP API work per contig on the per-contig path
// (allocations, ~40 runtime calls, two or three synchronising fetches): 20 000 scaffolds of 10 kb ran at 27 Mbases/s.
// Here mutate()'s loop body (mutator.py:111-141) runs for MANY small contigs in one pass: the host strips the FASTA
// text and walks the RNG chain contig by contig (plan_contig_host -- these contigs are below every device engine's
// threshold anyway), the record tables are concatenated with their positions shifted to the contig's place in ONE
// super-contig, the GPU runs ONE APPLY over it (records never cross a contig border, so the rewrite kernels need no
// change), and the host frames the mutated stream per contig and renders the VCF lines (msim_render_vcf).
namespace msim {

struct Batch {
    struct Item { uint64_t base = 0, len = 0, out_start = 0, out_len = 0, rec0 = 0, nrec = 0, pool0 = 0, npool = 0, text0 = 0, ntext = 0,
                  vcf0 = 0, nvcf = 0; uint32_t bpl = 0; bool empty = true;
                  const char *header = nullptr; uint32_t header_len = 0; bool nl_before = false; };   // the record's defline (caller's memory)
    std::vector<Item> items;
    std::vector<msim_record> recs_rel;       // per-contig coordinates (what the VCF renderer reads)
    std::vector<uint8_t> pool;
    uint8_t *h_in = nullptr, *h_out = nullptr;   // page-aligned host staging (see pinned_reserve)
    size_t cap_in = 0, cap_out = 0;
    uint8_t *fasta = nullptr; size_t fasta_cap = 0, fasta_len = 0;   // malloc'ed: never zero-filled; framed on demand (msim_batch_view)
    bool framed = false;
    char *vcf = nullptr; size_t vcf_cap = 0, vcf_len = 0;
    uint64_t last_line_bases = 0;            // bases on the (partial) last line of the FASTA text
    int key_contig = -1;
};

static void batch_free(Ctx *c) {
    if (!c->batch) return;
    file_channel_idle(c, 0);                               // (an output channel may still be writing the batch's texts)
    file_channel_idle(c, 1);
    static const bool prof = getenv("MSIM_BATCH_PROF") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { const auto n = std::chrono::steady_clock::now(); if (prof) fprintf(stderr, "  batch_free %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(n - tp).count()); tp = n; };
    free(c->batch->h_in); lap("h_in");
    free(c->batch->h_out); lap("h_out");
    free(c->batch->fasta); lap("fasta");
    free(c->batch->vcf); lap("vcf");
    delete c->batch; lap("vectors");
    c->batch = nullptr;
}

template <class T> static bool raw_reserve(T **p, size_t *cap, size_t want) {
    if (*cap >= want) return true;
    free(*p);
    *cap = want + want / 8 + 4096;
    *p = static_cast<T *>(malloc(*cap));
    if (!*p) *cap = 0;
    return *p != nullptr;
}

// Contigs [0, n) in `parts` consecutive slices, one host thread each (text work of a batch: ingest, framing, VCF lines).
static int batch_threads() {
    static const int t = [] {
        if (const char *e = getenv("MSIM_BATCH_THREADS")) return std::max(1, atoi(e));
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(16, std::max<unsigned>(1, hw / 2));
    }();
    return t;
}
template <class F> static void parallel_slices(int n, F f) {   // f(part, i0, i1)
    const int parts = std::max(1, std::min(batch_threads(), n / 64));
    if (parts == 1) { f(0, 0, n); return; }
    std::vector<std::thread> th;
    int started = 1;                                       // slices [0, started) run or have run; the rest falls to this thread
    try {
        for (int t = 1; t < parts; t++, started++)
            th.emplace_back(f, t, (int)((long long)n * t / parts), (int)((long long)n * (t + 1) / parts));
    } catch (const std::system_error &) {                  // no more threads to be had: the remaining slices run here
    }
    f(0, 0, (int)((long long)n / parts));
    for (int t = started; t < parts; t++) f(t, (int)((long long)n * t / parts), (int)((long long)n * (t + 1) / parts));
    for (auto &x : th) x.join();
}

// Staging of a batch's bases on the host: plain page-aligned memory.  Page-locked memory does not pay here: hipHostMalloc
// pins at ~4 GB/s (90 ms for the 400 MB of a 16 k-contig batch, and as long again to release), while copies to and from
// touched pageable memory run at the link's speed on this platform and into fresh pages at 20 GB/s.
static int pinned_reserve(Ctx *c, uint8_t **p, size_t *cap, size_t want) {
    if (*cap >= want) return MSIM_OK;
    free(*p);
    *p = nullptr; *cap = 0;
    const size_t huge = (size_t)2 << 20;
    const size_t sz = (want + want / 4 + huge) & ~(huge - 1);
    *p = static_cast<uint8_t *>(aligned_alloc(huge, sz));
    if (!*p) return fail(c, MSIM_ERR_NOMEM, "batch staging buffer");
#ifdef MADV_HUGEPAGE
    (void)madvise(*p, sz, MADV_HUGEPAGE);                  // 2 MB pages where the kernel grants them: 512x fewer faults and unmaps
#endif
    *cap = sz;
    return MSIM_OK;
}

static inline size_t batch_header_len(const msim_batch_contig &q) { return q.header_len ? q.header_len : strlen(q.header); }

// output length change of one record (apply.hip: rec_lengths)
static long long rec_delta(const msim_record &r) {
    const long long len = (long long)r.stop - (long long)r.pos + 1;
    switch (r.type) {
        case MSIM_IN: return len;
        case MSIM_DE: case MSIM_TL: return -len;
        case MSIM_DU: return len;
        case MSIM_TLI: return r.stop + 1 > r.extra ? (long long)r.stop + 1 - (long long)r.extra : 0;
        default: return 0;
    }
}

// FASTA framing of a batch's mutated bases into `dst` (B.fasta_len bytes): '\n' after every bpl bases, none after a partial
// last line (fasta_writer.py:40-58).  With a header per contig the text is the complete run of FASTA records as FastaWriter
// would have written them: '>' header '\n' body, and a '\n' before a header iff the previous body ended in a partial line.
static void batch_frame_into(const Batch &B, uint8_t *out) {
    parallel_slices((int)B.items.size(), [&](int, int i0, int i1) {
        for (int i = i0; i < i1; i++) {
            const Batch::Item &it = B.items[(size_t)i];
            const uint8_t *src = B.h_out + it.out_start;
            uint8_t *dst = out + it.text0;
            if (it.header) {
                if (it.nl_before) *dst++ = '\n';
                *dst++ = '>';
                memcpy(dst, it.header, it.header_len);
                dst += it.header_len;
                *dst++ = '\n';
            }
            if (!it.bpl) { if (it.out_len) memcpy(dst, src, it.out_len); continue; }
            uint64_t done = 0;
            while (done + it.bpl <= it.out_len) { memcpy(dst, src + done, it.bpl); dst[it.bpl] = '\n'; dst += it.bpl + 1; done += it.bpl; }
            if (done < it.out_len) memcpy(dst, src + done, it.out_len - done);
        }
    });
}

}  // namespace msim

extern "C" {

int msim_batch_run(msim_ctx *p, const msim_batch_contig *contigs, int n) {
    CTX_FLUSHED(c, p)
    if (!c || !contigs || n < 1) return MSIM_ERR_ARG;
    NEED_GPU(c);
    if (!c->have_params) return fail(c, MSIM_ERR_ARG, "msim_set_params has not been called");
    TraceRange tr("msim batch of small contigs");
    static const bool prof = getenv("MSIM_BATCH_PROF") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_0 = now();
    int rc = drain(c);
    if (rc) return rc;
    if (c->gpu) {                                          // the host planner continues from wherever the device streams stand
        rc = gpu_plan_sync_to_host(c, c->gpu);
        if (rc) return rc;
    }
    if (!c->batch) c->batch = new Batch();
    Batch &B = *c->batch;
    B.items.assign((size_t)n, Batch::Item());
    B.recs_rel.clear(); B.pool.clear(); B.fasta_len = 0; B.vcf_len = 0; B.last_line_bases = 0; B.framed = false;
    B.key_contig = -1;
    uint64_t total = 0;
    for (int i = 0; i < n; i++) {
        const msim_batch_contig &q = contigs[i];
        if ((!q.body && q.n_bases) || (q.n_ranges && !q.ranges) || q.n_ranges < 0 || !q.name) return fail(c, MSIM_ERR_ARG, "msim_batch_run: bad contig");
        if (q.n_bases) {
            if (q.lenc == 0 || q.lenb < q.lenc) return fail(c, MSIM_ERR_ARG, "line width / line stride of the FASTA body are inconsistent");
            const uint64_t last = (q.n_bases - 1) / q.lenc * q.lenb + (q.n_bases - 1) % q.lenc;
            if (last >= q.body_bytes) return fail(c, MSIM_ERR_ARG, "FASTA body shorter than n_bases at this line width");
        }
        B.items[(size_t)i].base = total;
        B.items[(size_t)i].len = q.n_bases;
        B.items[(size_t)i].bpl = q.lenc;
        total += q.n_bases;
    }
    if (total >= (1ull << 31)) return fail(c, MSIM_ERR_UNSUPPORTED, "batch of 2 GiB or more: split it");
    // ---- 1. FASTA text -> upper-cased bases (pyfaidx's sequence_always_upper, util.py:84-88) on the host and up to the
    //         device -- on a thread of its own, beside PLAN: the RNG chain never looks at a base
    rc = pinned_reserve(c, &B.h_in, &B.cap_in, total + 64);
    if (rc) return rc;
    Contig *g;
    rc = new_contig(c, total, &g);
    if (rc) return rc;
    const int gid = (int)c->contigs.size() - 1;
    // the temporary super-contig leaves the context on EVERY way out of this block (HIP failures included: a contig left
    // behind would leak its device buffers and shift the ids of the caller's later contigs)
    struct Drop {
        Ctx *c; int gid; bool armed = true;
        void now() { if (armed) { armed = false; (void)free_contig(c, c->contigs[(size_t)gid], false); c->contigs.pop_back(); } }
        ~Drop() { now(); }
    } guard{c, gid};
    auto drop = [&]() { guard.now(); };
    struct Side {                                           // a helper thread that is always joined before its captures die
        std::thread th;
        void join() { if (th.joinable()) th.join(); }
        ~Side() { join(); }
    };
    hipError_t ingest_err = hipSuccess;
    double ingest_ms = 0;
    uint8_t *const d_in = g->d_in;                          // (c->contigs does not grow during the batch: g stays valid)
    auto ingest = [&, d_in]() {
        const auto a0 = now();
        parallel_slices(n, [&](int, int i0, int i1) {
            for (int i = i0; i < i1; i++) {
                const msim_batch_contig &q = contigs[i];
                uint8_t *dst = B.h_in + B.items[(size_t)i].base;
                uint64_t done = 0;
                const uint8_t *src = q.body;
                while (done < q.n_bases) {
                    const uint64_t take = std::min<uint64_t>(q.lenc, q.n_bases - done);
                    memcpy(dst + done, src, take);
                    done += take;
                    src += q.lenb;
                }
                for (uint64_t k = 0; k < q.n_bases; k++) {       // (auto-vectorised)
                    const uint8_t b = dst[k];
                    dst[k] = (uint8_t)(b - ((b >= 'a' && b <= 'z') ? 32 : 0));
                }
            }
        });
        ingest_ms = ms(a0, now());
        if (total) {
            ingest_err = hipSetDevice(c->device);
            if (ingest_err == hipSuccess) ingest_err = hipMemcpyAsync(d_in + PAD, B.h_in, total, hipMemcpyHostToDevice, c->stream);
        }
    };
    Side side_in;
    try { side_in.th = std::thread(ingest); } catch (const std::system_error &) { ingest(); }
    const auto t_1 = now();
    for (int i = 0; i < n; i++) {                            // (kept for the framing, which runs when the text is asked for)
        Batch::Item &it = B.items[(size_t)i];
        it.header = contigs[i].header;
        it.header_len = contigs[i].header ? (uint32_t)batch_header_len(contigs[i]) : 0u;
    }
    // ---- 2. PLAN: the RNG chain, contig by contig (mutator.py:111-131 + the draws of :334-358)
    std::vector<msim_record> recs_abs;
    long long delta_total = 0;
    bool all_snp = true;
    HostPlan hp;
    for (int i = 0; i < n; i++) {
        const msim_batch_contig &q = contigs[i];
        Batch::Item &it = B.items[(size_t)i];
        const uint64_t w_py = c->py.words, w_np = c->np.words;
        rc = plan_contig_host(c, q.n_bases, q.ranges, q.n_ranges, hp);
        if (rc) return rc;
        c->t.contigs_batch++;
        c->t.py_words += c->py.words - w_py;
        c->t.np_words += c->np.words - w_np;
        it.empty = hp.empty;
        it.rec0 = B.recs_rel.size(); it.nrec = hp.recs.size();
        it.pool0 = B.pool.size(); it.npool = hp.pool.size();
        long long delta = 0;
        for (const msim_record &r : hp.recs) {
            delta += rec_delta(r);
            msim_record a = r;
            a.pos += (uint32_t)it.base;
            a.stop += (uint32_t)it.base;
            if (r.type == MSIM_IN) a.extra += (uint32_t)it.pool0;
            else if (r.type == MSIM_TLI) a.extra += (uint32_t)it.base;
            if (r.type != MSIM_SN) all_snp = false;
            recs_abs.push_back(a);
        }
        B.recs_rel.insert(B.recs_rel.end(), hp.recs.begin(), hp.recs.end());
        B.pool.insert(B.pool.end(), hp.pool.begin(), hp.pool.end());
        it.out_start = (uint64_t)((long long)it.base + delta_total);
        it.out_len = (uint64_t)((long long)it.len + delta);
        delta_total += delta;
    }
    if (B.pool.size() >= (1ull << 31)) return fail(c, MSIM_ERR_UNSUPPORTED, "batch insert pool of 2 GiB or more: split it");
    const uint64_t out_total = (uint64_t)((long long)total + delta_total);
    const auto t_2 = now();
    // ---- 3. ONE APPLY over the super-contig
    side_in.join();
    if (ingest_err != hipSuccess) { drop(); return hip_fail(c, ingest_err, "batch input upload"); }
    // ---- 3b. VCF record lines (mutator.py:334-399 + vcf_writer.py:118-126) per contig, on the host -- beside the APPLY: they
    //          read the records and the INPUT bases only
    int vcf_rc = MSIM_OK;
    const char *vcf_msg = "";
    double vcf_ms = 0;
    auto render_vcf = [&]() {
        const auto v0 = now();
    // VCF lines: every slice of contigs renders into a buffer of its own, sized by a cheap upper bound (a line is name +
        // fixed fields of < 96 bytes + REF and ALT, each at most span + insert + 1 long, twice for a duplication's ALT);
        // the slices are then joined in contig order
        const msim_record *all = B.recs_rel.data();
        const int max_parts = batch_threads();
        std::vector<char *> part_buf((size_t)max_parts, nullptr);      // malloc'ed: never zero-filled
        std::vector<uint64_t> part_len((size_t)max_parts, 0);
        std::vector<int> part_first((size_t)max_parts, -1);
        std::atomic<bool> bad{false};
        parallel_slices(n, [&](int part, int i0, int i1) {
            uint64_t bound = 0;
            for (int i = i0; i < i1; i++) {
                const Batch::Item &it = B.items[(size_t)i];
                const uint64_t nl = (contigs[i].name_len ? contigs[i].name_len : strlen(contigs[i].name)) + 96;
                for (uint64_t k = 0; k < it.nrec; k++) {
                    const msim_record &r = all[it.rec0 + k];
                    const uint64_t span = r.type == MSIM_SN ? 1 : (uint64_t)(r.stop >= r.pos ? r.stop - r.pos + 1 : 0) + 2;
                    const uint64_t tli = r.type == MSIM_TLI && r.stop + 1 > r.extra ? (uint64_t)r.stop + 1 - r.extra : 0;
                    bound += nl + 3 * (span + tli) + 8;
                }
            }
            char *out = static_cast<char *>(malloc((size_t)bound + 64));
            part_buf[(size_t)part] = out;
            part_first[(size_t)part] = i0;
            if (!out) { bad = true; return; }
            uint64_t at = 0;
            for (int i = i0; i < i1; i++) {
                Batch::Item &it = B.items[(size_t)i];
                it.vcf0 = at;                                  // relative to the slice until the join
                uint64_t need = 0;
                if (it.nrec) {
                    const std::string nm = contigs[i].name_len ? std::string(contigs[i].name, contigs[i].name_len) : std::string();
                    need = render_vcf_unchecked(all + it.rec0, it.nrec, B.pool.data() + it.pool0, B.h_in + it.base, it.len,
                                                contigs[i].name_len ? nm.c_str() : contigs[i].name, out + at);
                    if (at + need > bound) bad = true;
                }
                it.nvcf = need;
                at += need;
            }
            part_len[(size_t)part] = at;
        });
        auto free_parts = [&]() { for (char *q : part_buf) free(q); };
        if (bad) { free_parts(); vcf_rc = MSIM_ERR_HIP; vcf_msg = "internal: VCF text of a batch: bound too small or out of memory"; return; }
        // join in slice order (slices are consecutive contig ranges, part index ascending)
        std::vector<int> order;
        for (int t = 0; t < max_parts; t++) if (part_first[(size_t)t] >= 0) order.push_back(t);
        std::sort(order.begin(), order.end(), [&](int x, int y) { return part_first[(size_t)x] < part_first[(size_t)y]; });
        uint64_t total_vcf = 0;
        for (int t : order) total_vcf += part_len[(size_t)t];
        file_channel_idle(c, 1);                            // (the previous batch's VCF text may still be on its way to the file)
        if (!raw_reserve(&B.vcf, &B.vcf_cap, (size_t)total_vcf + 1)) { free_parts(); vcf_rc = MSIM_ERR_NOMEM; vcf_msg = "batch VCF text"; return; }
        uint64_t at = 0;
        for (size_t k = 0; k < order.size(); k++) {
            const int t = order[k];
            const int i0 = part_first[(size_t)t], i1 = k + 1 < order.size() ? part_first[(size_t)order[k + 1]] : n;
            for (int i = i0; i < i1; i++) B.items[(size_t)i].vcf0 += at;
            if (part_len[(size_t)t]) memcpy(B.vcf + at, part_buf[(size_t)t], (size_t)part_len[(size_t)t]);
            at += part_len[(size_t)t];
        }
        B.vcf_len = (size_t)at;
        free_parts();
        vcf_ms = ms(v0, now());
    };
    Side side_vcf;
    try { side_vcf.th = std::thread(render_vcf); } catch (const std::system_error &) { render_vcf(); }
    g->n_rec = recs_abs.size();
    g->pool_len = B.pool.size();
    g->plan_empty = recs_abs.empty();
    g->all_snp = all_snp;
    g->delta_known = true;
    g->known_delta = delta_total;
    if (g->n_rec) {
        rc = dev_reserve(c, (void **)&g->d_recs, &g->cap_recs, g->n_rec * sizeof(msim_record));
        if (rc) return rc;
        MSIM_HIP(c, hipMemcpyAsync(g->d_recs, recs_abs.data(), g->n_rec * sizeof(msim_record), hipMemcpyHostToDevice, c->stream));
    }
    rc = dev_reserve(c, (void **)&g->d_pool, &g->cap_pool, g->pool_len + 2 * PAD);
    if (rc) return rc;
    if (g->pool_len) MSIM_HIP(c, hipMemcpyAsync(g->d_pool + PAD, B.pool.data(), g->pool_len, hipMemcpyHostToDevice, c->stream));
    MSIM_HIP(c, wait_stream(c->stream));          // (the uploads read pageable vectors; APPLY runs on the emit stream)
    g->planned = true;
    rc = apply_contig_device(c, *g);
    if (rc) return rc;
    rc = pinned_reserve(c, &B.h_out, &B.cap_out, out_total + 64);
    if (rc) return rc;
    if (out_total) MSIM_HIP(c, hipMemcpyAsync(B.h_out, g->d_out, out_total, hipMemcpyDeviceToHost, c->emit_stream));
    rc = apply_finish(c);                                  // synchronises the emit stream: the copy above included
    if (rc) return rc;
    if (g->out_len != out_total) return fail(c, MSIM_ERR_HIP, "internal: batch planner and device disagree on the mutated length");
    if (g->key_error) {                                    // the reference's KeyError: which contig, which base
        const uint64_t kp = g->key_pos;
        int who = 0;
        for (int i = 0; i < n; i++) if (B.items[(size_t)i].base <= kp && kp < B.items[(size_t)i].base + B.items[(size_t)i].len) who = i;
        B.key_contig = who;
        const uint8_t kb = g->key_base;
        return fail(c, MSIM_ERR_KEY, std::string("KeyError: '") + (char)kb + "'");
    }
    drop();
    const auto t_3 = now();
    // ---- 4. the FASTA text's layout; the framing itself runs when the text is asked for, straight into the caller's
    //         memory (msim_batch_fetch) or into a buffer of the context (msim_batch_view)
    uint64_t text_total = 0;
    for (int i = 0; i < n; i++) {
        Batch::Item &it = B.items[(size_t)i];
        it.nl_before = i > 0 && B.items[(size_t)i - 1].bpl && B.items[(size_t)i - 1].out_len % B.items[(size_t)i - 1].bpl;
        const uint64_t head = it.header ? (it.nl_before ? 1 : 0) + 1 + it.header_len + 1 : 0;
        it.text0 = text_total;
        it.ntext = head + (it.bpl ? it.out_len + it.out_len / it.bpl : it.out_len);
        text_total += it.ntext;
    }
    B.fasta_len = (size_t)text_total;
    B.last_line_bases = B.items.back().bpl ? B.items.back().out_len % B.items.back().bpl : 0;
    side_vcf.join();
    if (vcf_rc) return fail(c, vcf_rc, vcf_msg);
    if (prof)
        fprintf(stderr, "msim_batch_run: %d contigs, %.1f Mb in %.1f ms: plan %.1f (beside it: ingest %.1f), upload+APPLY+download %.1f (beside it: VCF %.1f), wait for VCF %.1f\n",
                n, total / 1e6, ms(t_0, now()), ms(t_1, t_2), ingest_ms, ms(t_2, t_3), vcf_ms, ms(t_3, now()));
    return MSIM_OK;
}

int msim_batch_sizes(msim_ctx *p, int n, uint64_t *fasta_bytes, uint64_t *vcf_bytes, int32_t *empty, uint64_t *n_records) {
    CTX_FLUSHED(c, p)
    if (!c || !c->batch || (size_t)n != c->batch->items.size()) return MSIM_ERR_ARG;
    for (int i = 0; i < n; i++) {
        const Batch::Item &it = c->batch->items[(size_t)i];
        if (fasta_bytes) fasta_bytes[i] = it.ntext;
        if (vcf_bytes) vcf_bytes[i] = it.nvcf;
        if (empty) empty[i] = it.empty ? 1 : 0;
        if (n_records) n_records[i] = it.nrec;
    }
    return MSIM_OK;
}

int msim_batch_fetch(msim_ctx *p, uint8_t *fasta_text, uint64_t fasta_cap, char *vcf_text, uint64_t vcf_cap) {
    CTX_FLUSHED(c, p)
    if (!c || !c->batch) return MSIM_ERR_ARG;
    Batch &B = *c->batch;
    if (fasta_text) {                                      // framed straight into the caller's memory (a mapped file, say)
        if (fasta_cap < B.fasta_len) return fail(c, MSIM_ERR_ARG, "fasta buffer too small");
        if (B.fasta_len && B.framed) memcpy(fasta_text, B.fasta, B.fasta_len);
        else if (B.fasta_len) batch_frame_into(B, fasta_text);
    }
    if (vcf_text) {
        if (vcf_cap < B.vcf_len) return fail(c, MSIM_ERR_ARG, "vcf buffer too small");
        if (B.vcf_len) memcpy(vcf_text, B.vcf, B.vcf_len);
    }
    return MSIM_OK;
}

int msim_batch_fetch_file(msim_ctx *p, int fasta_fd, uint64_t fasta_offset, int vcf_fd, uint64_t vcf_offset) {
    CTX_FLUSHED(c, p)
    if (!c || !c->batch) return MSIM_ERR_ARG;
    Batch &B = *c->batch;
    if (fasta_fd >= 0) {
        int rc = file_check(c, fasta_fd);
        if (rc) return rc;
        file_channel_idle(c, 0);                           // (the previous batch's text may still be read from B.fasta)
        if (!B.framed) {
            if (!raw_reserve(&B.fasta, &B.fasta_cap, B.fasta_len + 1)) return fail(c, MSIM_ERR_NOMEM, "batch FASTA text");
            if (B.fasta_len) batch_frame_into(B, B.fasta);
            B.framed = true;
        }
        rc = file_enqueue_host(c, 0, B.fasta, B.fasta_len, fasta_fd, fasta_offset);
        if (rc) return rc;
    }
    if (vcf_fd >= 0) {
        int rc = file_check(c, vcf_fd);
        if (rc) return rc;
        rc = file_enqueue_host(c, 1, reinterpret_cast<const uint8_t *>(B.vcf), B.vcf_len, vcf_fd, vcf_offset);
        if (rc) return rc;
    }
    return MSIM_OK;
}

int msim_batch_view(msim_ctx *p, const uint8_t **fasta_text, uint64_t *fasta_bytes, const char **vcf_text, uint64_t *vcf_bytes,
                    uint64_t *last_line_bases) {
    CTX_FLUSHED(c, p)
    if (!c || !c->batch) return MSIM_ERR_ARG;
    Batch &B = *c->batch;
    if (fasta_text && !B.framed) {
        file_channel_idle(c, 0);                           // (the previous batch's text may still be read from B.fasta)
        if (!raw_reserve(&B.fasta, &B.fasta_cap, B.fasta_len + 1)) return fail(c, MSIM_ERR_NOMEM, "batch FASTA text");
        if (B.fasta_len) batch_frame_into(B, B.fasta);
        B.framed = true;
    }
    if (fasta_text) *fasta_text = B.fasta;
    if (fasta_bytes) *fasta_bytes = B.fasta_len;
    if (vcf_text) *vcf_text = B.vcf;
    if (vcf_bytes) *vcf_bytes = B.vcf_len;
    if (last_line_bases) *last_line_bases = B.last_line_bases;
    return MSIM_OK;
}

int msim_batch_key_contig(msim_ctx *p, int *contig) {
    CTX_FLUSHED(c, p)
    if (!c || !contig || !c->batch) return MSIM_ERR_ARG;
    *contig = c->batch->key_contig;
    return MSIM_OK;
}

}  // extern "C"
