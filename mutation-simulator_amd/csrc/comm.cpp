// Multi-GPU exchange step of libmsim: gather of the mutated contigs to one rank over RCCL (xGMI inside a node).
//
// One process per GPU.  PLAN is replayed on every rank (the two MT19937 streams chain across contigs, so it cannot
// shard in bit-compatible mode); APPLY is sharded by contig (mutate()'s loop, mutator.py:111-141, is independent
// per contig once the records exist).  The one collective north_star names is the final gather: every peer sends
// its mutated contigs straight to the root over its own direct xGMI link -- grouped ncclSend / ncclRecv, sizes
// differ per rank, a ring collective would be bound by one link.
//
// librccl is loaded with dlopen at msim_comm_init: single-GPU users of libmsim never need it, and a process that
// already carries an RCCL (e.g. one that imported torch) shares that copy.  No torch type crosses the ABI; the
// ncclUniqueId travels as 128 opaque bytes over whatever control plane the caller has (bench.py: gloo).
#include <dlfcn.h>

#include <cstring>
#include <vector>

#include "ctx.h"

namespace msim {

namespace {

typedef void *nccl_comm_t;
struct nccl_uid { char internal[128]; };
constexpr int NCCL_UINT8 = 1;                              // ncclUint8 (rccl.h)

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;
std::string g_rccl_error;

bool load_rccl() {
    if (g_rccl.lib) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { g_rccl_error = std::string("librccl not found: ") + dlerror(); return false; }
    Rccl r;
    r.lib = h;
#define MSIM_SYM(field, name)                                                             \
    *reinterpret_cast<void **>(&r.field) = dlsym(h, name);                                \
    if (!r.field) { g_rccl_error = std::string("librccl lacks ") + name; return false; }
    MSIM_SYM(GetUniqueId, "ncclGetUniqueId")
    MSIM_SYM(CommInitRank, "ncclCommInitRank")
    MSIM_SYM(CommDestroy, "ncclCommDestroy")
    MSIM_SYM(GroupStart, "ncclGroupStart")
    MSIM_SYM(GroupEnd, "ncclGroupEnd")
    MSIM_SYM(Send, "ncclSend")
    MSIM_SYM(Recv, "ncclRecv")
    MSIM_SYM(GetErrorString, "ncclGetErrorString")
#undef MSIM_SYM
    g_rccl = r;
    return true;
}

}  // namespace

struct Comm {
    nccl_comm_t comm = nullptr;
    int rank = 0, world = 1;
    std::vector<uint8_t *> recv;                           // root: one grow-only buffer per gathered contig slot
    std::vector<size_t> cap;
};

void comm_destroy(Ctx *c) {
    if (!c->comm) return;
    for (uint8_t *p : c->comm->recv) if (p) (void)hipFree(p);
    if (c->comm->comm && g_rccl.lib) (void)g_rccl.CommDestroy(c->comm->comm);
    delete c->comm;
    c->comm = nullptr;
}

}  // namespace msim

using namespace msim;

extern "C" {

int msim_comm_unique_id(uint8_t id[128]) {
    if (!id) return MSIM_ERR_ARG;
    if (!load_rccl()) return fail(nullptr, MSIM_ERR_HIP, g_rccl_error);
    nccl_uid u;
    const int rc = g_rccl.GetUniqueId(&u);
    if (rc) return fail(nullptr, MSIM_ERR_HIP, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(rc));
    memcpy(id, u.internal, sizeof u.internal);
    return MSIM_OK;
}

int msim_comm_init(msim_ctx *p, const uint8_t id[128], int rank, int world) {
    Ctx *c = reinterpret_cast<Ctx *>(p);
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return MSIM_ERR_ARG;
    if (c->host_only) return fail(c, MSIM_ERR_HIP, "host-only context: this call needs the GPU");
    if (!load_rccl()) return fail(c, MSIM_ERR_HIP, g_rccl_error);
    comm_destroy(c);
    MSIM_HIP(c, hipSetDevice(c->device));
    Comm *m = new Comm();
    m->rank = rank;
    m->world = world;
    nccl_uid u;
    memcpy(u.internal, id, sizeof u.internal);
    const int rc = g_rccl.CommInitRank(&m->comm, world, u, rank);
    if (rc) {
        delete m;
        return fail(c, MSIM_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc));
    }
    c->comm = m;
    return MSIM_OK;
}

int msim_comm_destroy(msim_ctx *p) {
    Ctx *c = reinterpret_cast<Ctx *>(p);
    if (!c) return MSIM_ERR_ARG;
    comm_destroy(c);
    return MSIM_OK;
}

// The bookkeeping of the gather, separated from the transport so that it can be checked without GPUs: which
// transfers does `rank` post, in which order?  A slot has three PARTS -- 0 the mutated stream, 1 its record table (16 bytes
// per record: the binary VCF, mutator.py:334-421), 2 its insert pool -- and op = {kind (0 send, 1 recv, 2 local), slot, part,
// peer, bytes}.  Empty parts are not transferred.  The order is (slot, part) on both sides of every pair, which is what makes
// grouped point-to-point calls match up.  n_records / pool_len may be NULL: streams only.
int msim_gather_plan(int n, const int *owner, const uint64_t *out_len, const uint64_t *n_records, const uint64_t *pool_len, int rank,
                     int world, int root, int64_t *ops /* 5 per op, capacity 3 n */, int *n_ops) {
    if (n < 0 || (n && (!owner || !out_len)) || !ops || !n_ops || world < 1 || root < 0 || root >= world ||
        rank < 0 || rank >= world || (!n_records) != (!pool_len))
        return MSIM_ERR_ARG;
    int k = 0;
    for (int i = 0; i < n; i++) {
        if (owner[i] < 0 || owner[i] >= world) return MSIM_ERR_ARG;
        int kind = -1, peer = -1;
        if (rank == root) { kind = owner[i] == root ? 2 : 1; peer = owner[i]; }
        else if (owner[i] == rank) { kind = 0; peer = root; }
        if (kind < 0) continue;
        const uint64_t bytes[3] = {out_len[i], n_records ? n_records[i] * sizeof(msim_record) : 0, pool_len ? pool_len[i] : 0};
        for (int part = 0; part < 3; part++) {
            if (part && !bytes[part]) continue;            // (the stream's op is always there: it carries the slot, even when empty)
            ops[5 * k] = kind; ops[5 * k + 1] = i; ops[5 * k + 2] = part; ops[5 * k + 3] = peer; ops[5 * k + 4] = (int64_t)bytes[part];
            k++;
        }
    }
    *n_ops = k;
    return MSIM_OK;
}

int msim_gather_to_root(msim_ctx *p, int n, const int *contig_ids, const int *owner, const uint64_t *out_len, const uint64_t *n_records,
                        const uint64_t *pool_len, int root, uint64_t *device_addrs) {
    Ctx *c = reinterpret_cast<Ctx *>(p);
    if (!c || n < 0 || (n && (!contig_ids || !owner || !out_len)) || (!n_records) != (!pool_len)) return MSIM_ERR_ARG;
    if (c->host_only) return fail(c, MSIM_ERR_HIP, "host-only context: this call needs the GPU");
    Comm *m = c->comm;
    const int rank = m ? m->rank : 0, world = m ? m->world : 1;
    if (!m && n) {
        for (int i = 0; i < n; i++)
            if (owner[i] != 0) return fail(c, MSIM_ERR_ARG, "msim_gather_to_root before msim_comm_init");
    }
    TraceRange tr("msim gather to root (RCCL)");
    std::vector<int64_t> ops((size_t)15 * (n ? n : 1));
    int n_ops = 0;
    int rc = msim_gather_plan(n, owner, out_len, n_records, pool_len, rank, world, root, ops.data(), &n_ops);
    if (rc) return fail(c, rc, "msim_gather_to_root: bad owner / root");
    // everything this rank produced has to exist before it is sent: collect the asynchronous APPLYs
    rc = msim_sync(p);
    if (rc) return rc;
    if (device_addrs) for (int i = 0; i < 3 * n; i++) device_addrs[i] = 0;
    if (m && rank == root && m->recv.size() < (size_t)3 * n) { m->recv.resize((size_t)3 * n, nullptr); m->cap.resize((size_t)3 * n, 0); }
    bool any_remote = false;
    auto local_ptr = [&](Contig &g, int part) -> uint8_t * {
        return part == 0 ? g.d_out : part == 1 ? reinterpret_cast<uint8_t *>(g.d_recs) : g.d_pool + PAD;
    };
    for (int k = 0; k < n_ops; k++) {                      // buffers first: no allocation inside the group
        const int kind = (int)ops[5 * k], slot = (int)ops[5 * k + 1], part = (int)ops[5 * k + 2];
        const uint64_t bytes = (uint64_t)ops[5 * k + 4];
        const size_t at = (size_t)3 * slot + part;
        if (kind == 1) {
            any_remote = true;
            if (m->cap[at] < bytes + PAD) {
                if (m->recv[at]) MSIM_HIP(c, hipFree(m->recv[at]));
                m->recv[at] = nullptr; m->cap[at] = 0;
                const size_t sz = bytes + (bytes >> 4) + PAD;
                MSIM_HIP(c, hipMalloc(&m->recv[at], sz));
                m->cap[at] = sz;
            }
            if (device_addrs) device_addrs[at] = (uint64_t)(uintptr_t)m->recv[at];
        } else {
            const int cid = contig_ids[slot];
            if (cid < 0 || (size_t)cid >= c->contigs.size()) return fail(c, MSIM_ERR_ARG, "no such contig");
            Contig &g = c->contigs[(size_t)cid];
            if (!g.applied) return fail(c, MSIM_ERR_ARG, "gather of a contig this rank has not applied");
            if (part == 0 && g.out_len != bytes) return fail(c, MSIM_ERR_ARG, "gather: out_len disagrees with the applied contig");
            if (part == 1 && g.n_rec * sizeof(msim_record) != bytes) return fail(c, MSIM_ERR_ARG, "gather: n_records disagrees with the contig's table");
            if (part == 2 && g.pool_len != bytes) return fail(c, MSIM_ERR_ARG, "gather: pool_len disagrees with the contig's insert pool");
            if (kind == 0) any_remote = true;
            if (device_addrs) device_addrs[at] = (uint64_t)(uintptr_t)local_ptr(g, part);
        }
    }
    if (!any_remote) return MSIM_OK;
    hipStream_t st = c->emit_stream;
    int nrc = g_rccl.GroupStart();
    for (int k = 0; k < n_ops && !nrc; k++) {
        const int kind = (int)ops[5 * k], slot = (int)ops[5 * k + 1], part = (int)ops[5 * k + 2], peer = (int)ops[5 * k + 3];
        const size_t bytes = (size_t)ops[5 * k + 4];
        if (!bytes) continue;
        if (kind == 0) nrc = g_rccl.Send(local_ptr(c->contigs[(size_t)contig_ids[slot]], part), bytes, NCCL_UINT8, peer, m->comm, st);
        else if (kind == 1) nrc = g_rccl.Recv(m->recv[(size_t)3 * slot + part], bytes, NCCL_UINT8, peer, m->comm, st);
    }
    const int erc = g_rccl.GroupEnd();
    if (!nrc) nrc = erc;
    if (nrc) return fail(c, MSIM_ERR_HIP, std::string("RCCL gather: ") + g_rccl.GetErrorString(nrc));
    MSIM_HIP(c, wait_stream(st));
    return MSIM_OK;
}

// Copy `bytes` bytes at a device address msim_gather_to_root reported (a gathered part on the root, a contig's own buffer
// elsewhere) to the host: how the root reads the record tables and insert pools of contigs it does not own.
int msim_gather_fetch(msim_ctx *p, uint64_t device_addr, uint64_t bytes, void *dst) {
    Ctx *c = reinterpret_cast<Ctx *>(p);
    if (!c || (bytes && (!dst || !device_addr))) return MSIM_ERR_ARG;
    if (c->host_only) return fail(c, MSIM_ERR_HIP, "host-only context: this call needs the GPU");
    if (bytes) MSIM_HIP(c, hipMemcpy(dst, reinterpret_cast<const void *>((uintptr_t)device_addr), bytes, hipMemcpyDeviceToHost));
    return MSIM_OK;
}

// ---- test hooks: NOT part of the ABI (absent from include/msim.h) -------------------------------------------------
// RCCL on one GPU: the contig's mutated stream is sent to and received from this rank itself inside one group;
// *sum = checksum of what came back (tests/test_gpu_parity.py).
int msim_dbg_comm_loopback(msim_ctx *p, int contig, uint64_t *sum) {
    Ctx *c = reinterpret_cast<Ctx *>(p);
    if (!c || !sum) return MSIM_ERR_ARG;
    Comm *m = c->comm;
    if (!m) return fail(c, MSIM_ERR_ARG, "msim_dbg_comm_loopback before msim_comm_init");
    if (contig < 0 || (size_t)contig >= c->contigs.size()) return fail(c, MSIM_ERR_ARG, "no such contig");
    int rc = msim_sync(p);
    if (rc) return rc;
    Contig &g = c->contigs[(size_t)contig];
    if (!g.applied) return fail(c, MSIM_ERR_ARG, "contig has not been applied");
    uint8_t *buf = nullptr;
    MSIM_HIP(c, hipMalloc(&buf, g.out_len + PAD));
    MSIM_HIP(c, hipMemset(buf, 0, g.out_len + PAD));
    int nrc = g_rccl.GroupStart();
    if (!nrc) nrc = g_rccl.Send(g.d_out, g.out_len, NCCL_UINT8, m->rank, m->comm, c->emit_stream);
    if (!nrc) nrc = g_rccl.Recv(buf, g.out_len, NCCL_UINT8, m->rank, m->comm, c->emit_stream);
    const int erc = g_rccl.GroupEnd();
    if (!nrc) nrc = erc;
    if (nrc) { (void)hipFree(buf); return fail(c, MSIM_ERR_HIP, std::string("RCCL loopback: ") + g_rccl.GetErrorString(nrc)); }
    MSIM_HIP(c, wait_stream(c->emit_stream));
    rc = checksum_device(c, buf, g.out_len, sum);
    (void)hipFree(buf);
    return rc;
}

int msim_dbg_checksum_device(msim_ctx *p, uint64_t device_addr, uint64_t len, uint64_t *sum) {
    Ctx *c = reinterpret_cast<Ctx *>(p);
    if (!c || !sum || c->host_only) return MSIM_ERR_ARG;
    return checksum_device(c, reinterpret_cast<const uint8_t *>((uintptr_t)device_addr), len, sum);
}

}  // extern "C"
