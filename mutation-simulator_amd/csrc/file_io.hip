// Device text -> output files, off the calling thread.
// gfx950 (MI355X) host side.                                              SURVEY.md section 8(f) rows 1-2 (egress).
//
// The CLI's outputs are files: the mutated Fasta (about as large as the genome) and the VCF.  What the bytes are is decided
// by the kernels of text_gpu.hip (fasta_writer.py:40-58, vcf_writer.py:118-126); this file only moves them, and the moving
// is what an end-to-end run spends its time on.  Measured on the MI355X box for 1.2 GiB into a fresh file (tmpfs / overlay):
//     mmap the span + copy into it (first-touch faults) + munmap     355 / 190 ms      <- what a D2H copy into a mapping pays
//     fallocate + mmap + populate + copy + munmap                    243 / 143 ms
//     the same with 4 threads on 4 slices of one file                386 / 515 ms      <- faults on one file do not scale
//     write() in 8 MiB pieces from a resident buffer                 144 /  85 ms      <- no page tables to build and tear down
// So: a channel per output file (0 the Fasta, 1 the VCF), each with its own thread, HIP stream and a small pinned ring.  An
// entry point renders the text into one of the channel's two device buffers on the context's stream and queues a job; the
// channel's thread copies it device -> pinned ring in 8 MiB pieces (the copy of piece k + 1 runs while piece k is written)
// and pwrite()s the pieces to the file.  The calling thread goes on with the next contig's ingest / PLAN / APPLY meanwhile;
// file_wait() joins.  Writes to ONE file serialise on its inode in the kernel anyway -- one thread per file is all there is.
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include "ctx.h"

namespace msim {

namespace {

constexpr size_t FILE_CHUNK = 8u << 20;
constexpr int FILE_SLOTS = 3;

struct FileJob {
    const uint8_t *d_src;                  // device text (a buffer of the channel) ...
    const uint8_t *h_src;                  // ... or host text the caller keeps untouched until the channel is idle
    int buf;
    uint64_t n;
    int fd;                                // a dup() of the caller's descriptor: closed when the job is done
    uint64_t offset;
};

struct FileChannel {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<FileJob> q;
    bool stop = false, running = false, started = false;
    int err = 0;
    std::string err_msg;
    int device = 0;
    // the thread's own
    hipStream_t st = nullptr;
    uint8_t *pin = nullptr;
    hipEvent_t ev[FILE_SLOTS] = {};
    // text buffers (device), filled by the calling thread on the context's stream
    uint8_t *d_buf[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0};
    bool in_flight[2] = {false, false};
    hipEvent_t ready[2] = {nullptr, nullptr};
};

}  // namespace

struct FileIo { FileChannel ch[2]; };

namespace {

void set_err(FileChannel &ch, int code, const std::string &msg) {
    std::lock_guard<std::mutex> lk(ch.mu);
    if (!ch.err) { ch.err = code; ch.err_msg = msg; }
}

bool pwrite_all(int fd, const uint8_t *p, size_t n, uint64_t off, std::string &why) {
    while (n) {
        const ssize_t w = pwrite(fd, p, n, (off_t)off);
        if (w < 0) {
            if (errno == EINTR) continue;
            why = std::string("pwrite: ") + strerror(errno);
            return false;
        }
        p += w; n -= (size_t)w; off += (uint64_t)w;
    }
    return true;
}

void run_job(FileChannel &ch, const FileJob &job) {
    static const bool prof = getenv("MSIM_IO_PROF") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    double wait_ms = 0, write_ms = 0;
    bool failed;
    { std::lock_guard<std::mutex> lk(ch.mu); failed = ch.err != 0; }
    hipError_t e = hipSuccess;
    if (!failed && job.h_src) {                             // host text (a batch of small contigs, framed on the host)
        const auto tb = std::chrono::steady_clock::now();
        std::string why;
        if (!pwrite_all(job.fd, job.h_src, job.n, job.offset, why)) set_err(ch, MSIM_ERR_IO, why);
        if (prof) write_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count();
        failed = true;                                      // (nothing left to do below)
    }
    if (!failed && !ch.st) {
        e = hipSetDevice(ch.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ch.st, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipHostMalloc((void **)&ch.pin, FILE_CHUNK * FILE_SLOTS, hipHostMallocDefault);
        for (int i = 0; i < FILE_SLOTS && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ch.ev[i], hipEventDisableTiming);
    }
    if (!failed && e == hipSuccess) e = hipStreamWaitEvent(ch.st, ch.ready[job.buf], 0);   // the text is complete
    if (!failed && e == hipSuccess) {
        const uint64_t pieces = (job.n + FILE_CHUNK - 1) / FILE_CHUNK;
        auto issue = [&](uint64_t k) {
            const uint64_t off = k * FILE_CHUNK, len = job.n - off < FILE_CHUNK ? job.n - off : FILE_CHUNK;
            const int slot = (int)(k % FILE_SLOTS);
            hipError_t r = hipMemcpyAsync(ch.pin + (size_t)slot * FILE_CHUNK, job.d_src + off, len, hipMemcpyDeviceToHost, ch.st);
            if (r == hipSuccess) r = hipEventRecord(ch.ev[slot], ch.st);
            return r;
        };
        for (uint64_t k = 0; k < pieces && k < (uint64_t)FILE_SLOTS && e == hipSuccess; k++) e = issue(k);
        for (uint64_t k = 0; k < pieces && e == hipSuccess; k++) {
            const int slot = (int)(k % FILE_SLOTS);
            const auto ta = std::chrono::steady_clock::now();
            e = wait_event(ch.ev[slot]);
            if (e != hipSuccess) break;
            const auto tb = std::chrono::steady_clock::now();
            const uint64_t off = k * FILE_CHUNK, len = job.n - off < FILE_CHUNK ? job.n - off : FILE_CHUNK;
            std::string why;
            const bool ok = pwrite_all(job.fd, ch.pin + (size_t)slot * FILE_CHUNK, len, job.offset + off, why);
            if (prof) {
                wait_ms += std::chrono::duration<double, std::milli>(tb - ta).count();
                write_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count();
            }
            if (!ok) {
                set_err(ch, MSIM_ERR_IO, why);
                (void)wait_stream(ch.st);
                break;
            }
            if (k + FILE_SLOTS < pieces) e = issue(k + FILE_SLOTS);
        }
    }
    if (e != hipSuccess) {
        set_err(ch, MSIM_ERR_HIP, std::string("output channel: ") + hipGetErrorString(e));
        if (ch.st) (void)wait_stream(ch.st);
    }
    (void)close(job.fd);
    if (prof)
        fprintf(stderr, "[msim io] t=%.4f %.1f MB at offset %llu: %.1f ms (waiting for the device copies %.1f, pwrite %.1f = %.2f GB/s)\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(), job.n / 1e6,
                (unsigned long long)job.offset, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                wait_ms, write_ms, write_ms > 0 ? job.n / 1e6 / write_ms : 0.0);
}

// A channel's thread runs on the CPUs of the NUMA node its GPU hangs on (sysfs: the PCI device's numa_node, the node's
// cpulist): the ring the DMA engine fills, the thread that reads it and the file pages it writes then share a socket -- on the
// two-socket hosts of the pool a CLI run pinned to the GPU's node took 0.31 s where the unpinned one took 0.40.
// MSIM_IO_CPUS overrides: a cpulist ("64-127,192-255"), or "none".
bool parse_cpulist(const char *s, cpu_set_t *set) {
    CPU_ZERO(set);
    bool any = false;
    while (*s) {
        char *end;
        long a = strtol(s, &end, 10);
        if (end == s) return false;
        long b = a;
        if (*end == '-') { s = end + 1; b = strtol(s, &end, 10); if (end == s) return false; }
        for (long q = a; q <= b && q < CPU_SETSIZE; q++) { CPU_SET((int)q, set); any = true; }
        s = end;
        while (*s == ',' || *s == ' ' || *s == '\n') s++;
    }
    return any;
}

bool read_line(const std::string &path, char *buf, size_t cap) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    const bool ok = fgets(buf, (int)cap, f) != nullptr;
    fclose(f);
    return ok;
}

// cpulist of the NUMA node the device hangs on ("" when sysfs does not tell or the host has one node)
void gpu_node_cpulist(int device, char *buf, size_t cap) {
    buf[0] = 0;
    char bus[64] = {0}, tmp[64];
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char *q = bus; *q; q++) *q = (char)tolower((unsigned char)*q);
    if (!read_line(std::string("/sys/bus/pci/devices/") + bus + "/numa_node", tmp, sizeof tmp)) return;
    const int node = atoi(tmp);
    if (node < 0) return;                                   // (a one-node host says -1)
    if (!read_line("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", buf, cap)) buf[0] = 0;
    for (char *q = buf; *q; q++) if (*q == '\n') *q = 0;
}

void pin_to_gpu_node(int device) {
    cpu_set_t set;
    char buf[4096];
    if (const char *e = getenv("MSIM_IO_CPUS")) {
        if (strcmp(e, "none") != 0 && parse_cpulist(e, &set)) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
        return;
    }
    gpu_node_cpulist(device, buf, sizeof buf);
    if (buf[0] && parse_cpulist(buf, &set)) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);   // (refused in a narrower cpuset: stays put)
}

void channel_main(FileChannel *chp) {
    FileChannel &ch = *chp;
    pin_to_gpu_node(ch.device);
    for (;;) {
        FileJob job;
        {
            std::unique_lock<std::mutex> lk(ch.mu);
            ch.cv_job.wait(lk, [&] { return ch.stop || !ch.q.empty(); });
            if (ch.q.empty()) break;                        // (stop: only once the queue is empty)
            job = ch.q.front();
            ch.q.pop_front();
            ch.running = true;
        }
        run_job(ch, job);
        {
            std::lock_guard<std::mutex> lk(ch.mu);
            ch.running = false;
            if (job.buf >= 0) ch.in_flight[job.buf] = false;
        }
        ch.cv_done.notify_all();
    }
    if (ch.st) {
        (void)wait_stream(ch.st);
        for (auto &e : ch.ev) if (e) (void)hipEventDestroy(e);
        if (ch.pin) (void)hipHostFree(ch.pin);
        (void)hipStreamDestroy(ch.st);
        ch.st = nullptr;
    }
}

FileIo *io_get(Ctx *c) {
    if (!c->file_io) {
        c->file_io = new FileIo;
        for (auto &ch : c->file_io->ch) ch.device = c->device;
    }
    return c->file_io;
}

}  // namespace

void device_host_cpus(int device, char *buf, size_t cap) { gpu_node_cpulist(device, buf, cap); }

// `fd` must be something pwrite() can address: a regular file.  MSIM_ERR_UNSUPPORTED otherwise (a pipe, a terminal): the
// caller fetches the text into a buffer and write()s it.
int file_check(Ctx *c, int fd) {
    struct stat sb;
    if (fstat(fd, &sb) != 0) return fail(c, MSIM_ERR_IO, std::string("fstat: ") + strerror(errno));
    if (!S_ISREG(sb.st_mode)) return fail(c, MSIM_ERR_UNSUPPORTED, "output is not a regular file: it cannot be written at an offset");
    return MSIM_OK;
}

// A device buffer of channel `ch` that no queued job reads (waits for the older of the two jobs when both are queued).
// *buf / *cap: the buffer's pointer and capacity (grow it with dev_reserve); *slot: for file_enqueue.
int file_text_buffer(Ctx *c, int ch_id, int *slot, uint8_t ***buf, size_t **cap) {
    FileChannel &ch = io_get(c)->ch[ch_id];
    std::unique_lock<std::mutex> lk(ch.mu);
    ch.cv_done.wait(lk, [&] { return !ch.in_flight[0] || !ch.in_flight[1]; });
    const int s = !ch.in_flight[0] ? 0 : 1;
    *slot = s;
    *buf = &ch.d_buf[s];
    *cap = &ch.cap[s];
    return MSIM_OK;
}

// The first n bytes of the channel's buffer `slot`, complete once everything now on the context's stream has run, go to
// bytes [offset, offset + n) of `fd`.  Returns at once; file_wait() tells when and whether they got there.
int file_enqueue(Ctx *c, int ch_id, int slot, uint64_t n, int fd, uint64_t offset) {
    if (!n) return MSIM_OK;
    FileChannel &ch = io_get(c)->ch[ch_id];
    if (!ch.ready[slot]) MSIM_HIP(c, hipEventCreateWithFlags(&ch.ready[slot], hipEventDisableTiming));
    MSIM_HIP(c, hipEventRecord(ch.ready[slot], c->stream));
    const int own = dup(fd);
    if (own < 0) return fail(c, MSIM_ERR_IO, std::string("dup: ") + strerror(errno));
    {
        std::lock_guard<std::mutex> lk(ch.mu);
        if (!ch.started) {
            try {
                ch.th = std::thread(channel_main, &ch);
            } catch (const std::system_error &e) {
                (void)close(own);
                return fail(c, MSIM_ERR_NOMEM, std::string("output channel thread: ") + e.what());
            }
            ch.started = true;
        }
        ch.in_flight[slot] = true;
        ch.q.push_back(FileJob{ch.d_buf[slot], nullptr, slot, n, own, offset});
    }
    ch.cv_job.notify_one();
    return MSIM_OK;
}

// Host text [src, src + n) -> bytes [offset, offset + n) of `fd` on the channel's thread.  The caller leaves the text alone
// until file_channel_idle(ch) / file_wait() has returned.
int file_enqueue_host(Ctx *c, int ch_id, const uint8_t *src, uint64_t n, int fd, uint64_t offset) {
    if (!n) return MSIM_OK;
    FileChannel &ch = io_get(c)->ch[ch_id];
    const int own = dup(fd);
    if (own < 0) return fail(c, MSIM_ERR_IO, std::string("dup: ") + strerror(errno));
    {
        std::lock_guard<std::mutex> lk(ch.mu);
        if (!ch.started) {
            try {
                ch.th = std::thread(channel_main, &ch);
            } catch (const std::system_error &e) {
                (void)close(own);
                return fail(c, MSIM_ERR_NOMEM, std::string("output channel thread: ") + e.what());
            }
            ch.started = true;
        }
        ch.q.push_back(FileJob{nullptr, src, -1, n, own, offset});
    }
    ch.cv_job.notify_one();
    return MSIM_OK;
}

// The channel has nothing queued or running (a failure stays recorded for file_wait).
void file_channel_idle(Ctx *c, int ch_id) {
    if (!c->file_io) return;
    FileChannel &ch = c->file_io->ch[ch_id];
    std::unique_lock<std::mutex> lk(ch.mu);
    ch.cv_done.wait(lk, [&] { return ch.q.empty() && !ch.running; });
}

// Everything queued is in its file (or failed: the first failure is reported, once).
int file_wait(Ctx *c) {
    if (!c->file_io) return MSIM_OK;
    int rc = MSIM_OK;
    for (auto &ch : c->file_io->ch) {
        std::unique_lock<std::mutex> lk(ch.mu);
        ch.cv_done.wait(lk, [&] { return ch.q.empty() && !ch.running; });
        if (ch.err && rc == MSIM_OK) {
            rc = ch.err;
            fail(c, ch.err, ch.err_msg);
        }
        ch.err = 0;
        ch.err_msg.clear();
    }
    return rc;
}

void file_io_destroy(Ctx *c) {
    FileIo *io = c->file_io;
    if (!io) return;
    for (auto &ch : io->ch) {
        {
            std::lock_guard<std::mutex> lk(ch.mu);
            ch.stop = true;
        }
        ch.cv_job.notify_all();
        if (ch.th.joinable()) ch.th.join();
        for (int s = 0; s < 2; s++) {
            if (ch.ready[s]) (void)hipEventDestroy(ch.ready[s]);
            if (ch.d_buf[s]) (void)hipFree(ch.d_buf[s]);
        }
    }
    delete io;
    c->file_io = nullptr;
}

}  // namespace msim
