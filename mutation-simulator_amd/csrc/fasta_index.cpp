// FASTA index pass of the host loader (mutation_simulator_amd/fasta_io.py), natively and on several host threads.
//
// The reference reads sequences through pyfaidx (util.py:77-91: Fasta(..., sequence_always_upper=True)), whose index
// holds, per record: name, length, offset of the first base, bases per line (lenc) and bytes per line (lenb), and which
// refuses files whose lines inside a record differ in length.  The loader needs the same per-record facts plus where the
// record's text sits, so that the text can go to the device as it is (msim_add_contig_text) -- nothing here touches a base.
// At 3 GB of text the NumPy formulation of this pass (a dozen whole-file passes over per-line arrays) was two thirds of
// the CLI's wall time.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/msim.h"

namespace {

int index_threads() {
    static const int t = [] {
        if (const char *e = getenv("MSIM_BATCH_THREADS")) return std::max(1, atoi(e));
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(16, std::max<unsigned>(1, hw / 2));
    }();
    return t;
}

// size from which a record's lines are checked in parallel slices (MSIM_INDEX_BIG_BYTES: tests lower it)
uint64_t big_record_bytes() {
    static const uint64_t v = [] {
        const char *e = getenv("MSIM_INDEX_BIG_BYTES");
        return e ? (uint64_t)strtoull(e, nullptr, 10) : (uint64_t)(64u << 20);
    }();
    return v;
}

// One record: [rec0, rec1) starts with its defline.  Lines are '\n'-separated; a trailing '\r' does not count as a base.
// pyfaidx's rule: every body line before the last non-empty one has the first line's length (and, for the fixed-stride
// ingest, its terminator); the last non-empty line may be shorter, never longer; trailing empty lines are ignored.
struct LineWalk {
    bool first = true, first_cr = false, seen_bad = false, seen_crdiff = false, bad = false, nonuni = false, any_nz = false;
    uint64_t lenc = 0, n_bases = 0, last_nz_len = 0, last_end = 0;
};

// lines from p up to rec1 (or `max_lines` of them); returns where it stopped
uint64_t walk_lines(const uint8_t *text, uint64_t p, uint64_t rec1, LineWalk &w, msim_fasta_record &r, uint64_t max_lines) {
    while (p < rec1 && max_lines--) {
        const uint8_t *q = static_cast<const uint8_t *>(memchr(text + p, '\n', rec1 - p));
        const uint64_t e = q ? (uint64_t)(q - text) : rec1;
        const bool cr = e > p && text[e - 1] == '\r';
        const uint64_t len = e - p - (cr ? 1 : 0);
        if (w.first) {
            w.first = false;
            w.lenc = len;
            w.first_cr = cr;
            r.b0 = p;
            r.lenc = (uint32_t)std::min<uint64_t>(len, 0xffffffffu);
            r.lenb = (uint32_t)std::min<uint64_t>(e - p + 1, 0xffffffffu);
        }
        w.n_bases += len;
        if (len > 0) {                                   // every line before this one is "before the last non-empty line"
            w.bad |= w.seen_bad;
            w.nonuni |= w.seen_crdiff;
            w.any_nz = true;
            w.last_nz_len = len;
        }
        w.seen_bad |= len != w.lenc;
        w.seen_crdiff |= cr != w.first_cr;
        w.last_end = e;
        p = e + 1;
    }
    return p;
}

// Lines [0, count) of a body that starts at b0 with `lenb` bytes per line: true iff every one of them is exactly like the
// first line -- `lenb - 1 - cr` bases, the same terminator, no line feed inside.  (The common case for a chromosome: tens
// of millions of identical lines.  Checked in slices on several threads; anything else goes through the sequential walk.)
bool lines_uniform(const uint8_t *text, uint64_t b0, uint64_t lenb, bool cr, uint64_t i0, uint64_t i1) {
    for (uint64_t i = i0; i < i1; i++) {
        const uint8_t *ln = text + b0 + i * lenb;
        if (ln[lenb - 1] != '\n') return false;
        if (lenb >= 2 && (ln[lenb - 2] == '\r') != cr) return false;
        if (memchr(ln, '\n', lenb - 1)) return false;
    }
    return true;
}

void index_record(const uint8_t *text, uint64_t rec0, uint64_t rec1, msim_fasta_record &r, int threads) {
    r = msim_fasta_record{};
    const uint8_t *nl = static_cast<const uint8_t *>(memchr(text + rec0, '\n', rec1 - rec0));
    const uint64_t h_end = nl ? (uint64_t)(nl - text) : rec1;                 // exclusive, at the '\n'
    const bool h_cr = h_end > rec0 + 1 && text[h_end - 1] == '\r';
    r.h0 = rec0 + 1;
    r.h1 = h_end - (h_cr ? 1 : 0);
    uint64_t p = h_end + 1;
    r.b0 = r.b1 = p;                                                          // (offset of an empty record: after its defline)
    if (!nl || p >= rec1) return;                                             // no line after the defline
    r.flags |= MSIM_FASTA_HAS_BODY;
    LineWalk w;
    w.last_end = p;
    p = walk_lines(text, p, rec1, w, r, 1);                                   // the first line fixes lenc / lenb
    const uint64_t lenb = (uint64_t)(w.last_end + 1 - r.b0);
    if (threads > 1 && w.lenc > 0 && p < rec1 && lenb == r.lenb && rec1 - p > big_record_bytes() / 2) {
        // a large record: the lines that are certainly complete -- all but the last two -- in parallel slices
        const uint64_t avail = (rec1 - p) / lenb;
        const uint64_t count = avail > 2 ? avail - 2 : 0;
        std::vector<char> ok((size_t)threads, 1);
        std::vector<std::thread> th;
        auto slice = [&](int t) {
            const uint64_t i0 = 1 + count * (uint64_t)t / (uint64_t)threads, i1 = 1 + count * (uint64_t)(t + 1) / (uint64_t)threads;
            ok[(size_t)t] = lines_uniform(text, r.b0, lenb, w.first_cr, i0, i1) ? 1 : 0;
        };
        int started = 1;
        try {
            for (int t = 1; t < threads; t++, started++) th.emplace_back(slice, t);
        } catch (const std::system_error &) {
        }
        slice(0);
        for (int t = started; t < threads; t++) slice(t);
        for (auto &x : th) x.join();
        bool all = true;
        for (char c : ok) all = all && c;
        if (all && count) {                              // as if walk_lines had seen `count` more lines like the first
            w.n_bases += count * w.lenc;
            w.any_nz = true;
            w.last_nz_len = w.lenc;
            p += count * lenb;
            w.last_end = p - 1;
        }
    }
    walk_lines(text, p, rec1, w, r, UINT64_MAX);
    if (w.any_nz && w.last_nz_len > w.lenc) w.bad = true;
    r.b1 = w.last_end;
    r.n_bases = w.n_bases;
    if (w.bad) r.flags |= MSIM_FASTA_BAD_LINES;
    if (w.nonuni) r.flags |= MSIM_FASTA_NONUNIFORM;
}

}  // namespace

extern "C" int msim_fasta_index(const uint8_t *text, uint64_t n, msim_fasta_record *records, uint64_t cap, uint64_t *n_records) {
    if ((!text && n) || !n_records) return MSIM_ERR_ARG;
    *n_records = 0;
    if (n == 0) return MSIM_OK;
    // ---- 1. deflines: '>' at the start of a line.  Slices of the file on the host threads, joined in order.
    const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)index_threads(), n >> 22));
    std::vector<std::vector<uint64_t>> found((size_t)T);
    auto scan = [&](int t) {
        const uint64_t a = n * (uint64_t)t / (uint64_t)T, b = n * (uint64_t)(t + 1) / (uint64_t)T;
        uint64_t p = a;
        while (p < b) {
            const uint8_t *q = static_cast<const uint8_t *>(memchr(text + p, '>', b - p));
            if (!q) break;
            const uint64_t at = (uint64_t)(q - text);
            if (at == 0 || text[at - 1] == '\n') found[(size_t)t].push_back(at);
            p = at + 1;
        }
    };
    {
        std::vector<std::thread> th;
        int started = 1;
        try {
            for (int t = 1; t < T; t++, started++) th.emplace_back(scan, t);
        } catch (const std::system_error &) {              // no more threads to be had: the remaining slices run here
        }
        scan(0);
        for (int t = started; t < T; t++) scan(t);
        for (auto &x : th) x.join();
    }
    std::vector<uint64_t> hdr;
    for (auto &v : found) hdr.insert(hdr.end(), v.begin(), v.end());
    // sequence text before the first defline (any byte that is not a line feed) is an error (pyfaidx: FastaIndexingError)
    const uint64_t lead = hdr.empty() ? n : hdr[0];
    for (uint64_t i = 0; i < lead; i++) if (text[i] != '\n') return MSIM_ERR_VALUE;
    *n_records = hdr.size();
    if (hdr.empty() || !records) return MSIM_OK;
    if (cap < hdr.size()) return MSIM_ERR_ARG;
    // ---- 2. records.  A genome has a few huge ones: each of those is walked by all threads together (index_record slices
    // its lines); an assembly has many small ones: those are shared out dynamically, one thread per record.
    const uint64_t R = hdr.size();
    const int TH = index_threads();
    auto end_of = [&](uint64_t k) { return k + 1 < R ? hdr[k + 1] : n; };
    std::vector<uint64_t> small;
    for (uint64_t k = 0; k < R; k++) {
        if (end_of(k) - hdr[k] > big_record_bytes() && TH > 1) index_record(text, hdr[k], end_of(k), records[k], TH);
        else small.push_back(k);
    }
    std::vector<std::thread> th;
    std::atomic<uint64_t> next{0};
    const uint64_t S = small.size();
    auto work = [&]() {
        for (;;) {
            const uint64_t k0 = next.fetch_add(64);
            if (k0 >= S) break;
            for (uint64_t q = k0; q < std::min(S, k0 + 64); q++) index_record(text, hdr[small[q]], end_of(small[q]), records[small[q]], 1);
        }
    };
    const int T2 = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)TH, (S + 63) / 64));
    try {
        for (int t = 1; t < T2; t++) th.emplace_back(work);
    } catch (const std::system_error &) {                  // fewer helpers: the shared counter hands their records to the others
    }
    work();
    for (auto &x : th) x.join();
    return MSIM_OK;
}
