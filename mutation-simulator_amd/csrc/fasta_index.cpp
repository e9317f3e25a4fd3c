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

// One record: [rec0, rec1) starts with its defline.  Lines are '\n'-separated; a trailing '\r' does not count as a base.
// pyfaidx's rule: every body line before the last non-empty one has the first line's length (and, for the fixed-stride
// ingest, its terminator); the last non-empty line may be shorter, never longer; trailing empty lines are ignored.
void index_record(const uint8_t *text, uint64_t rec0, uint64_t rec1, msim_fasta_record &r) {
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
    bool first = true, first_cr = false, seen_bad = false, seen_crdiff = false, bad = false, nonuni = false, any_nz = false;
    uint64_t lenc = 0, n_bases = 0, last_nz_len = 0, last_end = p;
    while (p < rec1) {
        const uint8_t *q = static_cast<const uint8_t *>(memchr(text + p, '\n', rec1 - p));
        const uint64_t e = q ? (uint64_t)(q - text) : rec1;
        const bool cr = e > p && text[e - 1] == '\r';
        const uint64_t len = e - p - (cr ? 1 : 0);
        if (first) {
            first = false;
            lenc = len;
            first_cr = cr;
            r.b0 = p;
            r.lenc = (uint32_t)std::min<uint64_t>(len, 0xffffffffu);
            r.lenb = (uint32_t)std::min<uint64_t>(e - p + 1, 0xffffffffu);
        }
        n_bases += len;
        if (len > 0) {                                   // every line before this one is "before the last non-empty line"
            bad |= seen_bad;
            nonuni |= seen_crdiff;
            any_nz = true;
            last_nz_len = len;
        }
        seen_bad |= len != lenc;
        seen_crdiff |= cr != first_cr;
        last_end = e;
        p = e + 1;
    }
    if (any_nz && last_nz_len > lenc) bad = true;
    r.b1 = last_end;
    r.n_bases = n_bases;
    if (bad) r.flags |= MSIM_FASTA_BAD_LINES;
    if (nonuni) r.flags |= MSIM_FASTA_NONUNIFORM;
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
    // ---- 2. one record at a time, records shared out dynamically (a genome has a few huge ones, an assembly many small)
    std::vector<std::thread> th;
    std::atomic<uint64_t> next{0};
    const uint64_t R = hdr.size();
    auto work = [&]() {
        for (;;) {
            const uint64_t k0 = next.fetch_add(64);
            if (k0 >= R) break;
            for (uint64_t k = k0; k < std::min(R, k0 + 64); k++) index_record(text, hdr[k], k + 1 < R ? hdr[k + 1] : n, records[k]);
        }
    };
    const int T2 = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)index_threads(), (R + 63) / 64));
    try {
        for (int t = 1; t < T2; t++) th.emplace_back(work);
    } catch (const std::system_error &) {                  // fewer helpers: the shared counter hands their records to the others
    }
    work();
    for (auto &x : th) x.join();
    return MSIM_OK;
}
