// VCF record text for one contig (host helper of the C-ABI; stateless).
//
// Restates how the reference builds a VcfRecord per mutation type (mutator.py:334-399) and prints
// it (vcf_writer.py:44-52, 118-126).  REF/ALT strings are a pure function of (record, input bases,
// insert pool), so the device only ever emits the 16-byte binary records; text is rendered here,
// outside the timed region (for an SV mix the VCF text is as large as the genome).
#include <cstring>
#include <string>

#include "../../include/msim.h"

namespace {

struct Tables {
    uint8_t conv[256], comp[256], ti[256], tv[2][256];
    Tables() {
        for (int i = 0; i < 256; i++) { conv[i] = comp[i] = ti[i] = (uint8_t)i; tv[0][i] = tv[1][i] = 0; }
        const char *a = "KSYMWRBDHV-", *b = "GCCAAACAAAN";
        for (int i = 0; a[i]; i++) conv[(uint8_t)a[i]] = (uint8_t)b[i];
        a = "ACGTUMRWSYKVHDB"; b = "TGCAAKYWSRMBDHV";
        for (int i = 0; a[i]; i++) comp[(uint8_t)a[i]] = (uint8_t)b[i];
        a = "AGTC"; b = "GACT";
        for (int i = 0; a[i]; i++) ti[(uint8_t)a[i]] = (uint8_t)b[i];
        const char *keys = "AGTCN";
        const char *cols[5] = {"TC", "CT", "GA", "AG", "NN"};
        for (int i = 0; i < 5; i++) { tv[0][(uint8_t)keys[i]] = (uint8_t)cols[i][0]; tv[1][(uint8_t)keys[i]] = (uint8_t)cols[i][1]; }
    }
};
const Tables T;

// Sink that either counts or writes.
struct Sink {
    char *p;
    uint64_t n = 0;
    explicit Sink(char *dst) : p(dst) {}
    inline void put(char c) { if (p) p[n] = c; n++; }
    inline void put(const char *s, size_t len) { if (p) memcpy(p + n, s, len); n += len; }
    inline void lit(const char *s) { put(s, strlen(s)); }
    inline void num(uint64_t v) {
        char t[24]; int k = 0;
        do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (k) put(t[--k]);
    }
};

inline void line_head(Sink &s, const char *name, size_t name_len, uint64_t start) {
    s.put(name, name_len); s.put('\t'); s.num(start); s.lit("\t.\t");
}
inline void line_tail(Sink &s, const char *svtype, uint64_t end, uint64_t len) {
    s.lit("\t.\t.\t");
    if (svtype) { s.lit("SVTYPE="); s.lit(svtype); s.lit(";END="); s.num(end); s.lit(";SVLEN="); s.num(len); }
    else s.put('.');
    s.lit("\tGT\t1\n");
}

uint64_t render(const msim_record *recs, uint64_t n, const uint8_t *pool, const uint8_t *in, uint64_t L,
                const char *name, char *out) {
    Sink s(out);
    const size_t name_len = strlen(name);
    for (uint64_t i = 0; i < n; i++) {
        const msim_record &r = recs[i];
        const uint64_t pos = r.pos, stop = r.stop;
        switch (r.type) {
            case MSIM_SN: {                                          // mutator.py:334-341
                const uint8_t ref = T.conv[in[pos]];
                const uint8_t alt = r.aux == 0 ? T.ti[ref] : T.tv[r.aux - 1][ref];
                if (ref == alt) break;                               // vcf_writer.py:123
                line_head(s, name, name_len, pos + 1);
                s.put((char)ref); s.put('\t'); s.put((char)alt);
                line_tail(s, nullptr, 0, 0);
                break;
            }
            case MSIM_IN: {                                          // mutator.py:343-358
                const uint64_t len = stop + 1 - pos;
                const uint8_t *ins = pool + r.extra;
                if (pos > 0) {
                    const char ref = (char)T.conv[in[pos - 1]];
                    line_head(s, name, name_len, pos);
                    s.put(ref); s.put('\t'); s.put(ref); s.put((const char *)ins, len);
                    line_tail(s, "INS", pos, len);
                } else {
                    const char ref = (char)T.conv[in[0]];
                    line_head(s, name, name_len, 1);
                    s.put(ref); s.put('\t'); s.put((const char *)ins, len); s.put(ref);
                    line_tail(s, "INS", 1, len);
                }
                break;
            }
            case MSIM_TLI: {                                         // mutator.py:401-421
                // insert = conv(seq[start : stop+1]) (Python slice: empty when stop+1 <= start)
                const uint64_t src = r.extra;
                const uint64_t hi = stop + 1 < L ? stop + 1 : L;
                const uint64_t ilen = hi > src ? hi - src : 0;
                const bool rev = r.aux & 1, after = r.aux & 2;
                if (ilen == 0) break;                                // REF == ALT: suppressed
                const uint64_t start = after ? pos : pos + 1;
                const char ref = (char)T.conv[in[after ? pos - 1 : pos]];
                line_head(s, name, name_len, start);
                s.put(ref); s.put('\t');
                if (after) s.put(ref);
                if (rev) for (uint64_t q = 0; q < ilen; q++) s.put((char)T.comp[T.conv[in[hi - 1 - q]]]);
                else for (uint64_t q = 0; q < ilen; q++) s.put((char)T.conv[in[src + q]]);
                if (!after) s.put(ref);
                line_tail(s, "INS:ME", start, ilen);
                break;
            }
            case MSIM_TL:
            case MSIM_DE: {                                          // mutator.py:360-377
                uint64_t start = pos, end = stop + 1, lo = pos - 1;
                if (pos == 0) { start = 1; end = stop + 2; lo = 0; }
                const uint64_t hi = end < L ? end : L;               // slice clamps at len(sequence)
                line_head(s, name, name_len, start);
                for (uint64_t q = lo; q < hi; q++) s.put((char)T.conv[in[q]]);
                s.put('\t');
                s.put((char)T.conv[in[pos > 0 ? lo : hi - 1]]);      // REF[0] / REF[-1]
                line_tail(s, r.type == MSIM_DE ? "DEL" : "DEL:ME", end, stop - pos + 1);
                break;
            }
            case MSIM_IV: {                                          // mutator.py:379-387
                const uint64_t len = stop - pos + 1;
                bool same = true;                                    // REF == ALT (palindrome): suppressed
                for (uint64_t q = 0; q < len && same; q++)
                    same = T.conv[in[pos + q]] == T.comp[T.conv[in[stop - q]]];
                if (same) break;
                line_head(s, name, name_len, pos + 1);
                for (uint64_t q = 0; q < len; q++) s.put((char)T.conv[in[pos + q]]);
                s.put('\t');
                for (uint64_t q = 0; q < len; q++) s.put((char)T.comp[T.conv[in[stop - q]]]);
                line_tail(s, "INV", stop + 1, 0);
                break;
            }
            case MSIM_DU: {                                          // mutator.py:389-399
                const uint64_t len = stop - pos + 1;
                line_head(s, name, name_len, pos + 1);
                s.put((const char *)in + pos, len); s.put('\t');
                s.put((const char *)in + pos, len); s.put((const char *)in + pos, len);
                line_tail(s, "DUP", pos + len, len);
                break;
            }
            default: break;
        }
    }
    return s.n;
}

}  // namespace

// single pass, no size check: the caller guarantees room (msim_batch_run sizes its buffer by an upper bound)
namespace msim {
uint64_t render_vcf_unchecked(const msim_record *recs, uint64_t n_records, const uint8_t *insert_pool, const uint8_t *bases,
                              uint64_t len, const char *seq_name, char *out) {
    return render(recs, n_records, insert_pool, bases, len, seq_name, out);
}
}  // namespace msim

extern "C" int msim_render_vcf(const msim_record *recs, uint64_t n_records, const uint8_t *insert_pool,
                               const uint8_t *bases, uint64_t len, const char *seq_name, char *out,
                               uint64_t cap, uint64_t *needed) {
    if ((n_records && !recs) || !seq_name || !needed || (len && !bases)) return MSIM_ERR_ARG;
    const uint64_t want = render(recs, n_records, insert_pool, bases, len, seq_name, nullptr);
    *needed = want;
    if (!out) return MSIM_OK;
    if (cap < want) return MSIM_ERR_ARG;
    render(recs, n_records, insert_pool, bases, len, seq_name, out);
    return MSIM_OK;
}
