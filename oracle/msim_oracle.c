/*
 * msim_oracle.c -- CPU ORACLE for the mutation-injection hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain, sequential C restatement of what the reference does on this path, written from the
 * reference's behaviour (never from its text) with every function citing the reference file:line
 * it follows (paths relative to /root/reference/mutation_simulator/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (mutation-simulator_amd/) never links, imports or calls it.
 *
 * Pinning: the reference ships no tests and no golden vectors, and the
 * arithmetic it relies on lives in two third-party pieces that are not under /root/reference:
 *   - CPython's `random` module (Lib/random.py + Modules/_randommodule.c, Python >= 3.10,
 *     pyproject.toml:29): MT19937, init_by_array seeding, getrandbits, _randbelow, sample,
 *     shuffle, random();
 *   - NumPy's legacy RandomState (numpy, unpinned in pyproject.toml:30-34; the legacy stream is
 *     frozen by NumPy policy): init_genrand seeding, random_sample, choice(p), choice(a).
 * This file restates their published algorithms and is pinned against goldens captured by
 * running the real reference in the build container (tests/golden/make_goldens.py, Python
 * 3.10.12 / NumPy 2.2.6, pyfaidx replaced by a sequence-access stand-in).
 *
 * Floating point: the few doubles on the path (rate sums, chances, titv) are computed by the
 * caller in Python with the reference's own expressions and passed in, or computed here with the
 * same IEEE-754 operations in the same order.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_VALUE 1   /* ValueError("Sample larger than population or is negative") */
#define ORC_ERR_KEY 2     /* KeyError(base) from the transversion table                  */
#define ORC_ERR_ARG 3
#define ORC_ERR_NOMEM 4

/* mut_types.py:6-12 */
enum { T_SN = 1, T_IN = 2, T_DE = 3, T_DU = 4, T_IV = 5, T_TL = 6, T_TLI = 7 };

/* ------------------------------------------------------------------ MT19937 (both libraries) */
typedef struct { uint32_t mt[624]; int idx; } mt_t;

/* init_genrand: _randommodule.c init_genrand / numpy mt19937_seed */
static void mt_init_genrand(mt_t *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}

/* init_by_array: _randommodule.c init_by_array (what random.seed(int) uses) */
static void mt_init_by_array(mt_t *s, const uint32_t *key, int klen) {
    mt_init_genrand(s, 19650218u);
    int i = 1, j = 0;
    int k = 624 > klen ? 624 : klen;
    for (; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
        if (j >= klen) j = 0;
    }
    for (k = 623; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
    }
    s->mt[0] = 0x80000000u;
    s->idx = 624;
}

static uint32_t mt_next(mt_t *s) {
    if (s->idx >= 624) {
        uint32_t *mt = s->mt;
        for (int k = 0; k < 624; k++) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* ------------------------------------------------------------------ oracle object */
/* One entry of the reference's `muts` dict (mutator.py:26-50): key `pos` -> Mutation(type, start,
 * stop, trans_reverse, trans_insert_pos).  start == pos except for a linked TLI. */
typedef struct { int64_t pos; int32_t type; int32_t rev; int64_t start; int64_t stop; int64_t ins_pos; } rec_t;

typedef struct {
    char *p; size_t n, cap;
} buf_t;

typedef struct {
    mt_t py, np;
    uint64_t py_words, np_words;   /* words drawn since creation / last reset (diagnostics) */
    int64_t block[8];              /* SimulationSettings.mut_block, indexed by type id */
    double titv;
    /* FastaWriter state (fasta_writer.py:24-25) */
    int64_t fw_written, fw_bpl;
    buf_t fasta, vcf;
    char key_error_base;
} orc_t;

static int buf_put(buf_t *b, const void *src, size_t n) {
    if (n == 0) return ORC_OK;               /* memcpy(NULL, ..., 0) is undefined */
    if (b->n + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 1 << 16;
        while (nc < b->n + n) nc *= 2;
        char *q = (char *)realloc(b->p, nc);
        if (!q) return ORC_ERR_NOMEM;
        b->p = q; b->cap = nc;
    }
    memcpy(b->p + b->n, src, n);
    b->n += n;
    return ORC_OK;
}
static int buf_putc(buf_t *b, char c) { return buf_put(b, &c, 1); }
static int buf_puts(buf_t *b, const char *s) { return buf_put(b, s, strlen(s)); }
static int buf_puti(buf_t *b, int64_t v) { char t[32]; snprintf(t, sizeof t, "%lld", (long long)v); return buf_puts(b, t); }

orc_t *orc_new(void) {
    orc_t *o = (orc_t *)calloc(1, sizeof *o);
    if (!o) return NULL;
    mt_init_genrand(&o->py, 5489u);
    mt_init_genrand(&o->np, 5489u);
    for (int i = 0; i < 8; i++) o->block[i] = 1;   /* defaults.py:27,35-42 */
    o->titv = 1.0;                                 /* defaults.py:25 */
    o->fw_bpl = 60;                                /* fasta_writer.py:25 */
    return o;
}
void orc_free(orc_t *o) { if (o) { free(o->fasta.p); free(o->vcf.p); free(o); } }

/* random.seed(int): key = abs(seed) as little-endian 32-bit limbs (>=1 limb); caller splits. */
void orc_seed_py(orc_t *o, const uint32_t *key, int n) { mt_init_by_array(&o->py, key, n); o->py_words = 0; }
/* numpy.random.seed(int) */
void orc_seed_np(orc_t *o, uint32_t seed) { mt_init_genrand(&o->np, seed); o->np_words = 0; }
void orc_set_state(orc_t *o, int stream, const uint32_t *mt, int idx) {
    mt_t *s = stream ? &o->np : &o->py; memcpy(s->mt, mt, sizeof s->mt); s->idx = idx;
}
void orc_get_state(orc_t *o, int stream, uint32_t *mt, int *idx) {
    mt_t *s = stream ? &o->np : &o->py; memcpy(mt, s->mt, sizeof s->mt); *idx = s->idx;
}
uint64_t orc_words(orc_t *o, int stream) { return stream ? o->np_words : o->py_words; }
void orc_set_block(orc_t *o, const int64_t *block8) { memcpy(o->block, block8, sizeof o->block); }
void orc_set_titv(orc_t *o, double titv) { o->titv = titv; }

static uint32_t py32(orc_t *o) { o->py_words++; return mt_next(&o->py); }
static uint32_t np32(orc_t *o) { o->np_words++; return mt_next(&o->np); }
uint32_t orc_py_next32(orc_t *o) { return py32(o); }
uint32_t orc_np_next32(orc_t *o) { return np32(o); }
/* n words of a stream consumed and dropped: the sequential generation a jump-ahead table is checked against
   (tests/test_jump_table.py); the same genrand_int32 steps as every draw of the reference (_randommodule.c) */
void orc_skip_words(orc_t *o, int stream, uint64_t n) {
    if (stream) for (uint64_t i = 0; i < n; i++) (void)np32(o);
    else for (uint64_t i = 0; i < n; i++) (void)py32(o);
}

/* ------------------------------------------------------------------ CPython random layer */
/* _randommodule.c getrandbits: k<=32 -> one word >> (32-k); else little-endian limbs, the top
 * limb shifted.  Supported here up to 64 bits. */
static uint64_t py_getrandbits(orc_t *o, int k) {
    if (k <= 32) return py32(o) >> (32 - k);
    uint64_t lo = py32(o);
    int rem = k - 32;
    uint64_t hi = py32(o);
    if (rem < 32) hi >>= (32 - rem);
    return lo | (hi << 32);
}
static int bit_length(uint64_t n) { int b = 0; while (n) { b++; n >>= 1; } return b; }

/* Lib/random.py _randbelow_with_getrandbits */
uint64_t orc_randbelow(orc_t *o, uint64_t n) {
    if (!n) return 0;
    int k = bit_length(n);
    uint64_t r = py_getrandbits(o, k);
    while (r >= n) r = py_getrandbits(o, k);
    return r;
}
/* Lib/random.py randint(a, b) = a + _randbelow(b - a + 1) */
int64_t orc_randint(orc_t *o, int64_t a, int64_t b) { return a + (int64_t)orc_randbelow(o, (uint64_t)(b - a + 1)); }
/* _randommodule.c random(): genrand_res53; uniform(0,1) = 0 + (1-0)*random() */
double orc_random(orc_t *o) {
    uint32_t a = py32(o) >> 5, b = py32(o) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

/* open-addressing set of int64 for sample()'s `selected` */
typedef struct { int64_t *slot; uint64_t mask; } iset_t;
static int iset_init(iset_t *s, uint64_t k) {
    uint64_t cap = 16; while (cap < 2 * k + 2) cap <<= 1;
    s->slot = (int64_t *)malloc(cap * sizeof(int64_t));
    if (!s->slot) return ORC_ERR_NOMEM;
    memset(s->slot, 0xff, cap * sizeof(int64_t));
    s->mask = cap - 1; return ORC_OK;
}
static int iset_add(iset_t *s, int64_t v) { /* returns 1 if newly added, 0 if present */
    uint64_t h = ((uint64_t)v * 0x9E3779B97F4A7C15ull) >> 20;
    for (;; h++) {
        int64_t *p = &s->slot[h & s->mask];
        if (*p == v) return 0;
        if (*p == -1) { *p = v; return 1; }
    }
}

/* Lib/random.py sample(range(n), k): selection-order result in out[0..k).  setsize as in CPython:
 * 21 + 4**ceil(log(3k, 4)) for k > 5 (math.log(x, 4) = log(x)/log(4)). */
int64_t orc_setsize(int64_t k) {
    int64_t setsize = 21;
    if (k > 5) setsize += (int64_t)llround(pow(4.0, ceil(log((double)k * 3.0) / log(4.0))));
    return setsize;
}
int orc_sample(orc_t *o, int64_t n, int64_t k, int64_t *out) {
    if (n < 0) n = 0;                       /* len(range(a, b)) with b < a */
    if (k < 0 || k > n) return ORC_ERR_VALUE;
    if (n <= orc_setsize(k)) {              /* pool path */
        int64_t *pool = (int64_t *)malloc((size_t)(n ? n : 1) * sizeof(int64_t));
        if (!pool) return ORC_ERR_NOMEM;
        for (int64_t i = 0; i < n; i++) pool[i] = i;
        for (int64_t i = 0; i < k; i++) {
            int64_t j = (int64_t)orc_randbelow(o, (uint64_t)(n - i));
            out[i] = pool[j];
            pool[j] = pool[n - i - 1];
        }
        free(pool);
    } else {                                /* set path */
        iset_t s;
        if (iset_init(&s, (uint64_t)k)) return ORC_ERR_NOMEM;
        for (int64_t i = 0; i < k; i++) {
            int64_t j = (int64_t)orc_randbelow(o, (uint64_t)n);
            while (!iset_add(&s, j)) j = (int64_t)orc_randbelow(o, (uint64_t)n);
            out[i] = j;
        }
        free(s.slot);
    }
    return ORC_OK;
}
/* Lib/random.py shuffle(x) */
void orc_shuffle(orc_t *o, int64_t *x, int64_t n) {
    for (int64_t i = n - 1; i >= 1; i--) {
        int64_t j = (int64_t)orc_randbelow(o, (uint64_t)(i + 1));
        int64_t t = x[i]; x[i] = x[j]; x[j] = t;
    }
}

static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b; return (x > y) - (x < y);
}
/* util.py:94-109 sample_with_minimum_distance: sorted(sample(range(start, stop-(k-1)d), k))[i] + d*i */
int orc_sample_min_dist(orc_t *o, int64_t start, int64_t stop, int64_t k, int64_t d, int64_t *out) {
    int64_t n = (stop - (k - 1) * d) - start;
    int rc = orc_sample(o, n, k, out);
    if (rc) return rc;
    if (k) qsort(out, (size_t)k, sizeof(int64_t), cmp_i64);
    for (int64_t i = 0; i < k; i++) out[i] = start + out[i] + d * i;
    return ORC_OK;
}

/* ------------------------------------------------------------------ NumPy legacy layer */
/* legacy random_sample: (a>>5, b>>6) -> (a*2^26+b)/2^53 */
double orc_np_double(orc_t *o) {
    int32_t a = (int32_t)(np32(o) >> 5), b = (int32_t)(np32(o) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}
/* RandomState.choice(a, p=p, size): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right') */
void orc_choice_p(orc_t *o, const double *p, int n, int64_t size, int32_t *out_idx) {
    double cdf[16];
    double acc = 0.0;
    for (int j = 0; j < n; j++) { acc = (j == 0) ? p[0] : acc + p[j]; cdf[j] = acc; }
    double last = cdf[n - 1];
    for (int j = 0; j < n; j++) cdf[j] /= last;
    for (int64_t i = 0; i < size; i++) {
        double u = orc_np_double(o);
        int idx = 0;
        while (idx < n && cdf[idx] <= u) idx++;
        out_idx[i] = idx;
    }
}
/* RandomState.choice(["A","T","G","C"], L) -> randint(0,4,L): masked rejection, mask 3, one word each */
void orc_choice_atgc(orc_t *o, int64_t len, char *out) {
    static const char ATGC[4] = { 'A', 'T', 'G', 'C' };
    for (int64_t i = 0; i < len; i++) out[i] = ATGC[np32(o) & 3u];
}

/* ------------------------------------------------------------------ settings handed in by tests */
typedef struct {
    int64_t start, stop;      /* RangeDefinition.start/stop, 0-based inclusive (rmt.py:166-189) */
    double rate_sum;          /* sum(mut_rates.values()) computed by the caller (mutator.py:160) */
    int32_t n_types;          /* len(mut_chances) in dict order (mutator.py:171-173) */
    int32_t types[8];
    double chances[8];
    int64_t min_len[8];       /* mut_lengs["min"/"max"], indexed by type id */
    int64_t max_len[8];
} orc_range;

/* mutator.py:228-265 __get_stop_position; returns 0 = keep, 1 = dropped (IV does not fit) */
static int get_stop_position(orc_t *o, rec_t *m, const orc_range *r, int64_t chrom_len) {
    int t = m->type;
    if (t == T_SN) {
        m->stop = m->pos;
    } else if (t == T_IV) {
        if (m->pos + r->max_len[T_IV] >= chrom_len - 1) return 1;
        m->stop = orc_randint(o, m->pos + r->min_len[T_IV] - 1, m->pos + r->max_len[T_IV] - 1);
    } else if (t == T_IN) {
        m->stop = orc_randint(o, m->pos + r->min_len[t] - 1, m->pos + r->max_len[t] - 1);
    } else if (t == T_DU || t == T_TL || t == T_DE) {
        m->stop = orc_randint(o, m->pos + r->min_len[t] - 1, m->pos + r->max_len[t] - 1);
        if (m->stop > chrom_len - 1) m->stop = chrom_len - 1;
    }   /* T_TLI: no branch matches, stop stays 0 (mutator.py:238-265) */
    return 0;
}

/* mutator.py:144-214 __get_mutations.  out must hold k records; returns kept count in *n_out.
 * tls/tlis receive the positions appended at mutator.py:210-213 (may be NULL when unused). */
int orc_get_mutations(orc_t *o, const orc_range *r, int64_t chrom_len, rec_t **out, int64_t *n_out,
                      int64_t **tls, int64_t *n_tls, int64_t **tlis, int64_t *n_tlis) {
    *out = NULL; *n_out = 0;
    if (tls) { *tls = NULL; *n_tls = 0; }
    if (tlis) { *tlis = NULL; *n_tlis = 0; }
    int64_t d = o->block[1];
    for (int t = 2; t <= 7; t++) if (o->block[t] < d) d = o->block[t];   /* min(mut_block.values()) */
    /* mutator.py:225  int(((stop - start) + 1) * mut_rate) */
    int64_t k = (int64_t)((double)((r->stop - r->start) + 1) * r->rate_sum);
    int64_t *pos = (int64_t *)malloc((size_t)(k > 0 ? k : 1) * sizeof(int64_t));
    if (!pos) return ORC_ERR_NOMEM;
    int rc = orc_sample_min_dist(o, r->start, r->stop, k, d, pos);
    if (rc) { free(pos); return rc; }
    if (k == 0) { free(pos); return ORC_OK; }              /* mutator.py:163-164 */
    int32_t *ti = (int32_t *)malloc((size_t)k * sizeof(int32_t));
    rec_t *recs = (rec_t *)malloc((size_t)k * sizeof(rec_t));
    int64_t *a = (int64_t *)malloc((size_t)k * sizeof(int64_t));
    int64_t *b = (int64_t *)malloc((size_t)k * sizeof(int64_t));
    if (!ti || !recs || !a || !b) { free(pos); free(ti); free(recs); free(a); free(b); return ORC_ERR_NOMEM; }
    orc_choice_p(o, r->chances, r->n_types, k, ti);        /* mutator.py:170-174 */
    int64_t kept = 0, na = 0, nb = 0;
    int64_t blk_lo = 0, blk_hi = 0;                        /* last_mut_range = range(0) */
    for (int64_t i = 0; i < k; i++) {                      /* mutator.py:185-213 */
        int64_t p = pos[i];
        if (p >= blk_lo && p < blk_hi) continue;
        rec_t m = { p, r->types[ti[i]], 0, p, 0, 0 };
        if (get_stop_position(o, &m, r, chrom_len)) continue;
        recs[kept++] = m;
        if (m.type == T_SN || m.type == T_IN) {
            blk_lo = m.pos; blk_hi = m.pos + 1 + o->block[m.type];
        } else {
            blk_lo = m.pos; blk_hi = m.stop + 1 + o->block[m.type];
            if (m.type == T_TL) a[na++] = p;
            if (m.type == T_TLI) b[nb++] = p;
        }
    }
    free(pos); free(ti);
    *out = recs; *n_out = kept;
    if (tls) { *tls = a; *n_tls = na; } else free(a);
    if (tlis) { *tlis = b; *n_tlis = nb; } else free(b);
    return ORC_OK;
}

/* ------------------------------------------------------------------ tables (mutator.py:75-77) */
static unsigned char TAB_NONAMB[256], TAB_COMP[256], TAB_TI[256];
static int tables_ready = 0;
static void init_tables(void) {
    if (tables_ready) return;
    for (int i = 0; i < 256; i++) TAB_NONAMB[i] = TAB_COMP[i] = TAB_TI[i] = (unsigned char)i;
    const char *a = "KSYMWRBDHV-", *b = "GCCAAACAAAN";
    for (int i = 0; a[i]; i++) TAB_NONAMB[(unsigned char)a[i]] = (unsigned char)b[i];
    a = "ACGTUMRWSYKVHDB"; b = "TGCAAKYWSRMBDHV";
    for (int i = 0; a[i]; i++) TAB_COMP[(unsigned char)a[i]] = (unsigned char)b[i];
    a = "AGTC"; b = "GACT";
    for (int i = 0; a[i]; i++) TAB_TI[(unsigned char)a[i]] = (unsigned char)b[i];
    tables_ready = 1;
}

/* ------------------------------------------------------------------ writers */
/* fasta_writer.py:49-58 write */
static int fw_write(orc_t *o, char base) {
    int rc = buf_putc(&o->fasta, base);
    o->fw_written++;
    if (o->fw_written % o->fw_bpl == 0) { rc |= buf_putc(&o->fasta, '\n'); o->fw_written = 0; }
    return rc;
}
/* fasta_writer.py:40-47 write_header */
static int fw_header(orc_t *o, const char *header) {
    int rc = 0;
    if (o->fw_written != 0) rc |= buf_putc(&o->fasta, '\n');
    rc |= buf_putc(&o->fasta, '>'); rc |= buf_puts(&o->fasta, header); rc |= buf_putc(&o->fasta, '\n');
    o->fw_written = 0;
    return rc;
}
/* vcf_writer.py:118-126 write + :44-52 info */
static int vcf_write(orc_t *o, const char *name, const char *svtype, int64_t start, int64_t end,
                     int64_t len, const char *ref, size_t nref, const char *alt, size_t nalt) {
    if (nref == nalt && memcmp(ref, alt, nref) == 0) return ORC_OK;
    buf_t *v = &o->vcf; int rc = 0;
    rc |= buf_puts(v, name); rc |= buf_putc(v, '\t'); rc |= buf_puti(v, start);
    rc |= buf_puts(v, "\t.\t"); rc |= buf_put(v, ref, nref); rc |= buf_putc(v, '\t');
    rc |= buf_put(v, alt, nalt); rc |= buf_puts(v, "\t.\t.\t");
    if (strcmp(svtype, "sn") != 0) {
        rc |= buf_puts(v, "SVTYPE="); rc |= buf_puts(v, svtype); rc |= buf_puts(v, ";END=");
        rc |= buf_puti(v, end); rc |= buf_puts(v, ";SVLEN="); rc |= buf_puti(v, len);
    } else rc |= buf_putc(v, '.');
    rc |= buf_puts(v, "\tGT\t1\n");
    return rc;
}

/* vcf_writer.py:74-116 write_header; `date` is passed in (wall clock in the reference) */
int orc_vcf_header(orc_t *o, const char *input_fasta_name, int n_contigs, const char **names,
                   const int64_t *lengths, const char *assembly, const char *species,
                   const char *sample, const char *date) {
    buf_t *v = &o->vcf; int rc = 0;
    rc |= buf_puts(v, "##fileformat=VCFv4.3\n##filedate="); rc |= buf_puts(v, date);
    rc |= buf_puts(v, "\n##source=Mutation-Simulator\n##reference="); rc |= buf_puts(v, input_fasta_name);
    rc |= buf_putc(v, '\n');
    for (int i = 0; i < n_contigs; i++) {
        rc |= buf_puts(v, "##contig=<ID="); rc |= buf_puts(v, names[i]); rc |= buf_puts(v, ",length=");
        rc |= buf_puti(v, lengths[i]); rc |= buf_puts(v, ",assembly="); rc |= buf_puts(v, assembly);
        rc |= buf_puts(v, ",species=\""); rc |= buf_puts(v, species); rc |= buf_puts(v, "\">\n");
    }
    rc |= buf_puts(v,
        "##INFO=<ID=SVTYPE,Number=1,Type=String,Description=\"Type of structural variant\">\n"
        "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End position of the variant described in this record\">\n"
        "##INFO=<ID=SVLEN,Number=.,Type=Integer,Description=\"Difference in length between REF and ALT alleles\">\n"
        "##ALT=<ID=INS,Description=\"Insert\">\n"
        "##ALT=<ID=DEL,Description=\"Deletion\">\n"
        "##ALT=<ID=DUP,Description=\"Duplication\">\n"
        "##ALT=<ID=INV,Description=\"Inversion\">\n"
        "##ALT=<ID=DEL:ME,Description=\"Deletion of mobile element\">\n"
        "##ALT=<ID=INS:ME,Description=\"Insertion of mobile element\">\n"
        "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t");
    rc |= buf_puts(v, sample); rc |= buf_putc(v, '\n');
    return rc;
}

/* ------------------------------------------------------------------ SNP helpers */
/* mutator.py:428-463 __get_snp / __get_ti_Base / __get_tv_Base */
static int get_snp(orc_t *o, unsigned char base, unsigned char *alt) {
    double p_ti = o->titv * (1 / (o->titv + 1));
    double p = orc_random(o);                 /* uniform(0, 1) */
    if (p <= p_ti) { *alt = TAB_TI[base]; return ORC_OK; }
    const char *pair;
    switch (base) {                           /* the dict lookup precedes the randint */
        case 'A': pair = "TC"; break;
        case 'G': pair = "CT"; break;
        case 'T': pair = "GA"; break;
        case 'C': pair = "AG"; break;
        case 'N': pair = "NN"; break;
        default: o->key_error_base = (char)base; return ORC_ERR_KEY;
    }
    *alt = (unsigned char)pair[orc_randint(o, 0, 1)];
    return ORC_OK;
}

/* scratch strings */
static int sb_conv(buf_t *b, const unsigned char *seq, int64_t lo, int64_t hi /*exclusive*/, int64_t L) {
    if (lo < 0) lo = 0;
    if (hi > L) hi = L;
    b->n = 0;
    for (int64_t i = lo; i < hi; i++) { if (buf_putc(b, (char)TAB_NONAMB[seq[i]])) return ORC_ERR_NOMEM; }
    return ORC_OK;
}

/* mutator.py:318-426 __mutate_sequence.  recs sorted by pos, unique pos (dict keys). */
int orc_mutate_sequence(orc_t *o, const unsigned char *seq, int64_t L, const char *name,
                        const rec_t *recs, int64_t n_recs) {
    init_tables();
    buf_t ref = {0}, alt = {0}, ins = {0};
    int rc = ORC_OK;
    int64_t ri = 0, pos = 0;
    while (pos < L && rc == ORC_OK) {
        while (ri < n_recs && recs[ri].pos < pos) ri++;          /* `pos in muts` */
        if (ri < n_recs && recs[ri].pos == pos) {
            const rec_t *m = &recs[ri];
            if (m->type == T_SN) {                               /* :334-341 */
                unsigned char r = TAB_NONAMB[seq[pos]], a;
                rc = get_snp(o, r, &a); if (rc) break;
                rc |= fw_write(o, (char)a);
                rc |= vcf_write(o, name, "sn", pos + 1, 0, 0, (char *)&r, 1, (char *)&a, 1);
            } else if (m->type == T_IN) {                        /* :343-358 */
                int64_t len = m->stop + 1 - pos;
                ins.n = 0;
                for (int64_t i = 0; i < len; i++) { char c; orc_choice_atgc(o, 1, &c); rc |= buf_putc(&ins, c); }
                int64_t start = pos, end = pos;
                ref.n = 0; alt.n = 0;
                if (pos > 0) {
                    rc |= buf_putc(&ref, (char)TAB_NONAMB[seq[pos - 1]]);
                    rc |= buf_put(&alt, ref.p, ref.n); rc |= buf_put(&alt, ins.p, ins.n);
                } else {
                    rc |= buf_putc(&ref, (char)TAB_NONAMB[seq[0]]);
                    rc |= buf_put(&alt, ins.p, ins.n); rc |= buf_put(&alt, ref.p, ref.n);
                    start += 1; end += 1;
                }
                for (size_t i = 0; i < ins.n; i++) rc |= fw_write(o, ins.p[i]);
                rc |= fw_write(o, (char)seq[pos]);
                rc |= vcf_write(o, name, "INS", start, end, (int64_t)ins.n, ref.p, ref.n, alt.p, alt.n);
            } else if (m->type == T_DE || m->type == T_TL) {     /* :360-377 */
                const char *sv = m->type == T_DE ? "DEL" : "DEL:ME";
                int64_t start = pos, end = m->stop + 1;
                alt.n = 0;
                if (pos > 0) {
                    rc |= sb_conv(&ref, seq, pos - 1, end, L);
                    rc |= buf_putc(&alt, ref.p[0]);
                } else {
                    start += 1; end += 1;
                    rc |= sb_conv(&ref, seq, 0, end, L);
                    rc |= buf_putc(&alt, ref.p[ref.n - 1]);
                }
                int64_t len = m->stop - pos + 1;
                pos = m->stop;
                rc |= vcf_write(o, name, sv, start, end, len, ref.p, ref.n, alt.p, alt.n);
            } else if (m->type == T_IV) {                        /* :379-387 */
                int64_t end = m->stop + 1;
                rc |= sb_conv(&ref, seq, pos, end, L);
                alt.n = 0;
                for (size_t i = ref.n; i > 0; i--) rc |= buf_putc(&alt, (char)TAB_COMP[(unsigned char)ref.p[i - 1]]);
                for (size_t i = 0; i < alt.n; i++) rc |= fw_write(o, alt.p[i]);
                int64_t start = pos + 1;
                pos = m->stop;
                rc |= vcf_write(o, name, "INV", start, end, 0, ref.p, ref.n, alt.p, alt.n);
            } else if (m->type == T_DU) {                        /* :389-399 */
                int64_t hi = m->stop + 1 > L ? L : m->stop + 1;
                ref.n = 0; alt.n = 0;
                rc |= buf_put(&ref, seq + pos, (size_t)(hi - pos));
                rc |= buf_put(&alt, ref.p, ref.n); rc |= buf_put(&alt, ref.p, ref.n);
                for (size_t i = 0; i < alt.n; i++) rc |= fw_write(o, alt.p[i]);
                int64_t start = pos + 1, len = (int64_t)ref.n, end = pos + len;
                pos = m->stop;
                rc |= vcf_write(o, name, "DUP", start, end, len, ref.p, ref.n, alt.p, alt.n);
            } else if (m->type == T_TLI) {                       /* :401-421 */
                rc |= sb_conv(&ins, seq, m->start, m->stop + 1, L);
                if (m->rev) {
                    for (size_t i = 0; i < ins.n / 2; i++) { char t = ins.p[i]; ins.p[i] = ins.p[ins.n - 1 - i]; ins.p[ins.n - 1 - i] = t; }
                    for (size_t i = 0; i < ins.n; i++) ins.p[i] = (char)TAB_COMP[(unsigned char)ins.p[i]];
                }
                int64_t start = pos;
                alt.n = 0;
                if (m->ins_pos > 0) {
                    rc |= sb_conv(&ref, seq, pos - 1, pos, L);
                    rc |= buf_put(&alt, ref.p, ref.n); rc |= buf_put(&alt, ins.p, ins.n);
                } else {
                    start += 1;
                    rc |= sb_conv(&ref, seq, pos, pos + 1, L);
                    rc |= buf_put(&alt, ins.p, ins.n); rc |= buf_put(&alt, ref.p, ref.n);
                }
                for (size_t i = 0; i < ins.n; i++) rc |= fw_write(o, ins.p[i]);
                rc |= fw_write(o, (char)seq[pos]);
                rc |= vcf_write(o, name, "INS:ME", start, start, (int64_t)ins.n, ref.p, ref.n, alt.p, alt.n);
            }
        } else {
            rc |= fw_write(o, (char)seq[pos]);                   /* :423 */
        }
        pos += 1;
    }
    free(ref.p); free(alt.p); free(ins.p);
    return rc;
}

static int cmp_rec(const void *a, const void *b) {
    const rec_t *x = (const rec_t *)a, *y = (const rec_t *)b;
    if (x->pos != y->pos) return (x->pos > y->pos) - (x->pos < y->pos);
    return (x->ins_pos > y->ins_pos) - (x->ins_pos < y->ins_pos);   /* insertion order tiebreak */
}

/* mutator.py:111-141: one iteration of mutate()'s contig loop.
 * Returns the merged, position-sorted record list through out_recs (caller frees with orc_release)
 * and *had_muts = whether `muts` was non-empty (drives the warning at mutator.py:125-129). */
int orc_mutate_contig(orc_t *o, const unsigned char *seq, int64_t L, const char *name,
                      const char *long_name, int64_t lenc, const orc_range *ranges, int n_ranges,
                      rec_t **out_recs, int64_t *n_out, int *had_muts) {
    init_tables();
    rec_t *all = NULL; int64_t n_all = 0, cap = 0;
    int64_t *tls = NULL, *tlis = NULL; int64_t n_tls = 0, n_tlis = 0;
    int rc = ORC_OK;
    for (int r = 0; r < n_ranges && rc == ORC_OK; r++) {
        rec_t *rr; int64_t nr; int64_t *a, *b; int64_t na, nb;
        rc = orc_get_mutations(o, &ranges[r], L, &rr, &nr, &a, &na, &b, &nb);
        if (rc) break;
        if (n_all + nr > cap) { cap = (n_all + nr) * 2 + 16; all = (rec_t *)realloc(all, (size_t)cap * sizeof(rec_t)); }
        for (int64_t i = 0; i < nr; i++) { all[n_all] = rr[i]; all[n_all].ins_pos = n_all; /* temp: insertion order */ n_all++; }
        tls = (int64_t *)realloc(tls, (size_t)(n_tls + na + 1) * sizeof(int64_t));
        if (na) memcpy(tls + n_tls, a, (size_t)na * sizeof(int64_t));
        n_tls += na;
        tlis = (int64_t *)realloc(tlis, (size_t)(n_tlis + nb + 1) * sizeof(int64_t));
        if (nb) memcpy(tlis + n_tlis, b, (size_t)nb * sizeof(int64_t));
        n_tlis += nb;
        free(rr); free(a); free(b);
    }
    if (rc) { free(all); free(tls); free(tlis); return rc; }
    /* muts.update(rng_muts): key = pos, later ranges win (mutator.py:121) */
    if (n_all) qsort(all, (size_t)n_all, sizeof(rec_t), cmp_rec);
    int64_t w = 0;
    for (int64_t i = 0; i < n_all; i++) {
        if (i + 1 < n_all && all[i + 1].pos == all[i].pos) continue;
        all[w++] = all[i];
    }
    n_all = w;
    /* Mutation defaults (mutator.py:30-35): trans_reverse False, trans_insert_pos 0; an unlinked
     * TLI keeps start = pos and the stop = 0 that __get_stop_position never touched */
    for (int64_t i = 0; i < n_all; i++) { all[i].ins_pos = 0; all[i].rev = 0; }
    *had_muts = n_all > 0;
    if (n_tls > 0) {                                            /* mutator.py:130-131, :267-316 */
        #define FIND(P, IDX) do { int64_t lo_ = 0, hi_ = n_all; while (lo_ < hi_) { int64_t mid_ = (lo_ + hi_) / 2; \
            if (all[mid_].pos < (P)) lo_ = mid_ + 1; else hi_ = mid_; } IDX = (lo_ < n_all && all[lo_].pos == (P)) ? lo_ : -1; } while (0)
        /* deletions from `muts` are tracked with a tombstone type 0 */
        while (n_tls < n_tlis) {                                /* __fix_tl_amount */
            int64_t idx = orc_randint(o, 0, n_tlis - 1), at; FIND(tlis[idx], at);
            if (at >= 0) all[at].type = 0;
            memmove(tlis + idx, tlis + idx + 1, (size_t)(n_tlis - idx - 1) * sizeof(int64_t)); n_tlis--;
        }
        while (n_tls > n_tlis) {
            int64_t idx = orc_randint(o, 0, n_tls - 1), at; FIND(tls[idx], at);
            if (at >= 0) all[at].type = 0;
            memmove(tls + idx, tls + idx + 1, (size_t)(n_tls - idx - 1) * sizeof(int64_t)); n_tls--;
        }
        orc_shuffle(o, tls, n_tls);
        for (int64_t i = 0; i < n_tls; i++) {
            int64_t tl_at, tli_at; FIND(tls[i], tl_at); FIND(tlis[i], tli_at);
            int64_t tl_start = all[tl_at].pos, tl_stop = all[tl_at].stop;
            int64_t tlen = tl_stop + 1 - tl_start;
            int rev = !(orc_randint(o, 0, 1) == 0 || tlen < 2);   /* __transloc_invert */
            /* muts[tli_pos] = Mutation(TLI, tl_pos, muts[tl_pos].stop, rev, tli_pos) */
            all[tli_at].type = T_TLI; all[tli_at].start = tl_start; all[tli_at].stop = tl_stop;
            all[tli_at].rev = rev; all[tli_at].ins_pos = tlis[i];
        }
        w = 0;
        for (int64_t i = 0; i < n_all; i++) if (all[i].type != 0) all[w++] = all[i];
        n_all = w;
        #undef FIND
    }
    free(tls); free(tlis);
    o->fw_bpl = lenc;                                           /* mutator.py:133-136 */
    rc = fw_header(o, long_name);
    if (!rc) rc = orc_mutate_sequence(o, seq, L, name, all, n_all);
    if (out_recs) { *out_recs = all; *n_out = n_all; } else free(all);
    return rc;
}

/* rec_t accessors for ctypes callers */
void orc_release(void *p) { free(p); }
void orc_set_bpl(orc_t *o, int64_t bpl) { o->fw_bpl = bpl; }
int orc_write_header(orc_t *o, const char *h) { return fw_header(o, h); }
const char *orc_fasta(orc_t *o, uint64_t *n) { *n = o->fasta.n; return o->fasta.p; }
const char *orc_vcf(orc_t *o, uint64_t *n) { *n = o->vcf.n; return o->vcf.p; }
void orc_clear_outputs(orc_t *o) { o->fasta.n = 0; o->vcf.n = 0; o->fw_written = 0; }
char orc_key_error_base(orc_t *o) { return o->key_error_base; }
int orc_sizeof_rec(void) { return (int)sizeof(rec_t); }
int orc_sizeof_range(void) { return (int)sizeof(orc_range); }
