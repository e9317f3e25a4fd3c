// A/B for the host-chain engine's sampling step (BASELINE configs[3]'s file with the SV mix as std line):
//   A  what plan_host.cpp does: the range's bitmap -> ALL k sorted positions (two passes: indices of the non-empty words, then
//      a loop of known length writing two slots per word), of which the boundary chain then gathers its 36 %;
//   B  what round 4's review proposed: keep the bitmap, RANK-SELECT only the chain candidates' positions (their ordinals are
//      known from the device): 512-bit blocks skipped by vpopcntq, the word inside the block by a short scan, the bit by
//      pdep + tzcnt -- and let the device regenerate all positions from the stream cuts.
// Ranges: (n, k) pairs of the real c4sv table (tools/select_vs_extract.py writes them), bitmaps filled with k random bits,
// chain ordinals drawn with probability q = 0.357 each.  Both variants clear the bitmap as the product does (memset).
//   g++ -O3 -march=native -o sve select_vs_extract.cpp && ./sve ranges.bin [reps]
#include <immintrin.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static size_t nonzero_index(const uint64_t *B, size_t nw, uint32_t *idx) {
    size_t nz = 0;
#if defined(__AVX512F__)
    const __m512i step = _mm512_set1_epi32(8);
    __m256i base = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
    size_t i = 0;
    for (; i + 8 <= nw; i += 8) {
        const __m512i v = _mm512_loadu_si512(B + i);
        const __mmask8 m = _mm512_test_epi64_mask(v, v);
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(idx + nz), _mm256_maskz_compress_epi32(m, base));
        nz += (size_t)__builtin_popcount(m);
        base = _mm256_add_epi32(base, _mm512_castsi512_si256(step));
    }
    for (; i < nw; i++) if (B[i]) idx[nz++] = (uint32_t)i;
#else
    for (size_t i = 0; i < nw; i++) if (B[i]) idx[nz++] = (uint32_t)i;
#endif
    return nz;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s ranges.bin [reps]\n", argv[0]); return 2; }
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("ranges"); return 2; }
    std::vector<uint32_t> nk;
    uint32_t buf[2];
    while (fread(buf, 4, 2, f) == 2) { nk.push_back(buf[0]); nk.push_back(buf[1]); }
    fclose(f);
    const size_t R = nk.size() / 2;
    std::mt19937_64 rng(12345);
    // per range: its k distinct values (draw order) and its chain ordinals
    std::vector<std::vector<uint32_t>> vals(R), ords(R);
    size_t K = 0, C = 0, max_nw = 0;
    for (size_t r = 0; r < R; r++) {
        const uint32_t n = nk[2 * r], k = nk[2 * r + 1];
        std::vector<uint64_t> bm((n + 63) / 64 + 1, 0);
        max_nw = std::max(max_nw, bm.size());
        vals[r].reserve(k);
        while (vals[r].size() < k) {
            const uint32_t v = (uint32_t)(rng() % n);
            if (bm[v >> 6] >> (v & 63) & 1) continue;
            bm[v >> 6] |= 1ull << (v & 63);
            vals[r].push_back(v);
        }
        for (uint32_t j = 0; j < k; j++) if ((rng() & 0xffff) < 0.357 * 65536) ords[r].push_back(j);
        K += k; C += ords[r].size();
    }
    std::vector<uint64_t> B(max_nw + 16, 0);
    std::vector<uint32_t> idx(max_nw + max_nw / 2 + 64), pos(1 << 22), sel(1 << 22);
    double tA = 1e30, tB = 1e30, tI = 1e30;
    uint64_t sumA = 0, sumB = 0;
    for (int rep = 0; rep < reps; rep++) {
        // inserts alone (common to both)
        auto t0 = std::chrono::steady_clock::now();
        for (size_t r = 0; r < R; r++) {
            const size_t nw = ((size_t)nk[2 * r] + 63) / 64;
            for (uint32_t v : vals[r]) B[v >> 6] |= 1ull << (v & 63);
            memset(B.data(), 0, nw * 8);
        }
        tI = std::min(tI, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        // A: inserts + full extraction + gather of the chain's positions
        t0 = std::chrono::steady_clock::now();
        sumA = 0;
        for (size_t r = 0; r < R; r++) {
            const uint32_t n = nk[2 * r], k = nk[2 * r + 1];
            const size_t nw = ((size_t)n + 63) / 64;
            for (uint32_t v : vals[r]) B[v >> 6] |= 1ull << (v & 63);
            const size_t nz = nonzero_index(B.data(), nw, idx.data());
            size_t at = 0;
            uint32_t drank = 0;
            const size_t safe = k >= 2 ? (size_t)k - 2 : 0;
            uint32_t *po = pos.data();
            for (size_t j = 0; j < nz; j++) {
                const size_t wi = idx[j];
                uint64_t x = B[wi];
                const uint32_t cn = (uint32_t)__builtin_popcountll(x);
                const uint32_t p0 = (uint32_t)(wi * 64) + drank;
                if (__builtin_expect(cn <= 2 && at <= safe, 1)) {
                    const uint64_t x1 = x & (x - 1), top = 1ull << 63;
                    po[at] = p0 + (uint32_t)__builtin_ctzll(x);
                    po[at + 1] = p0 + 1 + (uint32_t)__builtin_ctzll(x1 | top);
                } else {
                    uint32_t q = 0;
                    while (x) { po[at + q] = p0 + q + (uint32_t)__builtin_ctzll(x); q++; x &= x - 1; }
                }
                at += cn;
                drank += cn;
            }
            memset(B.data(), 0, nw * 8);
            for (uint32_t j : ords[r]) sumA += po[j];                  // (the chain's gather)
        }
        tA = std::min(tA, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        // B: inserts + rank-select of the chain's ordinals only
        t0 = std::chrono::steady_clock::now();
        sumB = 0;
        for (size_t r = 0; r < R; r++) {
            const uint32_t n = nk[2 * r];
            const size_t nw = ((size_t)n + 63) / 64;
            for (uint32_t v : vals[r]) B[v >> 6] |= 1ull << (v & 63);
            const uint64_t *b = B.data();
            size_t blk = 0;                                             // current 512-bit block
            uint32_t before = 0;                                        // set bits in front of it
#if defined(__AVX512VPOPCNTDQ__)
            auto blk_count = [&](size_t q) { return (uint32_t)_mm512_reduce_add_epi64(_mm512_popcnt_epi64(_mm512_loadu_si512(b + 8 * q))); };
#else
            auto blk_count = [&](size_t q) { uint32_t s = 0; for (int i = 0; i < 8; i++) s += (uint32_t)__builtin_popcountll(b[8 * q + i]); return s; };
#endif
            uint32_t cur = blk_count(0);
            for (uint32_t j : ords[r]) {
                while (before + cur <= j) { before += cur; cur = blk_count(++blk); }
                uint32_t rem = j - before;
                size_t wi = 8 * blk;
                uint32_t c;
                while ((c = (uint32_t)__builtin_popcountll(b[wi])) <= rem) { rem -= c; wi++; }
                const uint32_t bit = (uint32_t)__builtin_ctzll(_pdep_u64(1ull << rem, b[wi]));
                sumB += (uint32_t)(wi * 64) + bit + j;
            }
            memset(B.data(), 0, nw * 8);
        }
        tB = std::min(tB, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("%zu ranges, %zu candidates, %zu on the chain (%.1f %%)\n", R, K, C, 100.0 * C / K);
    printf("inserts + clear alone            : %7.2f ms  (%.2f ns per candidate)\n", tI * 1e3, tI * 1e9 / K);
    printf("A  extract all, gather the chain : %7.2f ms  (%.2f ns per candidate; minus inserts %.2f)\n", tA * 1e3, tA * 1e9 / K, (tA - tI) * 1e9 / K);
    printf("B  rank-select the chain only    : %7.2f ms  (%.2f ns per candidate; minus inserts %.2f)\n", tB * 1e3, tB * 1e9 / K, (tB - tI) * 1e9 / K);
    printf("checksums %s (%llu)\n", sumA == sumB ? "equal" : "DIFFER", (unsigned long long)sumA);
    return sumA == sumB ? 0 : 1;
}
