// MT19937 core shared by the host planner and the device kernels.
//
// Both generators on the reference's path are MT19937: CPython's `random` (seeded through
// init_by_array) and NumPy's legacy RandomState (seeded through init_genrand).  Only the seeding
// differs; the recurrence and tempering are the published algorithm (Matsumoto & Nishimura 1998).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MSIM_HD __host__ __device__ __forceinline__
#else
#define MSIM_HD inline
#endif

namespace msim {

constexpr int MT_N = 624;
constexpr int MT_M = 397;

MSIM_HD uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// x[k+624] from x[k], x[k+1], x[k+397]
MSIM_HD uint32_t mt_twist(uint32_t xk, uint32_t xk1, uint32_t xk397) {
    uint32_t y = (xk & 0x80000000u) | (xk1 & 0x7fffffffu);
    return xk397 ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// Sequential host generator.  `words` counts outputs drawn (stream position bookkeeping).
// A regeneration produces the next 624 raw words in three dependency-free loops and tempers them all
// at once into `out`, so both loops vectorise (an AVX2 clone is selected at load time where available).
struct HostMT {
    uint32_t mt[MT_N];                 // raw state (what random.getstate() / numpy get_state() hold)
    uint32_t out[MT_N];                // tempered outputs of the current block
    int idx = MT_N;
    uint64_t words = 0;

    void init_genrand(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < MT_N; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = MT_N;
    }
    void init_by_array(const uint32_t *key, int klen) {
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int k = (MT_N > klen ? MT_N : klen); k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
            if (++j >= klen) j = 0;
        }
        for (int k = MT_N - 1; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            if (++i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
        }
        mt[0] = 0x80000000u;
        idx = MT_N;
    }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __attribute__((target_clones("avx2", "default")))
#endif
    void block(bool advance) {
        if (advance) {
            int k = 0;
            for (; k < MT_N - MT_M; k++) mt[k] = mt_twist(mt[k], mt[k + 1], mt[k + MT_M]);
            for (; k < MT_N - 1; k++) mt[k] = mt_twist(mt[k], mt[k + 1], mt[k + MT_M - MT_N]);
            mt[MT_N - 1] = mt_twist(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
        }
        for (int k = 0; k < MT_N; k++) out[k] = mt_temper(mt[k]);
    }
    void regenerate() { block(true); idx = 0; }
    // the raw state was set from outside (msim_set_mt_state, device window): refresh the tempered block
    void state_changed() { block(false); }
    inline uint32_t next() {
        if (idx >= MT_N) regenerate();
        words++;
        return out[idx++];
    }
    // 53-bit sample shared by random.random() and NumPy's random_sample(): (a>>5)*2^26 + (b>>6)
    inline uint64_t next53() {
        uint64_t a = next() >> 5, b = next() >> 6;
        return (a << 26) | b;
    }
};

inline int bit_length64(uint64_t n) { return n ? 64 - __builtin_clzll(n) : 0; }

}  // namespace msim
