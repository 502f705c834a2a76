// EM seed stream on the host: `np.random.choice(N, K, replace=False)` once per episode in episode order
// (omniglot/model.py:351-356), restated so that a step's draws cost microseconds instead of milliseconds and can
// run on a helper thread without the interpreter lock.
//
// The arithmetic is numpy's (third-party dependency of the reference, not under /root/reference; this image has
// numpy 2.2.6): legacy RandomState.choice(int, size, replace=False, p=None) is permutation(N)[:K]; permutation(N)
// is arange(N) shuffled by RandomState._shuffle_raw — for i = N-1 .. 1: j = random_interval(i), swap(i, j) — and
// random_interval(max) masks 32-bit outputs of the MT19937 generator with the smallest all-ones mask >= max and
// rejects values above max (numpy/random/mtrand.pyx, numpy/random/src/distributions/distributions.c,
// numpy/random/src/mt19937/mt19937.{h,c}).  tests/test_seed_stream.py pins it against numpy itself: same indices,
// same generator state afterwards.  Host code only; no device work.
#include "common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;

inline void mt_gen(uint32_t* key) {
    constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;
    int kk = 0;
    uint32_t y;
    for (; kk < MT_N - MT_M; kk++) {
        y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
        key[kk] = key[kk + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    for (; kk < MT_N - 1; kk++) {
        y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
        key[kk] = key[kk + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    y = (key[MT_N - 1] & UPPER) | (key[0] & LOWER);
    key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
}

struct Mt {
    uint32_t* key;
    int pos;
    inline uint32_t next() {
        if (pos == MT_N) {
            mt_gen(key);
            pos = 0;
        }
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
};

}  // namespace

extern "C" int cmvae_em_seed_draw_host(uint32_t* mt_key_host, int* mt_pos_host, int n_samples, int k, int n_episodes,
                                       int64_t* out_host, int64_t* scratch_host) {
    CMVAE_REQUIRE(mt_key_host && mt_pos_host && out_host && scratch_host, "cmvae_em_seed_draw_host: null argument");
    CMVAE_REQUIRE(n_samples >= 1 && k >= 1 && k <= n_samples && n_episodes >= 0,
                  "cmvae_em_seed_draw_host: need 1 <= K <= N (K=%d, N=%d) and episodes >= 0 (%d)", k, n_samples,
                  n_episodes);
    CMVAE_REQUIRE(*mt_pos_host >= 0 && *mt_pos_host <= MT_N, "cmvae_em_seed_draw_host: MT19937 position %d outside [0, 624]",
                  *mt_pos_host);
    Mt mt{mt_key_host, *mt_pos_host};
    for (int e = 0; e < n_episodes; ++e) {
        for (int i = 0; i < n_samples; ++i) scratch_host[i] = i;
        // numpy: for i = N-1 .. 1: j = random_interval(i) (draw & mask until <= i), swap(i, j).  All i of one power-of-two
        // band share their mask, so the rejection is written branch-free: a rejected draw swaps element i with itself
        // and leaves i where it is (the exit of a rejection loop would be a mispredicted branch every third element).
        int64_t i = n_samples - 1;
        while (i >= 1) {
            uint32_t mask = (uint32_t)i;
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
            const int64_t lo = ((int64_t)mask + 1) >> 1;          // smallest i with this mask (>= 1)
            while (i >= lo) {
                const uint32_t v = mt.next() & mask;
                const bool ok = v <= (uint32_t)i;
                const int64_t j = ok ? (int64_t)v : i;
                const int64_t t = scratch_host[j];
                scratch_host[j] = scratch_host[i];
                scratch_host[i] = t;
                i -= ok;
            }
        }
        for (int j = 0; j < k; ++j) out_host[(int64_t)e * k + j] = scratch_host[j];
    }
    *mt_pos_host = mt.pos;
    return 0;
}
