// The windowed acceptance rate of adaptMCMC (adaptMCMC_source.cpp:194-203) on the device.
//
// The reference prepends every outcome from iteration 2 on to `acceptance_series`, drops the oldest once the series
// is longer than `acceptance_window`, and takes the mean: rate = (accepted among the last min(i-1, window)
// outcomes) / min(i-1, window), an exact small-integer quotient.
//   * window <= 64: the outcomes are the low bits of ONE 64-bit word per chain, newest in bit 0 (no memory).
//   * window  > 64: a ring of ceil(window / 64) words per chain in HBM, [word][chain]; outcome number p (0-based,
//     modulo window) is bit p % 64 of word p / 64.  The word being written stays in registers and touches memory once
//     per 64 iterations; `count` is the number of accepted outcomes held, updated by the new outcome and the one it
//     overwrites.  All chains of a launch are at the same iteration, so the ring position is wave-uniform.
#pragma once
#include "fastmath.h"

namespace mcmc {

constexpr int kMaxAcceptanceWindow = 65536; // 1 024 ring words = 8 KB per chain

struct AcceptWindow {
    uint64_t bits = 0; // window <= 64: the outcomes; window > 64: the ring word that holds position `pos`
    int len = 0;       // outcomes held (<= window)
    int count = 0;     // window > 64: accepted outcomes among them
};

__host__ __device__ constexpr int accept_window_words(int window) { return window > 64 ? (window + 63) / 64 : 0; }
// ring position of the next outcome when `i_done` iterations are behind the chain (iteration 1 is never entered)
__host__ __device__ constexpr int accept_window_pos(int i_done, int window)
{
    return (window > 64 && i_done > 1) ? (i_done - 1) % window : 0;
}

// Enters the outcome of one iteration (i >= 2) and returns the new rate.  ring: word 0 of THIS chain (stride n between
// words), only touched when window > 64; `store`: this thread writes the completed words (one thread per chain does).
__device__ __forceinline__ double accept_window_push(AcceptWindow &w, bool accept, int window, int &pos, unsigned long long *ring,
                                                     int64_t n, bool store)
{
    const int a1 = accept ? 1 : 0;
    if (window <= 64) {
        w.bits = (w.bits << 1) | (uint64_t)a1;
        if (w.len < window) ++w.len;
        const uint64_t m = (w.len >= 64) ? ~0ull : ((1ull << w.len) - 1ull);
        return div_rn((double)__popcll(w.bits & m), (double)w.len);
    }
    const int b = pos & 63;
    const bool full = w.len == window;
    const int old = full ? (int)((w.bits >> b) & 1ull) : 0; // the outcome `window` iterations back leaves the series
    w.bits = (w.bits & ~(1ull << b)) | ((uint64_t)a1 << b);
    w.count += a1 - old;
    if (!full) ++w.len;
    const double rate = div_rn((double)w.count, (double)w.len);
    const int npos = (pos + 1 == window) ? 0 : pos + 1;
    if ((npos & 63) == 0) { // this word is complete, or the ring wraps: put it away and take up the next one
        // One thread of the chain stores the finished word; ALL threads that carry the chain's scalars (the lanes of its
        // row, the waves of its workgroup) load the next one -- which the storing thread put away the last time the ring
        // passed it: a whole lap earlier for long windows, but only window - 64 >= 1 iterations earlier when the window
        // has two words (65..127).  The pair is a release store / acquire load at agent scope (the load must not be
        // served by a vector L1 line older than the store); between the two there is program order inside a wavefront
        // (row kernel) or the iteration's barriers (per-workgroup kernel), never less than one iteration.
        if (store) __hip_atomic_store(ring + (int64_t)(pos >> 6) * n, w.bits, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        w.bits = (w.len == window)
                     ? __hip_atomic_load(ring + (int64_t)(npos >> 6) * n, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                     : 0ull;
    }
    pos = npos;
    return rate;
}

} // namespace mcmc
