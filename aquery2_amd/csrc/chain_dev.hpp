// Chained links with decoupled look-back: the hand-off protocol of the single-pass scans (scan.hip).
#pragma once
#include "dev_common.hpp"

#include <type_traits>

namespace aqgchain {

template <class A> __device__ inline A shfl_xor_any(A x, int off) {
    if constexpr (std::is_same_v<A, aqg_i128>) return shfl_xor_i128(x, off);
    else return shfl_xor_t(x, off);
}

// Link ids are handed out by an atomic counter, so every predecessor of a running link has started (forward progress).  A link
// publishes its aggregate, then walks back over its predecessors adding aggregates until it meets a published inclusive prefix,
// then publishes its own inclusive prefix.
// Hand-off words: a link owns NW = sizeof(A) / 4 64-bit words, word k = {status : 32 | 32 bits of the value}.  Every word is one
// relaxed agent-scope atomic store (write-through, no drain, no fence: a release would cost a vmcnt(0) round trip per publication
// and a `buffer_wbl2` would write back this XCD's dirty output lines) and one L1-bypassing atomic load.  A reader accepts a link
// when all its words carry the same non-zero status: each word is written once per status, so equal flags mean one publication.
// Measured at 1e9 int32 rows (mins): separate status + payload words with a drain in between 1.95 ms, this protocol with the
// aggregate published before the sub-tile scans 1.50 ms; the same kernel without any look-back 1.32 ms.
// Spins are bounded: on a timeout the kernel raises `err` and the host falls back to the three-kernel scan.
enum : uint32_t { ST_NONE = 0, ST_AGG = 1, ST_PREFIX = 2 };
template <class A> constexpr int flagged_words() { return (int)((sizeof(A) + 3) / 4); }
template <class A> __device__ inline void publish_flagged(uint64_t* slot, uint32_t st, A v) {
    constexpr int NW = flagged_words<A>();
    uint32_t w[NW] = {};
    __builtin_memcpy(w, &v, sizeof(A));
#pragma unroll
    for (int k = 0; k < NW; ++k) __hip_atomic_store(slot + k, ((uint64_t)st << 32) | w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// waits until the link at `slot` has published something; returns its status (ST_NONE: gave up) and the value
template <class A> __device__ inline uint32_t poll_flagged(uint64_t* slot, A& v) {
    constexpr int NW = flagged_words<A>();
    for (uint32_t spins = 0; spins < (1u << 22); ++spins) {
        uint64_t r[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) r[k] = __hip_atomic_load(slot + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t st = (uint32_t)(r[0] >> 32);
        bool ok = st != ST_NONE;
#pragma unroll
        for (int k = 1; k < NW; ++k) ok = ok && (uint32_t)(r[k] >> 32) == st;
        if (ok) {
            uint32_t w[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) w[k] = (uint32_t)r[k];
            __builtin_memcpy(&v, w, sizeof(A));
            return st;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return ST_NONE;
}


// Look-back of link `tile` by ONE wavefront (all 64 lanes call it), 64 predecessors per step: returns the fold of every earlier
// link (ALG::identity() for link 0) after publishing this link's inclusive prefix.  The caller has already published `total`
// as ST_AGG (ST_PREFIX for link 0) -- as early as it could, so that successors rarely wait.  ALG::op must commute.
// `err` is raised when a predecessor never shows up (bounded spins); the result is then meaningless and the host re-runs.
template <class ALG, class A> __device__ inline A lookback(uint64_t* slots, uint32_t tile, A total, uint32_t* err) {
    constexpr int NW = flagged_words<A>();
    const int lane = lane_id();
    A prefix = ALG::identity();
    if (tile == 0) return prefix;
    int64_t p = (int64_t)tile - 1;
    while (true) {
        const int64_t idx = p - lane;                         // lane l inspects predecessor p - l
        uint32_t st = ST_PREFIX;                              // before link 0: an empty prefix
        A val = ALG::identity();
        if (idx >= 0) st = poll_flagged<A>(slots + (size_t)idx * NW, val);
        if (__ballot(st == ST_NONE)) { if (lane == 0) atomicExch(err, 1u); break; }
        const uint64_t pm = __ballot(st == ST_PREFIX);
        const int k = pm ? __ffsll((long long)pm) - 1 : 64;   // nearest predecessor that already has its inclusive prefix
        A contrib = lane <= k ? val : ALG::identity();
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) contrib = ALG::op(contrib, shfl_xor_any(contrib, off));
        prefix = ALG::op(contrib, prefix);
        if (pm) break;
        p -= 64;
    }
    if (lane == 0) publish_flagged<A>(slots + (size_t)tile * NW, ST_PREFIX, ALG::op(prefix, total));
    return prefix;
}

} // namespace aqgchain
