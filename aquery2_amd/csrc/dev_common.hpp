// dev_common.hpp -- device-side helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#define AQG_WAVE 64

// 128-bit integer result slot (reference: __int128 / unsigned __int128, server/types.h:152-160)
struct alignas(16) aqg_i128 {
    uint64_t lo;
    uint64_t hi;
};
__host__ __device__ static inline aqg_i128 i128_from_i64(int64_t v) { return {(uint64_t)v, v < 0 ? ~0ull : 0ull}; }
__host__ __device__ static inline aqg_i128 i128_from_u64(uint64_t v) { return {v, 0ull}; }
__host__ __device__ static inline aqg_i128 i128_add(aqg_i128 a, aqg_i128 b) {
    aqg_i128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
    return r;
}

// correctly rounded (nearest-even) conversions of 128-bit integers to double
__device__ static inline double u128_to_double(uint64_t hi, uint64_t lo) {
    if (hi == 0) return (double)lo;
    int lz = __clzll((long long)hi);
    uint64_t top = lz ? ((hi << lz) | (lo >> (64 - lz))) : hi;   // top 64 bits; everything below folds into a sticky bit
    uint64_t rest = lz ? (lo << lz) : lo;
    if (rest) top |= 1;
    return ldexp((double)top, 64 - lz);
}
__device__ static inline double i128_to_double(aqg_i128 v) {
    if ((int64_t)v.hi < 0) {
        uint64_t lo = ~v.lo + 1, hi = ~v.hi + (lo == 0 ? 1 : 0);
        return -u128_to_double(hi, lo);
    }
    return u128_to_double(v.hi, v.lo);
}

// 16-byte vector of T for coalesced dwordx4 loads/stores
template <class T> struct alignas(16) vec16 {
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};
template <class T> __device__ static inline vec16<T> load16(const T* p) { return *reinterpret_cast<const vec16<T>*>(p); }
template <class T> __device__ static inline void store16(T* p, const vec16<T>& x) { *reinterpret_cast<vec16<T>*>(p) = x; }

// generic aligned pack of N elements of T (N*sizeof(T) in {4, 8, 16, 32, 64})
template <class T, int N> struct alignas((N * sizeof(T)) > 16 ? 16 : (N * sizeof(T))) pack {
    T v[N];
};

// ---- wave-level primitives -----------------------------------------------------------------
// shuffles for every numeric T (1- and 2-byte types travel as int)
template <class T> __device__ static inline T shfl_xor_t(T x, int off) {
    if constexpr (sizeof(T) < 4) return (T)__shfl_xor((int)x, off, 64);
    else if constexpr (std::is_same_v<T, int64_t>) return (T)__shfl_xor((long long)x, off, 64);
    else if constexpr (std::is_same_v<T, uint64_t>) return (T)__shfl_xor((unsigned long long)x, off, 64);
    else return __shfl_xor(x, off, 64);
}
template <class T> __device__ static inline T shfl_up_t(T x, int off) {
    if constexpr (sizeof(T) < 4) return (T)__shfl_up((int)x, off, 64);
    else if constexpr (std::is_same_v<T, int64_t>) return (T)__shfl_up((long long)x, off, 64);
    else if constexpr (std::is_same_v<T, uint64_t>) return (T)__shfl_up((unsigned long long)x, off, 64);
    else return __shfl_up(x, off, 64);
}
template <class T> __device__ static inline T shfl_idx_t(T x, int src) {
    if constexpr (sizeof(T) < 4) return (T)__shfl((int)x, src, 64);
    else if constexpr (std::is_same_v<T, int64_t>) return (T)__shfl((long long)x, src, 64);
    else if constexpr (std::is_same_v<T, uint64_t>) return (T)__shfl((unsigned long long)x, src, 64);
    else return __shfl(x, src, 64);
}
template <class T, class Op> __device__ static inline T wave_reduce(T x, Op op) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x = op(x, shfl_xor_t(x, off));
    return x;
}
// inclusive scan across the 64 lanes of a wave
template <class T, class Op> __device__ static inline T wave_scan_incl(T x, Op op, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T y = shfl_up_t(x, off);
        if (lane >= off) x = op(y, x);
    }
    return x;
}
__device__ static inline aqg_i128 shfl_xor_i128(aqg_i128 x, int off) {
    aqg_i128 r;
    r.lo = __shfl_xor((unsigned long long)x.lo, off, 64);
    r.hi = __shfl_xor((unsigned long long)x.hi, off, 64);
    return r;
}

// one element of a result column, stored through a call: a run-time switch whose arms STORE, inlined into a loop with a 64-bit index live
// across it, is the shape hipcc 7.2 miscompiled (profiles/r2_hipcc_switch_miscompile.md) -- the arms of such switches store through these
template <class T> __device__ __noinline__ static void store_at(void* __restrict__ col, size_t i, T v) { static_cast<T*>(col)[i] = v; }

struct OpAdd { template <class T> __device__ T operator()(T a, T b) const { return a + b; } };
struct OpMin { template <class T> __device__ T operator()(T a, T b) const { return b < a ? b : a; } };
struct OpMax { template <class T> __device__ T operator()(T a, T b) const { return b > a ? b : a; } };

// Streaming kernels give every workgroup ONE CONTIGUOUS span of the work items instead of a grid-stride interleave: measured on
// MI355X the same kernels run 4-8 % faster that way (h2o Q1 row pass 1.39-1.48 -> 1.31-1.36 ms per 1e9 rows).
__device__ static inline void wg_span(uint32_t total, uint32_t& lo, uint32_t& hi, uint32_t multiple = 1) {
    uint32_t per = (total + gridDim.x - 1) / gridDim.x;
    per = (per + multiple - 1) / multiple * multiple;
    const uint64_t b = (uint64_t)blockIdx.x * per, e = b + per;
    lo = b < total ? (uint32_t)b : total;
    hi = e < total ? (uint32_t)e : total;
}
__device__ static inline int lane_id() { return threadIdx.x & 63; }
__device__ static inline int wave_id() { return threadIdx.x >> 6; }

// numeric_limits on device (reference seeds: max with ::min(), min with ::max(), aggregations.h:73,81)
template <class T> struct dlimits;
#define AQG_DLIM(T, MINV, MAXV)                                        \
    template <> struct dlimits<T> {                                    \
        __host__ __device__ static constexpr T min() { return MINV; }  \
        __host__ __device__ static constexpr T max() { return MAXV; }  \
    };
AQG_DLIM(int8_t, INT8_MIN, INT8_MAX)
AQG_DLIM(int16_t, INT16_MIN, INT16_MAX)
AQG_DLIM(int32_t, INT32_MIN, INT32_MAX)
AQG_DLIM(int64_t, INT64_MIN, INT64_MAX)
AQG_DLIM(uint8_t, 0, UINT8_MAX)
AQG_DLIM(uint16_t, 0, UINT16_MAX)
AQG_DLIM(uint32_t, 0, UINT32_MAX)
AQG_DLIM(uint64_t, 0, UINT64_MAX)
AQG_DLIM(float, 1.17549435e-38f, 3.40282347e+38f)   /* FLT_MIN (smallest positive normal), FLT_MAX */
AQG_DLIM(double, 2.2250738585072014e-308, 1.7976931348623157e+308)
#undef AQG_DLIM

// the reference's GetLongType restricted to what a device accumulator needs:
// signed ints -> int64 lanes (+ carry word for 8-byte inputs), unsigned -> uint64, fp -> double
template <class T> struct acc_of {
    using type = std::conditional_t<std::is_floating_point_v<T>, double,
                                    std::conditional_t<std::is_unsigned_v<T>, uint64_t, int64_t>>;
};
template <class T> using acc_t = typename acc_of<T>::type;

// splitmix64 / counter-based generator shared with oracle/aq_oracle.c
__host__ __device__ static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
