// scan_dev.hpp -- tile vocabulary shared by the scan kernels (scan.hip: whole columns; segscan.hip: per-group scans over the
// flat row-list layout of a grouping): accumulator algebras, the workgroup exclusive scan, blocked tile loads and the
// LDS-transposed tile store.
#pragma once
#include "aqg_internal.hpp"
#include "dev_common.hpp"
#include "chain_dev.hpp"

namespace aqgscan {

constexpr int SB = 256;        // lanes per workgroup
constexpr int IT = 8;          // consecutive elements per lane
constexpr int TS = SB * IT;    // tile
constexpr uint32_t HALO_MAX_BYTES = 96 * 1024;

// ---- accumulator algebra ----------------------------------------------------------------------
// any trivially copyable struct of whole dwords travels word by word
template <class A> __device__ inline A shfl_up_words(A x, int off) {
    static_assert(sizeof(A) % 4 == 0, "carry structs are whole dwords");
    uint32_t w[sizeof(A) / 4];
    __builtin_memcpy(w, &x, sizeof(A));
#pragma unroll
    for (size_t k = 0; k < sizeof(A) / 4; ++k) w[k] = (uint32_t)__shfl_up((int)w[k], off, 64);
    __builtin_memcpy(&x, w, sizeof(A));
    return x;
}
template <class A> __device__ inline A shfl_up_any(A x, int off) {
    if constexpr (std::is_class_v<A> && !std::is_same_v<A, aqg_i128>) return shfl_up_words(x, off);
    else if constexpr (std::is_same_v<A, aqg_i128>) {
        aqg_i128 r;
        r.lo = __shfl_up((unsigned long long)x.lo, off, 64);
        r.hi = __shfl_up((unsigned long long)x.hi, off, 64);
        return r;
    } else return shfl_up_t(x, off);
}
using aqgchain::shfl_xor_any;
template <class A> __device__ inline A shfl_idx_any(A x, int src) {
    if constexpr (std::is_same_v<A, aqg_i128>) {
        aqg_i128 r;
        r.lo = __shfl((unsigned long long)x.lo, src, 64);
        r.hi = __shfl((unsigned long long)x.hi, src, 64);
        return r;
    } else return shfl_idx_t(x, src);
}

// sum accumulator of T: exact integers (64 bits for <=4-byte inputs, 128 for 8-byte), double for fp
template <class T> struct sum_alg {
    using A = std::conditional_t<std::is_floating_point_v<T>, double,
              std::conditional_t<sizeof(T) == 8, aqg_i128, std::conditional_t<std::is_unsigned_v<T>, uint64_t, int64_t>>>;
    __device__ static A identity() { if constexpr (std::is_same_v<A, aqg_i128>) return {0, 0}; else return (A)0; }
    __device__ static A lift(T v) {
        if constexpr (std::is_same_v<A, aqg_i128>) { if constexpr (std::is_unsigned_v<T>) return i128_from_u64(v); else return i128_from_i64(v); }
        else return (A)v;
    }
    __device__ static A op(A a, A b) { if constexpr (std::is_same_v<A, aqg_i128>) return i128_add(a, b); else return a + b; }
    __device__ static A sub(A a, A b) {
        if constexpr (std::is_same_v<A, aqg_i128>) { aqg_i128 nb = {~b.lo + 1, ~b.hi + (b.lo == 0 ? 1ull : 0ull)}; return i128_add(a, nb); }
        else return a - b;
    }
    __device__ static aqg_i128 to_i128(A a) {
        if constexpr (std::is_same_v<A, aqg_i128>) return a;
        else if constexpr (std::is_unsigned_v<A>) return i128_from_u64(a);
        else return i128_from_i64((int64_t)a);
    }
    __device__ static double to_double(A a) {
        if constexpr (std::is_same_v<A, aqg_i128>) {
            if constexpr (std::is_unsigned_v<T>) return u128_to_double(a.hi, a.lo); else return i128_to_double(a);
        } else return (double)a;
    }
};
template <class T> struct min_alg {
    using A = T;
    __device__ static A identity() { return dlimits<T>::max(); }
    __device__ static A lift(T v) { return v; }
    __device__ static A op(A a, A b) { return b < a ? b : a; }
};
template <class T> struct max_alg {
    using A = T;
    // true lowest value, not the reference's seed (the seed is applied when writing `maxs`)
    __device__ static A identity() { if constexpr (std::is_floating_point_v<T>) return -dlimits<T>::max(); else return dlimits<T>::min(); }
    __device__ static A lift(T v) { return v; }
    __device__ static A op(A a, A b) { return b > a ? b : a; }
};

// exclusive scan of one value per lane across the workgroup; `total` = fold of all lanes
template <class ALG, class A> __device__ inline A block_scan_excl(A v, A* lds_w /* >= 5 */, A& total) {
    const int lane = lane_id(), wid = wave_id();
    A incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        A y = shfl_up_any(incl, off);
        if (lane >= off) incl = ALG::op(y, incl);
    }
    if (lane == 63) lds_w[wid] = incl;
    __syncthreads();
    A base = ALG::identity();
    for (int w = 0; w < wid; ++w) base = ALG::op(base, lds_w[w]);
    A tot = ALG::identity();
    for (int w = 0; w < SB / 64; ++w) tot = ALG::op(tot, lds_w[w]);
    total = tot;
    A prev = shfl_up_any(incl, 1);
    if (lane == 0) prev = ALG::identity();
    __syncthreads();
    return ALG::op(base, prev);
}

template <class T> __device__ inline void load_tile_items(const T* __restrict__ x, uint32_t n, uint32_t base, T (&v)[IT], uint32_t& cnt) {
    cnt = base >= n ? 0 : (n - base < (uint32_t)IT ? n - base : IT);
    if (cnt == IT && (((uintptr_t)(x + base)) & (sizeof(T) * IT > 16 ? 15 : sizeof(T) * IT - 1)) == 0) {
        pack<T, IT> p = *reinterpret_cast<const pack<T, IT>*>(x + base);
#pragma unroll
        for (int j = 0; j < IT; ++j) v[j] = p.v[j];
    } else {
#pragma unroll
        for (int j = 0; j < IT; ++j) if ((uint32_t)j < cnt) v[j] = x[base + j];
    }
}

// Blocked results (IT consecutive elements per lane) are written through LDS so that consecutive lanes store consecutive
// elements: a lane-blocked store of 16-byte results touches 64 different 128-B lines per instruction (sums: 2.4 TB/s),
// the transposed one writes whole lines.
template <class O> __device__ inline void store_tile_striped(O* __restrict__ out, uint32_t tile_base, const O (&v)[IT], uint32_t n, O* lds /* TS elements */) {
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < IT; ++j) lds[t * IT + j] = v[j];
    __syncthreads();
    const uint32_t live = tile_base + TS <= n ? TS : n - tile_base;
#pragma unroll
    for (int j = 0; j < IT; ++j) { uint32_t e = j * SB + t; if (e < live) out[tile_base + e] = lds[e]; }
}
template <class T> struct dsum_alg {
    using A = double;
    __device__ static double identity() { return 0; }
    __device__ static double lift(T v) { return (double)v; }
    __device__ static double op(double a, double b) { return a + b; }
    __device__ static double sub(double a, double b) { return a - b; }
    __device__ static double to_double(double a) { return a; }
    __device__ static aqg_i128 to_i128(double) { return {0, 0}; }
};
template <class T> struct sq_alg {   // tile aggregate of x*x in double
    using A = double;
    __device__ static double identity() { return 0; }
    __device__ static double lift(T v) { return (double)v * (double)v; }
    __device__ static double op(double a, double b) { return a + b; }
};


// K2: exclusive scan of the tile aggregates by one workgroup
template <class ALG> __global__ void __launch_bounds__(SB) agg_scan_kernel(typename ALG::A* __restrict__ tile_agg, uint32_t ntiles) {
    using A = typename ALG::A;
    __shared__ A lds_w[8];
    __shared__ A carry_s;
    if (threadIdx.x == 0) carry_s = ALG::identity();
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += SB * IT) {
        uint32_t b = base + threadIdx.x * IT;
        A v[IT];
        A a = ALG::identity();
#pragma unroll
        for (int j = 0; j < IT; ++j) { v[j] = (b + j < ntiles) ? tile_agg[b + j] : ALG::identity(); a = ALG::op(a, v[j]); }
        A total;
        A excl = block_scan_excl<ALG>(a, lds_w, total);
        A run = ALG::op(carry_s, excl);
#pragma unroll
        for (int j = 0; j < IT; ++j) { if (b + j < ntiles) tile_agg[b + j] = run; run = ALG::op(run, v[j]); }
        __syncthreads();
        if (threadIdx.x == 0) carry_s = ALG::op(carry_s, total);
        __syncthreads();
    }
}

// K2 for many tiles: chunks of CH aggregates are reduced by one workgroup each, the few chunk totals are scanned by one
// workgroup, then every chunk is scanned with its carry-in (one workgroup over 488k aggregates took 0.57 ms at 1e9 rows)
constexpr uint32_t CH = SB * IT;
template <class ALG> __global__ void __launch_bounds__(SB) agg_chunk_sum_kernel(const typename ALG::A* __restrict__ tile_agg, uint32_t ntiles, typename ALG::A* __restrict__ chunk_tot) {
    using A = typename ALG::A;
    __shared__ A lds_w[8];
    const uint32_t b = blockIdx.x * CH + threadIdx.x * IT;
    A a = ALG::identity();
#pragma unroll
    for (int j = 0; j < IT; ++j) if (b + j < ntiles) a = ALG::op(a, tile_agg[b + j]);
    A total;
    block_scan_excl<ALG>(a, lds_w, total);
    if (threadIdx.x == 0) chunk_tot[blockIdx.x] = total;
}
template <class ALG> __global__ void __launch_bounds__(SB) agg_chunk_scan_kernel(typename ALG::A* __restrict__ tile_agg, uint32_t ntiles, const typename ALG::A* __restrict__ chunk_excl) {
    using A = typename ALG::A;
    __shared__ A lds_w[8];
    const uint32_t b = blockIdx.x * CH + threadIdx.x * IT;
    A v[IT];
    A a = ALG::identity();
#pragma unroll
    for (int j = 0; j < IT; ++j) { v[j] = (b + j < ntiles) ? tile_agg[b + j] : ALG::identity(); a = ALG::op(a, v[j]); }
    A total;
    A run = ALG::op(chunk_excl[blockIdx.x], block_scan_excl<ALG>(a, lds_w, total));
#pragma unroll
    for (int j = 0; j < IT; ++j) { if (b + j < ntiles) tile_agg[b + j] = run; run = ALG::op(run, v[j]); }
}
// exclusive scan of the tile aggregates in place; `chunk_tot` holds ceil(ntiles / CH) + 1 values of workspace
template <class ALG> void launch_agg_scan(aqg_ctx* ctx, typename ALG::A* tile_agg, uint32_t ntiles, typename ALG::A* chunk_tot) {
    if (ntiles <= 4 * CH || !chunk_tot) { hipLaunchKernelGGL((agg_scan_kernel<ALG>), dim3(1), dim3(SB), 0, ctx->stream, tile_agg, ntiles); return; }
    const uint32_t nch = aqg_ceil_div(ntiles, CH);
    hipLaunchKernelGGL((agg_chunk_sum_kernel<ALG>), dim3(nch), dim3(SB), 0, ctx->stream, tile_agg, ntiles, chunk_tot);
    hipLaunchKernelGGL((agg_scan_kernel<ALG>), dim3(1), dim3(SB), 0, ctx->stream, chunk_tot, nch);
    hipLaunchKernelGGL((agg_chunk_scan_kernel<ALG>), dim3(nch), dim3(SB), 0, ctx->stream, tile_agg, ntiles, chunk_tot);
}


} // namespace aqgscan
