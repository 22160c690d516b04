// reduce.hip -- full-column reductions (reference server/aggregations.h:10-32,71-86,332-348,
// 383-416,487-497): count, sum, avg, min, max, var, stddev, corr, first, last.
//
// One streaming pass (16 B per lane per load, 4 loads in flight) produces per-block partials
// {sum, sum of squares, min, max}; a one-block kernel folds them in a fixed order, so results
// are reproducible run to run.  Integer sums are exact: <=4-byte inputs accumulate in 64 bits
// (|x| < 2^31, n < 2^32), 8-byte inputs in a 128-bit carry pair -- the reference's __int128.
// HBM-bound: algorithmic bytes = n * sizeof(T).
#include "aqg_internal.hpp"
#include "dev_common.hpp"

namespace {

enum : int { ST_SUM = 1, ST_SSQ = 2, ST_MINMAX = 4 };

// raw statistic block shared by every T (values stored in their accumulator representation)
struct alignas(16) stats_raw {
    aqg_i128 sum;   // integers: two's complement 128; fp: double in .lo
    aqg_i128 ssq;
    uint64_t mn;    // T bit pattern, zero-extended
    uint64_t mx;
};

template <class T> __device__ inline uint64_t to_bits(T v) {
    if constexpr (sizeof(T) == 8) return __builtin_bit_cast(uint64_t, v);
    else if constexpr (sizeof(T) == 4) return (uint64_t)__builtin_bit_cast(uint32_t, v);
    else if constexpr (sizeof(T) == 2) return (uint64_t)__builtin_bit_cast(uint16_t, v);
    else return (uint64_t)__builtin_bit_cast(uint8_t, v);
}
template <class T> __host__ __device__ inline T from_bits(uint64_t b) {
    if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, b);
    else if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, (uint32_t)b);
    else if constexpr (sizeof(T) == 2) return __builtin_bit_cast(T, (uint16_t)b);
    else return __builtin_bit_cast(T, (uint8_t)b);
}

// per-thread accumulator: exact for integers, double for fp
template <class T, bool WIDE = (sizeof(T) == 8 && std::is_integral_v<T>)> struct sum_acc;
template <class T> struct sum_acc<T, false> {
    acc_t<T> s;
    __device__ void init() { s = 0; }
    __device__ void add(acc_t<T> v) { s += v; }
    __device__ aqg_i128 get() const {
        if constexpr (std::is_floating_point_v<T>) return {__builtin_bit_cast(uint64_t, (double)s), 0};
        else if constexpr (std::is_unsigned_v<T>) return i128_from_u64((uint64_t)s);
        else return i128_from_i64((int64_t)s);
    }
};
template <class T> struct sum_acc<T, true> {
    aqg_i128 s;
    __device__ void init() { s = {0, 0}; }
    __device__ void add(T v) {
        uint64_t lo = s.lo + (uint64_t)v;
        uint64_t ext = std::is_signed_v<T> ? ((int64_t)v < 0 ? ~0ull : 0ull) : 0ull;
        s.hi += ext + (lo < s.lo ? 1ull : 0ull);
        s.lo = lo;
    }
    __device__ aqg_i128 get() const { return s; }
};

template <bool FP> __device__ inline aqg_i128 combine_sum(aqg_i128 a, aqg_i128 b) {
    if constexpr (FP) return {__builtin_bit_cast(uint64_t, __builtin_bit_cast(double, a.lo) + __builtin_bit_cast(double, b.lo)), 0};
    else return i128_add(a, b);
}
template <bool FP> __device__ inline aqg_i128 wave_sum128(aqg_i128 x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x = combine_sum<FP>(x, shfl_xor_i128(x, off));
    return x;
}

// `arr[i] * arr[i]` in the C++ promoted type of T, wrapping like the reference's x86 builds
template <class T> __device__ inline auto square_promoted(T v) {
    if constexpr (std::is_floating_point_v<T>) return v * v;
    else {
        using P = decltype(v * v);
        using UP = std::make_unsigned_t<P>;
        return (P)((UP)(P)v * (UP)(P)v);
    }
}

template <class T>
__global__ void __launch_bounds__(256) stats_kernel(const T* __restrict__ x, uint32_t n, int flags, stats_raw* __restrict__ partials) {
    constexpr int N = 16 / sizeof(T);
    constexpr bool FP = std::is_floating_point_v<T>;
    sum_acc<T> s, q;
    s.init(); q.init();
    T mn = dlimits<T>::max(), mx = dlimits<T>::min();   // reference seeds (aggregations.h:73,81)

    auto visit = [&](T v) {
        if (flags & ST_SUM) s.add(v);
        if (flags & ST_SSQ) {
            auto p = square_promoted(v);
            if constexpr (sizeof(T) == 8 && std::is_integral_v<T>) q.add((T)p); else q.add((acc_t<T>)p);
        }
        if (flags & ST_MINMAX) { mn = mn < v ? mn : v; mx = mx > v ? mx : v; }
    };

    uint32_t head = (uint32_t)(((16 - ((uintptr_t)x & 15)) & 15) / sizeof(T));
    if (head > n) head = n;
    const T* xb = x + head;
    uint32_t nvec = (n - head) / N;
    const uint32_t stride = blockDim.x;                 // a workgroup walks one contiguous span of the vectors (dev_common.hpp: wg_span)
    uint32_t v_lo, v_hi;
    wg_span(nvec, v_lo, v_hi);
    uint32_t i = v_lo + threadIdx.x;
    // 4 independent 16-byte loads in flight per lane
    for (; i + 3 * (uint64_t)stride < v_hi; i += 4 * stride) {
        vec16<T> a = load16(xb + (size_t)i * N), b = load16(xb + (size_t)(i + stride) * N);
        vec16<T> c = load16(xb + (size_t)(i + 2 * stride) * N), d = load16(xb + (size_t)(i + 3 * stride) * N);
#pragma unroll
        for (int j = 0; j < N; ++j) { visit(a.v[j]); visit(b.v[j]); visit(c.v[j]); visit(d.v[j]); }
    }
    for (; i < v_hi; i += stride) {
        vec16<T> a = load16(xb + (size_t)i * N);
#pragma unroll
        for (int j = 0; j < N; ++j) visit(a.v[j]);
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) visit(x[threadIdx.x]);
        uint32_t t = head + nvec * N + threadIdx.x;
        if (t < n) visit(x[t]);            // tail < N <= 16 elements
    }

    aqg_i128 ws = wave_sum128<FP>(s.get()), wq = wave_sum128<FP>(q.get());
    T wmn = wave_reduce(mn, OpMin{}), wmx = wave_reduce(mx, OpMax{});
    __shared__ stats_raw sh[4];
    if (lane_id() == 0) sh[wave_id()] = {ws, wq, to_bits(wmn), to_bits(wmx)};
    __syncthreads();
    if (threadIdx.x == 0) {
        stats_raw r = sh[0];
        for (int w = 1; w < 4; ++w) {
            r.sum = combine_sum<FP>(r.sum, sh[w].sum);
            r.ssq = combine_sum<FP>(r.ssq, sh[w].ssq);
            T a = from_bits<T>(r.mn), b = from_bits<T>(sh[w].mn);
            r.mn = to_bits(b < a ? b : a);
            a = from_bits<T>(r.mx); b = from_bits<T>(sh[w].mx);
            r.mx = to_bits(b > a ? b : a);
        }
        partials[blockIdx.x] = r;
    }
}

template <class T>
__global__ void __launch_bounds__(64) stats_final_kernel(const stats_raw* __restrict__ partials, uint32_t nparts, stats_raw* __restrict__ out) {
    constexpr bool FP = std::is_floating_point_v<T>;
    // fixed order: lane l folds partials l, l+64, ... then a butterfly
    aqg_i128 s = {FP ? __builtin_bit_cast(uint64_t, 0.0) : 0ull, 0}, q = s;
    T mn = dlimits<T>::max(), mx = dlimits<T>::min();
    for (uint32_t i = threadIdx.x; i < nparts; i += 64) {
        stats_raw p = partials[i];
        s = combine_sum<FP>(s, p.sum);
        q = combine_sum<FP>(q, p.ssq);
        T a = from_bits<T>(p.mn); mn = a < mn ? a : mn;
        a = from_bits<T>(p.mx); mx = a > mx ? a : mx;
    }
    s = wave_sum128<FP>(s); q = wave_sum128<FP>(q);
    mn = wave_reduce(mn, OpMin{}); mx = wave_reduce(mx, OpMax{});
    if (threadIdx.x == 0) *out = {s, q, to_bits(mn), to_bits(mx)};
}

// corr: five sums.  The reference accumulates them in __int128 for EVERY input type
// (aggregations.h:387-389: InnerType is the Coercion struct, so GetLongType gives __int128) and
// truncates a floating term at every step; that sequential truncation is reproduced only for
// integer inputs here (exact); floating inputs are rejected with AQG_ERR_DTYPE.
struct alignas(16) corr_raw { aqg_i128 sx, sy, sxy, sx2, sy2; };

template <class TX, class TY>
__global__ void __launch_bounds__(256) corr_kernel(const TX* __restrict__ x, const TY* __restrict__ y, uint32_t n, corr_raw* __restrict__ partials) {
    // products in the C++ type of the operands (wrapping); sums exact in 128 bits
    aqg_i128 sx = {0, 0}, sy = sx, sxy = sx, sx2 = sx, sy2 = sx;
    auto ext = [](auto v) -> aqg_i128 {
        using V = decltype(v);
        if constexpr (std::is_unsigned_v<V>) return i128_from_u64((uint64_t)v); else return i128_from_i64((int64_t)v);
    };
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        TX a = x[i]; TY b = y[i];
        using PXY = decltype(a * b);
        using UXY = std::make_unsigned_t<PXY>;
        sx = i128_add(sx, ext(a)); sy = i128_add(sy, ext(b));
        sx2 = i128_add(sx2, ext(square_promoted(a)));
        sy2 = i128_add(sy2, ext(square_promoted(b)));
        sxy = i128_add(sxy, ext((PXY)((UXY)(PXY)a * (UXY)(PXY)b)));
    }
    sx = wave_sum128<false>(sx); sy = wave_sum128<false>(sy); sxy = wave_sum128<false>(sxy);
    sx2 = wave_sum128<false>(sx2); sy2 = wave_sum128<false>(sy2);
    __shared__ corr_raw sh[4];
    if (lane_id() == 0) sh[wave_id()] = {sx, sy, sxy, sx2, sy2};
    __syncthreads();
    if (threadIdx.x == 0) {
        corr_raw r = sh[0];
        for (int w = 1; w < 4; ++w) {
            r.sx = i128_add(r.sx, sh[w].sx); r.sy = i128_add(r.sy, sh[w].sy); r.sxy = i128_add(r.sxy, sh[w].sxy);
            r.sx2 = i128_add(r.sx2, sh[w].sx2); r.sy2 = i128_add(r.sy2, sh[w].sy2);
        }
        partials[blockIdx.x] = r;
    }
}
__global__ void __launch_bounds__(64) corr_final_kernel(const corr_raw* __restrict__ partials, uint32_t nparts, corr_raw* __restrict__ out) {
    aqg_i128 z = {0, 0};
    corr_raw r = {z, z, z, z, z};
    for (uint32_t i = threadIdx.x; i < nparts; i += 64) {
        corr_raw p = partials[i];
        r.sx = i128_add(r.sx, p.sx); r.sy = i128_add(r.sy, p.sy); r.sxy = i128_add(r.sxy, p.sxy);
        r.sx2 = i128_add(r.sx2, p.sx2); r.sy2 = i128_add(r.sy2, p.sy2);
    }
    r.sx = wave_sum128<false>(r.sx); r.sy = wave_sum128<false>(r.sy); r.sxy = wave_sum128<false>(r.sxy);
    r.sx2 = wave_sum128<false>(r.sx2); r.sy2 = wave_sum128<false>(r.sy2);
    if (threadIdx.x == 0) *out = r;
}

template <class T> __global__ void pick_kernel(const T* x, uint32_t idx, stats_raw* out) { out->mn = to_bits(x[idx]); }

// host view of a 128-bit device result
inline __int128 as_i128(aqg_i128 v) { return (__int128)(((unsigned __int128)v.hi << 64) | v.lo); }
inline unsigned __int128 as_u128(aqg_i128 v) { return ((unsigned __int128)v.hi << 64) | v.lo; }

template <class T> int run_stats(aqg_ctx* ctx, const T* x, uint32_t n, int flags, stats_raw* host_out, stats_raw** dev_out) {
    // workgroups per CU at 1e9 rows: 1 -> 0.88 ms, 2-4 -> 0.69-0.72 ms, 8 -> 0.73, 32 -> 0.76, 256 -> 1.15 (the one-wavefront fold of the
    // partials grows with the grid)
    unsigned grid = aqg_grid(ctx, n / (16 / sizeof(T)) + 1, 256, 4, 4);
    AQG_TRY(aqg_ws_reset(ctx));
    AQG_TRY(aqg_ws_ensure(ctx, (size_t)(grid + 2) * sizeof(stats_raw) + 1024));
    stats_raw *parts, *fin;
    AQG_TRY(aqg_ws_get(ctx, grid, &parts));
    AQG_TRY(aqg_ws_get(ctx, 1, &fin));
    hipLaunchKernelGGL(stats_kernel<T>, dim3(grid), dim3(256), 0, ctx->stream, x, n, flags, parts);
    hipLaunchKernelGGL(stats_final_kernel<T>, dim3(1), dim3(64), 0, ctx->stream, parts, grid, fin);
    AQG_TRY(aqg_check_launch(ctx, "stats_kernel"));
    if (dev_out) *dev_out = fin;
    if (host_out) {
        AQG_HIP(ctx, hipMemcpyAsync(host_out, fin, sizeof(stats_raw), hipMemcpyDeviceToHost, ctx->stream));
        AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return AQG_OK;
}

int flags_for(int op) {
    switch (op) {
    case AQG_RED_SUM: case AQG_RED_AVG: return ST_SUM;
    case AQG_RED_VAR: case AQG_RED_STDDEV: return ST_SUM | ST_SSQ;
    case AQG_RED_MIN: case AQG_RED_MAX: return ST_MINMAX;
    }
    return 0;
}

} // namespace

extern "C" {

int aqg_reduce_out_dtype(int op, int t) {
    if (!dt_is_num(t)) return AQG_ERROR;
    switch (op) {
    case AQG_RED_SUM: case AQG_RED_SUMSQ: return aqg_long_type(t);
    case AQG_RED_MIN: case AQG_RED_MAX: case AQG_RED_FIRST: case AQG_RED_LAST: return t;
    case AQG_RED_COUNT: return AQG_UINT64;
    case AQG_RED_AVG: case AQG_RED_VAR: case AQG_RED_STDDEV: return AQG_DOUBLE;
    }
    return AQG_ERROR;
}

int aqg_reduce(aqg_ctx* ctx, int op, int t, const void* x, uint32_t n, void* out_host16) {
    if (!ctx || !out_host16 || (!x && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_reduce: bad argument");
    if (op < 0 || op > AQG_RED_LAST) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_reduce: bad op");
    AQG_CHECK_ROWS(ctx, n, "aqg_reduce");
    memset(out_host16, 0, 16);
    if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "reduction: the column dtype is not numeric (128-bit results are not inputs)");
    return aqg_dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        constexpr bool FP = std::is_floating_point_v<T>;
        const T* xp = static_cast<const T*>(x);
        if (op == AQG_RED_COUNT) { uint64_t c = n; memcpy(out_host16, &c, 8); return AQG_OK; }          // :10-13
        if (op == AQG_RED_FIRST || op == AQG_RED_LAST) {                                                  // :487-497
            T v = 0;
            if (n) AQG_TRY(aqg_d2h(ctx, &v, xp + (op == AQG_RED_FIRST ? 0 : n - 1), sizeof(T)));
            memcpy(out_host16, &v, sizeof(T));
            return AQG_OK;
        }
        stats_raw r;
        if (n == 0) {
            r.sum = r.ssq = {0, 0};
            r.mn = 0; r.mx = 0;
            T mn = dlimits<T>::max(), mx = dlimits<T>::min();
            memcpy(&r.mn, &mn, sizeof(T)); memcpy(&r.mx, &mx, sizeof(T));
        } else {
            AQG_TRY(run_stats<T>(ctx, xp, n, flags_for(op), &r, nullptr));
        }
        switch (op) {
        case AQG_RED_SUM: memcpy(out_host16, &r.sum, FP ? 8 : 16); break;                               // :19-27
        case AQG_RED_MIN: memcpy(out_host16, &r.mn, sizeof(T)); break;                                  // :79-86
        case AQG_RED_MAX: memcpy(out_host16, &r.mx, sizeof(T)); break;                                  // :71-78
        case AQG_RED_AVG: {                                                                             // :28-32
            double d;
            if constexpr (FP) d = __builtin_bit_cast(double, r.sum.lo) / (double)n;
            else if constexpr (std::is_unsigned_v<T>) d = (double)as_u128(r.sum) / (double)n;
            else d = (double)as_i128(r.sum) / (double)n;
            memcpy(out_host16, &d, 8);
        } break;
        case AQG_RED_VAR: case AQG_RED_STDDEV: {                                                        // :332-348,413-416
            double d, np1 = (double)(uint32_t)(n + 1);
            if constexpr (FP) {
                double s = __builtin_bit_cast(double, r.sum.lo), q = __builtin_bit_cast(double, r.ssq.lo);
                d = (q - s * s / np1) / np1;
            } else if constexpr (std::is_unsigned_v<T>) {
                unsigned __int128 s = as_u128(r.sum), q = as_u128(r.ssq);
                d = ((double)q - (double)(s * s) / np1) / np1;
            } else {
                __int128 s = as_i128(r.sum), q = as_i128(r.ssq);
                __int128 ss = (__int128)((unsigned __int128)s * (unsigned __int128)s);
                d = ((double)q - (double)ss / np1) / np1;
            }
            if (op == AQG_RED_STDDEV) d = sqrt(d);
            memcpy(out_host16, &d, 8);
        } break;
        }
        return AQG_OK;
    });
}

// asynchronous form: SUM / MIN / MAX / COUNT / FIRST / LAST leave their 16-byte slot in device
// memory without a host round trip (AVG / VAR / STDDEV need the host epilogue: use aqg_reduce).
int aqg_reduce_dev(aqg_ctx* ctx, int op, int t, const void* x, uint32_t n, void* out_dev16) {
    if (!ctx || !out_dev16 || (!x && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_reduce_dev: bad argument");
    if (op == AQG_RED_AVG || op == AQG_RED_VAR || op == AQG_RED_STDDEV) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_reduce_dev: op needs the host epilogue");
    AQG_CHECK_ROWS(ctx, n, "aqg_reduce_dev");
    AQG_HIP(ctx, hipMemsetAsync(out_dev16, 0, 16, ctx->stream));
    if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "reduction: the column dtype is not numeric (128-bit results are not inputs)");
    return aqg_dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        const T* xp = static_cast<const T*>(x);
        if (op == AQG_RED_COUNT) {
            uint64_t c = n;
            void* st; AQG_TRY(aqg_host_stage(ctx, 16, &st)); memcpy(st, &c, 8);
            AQG_HIP(ctx, hipMemcpyAsync(out_dev16, st, 8, hipMemcpyHostToDevice, ctx->stream));
            return AQG_OK;
        }
        if (op == AQG_RED_FIRST || op == AQG_RED_LAST) {
            if (n) AQG_HIP(ctx, hipMemcpyAsync(out_dev16, xp + (op == AQG_RED_FIRST ? 0 : n - 1), sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
            return AQG_OK;
        }
        if (n == 0) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_reduce_dev: empty column");
        stats_raw* fin = nullptr;
        AQG_TRY(run_stats<T>(ctx, xp, n, flags_for(op), nullptr, &fin));
        const void* src = op == AQG_RED_SUM ? (const void*)&fin->sum : op == AQG_RED_MIN ? (const void*)&fin->mn : (const void*)&fin->mx;
        size_t bytes = op == AQG_RED_SUM ? (std::is_floating_point_v<T> ? 8 : 16) : sizeof(T);
        AQG_HIP(ctx, hipMemcpyAsync(out_dev16, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        return AQG_OK;
    });
}

// internal (sharded.hip): {sum 16 B | sum of squares 16 B | min 8 B | max 8 B} of a column (the representation of aqg_reduce's epilogue:
// integer sums two's complement 128-bit, floating ones a double in the low word; min / max the element's bits) copied to `out_dev48`.
// flags: 1 sum, 2 sum of squares, 4 min / max.  n > 0.
int aqg_stats_dev(aqg_ctx* ctx, int t, const void* x, uint32_t n, int flags, void* out_dev48) {
    return aqg_dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        stats_raw* fin = nullptr;
        AQG_TRY(run_stats<T>(ctx, static_cast<const T*>(x), n, flags, nullptr, &fin));
        AQG_HIP(ctx, hipMemcpyAsync(out_dev48, fin, sizeof(stats_raw), hipMemcpyDeviceToDevice, ctx->stream));
        return AQG_OK;
    });
}
// internal (sharded.hip): the five 128-bit sums of corr (sx, sy, sxy, sx2, sy2) copied to `out_dev80`; integer columns, n > 0
int aqg_corr_sums_dev(aqg_ctx* ctx, int tx, const void* x, int ty, const void* y, uint32_t n, void* out_dev80) {
    auto int_only = [&](int dt, auto&& f) -> int {
        switch (dt) {
        case AQG_INT8: return f(aqg_tag<int8_t>{});
        case AQG_INT16: return f(aqg_tag<int16_t>{});
        case AQG_INT32: return f(aqg_tag<int32_t>{});
        case AQG_INT64: return f(aqg_tag<int64_t>{});
        case AQG_UINT8: return f(aqg_tag<uint8_t>{});
        case AQG_UINT16: return f(aqg_tag<uint16_t>{});
        case AQG_UINT32: return f(aqg_tag<uint32_t>{});
        }
        return aqg_fail(ctx, AQG_ERR_DTYPE, "corr: integer columns (floating inputs are accumulated with per-step truncation in the reference; not offered on device)");
    };
    return int_only(tx, [&](auto ta) -> int {
        return int_only(ty, [&](auto tb) -> int {
            using TX = typename decltype(ta)::type; using TY = typename decltype(tb)::type;
            unsigned grid = aqg_grid(ctx, n, 256, 8, 8);
            AQG_TRY(aqg_ws_reset(ctx));
            AQG_TRY(aqg_ws_ensure(ctx, (size_t)(grid + 2) * sizeof(corr_raw) + 1024));
            corr_raw *parts, *fin;
            AQG_TRY(aqg_ws_get(ctx, grid, &parts));
            AQG_TRY(aqg_ws_get(ctx, 1, &fin));
            hipLaunchKernelGGL((corr_kernel<TX, TY>), dim3(grid), dim3(256), 0, ctx->stream, (const TX*)x, (const TY*)y, n, parts);
            hipLaunchKernelGGL(corr_final_kernel, dim3(1), dim3(64), 0, ctx->stream, parts, grid, fin);
            AQG_TRY(aqg_check_launch(ctx, "corr_kernel"));
            AQG_HIP(ctx, hipMemcpyAsync(out_dev80, fin, sizeof(corr_raw), hipMemcpyDeviceToDevice, ctx->stream));
            return AQG_OK;
        });
    });
}

int aqg_corr(aqg_ctx* ctx, int tx, const void* x, int ty, const void* y, uint32_t n, double* out_host) {
    if (!ctx || !out_host || ((!x || !y) && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_corr: bad argument");
    AQG_CHECK_ROWS(ctx, n, "aqg_corr");
    if (dt_is_fp(tx) || dt_is_fp(ty)) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_corr: floating inputs are accumulated with per-step truncation in the reference; not offered on device");
    int inner = aqg_coercion(tx, ty);
    if (inner == AQG_ERROR || inner == AQG_STR) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_corr: no coercion");
    auto int_only = [&](int dt, auto&& f) -> int {
        switch (dt) {
        case AQG_INT8: return f(aqg_tag<int8_t>{});
        case AQG_INT16: return f(aqg_tag<int16_t>{});
        case AQG_INT32: return f(aqg_tag<int32_t>{});
        case AQG_INT64: return f(aqg_tag<int64_t>{});
        case AQG_UINT8: return f(aqg_tag<uint8_t>{});
        case AQG_UINT16: return f(aqg_tag<uint16_t>{});
        case AQG_UINT32: return f(aqg_tag<uint32_t>{});
        }
        return AQG_ERR_DTYPE;
    };
    return int_only(tx, [&](auto ta) -> int {
        return int_only(ty, [&](auto tb) -> int {
            using TX = typename decltype(ta)::type; using TY = typename decltype(tb)::type;
            corr_raw r;
            memset(&r, 0, sizeof r);
            if (n) {
                unsigned grid = aqg_grid(ctx, n, 256, 8, 8);
                AQG_TRY(aqg_ws_reset(ctx));
                AQG_TRY(aqg_ws_ensure(ctx, (size_t)(grid + 2) * sizeof(corr_raw) + 1024));
                corr_raw *parts, *fin;
                AQG_TRY(aqg_ws_get(ctx, grid, &parts));
                AQG_TRY(aqg_ws_get(ctx, 1, &fin));
                hipLaunchKernelGGL((corr_kernel<TX, TY>), dim3(grid), dim3(256), 0, ctx->stream, (const TX*)x, (const TY*)y, n, parts);
                hipLaunchKernelGGL(corr_final_kernel, dim3(1), dim3(64), 0, ctx->stream, parts, grid, fin);
                AQG_TRY(aqg_check_launch(ctx, "corr_kernel"));
                AQG_TRY(aqg_d2h(ctx, &r, fin, sizeof r));
            }
            // (len*sxy - FP(sx*sy)) / sqrt((len*sx2 - FP(sx*sx)) * (len*sy2 - FP(sy*sy)))  :401-406
            __int128 sx = as_i128(r.sx), sy = as_i128(r.sy), sxy = as_i128(r.sxy), sx2 = as_i128(r.sx2), sy2 = as_i128(r.sy2);
            auto mulw = [](__int128 a, __int128 b) { return (__int128)((unsigned __int128)a * (unsigned __int128)b); };
            *out_host = ((double)mulw(n, sxy) - (double)mulw(sx, sy)) /
                        sqrt(((double)mulw(n, sx2) - (double)mulw(sx, sx)) * ((double)mulw(n, sy2) - (double)mulw(sy, sy)));
            return AQG_OK;
        });
    });
}

} // extern "C"
