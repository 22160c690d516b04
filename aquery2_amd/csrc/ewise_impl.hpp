// ewise_impl.hpp -- kernels and dispatch of aqg_ewise, templated on the compute type C.  One translation unit per C
// (ewise_i32.hip ... ewise_f64.hip) instantiates aqgew::dispatch_ot<C>, so the 72 (C, OT) kernels compile in parallel;
// ewise.hip holds the C entry points.
#pragma once
#include "aqg_internal.hpp"
#include "dev_common.hpp"

namespace aqgew {

// elements per lane per vector: chosen so that a lane STORES exactly 16 bytes (int128 results: 1 element, bool: 16);
// each lane handles UNR such vectors per step, every one a separate fully coalesced access
template <class OT> constexpr int elems_for() { return (int)(16 / sizeof(OT)); }

template <class C, class T, int E> __device__ inline void load_chunk_t(const void* p, size_t base, C (&o)[E]) {
    pack<T, E> v = *reinterpret_cast<const pack<T, E>*>(static_cast<const T*>(p) + base);
#pragma unroll
    for (int j = 0; j < E; ++j) o[j] = (C)v.v[j];
}
template <class C, int E> __device__ inline void load_chunk(const void* p, int dt, size_t base, C (&o)[E]) {
    switch (dt) {
    case AQG_INT8: load_chunk_t<C, int8_t, E>(p, base, o); break;
    case AQG_INT16: load_chunk_t<C, int16_t, E>(p, base, o); break;
    case AQG_INT32: load_chunk_t<C, int32_t, E>(p, base, o); break;
    case AQG_INT64: load_chunk_t<C, int64_t, E>(p, base, o); break;
    case AQG_BOOL: case AQG_UINT8: load_chunk_t<C, uint8_t, E>(p, base, o); break;
    case AQG_UINT16: load_chunk_t<C, uint16_t, E>(p, base, o); break;
    case AQG_UINT32: load_chunk_t<C, uint32_t, E>(p, base, o); break;
    case AQG_UINT64: load_chunk_t<C, uint64_t, E>(p, base, o); break;
    case AQG_FLOAT: load_chunk_t<C, float, E>(p, base, o); break;
    default: load_chunk_t<C, double, E>(p, base, o); break;
    }
}
template <class C> __device__ inline C load_one(const void* p, int dt, size_t i) {
    switch (dt) {
    case AQG_INT8: return (C) static_cast<const int8_t*>(p)[i];
    case AQG_INT16: return (C) static_cast<const int16_t*>(p)[i];
    case AQG_INT32: return (C) static_cast<const int32_t*>(p)[i];
    case AQG_INT64: return (C) static_cast<const int64_t*>(p)[i];
    case AQG_BOOL: case AQG_UINT8: return (C) static_cast<const uint8_t*>(p)[i];
    case AQG_UINT16: return (C) static_cast<const uint16_t*>(p)[i];
    case AQG_UINT32: return (C) static_cast<const uint32_t*>(p)[i];
    case AQG_UINT64: return (C) static_cast<const uint64_t*>(p)[i];
    case AQG_FLOAT: return (C) static_cast<const float*>(p)[i];
    default: return (C) static_cast<const double*>(p)[i];
    }
}

// conversion of a computed value to the output element type
template <class OT, class R> __device__ inline OT convert_out(R r) {
    if constexpr (std::is_same_v<OT, aqg_i128>) {
        static_assert(std::is_integral_v<R>, "128-bit results come from integer arithmetic");
        if constexpr (std::is_unsigned_v<R>) return i128_from_u64((uint64_t)r); else return i128_from_i64((int64_t)r);
    } else if constexpr (std::is_same_v<OT, bool>) {
        return r != 0;
    } else return (OT)r;
}

template <class C> __device__ inline C safe_div(C a, C b) {
    if constexpr (std::is_floating_point_v<C>) return a / b;
    else {
        if (b == 0) return 0;                                   // the reference traps (SIGFPE); defined as 0 here
        if constexpr (std::is_signed_v<C>) if (b == (C)-1) return (C)(0 - (std::make_unsigned_t<C>)a);
        return a / b;
    }
}
template <class C> __device__ inline C safe_mod(C a, C b) {
    if constexpr (std::is_floating_point_v<C>) return 0;
    else {
        if (b == 0) return 0;
        if constexpr (std::is_signed_v<C>) if (b == (C)-1) return 0;
        return a % b;
    }
}

template <int OP, class C, class OT> __device__ inline OT apply(C a, C b) {
    if constexpr (OP == AQG_OP_ADD) return convert_out<OT>((C)(a + b));
    else if constexpr (OP == AQG_OP_SUB) return convert_out<OT>((C)(a - b));
    else if constexpr (OP == AQG_OP_MUL) return convert_out<OT>((C)(a * b));
    else if constexpr (OP == AQG_OP_DIV) return convert_out<OT>(safe_div(a, b));
    else if constexpr (OP == AQG_OP_MOD) return convert_out<OT>(safe_mod(a, b));
    else if constexpr (OP == AQG_OP_AND || OP == AQG_OP_OR || OP == AQG_OP_XOR) {
        if constexpr (std::is_integral_v<C>) {
            if constexpr (OP == AQG_OP_AND) return convert_out<OT>((C)(a & b));
            else if constexpr (OP == AQG_OP_OR) return convert_out<OT>((C)(a | b));
            else return convert_out<OT>((C)(a ^ b));
        } else return convert_out<OT>(0);
    }
    else if constexpr (OP == AQG_OP_GT) return convert_out<OT>((int)(a > b));
    else if constexpr (OP == AQG_OP_LT) return convert_out<OT>((int)(a < b));
    else if constexpr (OP == AQG_OP_GE) return convert_out<OT>((int)(a >= b));
    else if constexpr (OP == AQG_OP_LE) return convert_out<OT>((int)(a <= b));
    else if constexpr (OP == AQG_OP_EQ) return convert_out<OT>((int)(a == b));
    else return convert_out<OT>((int)(a != b));
}

// vectors per lane (and per workgroup: the grid is exact, the loop runs once).  Measured at 1e9 rows: int32 results (E = 4)
// 2.09 / 2.05 / 2.02 ms with 4 / 2 / 1 vectors; double results (E = 2) 2.08 / 2.07 / 2.53 ms with 8 / 4 / 1; int128 results
// (E = 1) 4.19 / 5.43 ms with 8 / 1.
template <int E> constexpr int ew_unr() { return E >= 4 ? 1 : (E == 2 ? 4 : 8); }

template <int OP, class C, class OT>
__device__ inline void ewise_body(int kind, int lt, const void* l, int rt, const void* r, C sc, OT* out, uint32_t n, int vec_ok) {
    constexpr int E = elems_for<OT>();
    constexpr int UNR = ew_unr<E>();
    const uint32_t nvec = vec_ok ? n / E : 0;                       // operands not 16-byte aligned: everything goes the scalar way
    const uint32_t stride = blockDim.x;                      // a workgroup covers UNR * 256 consecutive vectors per step (one 4-64 KB span)
    for (uint64_t c = blockIdx.x; c * UNR * blockDim.x < nvec; c += gridDim.x) {
        const uint32_t v0 = (uint32_t)(c * UNR * blockDim.x) + threadIdx.x;
        C a[UNR][E], b[UNR][E];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint32_t vid = v0 + u * stride;
            if (vid < nvec) {
                const size_t base = (size_t)vid * E;
                if (kind == AQG_SCALAR_VEC) {
#pragma unroll
                    for (int j = 0; j < E; ++j) a[u][j] = sc;
                } else load_chunk<C, E>(l, lt, base, a[u]);
                if (kind == AQG_VEC_SCALAR) {
#pragma unroll
                    for (int j = 0; j < E; ++j) b[u][j] = sc;
                } else load_chunk<C, E>(r, rt, base, b[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint32_t vid = v0 + u * stride;
            if (vid < nvec) {
                pack<OT, E> o;
#pragma unroll
                for (int j = 0; j < E; ++j) o.v[j] = apply<OP, C, OT>(a[u][j], b[u][j]);
                *reinterpret_cast<pack<OT, E>*>(out + (size_t)vid * E) = o;
            }
        }
    }
    for (uint64_t i = (uint64_t)nvec * E + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        C a = kind == AQG_SCALAR_VEC ? sc : load_one<C>(l, lt, i);
        C b = kind == AQG_VEC_SCALAR ? sc : load_one<C>(r, rt, i);
        out[i] = apply<OP, C, OT>(a, b);
    }
}

template <class C, class OT>
__global__ void __launch_bounds__(256) ewise_kernel(int op, int kind, int lt, const void* __restrict__ l, int rt,
                                                    const void* __restrict__ r, C sc, OT* __restrict__ out, uint32_t n, int vec_ok) {
    switch (op) { // wave-uniform
    case AQG_OP_ADD: ewise_body<AQG_OP_ADD>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_SUB: ewise_body<AQG_OP_SUB>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_MUL: ewise_body<AQG_OP_MUL>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_DIV: ewise_body<AQG_OP_DIV>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_MOD: ewise_body<AQG_OP_MOD>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_AND: ewise_body<AQG_OP_AND>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_OR: ewise_body<AQG_OP_OR>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_XOR: ewise_body<AQG_OP_XOR>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_GT: ewise_body<AQG_OP_GT>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_LT: ewise_body<AQG_OP_LT>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_GE: ewise_body<AQG_OP_GE>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_LE: ewise_body<AQG_OP_LE>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    case AQG_OP_EQ: ewise_body<AQG_OP_EQ>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    default: ewise_body<AQG_OP_NE>(kind, lt, l, rt, r, sc, out, n, vec_ok); break;
    }
}

// host: C++ integer promotion + usual arithmetic conversions -> compute class tag
inline int promote1(int dt) {
    switch (dt) {
    case AQG_FLOAT: case AQG_DOUBLE: case AQG_INT64: case AQG_UINT64: case AQG_UINT32: return dt;
    default: return AQG_INT32;
    }
}
inline int usual_conv(int lt, int rt) {
    int a = promote1(lt), b = promote1(rt);
    if (a == AQG_DOUBLE || b == AQG_DOUBLE) return AQG_DOUBLE;
    if (a == AQG_FLOAT || b == AQG_FLOAT) return AQG_FLOAT;
    if (a == AQG_UINT64 || b == AQG_UINT64) return AQG_UINT64;
    if (a == AQG_INT64 || b == AQG_INT64) return AQG_INT64;
    if (a == AQG_UINT32 || b == AQG_UINT32) return AQG_UINT32;
    return AQG_INT32;
}
template <class C> C host_scalar(int dt, const void* p) {
    switch (dt) {
    case AQG_INT8: return (C) * static_cast<const int8_t*>(p);
    case AQG_INT16: return (C) * static_cast<const int16_t*>(p);
    case AQG_INT32: return (C) * static_cast<const int32_t*>(p);
    case AQG_INT64: return (C) * static_cast<const int64_t*>(p);
    case AQG_BOOL: case AQG_UINT8: return (C) * static_cast<const uint8_t*>(p);
    case AQG_UINT16: return (C) * static_cast<const uint16_t*>(p);
    case AQG_UINT32: return (C) * static_cast<const uint32_t*>(p);
    case AQG_UINT64: return (C) * static_cast<const uint64_t*>(p);
    case AQG_FLOAT: return (C) * static_cast<const float*>(p);
    default: return (C) * static_cast<const double*>(p);
    }
}

template <class C, class OT>
int launch_ewise(aqg_ctx* ctx, int op, int kind, int lt, const void* l, int rt, const void* r, void* out, uint32_t n, int vec_ok) {
    if constexpr (std::is_same_v<OT, aqg_i128> && std::is_floating_point_v<C>) {
        return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_ewise: 128-bit result from floating arithmetic");
    } else {
        C sc = 0;
        if (kind == AQG_VEC_SCALAR) sc = host_scalar<C>(rt, r);
        if (kind == AQG_SCALAR_VEC) sc = host_scalar<C>(lt, l);
        // one chunk of UNR * 256 vectors per workgroup (the kernel's loop then runs once): measured 72-74 % of the HBM roofline
        // against 61-68 % with a capped grid and a grid-stride loop
        constexpr int E_ = (int)(16 / sizeof(OT));
        constexpr int UNR_ = ew_unr<E_>();
        const uint64_t per_wg = (uint64_t)UNR_ * 256, nvec_ = n / E_;
        const uint64_t want = (nvec_ + per_wg - 1) / per_wg;
        unsigned grid = (unsigned)(want < 1 ? 1 : want);
        hipLaunchKernelGGL((ewise_kernel<C, OT>), dim3(grid), dim3(256), 0, ctx->stream, op, kind, lt,
                           kind == AQG_SCALAR_VEC ? nullptr : l, rt, kind == AQG_VEC_SCALAR ? nullptr : r, sc,
                           static_cast<OT*>(out), n, vec_ok);
        return aqg_check_launch(ctx, "ewise_kernel");
    }
}

template <class C> int dispatch_ot(aqg_ctx* ctx, int ot, int op, int kind, int lt, const void* l, int rt, const void* r, void* out, uint32_t n, int vec_ok) {
    switch (ot) {
    case AQG_INT8: return launch_ewise<C, int8_t>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_INT16: return launch_ewise<C, int16_t>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_INT32: return launch_ewise<C, int32_t>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_INT64: return launch_ewise<C, int64_t>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_UINT8: return launch_ewise<C, uint8_t>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_UINT16: return launch_ewise<C, uint16_t>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_UINT32: return launch_ewise<C, uint32_t>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_UINT64: return launch_ewise<C, uint64_t>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_FLOAT: return launch_ewise<C, float>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_DOUBLE: return launch_ewise<C, double>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_BOOL: return launch_ewise<C, bool>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_INT128: case AQG_UINT128: return launch_ewise<C, aqg_i128>(ctx, op, kind, lt, l, rt, r, out, n, vec_ok);
    }
    return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_ewise: unsupported result dtype");
}


} // namespace aqgew
