// ewise.hip -- element-wise column arithmetic / compare (reference server/table.h:820-937 free
// operators, :954-973 aqop_*), sqrt / truncate (server/aggregations.h:34-69).
//
// ret[i] = l[i] OP r[i] is evaluated in the C++ usual-arithmetic-conversion type C of the two
// element types -- exactly what the reference's loop body computes -- and then converted to the
// caller's result dtype (Coercion / GetLongType / GetFPType / bool, or aqop's Ret).
// Kernels are instantiated per (C, OT); the input dtypes are wave-uniform runtime switches around
// 8-element chunk loaders, so every lane still issues 16-byte loads and stores.
// HBM-bound: algorithmic bytes per row = sizeof(TL) + sizeof(TR) + sizeof(OT).
#include "ewise_impl.hpp"

using namespace aqgew;
extern template int aqgew::dispatch_ot<int32_t>(aqg_ctx*, int, int, int, int, const void*, int, const void*, void*, uint32_t, int);
extern template int aqgew::dispatch_ot<uint32_t>(aqg_ctx*, int, int, int, int, const void*, int, const void*, void*, uint32_t, int);
extern template int aqgew::dispatch_ot<int64_t>(aqg_ctx*, int, int, int, int, const void*, int, const void*, void*, uint32_t, int);
extern template int aqgew::dispatch_ot<uint64_t>(aqg_ctx*, int, int, int, int, const void*, int, const void*, void*, uint32_t, int);
extern template int aqgew::dispatch_ot<float>(aqg_ctx*, int, int, int, int, const void*, int, const void*, void*, uint32_t, int);
extern template int aqgew::dispatch_ot<double>(aqg_ctx*, int, int, int, int, const void*, int, const void*, void*, uint32_t, int);

namespace {

// ---- unary -----------------------------------------------------------------------------------
// sqrt: `ret[i] = sqrt(v[i])` resolves to ::sqrt(double) for every T (aggregations.h:34-39)
template <class T> __global__ void __launch_bounds__(256) sqrt_kernel(const T* __restrict__ x, double* __restrict__ out, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = sqrt((double)x[i]);
}
// truncate (aggregations.h:57-69): round(v * 10^p) / 10^p in double, unless v >= max/10^p
template <class T> __global__ void __launch_bounds__(256) truncate_kernel(const T* __restrict__ x, T* __restrict__ out, uint32_t n,
                                                                           double multiplier, double max_truncate) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        T v = x[i];
        out[i] = (double)v < max_truncate ? (T)(round((double)v * multiplier) / multiplier) : v;
    }
}

} // namespace

extern "C" {

int aqg_ewise_out_dtype(int op, int lt, int rt) {
    int c = aqg_coercion(lt, rt);
    if (c == AQG_ERROR || c == AQG_STR) return AQG_ERROR;
    switch (op) {
    case AQG_OP_ADD: case AQG_OP_SUB: return c;                 // get_autoext_type  table.h:779-782
    case AQG_OP_MUL: return aqg_long_type(c);                   // get_long_type     :784-787
    case AQG_OP_DIV: return aqg_fp_type(c);                     // get_fp_type       :789-792
    case AQG_OP_MOD: case AQG_OP_AND: case AQG_OP_OR: case AQG_OP_XOR: return c;
    default: return AQG_BOOL;
    }
}

int aqg_ewise(aqg_ctx* ctx, int op, int kind, int lt, const void* l, int rt, const void* r, int ot, void* out, uint32_t n) {
    if (!ctx || op < 0 || op > AQG_OP_NE || kind < 0 || kind > 2) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_ewise: bad argument");
    if (!l || !r || (!out && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_ewise: null operand");
    AQG_CHECK_ROWS(ctx, n, "aqg_ewise");
    if (!(dt_is_num(lt) || lt == AQG_BOOL) || !(dt_is_num(rt) || rt == AQG_BOOL)) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_ewise: operand dtype");
    if (n == 0) return AQG_OK;
    int c = usual_conv(lt, rt);
    if ((c == AQG_FLOAT || c == AQG_DOUBLE) && (op == AQG_OP_MOD || op == AQG_OP_AND || op == AQG_OP_OR || op == AQG_OP_XOR))
        return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_ewise: bitwise/mod on floating operands");
    // chunk loads need 16-byte aligned bases (device allocations are; sub-views of a column may not be: those take the
    // element-at-a-time path of the same kernel)
    auto aligned = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const int vec_ok = (kind == AQG_SCALAR_VEC || aligned(l)) && (kind == AQG_VEC_SCALAR || aligned(r)) && aligned(out);
    switch (c) {
    case AQG_INT32: return dispatch_ot<int32_t>(ctx, ot, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_UINT32: return dispatch_ot<uint32_t>(ctx, ot, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_INT64: return dispatch_ot<int64_t>(ctx, ot, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_UINT64: return dispatch_ot<uint64_t>(ctx, ot, op, kind, lt, l, rt, r, out, n, vec_ok);
    case AQG_FLOAT: return dispatch_ot<float>(ctx, ot, op, kind, lt, l, rt, r, out, n, vec_ok);
    default: return dispatch_ot<double>(ctx, ot, op, kind, lt, l, rt, r, out, n, vec_ok);
    }
}

int aqg_unary(aqg_ctx* ctx, int op, int t, const void* x, uint32_t n, uint32_t param, int ot, void* out) {
    if (!ctx || (!x && n) || (!out && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_unary: bad argument");
    AQG_CHECK_ROWS(ctx, n, "aqg_unary");
    if (n == 0) return AQG_OK;
    unsigned grid = aqg_grid(ctx, n, 256, 4, 16);
    if (op == AQG_UN_SQRT) {
        if (ot != AQG_DOUBLE) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_unary: sqrt yields double");
        if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "unary: the column dtype is not numeric (128-bit results are not inputs)");
        return aqg_dispatch_num(t, [&](auto tt) -> int {
            using T = typename decltype(tt)::type;
            hipLaunchKernelGGL(sqrt_kernel<T>, dim3(grid), dim3(256), 0, ctx->stream, (const T*)x, (double*)out, n);
            return aqg_check_launch(ctx, "sqrt_kernel");
        });
    }
    if (op == AQG_UN_TRUNCATE) {
        if (ot != t || !dt_is_fp(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_unary: truncate is floating only");
        uint32_t prec = t == AQG_FLOAT ? 7 : 16;   // aq_fp_precision (server/types.h:464-475)
        if (prec <= param) {                        // :59-60 returns a copy
            AQG_HIP(ctx, hipMemcpyAsync(out, x, (size_t)n * aqg_dtype_size(t), hipMemcpyDeviceToDevice, ctx->stream));
            return AQG_OK;
        }
        double multiplier = pow(10, param);
        if (t == AQG_FLOAT) {
            hipLaunchKernelGGL(truncate_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)x, (float*)out, n,
                               multiplier, (double)3.40282347e+38f / multiplier);
        } else {
            hipLaunchKernelGGL(truncate_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)x, (double*)out, n,
                               multiplier, 1.7976931348623157e+308 / multiplier);
        }
        return aqg_check_launch(ctx, "truncate_kernel");
    }
    return aqg_fail(ctx, AQG_ERR_ARG, "aqg_unary: bad op");
}

} // extern "C"
