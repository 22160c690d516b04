/*
 * aq_oracle.c -- CPU restatement of the reference's column-batch hot path (plain C).
 * TEST INFRASTRUCTURE ONLY -- see aq_oracle.h.  Each function cites the reference
 * file:line it follows (paths relative to the reference tree root).
 *
 * Integer arithmetic that the reference leaves to signed-overflow UB is done
 * here in the unsigned type of the same width (the wrap every x86 build of the
 * reference shows).  Floating point is strict IEEE (build WITHOUT -ffast-math).
 */
#include "aq_oracle.h"
#include "../include/aqg.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef __int128 i128;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------- */
/* numeric dtypes of the reference (server/types.h:162-181)                  */
/*   X(suffix, ctype, tag, longtype, long_tag)                               */
#define AQO_INT_TYPES(X)                                       \
    X(i8, int8_t, AQG_INT8, i128, AQG_INT128)                  \
    X(i16, int16_t, AQG_INT16, i128, AQG_INT128)               \
    X(i32, int32_t, AQG_INT32, i128, AQG_INT128)               \
    X(i64, int64_t, AQG_INT64, i128, AQG_INT128)               \
    X(u8, uint8_t, AQG_UINT8, u128, AQG_UINT128)               \
    X(u16, uint16_t, AQG_UINT16, u128, AQG_UINT128)            \
    X(u32, uint32_t, AQG_UINT32, u128, AQG_UINT128)            \
    X(u64, uint64_t, AQG_UINT64, u128, AQG_UINT128)
#define AQO_FP_TYPES(X)                                        \
    X(f32, float, AQG_FLOAT, double, AQG_DOUBLE)               \
    X(f64, double, AQG_DOUBLE, double, AQG_DOUBLE)
#define AQO_NUM_TYPES(X) AQO_INT_TYPES(X) AQO_FP_TYPES(X)

static size_t dt_size(int dt) {
    switch (dt) {
    case AQG_INT8: case AQG_UINT8: case AQG_BOOL: case AQG_CHAR: return 1;
    case AQG_INT16: case AQG_UINT16: return 2;
    case AQG_INT32: case AQG_UINT32: case AQG_FLOAT: return 4;
    case AQG_INT64: case AQG_UINT64: case AQG_DOUBLE: return 8;
    case AQG_INT128: case AQG_UINT128: return 16;
    default: return 0;
    }
}
static int dt_is_fp(int dt) { return dt == AQG_FLOAT || dt == AQG_DOUBLE; }
static int dt_is_unsigned(int dt) {
    return dt == AQG_UINT8 || dt == AQG_UINT16 || dt == AQG_UINT32 || dt == AQG_UINT64 || dt == AQG_UINT128 || dt == AQG_BOOL;
}
static int int_tag(size_t sz, int uns) {
    switch (sz) {
    case 1: return uns ? AQG_UINT8 : AQG_INT8;
    case 2: return uns ? AQG_UINT16 : AQG_INT16;
    case 4: return uns ? AQG_UINT32 : AQG_INT32;
    case 8: return uns ? AQG_UINT64 : AQG_INT64;
    case 16: return uns ? AQG_UINT128 : AQG_INT128;
    }
    return AQG_ERROR;
}

/* types::GetLongType -- server/types.h:205-210: unsigned -> u128, fp -> double, else i128 */
int AQCHK(long_type)(int dt) {
    if (!dt_size(dt)) return AQG_ERROR;
    if (dt_is_fp(dt)) return AQG_DOUBLE;
    return dt_is_unsigned(dt) ? AQG_UINT128 : AQG_INT128;
}
/* types::GetFPType -- server/types.h:199-204: sizeof==4 -> float, everything else double */
int AQCHK(fp_type)(int dt) {
    if (!dt_size(dt)) return AQG_ERROR;
    return dt_size(dt) == 4 ? AQG_FLOAT : AQG_DOUBLE;
}
/* types::Coercion<T1,T2> -- server/types.h:269-275 */
int AQCHK(coercion)(int a, int b) {
    size_t sa = dt_size(a), sb = dt_size(b);
    if (!sa || !sb) return AQG_ERROR;
    /* t2: aqis_same (server/types.h:37-62): same signedness, fp-ness and size -> T1 */
    if (a == AQG_BOOL || b == AQG_BOOL) { if (a == b) return a; }
    else if (dt_is_unsigned(a) == dt_is_unsigned(b) && dt_is_fp(a) == dt_is_fp(b) && sa == sb) return a;
    /* reference quirk (:273): aqis_same<unsigned long, const char*> holds (both unsigned, 8 bytes), so
     * Coercion of uint64 with any different type is `const char*`; such operators do not compile there */
    if (a == AQG_UINT64 || b == AQG_UINT64) return AQG_STR;
    int t0;
    if (sa <= sb) {
        if (sa == sb) t0 = dt_is_fp(a) ? a : (dt_is_fp(b) ? b : (dt_is_unsigned(a) ? b : a));
        else t0 = b;
    } else t0 = a;
    if (dt_is_fp(a) || dt_is_fp(b)) return AQCHK(fp_type)(t0);
    if (!(dt_is_unsigned(a) && dt_is_unsigned(b))) return int_tag(dt_size(t0), 0); /* GetSignedType */
    return t0;
}

/* ------------------------------------------------------------------------- */
/* generic scalar: a value in one of the six C arithmetic "compute classes"  */
typedef enum { C_I32, C_U32, C_I64, C_U64, C_F32, C_F64, C_I128, C_U128 } cclass;
typedef struct { cclass c; union { int32_t i32; uint32_t u32; int64_t i64; uint64_t u64; float f32; double f64; i128 i128v; u128 u128v; } v; } val;

/* integer promotion + usual arithmetic conversions of C/C++ for (lt, rt) */
static cclass promote1(int dt) {
    switch (dt) {
    case AQG_FLOAT: return C_F32;
    case AQG_DOUBLE: return C_F64;
    case AQG_INT64: return C_I64;
    case AQG_UINT64: return C_U64;
    case AQG_UINT32: return C_U32;
    case AQG_INT128: return C_I128;
    case AQG_UINT128: return C_U128;
    default: return C_I32; /* bool, (u)int8, (u)int16, int32 -> int */
    }
}
static cclass usual_conv(int lt, int rt) {
    cclass a = promote1(lt), b = promote1(rt);
    if (a == C_F64 || b == C_F64) return C_F64;
    if (a == C_F32 || b == C_F32) return C_F32;
    if (a == C_U128 || b == C_U128) return C_U128;
    if (a == C_I128 || b == C_I128) return C_I128;
    if (a == C_U64 || b == C_U64) return C_U64;
    if (a == C_I64 || b == C_I64) return C_I64;
    if (a == C_U32 || b == C_U32) return C_U32;
    return C_I32;
}
static val load_as(int dt, const void* p, size_t i, cclass c) {
    val r; r.c = c;
    /* read as the widest faithful representation first */
    i128 iv = 0; u128 uv = 0; double dv = 0; int kind = 0; /* 0 signed, 1 unsigned, 2 fp */
    switch (dt) {
    case AQG_INT8: iv = ((const int8_t*)p)[i]; break;
    case AQG_INT16: iv = ((const int16_t*)p)[i]; break;
    case AQG_INT32: iv = ((const int32_t*)p)[i]; break;
    case AQG_INT64: iv = ((const int64_t*)p)[i]; break;
    case AQG_INT128: iv = ((const i128*)p)[i]; break;
    case AQG_BOOL: case AQG_UINT8: uv = ((const uint8_t*)p)[i]; kind = 1; break;
    case AQG_UINT16: uv = ((const uint16_t*)p)[i]; kind = 1; break;
    case AQG_UINT32: uv = ((const uint32_t*)p)[i]; kind = 1; break;
    case AQG_UINT64: uv = ((const uint64_t*)p)[i]; kind = 1; break;
    case AQG_UINT128: uv = ((const u128*)p)[i]; kind = 1; break;
    case AQG_FLOAT: dv = ((const float*)p)[i]; kind = 2; break;
    case AQG_DOUBLE: dv = ((const double*)p)[i]; kind = 2; break;
    }
#define CONV(field, T) r.v.field = kind == 0 ? (T)iv : kind == 1 ? (T)uv : (T)dv
    switch (c) {
    case C_I32: CONV(i32, int32_t); break;
    case C_U32: CONV(u32, uint32_t); break;
    case C_I64: CONV(i64, int64_t); break;
    case C_U64: CONV(u64, uint64_t); break;
    case C_F32: r.v.f32 = kind == 0 ? (float)iv : kind == 1 ? (float)uv : (float)dv; break;
    case C_F64: CONV(f64, double); break;
    case C_I128: CONV(i128v, i128); break;
    case C_U128: CONV(u128v, u128); break;
    }
#undef CONV
    return r;
}
static void store_val(int ot, void* out, size_t i, val x) {
#define ST(T)                                                                            \
    switch (x.c) {                                                                       \
    case C_I32: ((T*)out)[i] = (T)x.v.i32; break;                                        \
    case C_U32: ((T*)out)[i] = (T)x.v.u32; break;                                        \
    case C_I64: ((T*)out)[i] = (T)x.v.i64; break;                                        \
    case C_U64: ((T*)out)[i] = (T)x.v.u64; break;                                        \
    case C_F32: ((T*)out)[i] = (T)x.v.f32; break;                                        \
    case C_F64: ((T*)out)[i] = (T)x.v.f64; break;                                        \
    case C_I128: ((T*)out)[i] = (T)x.v.i128v; break;                                     \
    case C_U128: ((T*)out)[i] = (T)x.v.u128v; break;                                     \
    }
    switch (ot) {
    case AQG_INT8: ST(int8_t) break;
    case AQG_INT16: ST(int16_t) break;
    case AQG_INT32: ST(int32_t) break;
    case AQG_INT64: ST(int64_t) break;
    case AQG_INT128: ST(i128) break;
    case AQG_UINT8: ST(uint8_t) break;
    case AQG_UINT16: ST(uint16_t) break;
    case AQG_UINT32: ST(uint32_t) break;
    case AQG_UINT64: ST(uint64_t) break;
    case AQG_UINT128: ST(u128) break;
    case AQG_FLOAT: ST(float) break;
    case AQG_DOUBLE: ST(double) break;
    case AQG_BOOL: {
        int b = 0;
        switch (x.c) {
        case C_I32: b = x.v.i32 != 0; break; case C_U32: b = x.v.u32 != 0; break;
        case C_I64: b = x.v.i64 != 0; break; case C_U64: b = x.v.u64 != 0; break;
        case C_F32: b = x.v.f32 != 0; break; case C_F64: b = x.v.f64 != 0; break;
        case C_I128: b = x.v.i128v != 0; break; case C_U128: b = x.v.u128v != 0; break;
        }
        ((uint8_t*)out)[i] = (uint8_t)b;
    } break;
    }
#undef ST
}

/* one application of `l OP r` in compute class c.  Integer + - * wrap (unsigned
 * arithmetic); integer / and % by zero trap in the reference (SIGFPE) -> 0 here. */
static val apply_op(int op, val l, val r, int* is_bool) {
    val o; o.c = l.c; *is_bool = 0;
#define ARITH_INT(field, T, UT)                                                              \
    switch (op) {                                                                            \
    case AQG_OP_ADD: o.v.field = (T)((UT)l.v.field + (UT)r.v.field); break;                  \
    case AQG_OP_SUB: o.v.field = (T)((UT)l.v.field - (UT)r.v.field); break;                  \
    case AQG_OP_MUL: o.v.field = (T)((UT)l.v.field * (UT)r.v.field); break;                  \
    case AQG_OP_DIV: o.v.field = r.v.field == 0 ? 0 :                                        \
        ((T)-1 < 0 && r.v.field == (T)-1 ? (T)((UT)0 - (UT)l.v.field) : (T)(l.v.field / r.v.field)); break; \
    case AQG_OP_MOD: o.v.field = r.v.field == 0 ? 0 :                                        \
        ((T)-1 < 0 && r.v.field == (T)-1 ? 0 : (T)(l.v.field % r.v.field)); break;           \
    case AQG_OP_AND: o.v.field = l.v.field & r.v.field; break;                               \
    case AQG_OP_OR: o.v.field = l.v.field | r.v.field; break;                                \
    case AQG_OP_XOR: o.v.field = l.v.field ^ r.v.field; break;                               \
    case AQG_OP_GT: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field > r.v.field; break;       \
    case AQG_OP_LT: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field < r.v.field; break;       \
    case AQG_OP_GE: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field >= r.v.field; break;      \
    case AQG_OP_LE: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field <= r.v.field; break;      \
    case AQG_OP_EQ: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field == r.v.field; break;      \
    case AQG_OP_NE: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field != r.v.field; break;      \
    }
#define ARITH_FP(field)                                                                      \
    switch (op) {                                                                            \
    case AQG_OP_ADD: o.v.field = l.v.field + r.v.field; break;                               \
    case AQG_OP_SUB: o.v.field = l.v.field - r.v.field; break;                               \
    case AQG_OP_MUL: o.v.field = l.v.field * r.v.field; break;                               \
    case AQG_OP_DIV: o.v.field = l.v.field / r.v.field; break;                               \
    case AQG_OP_GT: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field > r.v.field; break;       \
    case AQG_OP_LT: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field < r.v.field; break;       \
    case AQG_OP_GE: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field >= r.v.field; break;      \
    case AQG_OP_LE: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field <= r.v.field; break;      \
    case AQG_OP_EQ: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field == r.v.field; break;      \
    case AQG_OP_NE: *is_bool = 1; o.c = C_I32; o.v.i32 = l.v.field != r.v.field; break;      \
    default: o.v.field = 0; break;                                                           \
    }
    switch (l.c) {
    case C_I32: ARITH_INT(i32, int32_t, uint32_t) break;
    case C_U32: ARITH_INT(u32, uint32_t, uint32_t) break;
    case C_I64: ARITH_INT(i64, int64_t, uint64_t) break;
    case C_U64: ARITH_INT(u64, uint64_t, uint64_t) break;
    case C_I128: ARITH_INT(i128v, i128, u128) break;
    case C_U128: ARITH_INT(u128v, u128, u128) break;
    case C_F32: ARITH_FP(f32) break;
    case C_F64: ARITH_FP(f64) break;
    }
    return o;
}

/* result dtype of the reference's FREE operators (server/table.h:779-818,820-937):
 * + -  -> Coercion ; *  -> GetLongType<Coercion> ; /  -> GetFPType<Coercion> ; > -> bool */
int AQCHK(ewise_out_dtype)(int op, int lt, int rt) {
    int c = AQCHK(coercion)(lt, rt);
    if (c == AQG_ERROR || c == AQG_STR) return AQG_ERROR;
    switch (op) {
    case AQG_OP_ADD: case AQG_OP_SUB: return c;
    case AQG_OP_MUL: return AQCHK(long_type)(c);
    case AQG_OP_DIV: return AQCHK(fp_type)(c);
    case AQG_OP_MOD: case AQG_OP_AND: case AQG_OP_OR: case AQG_OP_XOR: return c;
    default: return AQG_BOOL;
    }
}

/* server/table.h:820-937 (`ret[i] = lhs[i] OP rhs[i]`, scalar forms `lhs[i] OP rhs`)
 * and aqop_* :954-959.  The expression is evaluated in the C++ type of
 * (T1 OP T2) and then converted to the element type of `ret`.                   */
int AQCHK(ewise)(int op, int kind, int lt, const void* l, int rt, const void* r, int ot, void* out, uint32_t n) {
    if (!dt_size(lt) || !dt_size(rt) || !dt_size(ot)) return AQG_ERR_DTYPE;
    cclass c = usual_conv(lt, rt);
    if ((c == C_F32 || c == C_F64) && (op == AQG_OP_MOD || op == AQG_OP_AND || op == AQG_OP_OR || op == AQG_OP_XOR)) return AQG_ERR_DTYPE;
    for (uint32_t i = 0; i < n; ++i) {
        val a = load_as(lt, l, kind == AQG_SCALAR_VEC ? 0 : i, c);
        val b = load_as(rt, r, kind == AQG_VEC_SCALAR ? 0 : i, c);
        int isb;
        val o = apply_op(op, a, b, &isb);
        store_val(ot, out, i, o);
    }
    return AQG_OK;
}

/* sqrt: server/aggregations.h:34-46 (ret double); truncate: :57-69 */
static uint32_t fp_precision(int t) { /* aq_fp_precision, server/types.h:464-475: float 7, double 16 */
    if (t == AQG_FLOAT) { uint32_t r = 0; float v = FLT_EPSILON; while (v + FLT_EPSILON < 1) { v *= 10; r++; } return r; }
    if (t == AQG_DOUBLE) { uint32_t r = 0; double v = DBL_EPSILON; while (v + DBL_EPSILON < 1) { v *= 10; r++; } return r; }
    return 0;
}
int AQCHK(unary)(int op, int t, const void* x, uint32_t n, uint32_t param, int ot, void* out) {
    if (!dt_size(t)) return AQG_ERR_DTYPE;
    if (op == AQG_UN_SQRT) {
        if (ot != AQG_DOUBLE) return AQG_ERR_DTYPE;
        for (uint32_t i = 0; i < n; ++i) {
            /* unqualified `sqrt(v[i])` resolves to ::sqrt(double) for every T (float included) */
            val a = load_as(t, x, i, C_F64);
            ((double*)out)[i] = sqrt(a.v.f64);
        }
        return AQG_OK;
    }
    if (op == AQG_UN_TRUNCATE) {
        if (ot != t) return AQG_ERR_DTYPE;
        if (fp_precision(t) <= param) { memcpy(out, x, (size_t)n * dt_size(t)); return AQG_OK; } /* :59-60 */
        double multiplier = pow(10, param);                                                        /* :61 */
        if (t == AQG_FLOAT) {
            double max_truncate = FLT_MAX / multiplier;
            for (uint32_t i = 0; i < n; ++i) {
                float v = ((const float*)x)[i];
                ((float*)out)[i] = v < max_truncate ? (float)(round(v * multiplier) / multiplier) : v;
            }
        } else if (t == AQG_DOUBLE) {
            double max_truncate = DBL_MAX / multiplier;
            for (uint32_t i = 0; i < n; ++i) {
                double v = ((const double*)x)[i];
                ((double*)out)[i] = v < max_truncate ? round(v * multiplier) / multiplier : v;
            }
        } else return AQG_ERR_DTYPE;
        return AQG_OK;
    }
    return AQG_ERR_ARG;
}

/* ------------------------------------------------------------------------- */
/* reductions                                                                */
int AQCHK(reduce_out_dtype)(int op, int t) {
    if (!dt_size(t)) return AQG_ERROR;
    switch (op) {
    case AQG_RED_SUM: return AQCHK(long_type)(t);
    case AQG_RED_MIN: case AQG_RED_MAX: case AQG_RED_FIRST: case AQG_RED_LAST: return t;
    case AQG_RED_COUNT: return AQG_UINT64;
    case AQG_RED_AVG: case AQG_RED_VAR: case AQG_RED_STDDEV: return AQG_DOUBLE;
    }
    return AQG_ERROR;
}

#define TMIN_i8 INT8_MIN
#define TMAX_i8 INT8_MAX
#define TMIN_i16 INT16_MIN
#define TMAX_i16 INT16_MAX
#define TMIN_i32 INT32_MIN
#define TMAX_i32 INT32_MAX
#define TMIN_i64 INT64_MIN
#define TMAX_i64 INT64_MAX
#define TMIN_u8 0
#define TMAX_u8 UINT8_MAX
#define TMIN_u16 0
#define TMAX_u16 UINT16_MAX
#define TMIN_u32 0
#define TMAX_u32 UINT32_MAX
#define TMIN_u64 0
#define TMAX_u64 UINT64_MAX
/* numeric_limits<fp>::min() is the smallest POSITIVE normal (defect D8 of the survey) */
#define TMIN_f32 FLT_MIN
#define TMAX_f32 FLT_MAX
#define TMIN_f64 DBL_MIN
#define TMAX_f64 DBL_MAX

#define DEF_REDUCE(S, T, TAG, LT, LTAG)                                                            \
    /* sum: server/aggregations.h:19-27 */                                                         \
    static LT sum_##S(const T* v, uint32_t n) { LT ret = 0; for (uint32_t i = 0; i < n; ++i) ret += v[i]; return ret; } \
    /* max: :71-78 (seed numeric_limits<T>::min()), min: :79-86 */                                 \
    static T max_##S(const T* v, uint32_t n) { T m = TMIN_##S; for (uint32_t i = 0; i < n; ++i) m = m > v[i] ? m : v[i]; return m; } \
    static T min_##S(const T* v, uint32_t n) { T m = TMAX_##S; for (uint32_t i = 0; i < n; ++i) m = m < v[i] ? m : v[i]; return m; } \
    /* avg: :28-32  sum / static_cast<double>(size) */                                             \
    static double avg_##S(const T* v, uint32_t n) { return sum_##S(v, n) / (double)n; }            \
    /* var: :332-348  (ssq - s*s/(FP)(len+1)) / (FP)(len+1); products in the promoted type of T */ \
    static double var_##S(const T* a, uint32_t n) {                                                \
        LT s = 0, ssq = 0;                                                                         \
        if (n) { s = a[0]; ssq = a[0] * a[0]; }                                                    \
        for (uint32_t i = 1; i < n; ++i) { s += a[i]; ssq += a[i] * a[i]; }                        \
        return (ssq - s * s / (double)(uint32_t)(n + 1)) / (double)(uint32_t)(n + 1);              \
    }                                                                                              \
    static void reduce_##S(int op, const T* v, uint32_t n, void* out) {                            \
        memset(out, 0, 16);                                                                        \
        switch (op) {                                                                              \
        case AQG_RED_SUM: { LT s = sum_##S(v, n); memcpy(out, &s, sizeof s); } break;              \
        case AQG_RED_MIN: { T m = min_##S(v, n); memcpy(out, &m, sizeof m); } break;               \
        case AQG_RED_MAX: { T m = max_##S(v, n); memcpy(out, &m, sizeof m); } break;               \
        case AQG_RED_COUNT: { uint64_t c = n; memcpy(out, &c, 8); } break;           /* :10-13 */  \
        case AQG_RED_AVG: { double d = avg_##S(v, n); memcpy(out, &d, 8); } break;                 \
        case AQG_RED_VAR: { double d = var_##S(v, n); memcpy(out, &d, 8); } break;                 \
        case AQG_RED_STDDEV: { double d = sqrt(var_##S(v, n)); memcpy(out, &d, 8); } break; /* :413-416 */ \
        case AQG_RED_FIRST: { T m = n ? v[0] : 0; memcpy(out, &m, sizeof m); } break;  /* :493-497 */ \
        case AQG_RED_LAST: { T m = n ? v[n - 1] : 0; memcpy(out, &m, sizeof m); } break; /* :487-491 */ \
        }                                                                                          \
    }
/* signed narrow ints multiply as int (may wrap): do the product in the promoted type explicitly */
AQO_NUM_TYPES(DEF_REDUCE)

int AQCHK(reduce)(int op, int t, const void* x, uint32_t n, void* out16) {
    if (op < 0 || op > AQG_RED_LAST) return AQG_ERR_ARG;
    switch (t) {
#define CASE(S, T, TAG, LT, LTAG) case TAG: reduce_##S(op, (const T*)x, n, out16); return AQG_OK;
        AQO_NUM_TYPES(CASE)
#undef CASE
    }
    return AQG_ERR_DTYPE;
}

/* corr: server/aggregations.h:383-407.  `InnerType` there is the Coercion STRUCT (not its ::type), so
 * GetLongType<InnerType> is __int128 for EVERY input type: the five sums are __int128 and a floating
 * term is accumulated as  s = (__int128)((fp)s + term)  -- truncated every step (reference behaviour).
 * Products are evaluated in the C++ type of their operands.                                          */
static void corr_acc(i128* s, val v) {
    switch (v.c) {
    case C_I32: *s += v.v.i32; break;
    case C_U32: *s += v.v.u32; break;
    case C_I64: *s += v.v.i64; break;
    case C_U64: *s += v.v.u64; break;
    case C_I128: *s += v.v.i128v; break;
    case C_U128: *s = (i128)((u128)*s + v.v.u128v); break;
    case C_F32: *s = (i128)((float)*s + v.v.f32); break;
    case C_F64: *s = (i128)((double)*s + v.v.f64); break;
    }
}
int AQCHK(corr)(int tx, const void* x, int ty, const void* y, uint32_t n, double* out) {
    int inner = AQCHK(coercion)(tx, ty);
    if (inner == AQG_ERROR || inner == AQG_STR) return AQG_ERR_DTYPE;
    cclass cx = promote1(tx), cy = promote1(ty), cxx = usual_conv(tx, tx), cyy = usual_conv(ty, ty), cxy = usual_conv(tx, ty);
    int isb;
    i128 sx = 0, sy = 0, sxy = 0, sx2 = 0, sy2 = 0;
    for (uint32_t i = 0; i < n; ++i) {
        corr_acc(&sx, load_as(tx, x, i, cx));
        corr_acc(&sx2, apply_op(AQG_OP_MUL, load_as(tx, x, i, cxx), load_as(tx, x, i, cxx), &isb));
        corr_acc(&sy, load_as(ty, y, i, cy));
        corr_acc(&sxy, apply_op(AQG_OP_MUL, load_as(tx, x, i, cxy), load_as(ty, y, i, cxy), &isb));
        corr_acc(&sy2, apply_op(AQG_OP_MUL, load_as(ty, y, i, cyy), load_as(ty, y, i, cyy), &isb));
    }
    *out = (n * sxy - (double)(sx * sy)) / sqrt((n * sx2 - (double)(sx * sx)) * (n * sy2 - (double)(sy * sy)));
    return AQG_OK;
}

/* ------------------------------------------------------------------------- */
/* scans, windows, shifts                                                    */
int AQCHK(scan_out_dtype)(int op, int t) {
    if (!dt_size(t)) return AQG_ERROR;
    switch (op) {
    case AQG_SCAN_SUMS: case AQG_SCAN_SUMW: return AQCHK(long_type)(t);          /* :213,252 */
    case AQG_SCAN_AVGS: case AQG_SCAN_AVGW: case AQG_SCAN_VARS: case AQG_SCAN_STDDEVS:
    case AQG_SCAN_VARW: case AQG_SCAN_STDDEVW: return AQG_DOUBLE;                /* GetFPType<GetLongType<T>> */
    case AQG_SCAN_MINS: case AQG_SCAN_MAXS: case AQG_SCAN_MINW: case AQG_SCAN_MAXW:
    case AQG_SCAN_DELTAS: case AQG_SCAN_PREV: case AQG_SCAN_NEXT: return t;
    case AQG_SCAN_RATIOW: return AQCHK(fp_type)(t);                              /* :186 */
    }
    return AQG_ERROR;
}

/* monotonic deque of (value, index) pairs for minw/maxw (std::deque in the reference) */
#define DEF_SCAN(S, T, TAG, LT, LTAG)                                                              \
    /* mins :89-99, maxs :108-118 */                                                               \
    static void mins_##S(const T* a, uint32_t n, T* r) { T m = TMAX_##S; for (uint32_t i = 0; i < n; ++i) { if (a[i] < m) m = a[i]; r[i] = m; } } \
    static void maxs_##S(const T* a, uint32_t n, T* r) { T m = TMIN_##S; for (uint32_t i = 0; i < n; ++i) { if (a[i] > m) m = a[i]; r[i] = m; } } \
    /* minw :127-139 / maxw :148-160: pop front when its index == i - w, pop back while worse */    \
    static void minmaxw_##S(int is_max, uint32_t w, const T* a, uint32_t n, T* r) {                \
        uint32_t* q = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));                           \
        uint32_t head = 0, tail = 0;                                                               \
        for (uint32_t i = 0; i < n; ++i) {                                                         \
            if (head != tail && q[head] == (uint32_t)(i - w)) ++head;                              \
            if (is_max) while (head != tail && a[q[tail - 1]] < a[i]) --tail;                      \
            else        while (head != tail && a[q[tail - 1]] > a[i]) --tail;                      \
            q[tail++] = i;                                                                         \
            r[i] = a[q[head]];                                                                     \
        }                                                                                          \
        free(q);                                                                                   \
    }                                                                                              \
    /* sums :203-210 */                                                                            \
    static void sums_##S(const T* a, uint32_t n, LT* r) { if (n) r[0] = a[0]; for (uint32_t i = 1; i < n; ++i) r[i] = r[i - 1] + a[i]; } \
    /* avgs :219-228: s = ret[0] = arr[0]; ret[i] = (s += arr[i]) / (double)(i+1) */                \
    static void avgs_##S(const T* a, uint32_t n, double* r) {                                      \
        LT s = 0; if (n) s = (LT)(r[0] = (double)a[0]);                                            \
        for (uint32_t i = 1; i < n; ++i) r[i] = (s += a[i]) / (double)(i + 1);                     \
    }                                                                                              \
    /* sumw :238-249 */                                                                            \
    static void sumw_##S(uint32_t w, const T* a, uint32_t n, LT* r) {                              \
        w = w > n ? n : w; if (n) r[0] = a[0];                                                     \
        for (uint32_t i = 1; i < w; ++i) r[i] = r[i - 1] + a[i];                                   \
        for (uint32_t i = w; i < n; ++i) r[i] = r[i - 1] + a[i] - a[i - w];                        \
    }                                                                                              \
    /* avgw :258-272: running mean for i<w, then ret[i-1] + (arr[i]-arr[i-w])/(double)w */          \
    static void avgw_##S(uint32_t w, const T* a, uint32_t n, double* r) {                          \
        LT s = 0; w = w > n ? n : w; if (n) s = (LT)(r[0] = (double)a[0]);                         \
        for (uint32_t i = 1; i < w; ++i) r[i] = (s += a[i]) / (double)(i + 1);                     \
        for (uint32_t i = w; i < n; ++i) r[i] = r[i - 1] + (a[i] - a[i - w]) / (double)w;          \
    }                                                                                              \
    /* deltas :439-446, prev :455-462, aggnext :471-478 */                                         \
    static void deltas_##S(const T* a, uint32_t n, T* r) { if (n) r[0] = 0; for (uint32_t i = 1; i < n; ++i) r[i] = (T)(a[i] - a[i - 1]); } \
    static void prev_##S(const T* a, uint32_t n, T* r) { if (n) r[0] = a[0]; for (uint32_t i = 1; i < n; ++i) r[i] = a[i - 1]; } \
    static void next_##S(const T* a, uint32_t n, T* r) { for (uint32_t i = 1; i < n; ++i) r[i - 1] = a[i]; if (n > 0) r[n - 1] = a[n - 1]; } \
    /* vars :350-373 (running variance; the stray printf at :368 is not restated) */               \
    static void vars_##S(int sd, const T* a, uint32_t n, double* r) {                              \
        LT s = 0; double MnX = 0, EnX = 0;                                                         \
        if (n) { s = a[0]; MnX = 0; EnX = a[0]; r[0] = 0; }                                        \
        for (uint32_t i = 1; i < n; ++i) {                                                         \
            s += a[i];                                                                             \
            double _EnX = s / (double)(i + 1);                                                     \
            MnX += (a[i] - EnX) * (a[i] - _EnX);                                                   \
            EnX = _EnX;                                                                            \
            r[i] = MnX / (double)(i + 1);                                                          \
            if (sd) r[i] = sqrt(r[i]);                                                             \
        }                                                                                          \
    }                                                                                              \
    /* varw: the reference (:283-321) reads arr[i-w-1] (out of bounds at i==w) -- undefined, so    \
     * this restates the INTENDED population variance of the last w values (parity unpinned). */   \
    static void varw_##S(int sd, uint32_t w, const T* a, uint32_t n, double* r) {                  \
        w = w > n ? n : w;                                                                         \
        for (uint32_t i = 0; i < n; ++i) {                                                         \
            uint32_t lo = (w && i + 1 > w) ? i + 1 - w : 0, cnt = i + 1 - lo;                      \
            long double m = 0, q = 0;                                                              \
            for (uint32_t j = lo; j <= i; ++j) m += (long double)a[j];                             \
            m /= cnt;                                                                              \
            for (uint32_t j = lo; j <= i; ++j) { long double d = (long double)a[j] - m; q += d * d; } \
            double v = (double)(q / cnt);                                                          \
            r[i] = sd ? sqrt(v) : v;                                                               \
        }                                                                                          \
    }
AQO_NUM_TYPES(DEF_SCAN)

/* ratiow :169-183.  FPType = GetFPType<T>; `arr[i] / (FPType)arr[i-w]` divides in the
 * C++ type of (T / FPType) and stores as FPType.                                              */
#define DEF_RATIOW(S, T, TAG, FPT)                                                                 \
    static void ratiow_##S(uint32_t w, const T* a, uint32_t n, FPT* r) {                           \
        if (!n) return; /* the reference writes ret[0] on an empty vector (heap overrun): skipped */ \
        uint32_t len = n; if (n <= w) len = 1;                                                     \
        w = w > len ? len : w;                                                                     \
        r[0] = 0;                                                                                  \
        for (uint32_t i = 0; i < w; ++i) r[i] = (FPT)(a[i] / (FPT)a[0]);                           \
        for (uint32_t i = w; i < n; ++i) r[i] = (FPT)(a[i] / (FPT)a[i - w]);                       \
    }
DEF_RATIOW(i8, int8_t, AQG_INT8, double)
DEF_RATIOW(i16, int16_t, AQG_INT16, double)
DEF_RATIOW(i32, int32_t, AQG_INT32, float)
DEF_RATIOW(i64, int64_t, AQG_INT64, double)
DEF_RATIOW(u8, uint8_t, AQG_UINT8, double)
DEF_RATIOW(u16, uint16_t, AQG_UINT16, double)
DEF_RATIOW(u32, uint32_t, AQG_UINT32, float)
DEF_RATIOW(u64, uint64_t, AQG_UINT64, double)
DEF_RATIOW(f32, float, AQG_FLOAT, float)
DEF_RATIOW(f64, double, AQG_DOUBLE, double)

int AQCHK(scan)(int op, int t, const void* x, uint32_t n, uint32_t w, void* out) {
    /* w == 0: sumw/avgw read ret[-1] and arr[i-0] (aggregations.h:247-248,270-271) -- undefined in the
     * reference, rejected here and in the C-ABI */
    if (w == 0 && (op == AQG_SCAN_SUMW || op == AQG_SCAN_AVGW || op == AQG_SCAN_VARW || op == AQG_SCAN_STDDEVW)) return AQG_ERR_ARG;
    switch (t) {
#define CASE(S, T, TAG, LT, LTAG)                                                          \
    case TAG:                                                                              \
        switch (op) {                                                                      \
        case AQG_SCAN_SUMS: sums_##S((const T*)x, n, (LT*)out); return AQG_OK;             \
        case AQG_SCAN_AVGS: avgs_##S((const T*)x, n, (double*)out); return AQG_OK;         \
        case AQG_SCAN_MINS: mins_##S((const T*)x, n, (T*)out); return AQG_OK;              \
        case AQG_SCAN_MAXS: maxs_##S((const T*)x, n, (T*)out); return AQG_OK;              \
        case AQG_SCAN_SUMW: sumw_##S(w, (const T*)x, n, (LT*)out); return AQG_OK;          \
        case AQG_SCAN_AVGW: avgw_##S(w, (const T*)x, n, (double*)out); return AQG_OK;      \
        case AQG_SCAN_MINW: minmaxw_##S(0, w, (const T*)x, n, (T*)out); return AQG_OK;     \
        case AQG_SCAN_MAXW: minmaxw_##S(1, w, (const T*)x, n, (T*)out); return AQG_OK;     \
        case AQG_SCAN_RATIOW: ratiow_##S(w, (const T*)x, n, out); return AQG_OK;           \
        case AQG_SCAN_DELTAS: deltas_##S((const T*)x, n, (T*)out); return AQG_OK;          \
        case AQG_SCAN_PREV: prev_##S((const T*)x, n, (T*)out); return AQG_OK;              \
        case AQG_SCAN_NEXT: next_##S((const T*)x, n, (T*)out); return AQG_OK;              \
        case AQG_SCAN_VARS: vars_##S(0, (const T*)x, n, (double*)out); return AQG_OK;      \
        case AQG_SCAN_STDDEVS: vars_##S(1, (const T*)x, n, (double*)out); return AQG_OK;   \
        case AQG_SCAN_VARW: varw_##S(0, w, (const T*)x, n, (double*)out); return AQG_OK;   \
        case AQG_SCAN_STDDEVW: varw_##S(1, w, (const T*)x, n, (double*)out); return AQG_OK; \
        }                                                                                  \
        return AQG_ERR_ARG;
        AQO_NUM_TYPES(CASE)
#undef CASE
    }
    return AQG_ERR_DTYPE;
}

/* ------------------------------------------------------------------------- */
/* gather: server/table.h:184-189 ; mask filter: :190-198 (selected values only) */
int AQCHK(gather)(int t, const void* x, const uint32_t* idx, uint32_t m, void* out) {
    size_t sz = dt_size(t);
    if (!sz) return AQG_ERR_DTYPE;
    for (uint32_t i = 0; i < m; ++i) memcpy((char*)out + (size_t)i * sz, (const char*)x + (size_t)idx[i] * sz, sz);
    return AQG_OK;
}
int AQCHK(compact)(int t, const void* x, const uint8_t* mask, uint32_t n, void* out, uint32_t* m) {
    size_t sz = dt_size(t);
    if (!sz) return AQG_ERR_DTYPE;
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (mask[i]) { memcpy((char*)out + (size_t)k * sz, (const char*)x + (size_t)i * sz, sz); ++k; }
    *m = k;
    return AQG_OK;
}

/* ------------------------------------------------------------------------- */
/* hashing: wyhash mix (server/unordered_dense.h:112-145,212-214)             */
static inline uint64_t wy_mix(uint64_t a, uint64_t b) {
    u128 r = (u128)a * b;
    return (uint64_t)r ^ (uint64_t)(r >> 64);
}
static inline uint64_t wy_hash_u64(uint64_t x) { return wy_mix(x, UINT64_C(0x9E3779B97F4A7C15)); }

/* hash<T> for integral T = wyhash::hash(static_cast<uint64_t>(obj)) (:279-310);
 * float/double fall to std::hash (libstdc++: 0 for +-0.0, else _Hash_bytes) -- not restated */
static int key_as_u64(int t, const void* col, size_t i, uint64_t* out) {
    switch (t) {
    case AQG_INT8: *out = (uint64_t)(int64_t)((const int8_t*)col)[i]; return 1;
    case AQG_INT16: *out = (uint64_t)(int64_t)((const int16_t*)col)[i]; return 1;
    case AQG_INT32: *out = (uint64_t)(int64_t)((const int32_t*)col)[i]; return 1;
    case AQG_INT64: *out = (uint64_t)((const int64_t*)col)[i]; return 1;
    case AQG_BOOL: case AQG_UINT8: *out = ((const uint8_t*)col)[i]; return 1;
    case AQG_UINT16: *out = ((const uint16_t*)col)[i]; return 1;
    case AQG_UINT32: *out = ((const uint32_t*)col)[i]; return 1;
    case AQG_UINT64: *out = ((const uint64_t*)col)[i]; return 1;
    }
    return 0;
}
uint64_t AQCHK(hash_scalar)(int t, const void* v) {
    uint64_t k;
    if (!key_as_u64(t, v, 0, &k)) return 0;
    return wy_hash_u64(k);
}
/* hasher<Ts...>: XOR of field hashes, seeded 534235245539 (server/hasher.h:66-89);
 * the single-field specialisation (:90-95) is the bare field hash.                */
static inline uint64_t tuple_hash_row(int nkeys, const int* dts, const void* const* cols, size_t i) {
    uint64_t k;
    if (nkeys == 1) { key_as_u64(dts[0], cols[0], i, &k); return wy_hash_u64(k); }
    uint64_t h = UINT64_C(534235245539);
    for (int j = 0; j < nkeys; ++j) { key_as_u64(dts[j], cols[j], i, &k); h ^= wy_hash_u64(k); }
    return h;
}
uint64_t AQCHK(hash_tuple)(int nkeys, const int* dts, const void* const* vals) {
    return tuple_hash_row(nkeys, dts, vals, 0);
}

/* ------------------------------------------------------------------------- */
/* AQHashTable: robin-hood set with a dense value vector.
 * server/unordered_dense.h: bucket {dist_and_fingerprint, value_idx} (:320-327),
 * dist_inc = 1<<8, fingerprint = low byte; max load 0.8 (:406); mixed_hash
 * re-mixes a non-avalanching hash (:465-478); hashtable_push (:1117-1147);
 * place_and_shift_up (:508-515); increase_size (:573-581).                      */
typedef struct { uint32_t daf; uint32_t vidx; } rh_bucket;
typedef struct {
    rh_bucket* b; uint64_t nb; uint64_t maxcap; uint8_t shifts;
    uint32_t* first_row; uint32_t nvals; /* dense values: the row that introduced each key */
    int nkeys; const int* dts; const void* const* cols;
} rh_table;

static uint8_t rh_shifts_for(uint64_t s) { /* calc_shifts_for_size :521-527 */
    uint8_t sh = 64 - 3;
    while (sh > 0 && (uint64_t)((float)((uint64_t)1 << (64 - sh)) * 0.8f) < s) --sh;
    return sh;
}
static void rh_alloc(rh_table* t) { /* allocate_buckets_from_shift :551-561 */
    t->nb = (uint64_t)1 << (64 - t->shifts);
    t->b = (rh_bucket*)calloc(t->nb, sizeof(rh_bucket));
    t->maxcap = (uint32_t)((float)t->nb * 0.8f);
}
static int rh_keys_equal(const rh_table* t, size_t i, size_t j) { /* std::tuple operator== */
    for (int k = 0; k < t->nkeys; ++k) {
        size_t sz = dt_size(t->dts[k]);
        if (memcmp((const char*)t->cols[k] + i * sz, (const char*)t->cols[k] + j * sz, sz)) return 0;
    }
    return 1;
}
static inline uint64_t rh_next(const rh_table* t, uint64_t i) { return i + 1 == t->nb ? 0 : i + 1; }
static void rh_place(rh_table* t, rh_bucket bk, uint64_t place) {
    while (t->b[place].daf != 0) {
        rh_bucket tmp = t->b[place]; t->b[place] = bk; bk = tmp;
        bk.daf += 1u << 8;
        place = rh_next(t, place);
    }
    t->b[place] = bk;
}
static void rh_grow(rh_table* t) {
    --t->shifts; free(t->b); rh_alloc(t);
    for (uint32_t v = 0; v < t->nvals; ++v) { /* clear_and_fill_buckets_from_values :563-571 */
        uint64_t h = wy_hash_u64(tuple_hash_row(t->nkeys, t->dts, t->cols, t->first_row[v]));
        uint32_t daf = (1u << 8) | (uint32_t)(h & 0xFF);
        uint64_t bi = h >> t->shifts;
        while (daf < t->b[bi].daf) { daf += 1u << 8; bi = rh_next(t, bi); }
        rh_bucket bk = { daf, v }; rh_place(t, bk, bi);
    }
}
static uint32_t rh_push(rh_table* t, uint32_t row) {
    if (t->nvals >= t->maxcap) rh_grow(t);
    uint64_t h = wy_hash_u64(tuple_hash_row(t->nkeys, t->dts, t->cols, row));
    uint32_t daf = (1u << 8) | (uint32_t)(h & 0xFF);
    uint64_t bi = h >> t->shifts;
    while (daf <= t->b[bi].daf) {
        if (daf == t->b[bi].daf && rh_keys_equal(t, row, t->first_row[t->b[bi].vidx])) return t->b[bi].vidx;
        daf += 1u << 8; bi = rh_next(t, bi);
    }
    uint32_t v = t->nvals++;
    t->first_row[v] = row;
    rh_bucket bk = { daf, v }; rh_place(t, bk, bi);
    return v;
}

int AQCHK(groupby)(int nkeys, const int* key_dts, const void* const* keys, uint32_t n,
                   uint32_t* reversemap, uint32_t* ngroups, uint32_t* counts,
                   uint32_t* offsets, uint32_t* row_ids, uint32_t* first_rows) {
    for (int k = 0; k < nkeys; ++k) { uint64_t d; if (n && !key_as_u64(key_dts[k], keys[k], 0, &d)) return AQG_ERR_DTYPE; }
    rh_table t; memset(&t, 0, sizeof t);
    t.nkeys = nkeys; t.dts = key_dts; t.cols = keys;
    t.first_row = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    t.shifts = rh_shifts_for(n);                 /* AQHashTable(sz): reserve(sz) hasher.h:151-158 */
    rh_alloc(&t);
    memset(counts, 0, sizeof(uint32_t) * n);     /* ht_base = calloc(sz) */
    for (uint32_t i = 0; i < n; ++i) {           /* hashtable_push(Key&&, i) hasher.h:176-179 */
        reversemap[i] = rh_push(&t, i);
        ++counts[reversemap[i]];
    }
    uint32_t G = t.nvals;
    *ngroups = G;
    if (first_rows) memcpy(first_rows, t.first_row, sizeof(uint32_t) * G);
    if (offsets && row_ids) {                    /* ht_postproc hasher.h:181-198 */
        uint32_t* hb = offsets;
        for (uint32_t g = 0; g < G; ++g) hb[g] = counts[g];
        for (uint32_t g = 1; g < G; ++g) hb[g] += hb[g - 1];
        for (uint32_t i = 0; i < n; ++i) row_ids[--hb[reversemap[i]]] = i;
    }
    free(t.b); free(t.first_row);
    return AQG_OK;
}

/* ------------------------------------------------------------------------- */
/* Group-by over key columns that are not plain integers: dense ids in first-occurrence order under the reference's tuple ==
 * (std::equal_to on std::tuple; element == as defined in server/types.h and server/libaquery.cpp), hashed by
 * server/hasher.h:97-144.  Only EQUALITY decides the ids (the table mechanics do not), so this restates equality per type:
 *   AQG_DATE      {uchar day, month; short year}: all 4 bytes (types.h:82-104, operator== compares the three fields)
 *   AQG_TIME      {uint ms; uchar seconds, minutes, hours}: the 7 field bytes of the 8-byte struct (types.h:106-128)
 *   AQG_TIMESTAMP {date_t; time_t}: 12-byte struct, date bytes 0-3, time fields at bytes 4-10 (types.h:129-146)
 *   AQG_INT128 / AQG_UINT128: all 16 bytes (hash: int128_struct low ^ high, hasher.h:135-141)
 *   AQG_FLOAT / AQG_DOUBLE: by value -- 0.0 == -0.0 (and libstdc++'s std::hash maps both to 0, so they meet in the table);
 *                 NaN != NaN although the hashes agree, so every NaN row becomes a group of its own
 *   AQG_STR       astring_view: the NUL-terminated contents (types.h:299-309; hash of the string view, hasher.h:99-106);
 *                 the column holds `const char*` pointers
 *   raw `const char*` keys are POINTERS under tuple == : pass them as AQG_UINT64
 * Pinned against the reference itself: tests/golden/ref_golden_keys.json (oracle/gen_golden.py, ref_harness.cpp).             */
static size_t typed_key_size(int t) {
    switch (t) {
    case AQG_DATE: return 4;
    case AQG_TIME: return 8;
    case AQG_TIMESTAMP: return 12;
    case AQG_INT128: case AQG_UINT128: return 16;
    case AQG_STR: return sizeof(char*);
    default: return dt_size(t);
    }
}
/* 1 if element i of column a equals element j (same column) */
static int typed_equal(int t, const void* col, size_t i, size_t j) {
    const unsigned char* b = (const unsigned char*)col;
    switch (t) {
    case AQG_DATE: return memcmp(b + 4 * i, b + 4 * j, 4) == 0;
    case AQG_TIME: return memcmp(b + 8 * i, b + 8 * j, 7) == 0;
    case AQG_TIMESTAMP: return memcmp(b + 12 * i, b + 12 * j, 11) == 0;
    case AQG_INT128: case AQG_UINT128: return memcmp(b + 16 * i, b + 16 * j, 16) == 0;
    case AQG_FLOAT: return ((const float*)col)[i] == ((const float*)col)[j];
    case AQG_DOUBLE: return ((const double*)col)[i] == ((const double*)col)[j];
    case AQG_STR: return strcmp(((const char* const*)col)[i], ((const char* const*)col)[j]) == 0;
    default: { size_t z = dt_size(t); return z && memcmp(b + z * i, b + z * j, z) == 0; }
    }
}
/* a bucket hash that is equal for equal elements (any such function yields the same ids) */
static uint64_t typed_bucket(int t, const void* col, size_t i) {
    const unsigned char* b = (const unsigned char*)col;
    uint64_t h = UINT64_C(1469598103934665603);
    size_t len = typed_key_size(t);
    const unsigned char* p = b + len * i;
    if (t == AQG_TIME) len = 7;
    if (t == AQG_TIMESTAMP) len = 11;
    if (t == AQG_FLOAT) { float f = ((const float*)col)[i]; if (f == 0.0f) return 7; }
    if (t == AQG_DOUBLE) { double d = ((const double*)col)[i]; if (d == 0.0) return 7; }
    if (t == AQG_STR) { p = (const unsigned char*)((const char* const*)col)[i]; len = strlen((const char*)p); }
    for (size_t k = 0; k < len; ++k) h = (h ^ p[k]) * UINT64_C(1099511628211);
    return h;
}
int AQCHK(groupby_typed)(int nkeys, const int* key_dts, const void* const* keys, uint32_t n,
                         uint32_t* reversemap, uint32_t* ngroups, uint32_t* first_rows) {
    if (nkeys < 1 || nkeys > 8) return AQG_ERR_ARG;
    for (int k = 0; k < nkeys; ++k) if (!typed_key_size(key_dts[k])) return AQG_ERR_DTYPE;
    uint64_t nb = 16; while (nb < (uint64_t)n * 2) nb <<= 1;
    uint32_t* head = (uint32_t*)malloc(sizeof(uint32_t) * nb);      /* chained: bucket -> first group, groups chained by next */
    uint32_t* next = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    memset(head, 0xFF, sizeof(uint32_t) * nb);
    uint32_t G = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t h = UINT64_C(534235245539);
        for (int k = 0; k < nkeys; ++k) h = (h ^ typed_bucket(key_dts[k], keys[k], i)) * UINT64_C(0x9E3779B97F4A7C15);
        uint64_t bk = (h ^ (h >> 29)) & (nb - 1);
        uint32_t g = head[bk];
        for (; g != 0xFFFFFFFFu; g = next[g]) {
            int eq = 1;
            for (int k = 0; k < nkeys && eq; ++k) eq = typed_equal(key_dts[k], keys[k], first_rows[g], i);
            if (eq) break;
        }
        if (g == 0xFFFFFFFFu) { g = G++; first_rows[g] = i; next[g] = head[bk]; head[bk] = g; }
        reversemap[i] = g;
    }
    *ngroups = G;
    free(head); free(next);
    return AQG_OK;
}

/* generated per-group loop: out[g] = op(col[vecs[g]]) (engine/ast.py:722-789) */
int AQCHK(grouped_reduce)(int op, int t, const void* x, uint32_t G, const uint32_t* offsets,
                          const uint32_t* counts, const uint32_t* row_ids, void* out) {
    size_t sz = dt_size(t);
    int odt = AQCHK(reduce_out_dtype)(op, t);
    if (!sz || odt == AQG_ERROR) return AQG_ERR_DTYPE;
    size_t osz = dt_size(odt);
    uint32_t maxc = 0;
    for (uint32_t g = 0; g < G; ++g) if (counts[g] > maxc) maxc = counts[g];
    void* tmp = malloc((size_t)(maxc ? maxc : 1) * sz);
    for (uint32_t g = 0; g < G; ++g) {
        unsigned char r16[16];
        AQCHK(gather)(t, x, row_ids + offsets[g], counts[g], tmp);
        int rc = AQCHK(reduce)(op, t, tmp, counts[g], r16);
        if (rc) { free(tmp); return rc; }
        memcpy((char*)out + (size_t)g * osz, r16, osz);
    }
    free(tmp);
    return AQG_OK;
}

/* ------------------------------------------------------------------------- */
/* join: the reference has no C++ join.  aq_map<key, rows> semantic:
 * pairs ordered by probe row, then build row ascending.  PARITY UNPINNED.       */
int AQCHK(join_pairs)(int t, const void* build_keys, uint32_t nb, const void* probe_keys, uint32_t np,
                      uint32_t* probe_rows, uint32_t* build_rows, uint64_t cap, uint64_t* m) {
    uint64_t d;
    if ((nb && !key_as_u64(t, build_keys, 0, &d)) || (np && !key_as_u64(t, probe_keys, 0, &d))) return AQG_ERR_DTYPE;
    /* chained hash on build side, chains kept in ascending build-row order */
    uint64_t nbk = 16; while (nbk < (uint64_t)nb * 2) nbk <<= 1;
    uint32_t* head = (uint32_t*)malloc(sizeof(uint32_t) * nbk);
    uint32_t* next = (uint32_t*)malloc(sizeof(uint32_t) * (nb ? nb : 1));
    memset(head, 0xFF, sizeof(uint32_t) * nbk);
    for (uint32_t ii = nb; ii-- > 0;) {
        uint64_t k; key_as_u64(t, build_keys, ii, &k);
        uint64_t h = wy_hash_u64(k) & (nbk - 1);
        next[ii] = head[h]; head[h] = ii;
    }
    uint64_t cnt = 0;
    for (uint32_t i = 0; i < np; ++i) {
        uint64_t k; key_as_u64(t, probe_keys, i, &k);
        for (uint32_t j = head[wy_hash_u64(k) & (nbk - 1)]; j != 0xFFFFFFFFu; j = next[j]) {
            uint64_t kb; key_as_u64(t, build_keys, j, &kb);
            if (kb == k) {
                if (probe_rows && cnt < cap) { probe_rows[cnt] = i; build_rows[cnt] = j; }
                ++cnt;
            }
        }
    }
    *m = cnt;
    free(head); free(next);
    return AQG_OK;
}

/* ------------------------------------------------------------------------- */
/* synthetic h2o / time-series columns.  MUST stay identical to
 * aquery2_amd/csrc/gen.hip.                                                    */
static inline uint64_t splitmix64(uint64_t z) {
    z += UINT64_C(0x9E3779B97F4A7C15);
    z = (z ^ (z >> 30)) * UINT64_C(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)) * UINT64_C(0x94D049BB133111EB);
    return z ^ (z >> 31);
}
static inline uint64_t gen_rnd(uint64_t seed, int col, uint64_t row) {
    return splitmix64(splitmix64(seed * 256 + (uint64_t)col) ^ row);
}
static inline uint32_t gen_uniform(uint64_t r, uint32_t range) { /* [0, range) multiply-shift */
    return (uint32_t)(((r >> 32) * (uint64_t)range) >> 32);
}
static inline int32_t tri_wave(uint64_t i, uint32_t period, int32_t amp) { /* integer triangle wave in [-amp, amp] */
    uint32_t ph = (uint32_t)(i % period);
    uint32_t half = period / 2;
    int64_t up = ph < half ? (int64_t)ph : (int64_t)(period - ph);
    return (int32_t)(2 * (int64_t)amp * up / (int64_t)half) - amp;
}
int AQCHK(gen_column)(int col, uint64_t seed, uint64_t row_base, uint32_t n, uint64_t n_total, uint32_t K, void* out) {
    if (K == 0) return AQG_ERR_ARG;
    uint64_t big = n_total / K; if (big < 1) big = 1; if (big > 0x7FFFFFFFu) big = 0x7FFFFFFFu;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t row = row_base + i;
        uint64_t r = gen_rnd(seed, col, row);
        switch (col) {
        case AQG_GEN_ID1: case AQG_GEN_ID2: case AQG_GEN_ID4: case AQG_GEN_ID5:
            ((int32_t*)out)[i] = 1 + (int32_t)gen_uniform(r, K); break;
        case AQG_GEN_ID3: case AQG_GEN_ID6:
            ((int32_t*)out)[i] = 1 + (int32_t)gen_uniform(r, (uint32_t)big); break;
        case AQG_GEN_V1: ((int32_t*)out)[i] = 1 + (int32_t)gen_uniform(r, 5); break;
        case AQG_GEN_V2: ((int32_t*)out)[i] = 1 + (int32_t)gen_uniform(r, 15); break;
        case AQG_GEN_V3: { /* round(U[0,100), 6 decimals) as float32 */
            uint64_t micro = ((r >> 32) * UINT64_C(100000000)) >> 32;
            ((float*)out)[i] = (float)((double)micro / 1e6);
        } break;
        case AQG_GEN_TIMESTAMP: ((int32_t*)out)[i] = (int32_t)(row + 1); break;
        case AQG_GEN_PRICE: { /* bounded pseudo-walk in [50, 500] */
            int32_t p = 275 + tri_wave(row, 1009, 100) + tri_wave(row, 104729, 100) + (int32_t)gen_uniform(r, 51) - 25;
            ((int32_t*)out)[i] = p < 50 ? 50 : (p > 500 ? 500 : p);
        } break;
        default: return AQG_ERR_ARG;
        }
    }
    return AQG_OK;
}

/* ------------------------------------------------------------------------- */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* reference-shaped group-by + sum path, single thread, timed (bench cpu_baseline "port") */
double AQCHK(time_groupby_sum)(int nkeys, const int* key_dts, const void* const* keys,
                               int nvals, const int* val_dts, const void* const* vals, uint32_t n,
                               uint32_t* ngroups_out, double* split3) {
    uint32_t* reversemap = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)(n ? n : 1));
    uint32_t* row_ids = reversemap + n;
    uint32_t* counts = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
    uint32_t* offsets = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
    uint32_t G = 0;
    double t0 = now_s();
    AQCHK(groupby)(nkeys, key_dts, keys, n, reversemap, &G, counts, NULL, NULL, NULL);
    double t1 = now_s();
    for (uint32_t g = 0; g < G; ++g) offsets[g] = counts[g];
    for (uint32_t g = 1; g < G; ++g) offsets[g] += offsets[g - 1];
    for (uint32_t i = 0; i < n; ++i) row_ids[--offsets[reversemap[i]]] = i;
    double t2 = now_s();
    for (int j = 0; j < nvals; ++j) {
        void* out = malloc((size_t)(G ? G : 1) * 16);
        AQCHK(grouped_reduce)(AQG_RED_SUM, val_dts[j], vals[j], G, offsets, counts, row_ids, out);
        free(out);
    }
    double t3 = now_s();
    if (ngroups_out) *ngroups_out = G;
    if (split3) { split3[0] = t1 - t0; split3[1] = t2 - t1; split3[2] = t3 - t2; }
    free(reversemap); free(counts); free(offsets);
    return t3 - t0;
}
