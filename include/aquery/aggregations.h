// aggregations.h -- reductions, prefix scans, sliding windows and shifts of the AQuery library API
// (names, argument order and result types of the reference's server/aggregations.h; `avgs(w, x)` in SQL is
// emitted as `avgw(w, x)`, `next` as `aggnext`: common/types.py:285-290,334).  Every vector argument is
// reduced / scanned by one HIP kernel through the C-ABI; scalars fall through to the identity overloads
// at the bottom exactly like the reference's.
#pragma once
#include <cmath>
#include <limits>
#include <utility>

#include "gc.h"
#include "types.h"
#include "vector_type.hpp"
#undef max
#undef min

namespace aq {

template <template <typename...> class VT, class T> constexpr bool is_vt = std::is_base_of_v<vector_base<T>, VT<T>>;

template <class R, class T, template <typename...> class VT> inline R device_reduce(int op, const VT<T>& v) {
    auto& rt = dev::Runtime::get();
    alignas(16) unsigned char buf[16];
    if (!rt.deferred_reduce(v.container, op, buf)) {      // not a deferred `col[vecs[g]]`: reduce this very vector
        dev::In in(v.container, (size_t)v.size * sizeof(T), v.capacity == 0);
        dev::check(aqg_reduce(rt.ctx(), op, dev::tag_of<T>::value, in.d, v.size, buf), "aqg_reduce");
    }
    R r;
    std::memcpy(&r, buf, sizeof(R));
    return r;
}

// ret = scan(op, w, arr): `ret` may be a fresh vector or the caller's out-parameter
template <class T, template <typename...> class VT, class Ret> inline void device_scan(int op, uint32_t w, const VT<T>& arr, Ret& ret) {
    using RT = std::remove_cv_t<std::remove_pointer_t<decltype(ret.container)>>;
    auto& rt = dev::Runtime::get();
    const uint32_t n = arr.size;
    if (n == 0) return;
    const int want = aqg_scan_out_dtype(op, dev::tag_of<T>::value);
    if (want != dev::tag_of<RT>::value) { std::fprintf(stderr, "[aquery-mi355x] scan result element type mismatch (op %d)\n", op); std::abort(); }
    // a per-group temporary of the generated loop (`avgw(10, sales[vecs[i]], col[i])`, mem_opt.cpp:61; engine/ast.py:749-784): the scan
    // runs ONCE for all groups over the flat layout (aqg_grouped_scan_flat) and `ret` becomes this group's slice of that result
    if (dev::Entry* e = rt.deferred_at(arr.container)) {
        dev::GroupCtx* gc = e->dgroup;
        const uint32_t g = e->dg;
        const int r = rt.vcol_scan(gc, e->dv, op, w);
        rt.defer_slice(ret.container, (size_t)n * sizeof(RT), gc, g, r);
        if (ret.capacity == 0 && GC::scratch_space == nullptr) rt.touch(ret.container);
        return;
    }
    void* dout = rt.result(ret.container, (size_t)n * sizeof(RT));
    dev::In in(arr.container, (size_t)n * sizeof(T), arr.capacity == 0);
    dev::check(aqg_scan(rt.ctx(), op, dev::tag_of<T>::value, in.d, n, w, dout), "aqg_scan");
    if (ret.capacity == 0 && GC::scratch_space == nullptr) rt.touch(ret.container);
}
template <class T> using fp_of_long = types::GetFPType<types::GetLongType<T>>;   // double for every numeric T
} // namespace aq

// ---- reductions ---------------------------------------------------------------------------------------------------
template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
size_t count(const VT<T>& v) { return v.size; }

template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
types::GetLongType<T> sum(const VT<T>& v) { return aq::device_reduce<types::GetLongType<T>>(AQG_RED_SUM, v); }

template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
double avg(const VT<T>& v) { return aq::device_reduce<double>(AQG_RED_AVG, v); }

// max seeds with numeric_limits<T>::min() like the reference (defect D8 kept: max({-1.,-2.}) == DBL_MIN)
template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
T max(const VT<T>& v) { return aq::device_reduce<T>(AQG_RED_MAX, v); }
template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
T min(const VT<T>& v) { return aq::device_reduce<T>(AQG_RED_MIN, v); }

template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
aq::fp_of_long<decays<T>> var(const VT<T>& v) { return aq::device_reduce<double>(AQG_RED_VAR, v); }
template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
aq::fp_of_long<decays<T>> stddev(const VT<T>& v) { return aq::device_reduce<double>(AQG_RED_STDDEV, v); }

template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
T first(const VT<T>& v) { return v.size ? aq::device_reduce<T>(AQG_RED_FIRST, v) : T(0); }
template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
T last(const VT<T>& v) { return v.size ? aq::device_reduce<T>(AQG_RED_LAST, v) : T(0); }

template <class T, template <typename...> class VT, class T2, template <typename...> class VT2,
          std::enable_if_t<aq::is_vt<VT, T> && aq::is_vt<VT2, T2>>* = nullptr>
double corr(const VT<T>& x, const VT2<T2>& y) {
    auto& rt = aq::dev::Runtime::get();
    double r = 0;
    if (rt.deferred_corr(x.container, y.container, &r)) return r;      // `corr(v1[val], v2[val])` (h2o Q9): one grouped pass for all groups
    aq::dev::In a(x.container, (size_t)x.size * sizeof(T), x.capacity == 0), b(y.container, (size_t)y.size * sizeof(T2), y.capacity == 0);
    aq::dev::check(aqg_corr(rt.ctx(), aq::dev::tag_of<T>::value, a.d, aq::dev::tag_of<T2>::value, b.d, x.size, &r), "aqg_corr");
    return r;
}

// ---- element-wise ----------------------------------------------------------------------------------------------------
template <class T, template <typename...> class VT, class Ret, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
void sqrt(const VT<T>& v, Ret& ret) {
    auto& rt = aq::dev::Runtime::get();
    if (!v.size) return;
    void* dout = rt.result(ret.container, (size_t)v.size * sizeof(double));
    aq::dev::In in(v.container, (size_t)v.size * sizeof(T), v.capacity == 0);
    aq::dev::check(aqg_unary(rt.ctx(), AQG_UN_SQRT, aq::dev::tag_of<T>::value, in.d, v.size, 0, AQG_DOUBLE, dout), "aqg_unary");
    if (ret.capacity == 0 && GC::scratch_space == nullptr) rt.touch(ret.container);
}
template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
VT<double> sqrt(const VT<T>& v) { VT<double> ret(v.size); sqrt(v, ret); return ret; }

template <class T, std::enable_if_t<std::is_arithmetic_v<T>>* = nullptr>
T truncate(const T& v, const uint32_t precision) {
    auto m = std::pow(10, precision);
    if (v >= std::numeric_limits<T>::max() / m || aq_fp_precision<T> <= precision) return v;
    return std::round(v * m) / m;
}
template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
VT<T> truncate(const VT<T>& v, const uint32_t precision) {
    VT<T> ret(v.size);
    if (!v.size) return ret;
    auto& rt = aq::dev::Runtime::get();
    void* dout = rt.result(ret.container, (size_t)v.size * sizeof(T));
    aq::dev::In in(v.container, (size_t)v.size * sizeof(T), v.capacity == 0);
    aq::dev::check(aqg_unary(rt.ctx(), AQG_UN_TRUNCATE, aq::dev::tag_of<T>::value, in.d, v.size, precision, aq::dev::tag_of<T>::value, dout), "aqg_unary");
    return ret;
}
template <class X, class Y, class Z> void pow(X x, Y y, Z& z) { z = std::pow(x, y); }

// ---- scans, windows, shifts: out-parameter form + value-returning form -------------------------------------------------
#define AQ_SCAN(name, code, RET_T)                                                                   \
    template <class T, template <typename...> class VT, class Ret, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr> \
    void name(const VT<T>& arr, Ret& ret) { aq::device_scan(code, 0, arr, ret); }                    \
    template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr> \
    inline decayed_t<VT, RET_T> name(const VT<T>& arr) { decayed_t<VT, RET_T> ret(arr.size); aq::device_scan(code, 0, arr, ret); return ret; }
#define AQ_WINDOW(name, code, RET_T)                                                                 \
    template <class T, template <typename...> class VT, class Ret, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr> \
    void name(uint32_t w, const VT<T>& arr, Ret& ret) { aq::device_scan(code, w, arr, ret); }        \
    template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr> \
    inline decayed_t<VT, RET_T> name(uint32_t w, const VT<T>& arr) { decayed_t<VT, RET_T> ret(arr.size); aq::device_scan(code, w, arr, ret); return ret; }

AQ_SCAN(mins, AQG_SCAN_MINS, T)
AQ_SCAN(maxs, AQG_SCAN_MAXS, T)
AQ_SCAN(sums, AQG_SCAN_SUMS, types::GetLongType<T>)
AQ_SCAN(avgs, AQG_SCAN_AVGS, aq::fp_of_long<T>)
AQ_SCAN(vars, AQG_SCAN_VARS, aq::fp_of_long<T>)
AQ_SCAN(stddevs, AQG_SCAN_STDDEVS, aq::fp_of_long<T>)
AQ_SCAN(deltas, AQG_SCAN_DELTAS, T)
AQ_SCAN(prev, AQG_SCAN_PREV, T)
AQ_SCAN(aggnext, AQG_SCAN_NEXT, T)
AQ_WINDOW(minw, AQG_SCAN_MINW, T)
AQ_WINDOW(maxw, AQG_SCAN_MAXW, T)
AQ_WINDOW(sumw, AQG_SCAN_SUMW, types::GetLongType<T>)
AQ_WINDOW(avgw, AQG_SCAN_AVGW, aq::fp_of_long<T>)
AQ_WINDOW(varw, AQG_SCAN_VARW, aq::fp_of_long<T>)
AQ_WINDOW(stddevw, AQG_SCAN_STDDEVW, aq::fp_of_long<T>)
AQ_WINDOW(ratiow, AQG_SCAN_RATIOW, types::GetFPType<T>)
#undef AQ_SCAN
#undef AQ_WINDOW

template <class T, template <typename...> class VT, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
inline decayed_t<VT, types::GetFPType<T>> ratios(const VT<T>& arr) { return ratiow(1, arr); }
template <class T, template <typename...> class VT, class Ret, std::enable_if_t<aq::is_vt<VT, T>>* = nullptr>
inline void ratios(const VT<T>& arr, Ret& ret) { ratiow(1, arr, ret); }

// ---- scalar fall-backs (non-vector arguments): identity-like, as in the reference (:499-527) --------------------------
template <class T, std::enable_if_t<std::is_arithmetic_v<T>>* = nullptr> constexpr size_t count(const T&) { return 1; }
#define AQ_SCALAR_ID(name) template <class T, std::enable_if_t<std::is_arithmetic_v<T>>* = nullptr> constexpr T name(const T& v) { return v; }
#define AQ_SCALAR_ZERO(name) template <class T, std::enable_if_t<std::is_arithmetic_v<T>>* = nullptr> constexpr T name(const T&) { return 0; }
#define AQ_SCALAR_WID(name) template <class T, std::enable_if_t<std::is_arithmetic_v<T>>* = nullptr> constexpr T name(uint32_t, const T& v) { return v; }
AQ_SCALAR_ID(max) AQ_SCALAR_ID(min) AQ_SCALAR_ID(avg) AQ_SCALAR_ID(sum) AQ_SCALAR_ID(maxs) AQ_SCALAR_ID(mins) AQ_SCALAR_ID(avgs)
AQ_SCALAR_ID(sums) AQ_SCALAR_ID(last) AQ_SCALAR_ID(first) AQ_SCALAR_ID(prev) AQ_SCALAR_ID(aggnext)
AQ_SCALAR_ZERO(var) AQ_SCALAR_ZERO(vars) AQ_SCALAR_ZERO(stddev) AQ_SCALAR_ZERO(stddevs) AQ_SCALAR_ZERO(deltas)
AQ_SCALAR_WID(maxw) AQ_SCALAR_WID(minw) AQ_SCALAR_WID(avgw) AQ_SCALAR_WID(sumw)
template <class T, std::enable_if_t<std::is_arithmetic_v<T>>* = nullptr> constexpr T varw(uint32_t, const T&) { return 0; }
template <class T, std::enable_if_t<std::is_arithmetic_v<T>>* = nullptr> constexpr T stddevw(uint32_t, const T&) { return 0; }
template <class T, std::enable_if_t<std::is_arithmetic_v<T>>* = nullptr> constexpr T ratiow(uint32_t, const T&) { return 1; }
template <class T, std::enable_if_t<std::is_arithmetic_v<T>>* = nullptr> constexpr T ratios(const T&) { return 1; }
#undef AQ_SCALAR_ID
#undef AQ_SCALAR_ZERO
#undef AQ_SCALAR_WID
