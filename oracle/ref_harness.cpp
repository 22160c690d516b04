/*
 * ref_harness.cpp -- exposes the REAL reference library through the checker ABI
 * of aq_oracle.h (prefix aqr_).  TEST INFRASTRUCTURE ONLY.
 *
 * Built only where /root/reference is mounted, from the reference sources where
 * they lie (nothing is copied): see oracle/Makefile, target _ref/libaqref.so.
 * The unity include of server/libaquery.cpp provides GC/ScratchSpace/Context
 * (the reference defines ScratchSpace's members `inline` in that .cpp).
 * This file contains no reference code: it only instantiates and calls the
 * reference's templates.
 */
#include "server/libaquery.cpp"
#include "server/aggregations.h"
#include "server/hasher.h"
#include "server/table.h"

#include <cstring>
#include <chrono>
#include <tuple>
#include <type_traits>

#define AQCHK(name) aqr_##name
#include "aq_oracle.h"
#include "../include/aqg.h"

namespace {

struct Boot {
    Boot() {
        new Context();               // creates the GC; every owning vector's dtor needs it
        GC::scratch_space = nullptr; // value-returning forms malloc, as in generated code outside group loops
    }
} boot;

template <class T> struct tag { using type = T; };

template <class F> int dispatch_num(int dt, F&& f) {
    switch (dt) {
    case AQG_INT8: return f(tag<signed char>{});
    case AQG_INT16: return f(tag<short>{});
    case AQG_INT32: return f(tag<int>{});
    case AQG_INT64: return f(tag<long>{});
    case AQG_UINT8: return f(tag<unsigned char>{});
    case AQG_UINT16: return f(tag<unsigned short>{});
    case AQG_UINT32: return f(tag<unsigned int>{});
    case AQG_UINT64: return f(tag<unsigned long>{});
    case AQG_FLOAT: return f(tag<float>{});
    case AQG_DOUBLE: return f(tag<double>{});
    }
    return AQG_ERR_DTYPE;
}
/* the binary-operator matrix is restricted to keep the build quick */
template <class F> int dispatch_bin(int dt, F&& f) {
    switch (dt) {
    case AQG_INT16: return f(tag<short>{});
    case AQG_INT32: return f(tag<int>{});
    case AQG_INT64: return f(tag<long>{});
    case AQG_UINT32: return f(tag<unsigned int>{});
    case AQG_FLOAT: return f(tag<float>{});
    case AQG_DOUBLE: return f(tag<double>{});
    }
    return AQG_ERR_DTYPE;
}

template <class T> constexpr int tag_of() {
    if constexpr (std::is_same_v<T, bool>) return AQG_BOOL;
    else if constexpr (std::is_same_v<T, const char*>) return AQG_STR;
    else if constexpr (std::is_same_v<T, __int128_t>) return AQG_INT128;
    else if constexpr (std::is_same_v<T, __uint128_t>) return AQG_UINT128;
    else if constexpr (std::is_same_v<T, float>) return AQG_FLOAT;
    else if constexpr (std::is_same_v<T, double>) return AQG_DOUBLE;
    else if constexpr (std::is_integral_v<T>) {
        constexpr bool u = std::is_unsigned_v<T>;
        if constexpr (sizeof(T) == 1) return u ? AQG_UINT8 : AQG_INT8;
        else if constexpr (sizeof(T) == 2) return u ? AQG_UINT16 : AQG_INT16;
        else if constexpr (sizeof(T) == 4) return u ? AQG_UINT32 : AQG_INT32;
        else return u ? AQG_UINT64 : AQG_INT64;
    } else return AQG_ERROR;
}

template <class T> vector_type<T> view(const void* p, uint32_t n) {
    return vector_type<T>(const_cast<T*>(static_cast<const T*>(p)), n);
}
template <class V> void copy_out(void* out, const V& v) {
    using E = std::decay_t<decltype(v[0])>;
    if (v.size) std::memcpy(out, v.container, sizeof(E) * (size_t)v.size);
}

} // namespace

namespace {
template <class Rec, class Push>
int run_groupby(uint32_t n, Push&& push, uint32_t* reversemap, uint32_t* ngroups, uint32_t* counts,
                uint32_t* offsets, uint32_t* row_ids, uint32_t* first_rows, const void* const* keys, int nkeys) {
    AQHashTable<Rec, transTypes<Rec, hasher>> g{n ? n : 1u};
    for (uint32_t i = 0; i < n; ++i) push(g, i);                     /* hashtable_push(Key&&, i) */
    uint32_t G = (uint32_t)g.size();
    *ngroups = G;
    std::memcpy(reversemap, g.reversemap, sizeof(uint32_t) * (size_t)n);
    std::memcpy(counts, g.ht_base, sizeof(uint32_t) * (size_t)G);
    if (first_rows) {
        /* values() holds the key tuples in group order; report the first row carrying each */
        std::vector<char> seen(G, 0);
        for (uint32_t i = 0; i < n; ++i) if (!seen[reversemap[i]]) { seen[reversemap[i]] = 1; first_rows[reversemap[i]] = i; }
    }
    if (offsets && row_ids && G) {
        auto vecs = g.ht_postproc(n);                                /* hasher.h:181-198 */
        std::memcpy(offsets, g.ht_base, sizeof(uint32_t) * (size_t)G);
        std::memcpy(row_ids, g.mapbase, sizeof(uint32_t) * (size_t)n);
        free(vecs);
    }
    free(g.reversemap); free(g.ht_base);
    (void)keys; (void)nkeys;
    return AQG_OK;
}
} // namespace

extern "C" {

int aqr_long_type(int dt) {
    int r = AQG_ERROR;
    dispatch_num(dt, [&](auto t) { r = tag_of<types::GetLongType<typename decltype(t)::type>>(); return 0; });
    return r;
}
int aqr_fp_type(int dt) {
    int r = AQG_ERROR;
    dispatch_num(dt, [&](auto t) { r = tag_of<types::GetFPType<typename decltype(t)::type>>(); return 0; });
    return r;
}
int aqr_coercion(int a, int b) {
    int r = AQG_ERROR;
    dispatch_num(a, [&](auto ta) {
        return dispatch_num(b, [&](auto tb) {
            r = tag_of<typename types::Coercion<typename decltype(ta)::type, typename decltype(tb)::type>::type>();
            return 0;
        });
    });
    return r;
}

/* free operators of server/table.h:820-937; result dtype reported from the
 * reference's own return type */
int aqr_ewise_out_dtype(int op, int lt, int rt) {
    int r = AQG_ERROR;
    dispatch_bin(lt, [&](auto ta) {
        return dispatch_bin(rt, [&](auto tb) {
            using A = typename decltype(ta)::type; using B = typename decltype(tb)::type;
            vector_type<A> a; vector_type<B> b;
            switch (op) {
            case AQG_OP_ADD: r = tag_of<std::decay_t<decltype((a + b)[0])>>(); break;
            case AQG_OP_SUB: r = tag_of<std::decay_t<decltype((a - b)[0])>>(); break;
            case AQG_OP_MUL: r = tag_of<std::decay_t<decltype(operator*(a, b)[0])>>(); break;
            case AQG_OP_DIV: r = tag_of<std::decay_t<decltype(operator/(a, b)[0])>>(); break;
            case AQG_OP_GT: r = AQG_BOOL; break;
            }
            return 0;
        });
    });
    return r;
}

/* kind VEC_VEC / VEC_SCALAR / SCALAR_VEC; ops + - * / > use the FREE operators
 * (table.h:820-937); the other comparisons and & | ^ use aqop_* (:954-973) with
 * the caller's `ot` as Ret, which only exist vec-vec with one VT.                */
int aqr_ewise(int op, int kind, int lt, const void* l, int rt, const void* r, int ot, void* out, uint32_t n) {
    return dispatch_bin(lt, [&](auto ta) {
        return dispatch_bin(rt, [&](auto tb) -> int {
            using A = typename decltype(ta)::type; using B = typename decltype(tb)::type;
            auto run_free = [&](auto fn) -> int {
                if (kind == AQG_VEC_VEC) {
                    auto a = view<A>(l, n); auto b = view<B>(r, n);
                    auto res = fn(a, b);
                    if (tag_of<std::decay_t<decltype(res[0])>>() != ot) return AQG_ERR_DTYPE;
                    copy_out(out, res);
                } else if (kind == AQG_VEC_SCALAR) {
                    auto a = view<A>(l, n); B b = *static_cast<const B*>(r);
                    auto res = fn(a, b);
                    if (tag_of<std::decay_t<decltype(res[0])>>() != ot) return AQG_ERR_DTYPE;
                    copy_out(out, res);
                } else {
                    A a = *static_cast<const A*>(l); auto b = view<B>(r, n);
                    auto res = fn(a, b);
                    if (tag_of<std::decay_t<decltype(res[0])>>() != ot) return AQG_ERR_DTYPE;
                    copy_out(out, res);
                }
                return AQG_OK;
            };
            switch (op) {
            case AQG_OP_ADD: return run_free([](const auto& x, const auto& y) { return x + y; });
            case AQG_OP_SUB: return run_free([](const auto& x, const auto& y) { return x - y; });
            case AQG_OP_MUL: return run_free([](const auto& x, const auto& y) { return operator*(x, y); });
            case AQG_OP_DIV: return run_free([](const auto& x, const auto& y) { return operator/(x, y); });
            case AQG_OP_GT: return run_free([](const auto& x, const auto& y) { return operator>(x, y); });
            default: break;
            }
            if (kind != AQG_VEC_VEC) return AQG_ERR_ARG;
            auto a = view<A>(l, n); auto b = view<B>(r, n);
            auto run_aqop = [&](auto rt_tag) -> int {
                using R = typename decltype(rt_tag)::type;
                vector_type<R> ret(static_cast<R*>(out), n);
                switch (op) {
                case AQG_OP_LT: aqop_lt(a, b, ret); break;
                case AQG_OP_GE: aqop_gte(a, b, ret); break;
                case AQG_OP_LE: aqop_lte(a, b, ret); break;
                case AQG_OP_EQ: aqop_eq(a, b, ret); break;
                case AQG_OP_NE: aqop_neq(a, b, ret); break;
                default:
                    if constexpr (std::is_integral_v<A> && std::is_integral_v<B>) {
                        switch (op) {
                        case AQG_OP_AND: aqop_and(a, b, ret); break;
                        case AQG_OP_OR: aqop_or(a, b, ret); break;
                        case AQG_OP_XOR: aqop_xor(a, b, ret); break;
                        default: return AQG_ERR_ARG;
                        }
                    } else return AQG_ERR_DTYPE;
                }
                return AQG_OK;
            };
            if (ot == AQG_BOOL) return run_aqop(tag<bool>{});
            if (ot == AQG_INT32) return run_aqop(tag<int>{});
            if (ot == AQG_INT64) return run_aqop(tag<long>{});
            return AQG_ERR_DTYPE;
        });
    });
}

int aqr_unary(int op, int t, const void* x, uint32_t n, uint32_t param, int ot, void* out) {
    return dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        auto v = view<T>(x, n);
        if (op == AQG_UN_SQRT) {
            if (ot != AQG_DOUBLE) return AQG_ERR_DTYPE;
            auto res = sqrt(v);
            copy_out(out, res);
            return AQG_OK;
        }
        if (op == AQG_UN_TRUNCATE) {
            if constexpr (std::is_floating_point_v<T>) {
                auto res = truncate(v, param);
                copy_out(out, res);
                return AQG_OK;
            }
            return AQG_ERR_DTYPE;
        }
        return AQG_ERR_ARG;
    });
}

int aqr_reduce_out_dtype(int op, int t) {
    int r = AQG_ERROR;
    dispatch_num(t, [&](auto tt) {
        using T = typename decltype(tt)::type;
        vector_type<T> v;
        switch (op) {
        case AQG_RED_SUM: r = tag_of<decltype(sum(v))>(); break;
        case AQG_RED_MIN: r = tag_of<decltype(min(v))>(); break;
        case AQG_RED_MAX: r = tag_of<decltype(max(v))>(); break;
        case AQG_RED_COUNT: r = AQG_UINT64; break;
        case AQG_RED_AVG: r = tag_of<decltype(avg(v))>(); break;
        case AQG_RED_VAR: r = tag_of<decltype(var(v))>(); break;
        case AQG_RED_STDDEV: r = tag_of<decltype(stddev(v))>(); break;
        case AQG_RED_FIRST: r = tag_of<decltype(first(v))>(); break;
        case AQG_RED_LAST: r = tag_of<decltype(last(v))>(); break;
        }
        return 0;
    });
    return r;
}

int aqr_reduce(int op, int t, const void* x, uint32_t n, void* out16) {
    std::memset(out16, 0, 16);
    return dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        auto v = view<T>(x, n);
        auto put = [&](auto val) { std::memcpy(out16, &val, sizeof val); };
        switch (op) {
        case AQG_RED_SUM: put(sum(v)); break;
        case AQG_RED_MIN: put(min(v)); break;
        case AQG_RED_MAX: put(max(v)); break;
        case AQG_RED_COUNT: put((uint64_t)count(v)); break;
        case AQG_RED_AVG: put(avg(v)); break;
        case AQG_RED_VAR: put(var(v)); break;
        case AQG_RED_STDDEV: put(stddev(v)); break;
        case AQG_RED_FIRST: put(first(v)); break;
        case AQG_RED_LAST: put(last(v)); break;
        default: return AQG_ERR_ARG;
        }
        return AQG_OK;
    });
}

int aqr_corr(int tx, const void* x, int ty, const void* y, uint32_t n, double* out) {
    return dispatch_bin(tx, [&](auto ta) {
        return dispatch_bin(ty, [&](auto tb) -> int {
            using A = typename decltype(ta)::type; using B = typename decltype(tb)::type;
            auto a = view<A>(x, n); auto b = view<B>(y, n);
            *out = (double)corr(a, b);
            return AQG_OK;
        });
    });
}

int aqr_scan_out_dtype(int op, int t) {
    int r = AQG_ERROR;
    dispatch_num(t, [&](auto tt) {
        using T = typename decltype(tt)::type;
        vector_type<T> v;
        switch (op) {
        case AQG_SCAN_SUMS: r = tag_of<std::decay_t<decltype(sums(v)[0])>>(); break;
        case AQG_SCAN_AVGS: r = tag_of<std::decay_t<decltype(avgs(v)[0])>>(); break;
        case AQG_SCAN_MINS: r = tag_of<std::decay_t<decltype(mins(v)[0])>>(); break;
        case AQG_SCAN_MAXS: r = tag_of<std::decay_t<decltype(maxs(v)[0])>>(); break;
        case AQG_SCAN_SUMW: r = tag_of<std::decay_t<decltype(sumw(1u, v)[0])>>(); break;
        case AQG_SCAN_AVGW: r = tag_of<std::decay_t<decltype(avgw(1u, v)[0])>>(); break;
        case AQG_SCAN_MINW: r = tag_of<std::decay_t<decltype(minw(1u, v)[0])>>(); break;
        case AQG_SCAN_MAXW: r = tag_of<std::decay_t<decltype(maxw(1u, v)[0])>>(); break;
        case AQG_SCAN_RATIOW: r = tag_of<std::decay_t<decltype(ratiow(1u, v)[0])>>(); break;
        case AQG_SCAN_DELTAS: r = tag_of<std::decay_t<decltype(deltas(v)[0])>>(); break;
        case AQG_SCAN_PREV: r = tag_of<std::decay_t<decltype(prev(v)[0])>>(); break;
        case AQG_SCAN_NEXT: r = tag_of<std::decay_t<decltype(aggnext(v)[0])>>(); break;
        default: r = AQG_DOUBLE; break;
        }
        return 0;
    });
    return r;
}

/* vars/stddevs print from inside the loop (aggregations.h:368) and varw/stddevw read
 * out of bounds (:311-312): neither is callable as a pin, so they return ERR_DTYPE.  */
int aqr_scan(int op, int t, const void* x, uint32_t n, uint32_t w, void* out) {
    return dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        auto v = view<T>(x, n);
        switch (op) {
        case AQG_SCAN_SUMS: copy_out(out, sums(v)); break;
        case AQG_SCAN_AVGS: copy_out(out, avgs(v)); break;
        case AQG_SCAN_MINS: copy_out(out, mins(v)); break;
        case AQG_SCAN_MAXS: copy_out(out, maxs(v)); break;
        case AQG_SCAN_SUMW: copy_out(out, sumw(w, v)); break;
        case AQG_SCAN_AVGW: copy_out(out, avgw(w, v)); break;
        case AQG_SCAN_MINW: copy_out(out, minw(w, v)); break;
        case AQG_SCAN_MAXW: copy_out(out, maxw(w, v)); break;
        case AQG_SCAN_RATIOW: if (!n) return AQG_OK; copy_out(out, ratiow(w, v)); break;
        case AQG_SCAN_DELTAS: copy_out(out, deltas(v)); break;
        case AQG_SCAN_PREV: copy_out(out, prev(v)); break;
        case AQG_SCAN_NEXT: copy_out(out, aggnext(v)); break;
        default: return AQG_ERR_DTYPE;
        }
        return AQG_OK;
    });
}

int aqr_gather(int t, const void* x, const uint32_t* idx, uint32_t m, void* out) {
    return dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        ColRef<T> c(0u, const_cast<void*>(x));
        vector_type<uint32_t> iv(const_cast<uint32_t*>(idx), m);
        auto res = c[iv];
        copy_out(out, res);
        return AQG_OK;
    });
}

/* ColRef::operator[](const std::vector<bool>&) (table.h:190-198) returns `size`
 * uninitialised slots followed by the selected values (defect D11); only that
 * selected tail is handed back.                                                   */
int aqr_compact(int t, const void* x, const uint8_t* mask, uint32_t n, void* out, uint32_t* m) {
    return dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        ColRef<T> c(n, const_cast<void*>(x));
        std::vector<bool> mk(n);
        for (uint32_t i = 0; i < n; ++i) mk[i] = mask[i] != 0;
        auto res = c[mk];
        uint32_t sel = res.size - n;
        std::memcpy(out, res.container + n, sizeof(T) * (size_t)sel);
        *m = sel;
        return AQG_OK;
    });
}

uint64_t aqr_hash_scalar(int t, const void* v) {
    uint64_t h = 0;
    dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        if constexpr (std::is_integral_v<T>) h = ankerl::unordered_dense::hash<T>()(*static_cast<const T*>(v));
        return 0;
    });
    return h;
}

/* tuple hashes for the arities/types the parity tests use (int32 fields) */
uint64_t aqr_hash_tuple(int nkeys, const int* dts, const void* const* vals) {
    for (int i = 0; i < nkeys; ++i) if (dts[i] != AQG_INT32) return 0;
    auto g = [&](int i) { return *static_cast<const int*>(vals[i]); };
    switch (nkeys) {
    case 1: return hasher<int>()(std::make_tuple(g(0)));
    case 2: return hasher<int, int>()(std::make_tuple(g(0), g(1)));
    case 3: return hasher<int, int, int>()(std::make_tuple(g(0), g(1), g(2)));
    case 6: return hasher<int, int, int, int, int, int>()(std::make_tuple(g(0), g(1), g(2), g(3), g(4), g(5)));
    }
    return 0;
}


/* int32 key columns, arity 1..3 and 6 (h2o Q1/Q2/Q10 shapes), plus one int64 key */
int aqr_groupby(int nkeys, const int* key_dts, const void* const* keys, uint32_t n,
                uint32_t* reversemap, uint32_t* ngroups, uint32_t* counts,
                uint32_t* offsets, uint32_t* row_ids, uint32_t* first_rows) {
    if (nkeys == 1 && key_dts[0] == AQG_INT64) {
        using R = record<long>; auto k0 = static_cast<const long*>(keys[0]);
        return run_groupby<R>(n, [&](auto& g, uint32_t i) { g.hashtable_push(std::forward_as_tuple(k0[i]), i); },
                              reversemap, ngroups, counts, offsets, row_ids, first_rows, keys, nkeys);
    }
    for (int i = 0; i < nkeys; ++i) if (key_dts[i] != AQG_INT32) return AQG_ERR_DTYPE;
    auto K = [&](int j) { return static_cast<const int*>(keys[j]); };
    switch (nkeys) {
    case 1: { using R = record<int>; auto a = K(0);
        return run_groupby<R>(n, [&](auto& g, uint32_t i) { g.hashtable_push(std::forward_as_tuple(a[i]), i); },
                              reversemap, ngroups, counts, offsets, row_ids, first_rows, keys, nkeys); }
    case 2: { using R = record<int, int>; auto a = K(0), b = K(1);
        return run_groupby<R>(n, [&](auto& g, uint32_t i) { g.hashtable_push(std::forward_as_tuple(a[i], b[i]), i); },
                              reversemap, ngroups, counts, offsets, row_ids, first_rows, keys, nkeys); }
    case 3: { using R = record<int, int, int>; auto a = K(0), b = K(1), c = K(2);
        return run_groupby<R>(n, [&](auto& g, uint32_t i) { g.hashtable_push(std::forward_as_tuple(a[i], b[i], c[i]), i); },
                              reversemap, ngroups, counts, offsets, row_ids, first_rows, keys, nkeys); }
    case 6: { using R = record<int, int, int, int, int, int>; auto a = K(0), b = K(1), c = K(2), d = K(3), e = K(4), f = K(5);
        return run_groupby<R>(n, [&](auto& g, uint32_t i) { g.hashtable_push(std::forward_as_tuple(a[i], b[i], c[i], d[i], e[i], f[i]), i); },
                              reversemap, ngroups, counts, offsets, row_ids, first_rows, keys, nkeys); }
    }
    return AQG_ERR_ARG;
}

/* the generated group loop: scratch arena on, out[g] = op(col[vecs[g]]), release per group
 * (engine/ast.py:720-790) */
int aqr_grouped_reduce(int op, int t, const void* x, uint32_t G, const uint32_t* offsets,
                       const uint32_t* counts, const uint32_t* row_ids, void* out) {
    return dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        ColRef<T> col(0u, const_cast<void*>(x));
        int odt = aqr_reduce_out_dtype(op, t);
        size_t osz = odt == AQG_INT128 || odt == AQG_UINT128 ? 16 : (odt == AQG_DOUBLE || odt == AQG_UINT64 || odt == AQG_INT64 ? 8 : sizeof(T));
        GC::scratch_space = &GC::gc_handle->scratch;
        for (uint32_t g = 0; g < G; ++g) {
            vector_type<uint32_t> val(const_cast<uint32_t*>(row_ids) + offsets[g], counts[g]);
            char* o = static_cast<char*>(out) + (size_t)g * osz;
            auto put = [&](auto v) { std::memcpy(o, &v, sizeof v); };
            switch (op) {
            case AQG_RED_SUM: put(sum(col[val])); break;
            case AQG_RED_MIN: put(min(col[val])); break;
            case AQG_RED_MAX: put(max(col[val])); break;
            case AQG_RED_COUNT: put((uint64_t)val.size); break;
            case AQG_RED_AVG: put(avg(col[val])); break;
            case AQG_RED_VAR: put(var(col[val])); break;
            case AQG_RED_STDDEV: put(stddev(col[val])); break;
            case AQG_RED_FIRST: put(first(col[val])); break;
            case AQG_RED_LAST: put(last(col[val])); break;
            }
            GC::scratch_space->release();
        }
        GC::scratch_space = nullptr;
        return AQG_OK;
    });
}

int aqr_join_pairs(int, const void*, uint32_t, const void*, uint32_t, uint32_t*, uint32_t*, uint64_t, uint64_t*) {
    return AQG_ERR_DTYPE; /* the reference has no C++ join */
}
int aqr_gen_column(int, uint64_t, uint64_t, uint32_t, uint64_t, uint32_t, void*) { return AQG_ERR_DTYPE; }

/* reference Q1/Q5-shaped path, timed: AQHashTable build -> ht_postproc -> per-group gather+sum */
double aqr_time_groupby_sum(int nkeys, const int* key_dts, const void* const* keys,
                            int nvals, const int* val_dts, const void* const* vals, uint32_t n,
                            uint32_t* ngroups_out, double* split3) {
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    if (nkeys != 1 || key_dts[0] != AQG_INT32) return -1.0;
    using R = record<int>;
    auto k0 = static_cast<const int*>(keys[0]);
    auto t0 = clk::now();
    AQHashTable<R, transTypes<R, hasher>> g{n ? n : 1u};
    for (uint32_t i = 0; i < n; ++i) g.hashtable_push(std::forward_as_tuple(k0[i]), i);
    auto t1 = clk::now();
    auto vecs = g.ht_postproc(n);
    auto t2 = clk::now();
    uint32_t G = (uint32_t)g.size();
    for (int j = 0; j < nvals; ++j) {
        if (val_dts[j] == AQG_INT32) {
            ColRef<int> col(0u, const_cast<void*>(vals[j]));
            vector_type<__int128_t> o(G);                 /* output column: malloc'd, outside the arena */
            GC::scratch_space = &GC::gc_handle->scratch;
            for (uint32_t q = 0; q < G; ++q) { o[q] = sum(col[vecs[q]]); GC::scratch_space->release(); }
            GC::scratch_space = nullptr;
        } else if (val_dts[j] == AQG_FLOAT) {
            ColRef<float> col(0u, const_cast<void*>(vals[j]));
            vector_type<double> o(G);
            GC::scratch_space = &GC::gc_handle->scratch;
            for (uint32_t q = 0; q < G; ++q) { o[q] = sum(col[vecs[q]]); GC::scratch_space->release(); }
            GC::scratch_space = nullptr;
        }
    }
    auto t3 = clk::now();
    if (ngroups_out) *ngroups_out = G;
    if (split3) { split3[0] = secs(t0, t1); split3[1] = secs(t1, t2); split3[2] = secs(t2, t3); }
    free(vecs); free(g.reversemap); free(g.ht_base);
    return secs(t0, t3);
}

} // extern "C"

/* key columns that are not plain integers, through the reference's own AQHashTable / hasher (server/hasher.h:66-199): the menu of
 * tuple shapes the golden vectors cover (one column of each type; some paired with an int column: the multi-column hasher) */
namespace {
template <class Rec, class Push> int run_typed(uint32_t n, Push&& push, uint32_t* reversemap, uint32_t* ngroups, uint32_t* first_rows) {
    AQHashTable<Rec, transTypes<Rec, hasher>> g{n ? n : 1u};
    for (uint32_t i = 0; i < n; ++i) push(g, i);
    const uint32_t G = (uint32_t)g.size();
    *ngroups = G;
    std::memcpy(reversemap, g.reversemap, sizeof(uint32_t) * (size_t)n);
    std::vector<char> seen(G, 0);
    for (uint32_t i = 0; i < n; ++i) if (!seen[reversemap[i]]) { seen[reversemap[i]] = 1; first_rows[reversemap[i]] = i; }
    free(g.reversemap); free(g.ht_base);
    return AQG_OK;
}
template <class T> int typed1(const void* col, uint32_t n, uint32_t* rm, uint32_t* ng, uint32_t* fr) {
    auto k = static_cast<const T*>(col);
    return run_typed<record<T>>(n, [&](auto& g, uint32_t i) { g.hashtable_push(std::forward_as_tuple(k[i]), i); }, rm, ng, fr);
}
template <class T> int typed2(const void* col, const void* icol, uint32_t n, uint32_t* rm, uint32_t* ng, uint32_t* fr) {
    auto k = static_cast<const T*>(col);
    auto j = static_cast<const int*>(icol);
    return run_typed<record<T, int>>(n, [&](auto& g, uint32_t i) { g.hashtable_push(std::forward_as_tuple(k[i], j[i]), i); }, rm, ng, fr);
}
} // namespace
extern "C" int aqr_groupby_typed(int nkeys, const int* dts, const void* const* keys, uint32_t n, uint32_t* rm, uint32_t* ng, uint32_t* fr) {
    if (nkeys == 1) {
        switch (dts[0]) {
        case AQG_FLOAT: return typed1<float>(keys[0], n, rm, ng, fr);
        case AQG_DOUBLE: return typed1<double>(keys[0], n, rm, ng, fr);
        case AQG_DATE: return typed1<types::date_t>(keys[0], n, rm, ng, fr);
        case AQG_TIME: return typed1<types::time_t>(keys[0], n, rm, ng, fr);
        case AQG_TIMESTAMP: return typed1<types::timestamp_t>(keys[0], n, rm, ng, fr);
        case AQG_STR: return typed1<astring_view>(keys[0], n, rm, ng, fr);
        case AQG_UINT64: return typed1<const char*>(keys[0], n, rm, ng, fr);      /* raw string pointers: pointer equality */
        }
    } else if (nkeys == 2 && dts[1] == AQG_INT32) {
        switch (dts[0]) {
        case AQG_DOUBLE: return typed2<double>(keys[0], keys[1], n, rm, ng, fr);
        case AQG_FLOAT: return typed2<float>(keys[0], keys[1], n, rm, ng, fr);
        case AQG_DATE: return typed2<types::date_t>(keys[0], keys[1], n, rm, ng, fr);
        case AQG_TIMESTAMP: return typed2<types::timestamp_t>(keys[0], keys[1], n, rm, ng, fr);
        case AQG_INT128: return typed2<__int128>(keys[0], keys[1], n, rm, ng, fr);
        case AQG_STR: return typed2<astring_view>(keys[0], keys[1], n, rm, ng, fr);
        }
    }
    return AQG_ERR_DTYPE;
}

/* result egress: the shapes of tests/emitted/print_shapes.inc printed by the REFERENCE's TableInfo::print / printall.  Built as a
 * small executable of its own (oracle/Makefile: _ref/print_shapes_ref, -DAQR_PRINT_SHAPES_MAIN); its stdout is the golden text. */
#ifdef AQR_PRINT_SHAPES_MAIN
#include "../tests/emitted/print_shapes.inc"
int main() { aq_print_shapes(); return 0; }
#endif
