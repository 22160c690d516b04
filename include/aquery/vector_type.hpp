// vector_type.hpp -- the column batch of the AQuery library API: `{T* container; uint32_t size, capacity}`,
// packed to 16 bytes, owning iff capacity > 0 (reference server/vector_type.hpp:30-431; C view
// vectortype_cstorage :25-28).  Clean-room: same names / layout / ownership rules, but element-wise
// operators and comparisons run as HIP kernels through the C-ABI, and a vector produced on the device
// downloads its host buffer only when the host first looks at it (see device.h).
// The six member comparisons are implemented correctly (the reference's all compute `>`: defect D7).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <iostream>
#include <unordered_set>

template <typename _Ty> class vector_type;

#include "device.h"
#include "gc.h"
#include "types.h"

#pragma pack(push, 1)

struct vectortype_cstorage {
    void* container;
    unsigned int size, capacity;
};

namespace aq {
// ret = l OP r on the device.  L / R are vector-like ({container,size,capacity}) or arithmetic scalars.
template <class RT, class L, class R> inline void device_binary(int op, const L& l, const R& r, vector_type<RT>& ret);
template <class T> inline vector_type<T> device_result(uint32_t n, void** dptr);
}

template <typename _Ty>
class vector_type : public vector_base<_Ty> {
public:
    typedef vector_type<_Ty> Decayed_t;
    _Ty* container;
    uint32_t size, capacity;
    typedef _Ty* iterator_t;
    typedef const _Ty* const_iterator;
    typedef std::conditional_t<is_cstr<_Ty>(), astring_view, _Ty> value_t;

    // ---- construction -----------------------------------------------------------------------------
    explicit vector_type(const uint32_t& n) : size(n), capacity(n) {
        if (GC::scratch_space != nullptr) {            // per-group temporaries live in the arena (engine/ast.py:720,786)
            container = static_cast<_Ty*>(GC::scratch_space->alloc(n * sizeof(_Ty)));
            capacity = 0;
        } else container = static_cast<_Ty*>(std::malloc((n ? n : 1) * sizeof(_Ty)));
    }
    vector_type(std::initializer_list<_Ty> l) : size((uint32_t)l.size()), capacity((uint32_t)l.size()) {
        container = static_cast<_Ty*>(std::malloc((capacity ? capacity : 1) * sizeof(_Ty)));
        uint32_t i = 0;
        for (const auto& v : l) container[i++] = v;
    }
    constexpr vector_type() noexcept : container(nullptr), size(0), capacity(0) {}
    constexpr vector_type(_Ty* data, uint32_t len) noexcept : container(data), size(len), capacity(0) {}
    vector_type(const uint32_t n, void* data) : container(static_cast<_Ty*>(data)), size(n), capacity(0) {}
    explicit vector_type(vectortype_cstorage s) noexcept : container(static_cast<_Ty*>(s.container)), size(s.size), capacity(s.capacity) {}
    // copy of an lvalue is a non-owning reference; of a const one a deep copy when the source owns
    vector_type(vector_type<_Ty>& o) noexcept : container(o.container), size(o.size), capacity(0) {}
    explicit vector_type(const vector_type<_Ty>& o) : container(nullptr), size(0), capacity(0) { copy_from(o); }
    vector_type(vector_type<_Ty>&& o) noexcept : container(o.container), size(o.size), capacity(o.capacity) { o.container = nullptr; o.size = o.capacity = 0; }

    void init_from(const uint32_t n, void* data) { container = static_cast<_Ty*>(data); size = n; capacity = 0; }

    ~vector_type() {
        if (capacity > 0 && container) {
            aq::dev::Runtime::get().forget(container);
            if (GC::gc_handle) GC::gc_handle->reg(container, (uint32_t)(sizeof(_Ty) * capacity)); else std::free(container);
        }
        container = nullptr;
        size = capacity = 0;
    }

    // ---- assignment ---------------------------------------------------------------------------------
    vector_type<_Ty>& operator=(const _Ty& v) {
        if (!container) { container = static_cast<_Ty*>(std::malloc(sizeof(_Ty))); capacity = 1; }
        host();
        size = 1;
        container[0] = v;
        return *this;
    }
    vector_type<_Ty>& operator=(const vector_type<_Ty>& o) { if (this != &o) { drop(); copy_from(o); } return *this; }
    vector_type<_Ty>& operator=(vector_type<_Ty>&& o) noexcept {
        if (this == &o) return *this;
        // A non-owning view into the per-group scratch arena (`col[i] = v3[val].subvec(0, 2)`, benchmark/h2o/groupby.sql:17 as engine/expr.py:237
        // emits it) would dangle as soon as the iteration releases the arena -- in the reference every group ends up showing the last
        // group's bytes.  The receiver takes a copy instead (host access: a deferred per-group temporary is materialised first).
        if (o.capacity == 0 && o.container && o.size && GC::scratch_space != nullptr && GC::scratch_space->owns(o.container) && !GC::scratch_space->owns(this)) {
            drop();
            size = capacity = o.size;
            container = static_cast<_Ty*>(std::malloc(size * sizeof(_Ty)));
            std::memcpy((void*)container, (const void*)o.host(), size * sizeof(_Ty));
            return *this;
        }
        drop(); container = o.container; size = o.size; capacity = o.capacity; o.container = nullptr; o.size = o.capacity = 0;
        return *this;
    }
    template <template <class> class VT> vector_type<_Ty>& operator=(const VT<_Ty>& o) {
        drop();
        size = capacity = o.size;
        container = static_cast<_Ty*>(std::malloc((size ? size : 1) * sizeof(_Ty)));
        for (uint32_t i = 0; i < size; ++i) container[i] = o[i];
        return *this;
    }

    // ---- host access (downloads a device-produced buffer on first use) ---------------------------------
    inline _Ty* host() const { if (aq::dev::Runtime::get().stale) aq::dev::Runtime::get().touch(container); return container; }
    inline _Ty& operator[](const uint32_t i) const { return host()[i]; }
    inline iterator_t begin() const { return host(); }
    inline iterator_t end() const { return host() + size; }
    inline _Ty& back() { return host()[size - 1]; }
    iterator_t find(const _Ty item) const { iterator_t c = begin(), e = c + size; while (c != e && *c != item) ++c; return c; }

    // ---- growth ---------------------------------------------------------------------------------------
    template <bool _grow = true, bool _resize = false> inline void grow(uint32_t sz = 0) {
        if constexpr (_grow) sz = size;
        if (sz >= capacity) {
            uint32_t ncap = _grow ? size + 1 + (size >> 1) : sz;
            _Ty* old = host();
            _Ty* n;
            if (capacity == 0) {                         // borrowed / empty: take ownership of a copy
                n = static_cast<_Ty*>(std::malloc((ncap ? ncap : 1) * sizeof(_Ty)));
                if (size && old) std::memcpy((void*)n, (const void*)old, sizeof(_Ty) * size);
            } else {
                aq::dev::Runtime::get().forget(old);
                n = static_cast<_Ty*>(std::realloc((void*)old, (ncap ? ncap : 1) * sizeof(_Ty)));
            }
            if (ncap > size) std::memset((void*)(n + size), 0, sizeof(_Ty) * (ncap - size));   // (before `size` moves: resize()'s new elements are zero --
            if constexpr (_resize) size = sz;                                                   //  a column of vectors assigns INTO them, and assignment drops what was there)
            container = n;
            capacity = ncap;
        } else if constexpr (_resize) size = sz;
    }
    inline void resize(const uint32_t sz) { grow<false, true>(sz); }
    inline void reserve(const uint32_t sz) { grow<false>(sz); }
    inline void emplace_back(const _Ty& v) { grow(); host()[size++] = v; }
    inline void emplace_back(_Ty&& v) { grow(); host()[size++] = std::move(v); }
    inline void clear() { size = 0; }
    inline void qpop() { size = size ? size - 1 : size; }
    inline _Ty pop() { return host()[--size]; }
    inline void shrink_to_fit() {
        if (size && capacity != size && capacity > 0) {
            _Ty* n = static_cast<_Ty*>(std::malloc(sizeof(_Ty) * size));
            std::memcpy((void*)n, (const void*)host(), sizeof(_Ty) * size);
            aq::dev::Runtime::get().forget(container);
            std::free(container);
            container = n;
            capacity = size;
        }
    }
    iterator_t erase(iterator_t it) {
        host();
        for (iterator_t c = it + 1, e = container + size; c < e; ++c) *(c - 1) = *c;
        --size;
        return it;
    }
    void merge(vector_type<_Ty>& o) {
        uint32_t total = size + o.size;
        if (capacity < total) { reserve(total); }
        std::memcpy((void*)(host() + size), (const void*)o.host(), sizeof(_Ty) * o.size);
        size = total;
    }

    // ---- views and copies --------------------------------------------------------------------------------
    inline vector_type<_Ty> subvec(uint32_t start, uint32_t end) const { return vector_type<_Ty>(container + start, end - start); }
    vector_type<_Ty> subvec_memcpy(uint32_t start, uint32_t end) const {
        vector_type<_Ty> r(end - start);
        std::memcpy((void*)r.container, (const void*)(host() + start), sizeof(_Ty) * (end - start));
        return r;
    }
    vector_type<_Ty> subvec_deep(uint32_t start, uint32_t end) const { return subvec_memcpy(start, end); }
    inline vector_type<_Ty> subvec(uint32_t start = 0) { return subvec(start, size); }
    inline vector_type<_Ty> subvec_memcpy(uint32_t start = 0) const { return subvec_memcpy(start, size); }
    inline vector_type<_Ty> subvec_deep(uint32_t start = 0) const { return subvec_deep(start, size); }
    vector_type<_Ty> getRef() { return vector_type<_Ty>(container, size); }

    inline std::unordered_set<value_t> distinct_common() { return std::unordered_set<value_t>(begin(), end()); }
    uint32_t distinct_size() { return (uint32_t)distinct_common().size(); }
    vector_type<_Ty> distinct_copy() {
        auto d = distinct_common();
        vector_type<_Ty> r((uint32_t)d.size());
        uint32_t i = 0;
        for (const auto& v : d) r.container[i++] = v;
        return r;
    }
    vector_type<_Ty>& distinct_inplace() {
        auto d = distinct_common();
        uint32_t i = 0;
        for (const auto& v : d) container[i++] = v;
        size = i;
        return *this;
    }
    vector_type<_Ty> distinct() { if (capacity) return distinct_inplace(); return distinct_copy(); }

    inline void out(uint32_t n = 4000, const char* sep = " ") const {
        const char* more = "";
        if (n < size) more = " ... "; else n = size;
        std::cout << '(';
        for (uint32_t i = 0; i < n; ++i) { std::cout << this->operator[](i); if (i + 1 < n) std::cout << sep; }
        std::cout << more << ')';
    }

    // ---- member arithmetic: result element = Coercion<_Ty, T> (reference :387-430) ------------------------------
#define AQ_MEMBER_OP(sym, name, code)                                                                          \
    template <typename T> vector_type<typename types::Coercion<_Ty, T>::type> name(const vector_type<T>& r) const { \
        vector_type<typename types::Coercion<_Ty, T>::type> ret(size);                                       \
        aq::device_binary(code, *this, r, ret);                                                              \
        return ret;                                                                                          \
    }                                                                                                        \
    template <typename T> vector_type<typename types::Coercion<_Ty, T>::type> operator sym(const vector_type<T>& r) const { return name(r); }
    AQ_MEMBER_OP(+, add, AQG_OP_ADD)
    AQ_MEMBER_OP(-, minus, AQG_OP_SUB)
    AQ_MEMBER_OP(*, multi, AQG_OP_MUL)
    AQ_MEMBER_OP(/, div, AQG_OP_DIV)
    AQ_MEMBER_OP(%, mod, AQG_OP_MOD)
#undef AQ_MEMBER_OP
    // ---- member comparisons against a scalar -> vector_type<bool> ----------------------------------------------
#define AQ_MEMBER_CMP(sym, code)                                                       \
    template <typename T, std::enable_if_t<std::is_arithmetic_v<T>>* = nullptr>        \
    inline vector_type<bool> operator sym(const T& v) const {                          \
        vector_type<bool> ret(size);                                                   \
        aq::device_binary(code, *this, v, ret);                                        \
        return ret;                                                                    \
    }
    AQ_MEMBER_CMP(>, AQG_OP_GT)
    AQ_MEMBER_CMP(<, AQG_OP_LT)
    AQ_MEMBER_CMP(>=, AQG_OP_GE)
    AQ_MEMBER_CMP(<=, AQG_OP_LE)
    AQ_MEMBER_CMP(==, AQG_OP_EQ)
    AQ_MEMBER_CMP(!=, AQG_OP_NE)
#undef AQ_MEMBER_CMP

private:
    void drop() {
        if (capacity > 0 && container) { aq::dev::Runtime::get().forget(container); std::free(container); }
        container = nullptr;
        size = capacity = 0;
    }
    void copy_from(const vector_type<_Ty>& o) {
        size = o.size;
        capacity = o.capacity;
        if (capacity) {
            container = static_cast<_Ty*>(std::malloc((size ? size : 1) * sizeof(_Ty)));
            std::memcpy((void*)container, (const void*)o.host(), sizeof(_Ty) * size);
            capacity = size ? size : 1;
        } else container = o.container;
    }
};

template <>
class vector_type<void> {
public:
    void* container;
    uint32_t size, capacity;
    typedef void* iterator_t;
    vector_type(uint32_t n) : container(std::malloc(n ? n : 1)), size(n), capacity(n) {}
    constexpr vector_type() : container(nullptr), size(0), capacity(0) {}
};
#pragma pack(pop)

template <class T> struct vector_type_std : vector_type<T> {
    vector_type_std() = default;
    vector_type_std(vector_type<T> v) : vector_type<T>(v) {}
    uint32_t size() const { return vector_type<T>::size; }
};

// ---- device plumbing shared by vector_type / table.h / aggregations.h ------------------------------------------
namespace aq {

template <class V> struct elem_of;
template <template <class> class VT, class T> struct elem_of<VT<T>> { using type = T; };

// is X a column (has container/size) or a scalar?
template <class X, class = void> struct is_column : std::false_type {};
template <class X> struct is_column<X, std::void_t<decltype(std::declval<const X&>().container), decltype(std::declval<const X&>().size)>> : std::true_type {};

template <class RT, class L, class R>
inline void device_binary(int op, const L& l, const R& r, vector_type<RT>& ret) {
    using namespace aq::dev;
    Runtime& rt = Runtime::get();
    const uint32_t n = ret.size;
    constexpr int ot = tag_of<RT>::value;
    static_assert(ot != AQG_ERROR, "result element type is not a device dtype");
    if (n == 0) return;
    // operands that are per-group temporaries of the generated loop (engine/ast.py:749-784 rewrites every column to `col[val]`): the operator
    // runs ONCE over the whole columns -- gather commutes with it, f(a[val], b[val]) = f(a, b)[val] -- and `ret` is this group's slice
    if constexpr (is_column<L>::value && is_column<R>::value) {
        using TL = std::remove_cv_t<std::remove_pointer_t<decltype(l.container)>>;
        using TR = std::remove_cv_t<std::remove_pointer_t<decltype(r.container)>>;
        Entry *el = rt.deferred_at(l.container), *er = rt.deferred_at(r.container);
        if (el && er && el->dgroup == er->dgroup && el->dg == er->dg && l.size == r.size &&
            el->dgroup->vcols[el->dv].tag == tag_of<TL>::value && er->dgroup->vcols[er->dv].tag == tag_of<TR>::value) {
            GroupCtx* gc = el->dgroup;
            const uint32_t g = el->dg;
            const int v = rt.vcol_ewise(gc, op, el->dv, er->dv, ot);
            rt.defer_slice(ret.container, (size_t)n * sizeof(RT), gc, g, v);
            if (ret.capacity == 0 && GC::scratch_space == nullptr) rt.touch(ret.container);
            return;
        }
    } else {
        constexpr bool lcol = is_column<L>::value;
        const void* cp;
        if constexpr (lcol) cp = l.container; else cp = r.container;
        if (Entry* e = rt.deferred_at(cp)) {
            GroupCtx* gc = e->dgroup;
            const uint32_t g = e->dg;
            int v;
            if constexpr (lcol) { using TS = std::remove_cv_t<R>; TS sc = r; v = rt.vcol_ewise_scalar(gc, op, AQG_VEC_SCALAR, e->dv, tag_of<TS>::value, &sc, sizeof(TS), ot); }
            else { using TS = std::remove_cv_t<L>; TS sc = l; v = rt.vcol_ewise_scalar(gc, op, AQG_SCALAR_VEC, e->dv, tag_of<TS>::value, &sc, sizeof(TS), ot); }
            if (v >= 0) {
                rt.defer_slice(ret.container, (size_t)n * sizeof(RT), gc, g, v);
                if (ret.capacity == 0 && GC::scratch_space == nullptr) rt.touch(ret.container);
                return;
            }
        }
    }
    void* dout = rt.result(ret.container, (size_t)n * sizeof(RT));
    if constexpr (is_column<L>::value && is_column<R>::value) {
        using TL = std::remove_cv_t<std::remove_pointer_t<decltype(l.container)>>;
        using TR = std::remove_cv_t<std::remove_pointer_t<decltype(r.container)>>;
        In a(l.container, (size_t)l.size * sizeof(TL), l.capacity == 0), b(r.container, (size_t)r.size * sizeof(TR), r.capacity == 0);
        check(aqg_ewise(rt.ctx(), op, AQG_VEC_VEC, tag_of<TL>::value, a.d, tag_of<TR>::value, b.d, ot, dout, n), "aqg_ewise");
    } else if constexpr (is_column<L>::value) {
        using TL = std::remove_cv_t<std::remove_pointer_t<decltype(l.container)>>;
        using TR = std::remove_cv_t<R>;
        In a(l.container, (size_t)l.size * sizeof(TL), l.capacity == 0);
        TR s = r;
        check(aqg_ewise(rt.ctx(), op, AQG_VEC_SCALAR, tag_of<TL>::value, a.d, tag_of<TR>::value, &s, ot, dout, n), "aqg_ewise");
    } else {
        using TL = std::remove_cv_t<L>;
        using TR = std::remove_cv_t<std::remove_pointer_t<decltype(r.container)>>;
        In b(r.container, (size_t)r.size * sizeof(TR), r.capacity == 0);
        TL s = l;
        check(aqg_ewise(rt.ctx(), op, AQG_SCALAR_VEC, tag_of<TL>::value, &s, tag_of<TR>::value, b.d, ot, dout, n), "aqg_ewise");
    }
    // out-parameter results that alias caller-owned host memory (init_from views) must be visible to plain
    // pointer reads too: vectors that do not own their buffer are downloaded right away
    if (ret.capacity == 0 && GC::scratch_space == nullptr) rt.touch(ret.container);
}

} // namespace aq
