// table.h -- ColRef<T>, TableInfo<Ts...> and the free element-wise operators of the AQuery library API
// (reference server/table.h:60-258 ColRef, :380-617 TableInfo, :820-937 operators, :939-973 aqop_*).
// Clean-room; operators dispatch to the HIP kernels through the C-ABI (aqg_ewise), result element types
// follow the reference: `+ -` Coercion, `*` GetLongType, `/` GetFPType, comparisons bool.
#pragma once
#include <algorithm>
#include <map>
#include <memory>
#include <unordered_set>
#include <cstdarg>
#include <cstdio>
#include <iostream>
#include <string>
#include <tuple>
#include <vector>

#include "types.h"
#include "vector_type.hpp"

struct ColRef_cstorage {
    void* container;
    unsigned int size, capacity;
    const char* name;
    int ty;
};

#ifdef __AQ__HAS__INT128__
inline std::ostream& operator<<(std::ostream& os, __int128 v) {
    if (v == 0) return os << '0';
    bool neg = v < 0;
    unsigned __int128 u = neg ? (unsigned __int128)(-(v + 1)) + 1 : (unsigned __int128)v;
    char buf[48]; int i = 47; buf[i] = 0;
    while (u) { buf[--i] = char('0' + (int)(u % 10)); u /= 10; }
    if (neg) buf[--i] = '-';
    return os << (buf + i);
}
inline std::ostream& operator<<(std::ostream& os, unsigned __int128 u) {
    if (u == 0) return os << '0';
    char buf[48]; int i = 47; buf[i] = 0;
    while (u) { buf[--i] = char('0' + (int)(u % 10)); u /= 10; }
    return os << (buf + i);
}
#endif

template <class T> struct TableStats {
    T minima{};
    unsigned char bits = 255;
};

template <typename _Ty>
class ColRef : public vector_type<_Ty> {
public:
    typedef ColRef<_Ty> Decayed_t;
    const char* name = "";
    types::Type_t ty = types::Type_t::ERROR;
    TableStats<_Ty> stats;

    ColRef() : vector_type<_Ty>(), name("") {}
    ColRef(const uint32_t& n, const char* nm = "") : vector_type<_Ty>(n), name(nm) {}
    ColRef(const char* nm) : name(nm) {}
    ColRef(const uint32_t n, void* data, const char* nm = "") : vector_type<_Ty>(n, data), name(nm) {}
    ColRef(const char* nm, types::Type_t t) : name(nm), ty(t) {}
    ColRef(const ColRef<_Ty>& o) : vector_type<_Ty>(static_cast<const vector_type<_Ty>&>(o)), name(o.name), ty(o.ty) {}
    ColRef(ColRef<_Ty>&& o) : vector_type<_Ty>(std::move(static_cast<vector_type<_Ty>&>(o))), name(o.name), ty(o.ty) {}

    void init(const char* nm = "") { ty = types::Types<_Ty>::getType(); this->size = this->capacity = 0; this->container = nullptr; name = nm; }
    // bind a borrowed buffer (zero-copy view of the data source)
    void initfrom(uint32_t sz, void* data, const char* nm = "") { ty = types::Types<_Ty>::getType(); this->size = sz; this->capacity = 0; this->container = (_Ty*)data; name = nm; }
    // take over the buffer of a temporary (the usual `out->get_col<k>().initfrom(<expr>, "name")`)
    template <template <typename> class VT, typename T> void initfrom(VT<T>&& v, const char* nm = "") {
        ty = types::Types<_Ty>::getType();
        this->size = v.size; this->capacity = v.capacity; this->container = (_Ty*)v.container; name = nm;
        v.capacity = 0;
    }
    template <template <typename> class VT, typename T> void initfrom(VT<T>& v, const char* nm = "") {
        ty = types::Types<_Ty>::getType();
        this->size = v.size; this->capacity = 0; this->container = (_Ty*)v.container; name = nm;
    }
    template <template <typename> class VT, typename T> void initfrom(const VT<T>& v, const char* nm = "") {
        ty = types::Types<_Ty>::getType();
        this->size = v.size; this->capacity = 0; this->container = (_Ty*)v.container; name = nm;
    }
    void initfrom(vectortype_cstorage v, const char* nm = "") { ty = types::Types<_Ty>::getType(); this->size = v.size; this->capacity = v.capacity; this->container = (_Ty*)v.container; name = nm; }
    // a scalar result becomes a one-row column
    template <typename T, std::enable_if_t<!aq::is_column<T>::value>* = nullptr> void initfrom(const T& v, const char* nm = "") {
        ty = types::Types<_Ty>::getType();
        this->size = 0; this->capacity = 0; this->container = nullptr;
        this->emplace_back((_Ty)v);
        name = nm;
    }
    template <class T> ColRef<_Ty>& operator=(ColRef<T>&& o) { this->container = (_Ty*)o.container; this->size = o.size; this->capacity = o.capacity; o.capacity = 0; return *this; }
    ColRef<_Ty>& operator=(const _Ty& v) { vector_type<_Ty>::operator=(v); return *this; }
    ColRef<_Ty>& operator=(const ColRef<_Ty>& o) { vector_type<_Ty>::operator=(static_cast<const vector_type<_Ty>&>(o)); return *this; }
    ColRef<_Ty>& operator=(ColRef<_Ty>&& o) noexcept { vector_type<_Ty>::operator=(std::move(static_cast<vector_type<_Ty>&>(o))); return *this; }

    using vector_type<_Ty>::operator[];
    using vector_type<_Ty>::subvec;
    using vector_type<_Ty>::subvec_memcpy;
    using vector_type<_Ty>::subvec_deep;

    // gather by row ids: ret[i] = col[idx[i]] (reference :184-189) -- one HIP gather
    vector_type<_Ty> operator[](const vector_type<uint32_t>& idxs) const {
        vector_type<_Ty> ret(idxs.size);
        if (idxs.size == 0) return ret;
        if constexpr (aq::dev::on_device<_Ty>) {
            auto& rt = aq::dev::Runtime::get();
            // `col[vecs[g]]` of a device grouping: defer -- a following sum/avg/min/max/... is answered for ALL groups by one
            // kernel (the generated per-group loop, engine/ast.py:722-789, then costs one launch per aggregate, not per group)
            aq::dev::GroupCtx* gc = nullptr;
            uint32_t g = 0;
            if (rt.group_of(idxs.container, idxs.size, &gc, &g)) {
                rt.defer_gather(ret.container, (size_t)idxs.size * sizeof(_Ty), gc, g, this->container, (size_t)this->size * sizeof(_Ty),
                                this->capacity == 0, aq::dev::tag_of<_Ty>::value);
                return ret;
            }
            void* dout = rt.result(ret.container, (size_t)idxs.size * sizeof(_Ty));
            // the column's extent is unknown for `ColRef<T>(0, ptr)`-style views: the device copy must already exist or be borrowed
            aq::dev::In col(this->container, (size_t)this->size * sizeof(_Ty), this->capacity == 0);
            aq::dev::In ids(idxs.container, (size_t)idxs.size * 4, idxs.capacity == 0);
            aq::dev::check(aqg_gather(rt.ctx(), aq::dev::tag_of<_Ty>::value, col.d, static_cast<const uint32_t*>(ids.d), idxs.size, dout), "aqg_gather");
        } else {
            for (uint32_t i = 0; i < idxs.size; ++i) ret.container[i] = this->container[idxs[i]];
        }
        return ret;
    }
    // boolean mask filter as a true stream compaction (the reference prepends `size` junk slots: defect D11)
    vector_type<_Ty> operator[](const std::vector<bool>& mask) const {
        std::vector<uint8_t> m(mask.size());
        for (size_t i = 0; i < mask.size(); ++i) m[i] = mask[i];
        return filter(m.data(), (uint32_t)m.size());
    }
    vector_type<_Ty> operator[](const vector_type<bool>& mask) const {
        vector_type<_Ty> ret(this->size);
        uint32_t m = 0;
        if (this->size) {
            auto& rt = aq::dev::Runtime::get();
            void* dout = rt.result(ret.container, (size_t)this->size * sizeof(_Ty));
            aq::dev::In col(this->container, (size_t)this->size * sizeof(_Ty), this->capacity == 0);
            aq::dev::In mk(mask.container, (size_t)mask.size, mask.capacity == 0);
            aq::dev::check(aqg_compact(rt.ctx(), aq::dev::tag_of<_Ty>::value, col.d, static_cast<const uint8_t*>(mk.d), this->size, dout, &m), "aqg_compact");
        }
        ret.size = m;
        return ret;
    }
    vector_type<_Ty> filter(const uint8_t* mask, uint32_t n) const {
        vector_type<bool> mk((bool*)mask, n);
        return (*this)[mk];
    }

    // min / width statistics used by the perfect-hash plan of the reference (:76-91); computed on the device
    bool populate_stats() {
        if constexpr (std::is_integral_v<_Ty>) {
            if (stats.bits <= 128) return true;
            if (this->size == 0) { stats.minima = 0; stats.bits = 0; return true; }
            auto& rt = aq::dev::Runtime::get();
            aq::dev::In col(this->container, (size_t)this->size * sizeof(_Ty), this->capacity == 0);
            unsigned char lo[16], hi[16];
            aq::dev::check(aqg_reduce(rt.ctx(), AQG_RED_MIN, aq::dev::tag_of<_Ty>::value, col.d, this->size, lo), "aqg_reduce");
            aq::dev::check(aqg_reduce(rt.ctx(), AQG_RED_MAX, aq::dev::tag_of<_Ty>::value, col.d, this->size, hi), "aqg_reduce");
            _Ty mn, mx;
            std::memcpy(&mn, lo, sizeof(_Ty)); std::memcpy(&mx, hi, sizeof(_Ty));
            stats.minima = mn;
            unsigned long long range = (unsigned long long)(mx - mn);
            unsigned char b = 0;
            while (b < 64 && (1ull << b) < range + 1) ++b;
            stats.bits = b;
            return true;
        }
        return false;
    }

    void out(uint32_t n = 1000, const char* sep = " ") const { vector_type<_Ty>::out(n, sep); }
    ColRef<_Ty>* rename(const char* nm) { name = nm; return this; }
    template <typename T> ColRef<T> scast() { this->ty = types::Types<T>::getType(); return *(ColRef<T>*)this; }
    ColRef_cstorage s() { ColRef_cstorage c{this->container, this->size, this->capacity, name, (int)ty}; return c; }
};
template <> class ColRef<void> : public ColRef<int> {};

// ColView: a column seen through a row-id list (reference :262-343).  Element access goes through the host copies; converting
// to a ColRef materialises the view with one device gather.
template <typename _Ty>
class ColView : public vector_base<_Ty> {
public:
    typedef ColRef<_Ty> Decayed_t;
    const uint32_t size;
    const ColRef<_Ty>& orig;
    vector_type<uint32_t> idxs;
    ColView(const ColRef<_Ty>& o, vector_type<uint32_t>&& ix) : size(ix.size), orig(o), idxs(std::move(ix)) {}
    ColView(const ColRef<_Ty>& o, const vector_type<uint32_t>& ix) : size(ix.size), orig(o), idxs(ix) {}
    ColView(const ColView<_Ty>& v, const vector_type<uint32_t>& ix) : size(ix.size), orig(v.orig), idxs(ix) {
        for (uint32_t i = 0; i < size; ++i) idxs[i] = v.idxs[ix[i]];
    }
    _Ty& operator[](const uint32_t& i) const { return orig[idxs[i]]; }
    operator ColRef<_Ty>() { ColRef<_Ty> ret; static_cast<vector_type<_Ty>&>(ret) = orig[idxs]; return ret; }
    ColView<_Ty> subvec(uint32_t start, uint32_t end) const { return ColView<_Ty>(orig, idxs.subvec(start, end)); }
    ColRef<_Ty> subvec_deep(uint32_t start, uint32_t end) const {
        ColRef<_Ty> sub(end - start);
        for (uint32_t i = 0; i < end - start; ++i) sub[i] = (*this)[start + i];
        return sub;
    }
    std::unordered_set<_Ty> distinct_common() const { std::unordered_set<_Ty> s; for (uint32_t i = 0; i < size; ++i) s.insert((*this)[i]); return s; }
    uint32_t distinct_size() const { return (uint32_t)distinct_common().size(); }
    void out(uint32_t n = 1000, const char* sep = " ") const {
        n = n > size ? size : n;
        std::cout << '(';
        for (uint32_t i = 0; i < n; ++i) std::cout << (*this)[i] << sep;
        std::cout << ')';
    }
};
template <class V> struct is_vector_impl<ColView<V>> : std::true_type {};
using uColRef = ColRef<void>;

template <class V> struct is_vector_impl<ColRef<V>> : std::true_type {};
template <class V> struct is_vector_impl<vector_type<V>> : std::true_type {};
template <class T> constexpr static bool is_vector(const ColRef<T>&) { return true; }
template <class T> constexpr static bool is_vector(const vector_type<T>&) { return true; }

template <template <class...> class VT, class T, std::enable_if_t<std::is_base_of_v<vector_base<T>, VT<T>>>* = nullptr>
std::ostream& operator<<(std::ostream& os, const VT<T>& v) { v.out(); return os; }

// ---- output table ---------------------------------------------------------------------------------------------
// Columns are stored as a tuple of correctly typed ColRefs (the reference reinterprets a ColRef<void>[] with a
// 40-byte stride, which overlaps 8-byte columns: defect D10).
template <class... Types> struct TableInfo;

template <long long _Index, class... _Types> constexpr auto& get(const TableInfo<_Types...>& table) noexcept {
    return const_cast<TableInfo<_Types...>&>(table).template get_col<(size_t)_Index>();
}

namespace aq {
inline void append_value(std::string& s, const char* v) { s += v ? v : ""; }
inline void append_value(std::string& s, astring_view v) { s += v.rstr ? v.rstr : ""; }
inline void append_value(std::string& s, bool v) { s += v ? "true" : "false"; }
// %f of a large double runs to 300+ digits (1e300): size the buffer from the first call instead of cutting the text
template <class V> inline void append_printf(std::string& s, const char* fmt, V v) {
    char b[64];
    const int len = std::snprintf(b, sizeof b, fmt, v);
    if (len < (int)sizeof b) { s += b; return; }
    std::string big((size_t)len + 1, '\0');
    std::snprintf(big.data(), big.size(), fmt, v);
    big.resize((size_t)len);
    s += big;
}
inline void append_value(std::string& s, float v) { append_printf(s, "%f", (double)v); }
inline void append_value(std::string& s, double v) { append_printf(s, "%lf", v); }
inline void append_value(std::string& s, long double v) { append_printf(s, "%Lf", v); }
#ifdef __AQ__HAS__INT128__
inline void append_value(std::string& s, __int128 v) {
    if (v == 0) { s += '0'; return; }
    bool neg = v < 0;
    unsigned __int128 u = neg ? (unsigned __int128)(-(v + 1)) + 1 : (unsigned __int128)v;
    char buf[48]; int i = 47; buf[i] = 0;
    while (u) { buf[--i] = char('0' + (int)(u % 10)); u /= 10; }
    if (neg) buf[--i] = '-';
    s += buf + i;
}
inline void append_value(std::string& s, unsigned __int128 u) {
    if (u == 0) { s += '0'; return; }
    char buf[48]; int i = 47; buf[i] = 0;
    while (u) { buf[--i] = char('0' + (int)(u % 10)); u /= 10; }
    s += buf + i;
}
#endif
template <class T, std::enable_if_t<std::is_integral_v<T> && !std::is_same_v<T, bool>>* = nullptr>
inline void append_value(std::string& s, T v) { s += std::to_string(v); }
} // namespace aq

template <class... Types>
struct TableInfo {
    const char* name;
    uint32_t n_cols;
    typedef std::tuple<Types...> tuple_type;
    std::tuple<ColRef<Types>...> cols;

    TableInfo(const char* nm, uint32_t) : name(nm), n_cols(sizeof...(Types)) { init_names(nullptr, std::index_sequence_for<Types...>{}); }
    TableInfo(const char* nm = "", const char** col_names = nullptr) : name(nm), n_cols(sizeof...(Types)) { init_names(col_names, std::index_sequence_for<Types...>{}); }

    template <size_t i = 0> auto& get_col() { return std::get<i>(cols); }
    // write-back into the data source (reference server/table_ext_monetdb.hpp:34-87; emitted for INSERT INTO ... SELECT and SELECT ... INTO,
    // engine/ast.py:507,1494): CREATE TABLE IF NOT EXISTS through the source's SQL door, then the columns by pointer.  Result columns still
    // in HBM come down together: asynchronous egress of all of them, one wait (device.h fetch_all).  Defined in libaquery.h (needs DataSource).
    void monetdb_append_table(void* srv, const char* alt_name = nullptr);
    TableInfo<Types...>* rename(const char* nm) { name = nm; return this; }
    uint32_t rows() const { return std::get<0>(cols).size; }

    // ---- host utilities of the reference (:429-461, :601-614) ------------------------------------------------------------
    // materialize: every column gathered by the same row ids (one device gather per column)
    template <int prog = 0>
    inline void materialize(const vector_type<uint32_t>& idxs, TableInfo<Types...>* tbl = nullptr) {
        if constexpr (prog == 0) tbl = (tbl == nullptr ? this : tbl);
        if constexpr (prog == sizeof...(Types)) return;
        else {
            auto& col = std::get<prog>(cols);
            auto gathered = col[idxs];
            const char* nm = col.name;
            static_cast<std::remove_reference_t<decltype(gathered)>&>(std::get<prog>(tbl->cols)) = std::move(gathered);
            std::get<prog>(tbl->cols).name = nm;
            materialize<prog + 1>(idxs, tbl);
        }
    }
    inline TableInfo<Types...>* materialize_copy(const vector_type<uint32_t>& idxs) {
        auto tbl = new TableInfo<Types...>(this->name, (uint32_t)sizeof...(Types));
        materialize<0>(idxs, tbl);
        return tbl;
    }
    // order_by<c0, c1, ...>: row ids sorted by the listed columns (column c >= 0 ascending; -1 - c descending), std::sort on
    // the host like the reference (not stable, ties in unspecified order)
    template <int... ocols>
    inline vector_type<uint32_t>* order_by(vector_type<uint32_t>* ord = nullptr) {
        const uint32_t n = rows();
        if (!ord) {
            ord = new vector_type<uint32_t>(n);
            for (uint32_t i = 0; i < n; ++i) (*ord)[i] = i;
        }
        std::sort(ord->begin(), ord->end(), [this](const uint32_t& l, const uint32_t& r) {
            return std::make_tuple(order_key<ocols>(l)...) < std::make_tuple(order_key<ocols>(r)...);
        });
        return ord;
    }
    template <int... ocols> auto order_by_view();
    // distinct rows, in place.  The reference iterates an unordered_set (order unspecified); here: first-occurrence order --
    // one device group-by over all columns when they are all integral, a host pass otherwise.
    TableInfo<Types...>* distinct() {
        const uint32_t n = rows();
        if (n == 0) return this;
        vector_type<uint32_t> keep(0u);
        if constexpr ((std::is_integral_v<Types> && ...) && (aq::dev::on_device<Types> && ...)) {
            auto& rt = aq::dev::Runtime::get();
            int dts[sizeof...(Types)];
            const void* ptrs[sizeof...(Types)];
            std::vector<std::unique_ptr<aq::dev::In>> ins;
            distinct_bind(dts, ptrs, ins, std::index_sequence_for<Types...>{});
            aqg_groupby* h = nullptr;
            aq::dev::check(aqg_groupby_build(rt.ctx(), (int)sizeof...(Types), dts, ptrs, n, 0, &h), "aqg_groupby_build");
            const uint32_t G = aqg_groupby_ngroups(h);
            keep = vector_type<uint32_t>(G);
            aq::dev::check(aqg_d2h(rt.ctx(), keep.container, aqg_groupby_first_rows(h), (size_t)G * 4), "aqg_d2h");
            aq::dev::check(aqg_sync(rt.ctx()), "aqg_sync");
            aqg_groupby_destroy(h);
        } else {
            std::map<tuple_type, uint32_t> seen;
            std::vector<uint32_t> firsts;
            for (uint32_t i = 0; i < n; ++i)
                if (seen.emplace(row_tuple(i, std::index_sequence_for<Types...>{}), i).second) firsts.push_back(i);
            keep = vector_type<uint32_t>((uint32_t)firsts.size());
            for (uint32_t i = 0; i < keep.size; ++i) keep[i] = firsts[i];
        }
        materialize<0>(keep);
        return this;
    }

    // every name followed by sep '|' sep, then strlen(sep) + 1 characters trimmed -- which leaves the LAST separator standing
    // ("Mont | avgw3ysales " for sep " ", "Mont,|,avgw3ysales," for sep ","): the reference's text, byte for byte (table.h:484-493)
    std::string get_header_string(const char* __restrict sep, const char* __restrict end) const {
        std::string h;
        header_names(h, sep, std::index_sequence_for<Types...>{});
        if (const size_t l_sep = std::strlen(sep) + 1; h.size() >= l_sep) h.resize(h.size() - l_sep);
        std::string line(h.size(), '=');
        return h + end + line + end;
    }
    // `print(*tbl)`: header, rule, one row per line through operator<<
    void print(const char* __restrict sep, const char* __restrict end) const {
        std::cout << get_header_string(sep, end);
        const uint32_t n = n_cols ? rows() : 0;
        for (uint32_t i = 0; i < n; ++i) { print_row(i, sep, std::index_sequence_for<Types...>{}); std::cout << end; }
    }
    // `printall(sep, end, view, fp, limit)`: printf-style values; vector-valued columns are flattened (one output row per
    // element, the other columns repeated), as the reference's print2 does
    void printall(const char* __restrict sep = ",", const char* __restrict end = "\n", const vector_type<uint32_t>* __restrict view = nullptr,
                  FILE* __restrict fp = nullptr, uint32_t limit = std::numeric_limits<uint32_t>::max()) const {
        FILE* o = fp ? fp : stdout;
        std::string h;
        if (fp) { csv_names(h, sep, std::index_sequence_for<Types...>{}); }
        else { h = get_header_string(sep, end); h.resize(h.size() - std::strlen(end)); }
        std::fprintf(o, "%s%s", h.c_str(), end);
        const uint32_t total = view ? view->size : rows();
        const uint32_t n = limit > total ? total : limit;
        for (uint32_t r = 0; r < n; ++r) {
            std::vector<std::string> fields;
            flatten<0>(view ? (*view)[r] : r, fields, o, sep, end);
        }
    }

private:
    template <int c> auto order_key(uint32_t i) {
        if constexpr (c >= 0) return std::get<(size_t)c>(cols)[i];
        else return -std::get<(size_t)(-1 - c)>(cols)[i];
    }
    template <size_t... Is> tuple_type row_tuple(uint32_t i, std::index_sequence<Is...>) { return tuple_type(std::get<Is>(cols)[i]...); }
    template <size_t... Is> void distinct_bind(int* dts, const void** ptrs, std::vector<std::unique_ptr<aq::dev::In>>& ins, std::index_sequence<Is...>) {
        ((dts[Is] = aq::dev::tag_of<std::tuple_element_t<Is, tuple_type>>::value,
          ins.push_back(std::make_unique<aq::dev::In>(std::get<Is>(cols).container, (size_t)std::get<Is>(cols).size * sizeof(std::tuple_element_t<Is, tuple_type>),
                                                      std::get<Is>(cols).capacity == 0)),
          ptrs[Is] = ins.back()->d), ...);
    }
    template <size_t... Is> void init_names(const char** names, std::index_sequence<Is...>) { (std::get<Is>(cols).init(names ? names[Is] : ""), ...); }
    template <size_t... Is> void header_names(std::string& h, const char* sep, std::index_sequence<Is...>) const {
        ((h += std::string(std::get<Is>(cols).name) + sep + '|' + sep), ...);
    }
    template <size_t... Is> void csv_names(std::string& h, const char* sep, std::index_sequence<Is...>) const {
        size_t k = 0;
        ((h += std::string(std::get<Is>(cols).name) + (++k < sizeof...(Is) ? std::string(sep) : std::string())), ...);
    }
    template <size_t... Is> void print_row(uint32_t i, const char* sep, std::index_sequence<Is...>) const {
        size_t k = 0;
        ((std::cout << std::get<Is>(cols)[i], (++k < sizeof...(Is) ? (void)(std::cout << sep) : (void)0)), ...);
    }
    template <size_t j> void flatten(uint32_t row, std::vector<std::string>& fields, FILE* o, const char* sep, const char* end) const {
        if constexpr (j == sizeof...(Types)) {
            std::string line;
            for (size_t k = 0; k < fields.size(); ++k) { line += fields[k]; if (k + 1 < fields.size()) line += sep; }
            std::fprintf(o, "%s%s", line.c_str(), end);
        } else {
            using CT = std::tuple_element_t<j, tuple_type>;
            const auto& v = std::get<j>(cols)[row];
            if constexpr (is_vector_type<CT>) {
                for (uint32_t e = 0; e < v.size; ++e) {
                    std::string s; aq::append_value(s, v[e]);
                    fields.push_back(s);
                    flatten<j + 1>(row, fields, o, sep, end);
                    fields.pop_back();
                }
            } else {
                std::string s; aq::append_value(s, v);
                fields.push_back(s);
                flatten<j + 1>(row, fields, o, sep, end);
                fields.pop_back();
            }
        }
    }
};

// TableView: a table seen through a row-id list (reference :620-690); printing goes through `printall`'s view argument
template <class... Types>
struct TableView {
    typedef std::tuple<Types...> tuple_type;
    const vector_type<uint32_t>* idxs;
    const TableInfo<Types...>& info;
    constexpr TableView(const vector_type<uint32_t>* ix, const TableInfo<Types...>& t) noexcept : idxs(ix), info(t) {}
    void print(const char* __restrict sep, const char* __restrict end) const { info.printall(sep, end, idxs); }
    TableInfo<Types...>* materialize(const char* name = nullptr, const char** = nullptr) {
        auto t = const_cast<TableInfo<Types...>&>(info).materialize_copy(*idxs);
        if (name) t->name = name;
        return t;
    }
};
template <class... Types> template <int... ocols> auto TableInfo<Types...>::order_by_view() { return TableView<Types...>(order_by<ocols...>(), *this); }

template <class... Types> void print(const TableInfo<Types...>& v, const char* delimiter = " ", const char* endline = "\n") { v.print(delimiter, endline); }
template <class T, std::enable_if_t<!aq::is_column<T>::value>* = nullptr> void print(const T& v, const char* delimiter = " ") { std::cout << v << delimiter; }
// a column: every element through print(elem) followed by the delimiter, then the end of line
template <class T, template <typename> class VT, std::enable_if_t<aq::is_column<VT<T>>::value>* = nullptr>
void print(const VT<T>& v, const char* delimiter = " ", const char* endline = "\n") {
    for (uint32_t i = 0; i < v.size; ++i) { print(v[i]); std::cout << delimiter; }
    std::cout << endline;
}

// ---- free element-wise operators (reference :820-937) ------------------------------------------------------------------
namespace aq {
template <class T1, template <typename> class VT> constexpr bool vt_ok = std::is_base_of_v<vector_base<T1>, VT<T1>>;
template <class T1, class T2> using co_t = typename types::Coercion<T1, T2>::type;
}

#define AQ_FREE_OP(sym, code, RESULT)                                                                                           \
    template <class T1, class T2, template <typename> class VT, template <typename> class VT2,                                  \
              std::enable_if_t<aq::vt_ok<T1, VT> && aq::vt_ok<T2, VT2>>* = nullptr>                                              \
    decayed_t<VT, RESULT> operator sym(const VT<T1>& lhs, const VT2<T2>& rhs) {                                                 \
        decayed_t<VT, RESULT> ret(lhs.size);                                                                                    \
        aq::device_binary(code, lhs, rhs, static_cast<vector_type<RESULT>&>(ret));                                              \
        return ret;                                                                                                             \
    }                                                                                                                           \
    template <class T1, class T2, template <typename> class VT, std::enable_if_t<aq::vt_ok<T1, VT> && std::is_arithmetic_v<T2>>* = nullptr> \
    decayed_t<VT, RESULT> operator sym(const VT<T1>& lhs, const T2& rhs) {                                                      \
        decayed_t<VT, RESULT> ret(lhs.size);                                                                                    \
        aq::device_binary(code, lhs, rhs, static_cast<vector_type<RESULT>&>(ret));                                              \
        return ret;                                                                                                             \
    }                                                                                                                           \
    template <class T1, class T2, template <typename> class VT, std::enable_if_t<aq::vt_ok<T1, VT> && std::is_arithmetic_v<T2>>* = nullptr> \
    decayed_t<VT, RESULT> operator sym(const T2& lhs, const VT<T1>& rhs) {                                                      \
        decayed_t<VT, RESULT> ret(rhs.size);                                                                                    \
        aq::device_binary(code, lhs, rhs, static_cast<vector_type<RESULT>&>(ret));                                              \
        return ret;                                                                                                             \
    }
#define AQ_COMMA ,
AQ_FREE_OP(-, AQG_OP_SUB, aq::co_t<T1 AQ_COMMA T2>)
AQ_FREE_OP(+, AQG_OP_ADD, aq::co_t<T1 AQ_COMMA T2>)
AQ_FREE_OP(*, AQG_OP_MUL, types::GetLongType<aq::co_t<T1 AQ_COMMA T2>>)
AQ_FREE_OP(/, AQG_OP_DIV, types::GetFPType<aq::co_t<T1 AQ_COMMA T2>>)
#undef AQ_FREE_OP

// free comparisons -> VT<bool>; the reference only has `>` (:917-937), the other five are additions
#define AQ_FREE_CMP(sym, code)                                                                                                  \
    template <class T1, class T2, template <typename> class VT, template <typename> class VT2,                                  \
              std::enable_if_t<aq::vt_ok<T1, VT> && aq::vt_ok<T2, VT2>>* = nullptr>                                              \
    VT<bool> operator sym(const VT<T1>& lhs, const VT2<T2>& rhs) {                                                              \
        VT<bool> ret(lhs.size);                                                                                                 \
        aq::device_binary(code, lhs, rhs, static_cast<vector_type<bool>&>(ret));                                                \
        return ret;                                                                                                             \
    }
AQ_FREE_CMP(>, AQG_OP_GT)
AQ_FREE_CMP(<, AQG_OP_LT)
AQ_FREE_CMP(>=, AQG_OP_GE)
AQ_FREE_CMP(<=, AQG_OP_LE)
#undef AQ_FREE_CMP
#define AQ_FREE_CMP_S(sym, code, rcode)                                                                                          \
    template <class T1, class T2, template <typename> class VT, std::enable_if_t<aq::vt_ok<T1, VT> && !std::is_same_v<VT<T1>, vector_type<T1>> && std::is_arithmetic_v<T2>>* = nullptr> \
    VT<bool> operator sym(const VT<T1>& lhs, const T2& rhs) {                                                                   \
        VT<bool> ret(lhs.size);                                                                                                 \
        aq::device_binary(code, lhs, rhs, static_cast<vector_type<bool>&>(ret));                                                \
        return ret;                                                                                                             \
    }                                                                                                                           \
    template <class T1, class T2, template <typename> class VT, std::enable_if_t<aq::vt_ok<T1, VT> && std::is_arithmetic_v<T2>>* = nullptr> \
    VT<bool> operator sym(const T2& lhs, const VT<T1>& rhs) {                                                                   \
        VT<bool> ret(rhs.size);                                                                                                 \
        aq::device_binary(code, lhs, rhs, static_cast<vector_type<bool>&>(ret));                                                \
        return ret;                                                                                                             \
    }
AQ_FREE_CMP_S(>, AQG_OP_GT, AQG_OP_LT)
AQ_FREE_CMP_S(<, AQG_OP_LT, AQG_OP_GT)
AQ_FREE_CMP_S(>=, AQG_OP_GE, AQG_OP_LE)
AQ_FREE_CMP_S(<=, AQG_OP_LE, AQG_OP_GE)
#undef AQ_FREE_CMP_S

// out-parameter forms `aqop_<op>(l, r, ret)` (reference :939-973): ret[i] = l[i] OP r[i], ret keeps its own element type
#define AQ_AQOP(x, code)                                                                          \
    template <class T1, class T2, template <typename> class VT, class Ret>                        \
    void aqop_##x(const VT<T1>& lhs, const VT<T2>& rhs, Ret& ret) {                               \
        using RT = std::remove_cv_t<std::remove_pointer_t<decltype(ret.container)>>;              \
        aq::device_binary(code, lhs, rhs, static_cast<vector_type<RT>&>(ret));                    \
    }
AQ_AQOP(add, AQG_OP_ADD)
AQ_AQOP(minus, AQG_OP_SUB)
AQ_AQOP(mul, AQG_OP_MUL)
AQ_AQOP(div, AQG_OP_DIV)
AQ_AQOP(and, AQG_OP_AND)
AQ_AQOP(or, AQG_OP_OR)
AQ_AQOP(xor, AQG_OP_XOR)
AQ_AQOP(gt, AQG_OP_GT)
AQ_AQOP(lt, AQG_OP_LT)
AQ_AQOP(gte, AQG_OP_GE)
AQ_AQOP(lte, AQG_OP_LE)
AQ_AQOP(eq, AQG_OP_EQ)
AQ_AQOP(neq, AQG_OP_NE)
#undef AQ_AQOP

// ---- ABI facts shared with the reference's C views (server/vector_type.hpp:25-28, server/table.h:34-39 == sdk/aquery.h:105-128)
// and probed against the reference headers themselves: one edit of these classes must not silently break
// user modules that reinterpret a ColRef as a ColRef_storage.
#include <cstddef>
static_assert(sizeof(vectortype_cstorage) == 16 && sizeof(ColRef_cstorage) == 32, "C views of vector_type / ColRef");
static_assert(sizeof(vector_type<int>) == 16 && sizeof(vector_type<double>) == 16 && sizeof(vector_type<void>) == 16, "vector_type is the packed {container, size, capacity} triple");
static_assert(sizeof(ColRef<int>) == 40 && sizeof(ColRef<float>) == 40 && sizeof(ColRef<void>) == 40, "ColRef<4-byte T>: 40 bytes (the stride of TableInfo::colrefs)");
static_assert(sizeof(ColRef<double>) == 48 && sizeof(ColRef<long long>) == 48, "ColRef<8-byte T>: 48 bytes");
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Winvalid-offsetof"
namespace aq { namespace abi {
struct ColRefProbe : ColRef<int> { static constexpr size_t name_off() { return offsetof(ColRefProbe, name); } static constexpr size_t ty_off() { return offsetof(ColRefProbe, ty); } };
} }
static_assert(aq::abi::ColRefProbe::name_off() == 16 && aq::abi::ColRefProbe::ty_off() == 24, "ColRef: name @16, ty @24 (ColRef_cstorage / sdk ColRef_storage)");
#pragma GCC diagnostic pop
