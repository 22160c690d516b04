// hasher.h -- hash group-by front door of the AQuery library API (reference server/hasher.h).
//   hasher<Ts...>            tuple hash functor (spelled `transTypes<record_t, hasher>` by generated code)
//   AQHashTable<Key, Hash>   per-row `hashtable_push(Key&&, i)` + `ht_postproc(n)` (:146-199)
//   HashTableFactory<K,H>::get(cols...) -> HashTableComponents{size, keys, values, offsets} (:201-207,:327-357;
//                            does not compile in the reference -- defects D2/D3 -- implemented here for real)
// The table itself lives on the MI355X (aqg_groupby_build + aqg_groupby_postproc through the C-ABI).
// Contract = the reference's executable path: dense group ids in FIRST-OCCURRENCE order, row-id lists in
// DESCENDING row order inside a group, offsets[g] = start of group g in the flat row-id buffer.
// The hash function is not observable in results; the device uses its own.
#pragma once
#include <functional>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "types.h"
#include "vector_type.hpp"

template <class... Types> struct hasher {
    size_t operator()(const std::tuple<Types...>& rec) const {
        size_t h = 0x7c5f3e9a1b2d4c6bULL;
        std::apply([&](const auto&... f) { ((h = (h ^ std::hash<std::decay_t<decltype(f)>>()(f)) * 0x9E3779B97F4A7C15ULL), ...); }, rec);
        return h;
    }
};
template <typename Key, typename Val> using aq_map = std::unordered_map<Key, Val>;
template <typename Key> using aq_set = std::unordered_set<Key>;

template <class> class ColRef;

namespace aq {

using GroupTable = dev::GroupCtx;   // host-side view of one device group-by (offsets/counts host valid, row_ids filled lazily)

// dtype tag of a KEY column's element type (the reference hashes these, server/hasher.h:66-144): plain integers and bools,
// floating columns (grouped by value), raw `const char*` (a POINTER in the reference's tuple ==: an 8-byte integer),
// astring_view (string contents: dictionary codes made on the host, aqg_str_encode), dates / times / timestamps, 128-bit integers
template <class T> struct key_tag { static constexpr int value = dev::tag_of<std::remove_cv_t<T>>::value; };
template <class T> struct key_tag<T*> { static constexpr int value = AQG_UINT64; };
template <> struct key_tag<astring_view> { static constexpr int value = AQG_STR; };
template <> struct key_tag<types::date_t> { static constexpr int value = AQG_DATE; };
template <> struct key_tag<types::time_t> { static constexpr int value = AQG_TIME; };
template <> struct key_tag<types::timestamp_t> { static constexpr int value = AQG_TIMESTAMP; };

// device view of one key column for the duration of a build: uploads (or finds the mirror of) the host column; string views are
// encoded into a temporary code column
struct KeyIn {
    int dt = AQG_ERROR;
    const void* d = nullptr;
    void* temp = nullptr;       // upload owned by this view
    void* codes = nullptr;      // astring_view: the code column
    template <class T> KeyIn(const T* host, uint32_t n, bool borrowed) {
        auto& rt = dev::Runtime::get();
        if constexpr (std::is_same_v<std::remove_cv_t<T>, astring_view>) {
            static_assert(sizeof(astring_view) == sizeof(const char*), "astring_view is one pointer");
            dev::check(aqg_malloc(rt.ctx(), (size_t)(n ? n : 1) * 4, &codes), "aqg_malloc");
            dev::check(aqg_str_encode(rt.ctx(), reinterpret_cast<const char* const*>(host), n, static_cast<uint32_t*>(codes), nullptr), "aqg_str_encode");
            dt = AQG_UINT32; d = codes;
        } else {
            static_assert(key_tag<T>::value != AQG_ERROR, "element type cannot be a group-by key");
            dt = key_tag<T>::value;
            d = rt.input(host, (size_t)n * sizeof(T), borrowed, &temp);
        }
    }
    ~KeyIn() { auto& rt = dev::Runtime::get(); rt.release(temp); if (codes) aqg_free(rt.ctx(), codes); }
    KeyIn(const KeyIn&) = delete;
    KeyIn& operator=(const KeyIn&) = delete;
};

inline GroupTable& build_groups(int nkeys, const int* dts, const void* const* dev_cols, uint32_t n, bool want_reversemap, uint32_t* reversemap_host,
                                uint32_t max_groups_hint = 0) {
    auto& rt = dev::Runtime::get();
    GroupTable& t = *new GroupTable();      // owned by whoever holds the grouping (AQHashTable / the module's session)
    t.n = n;
    dev::check(aqg_groupby_build(rt.ctx(), nkeys, dts, dev_cols, n, max_groups_hint, &t.handle), "aqg_groupby_build");
    t.G = aqg_groupby_ngroups(t.handle);
    t.offsets = static_cast<uint32_t*>(std::malloc(((size_t)t.G + 1) * 4));
    t.counts = static_cast<uint32_t*>(std::malloc(((size_t)t.G + 1) * 4));
    t.row_ids = static_cast<uint32_t*>(std::malloc(((size_t)n + 1) * 4));
    // offsets = exclusive scan of the group sizes (what aqg_groupby_postproc would return); the row lists themselves are made on
    // first use (device.h fill_rows): generated code that only reduces `col[vecs[g]]` never asks for them
    if (t.G) dev::check(aqg_d2h(rt.ctx(), t.counts, aqg_groupby_counts(t.handle), (size_t)t.G * 4), "aqg_d2h");
    t.offsets[0] = 0;
    for (uint32_t g = 0; g < t.G; ++g) t.offsets[g + 1] = t.offsets[g] + t.counts[g];
    if (want_reversemap && n) dev::check(aqg_d2h(rt.ctx(), reversemap_host, aqg_groupby_reversemap(t.handle), (size_t)n * 4), "aqg_d2h");
    rt.adopt_group(&t, nullptr);  // vecs[g] views identify their group; their device copy is filled lazily
    return t;
}

template <class T> inline void fetch_key_column(const GroupTable& t, int k, std::vector<T>& out, const T* host_col = nullptr) {
    auto& rt = dev::Runtime::get();
    out.resize(t.G);
    if (!t.G) return;
    if constexpr (std::is_same_v<T, astring_view>) {       // the key of a group is the string view of its first row (the first one pushed)
        std::vector<uint32_t> first(t.G);
        dev::check(aqg_d2h(rt.ctx(), first.data(), aqg_groupby_first_rows(t.handle), (size_t)t.G * 4), "aqg_d2h");
        for (uint32_t g = 0; g < t.G; ++g) out[g] = host_col[first[g]];
        return;
    }
    void* d = nullptr;
    dev::check(aqg_malloc(rt.ctx(), (size_t)t.G * sizeof(T), &d), "aqg_malloc");
    dev::check(aqg_groupby_keys(t.handle, k, d), "aqg_groupby_keys");
    dev::check(aqg_d2h(rt.ctx(), out.data(), d, (size_t)t.G * sizeof(T)), "aqg_d2h");
    aqg_free(rt.ctx(), d);
}

template <class Tuple, size_t... Is> inline void fetch_keys(const GroupTable& t, std::vector<Tuple>& keys, std::index_sequence<Is...>, const void* const* host_cols = nullptr) {
    std::tuple<std::vector<std::tuple_element_t<Is, Tuple>>...> cols;
    (fetch_key_column(t, (int)Is, std::get<Is>(cols), host_cols ? static_cast<const std::tuple_element_t<Is, Tuple>*>(host_cols[Is]) : nullptr), ...);
    keys.resize(t.G);
    for (uint32_t g = 0; g < t.G; ++g) keys[g] = Tuple(std::get<Is>(cols)[g]...);
}

inline vector_type<uint32_t>* make_vecs(const GroupTable& t) {
    auto vecs = static_cast<vector_type<uint32_t>*>(std::malloc(sizeof(vector_type<uint32_t>) * (t.G ? t.G : 1)));
    for (uint32_t g = 0; g < t.G; ++g) vecs[g].init_from(t.counts[g], t.row_ids + t.offsets[g]);
    return vecs;
}

} // namespace aq

template <class... Ty> struct HashTableComponents {
    uint32_t size;
    std::vector<std::tuple<Ty...>>* keys;
    vector_type<uint32_t>* values;
    uint32_t* offsets;
};

template <class Key, class Hash>
class AQHashTable {
public:
    uint32_t *reversemap = nullptr, *mapbase = nullptr, *ht_base = nullptr;
    AQHashTable() = default;
    explicit AQHashTable(uint32_t sz) { init(sz); }
    void init(uint32_t sz) {
        staged_.clear();
        staged_.reserve(sz);
        cap_ = sz;
        reversemap = static_cast<uint32_t*>(std::malloc(sizeof(uint32_t) * ((size_t)sz * 2 + 2)));
        mapbase = reversemap + sz;
        ht_base = static_cast<uint32_t*>(std::calloc((size_t)sz + 1, sizeof(uint32_t)));
    }
    // the reference assigns the dense id here; the device assigns all ids at once in finish()
    inline void hashtable_push(Key&& k, uint32_t i) {
        if (staged_.size() <= i) staged_.resize((size_t)i + 1);
        staged_[i] = std::move(k);
        done_ = false;
    }
    template <typename... Keys_t> inline void hashtable_push_all(Keys_t&... keys, uint32_t len) {
        build_from_columns(len, keys...);
    }
    // counting sort of row ids by group: vecs[g] = rows of group g, DESCENDING; ht_base[g] = start offset
    vector_type<uint32_t>* ht_postproc(uint32_t sz) {
        finish(sz);
        std::memcpy(ht_base, table().offsets, (size_t)table().G * 4);
        auto vecs = aq::make_vecs(table());
        // expose the row lists through the reference's `mapbase` too (host copy)
        if (sz) { aq::dev::Runtime::get().touch(table().row_ids); std::memcpy(mapbase, table().row_ids, (size_t)sz * 4); }
        return vecs;
    }
    std::vector<Key>& values() { finish((uint32_t)staged_.size()); return keys_; }
    size_t size() { finish((uint32_t)staged_.size()); return table().G; }
    ~AQHashTable() { drop_table(); }

private:
    template <size_t... Is> void finish_staged(uint32_t n, std::index_sequence<Is...>) {
        // unzip the staged tuples into columns, upload, group on the device
        std::tuple<std::vector<std::tuple_element_t<Is, Key>>...> cols;
        ((std::get<Is>(cols).resize(n)), ...);
        for (uint32_t i = 0; i < n; ++i) ((std::get<Is>(cols)[i] = std::get<Is>(staged_[i])), ...);
        aq::KeyIn ins[] = {aq::KeyIn(std::get<Is>(cols).data(), n, false)...};
        int dts[sizeof...(Is)];
        const void* ptrs[sizeof...(Is)];
        const void* hosts[] = {static_cast<const void*>(std::get<Is>(cols).data())...};
        for (size_t k = 0; k < sizeof...(Is); ++k) { dts[k] = ins[k].dt; ptrs[k] = ins[k].d; }
        drop_table();
        tp_ = &aq::build_groups((int)sizeof...(Is), dts, ptrs, n, true, reversemap);
        aq::fetch_keys(*tp_, keys_, std::index_sequence<Is...>{}, hosts);
    }
    template <class... Cols> void build_from_columns(uint32_t n, Cols&... cols) {
        aq::KeyIn ins[] = {aq::KeyIn(cols.container, n, cols.capacity == 0)...};
        int dts[sizeof...(Cols)];
        const void* ptrs[sizeof...(Cols)];
        const void* hosts[] = {static_cast<const void*>(cols.container)...};
        for (size_t k = 0; k < sizeof...(Cols); ++k) { dts[k] = ins[k].dt; ptrs[k] = ins[k].d; }
        drop_table();
        tp_ = &aq::build_groups((int)sizeof...(Cols), dts, ptrs, n, true, reversemap);
        aq::fetch_keys(*tp_, keys_, std::make_index_sequence<std::tuple_size_v<Key>>{}, hosts);
        for (uint32_t g = 0; g < tp_->G; ++g) ht_base[g] = tp_->counts[g];
        done_ = true;
    }
    void finish(uint32_t n) {
        if (done_) return;
        finish_staged(n, std::make_index_sequence<std::tuple_size_v<Key>>{});
        for (uint32_t g = 0; g < table().G; ++g) ht_base[g] = table().counts[g];
        done_ = true;
    }
    std::vector<Key> staged_, keys_;
    aq::GroupTable* tp_ = nullptr;
    aq::GroupTable empty_;
    aq::GroupTable& table() { return tp_ ? *tp_ : empty_; }
    void drop_table() {
        if (!tp_) return;
        aq::dev::Runtime::get().free_vcols(tp_);
        if (tp_->row_ids) { aq::dev::Runtime::get().forget(tp_->row_ids); std::free(tp_->row_ids); }
        if (tp_->handle) aqg_groupby_destroy(tp_->handle);
        std::free(tp_->offsets); std::free(tp_->counts);
        delete tp_;
        tp_ = nullptr;
    }
    uint32_t cap_ = 0;
    bool done_ = false;
};

// `HashTableFactory<record_t, transTypes<record_t, hasher>>::get<decays<decltype(c)>...>(c...)` (engine/ast.py:668-670):
// the explicit template arguments are the COLUMN types, so `get` takes them as they are.
template <class Key, class Hash, int PerfectHashingThreshold = 18>
class HashTableFactory {
public:
    template <class... Cols>
    static HashTableComponents<value_type_r<std::decay_t<Cols>>...> get(Cols&... cols) {
        using Tuple = std::tuple<value_type_r<std::decay_t<Cols>>...>;
        uint32_t n = 0;
        ((n = cols.size), ...);
        aq::KeyIn ins[] = {aq::KeyIn(cols.container, n, cols.capacity == 0)...};
        int dts[sizeof...(Cols)];
        const void* ptrs[sizeof...(Cols)];
        const void* hosts[] = {static_cast<const void*>(cols.container)...};
        for (size_t k = 0; k < sizeof...(Cols); ++k) { dts[k] = ins[k].dt; ptrs[k] = ins[k].d; }
        // The reference's perfect-hash decision (hasher.h:334-343): integral keys whose TableStats widths add up to at most
        // PerfectHashingThreshold bits span a domain of 2^bits tuples.  Here that bound sizes the device plan (the direct-indexed
        // table of dense.hip is the device form of PerfectHashTable) instead of being discovered from a sample; group order stays
        // first occurrence, the only executable order of the reference (its PerfectHashTable::construct does not compile).
        uint32_t hint = 0;
        if constexpr ((std::is_integral_v<value_type_r<std::decay_t<Cols>>> && ...)) {
            unsigned bits = 0;
            ((bits += cols.stats.bits), ...);
            if (bits <= (unsigned)PerfectHashingThreshold) { const uint64_t dom = 1ull << bits; hint = (uint32_t)(dom < n ? dom : n); if (!hint) hint = 1; }
        }
        aq::GroupTable& t = aq::build_groups((int)sizeof...(Cols), dts, ptrs, n, false, nullptr, hint);
        auto* keys = new std::vector<Tuple>();
        aq::fetch_keys(t, *keys, std::index_sequence_for<Cols...>{}, hosts);
        auto* vecs = aq::make_vecs(t);
        HashTableComponents<value_type_r<std::decay_t<Cols>>...> c{t.G, keys, vecs, t.offsets};
        // the grouping (device handle, row ids, offsets, counts), `keys` and `vecs` live until the module's session ends
        aq::dev::Runtime::get().session_items.push_back({&t, keys, [](void* p) { delete static_cast<std::vector<Tuple>*>(p); }, vecs});
        return c;
    }
};
