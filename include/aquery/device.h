// device.h -- host-side runtime that ties the header-level API (vector_type / ColRef / aggregations /
// hasher) to the C-ABI of the MI355X library (include/aqg.h).  Header-only; state lives in one
// process-wide Runtime (the reference's post-processor is single-caller: one engine thread,
// SURVEY 8b).
//
// Model: a column keeps the reference's 16-byte {container, size, capacity} triple.  Device residency
// is tracked out of band, keyed by the HOST address range of the buffer:
//   PINNED  a borrowed host column (capacity == 0: `ColRef<T>(len, server->getCol(i))`) uploaded on first
//           use and cached -- the reference's zero-copy view of the data source becomes a device mirror;
//           the host data is assumed immutable while the mirror exists (drop_pins() at session end).
//   RESULT  a column produced by a device kernel.  Its host buffer is allocated exactly as the reference
//           would (malloc / scratch arena) but filled lazily: the first host access (operator[], begin(),
//           out(), ...) downloads it AND DROPS THE DEVICE COPY -- vector_type hands out mutable host access
//           (operator[] returns _Ty&), so a mirror kept past that point could go stale behind a host write
//           (`auto x = a + b; x[0] = 5; auto y = x * c;`): a later device use uploads the host data again.
//           Chained expressions such as max(price - mins(price)) never touch the host and never leave HBM.
//           (A grouping's row-id buffer keeps its registration after a download: `col[vecs[g]]` is recognised through it.
//           Generated code only reads vecs[g]; its device copy and the per-grouping aggregate cache assume that, like PINNED.)
// There is no CPU fallback: without the library or a GPU every operation aborts with a message.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#include "../aqg.h"

namespace aq {
namespace dev {

[[noreturn]] inline void die(const char* what, int rc, aqg_ctx* ctx) {
    std::fprintf(stderr, "[aquery-mi355x] %s failed (status %d): %s\n", what, rc, ctx ? aqg_last_error(ctx) : "no context");
    std::abort();
}

// One device group-by (the table behind HashTableFactory::get / AQHashTable).  Its row-id buffer is registered like any
// other RESULT; `vecs[g]` views into it let the runtime recognise `col[vecs[g]]` as "group g of this grouping".
struct GroupCtx {
    aqg_groupby* handle = nullptr;
    uint32_t n = 0, G = 0;
    uint32_t* offsets = nullptr;   // [G+1] host
    uint32_t* counts = nullptr;    // [G]   host
    uint32_t* row_ids = nullptr;   // [n]   host address (device copy registered)
    // (source column, op) -> G result slots of 16 bytes, filled for ALL groups by one kernel on first request
    std::map<std::pair<const void*, int>, std::vector<unsigned char>> cache;
};

struct Entry {
    void* dptr = nullptr;
    size_t bytes = 0;
    bool host_stale = false;   // RESULT not yet downloaded
    bool pinned = false;       // PINNED borrowed column
    GroupCtx* gctx = nullptr;  // set on a grouping's row-id buffer
    // DEFERRED gather `col[vecs[g]]`: nothing has run yet; reductions are answered from the grouping's cache,
    // anything else materialises the gather first
    bool deferred = false;
    GroupCtx* dgroup = nullptr;
    uint32_t dg = 0;
    const void* dsrc = nullptr;
    size_t dsrc_bytes = 0;
    int dtag = 0;
    bool dsrc_borrowed = false;
};

class Runtime {
public:
    static Runtime& get() {
        static Runtime r;
        return r;
    }
    aqg_ctx* ctx() {
        if (!ctx_) {
            int dev = 0;
            if (const char* e = std::getenv("AQ_GPU_DEVICE")) dev = std::atoi(e);
            int rc = aqg_ctx_create(dev, nullptr, &ctx_);
            if (rc != AQG_OK) die("aqg_ctx_create (no MI355X visible; this library has no CPU fallback)", rc, nullptr);
        }
        return ctx_;
    }
    size_t stale = 0;   // number of entries whose host copy is stale (fast path test in operator[])

    // containing entry of a host address, or end()
    std::map<uintptr_t, Entry>::iterator find(const void* p) {
        if (map_.empty()) return map_.end();
        auto it = map_.upper_bound((uintptr_t)p);
        if (it == map_.begin()) return map_.end();
        --it;
        if ((uintptr_t)p < it->first + it->second.bytes || ((uintptr_t)p == it->first && it->second.bytes == 0)) return it;
        return map_.end();
    }
    bool end(std::map<uintptr_t, Entry>::iterator it) { return it == map_.end(); }

    // device address of `bytes` bytes of host data at p.  borrowed != 0: cache the upload (PINNED).
    // `temp_out` receives a temporary device buffer the caller must release() when the data was not cacheable.
    const void* input(const void* p, size_t bytes, bool borrowed, void** temp_out) {
        *temp_out = nullptr;
        if (bytes == 0) return nullptr;
        auto it = find(p);
        if (it != map_.end() && it->second.deferred) materialize(it);
        if (it != map_.end() && it->second.gctx && !it->second.dptr) fill_rows(it->second);
        if (it != map_.end() && (uintptr_t)p + bytes <= it->first + it->second.bytes)
            return static_cast<char*>(it->second.dptr) + ((uintptr_t)p - it->first);
        void* d = nullptr;
        int rc = aqg_malloc(ctx(), bytes, &d);
        if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
        rc = aqg_h2d(ctx_, d, p, bytes);
        if (rc != AQG_OK) die("aqg_h2d", rc, ctx_);
        if (borrowed) {
            Entry e; e.dptr = d; e.bytes = bytes; e.pinned = true;
            map_[(uintptr_t)p] = e;
        } else *temp_out = d;
        return d;
    }
    void release(void* temp) { if (temp) aqg_free(ctx(), temp); }
    // register a host buffer whose device copy already exists (ownership of dptr moves to the registry)
    void adopt(void* p, size_t bytes, void* dptr, bool host_valid) {
        forget_range(p, bytes);
        Entry e; e.dptr = dptr; e.bytes = bytes; e.host_stale = !host_valid;
        map_[(uintptr_t)p] = e;
        if (!host_valid) ++stale;
    }

    // register a fresh RESULT buffer for the host range [p, p+bytes); returns its device address
    void* result(void* p, size_t bytes) {
        forget_range(p, bytes);
        void* d = nullptr;
        int rc = aqg_malloc(ctx(), bytes ? bytes : 16, &d);
        if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
        Entry e; e.dptr = d; e.bytes = bytes; e.host_stale = true;
        map_[(uintptr_t)p] = e;
        ++stale;
        return d;
    }
    // ---- grouped fast path ------------------------------------------------------------------------------------
    // The row-id lists of a grouping (aqg_groupby_postproc: a radix pass over the group-id column, 4.7 ms per 1e9 rows) are made
    // the first time somebody needs them on the device or on the host.  `col[vecs[g]]` followed by a reduction -- the shape the
    // code generator emits -- never does: it is answered by aqg_grouped_reduce from the group-id column of the build.
    void adopt_group(GroupCtx* g, void* drows) {
        adopt(g->row_ids, (size_t)g->n * 4, drows, /*host_valid=*/false);
        auto it = map_.find((uintptr_t)g->row_ids);
        if (it != map_.end()) it->second.gctx = g;
    }
    void fill_rows(Entry& e) {
        GroupCtx* c = e.gctx;
        void *doff = nullptr, *drows = nullptr;
        int rc = aqg_malloc(ctx(), ((size_t)c->G + 1) * 4, &doff);
        if (rc == AQG_OK) rc = aqg_malloc(ctx_, ((size_t)c->n + 1) * 4, &drows);
        if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
        rc = aqg_groupby_postproc(c->handle, static_cast<uint32_t*>(doff), static_cast<uint32_t*>(drows));
        if (rc != AQG_OK) die("aqg_groupby_postproc", rc, ctx_);
        aqg_free(ctx_, doff);
        e.dptr = drows;
    }
    // is [idx, idx+count) exactly the row list of one group of a registered grouping?
    bool group_of(const uint32_t* idx, uint32_t count, GroupCtx** gc, uint32_t* g) {
        auto it = find(idx);
        if (it == map_.end() || !it->second.gctx) return false;
        GroupCtx* c = it->second.gctx;
        uint32_t off = (uint32_t)(idx - c->row_ids);
        uint32_t lo = 0, hi = c->G;                       // largest g with offsets[g] <= off
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (c->offsets[mid] <= off) lo = mid; else hi = mid; }
        if (c->G == 0 || c->offsets[lo] != off || c->counts[lo] != count) return false;
        *gc = c; *g = lo;
        return true;
    }
    void defer_gather(void* host_out, size_t bytes, GroupCtx* gc, uint32_t g, const void* src, size_t src_bytes, bool src_borrowed, int tag) {
        forget_range(host_out, bytes);
        Entry e; e.bytes = bytes; e.host_stale = true; e.deferred = true; e.dgroup = gc; e.dg = g;
        e.dsrc = src; e.dsrc_bytes = src_bytes; e.dsrc_borrowed = src_borrowed; e.dtag = tag;
        map_[(uintptr_t)host_out] = e;
        ++stale;
    }
    void materialize(std::map<uintptr_t, Entry>::iterator it) {
        Entry& e = it->second;
        GroupCtx* c = e.dgroup;
        void* d = nullptr;
        int rc = aqg_malloc(ctx(), e.bytes ? e.bytes : 16, &d);
        if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
        e.deferred = false;                                // (before input(): the source may be this very registry)
        e.dptr = d;
        void* tmp = nullptr;
        const void* dsrc = input(e.dsrc, e.dsrc_bytes, e.dsrc_borrowed, &tmp);
        void* tmp2 = nullptr;
        const void* drows = input(c->row_ids + c->offsets[e.dg], (size_t)c->counts[e.dg] * 4, false, &tmp2);
        rc = aqg_gather(ctx_, e.dtag, dsrc, static_cast<const uint32_t*>(drows), c->counts[e.dg], d);
        if (rc != AQG_OK) die("aqg_gather", rc, ctx_);
        release(tmp); release(tmp2);
    }
    // op(col[vecs[g]]) for a deferred gather at p: answered from the per-grouping cache (one kernel for all groups)
    bool deferred_reduce(const void* p, int op, void* out16) {
        auto it = map_.find((uintptr_t)p);
        if (it == map_.end() || !it->second.deferred) return false;
        Entry& e = it->second;
        GroupCtx* c = e.dgroup;
        auto key = std::make_pair(e.dsrc, op);
        auto hit = c->cache.find(key);
        if (hit == c->cache.end()) {
            const int ot = aqg_reduce_out_dtype(op, e.dtag);
            const size_t osz = aqg_dtype_size(ot);
            void* dout = nullptr;
            int rc = aqg_malloc(ctx(), (size_t)c->G * 16 + 16, &dout);
            if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
            void* tmp = nullptr;
            const void* dsrc = input(e.dsrc, e.dsrc_bytes, e.dsrc_borrowed, &tmp);
            rc = aqg_grouped_reduce(ctx_, c->handle, op, e.dtag, dsrc, dout);
            if (rc != AQG_OK) die("aqg_grouped_reduce", rc, ctx_);
            std::vector<unsigned char> packed((size_t)c->G * osz), slots((size_t)c->G * 16, 0);
            rc = aqg_d2h(ctx_, packed.data(), dout, packed.size());
            if (rc != AQG_OK) die("aqg_d2h", rc, ctx_);
            for (uint32_t g = 0; g < c->G; ++g) std::memcpy(&slots[(size_t)g * 16], &packed[(size_t)g * osz], osz);
            release(tmp);
            aqg_free(ctx_, dout);
            hit = c->cache.emplace(key, std::move(slots)).first;
        }
        std::memcpy(out16, &hit->second[(size_t)e.dg * 16], 16);
        return true;
    }

    // make the host copy of the buffer containing p valid
    void touch(const void* p) {
        auto it = find(p);
        if (it != map_.end() && it->second.deferred) materialize(it);
        if (it == map_.end() || !it->second.host_stale) return;
        if (it->second.gctx && !it->second.dptr) fill_rows(it->second);
        int rc = aqg_d2h(ctx(), (void*)it->first, it->second.dptr, it->second.bytes);
        if (rc != AQG_OK) die("aqg_d2h", rc, ctx_);
        it->second.host_stale = false;
        --stale;
        // mutable host access follows: a RESULT's device copy cannot be trusted from here on
        if (!it->second.pinned && !it->second.gctx) { if (it->second.dptr) aqg_free(ctx_, it->second.dptr); map_.erase(it); }
    }
    // the host buffer at p is going away / being rewritten by the host
    void forget(const void* p) {
        auto it = map_.find((uintptr_t)p);
        if (it == map_.end()) return;
        if (it->second.host_stale) --stale;
        if (it->second.dptr) aqg_free(ctx(), it->second.dptr);
        map_.erase(it);
    }
    void forget_range(const void* p, size_t bytes) {
        if (map_.empty()) return;
        auto it = map_.lower_bound((uintptr_t)p);
        while (it != map_.end() && it->first < (uintptr_t)p + (bytes ? bytes : 1)) {
            if (it->second.host_stale) --stale;
            if (it->second.dptr) aqg_free(ctx(), it->second.dptr);
            it = map_.erase(it);
        }
    }
    void drop_pins() {
        for (auto it = map_.begin(); it != map_.end();) {
            if (it->second.pinned) { aqg_free(ctx(), it->second.dptr); it = map_.erase(it); } else ++it;
        }
    }
    void sync() { aqg_sync(ctx()); }

    // Groupings made through HashTableFactory::get belong to the SESSION of the module that made them (the reference leaks them:
    // "Memory leak here, cleanup after module is done", hasher.h:255): device handle, row ids, offsets / counts, the key vector and
    // the vecs array are released by Context::end_session() / the module's __AQ_End_Session__ hook / the unload of the module.
    struct SessionItem { GroupCtx* table; void* keys; void (*free_keys)(void*); void* vecs; };
    std::vector<SessionItem> session_items;
    void release_session() {
        for (auto& it : session_items) {
            if (it.table) {
                if (it.table->row_ids) { forget(it.table->row_ids); std::free(it.table->row_ids); }
                if (it.table->handle) aqg_groupby_destroy(it.table->handle);
                std::free(it.table->offsets); std::free(it.table->counts);
                delete it.table;
            }
            if (it.keys && it.free_keys) it.free_keys(it.keys);
            std::free(it.vecs);
        }
        session_items.clear();
    }

private:
    Runtime() = default;
    ~Runtime() {
        release_session();
        if (ctx_) {
            for (auto& kv : map_) if (kv.second.dptr) aqg_free(ctx_, kv.second.dptr);
            aqg_ctx_destroy(ctx_);
        }
    }
    aqg_ctx* ctx_ = nullptr;
    std::map<uintptr_t, Entry> map_;
};

inline void host_touch(const void* p) {
    Runtime& r = Runtime::get();
    if (r.stale) r.touch(p);
}

inline void check(int rc, const char* what) {
    if (rc != AQG_OK) die(what, rc, Runtime::get().ctx());
}

// dtype tag of a C++ element type (reference server/types.h:162-190 mapping)
template <class T> struct tag_of { static constexpr int value = AQG_ERROR; };
#define AQ_TAG(T, V) template <> struct tag_of<T> { static constexpr int value = V; };
AQ_TAG(int, AQG_INT32) AQ_TAG(float, AQG_FLOAT) AQ_TAG(double, AQG_DOUBLE) AQ_TAG(long, AQG_INT64) AQ_TAG(long long, AQG_INT64)
AQ_TAG(short, AQG_INT16) AQ_TAG(signed char, AQG_INT8) AQ_TAG(char, AQG_INT8) AQ_TAG(unsigned char, AQG_UINT8)
AQ_TAG(unsigned short, AQG_UINT16) AQ_TAG(unsigned int, AQG_UINT32) AQ_TAG(unsigned long, AQG_UINT64) AQ_TAG(unsigned long long, AQG_UINT64)
AQ_TAG(bool, AQG_BOOL)
#ifdef __SIZEOF_INT128__
AQ_TAG(__int128, AQG_INT128) AQ_TAG(unsigned __int128, AQG_UINT128)
#endif
#undef AQ_TAG
template <class T> constexpr bool on_device = tag_of<std::remove_cv_t<T>>::value != AQG_ERROR;

// RAII view of a column's device address for the duration of one call
struct In {
    const void* d = nullptr;
    void* temp = nullptr;
    In(const void* host, size_t bytes, bool borrowed) { d = Runtime::get().input(host, bytes, borrowed, &temp); }
    ~In() { Runtime::get().release(temp); }
    In(const In&) = delete;
    In& operator=(const In&) = delete;
};

} // namespace dev
} // namespace aq
