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
//           out(), ...) downloads it.  Chained expressions such as max(price - mins(price)) never leave HBM.
// There is no CPU fallback: without the library or a GPU every operation aborts with a message.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "../aqg.h"

namespace aq {
namespace dev {

[[noreturn]] inline void die(const char* what, int rc, aqg_ctx* ctx) {
    std::fprintf(stderr, "[aquery-mi355x] %s failed (status %d): %s\n", what, rc, ctx ? aqg_last_error(ctx) : "no context");
    std::abort();
}

struct Entry {
    void* dptr = nullptr;
    size_t bytes = 0;
    bool host_stale = false;   // RESULT not yet downloaded
    bool pinned = false;       // PINNED borrowed column
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
    // make the host copy of the buffer containing p valid
    void touch(const void* p) {
        auto it = find(p);
        if (it == map_.end() || !it->second.host_stale) return;
        int rc = aqg_d2h(ctx(), (void*)it->first, it->second.dptr, it->second.bytes);
        if (rc != AQG_OK) die("aqg_d2h", rc, ctx_);
        it->second.host_stale = false;
        --stale;
    }
    // the host buffer at p is going away / being rewritten by the host
    void forget(const void* p) {
        auto it = map_.find((uintptr_t)p);
        if (it == map_.end()) return;
        if (it->second.host_stale) --stale;
        aqg_free(ctx(), it->second.dptr);
        map_.erase(it);
    }
    void forget_range(const void* p, size_t bytes) {
        if (map_.empty()) return;
        auto it = map_.lower_bound((uintptr_t)p);
        while (it != map_.end() && it->first < (uintptr_t)p + (bytes ? bytes : 1)) {
            if (it->second.host_stale) --stale;
            aqg_free(ctx(), it->second.dptr);
            it = map_.erase(it);
        }
    }
    void drop_pins() {
        for (auto it = map_.begin(); it != map_.end();) {
            if (it->second.pinned) { aqg_free(ctx(), it->second.dptr); it = map_.erase(it); } else ++it;
        }
    }
    void sync() { aqg_sync(ctx()); }

private:
    Runtime() = default;
    ~Runtime() {
        if (ctx_) {
            for (auto& kv : map_) aqg_free(ctx_, kv.second.dptr);
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
