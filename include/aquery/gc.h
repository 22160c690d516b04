// gc.h -- deferred-free queue and bump arena used by the library API (names as in the reference's
// server/gc.h so generated code that toggles `GC::scratch_space` compiles unchanged).
// Differences on purpose (SURVEY defects D5/D6): everything is inline in the header, and the deferred
// queue is a growable vector behind a mutex instead of a fixed 8192-slot array with a daemon thread.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "device.h"

class ScratchSpace {
public:
    void* ret = nullptr;
    char* scratchspace = nullptr;
    size_t ptr = 0, cnt = 0, capacity = 0, initial_capacity = 0;
    std::vector<std::pair<char*, size_t>> retired;   // arenas outgrown since the last release(), and their sizes

    void init(size_t initial) {
        scratchspace = static_cast<char*>(std::malloc(initial));
        ptr = cnt = 0;
        capacity = initial_capacity = initial;
    }
    void* alloc(uint32_t sz) {
        size_t need = (size_t)sz;
        size_t at = (cnt + 15) & ~(size_t)15;             // 16-byte aligned: device-side vector loads
        if (at + need > capacity) {
            retired.push_back({scratchspace, capacity});   // live temporaries keep pointing into the old arena
            capacity = at + need + (capacity >> 1);
            scratchspace = static_cast<char*>(std::malloc(capacity));
            at = 0;
        }
        ptr = at;
        cnt = at + need;
        return scratchspace + at;
    }
    void register_ret(void* r) { ret = r; }
    // does `p` point into memory that the next release() gives up?
    bool owns(const void* p) const {
        const char* c = static_cast<const char*>(p);
        if (scratchspace && c >= scratchspace && c < scratchspace + capacity) return true;
        for (const auto& m : retired) if (c >= m.first && c < m.first + m.second) return true;
        return false;
    }
    void release() {
        aq::dev::Runtime::get().forget_range(scratchspace, capacity);   // device mirrors of arena temporaries die with them
        for (auto& m : retired) { aq::dev::Runtime::get().forget_range(m.first, m.second); std::free(m.first); }
        retired.clear();
        ptr = cnt = 0;
    }
    void reset() {
        release();
        ret = nullptr;
        if (capacity != initial_capacity) {
            capacity = initial_capacity;
            scratchspace = static_cast<char*>(std::realloc(scratchspace, capacity));
        }
    }
    void cleanup() {
        release();
        std::free(scratchspace);
        scratchspace = nullptr;
    }
};

class GC {
public:
    using gc_deallocator_t = void (*)(void*);
    ScratchSpace scratch;

    explicit GC(uint64_t max_bytes = 0xfffffff, uint32_t threshold = 64) : max_bytes_(max_bytes), threshold_(threshold) {
        GC::gc_handle = this;
        scratch.init(65536);
        GC::scratch_space = nullptr;
    }
    ~GC() {
        collect();
        scratch.cleanup();
        if (GC::gc_handle == this) GC::gc_handle = nullptr;
    }
    // deferred free: small blocks go straight back, large ones are queued and released in batches
    void reg(void* v, uint32_t sz = 0xffffffff, void (*f)(void*) = std::free) {
        if (!v || !f) return;
        if (sz < threshold_) { f(v); return; }
        std::lock_guard<std::mutex> g(mu_);
        q_.push_back({v, f});
        pending_ += sz == 0xffffffff ? threshold_ : sz;
        if (pending_ > max_bytes_ || q_.size() > 4096) collect_locked();
    }
    void collect() { std::lock_guard<std::mutex> g(mu_); collect_locked(); }
    uint32_t get_threshold() const { return threshold_; }

    template <class T> static inline gc_deallocator_t _delete(T*) { return [](void* v) { delete (T*)v; }; }
    constexpr static void (*_free)(void*) = std::free;

    static inline GC* gc_handle = nullptr;
    static inline ScratchSpace* scratch_space = nullptr;

private:
    struct item { void* p; void (*f)(void*); };
    void collect_locked() { for (auto& i : q_) i.f(i.p); q_.clear(); pending_ = 0; }
    std::mutex mu_;
    std::vector<item> q_;
    uint64_t pending_ = 0, max_bytes_;
    uint32_t threshold_;
};
