// aqg_internal.hpp -- host-side internals of libaqg (context, workspace, dtype dispatch).
// gfx950 only: no portability layer, no CPU fallback (the product fails loudly without a GPU).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/aqg.h"

// device mirror of a borrowed host column: `regs` = the host ranges page-locked for its upload (released at unpin), `ev` = upload done
struct aqg_pin { void* dptr; size_t bytes; std::vector<std::pair<void*, size_t>> regs; hipEvent_t ev; };

struct aqg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    std::string err;
    // workspace arena: one allocation, bump-allocated per API call
    char* ws = nullptr;
    size_t ws_cap = 0, ws_off = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evk0 = nullptr, evk1 = nullptr;   // bracket the dominant kernel of the last call
    hipEvent_t ev_flags = nullptr;               // recorded behind the copy of a group table's flag words (run_agg)
    hipStream_t copy_stream = nullptr;           // uploads of borrowed host columns (aqg_col_pin), created on first use
    void* up_buf[2] = {nullptr, nullptr};        // pinned staging of the fallback upload path (ranges the runtime refuses to copy directly)
    // device buffers of destroyed / outgrown result handles, kept for the next handle (a fresh handle per query -- what the header
    // layer does -- paid five hipMalloc + five hipFree, the latter each a device-wide synchronisation: ~0.1 ms of a 1.5 ms Q1 call)
    std::vector<std::pair<void*, size_t>> pool;
    size_t pool_bytes = 0;
    void* pool_big = nullptr; size_t pool_big_cap = 0;      // ONE large buffer (a row-to-group map of a build: 4 bytes per row) kept for the next handle
    uint32_t* rank_bm = nullptr;                 // a bitmap over row ids that is ALL ZERO between calls (group ranking of mid-size tables sets and clears only its own bits)
    size_t rank_bm_words = 0;
    hipEvent_t up_ev[2] = {nullptr, nullptr};
    bool tail_in_flight = false;                 // the last group-by returned with its tail kernels still queued (stream-ordered)
    bool evk_valid = false;
    bool evk_frozen = false;                     // the exchange's re-aggregation must not replace the row pass as "the dominant kernel"
    std::unordered_map<const void*, aqg_pin> pins;
    std::vector<std::pair<void*, size_t>> fetch_regs;   // host ranges page-locked for egress in flight (aqg_col_fetch), released by aqg_col_fetch_wait
    bool fetch_pending = false;
    hipEvent_t ev_fetch = nullptr;
    uint32_t pin_chunks[3] = {0, 0, 0};          // chunks of the last aqg_col_pin upload: page-locked + DMA / staged through pinned buffers / plain pageable copy
    std::unordered_map<const void*, int> max_lds;   // largest dynamic LDS size already granted per kernel (aqg_allow_lds)
    // pinned host staging for small results
    void* host_stage = nullptr;
    size_t host_stage_cap = 0;
};

#define AQG_HIP(ctx, call)                                                                     \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(_e);                    \
            (void)hipGetLastError(); /* do not leave a sticky error for the next launch check */ \
            return AQG_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define AQG_TRY(expr)                          \
    do {                                       \
        int _rc = (expr);                      \
        if (_rc != AQG_OK) return _rc;         \
    } while (0)

static inline int aqg_fail(aqg_ctx* ctx, int code, const char* msg) {
    if (ctx) ctx->err = msg;
    return code;
}
#define AQG_CHECK_ROWS(ctx, n, what) do { if ((uint64_t)(n) > (uint64_t)AQG_MAX_ROWS) return aqg_fail(ctx, AQG_ERR_ARG, what ": more than AQG_MAX_ROWS rows"); } while (0)
static inline uint32_t aqg_ceil_div(uint32_t n, uint32_t d) { return (uint32_t)(((uint64_t)n + d - 1) / d); }

// workspace: reset at the start of an API call, then bump.  Growing synchronises the
// stream (earlier kernels may still read the old arena) -- call aqg_reserve_workspace
// ahead of timed regions.
int aqg_ws_reset(aqg_ctx* ctx);
// small-buffer pool of a context (stream-ordered reuse: every user of these buffers runs on ctx->stream)
static inline void* aqg_pool_take(aqg_ctx* ctx, size_t need, size_t* cap) {
    if (ctx->pool_big && ctx->pool_big_cap >= need && ctx->pool_big_cap <= 2 * need) { void* p = ctx->pool_big; *cap = ctx->pool_big_cap; ctx->pool_big = nullptr; ctx->pool_big_cap = 0; return p; }
    int best = -1;
    const size_t slack = need * 4 > ((size_t)64 << 10) ? need * 4 : ((size_t)64 << 10);
    for (int i = 0; i < (int)ctx->pool.size(); ++i)
        if (ctx->pool[i].second >= need && ctx->pool[i].second <= slack && (best < 0 || ctx->pool[i].second < ctx->pool[best].second)) best = i;
    if (best < 0) return nullptr;
    void* p = ctx->pool[best].first;
    *cap = ctx->pool[best].second;
    ctx->pool_bytes -= *cap;
    ctx->pool[best] = ctx->pool.back();
    ctx->pool.pop_back();
    return p;
}
static inline void aqg_pool_give(aqg_ctx* ctx, void* p, size_t cap) {
    if (!p) return;
    if (ctx && cap && cap <= ((size_t)64 << 20) && ctx->pool.size() < 64 && ctx->pool_bytes + cap <= ((size_t)512 << 20)) { ctx->pool.emplace_back(p, cap); ctx->pool_bytes += cap; return; }
    if (ctx && cap > ((size_t)64 << 20) && cap <= ((size_t)8 << 30) && cap > ctx->pool_big_cap) {        // the larger one stays
        if (ctx->pool_big) (void)hipFree(ctx->pool_big);
        ctx->pool_big = p; ctx->pool_big_cap = cap;
        return;
    }
    (void)hipFree(p);
}
int aqg_ws_alloc(aqg_ctx* ctx, size_t bytes, void** out);
int aqg_ws_ensure(aqg_ctx* ctx, size_t bytes);
template <class T> static inline int aqg_ws_get(aqg_ctx* ctx, size_t count, T** out) {
    void* p = nullptr;
    int rc = aqg_ws_alloc(ctx, count * sizeof(T), &p);
    *out = static_cast<T*>(p);
    return rc;
}
int aqg_host_stage(aqg_ctx* ctx, size_t bytes, void** out);
// in-place exclusive scan of `count` uint32 words; bsum: scratch of ceil(count/2048) words (postproc.hip)
int aqg_exclusive_scan_u32(aqg_ctx* ctx, uint32_t* d, uint64_t count, uint32_t* bsum);

// postproc.hip: the stable radix passes over a build's group-id column.  x == nullptr: descending row ids per group -> row_ids_dev; else the
// `esz`-byte elements of column x in the flat row-list layout -> xout.  ws_managed: the caller reset the workspace and sized it with
// aqg_postproc_ws_bytes (so that its own sub-allocations survive)
struct aqg_groupby;
size_t aqg_postproc_ws_bytes(uint32_t n, uint32_t G, int esz);
int aqg_radix_by_group(aqg_ctx* ctx, aqg_groupby* g, uint32_t* row_ids_dev, const void* x, int esz, void* xout, bool ws_managed);
int aqg_group_offsets(aqg_ctx* ctx, const aqg_groupby* g, uint32_t* offsets_dev, uint32_t* bsum);
// groupby.hip: out[g] = op(x[rows whose id in gid_col is g]) through the group-by plans (gid_col: n dense ids in first-occurrence order)
extern "C" int aqg_grouped_reduce_keyed(aqg_ctx* ctx, aqg_groupby* g, const uint32_t* gid_col, int op, int t, const void* x, void* out_dev);

// exchange.hip: the all-gather of a communicator (`bytes` bytes of every rank, rank order, on / ordered behind the context's stream)
struct aqg_comm;
int aqg_comm_allgather_internal(aqg_comm* c, const void* send_dev, void* recv_dev, size_t bytes);
aqg_ctx* aqg_comm_ctx(aqg_comm* c);
// scan.hip: running min / max of one row-range shard, seeded with the fold of the earlier shards
extern "C" int aqg_scan_minmax_seeded(aqg_ctx* ctx, int op, int t, const void* x, uint32_t n, const void* seed_host, void* out);

// reduce.hip: raw moments of a column / of a pair of columns left in device memory (the sharded reductions ship them)
extern "C" int aqg_stats_dev(aqg_ctx* ctx, int t, const void* x, uint32_t n, int flags, void* out_dev48);
extern "C" int aqg_corr_sums_dev(aqg_ctx* ctx, int tx, const void* x, int ty, const void* y, uint32_t n, void* out_dev80);
// exchange.hip: grow-only device scratch of a communicator (send / receive sides of its small exchanges)
int aqg_comm_scratch(aqg_comm* c, size_t send_bytes, size_t recv_bytes, void** send, void** recv);

// internal aggregate of aqg_groupby_agg (not in aqg.h): the sum of squares, typed like SUM -- the second moment the sharded call ships for VAR / STDDEV
constexpr int AQG_RED_SUMSQ = 64;

// HIP events around the dominant kernel of a call (read back by aqg_last_kernel_ms)
static inline void aqg_kernel_timer_begin(aqg_ctx* ctx) { if (!ctx->evk_frozen) (void)hipEventRecord(ctx->evk0, ctx->stream); }
static inline void aqg_kernel_timer_end(aqg_ctx* ctx) { if (!ctx->evk_frozen) { (void)hipEventRecord(ctx->evk1, ctx->stream); ctx->evk_valid = true; } }

// launch check
static inline int aqg_check_launch(aqg_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ctx->err = std::string(what) + ": " + hipGetErrorString(e);
        return AQG_ERR_HIP;
    }
    return AQG_OK;
}

// grid sizing for memory-bound grid-stride kernels: enough blocks to fill 256 CUs,
// capped (guide: Guideline 11)
// hipFuncAttributeMaxDynamicSharedMemorySize once per kernel and size, not on every launch (a few microseconds of host time each)
static inline int aqg_allow_lds(aqg_ctx* ctx, const void* kernel, size_t lds) {
    auto it = ctx->max_lds.find(kernel);
    if (it != ctx->max_lds.end() && (size_t)it->second >= lds) return AQG_OK;
    AQG_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->max_lds[kernel] = (int)lds;
    return AQG_OK;
}

static inline unsigned aqg_grid(const aqg_ctx* ctx, uint64_t work_items, unsigned block, unsigned items_per_thread,
                                unsigned blocks_per_cu = 8) {
    uint64_t per_block = (uint64_t)block * items_per_thread;
    uint64_t need = (work_items + per_block - 1) / per_block;
    uint64_t cap = (uint64_t)ctx->num_cu * blocks_per_cu;
    if (need < 1) need = 1;
    return (unsigned)(need < cap ? need : cap);
}

// ---- dtype helpers (host) ----------------------------------------------------------------
static inline size_t dt_size(int dt) { return aqg_dtype_size(dt); }
static inline bool dt_is_fp(int dt) { return dt == AQG_FLOAT || dt == AQG_DOUBLE; }
static inline bool dt_is_unsigned(int dt) {
    return dt == AQG_UINT8 || dt == AQG_UINT16 || dt == AQG_UINT32 || dt == AQG_UINT64 || dt == AQG_UINT128 || dt == AQG_BOOL;
}
static inline bool dt_is_num(int dt) {
    switch (dt) {
    case AQG_INT8: case AQG_INT16: case AQG_INT32: case AQG_INT64: case AQG_UINT8: case AQG_UINT16:
    case AQG_UINT32: case AQG_UINT64: case AQG_FLOAT: case AQG_DOUBLE: return true;
    }
    return false;
}

// Dispatch a generic lambda on the C type of a numeric dtype tag.
template <class T> struct aqg_tag { using type = T; };
template <class F> static inline int aqg_dispatch_num(int dt, F&& f) {
    switch (dt) {
    case AQG_INT8: return f(aqg_tag<int8_t>{});
    case AQG_INT16: return f(aqg_tag<int16_t>{});
    case AQG_INT32: return f(aqg_tag<int32_t>{});
    case AQG_INT64: return f(aqg_tag<int64_t>{});
    case AQG_UINT8: return f(aqg_tag<uint8_t>{});
    case AQG_UINT16: return f(aqg_tag<uint16_t>{});
    case AQG_UINT32: return f(aqg_tag<uint32_t>{});
    case AQG_UINT64: return f(aqg_tag<uint64_t>{});
    case AQG_FLOAT: return f(aqg_tag<float>{});
    case AQG_DOUBLE: return f(aqg_tag<double>{});
    }
    return AQG_ERR_DTYPE;
}
