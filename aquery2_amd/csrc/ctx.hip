// ctx.hip -- context, stream, device memory, column pins, type rules, HIP-event timer.
#include <string_view>

#include <mutex>

#include "aqg_internal.hpp"

extern "C" {

const char* aqg_version(void) { return "aquery2_amd 0.1 (gfx950)"; }

// ---- type rules --------------------------------------------------------------------------
// server/types.h:192-193 (AType_sizes) for the numeric tags
size_t aqg_dtype_size(int dt) {
    switch (dt) {
    case AQG_INT8: case AQG_UINT8: case AQG_BOOL: case AQG_CHAR: return 1;
    case AQG_INT16: case AQG_UINT16: return 2;
    case AQG_INT32: case AQG_UINT32: case AQG_FLOAT: return 4;
    case AQG_INT64: case AQG_UINT64: case AQG_DOUBLE: return 8;
    case AQG_INT128: case AQG_UINT128: return 16;
    default: return 0;
    }
}
// types::GetLongType server/types.h:205-210
int aqg_long_type(int dt) {
    if (!aqg_dtype_size(dt)) return AQG_ERROR;
    if (dt_is_fp(dt)) return AQG_DOUBLE;
    return dt_is_unsigned(dt) ? AQG_UINT128 : AQG_INT128;
}
// types::GetFPType server/types.h:199-204
int aqg_fp_type(int dt) {
    if (!aqg_dtype_size(dt)) return AQG_ERROR;
    return aqg_dtype_size(dt) == 4 ? AQG_FLOAT : AQG_DOUBLE;
}
static int int_tag(size_t sz, bool uns) {
    switch (sz) {
    case 1: return uns ? AQG_UINT8 : AQG_INT8;
    case 2: return uns ? AQG_UINT16 : AQG_INT16;
    case 4: return uns ? AQG_UINT32 : AQG_INT32;
    case 8: return uns ? AQG_UINT64 : AQG_INT64;
    case 16: return uns ? AQG_UINT128 : AQG_INT128;
    }
    return AQG_ERROR;
}
// types::Coercion server/types.h:264-275, including its uint64 quirk (aqis_same<unsigned long,
// const char*> holds, so uint64 mixed with another type coerces to `const char*`)
int aqg_coercion(int a, int b) {
    size_t sa = aqg_dtype_size(a), sb = aqg_dtype_size(b);
    if (!sa || !sb) return AQG_ERROR;
    if (a == AQG_BOOL || b == AQG_BOOL) { if (a == b) return a; }
    else if (dt_is_unsigned(a) == dt_is_unsigned(b) && dt_is_fp(a) == dt_is_fp(b) && sa == sb) return a;
    if (a == AQG_UINT64 || b == AQG_UINT64) return AQG_STR;
    int t0;
    if (sa <= sb) {
        if (sa == sb) t0 = dt_is_fp(a) ? a : (dt_is_fp(b) ? b : (dt_is_unsigned(a) ? b : a));
        else t0 = b;
    } else t0 = a;
    if (dt_is_fp(a) || dt_is_fp(b)) return aqg_fp_type(t0);
    if (!(dt_is_unsigned(a) && dt_is_unsigned(b))) return int_tag(aqg_dtype_size(t0), false);
    return t0;
}

// ---- context -------------------------------------------------------------------------------
int aqg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int aqg_ctx_create(int device, void* hip_stream, aqg_ctx** out) {
    if (!out) return AQG_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return AQG_ERR_NODEVICE;
    aqg_ctx* ctx = new aqg_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return AQG_ERR_NODEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hip_stream) {
        ctx->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return AQG_ERR_HIP; }
        ctx->own_stream = true;
    }
    hipEventCreate(&ctx->ev0);
    hipEventCreate(&ctx->ev1);
    hipEventCreate(&ctx->evk0);
    hipEventCreate(&ctx->evk1);
    hipEventCreate(&ctx->ev_flags);
    *out = ctx;
    return AQG_OK;
}

void aqg_ctx_destroy(aqg_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->pins) {
        if (kv.second.ev) { hipEventSynchronize(kv.second.ev); hipEventDestroy(kv.second.ev); }
        for (auto& r : kv.second.regs) (void)hipHostUnregister(r.first);
        hipFree(kv.second.dptr);
    }
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    for (int k = 0; k < 2; ++k) { if (ctx->up_buf[k]) hipHostFree(ctx->up_buf[k]); if (ctx->up_ev[k]) hipEventDestroy(ctx->up_ev[k]); }
    if (ctx->ws) hipFree(ctx->ws);
    if (ctx->rank_bm) hipFree(ctx->rank_bm);
    for (auto& e : ctx->pool) hipFree(e.first);
    if (ctx->pool_big) hipFree(ctx->pool_big);
    if (ctx->host_stage) hipHostFree(ctx->host_stage);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->evk0) hipEventDestroy(ctx->evk0);
    if (ctx->evk1) hipEventDestroy(ctx->evk1);
    if (ctx->ev_flags) hipEventDestroy(ctx->ev_flags);
    (void)aqg_col_fetch_wait(ctx);
    if (ctx->ev_fetch) hipEventDestroy(ctx->ev_fetch);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* aqg_last_error(aqg_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }
void* aqg_ctx_stream(aqg_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int aqg_sync(aqg_ctx* ctx) {
    if (!ctx) return AQG_ERR_ARG;
    AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AQG_OK;
}

int aqg_reserve_workspace(aqg_ctx* ctx, size_t bytes) {
    if (!ctx) return AQG_ERR_ARG;
    return aqg_ws_ensure(ctx, bytes);
}

int aqg_malloc(aqg_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return AQG_ERR_ARG;
    AQG_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); *dptr = nullptr; return AQG_ERR_NOMEM; }
    return AQG_OK;
}
int aqg_free(aqg_ctx* ctx, void* dptr) {
    if (!ctx) return AQG_ERR_ARG;
    if (!dptr) return AQG_OK;
    AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    AQG_HIP(ctx, hipFree(dptr));
    return AQG_OK;
}
int aqg_h2d(aqg_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return AQG_ERR_ARG;
    if (!bytes) return AQG_OK;
    AQG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AQG_OK;
}
int aqg_d2h(aqg_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return AQG_ERR_ARG;
    if (!bytes) return AQG_OK;
    AQG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AQG_OK;
}
int aqg_d2d(aqg_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx || ((!dst || !src) && bytes)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_d2d: bad argument");
    if (!bytes) return AQG_OK;
    AQG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return AQG_OK;
}
int aqg_memset(aqg_ctx* ctx, void* dst, int byte, size_t bytes) {
    if (!ctx) return AQG_ERR_ARG;
    if (!bytes) return AQG_OK;
    AQG_HIP(ctx, hipMemsetAsync(dst, byte, bytes, ctx->stream));
    return AQG_OK;
}

// Ingest of a borrowed host column (the data source's result buffer: reference server/monetdb_conn.cpp:203-224 hands out zero-copy
// pointers into MonetDB's memory).  The upload is ASYNCHRONOUS and stream-ordered: the host range is page-locked chunk by chunk
// (hipHostRegister) and copied by DMA on a copy stream of its own while this call returns; the context's stream waits for the
// completion event, so every later call of this library sees the data.  Measured on the MI355X box (4 GB column): 56-57 GB/s
// including the registration (PCIe line rate) against 21 GB/s for a first pageable hipMemcpy.
//
// Page-locked ranges must never overlap (profiles/r3_hostregister_abort.md: the runtime keeps registered ranges in a map keyed by
// their start address; overlapping registrations are accepted, cannot all be undone, and a later hipHostUnregister aborts the
// process), whoever locked them: every range this LIBRARY has registered is in one process-wide list (several contexts of a process
// -- one per GPU thread, the header layer's next to a harness's -- see each other's), and pages somebody ELSE has page-locked (the
// host application, a pinned torch tensor, the data source itself) are found by asking the runtime about the chunk's first and last
// page and every 2 MB in between.  A chunk that touches either kind is never registered and never copied directly -- a direct copy
// that starts in page-locked memory and runs past its end faults on the device -- but staged through two pinned buffers of our own.
namespace {
struct RegRange { char* b; size_t len; };
std::mutex g_reg_mu;
std::vector<RegRange> g_regs;                       // every range registered through aqg_col_pin, all contexts of the process
bool lib_registered(const char* b, const char* e) {  // (g_reg_mu held)
    for (const auto& r : g_regs) if (b < r.b + r.len && r.b < e) return true;
    return false;
}
// is any probed page of [b, e) page-locked by somebody else?
bool foreign_registered(const char* b, const char* e) {
    constexpr size_t STEP = (size_t)2 << 20;
    auto probe = [](const char* p) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        const hipError_t rc = hipPointerGetAttributes(&at, p);
        if (rc != hipSuccess) { (void)hipGetLastError(); return false; }     // (older runtimes: "invalid value" for pageable memory)
        return at.type == hipMemoryTypeHost || at.type == hipMemoryTypeManaged;
    };
    if (e <= b) return false;
    if (probe(b) || probe(e - 1)) return true;
    for (const char* p = b + STEP; p < e; p += STEP) if (probe(p)) return true;
    return false;
}
} // namespace
static void pin_release(aqg_ctx* ctx, aqg_pin& p) {
    // no page is unlocked while a DMA of this context may still read it (every upload in flight, not only this column's: chunks are
    // issued back to back on the one copy stream)
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (p.ev) { hipEventSynchronize(p.ev); hipEventDestroy(p.ev); p.ev = nullptr; }
    {
        std::lock_guard<std::mutex> lock(g_reg_mu);
        for (auto& r : p.regs) {
            (void)hipHostUnregister(r.first);
            for (size_t i = 0; i < g_regs.size(); ++i) if (g_regs[i].b == static_cast<char*>(r.first)) { g_regs[i] = g_regs.back(); g_regs.pop_back(); break; }
        }
    }
    p.regs.clear();
    (void)hipGetLastError();
}
int aqg_col_pin(aqg_ctx* ctx, const void* host_ptr, size_t bytes, void** dptr) {
    if (!ctx || !host_ptr || !dptr) return AQG_ERR_ARG;
    auto it = ctx->pins.find(host_ptr);
    if (it != ctx->pins.end() && it->second.bytes >= bytes) { *dptr = it->second.dptr; return AQG_OK; }
    if (it != ctx->pins.end()) { pin_release(ctx, it->second); aqg_free(ctx, it->second.dptr); ctx->pins.erase(it); }
    void* d = nullptr;
    AQG_TRY(aqg_malloc(ctx, bytes, &d));
    aqg_pin pin{d, bytes, {}, nullptr};
    // every exit below this line releases what the call has locked and allocated so far
    auto fail = [&](int rc) { if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream); pin_release(ctx, pin); (void)hipFree(d); (void)hipGetLastError(); return rc; };
    if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) { ctx->err = "aqg_col_pin: hipStreamCreateWithFlags failed"; return fail(AQG_ERR_HIP); }
    static const bool no_register = getenv("AQG_PIN_PAGEABLE") != nullptr;    // A/B measurements only
    constexpr size_t CHUNK = (size_t)256 << 20, PAGE = 4096;
    auto staged = [&](size_t off, size_t len) -> bool {
        constexpr size_t SB = (size_t)32 << 20;
        for (int k = 0; k < 2; ++k) if (!ctx->up_buf[k]) {
            if (hipHostMalloc(&ctx->up_buf[k], SB, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ctx->up_ev[k], hipEventDisableTiming) != hipSuccess) {
                ctx->err = "aqg_col_pin: no pinned staging memory"; (void)hipGetLastError(); return false;
            }
        }
        for (int k = 0; len; k ^= 1) {
            const size_t c = len < SB ? len : SB;
            (void)hipEventSynchronize(ctx->up_ev[k]);                      // the copy that last used this buffer
            memcpy(ctx->up_buf[k], static_cast<const char*>(host_ptr) + off, c);
            if (hipMemcpyAsync(static_cast<char*>(d) + off, ctx->up_buf[k], c, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess) {
                ctx->err = "aqg_col_pin: staged hipMemcpyAsync failed"; (void)hipGetLastError(); return false;
            }
            (void)hipEventRecord(ctx->up_ev[k], ctx->copy_stream);
            off += c; len -= c;
        }
        return true;
    };
    auto copy = [&](size_t off, size_t len) -> bool {
        if (!len) return true;
        hipError_t e = hipMemcpyAsync(static_cast<char*>(d) + off, static_cast<const char*>(host_ptr) + off, len, hipMemcpyHostToDevice, ctx->copy_stream);
        if (e == hipSuccess) return true;
        (void)hipGetLastError();
        return staged(off, len);
    };
    bool ok = true;
    ctx->pin_chunks[0] = ctx->pin_chunks[1] = ctx->pin_chunks[2] = 0;
    for (size_t o = 0; o < bytes && ok; o += CHUNK) {
        const size_t c = bytes - o < CHUNK ? bytes - o : CHUNK;
        const char* src = static_cast<const char*>(host_ptr) + o;
        // whole pages inside [src, src + c): neighbouring chunks (and neighbouring columns) never share a registered page.  The
        // page-locked body and the pageable edges are copied by SEPARATE calls: the runtime classifies a copy by its start address,
        // and a copy that starts in registered memory and runs past its end faults on the device (found by bench.py, 4 GB column).
        char* rb = reinterpret_cast<char*>(((uintptr_t)src + PAGE - 1) & ~(uintptr_t)(PAGE - 1));
        char* re = reinterpret_cast<char*>(((uintptr_t)src + c) & ~(uintptr_t)(PAGE - 1));
        // the pages the chunk TOUCHES (its partial first and last page included): a direct copy must not start or end in locked memory either
        const char* tb = reinterpret_cast<const char*>((uintptr_t)src & ~(uintptr_t)(PAGE - 1));
        const char* te = reinterpret_cast<const char*>(((uintptr_t)src + c + PAGE - 1) & ~(uintptr_t)(PAGE - 1));
        bool reg = false, locked_by_others;
        {
            std::lock_guard<std::mutex> lock(g_reg_mu);        // check and register in one step: two GPU threads pinning slices of one array
            locked_by_others = lib_registered(tb, te) || foreign_registered(tb, te);
            if (!locked_by_others && !no_register && c >= ((size_t)1 << 20) && re > rb) {
                if (hipHostRegister(rb, (size_t)(re - rb), hipHostRegisterDefault) == hipSuccess) {
                    pin.regs.emplace_back(rb, (size_t)(re - rb));
                    g_regs.push_back(RegRange{rb, (size_t)(re - rb)});
                    reg = true;
                } else (void)hipGetLastError();
            }
        }
        ++ctx->pin_chunks[locked_by_others ? 1 : reg ? 0 : 2];
        if (locked_by_others) { ok = staged(o, c); continue; }
        if (reg) ok = copy(o, (size_t)(rb - src)) && copy(o + (size_t)(rb - src), (size_t)(re - rb)) && copy(o + (size_t)(re - src), (size_t)(src + c - re));
        else ok = copy(o, c);
    }
    if (!ok) return fail(AQG_ERR_HIP);
    if (hipEventCreateWithFlags(&pin.ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(pin.ev, ctx->copy_stream) != hipSuccess ||
        hipStreamWaitEvent(ctx->stream, pin.ev, 0) != hipSuccess) { ctx->err = "aqg_col_pin: completion event"; return fail(AQG_ERR_HIP); }
    ctx->pins[host_ptr] = pin;
    *dptr = d;
    return AQG_OK;
}
// Egress of a result column to caller-owned host memory (the write-back half of the ingest seam: TableInfo::monetdb_append_table,
// reference server/table_ext_monetdb.hpp:34-87, hands the data source POINTERS to the result columns).  Asynchronous: ordered behind
// everything queued on the context's stream (an event the copy stream waits for), the destination page-locked chunk by chunk under
// the same rules as aqg_col_pin's source (one process-wide registry, foreign locks probed; such chunks take the runtime's pageable
// path), DMA on the copy stream -- several columns' copies overlap each other and the tail kernels of the query.  aqg_col_fetch_wait
// completes them and unlocks the pages.
int aqg_col_fetch(aqg_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    if (!ctx || ((!dst_host || !src_dev) && bytes)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_col_fetch: bad argument");
    if (!bytes) return AQG_OK;
    if (!ctx->copy_stream) AQG_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!ctx->ev_fetch) AQG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fetch, hipEventDisableTiming));
    AQG_HIP(ctx, hipEventRecord(ctx->ev_fetch, ctx->stream));
    AQG_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_fetch, 0));
    constexpr size_t CHUNK = (size_t)256 << 20, PAGE = 4096;
    for (size_t o = 0; o < bytes; o += CHUNK) {
        const size_t c = bytes - o < CHUNK ? bytes - o : CHUNK;
        char* dst = static_cast<char*>(dst_host) + o;
        char* rb = reinterpret_cast<char*>(((uintptr_t)dst + PAGE - 1) & ~(uintptr_t)(PAGE - 1));
        char* re = reinterpret_cast<char*>(((uintptr_t)dst + c) & ~(uintptr_t)(PAGE - 1));
        const char* tb = reinterpret_cast<const char*>((uintptr_t)dst & ~(uintptr_t)(PAGE - 1));
        const char* te = reinterpret_cast<const char*>(((uintptr_t)dst + c + PAGE - 1) & ~(uintptr_t)(PAGE - 1));
        bool reg = false;
        {
            std::lock_guard<std::mutex> lock(g_reg_mu);
            if (c >= ((size_t)1 << 20) && re > rb && !lib_registered(tb, te) && !foreign_registered(tb, te) &&
                hipHostRegister(rb, (size_t)(re - rb), hipHostRegisterDefault) == hipSuccess) {
                ctx->fetch_regs.emplace_back(rb, (size_t)(re - rb));
                g_regs.push_back(RegRange{rb, (size_t)(re - rb)});
                reg = true;
            } else (void)hipGetLastError();
        }
        const char* src = static_cast<const char*>(src_dev) + o;
        auto cp = [&](size_t off, size_t len) -> hipError_t { return len ? hipMemcpyAsync(dst + off, src + off, len, hipMemcpyDeviceToHost, ctx->copy_stream) : hipSuccess; };
        // (page-locked body and pageable edges by separate calls, as in aqg_col_pin: the runtime classifies a copy by its start address)
        if (reg) { AQG_HIP(ctx, cp(0, (size_t)(rb - dst))); AQG_HIP(ctx, cp((size_t)(rb - dst), (size_t)(re - rb))); AQG_HIP(ctx, cp((size_t)(re - dst), (size_t)(dst + c - re))); }
        else AQG_HIP(ctx, cp(0, c));
    }
    ctx->fetch_pending = true;
    return AQG_OK;
}
int aqg_col_fetch_wait(aqg_ctx* ctx) {
    if (!ctx) return AQG_ERR_ARG;
    if (!ctx->fetch_pending && ctx->fetch_regs.empty()) return AQG_OK;
    hipError_t e = ctx->copy_stream ? hipStreamSynchronize(ctx->copy_stream) : hipSuccess;
    {
        std::lock_guard<std::mutex> lock(g_reg_mu);
        for (auto& r : ctx->fetch_regs) {
            (void)hipHostUnregister(r.first);
            for (size_t i = 0; i < g_regs.size(); ++i) if (g_regs[i].b == static_cast<char*>(r.first)) { g_regs[i] = g_regs.back(); g_regs.pop_back(); break; }
        }
    }
    ctx->fetch_regs.clear();
    ctx->fetch_pending = false;
    (void)hipGetLastError();
    if (e != hipSuccess) { ctx->err = std::string("aqg_col_fetch_wait: ") + hipGetErrorString(e); return AQG_ERR_HIP; }
    return AQG_OK;
}
int aqg_col_pin_last(aqg_ctx* ctx, uint32_t* registered, uint32_t* staged, uint32_t* pageable) {
    if (!ctx) return AQG_ERR_ARG;
    if (registered) *registered = ctx->pin_chunks[0];
    if (staged) *staged = ctx->pin_chunks[1];
    if (pageable) *pageable = ctx->pin_chunks[2];
    return AQG_OK;
}
int aqg_col_unpin(aqg_ctx* ctx, const void* host_ptr) {
    if (!ctx) return AQG_ERR_ARG;
    auto it = ctx->pins.find(host_ptr);
    if (it == ctx->pins.end()) return AQG_OK;
    pin_release(ctx, it->second);
    int rc = aqg_free(ctx, it->second.dptr);
    ctx->pins.erase(it);
    return rc;
}
int aqg_col_unpin_all(aqg_ctx* ctx) {
    if (!ctx) return AQG_ERR_ARG;
    hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->pins) { pin_release(ctx, kv.second); hipFree(kv.second.dptr); }
    ctx->pins.clear();
    return AQG_OK;
}

int aqg_timer_start(aqg_ctx* ctx) {
    if (!ctx) return AQG_ERR_ARG;
    AQG_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return AQG_OK;
}
int aqg_timer_stop_ms(aqg_ctx* ctx, float* ms) {
    if (!ctx || !ms) return AQG_ERR_ARG;
    AQG_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    AQG_HIP(ctx, hipEventSynchronize(ctx->ev1));
    AQG_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return AQG_OK;
}

int aqg_last_kernel_ms(aqg_ctx* ctx, float* ms) {
    if (!ctx || !ms) return AQG_ERR_ARG;
    if (!ctx->evk_valid) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_last_kernel_ms: no timed kernel yet");
    AQG_HIP(ctx, hipEventSynchronize(ctx->evk1));
    AQG_HIP(ctx, hipEventElapsedTime(ms, ctx->evk0, ctx->evk1));
    return AQG_OK;
}

} // extern "C"

// ---- workspace arena ---------------------------------------------------------------------------
int aqg_ws_reset(aqg_ctx* ctx) {
    ctx->ws_off = 0;
    return AQG_OK;
}
int aqg_ws_ensure(aqg_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_cap) return AQG_OK;
    AQG_HIP(ctx, hipSetDevice(ctx->device));
    AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    size_t cap = ctx->ws_cap ? ctx->ws_cap : (size_t)1 << 20;
    while (cap < bytes) cap *= 2;
    // large arenas grow to what is asked for (rounded to 1 GiB), not to the next power of two: a 130 GB need must not take 256 GB
    if (bytes > ((size_t)4 << 30)) cap = (bytes + (((size_t)1 << 30) - 1)) & ~(((size_t)1 << 30) - 1);
    // the arena holds nothing across calls: release it first, so that a 100 GB arena can be replaced by a larger one
    if (ctx->ws) { hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_cap = 0; }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) { ctx->err = std::string("workspace hipMalloc: ") + hipGetErrorString(e); (void)hipGetLastError(); return AQG_ERR_NOMEM; }
    ctx->ws = static_cast<char*>(p);
    ctx->ws_cap = cap;
    return AQG_OK;
}
// NOTE: a grow in the middle of a call would invalidate earlier sub-allocations of the same
// call, so every API entry computes its total need first and calls aqg_ws_ensure once.
int aqg_ws_alloc(aqg_ctx* ctx, size_t bytes, void** out) {
    size_t off = (ctx->ws_off + 255) & ~(size_t)255;
    if (off + bytes > ctx->ws_cap) {
        if (off != 0) return aqg_fail(ctx, AQG_ERR_NOMEM, "workspace overflow inside a call (internal sizing bug)");
        AQG_TRY(aqg_ws_ensure(ctx, bytes));
    }
    *out = ctx->ws + off;
    ctx->ws_off = off + bytes;
    return AQG_OK;
}
int aqg_host_stage(aqg_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->host_stage_cap) {
        if (ctx->host_stage) { hipStreamSynchronize(ctx->stream); hipHostFree(ctx->host_stage); }
        size_t cap = bytes < 4096 ? 4096 : bytes;
        AQG_HIP(ctx, hipHostMalloc(&ctx->host_stage, cap, hipHostMallocDefault));
        ctx->host_stage_cap = cap;
    }
    *out = ctx->host_stage;
    return AQG_OK;
}
