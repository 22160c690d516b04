// exchange.hip -- the exchange step of row-sharded group-bys INSIDE the library (BASELINE north_star: "tables shard by
// row range across the 8 GPUs of one node with a single RCCL reduce/all-gather over xGMI only for the final global aggregate or
// low-cardinality group merge").  The reference has no distributed form; this replaces nothing there.
//
//   aqg_comm_*                 one communicator per (process, GPU): RCCL (ncclAllGather on the context's stream; librccl is opened
//                              with dlopen when the first RCCL communicator is made, so single-GPU users never load it) or a
//                              caller-supplied all-gather (tests and rehearsals on one GPU, hosts with their own transport)
//   aqg_groupby_agg_sharded    group-by + aggregates over THIS rank's row range, then ONE all-gather of the shard's group table
//                              -- k key columns, the global first row, one partial per aggregate (SUM -> sum, COUNT -> count,
//                              MIN / MAX -> itself, AVG -> sum and count, VAR / STDDEV -> sum, sum of squares and count; 128-bit
//                              partials as two 8-byte columns) -- and a re-aggregation of the concatenation on every
//                              rank.  Shards are contiguous row ranges gathered in rank order, so first occurrence in the
//                              concatenation is the global first occurrence: the merged groups come out in the reference's order
//                              (server/hasher.h:176-198) without any row id but each group's first crossing the wire.
// Payload of a rank (gcap = group capacity of the exchange): 8-byte words, column-major --
//   {ngroups, 0} | key_0[gcap] ... key_{k-1}[gcap] | first_row[gcap] | partial_0[gcap] ... partial_{m-1}[gcap]
#include <dlfcn.h>

#include "aqg_internal.hpp"
#include <mutex>

#include "dev_common.hpp"
#include "groupby_handle.hpp"

namespace {

constexpr int MAXPART = 2 * MAXAGG;      // payload columns of one exchange (AVG ships two, a 128-bit partial two)
constexpr int DT_HI128 = 1001;           // pseudo dtype of a payload column: the HIGH 8 bytes of a 128-bit result column

// ---- RCCL through dlopen ------------------------------------------------------------------------------------------------------
struct NcclId { char internal[128]; };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
// opened once per process, under a mutex: a host with one thread per GPU calls aqg_comm_unique_id / aqg_comm_init_rccl from several
// threads at the same time, and a half-filled table must never be visible.  A failed load is retried by the next caller.
Rccl* rccl(std::string* err) {
    static std::mutex mu;
    static Rccl ready;
    static bool ok = false;
    std::lock_guard<std::mutex> lock(mu);
    if (ok) return &ready;
    Rccl r;
    // The RCCL that sits NEXT TO the HIP runtime this library is bound to: a process may hold two ROCm stacks (PyTorch wheels bundle
    // their own libamdhip64 / libhsa-runtime64 / librccl), and an RCCL from the other one opens its own, uninitialised HSA runtime and
    // reports "no ROCm-capable device" (seen when torch was imported after this library and before the first communicator).
    {
        Dl_info di;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash);
                for (const char* leaf : {"/librccl.so.1", "/librccl.so"}) { r.lib = dlopen((dir + leaf).c_str(), RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
            }
        }
    }
    if (!r.lib) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
    if (!r.lib) { if (err) *err = std::string("librccl not found: ") + dlerror(); return nullptr; }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) { if (err) *err = "librccl lacks an expected symbol"; dlclose(r.lib); return nullptr; }
    ready = r;                                 // published complete, under the lock
    ok = true;
    return &ready;
}

// one partial of the local group-by.  `wide`: its result is a 128-bit integer that does not fit 64 bits per shard (sums of 8-byte
// integers, sums of squares): it travels as two payload columns -- low halves summed as uint64, high halves as int64 / uint64 -- and
// total = (sum of the high halves << 64) + sum of the low halves, all mod 2^128
struct Partial { int local_op; int val_dt; int val_index; int part_dt; int merge_op; int wide; int col0; };   // col0: its first payload column

} // namespace

struct aqg_comm {
    aqg_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    void* nccl = nullptr;                    // ncclComm_t
    aqg_allgather_fn fn = nullptr;
    void* user = nullptr;
    // grow-only device buffers
    void *send = nullptr, *recv = nullptr, *cat = nullptr, *hdr = nullptr;
    void *xsend = nullptr, *xrecv = nullptr; size_t xsend_cap = 0, xrecv_cap = 0;   // aqg_reduce_sharded / aqg_scan_sharded
    size_t send_cap = 0, recv_cap = 0, cat_cap = 0, hdr_cap = 0;
    aqg_groupby *local = nullptr, *merged = nullptr;
    aqg_groupby* merged_x[3] = {nullptr, nullptr, nullptr};     // further merge calls when one cannot hold all payload columns (8 accumulators per call)
    // the merged columns after exchange_core: [0] global first rows (int64), [1 + c] payload column c re-aggregated
    const void* mres[1 + 2 * MAXAGG] = {};
    int mres_dt[1 + 2 * MAXAGG] = {};
};

namespace {

bool small_int(int dt) { return dt == AQG_INT8 || dt == AQG_INT16 || dt == AQG_INT32 || dt == AQG_UINT8 || dt == AQG_UINT16 || dt == AQG_UINT32 || dt == AQG_BOOL; }
bool is_fp(int dt) { return dt == AQG_FLOAT || dt == AQG_DOUBLE; }

int grow(aqg_ctx* ctx, void** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return AQG_OK;
    if (*p) { AQG_HIP(ctx, hipStreamSynchronize(ctx->stream)); AQG_HIP(ctx, hipFree(*p)); *p = nullptr; *cap = 0; }
    const size_t want = need < 4096 ? 4096 : need;
    if (hipMalloc(p, want) != hipSuccess) { ctx->err = "exchange: hipMalloc failed"; (void)hipGetLastError(); return AQG_ERR_NOMEM; }
    *cap = want;
    return AQG_OK;
}

int allgather(aqg_comm* c, const void* send, void* recv, size_t bytes) {
    aqg_ctx* ctx = c->ctx;
    if (c->fn) {
        const int rc = c->fn(c->user, send, recv, bytes, (void*)ctx->stream);
        if (rc != 0) return aqg_fail(ctx, AQG_ERR_HIP, "exchange: the caller's all-gather failed");
        return AQG_OK;
    }
    Rccl* r = rccl(&ctx->err);
    if (!r) return AQG_ERR_HIP;
    const int rc = r->AllGather(send, recv, bytes, /*ncclInt8*/ 0, c->nccl, ctx->stream);
    if (rc != 0) { ctx->err = std::string("ncclAllGather: ") + (r->GetErrorString ? r->GetErrorString(rc) : "error"); return AQG_ERR_HIP; }
    return AQG_OK;
}

__device__ inline uint64_t load_native(int dt, const void* col, size_t i) {
    switch (dt) {
    case AQG_INT8: return (uint64_t)(int64_t)static_cast<const int8_t*>(col)[i];
    case AQG_INT16: return (uint64_t)(int64_t)static_cast<const int16_t*>(col)[i];
    case AQG_INT32: return (uint64_t)(int64_t)static_cast<const int32_t*>(col)[i];
    case AQG_UINT8: case AQG_BOOL: return static_cast<const uint8_t*>(col)[i];
    case AQG_UINT16: return static_cast<const uint16_t*>(col)[i];
    case AQG_UINT32: case AQG_FLOAT: return static_cast<const uint32_t*>(col)[i];
    case DT_HI128: return static_cast<const uint64_t*>(col)[2 * i + 1];
    case AQG_INT128: case AQG_UINT128: return static_cast<const uint64_t*>(col)[2 * i];          // low half: partial sums of <= 4-byte integers over < 2^32 rows fit
    default: return static_cast<const uint64_t*>(col)[i];
    }
}
// (not inlined: see store_sized in groupby_dev.hpp and profiles/r2_hipcc_switch_miscompile.md)
__device__ __noinline__ void store_native(int dt, void* col, size_t i, uint64_t bits) {
    switch (dt) {
    case AQG_INT8: case AQG_UINT8: case AQG_BOOL: static_cast<uint8_t*>(col)[i] = (uint8_t)bits; break;
    case AQG_INT16: case AQG_UINT16: static_cast<uint16_t*>(col)[i] = (uint16_t)bits; break;
    case AQG_INT32: case AQG_UINT32: case AQG_FLOAT: static_cast<uint32_t*>(col)[i] = (uint32_t)bits; break;
    default: static_cast<uint64_t*>(col)[i] = bits; break;
    }
}

struct PackSpec {
    int ncols;                        // nkeys + 1 + nparts
    int nkeys;
    const void* src[MAXKEYS + 1 + MAXPART];
    int src_dt[MAXKEYS + 1 + MAXPART];
    uint64_t row_base;
    uint32_t status;                  // this rank's failure before the exchange (AQG_ERR_*, 0 = fine): it ships no groups and every rank returns it
};
// word (c, g) of the payload: column c of group g (column nkeys = the global first row); header = {groups, status}
__global__ void __launch_bounds__(256) xpack_kernel(PackSpec ps, uint32_t G, uint32_t gcap, uint64_t* __restrict__ out) {
    const size_t total = (size_t)ps.ncols * G;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = G; out[1] = ps.status; }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i / G);
        const uint32_t g = (uint32_t)(i - (size_t)c * G);
        uint64_t w = load_native(ps.src_dt[c], ps.src[c], g);
        if (c == ps.nkeys) w = (w & 0xFFFFFFFFull) + ps.row_base;             // local first row (uint32) -> global row id
        out[2 + (size_t)c * gcap + g] = w;
    }
}
struct UnpackSpec {
    int ncols;
    void* dst[MAXKEYS + 1 + MAXPART];
    int dst_dt[MAXKEYS + 1 + MAXPART];
};
// concatenation in rank order: rank r's groups start at the sum of the group counts before it
__global__ void __launch_bounds__(256) xunpack_kernel(const uint64_t* __restrict__ gathered, uint32_t world, uint32_t gcap, size_t words_per_rank, UnpackSpec us) {
    __shared__ uint32_t start[65];
    if (threadIdx.x == 0) { uint32_t s = 0; for (uint32_t r = 0; r < world; ++r) { start[r] = s; s += (uint32_t)gathered[(size_t)r * words_per_rank]; } start[world] = s; }
    __syncthreads();
    for (uint32_t r = 0; r < world; ++r) {
        const uint64_t* p = gathered + (size_t)r * words_per_rank;
        const uint32_t G = start[r + 1] - start[r];
        const size_t total = (size_t)us.ncols * G;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const int c = (int)(i / G);
            const uint32_t g = (uint32_t)(i - (size_t)c * G);
            store_native(us.dst_dt[c], us.dst[c], (size_t)start[r] + g, p[2 + (size_t)c * gcap + g]);
        }
    }
}

struct FinalSpec {
    int nagg;
    int op[MAXAGG], dt[MAXAGG];
    const void* a[MAXAGG];            // merged partial (SUM / MIN / MAX / COUNT), or the merged sum of AVG / VAR (low halves when wide)
    const void* a_hi[MAXAGG];         // merged high halves of a wide sum (null: `a` is the whole sum)
    const void* b[MAXAGG];            // merged count of AVG / VAR (128-bit)
    const void* q[MAXAGG];            // VAR / STDDEV: merged sum of squares (double, or the low halves of the 128-bit one)
    const void* q_hi[MAXAGG];         //               ... its high halves (integers)
    void* out[MAXAGG];
    int out_size[MAXAGG];
};
// (sum of the high halves << 64) + sum of the low halves, mod 2^128; both sums are 128-bit results of the merge
__device__ inline aqg_i128 join_halves(const void* lo, const void* hi, uint32_t g) {
    const aqg_i128 l = static_cast<const aqg_i128*>(lo)[g], h = static_cast<const aqg_i128*>(hi)[g];
    return {l.lo, h.lo + l.hi};
}
__device__ inline aqg_i128 xmul_128(aqg_i128 a, aqg_i128 b) { return {a.lo * b.lo, __umul64hi(a.lo, b.lo) + a.lo * b.hi + a.hi * b.lo}; }
__device__ __noinline__ static void copy_element(void* __restrict__ dst_col, const void* __restrict__ src_col, size_t g, int size) {
    const unsigned char* src = static_cast<const unsigned char*>(src_col) + g * size;
    unsigned char* dst = static_cast<unsigned char*>(dst_col) + g * size;
    for (int k = 0; k < size; ++k) dst[k] = src[k];
}
__global__ void __launch_bounds__(256) xfinal_kernel(FinalSpec fs, uint32_t G) {
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < G; g += gridDim.x * 256) {
        for (int j = 0; j < fs.nagg; ++j) {
            const bool fp = fs.dt[j] == AQG_FLOAT || fs.dt[j] == AQG_DOUBLE;
            const bool uns = fs.dt[j] == AQG_UINT8 || fs.dt[j] == AQG_UINT16 || fs.dt[j] == AQG_UINT32 || fs.dt[j] == AQG_UINT64 || fs.dt[j] == AQG_BOOL;
            auto int_sum = [&]() -> aqg_i128 { return fs.a_hi[j] ? join_halves(fs.a[j], fs.a_hi[j], g) : static_cast<const aqg_i128*>(fs.a[j])[g]; };
            auto to_double = [&](aqg_i128 v) -> double { return uns ? u128_to_double(v.hi, v.lo) : i128_to_double(v); };
            switch (fs.op[j]) {
            case AQG_RED_COUNT: store_at<uint64_t>(fs.out[j], g, static_cast<const aqg_i128*>(fs.a[j])[g].lo); break;   // sum of uint32 counts: unsigned 128-bit
            case AQG_RED_AVG: {                                               // sum / (double)size (aggregations.h:28-32)
                const aqg_i128 cn = static_cast<const aqg_i128*>(fs.b[j])[g];
                const double s = fp ? static_cast<const double*>(fs.a[j])[g] : to_double(int_sum());
                store_at<double>(fs.out[j], g, s / (double)cn.lo);
            } break;
            case AQG_RED_VAR: case AQG_RED_STDDEV: {                          // (ssq - s*s/(double)(n+1)) / (double)(n+1), as emit_record (groupby.hip)
                const double np1 = (double)(static_cast<const aqg_i128*>(fs.b[j])[g].lo + 1);
                double d;
                if (fp) {
                    const double sd = static_cast<const double*>(fs.a[j])[g], qq = static_cast<const double*>(fs.q[j])[g];
                    d = (qq - sd * sd / np1) / np1;
                } else {
                    const aqg_i128 sm = int_sum(), qq = join_halves(fs.q[j], fs.q_hi[j], g);
                    const aqg_i128 ss = xmul_128(sm, sm);                     // s * s in the 128-bit LongType (wraps like the reference)
                    d = (to_double(qq) - to_double(ss) / np1) / np1;
                }
                store_at<double>(fs.out[j], g, fs.op[j] == AQG_RED_STDDEV ? sqrt(d) : d);
            } break;
            case AQG_RED_SUM:
                if (!fp && fs.a_hi[j]) { store_at<aqg_i128>(fs.out[j], g, int_sum()); break; }
                [[fallthrough]];
            default: {                                                        // SUM / MIN / MAX: the merged column is the result
                copy_element(fs.out[j], fs.a[j], g, fs.out_size[j]);
            } break;
            }
        }
    }
}

// ---- the merge of SMALL exchanges in one launch -------------------------------------------------------------------------------------
// world x gcap <= 1024 rows, one key column (h2o Q1 / Q4, config 4: 8 ranks x 128 groups).  ONE workgroup reads the gathered
// payloads directly (no concatenation), groups the rows in an LDS table, numbers the groups by first occurrence in rank order (a
// bitmap of the rows that lead a group + popcounts) and reduces every payload column into dense per-group accumulators: integer
// sums into 128 bits (low word by atomic add, carry and sign into the high word), doubles by atomic add, MIN / MAX through an
// order-preserving 64-bit map.  Replaces concatenate + host read of the row count + a whole aqg_groupby_agg call (ten launches, two
// host round trips): what the exchange adds to a 1.4 ms Q1 step went from ~0.10 to ~0.05 ms at a world of one.
constexpr uint32_t XS_ROWS = 1024, XS_CAP = 2048, XS_MAXCOL = 6;
constexpr unsigned long long XS_EMPTY = 0x8000000000000001ull;
struct SmallMerge {
    int ncols;                         // payload columns
    int kind[XS_MAXCOL];               // 0 signed 128-bit sum, 1 unsigned 128-bit sum, 2 double sum, 3 / 4 signed min / max, 5 / 6 unsigned min / max
    int out_size[XS_MAXCOL];           // bytes per group of the result column (16 / 8 / the native size)
    void* out[XS_MAXCOL];
    void* keys_out; int key_size;
    long long* first_out;
    uint32_t* info;                    // [0] groups, [1] bad header, [2] the first non-zero status word of a shard header (a rank that failed before the exchange)
};
__global__ void __launch_bounds__(1024) xmerge_small_kernel(const uint64_t* __restrict__ gathered, uint32_t world, uint32_t gcap, size_t words_per_rank, SmallMerge sm) {
    __shared__ unsigned long long tkey[XS_CAP + 1];          // (last: the key equal to the empty mark)
    __shared__ uint32_t tlead[XS_CAP + 1];                   // lowest row of the slot's key, later its group id
    __shared__ unsigned long long lead_bits[XS_ROWS / 64];
    __shared__ uint32_t lead_before[XS_ROWS / 64 + 1];
    __shared__ unsigned long long alo[XS_MAXCOL][XS_ROWS], ahi[XS_MAXCOL][XS_ROWS];
    __shared__ long long gfirst[XS_ROWS];
    __shared__ uint32_t off[65];
    __shared__ uint32_t s_bad;
    if (threadIdx.x == 0) {
        uint32_t o = 0, bad = 0, st = 0;
        for (uint32_t r = 0; r < world; ++r) {
            const uint64_t c = gathered[(size_t)r * words_per_rank];
            const uint32_t rs = (uint32_t)gathered[(size_t)r * words_per_rank + 1];
            if (rs && !st) st = rs;
            if (c > gcap) bad = 1;
            off[r] = o;
            o += bad ? 0u : (uint32_t)c;
        }
        off[world] = o;
        s_bad = bad | (st ? 2u : 0u);
        sm.info[2] = st;
    }
    for (uint32_t t = threadIdx.x; t <= XS_CAP; t += 1024) { tkey[t] = XS_EMPTY; tlead[t] = 0xFFFFFFFFu; }
    if (threadIdx.x < XS_ROWS / 64) lead_bits[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t total = s_bad ? 0u : off[world];
    const uint32_t i = threadIdx.x;                          // one row per lane
    const bool live = i < total;
    uint32_t slot = XS_CAP, r = 0, g_in = 0;
    unsigned long long key = 0;
    if (live) {
        while (i >= off[r + 1]) ++r;                           // world <= 64
        g_in = i - off[r];
        key = gathered[(size_t)r * words_per_rank + 2 + g_in];
        if (key != XS_EMPTY) {
            slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 53);          // top 11 bits: XS_CAP slots
            while (true) {
                const unsigned long long cur = tkey[slot];
                if (cur == key) break;
                if (cur == XS_EMPTY) {
                    const unsigned long long old = atomicCAS(&tkey[slot], XS_EMPTY, key);
                    if (old == XS_EMPTY || old == key) break;
                }
                slot = (slot + 1) & (XS_CAP - 1);                             // at most XS_ROWS keys in XS_CAP slots: always ends
            }
        }
        atomicMin(&tlead[slot], i);
    }
    __syncthreads();
    const bool leader = live && tlead[slot] == i;
    if (leader) atomicOr(&lead_bits[i >> 6], 1ull << (i & 63));
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t c = 0; for (uint32_t w = 0; w < XS_ROWS / 64; ++w) { lead_before[w] = c; c += (uint32_t)__popcll(lead_bits[w]); } lead_before[XS_ROWS / 64] = c; }
    __syncthreads();
    const uint32_t G = lead_before[XS_ROWS / 64];
    // dense accumulators of the G groups
    for (uint32_t g = threadIdx.x; g < G; g += 1024) {
        gfirst[g] = 0x7FFFFFFFFFFFFFFFll;
        for (int c = 0; c < sm.ncols; ++c) {
            const int k = sm.kind[c];
            alo[c][g] = k == 3 || k == 5 ? ~0ull : 0ull;       // min: all ones in the mapped order; sums and max: zero
            ahi[c][g] = 0;
        }
    }
    uint32_t gid = 0;
    if (leader) { gid = lead_before[i >> 6] + (uint32_t)__popcll(lead_bits[i >> 6] & ((1ull << (i & 63)) - 1ull)); }
    __syncthreads();
    if (leader) tlead[slot] = gid;                             // (every row of the key read its leader's row above)
    __syncthreads();
    if (live) {
        const uint32_t g = tlead[slot];
        const uint64_t* base = gathered + (size_t)r * words_per_rank + 2;
        atomicMin(reinterpret_cast<long long*>(&gfirst[g]), (long long)base[(size_t)gcap + g_in]);
        for (int c = 0; c < sm.ncols; ++c) {
            const unsigned long long v = base[(size_t)(2 + c) * gcap + g_in];
            switch (sm.kind[c]) {
            case 0: case 1: {
                const unsigned long long old = atomicAdd(&alo[c][g], v);
                const unsigned long long hi_add = (sm.kind[c] == 0 && (long long)v < 0 ? ~0ull : 0ull) + (old + v < old ? 1ull : 0ull);
                if (hi_add) atomicAdd(&ahi[c][g], hi_add);
            } break;
            case 2: atomicAdd(reinterpret_cast<double*>(&alo[c][g]), __builtin_bit_cast(double, v)); break;
            case 3: atomicMin(&alo[c][g], v ^ 0x8000000000000000ull); break;
            case 4: atomicMax(&alo[c][g], v ^ 0x8000000000000000ull); break;
            case 5: atomicMin(&alo[c][g], v); break;
            default: atomicMax(&alo[c][g], v); break;
            }
        }
    }
    __syncthreads();
    if (leader) {
        unsigned char* kd = static_cast<unsigned char*>(sm.keys_out) + (size_t)gid * sm.key_size;
        for (int b = 0; b < sm.key_size; ++b) kd[b] = (unsigned char)(key >> (8 * b));
    }
    for (uint32_t g = threadIdx.x; g < G; g += 1024) {
        sm.first_out[g] = gfirst[g];
        for (int c = 0; c < sm.ncols; ++c) {
            unsigned long long lo = alo[c][g];
            const int k = sm.kind[c];
            if (k == 3 || k == 4) lo ^= 0x8000000000000000ull;
            unsigned char* dst = static_cast<unsigned char*>(sm.out[c]) + (size_t)g * sm.out_size[c];
            if (sm.out_size[c] == 16) { reinterpret_cast<unsigned long long*>(dst)[0] = lo; reinterpret_cast<unsigned long long*>(dst)[1] = ahi[c][g]; }
            else for (int b = 0; b < sm.out_size[c]; ++b) dst[b] = (unsigned char)(lo >> (8 * b));
        }
    }
    if (threadIdx.x == 0) { sm.info[0] = G; sm.info[1] = s_bad; }
}

// Key columns that are not plain integers travel as the NORMALISED integer columns the local group-by made of them (groupby.hip
// normalize_keys): date_t -> uint32, time_t -> uint64 (padding byte cleared), timestamp_t -> {uint32 date, uint64 time}, 128-bit integers
// -> {low, high}.  The layout depends on the key dtypes alone, so every rank -- one whose local call failed too -- sizes the payload alike.
// Floating keys (their NaN rows are numbered by LOCAL row) and strings (codes of a per-rank dictionary: aqg_str_encode_sharded makes global
// ones) are not offered here.
int norm_layout(int nkeys, const int* dts, int* ndt, int* first_of /* [nkeys] first normalised column of user key k */) {
    int m = 0;
    for (int k = 0; k < nkeys; ++k) {
        if (first_of) first_of[k] = m;
        const int dt = dts[k];
        if (m + 2 > MAXKEYS + 1) return -1;
        if (dt == AQG_DATE) ndt[m++] = AQG_UINT32;
        else if (dt == AQG_TIME) ndt[m++] = AQG_UINT64;
        else if (dt == AQG_TIMESTAMP) { ndt[m++] = AQG_UINT32; ndt[m++] = AQG_UINT64; }
        else if (dt == AQG_INT128 || dt == AQG_UINT128) { ndt[m++] = AQG_UINT64; ndt[m++] = AQG_UINT64; }
        else if (small_int(dt) || dt == AQG_INT64 || dt == AQG_UINT64) ndt[m++] = dt;
        else return -1;
        if (m > MAXKEYS) return -1;
    }
    return m;
}
__global__ void __launch_bounds__(256) denorm_timestamp_kernel(const uint32_t* __restrict__ date, const uint64_t* __restrict__ time, uint32_t G, uint32_t* __restrict__ out) {
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < G; g += gridDim.x * 256) { out[3 * (size_t)g] = date[g]; out[3 * (size_t)g + 1] = (uint32_t)time[g]; out[3 * (size_t)g + 2] = (uint32_t)(time[g] >> 32); }
}
__global__ void __launch_bounds__(256) denorm_i128_kernel(const uint64_t* __restrict__ lo, const uint64_t* __restrict__ hi, uint32_t G, uint64_t* __restrict__ out) {
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < G; g += gridDim.x * 256) { out[2 * (size_t)g] = lo[g]; out[2 * (size_t)g + 1] = hi[g]; }
}

// every rank's return when some rank failed before the exchange (its own message stays on the rank that failed)
int remote_failure(aqg_ctx* ctx, int status, bool mine = false) {
    if (!mine || ctx->err.empty()) ctx->err = status == AQG_ERR_OVERFLOW ? "sharded call: a rank's group table exceeded its capacity (gmax / table overflow)" :
                                              status == AQG_ERR_NOMEM ? "sharded call: a rank ran out of device memory" : "sharded call: a rank failed before the exchange";
    return status;
}

// steps 2-4 of the sharded group-by over an existing shard table L (keys, 32-bit first rows, one result column per partial):
// sizes the payload, packs, ONE all-gather, concatenates, re-aggregates into comm->merged (aggregate 0 = MIN of the global first rows)
// payload column p reads result column src_res[p] of L (null: p itself), its high 8 bytes when src_hi[p]
int exchange_core(aqg_comm* comm, aqg_groupby* L, int nkeys, const int* key_dtypes, int nparts, const int* part_dt, const int* merge_op,
                  uint64_t row_base, uint32_t max_groups_hint, uint32_t gmax, const int* src_res = nullptr, const int* src_hi = nullptr, int local_status = AQG_OK) {
    aqg_ctx* ctx = comm->ctx;
    // A failure of THIS rank's part (its group-by ran out of table or memory, its table exceeds gmax) must not keep it out of the
    // collective: the other ranks would wait in the all-gather for ever.  The rank ships an empty table whose header carries the status,
    // and EVERY rank returns that status after the all-gather (the first failed rank's, in rank order).
    uint32_t G = local_status == AQG_OK ? L->ngroups : 0;
    // ---- 2. capacity of the exchange: the caller's bound, or the largest shard table (one 8-byte all-gather and a host read) -------
    const uint32_t world = (uint32_t)comm->world;
    uint32_t gcap = gmax;
    if (gmax) { if (G > gmax) { local_status = AQG_ERR_OVERFLOW; ctx->err = "aqg_groupby_agg_sharded: a shard has more groups than gmax"; G = 0; } }
    else {
        AQG_TRY(grow(ctx, &comm->hdr, &comm->hdr_cap, 8 * ((size_t)world + 1)));
        uint64_t* h = static_cast<uint64_t*>(comm->hdr);
        const uint64_t mine = (uint64_t)G | ((uint64_t)(uint32_t)local_status << 32);
        AQG_HIP(ctx, hipMemcpyAsync(h + world, &mine, 8, hipMemcpyHostToDevice, ctx->stream));
        AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));                    // (`mine` is a stack variable)
        AQG_TRY(allgather(comm, h + world, h, 8));
        uint64_t all[64];
        AQG_HIP(ctx, hipMemcpyAsync(all, h, 8 * (size_t)world, hipMemcpyDeviceToHost, ctx->stream));
        AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        uint64_t mx = 1;
        for (uint32_t r = 0; r < world; ++r) if (all[r] >> 32) return remote_failure(ctx, (int)(all[r] >> 32), (int)r == comm->rank);   // (the same on every rank)
        for (uint32_t r = 0; r < world; ++r) mx = (all[r] & 0xFFFFFFFFull) > mx ? (all[r] & 0xFFFFFFFFull) : mx;
        gcap = (uint32_t)mx;
    }
    const int ncols = nkeys + 1 + nparts;
    const size_t words = 2 + (size_t)ncols * gcap, bytes = words * 8;
    if ((uint64_t)gcap * world > AQG_MAX_ROWS) return aqg_fail(ctx, AQG_ERR_OVERFLOW, "aqg_groupby_agg_sharded: world x group capacity exceeds AQG_MAX_ROWS");
    AQG_TRY(grow(ctx, &comm->send, &comm->send_cap, bytes));
    AQG_TRY(grow(ctx, &comm->recv, &comm->recv_cap, bytes * world));
    // ---- 3. pack, ONE all-gather, concatenate ----------------------------------------------------------------------------------------
    PackSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.ncols = ncols; ps.nkeys = nkeys; ps.row_base = row_base; ps.status = (uint32_t)local_status;
    for (int k = 0; k < nkeys; ++k) { ps.src[k] = L->keys_out[k]; ps.src_dt[k] = key_dtypes[k]; }
    ps.src[nkeys] = L->first_rows; ps.src_dt[nkeys] = AQG_UINT32;
    for (int p = 0; p < nparts; ++p) {
        const int r = src_res ? src_res[p] : p;
        ps.src[nkeys + 1 + p] = L->results[r]; ps.src_dt[nkeys + 1 + p] = src_hi && src_hi[p] ? DT_HI128 : L->res_dt[r];
    }
    hipLaunchKernelGGL(xpack_kernel, dim3(aqg_grid(ctx, (uint64_t)ncols * G + 1, 256, 2, 4)), dim3(256), 0, ctx->stream, ps, G, gcap, static_cast<uint64_t*>(comm->send));
    AQG_TRY(aqg_check_launch(ctx, "xpack_kernel"));
    AQG_TRY(allgather(comm, comm->send, comm->recv, bytes));
    // ---- 3'. small exchanges: one merge kernel over the gathered payloads --------------------------------------------------------------
    {
        static const bool small_off = getenv("AQG_DISABLE_SMALL_MERGE") != nullptr;
        bool small = !small_off && gmax && nkeys == 1 && nparts >= 0 && nparts <= (int)XS_MAXCOL && (uint64_t)gcap * world <= XS_ROWS && aqg_dtype_size(key_dtypes[0]) <= 8 && !is_fp(key_dtypes[0]);
        SmallMerge sm;
        memset(&sm, 0, sizeof sm);
        int rdt[XS_MAXCOL];
        for (int p = 0; p < nparts && small; ++p) {
            const int dt = part_dt[p], op = merge_op[p];
            const bool uns = dt == AQG_UINT8 || dt == AQG_UINT16 || dt == AQG_UINT32 || dt == AQG_UINT64 || dt == AQG_BOOL;
            if (op == AQG_RED_SUM && is_fp(dt)) { if (dt != AQG_DOUBLE) small = false; sm.kind[p] = 2; rdt[p] = AQG_DOUBLE; sm.out_size[p] = 8; }
            else if (op == AQG_RED_SUM) { sm.kind[p] = uns ? 1 : 0; rdt[p] = uns ? AQG_UINT128 : AQG_INT128; sm.out_size[p] = 16; }
            else if ((op == AQG_RED_MIN || op == AQG_RED_MAX) && !is_fp(dt) && aqg_dtype_size(dt) <= 8) { sm.kind[p] = (uns ? 5 : 3) + (op == AQG_RED_MAX ? 1 : 0); rdt[p] = dt; sm.out_size[p] = (int)aqg_dtype_size(dt); }
            else small = false;
        }
        if (small) {
            aqg_groupby* M = comm->merged ? comm->merged : new aqg_groupby();
            comm->merged = M;
            M->ctx = ctx; M->nkeys = 1; M->key_dt[0] = key_dtypes[0]; M->nagg = nparts + 1; M->has_counts = false; M->has_reversemap = false; M->sharded = false;
            AQG_TRY(grow(ctx, &M->keys_out[0], &M->cap_keys[0], XS_ROWS * 8));
            for (int a = 0; a <= nparts; ++a) AQG_TRY(grow(ctx, &M->results[a], &M->cap_results[a], XS_ROWS * 16));
            AQG_TRY(grow(ctx, &comm->hdr, &comm->hdr_cap, 8 * ((size_t)world + 1) + 64));
            sm.ncols = nparts;
            for (int p = 0; p < nparts; ++p) { sm.out[p] = M->results[1 + p]; M->res_dt[1 + p] = rdt[p]; }
            M->res_dt[0] = AQG_INT64;
            sm.keys_out = M->keys_out[0]; sm.key_size = (int)aqg_dtype_size(key_dtypes[0]);
            sm.first_out = static_cast<long long*>(M->results[0]);
            sm.info = reinterpret_cast<uint32_t*>(static_cast<char*>(comm->hdr) + 8 * ((size_t)world + 1));
            hipLaunchKernelGGL(xmerge_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, static_cast<const uint64_t*>(comm->recv), world, gcap, words, sm);
            AQG_TRY(aqg_check_launch(ctx, "xmerge_small_kernel"));
            uint32_t info[3] = {0, 0, 0};
            AQG_HIP(ctx, hipMemcpyAsync(info, sm.info, 12, hipMemcpyDeviceToHost, ctx->stream));
            AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (info[2]) return remote_failure(ctx, (int)info[2]);
            if (info[1]) return aqg_fail(ctx, AQG_ERR_ARG, "exchange: corrupt shard header");
            M->ngroups = info[0]; M->n = info[0];
            for (int a = 0; a <= nparts; ++a) { comm->mres[a] = M->results[a]; comm->mres_dt[a] = M->res_dt[a]; }
            return AQG_OK;
        }
    }
    // concatenated columns: keys in their own dtypes, first rows and partials
    const size_t cat_rows = (size_t)gcap * world;
    size_t col_off[MAXKEYS + 1 + MAXPART], cat_bytes = 0;
    int col_dt[MAXKEYS + 1 + MAXPART];
    for (int c = 0; c < ncols; ++c) {
        col_dt[c] = c < nkeys ? key_dtypes[c] : c == nkeys ? AQG_INT64 : part_dt[c - nkeys - 1];
        col_off[c] = cat_bytes;
        cat_bytes += (cat_rows * aqg_dtype_size(col_dt[c]) + 255) & ~(size_t)255;
    }
    AQG_TRY(grow(ctx, &comm->cat, &comm->cat_cap, cat_bytes + 256));
    UnpackSpec us;
    memset(&us, 0, sizeof us);
    us.ncols = ncols;
    for (int c = 0; c < ncols; ++c) { us.dst[c] = static_cast<char*>(comm->cat) + col_off[c]; us.dst_dt[c] = col_dt[c]; }
    // the shard headers {groups, status}: a failed rank fails the call on every rank; counts are checked before anything trusts them
    uint64_t total = 0;
    {
        uint64_t hd[64][2];
        for (uint32_t r = 0; r < world; ++r) AQG_HIP(ctx, hipMemcpyAsync(&hd[r][0], static_cast<const uint64_t*>(comm->recv) + (size_t)r * words, 16, hipMemcpyDeviceToHost, ctx->stream));
        AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t r = 0; r < world; ++r) if ((uint32_t)hd[r][1]) return remote_failure(ctx, (int)(uint32_t)hd[r][1], (int)r == comm->rank);
        for (uint32_t r = 0; r < world; ++r) { if (hd[r][0] > gcap) return aqg_fail(ctx, AQG_ERR_ARG, "exchange: corrupt shard header"); total += hd[r][0]; }
    }
    hipLaunchKernelGGL(xunpack_kernel, dim3(aqg_grid(ctx, (uint64_t)ncols * gcap + 1, 256, 2, 4)), dim3(256), 0, ctx->stream, static_cast<const uint64_t*>(comm->recv), world, gcap, words, us);
    AQG_TRY(aqg_check_launch(ctx, "xunpack_kernel"));
    if (total > AQG_MAX_ROWS) return aqg_fail(ctx, AQG_ERR_OVERFLOW, "aqg_groupby_agg_sharded: more shard groups than AQG_MAX_ROWS");
    // ---- 4. re-aggregate the concatenation (first occurrence in it = global first occurrence) ------------------------------------------
    const void* mkeys[MAXKEYS];
    for (int k = 0; k < nkeys; ++k) mkeys[k] = us.dst[k];
    uint64_t mhint = total < 64 ? 64 : total;
    if (max_groups_hint && (uint64_t)max_groups_hint * world < mhint) mhint = (uint64_t)max_groups_hint * world;
    // one merge call holds 8 aggregates and 8 accumulators (an 8-byte SUM takes two: its result is 128 bits): the payload columns go
    // through as many calls over the same concatenation as that takes -- the groups come out in the same order every time
    auto acc_cost = [](int dt, int op) { return op == AQG_RED_SUM && (dt == AQG_INT64 || dt == AQG_UINT64) ? 2 : 1; };
    ctx->evk_frozen = true;                  // aqg_last_kernel_ms keeps naming the pass over the shard's rows
    int c = 0, batch = 0, mrc = AQG_OK;
    uint32_t mgroups = 0;
    do {
        int mops[MAXAGG], mdts[MAXAGG], na = 0, acc = 0, col_of[MAXAGG];
        const void* mvals[MAXAGG];
        if (batch == 0) { mops[0] = AQG_RED_MIN; mdts[0] = AQG_INT64; mvals[0] = us.dst[nkeys]; col_of[0] = -1; na = 1; acc = 1; }
        while (c < nparts && na < MAXAGG && acc + acc_cost(part_dt[c], merge_op[c]) <= MAXACC) {
            mops[na] = merge_op[c]; mdts[na] = part_dt[c]; mvals[na] = us.dst[nkeys + 1 + c]; col_of[na] = c;
            acc += acc_cost(part_dt[c], merge_op[c]); ++na; ++c;
        }
        if (batch > 3) { mrc = aqg_fail(ctx, AQG_ERR_ARG, "exchange: too many payload columns"); break; }
        aqg_groupby** slot = batch == 0 ? &comm->merged : &comm->merged_x[batch - 1];
        mrc = aqg_groupby_agg(ctx, nkeys, key_dtypes, mkeys, na, mops, mdts, mvals, (uint32_t)total, (uint32_t)mhint, slot);
        if (mrc != AQG_OK) break;
        if (batch == 0) mgroups = (*slot)->ngroups;
        else if ((*slot)->ngroups != mgroups) { mrc = aqg_fail(ctx, AQG_ERR_HIP, "exchange: merge calls disagree on the group count"); break; }
        for (int a = 0; a < na; ++a) { comm->mres[1 + col_of[a]] = (*slot)->results[a]; comm->mres_dt[1 + col_of[a]] = (*slot)->res_dt[a]; }
        ++batch;
    } while (c < nparts);
    ctx->evk_frozen = false;
    AQG_TRY(mrc);
    return AQG_OK;
}


} // namespace

// internal (reduce.hip / sharded.hip): the communicator's all-gather and context
int aqg_comm_allgather_internal(aqg_comm* c, const void* send_dev, void* recv_dev, size_t bytes) { return allgather(c, send_dev, recv_dev, bytes); }
aqg_ctx* aqg_comm_ctx(aqg_comm* c) { return c ? c->ctx : nullptr; }
int aqg_comm_scratch(aqg_comm* c, size_t send_bytes, size_t recv_bytes, void** send, void** recv) {
    AQG_TRY(grow(c->ctx, &c->xsend, &c->xsend_cap, send_bytes));
    AQG_TRY(grow(c->ctx, &c->xrecv, &c->xrecv_cap, recv_bytes));
    *send = c->xsend; *recv = c->xrecv;
    return AQG_OK;
}

extern "C" {

int aqg_comm_unique_id(void* id_out) {
    if (!id_out) return AQG_ERR_ARG;
    std::string err;
    Rccl* r = rccl(&err);
    if (!r) return AQG_ERR_HIP;
    NcclId id;
    if (r->GetUniqueId(&id) != 0) return AQG_ERR_HIP;
    memcpy(id_out, &id, sizeof id);
    return AQG_OK;
}

int aqg_comm_init_rccl(aqg_ctx* ctx, int rank, int world, const void* id, aqg_comm** out) {
    if (!ctx || !id || !out || world < 1 || world > 64 || rank < 0 || rank >= world) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_comm_init_rccl: bad argument (1..64 ranks)");
    Rccl* r = rccl(&ctx->err);
    if (!r) return AQG_ERR_HIP;
    AQG_HIP(ctx, hipSetDevice(ctx->device));
    NcclId nid;
    memcpy(&nid, id, sizeof nid);
    void* comm = nullptr;
    const int rc = r->CommInitRank(&comm, world, nid, rank);
    if (rc != 0) { ctx->err = std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(rc) : "error"); return AQG_ERR_HIP; }
    aqg_comm* c = new aqg_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->nccl = comm;
    *out = c;
    return AQG_OK;
}

int aqg_comm_init_custom(aqg_ctx* ctx, int rank, int world, aqg_allgather_fn fn, void* user, aqg_comm** out) {
    if (!ctx || !fn || !out || world < 1 || world > 64 || rank < 0 || rank >= world) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_comm_init_custom: bad argument (1..64 ranks)");
    aqg_comm* c = new aqg_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->fn = fn; c->user = user;
    *out = c;
    return AQG_OK;
}

void aqg_comm_destroy(aqg_comm* c) {
    if (!c) return;
    if (c->ctx) hipStreamSynchronize(c->ctx->stream);
    if (c->nccl) { if (Rccl* r = rccl(nullptr)) r->CommDestroy(c->nccl); }
    for (void* p : {c->send, c->recv, c->cat, c->hdr, c->xsend, c->xrecv}) if (p) hipFree(p);
    if (c->local) aqg_groupby_destroy(c->local);
    if (c->merged) aqg_groupby_destroy(c->merged);
    for (aqg_groupby* m : c->merged_x) if (m) aqg_groupby_destroy(m);
    delete c;
}
int aqg_comm_rank(const aqg_comm* c) { return c ? c->rank : -1; }
int aqg_comm_world(const aqg_comm* c) { return c ? c->world : 0; }

const int64_t* aqg_groupby_first_rows64(const aqg_groupby* g) { return g ? g->first_rows64 : nullptr; }

int aqg_groupby_exchange(aqg_comm* comm, aqg_groupby* local, int nparts, const int* merge_ops, uint64_t row_base, uint32_t gmax, aqg_groupby** out) {
    if (!comm || !local || !out || nparts < 0 || nparts > MAXAGG || nparts > local->nagg) return aqg_fail(comm ? comm->ctx : nullptr, AQG_ERR_ARG, "aqg_groupby_exchange: bad argument");
    aqg_ctx* ctx = comm->ctx;
    if (local->nuser) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_groupby_exchange: plain integer key columns only");
    int pdt[MAXPART];
    for (int p = 0; p < nparts; ++p) {
        const int rdt = local->res_dt[p], op = merge_ops[p];
        if (!(op == AQG_RED_SUM || op == AQG_RED_MIN || op == AQG_RED_MAX)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_exchange: partials combine with SUM / MIN / MAX");
        if (op == AQG_RED_SUM) pdt[p] = (rdt == AQG_DOUBLE || rdt == AQG_FLOAT) ? AQG_DOUBLE : rdt == AQG_UINT64 ? AQG_UINT32 : AQG_INT64;   // 128-bit sums travel as their low 64 bits, counts as 32
        else pdt[p] = rdt;
        if (op == AQG_RED_SUM && rdt == AQG_FLOAT) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_groupby_exchange: a float column is not a partial sum");
    }
    AQG_TRY(exchange_core(comm, local, local->nkeys, local->key_dt, nparts, pdt, merge_ops, row_base, 0, gmax));
    aqg_groupby* M = comm->merged;
    aqg_groupby* H = *out ? *out : new aqg_groupby();
    H->ctx = ctx; H->n = local->n; H->ngroups = M->ngroups; H->nkeys = local->nkeys; H->nagg = nparts;
    H->has_counts = false; H->has_reversemap = false; H->sharded = true;
    const size_t GG = M->ngroups ? M->ngroups : 1;
    int rc = AQG_OK;
    auto d2d = [&](void* dst, const void* src, size_t bytes) { return !bytes || hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess ? AQG_OK : AQG_ERR_HIP; };
    for (int k = 0; k < local->nkeys && rc == AQG_OK; ++k) {
        H->key_dt[k] = local->key_dt[k];
        rc = grow(ctx, &H->keys_out[k], &H->cap_keys[k], GG * 8);
        if (rc == AQG_OK) rc = d2d(H->keys_out[k], M->keys_out[k], (size_t)M->ngroups * aqg_dtype_size(local->key_dt[k]));
    }
    if (rc == AQG_OK) rc = grow(ctx, reinterpret_cast<void**>(&H->first_rows64), &H->cap_first64, GG * 8);
    if (rc == AQG_OK) rc = d2d(H->first_rows64, comm->mres[0], (size_t)M->ngroups * 8);
    for (int p = 0; p < nparts && rc == AQG_OK; ++p) {
        H->res_dt[p] = comm->mres_dt[1 + p];
        rc = grow(ctx, &H->results[p], &H->cap_results[p], GG * 16);
        if (rc == AQG_OK) rc = d2d(H->results[p], comm->mres[1 + p], (size_t)M->ngroups * aqg_dtype_size(comm->mres_dt[1 + p]));
    }
    if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(H); return rc == AQG_ERR_HIP ? aqg_fail(ctx, rc, "aqg_groupby_exchange: device copy failed") : rc; }
    *out = H;
    return AQG_OK;
}

int aqg_groupby_agg_sharded(aqg_comm* comm, int nkeys, const int* key_dtypes, const void* const* keys, int naggs, const int* ops, const int* val_dtypes,
                            const void* const* vals, uint32_t n, uint64_t row_base, uint32_t max_groups_hint, uint32_t gmax, aqg_groupby** out) {
    if (!comm || !out) return AQG_ERR_ARG;
    aqg_ctx* ctx = comm->ctx;
    if (nkeys < 1 || nkeys > MAXKEYS || naggs < 0 || naggs > MAXAGG) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_agg_sharded: 1..8 key columns, 0..8 aggregates");
    // ---- what travels: one partial per aggregate (AVG: sum and count; VAR / STDDEV: sum, sum of squares, count), identical (op, column) pairs once
    Partial parts[MAXPART];
    int nparts = 0, ncols = 0, part_of[MAXAGG], cnt_of[MAXAGG], sq_of[MAXAGG];
    auto is_unsigned = [](int dt) { return dt == AQG_UINT8 || dt == AQG_UINT16 || dt == AQG_UINT32 || dt == AQG_UINT64 || dt == AQG_BOOL; };
    auto add_part = [&](int lop, int j) -> int {
        const int dt = val_dtypes[j];
        for (int p = 0; p < nparts; ++p) if (parts[p].local_op == lop && (lop == AQG_RED_COUNT || (vals[parts[p].val_index] == vals[j] && parts[p].val_dt == dt))) return p;
        if (nparts >= MAXPART) return -1;
        Partial& P = parts[nparts];
        P.local_op = lop; P.val_dt = dt; P.val_index = j; P.wide = 0; P.col0 = ncols;
        if (lop == AQG_RED_SUM || lop == AQG_RED_SUMSQ) {
            P.merge_op = AQG_RED_SUM;
            if (is_fp(dt)) P.part_dt = AQG_DOUBLE;
            else if (lop == AQG_RED_SUM && small_int(dt)) P.part_dt = is_unsigned(dt) ? AQG_UINT64 : AQG_INT64;   // a shard's sum of <= 4-byte integers over < 2^32 rows fits 64 bits (unsigned ones: uint64 -- 2^31 rows of 0xFFFFFFFF pass 2^63)
            else { P.part_dt = is_unsigned(dt) ? AQG_UINT64 : AQG_INT64; P.wide = 1; }  // (dtype of the HIGH halves; the low ones are uint64)
        }
        else if (lop == AQG_RED_COUNT) { P.part_dt = AQG_UINT32; P.merge_op = AQG_RED_SUM; }   // a shard has < 2^32 rows: one accumulator in the merge
        else { P.part_dt = dt; P.merge_op = lop; }
        ncols += P.wide ? 2 : 1;
        return nparts++;
    };
    for (int j = 0; j < naggs; ++j) {
        const int op = ops[j], dt = val_dtypes[j];
        part_of[j] = cnt_of[j] = sq_of[j] = -1;
        const bool num = small_int(dt) || is_fp(dt) || dt == AQG_INT64 || dt == AQG_UINT64;
        if (op != AQG_RED_COUNT && !num) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_groupby_agg_sharded: integer columns of 1 to 8 bytes and floating columns");
        bool ok = true;
        switch (op) {
        case AQG_RED_SUM: case AQG_RED_MIN: case AQG_RED_MAX: case AQG_RED_COUNT: part_of[j] = add_part(op, j); ok = part_of[j] >= 0; break;
        case AQG_RED_AVG: part_of[j] = add_part(AQG_RED_SUM, j); cnt_of[j] = add_part(AQG_RED_COUNT, j); ok = part_of[j] >= 0 && cnt_of[j] >= 0; break;
        case AQG_RED_VAR: case AQG_RED_STDDEV:
            part_of[j] = add_part(AQG_RED_SUM, j); sq_of[j] = add_part(AQG_RED_SUMSQ, j); cnt_of[j] = add_part(AQG_RED_COUNT, j);
            ok = part_of[j] >= 0 && sq_of[j] >= 0 && cnt_of[j] >= 0;
            break;
        default: return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_groupby_agg_sharded: SUM / COUNT / MIN / MAX / AVG / VAR / STDDEV (FIRST / LAST do not decompose through this call)");
        }
        if (!ok) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_agg_sharded: too many partial columns");
    }
    if (nparts > MAXAGG || ncols > MAXPART) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_agg_sharded: too many partials (8 local aggregates, 16 payload columns per call; a 128-bit partial takes two columns)");
    // ---- 1. this shard ------------------------------------------------------------------------------------------------------------
    int lops[MAXAGG], ldts[MAXAGG];
    const void* lvals[MAXAGG];
    for (int p = 0; p < nparts; ++p) { lops[p] = parts[p].local_op; ldts[p] = parts[p].val_dt; lvals[p] = vals[parts[p].val_index]; }
    // (argument errors above are the same on every rank; from here on a failure is this rank's alone and travels through the exchange)
    int lrc = aqg_groupby_agg(ctx, nkeys, key_dtypes, keys, nparts, lops, ldts, lvals, n, max_groups_hint, &comm->local);
    if (!comm->local) { comm->local = new aqg_groupby(); comm->local->ctx = ctx; }
    aqg_groupby* L = comm->local;
    // the key columns the exchange moves: the caller's, or their normalised integer forms (dates, times, timestamps, 128-bit integers)
    int xdt[MAXKEYS + 2], first_of[MAXKEYS];
    const int xk = norm_layout(nkeys, key_dtypes, xdt, first_of);
    if (xk < 0) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_groupby_agg_sharded: integer / date / time / timestamp / 128-bit key columns normalising to at most 8 integer columns (floating and string keys: see aqg.h)");
    if (lrc == AQG_OK && L->nkeys != xk) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_agg_sharded: internal: the normalised key layout differs from the local call's");
    int pdt[MAXPART], mop[MAXPART], sres[MAXPART], shi[MAXPART];
    for (int p = 0; p < nparts; ++p) {
        int c = parts[p].col0;
        if (parts[p].wide) { pdt[c] = AQG_UINT64; mop[c] = AQG_RED_SUM; sres[c] = p; shi[c] = 0; ++c; }      // low halves
        pdt[c] = parts[p].part_dt; mop[c] = parts[p].merge_op; sres[c] = p; shi[c] = parts[p].wide;
    }
    AQG_TRY(exchange_core(comm, L, xk, xdt, ncols, pdt, mop, row_base, max_groups_hint, gmax, sres, shi, lrc));
    aqg_groupby* M = comm->merged;
    // ---- 5. the result handle: keys and global first rows of the merged table, every aggregate in its own result dtype --------------
    aqg_groupby* H = *out ? *out : new aqg_groupby();
    H->ctx = ctx; H->n = n; H->ngroups = M->ngroups; H->nkeys = nkeys; H->nagg = naggs;
    H->has_counts = false; H->has_reversemap = false; H->sharded = true;
    const size_t GG = M->ngroups ? M->ngroups : 1;
    auto grow_h = [&](void** p, size_t* cap, size_t need) -> int { return grow(ctx, p, cap, need); };
    int rc = AQG_OK;
    H->nuser = 0;
    for (int k = 0; k < nkeys && rc == AQG_OK; ++k) {
        const int dt = key_dtypes[k], m0 = first_of[k];
        const size_t esz = dt == AQG_DATE ? 4 : dt == AQG_TIME ? 8 : dt == AQG_TIMESTAMP ? 12 : aqg_dtype_size(dt);
        H->key_dt[k] = dt;
        H->key_esz[k] = (int)esz;                              // (aqg_groupby_keys copies elements of this size: the merged keys in the CALLER's types)
        rc = grow_h(&H->keys_out[k], &H->cap_keys[k], GG * 16);
        if (rc != AQG_OK || !M->ngroups) continue;
        const unsigned grid = aqg_grid(ctx, M->ngroups, 256, 1, 8);
        if (dt == AQG_TIMESTAMP) hipLaunchKernelGGL(denorm_timestamp_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*)M->keys_out[m0], (const uint64_t*)M->keys_out[m0 + 1], M->ngroups, (uint32_t*)H->keys_out[k]);
        else if (dt == AQG_INT128 || dt == AQG_UINT128) hipLaunchKernelGGL(denorm_i128_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const uint64_t*)M->keys_out[m0], (const uint64_t*)M->keys_out[m0 + 1], M->ngroups, (uint64_t*)H->keys_out[k]);
        else rc = hipMemcpyAsync(H->keys_out[k], M->keys_out[m0], (size_t)M->ngroups * esz, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess ? AQG_OK : AQG_ERR_HIP;
    }
    if (rc == AQG_OK) rc = grow_h(reinterpret_cast<void**>(&H->first_rows64), &H->cap_first64, GG * 8);
    if (rc == AQG_OK && M->ngroups) rc = hipMemcpyAsync(H->first_rows64, comm->mres[0], (size_t)M->ngroups * 8, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess ? AQG_OK : AQG_ERR_HIP;
    FinalSpec fs;
    memset(&fs, 0, sizeof fs);
    fs.nagg = naggs;
    for (int j = 0; j < naggs && rc == AQG_OK; ++j) {
        H->res_dt[j] = aqg_reduce_out_dtype(ops[j], val_dtypes[j]);
        rc = grow_h(&H->results[j], &H->cap_results[j], GG * 16);
        fs.op[j] = ops[j]; fs.dt[j] = val_dtypes[j]; fs.out[j] = H->results[j]; fs.out_size[j] = (int)aqg_dtype_size(H->res_dt[j]);
        const Partial& P = parts[part_of[j]];
        fs.a[j] = comm->mres[1 + P.col0];
        fs.a_hi[j] = P.wide ? comm->mres[1 + P.col0 + 1] : nullptr;
        fs.b[j] = cnt_of[j] >= 0 ? comm->mres[1 + parts[cnt_of[j]].col0] : nullptr;
        if (sq_of[j] >= 0) {
            const Partial& Q = parts[sq_of[j]];
            fs.q[j] = comm->mres[1 + Q.col0];
            fs.q_hi[j] = Q.wide ? comm->mres[1 + Q.col0 + 1] : nullptr;
        }
    }
    if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(H); return rc == AQG_ERR_HIP ? aqg_fail(ctx, rc, "aqg_groupby_agg_sharded: device copy failed") : rc; }
    if (M->ngroups && naggs) hipLaunchKernelGGL(xfinal_kernel, dim3(aqg_grid(ctx, M->ngroups, 256, 1, 4)), dim3(256), 0, ctx->stream, fs, M->ngroups);
    AQG_TRY(aqg_check_launch(ctx, "xfinal_kernel"));
    *out = H;
    return AQG_OK;
}

} // extern "C"
