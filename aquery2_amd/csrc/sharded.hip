// sharded.hip -- full-column reductions and scans of a column sharded by ROW RANGE over the ranks of a communicator (SURVEY 8e:
// "Full reductions: one all-reduce of a few 16-byte values", "Prefix scans: local scan + all-gather of the block totals + fix-up",
// "Windows: a halo of w - 1 rows").  Reference semantics: server/aggregations.h:19-32,71-86,332-407 (reductions), :89-281,439-485
// (scans, windows, shifts) -- of the WHOLE column, i.e. of the shards laid end to end in rank order.
//
// Every call makes ONE all-gather of a small fixed-size record per rank {status, rows, raw moments, first / last row, the shard's
// last rows (the halo its successors need)} and every rank folds the records in rank order on the host: integer results are
// bit-identical to the single-GPU call, floating sums are the rank-order sum of the shards' sums.  A rank whose local part fails
// still joins the all-gather (status in its record) and EVERY rank returns that status.
//   reductions   local moments (aqg_stats_dev) -> all-gather -> fold
//   sums / avgs  local total -> all-gather -> aqg_scan_resume with the carry and the row offset of the earlier shards
//   mins / maxs  local min / max -> all-gather -> the chained scan seeded with the fold of the earlier shards
//   windows, shifts  all-gather of the shards' last w rows -> the unchanged single-GPU kernel over the shard, then the first w rows
//                (the only ones that see the halo) recomputed by the same kernel over [halo | first rows] and copied over
#include <vector>

#include "aqg_internal.hpp"
#include "dev_common.hpp"

namespace {

constexpr size_t HDR_WORDS = 18;                       // status, rows, 10 words of moments, first, last, tail rows, padding
constexpr size_t HDR_BYTES = HDR_WORDS * 8;
constexpr uint64_t HALO_MAX_BYTES_TOTAL = (uint64_t)1 << 30;

struct PackArgs {
    const unsigned char* moments; uint32_t moment_bytes;     // device (null: none)
    const unsigned char* x; uint32_t esz; uint32_t n; uint32_t tail;
    uint32_t status;
};
// the record of this rank: header words, then the last `tail` rows of the shard
__global__ void __launch_bounds__(256) shard_pack_kernel(PackArgs a, uint64_t* __restrict__ out) {
    unsigned char* ob = reinterpret_cast<unsigned char*>(out);
    if (blockIdx.x == 0) {
        if (threadIdx.x < HDR_WORDS) out[threadIdx.x] = threadIdx.x == 0 ? a.status : threadIdx.x == 1 ? a.n : threadIdx.x == 14 ? a.tail : 0;
        __syncthreads();
        if (a.moments) for (uint32_t b = threadIdx.x; b < a.moment_bytes; b += 256) ob[16 + b] = a.moments[b];
        if (a.n && threadIdx.x < a.esz) { ob[96 + threadIdx.x] = a.x[threadIdx.x]; ob[104 + threadIdx.x] = a.x[(size_t)(a.n - 1) * a.esz + threadIdx.x]; }
    }
    const size_t tb = (size_t)a.tail * a.esz;
    const unsigned char* src = a.x + (size_t)(a.n - a.tail) * a.esz;
    for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < tb; b += (size_t)gridDim.x * 256) ob[HDR_BYTES + b] = src[b];
}

struct Gathered {
    std::vector<uint64_t> hdr;                           // world x HDR_WORDS
    unsigned char* recv = nullptr;                       // device: world records of `rec_bytes`
    size_t rec_bytes = 0;
    int world = 0, rank = 0;
    uint64_t rows(int r) const { return hdr[(size_t)r * HDR_WORDS + 1]; }
    uint32_t tail(int r) const { return (uint32_t)hdr[(size_t)r * HDR_WORDS + 14]; }
    const uint64_t* moments(int r) const { return &hdr[(size_t)r * HDR_WORDS + 2]; }
    uint64_t first_bits(int r) const { return hdr[(size_t)r * HDR_WORDS + 12]; }
    uint64_t last_bits(int r) const { return hdr[(size_t)r * HDR_WORDS + 13]; }
    uint64_t rows_before() const { uint64_t s = 0; for (int r = 0; r < rank; ++r) s += rows(r); return s; }
    uint64_t rows_total() const { uint64_t s = 0; for (int r = 0; r < world; ++r) s += rows(r); return s; }
};

// pack this rank's record, ONE all-gather, headers to the host; a failed rank fails the call on every rank
int exchange_records(aqg_comm* comm, int local_status, const void* moments_dev, uint32_t moment_bytes, const void* x, uint32_t esz, uint32_t n, uint32_t tail_rows,
                     uint32_t tail_cap /* the same on every rank */, Gathered* g) {
    aqg_ctx* ctx = aqg_comm_ctx(comm);
    const int world = aqg_comm_world(comm);
    g->world = world; g->rank = aqg_comm_rank(comm);
    g->rec_bytes = (HDR_BYTES + (size_t)tail_cap * esz + 7) & ~(size_t)7;
    void *send, *recv;
    AQG_TRY(aqg_comm_scratch(comm, g->rec_bytes, g->rec_bytes * world, &send, &recv));
    g->recv = static_cast<unsigned char*>(recv);
    PackArgs a;
    a.moments = local_status == AQG_OK ? static_cast<const unsigned char*>(moments_dev) : nullptr; a.moment_bytes = moment_bytes;
    a.x = static_cast<const unsigned char*>(x); a.esz = esz; a.n = local_status == AQG_OK ? n : 0; a.tail = local_status == AQG_OK ? tail_rows : 0;
    a.status = (uint32_t)local_status;
    hipLaunchKernelGGL(shard_pack_kernel, dim3(aqg_grid(ctx, (uint64_t)a.tail * esz + 1, 256, 16, 4)), dim3(256), 0, ctx->stream, a, static_cast<uint64_t*>(send));
    AQG_TRY(aqg_check_launch(ctx, "shard_pack_kernel"));
    AQG_TRY(aqg_comm_allgather_internal(comm, send, recv, g->rec_bytes));
    g->hdr.resize((size_t)world * HDR_WORDS);
    for (int r = 0; r < world; ++r)
        AQG_HIP(ctx, hipMemcpyAsync(&g->hdr[(size_t)r * HDR_WORDS], g->recv + (size_t)r * g->rec_bytes, HDR_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < world; ++r) {
        const int st = (int)(uint32_t)g->hdr[(size_t)r * HDR_WORDS];
        if (st != AQG_OK) { if (r != g->rank || ctx->err.empty()) ctx->err = "sharded call: a rank failed before the exchange"; return st; }
    }
    return AQG_OK;
}

template <class T> T from_word(uint64_t w) { T v; memcpy(&v, &w, sizeof(T)); return v; }
inline __int128 i128_of(const uint64_t* w) { return (__int128)(((unsigned __int128)w[1] << 64) | w[0]); }

int esz_of(int t) { return (int)aqg_dtype_size(t); }

} // namespace

extern "C" {

// aqg_reduce over a column sharded by row range: this rank holds rows of its own (n may be 0); every rank gets the same result
int aqg_reduce_sharded(aqg_comm* comm, int op, int t, const void* x, uint32_t n, void* out_host16) {
    aqg_ctx* ctx = aqg_comm_ctx(comm);
    if (!comm || !out_host16 || (!x && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_reduce_sharded: bad argument");
    if (op < 0 || op > AQG_RED_LAST) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_reduce_sharded: bad op");
    if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "reduction: the column dtype is not numeric");
    memset(out_host16, 0, 16);
    int flags = 0;
    switch (op) {
    case AQG_RED_SUM: case AQG_RED_AVG: flags = 1; break;
    case AQG_RED_VAR: case AQG_RED_STDDEV: flags = 3; break;
    case AQG_RED_MIN: case AQG_RED_MAX: flags = 4; break;
    }
    void* mom = nullptr;
    int lrc = AQG_OK;
    if (flags && n) {
        lrc = aqg_malloc(ctx, 64, &mom);
        if (lrc == AQG_OK) lrc = aqg_stats_dev(ctx, t, x, n, flags, mom);
    }
    Gathered g;
    const int rc = exchange_records(comm, lrc, mom, 48, x, (uint32_t)esz_of(t), n, 0, 0, &g);
    if (mom) aqg_free(ctx, mom);
    AQG_TRY(rc);
    const uint64_t total = g.rows_total();
    return aqg_dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        constexpr bool FP = std::is_floating_point_v<T>;
        if (op == AQG_RED_COUNT) { memcpy(out_host16, &total, 8); return AQG_OK; }
        if (op == AQG_RED_FIRST || op == AQG_RED_LAST) {
            T v = 0;
            if (op == AQG_RED_FIRST) { for (int r = 0; r < g.world; ++r) if (g.rows(r)) { v = from_word<T>(g.first_bits(r)); break; } }
            else for (int r = g.world - 1; r >= 0; --r) if (g.rows(r)) { v = from_word<T>(g.last_bits(r)); break; }
            memcpy(out_host16, &v, sizeof(T));
            return AQG_OK;
        }
        // fold in rank order (moments: {sum 2 words, ssq 2 words, min, max})
        __int128 si = 0, qi = 0;
        double sd = 0, qd = 0;
        T mn = dlimits<T>::max(), mx = dlimits<T>::min();                  // the reference's seeds (aggregations.h:73,81)
        for (int r = 0; r < g.world; ++r) {
            if (!g.rows(r)) continue;
            const uint64_t* m = g.moments(r);
            if constexpr (FP) { double a, b; memcpy(&a, &m[0], 8); memcpy(&b, &m[2], 8); sd += a; qd += b; }
            else { si = (__int128)((unsigned __int128)si + (unsigned __int128)i128_of(m)); qi = (__int128)((unsigned __int128)qi + (unsigned __int128)i128_of(m + 2)); }
            const T a = from_word<T>(m[4]), b = from_word<T>(m[5]);
            mn = a < mn ? a : mn; mx = b > mx ? b : mx;
        }
        switch (op) {
        case AQG_RED_SUM: if constexpr (FP) memcpy(out_host16, &sd, 8); else memcpy(out_host16, &si, 16); break;
        case AQG_RED_MIN: memcpy(out_host16, &mn, sizeof(T)); break;
        case AQG_RED_MAX: memcpy(out_host16, &mx, sizeof(T)); break;
        case AQG_RED_AVG: {
            double d;
            if constexpr (FP) d = sd / (double)total;
            else if constexpr (std::is_unsigned_v<T>) d = (double)(unsigned __int128)si / (double)total;
            else d = (double)si / (double)total;
            memcpy(out_host16, &d, 8);
        } break;
        case AQG_RED_VAR: case AQG_RED_STDDEV: {                             // (ssq - s * s / (len + 1)) / (len + 1): D9 kept; len is the whole column's
            const double np1 = total < 0xFFFFFFFFull ? (double)(uint32_t)(total + 1) : (double)(total + 1);
            double d;
            if constexpr (FP) d = (qd - sd * sd / np1) / np1;
            else if constexpr (std::is_unsigned_v<T>) { const unsigned __int128 s = (unsigned __int128)si; d = ((double)(unsigned __int128)qi - (double)(s * s) / np1) / np1; }
            else { const __int128 ss = (__int128)((unsigned __int128)si * (unsigned __int128)si); d = ((double)qi - (double)ss / np1) / np1; }
            if (op == AQG_RED_STDDEV) d = sqrt(d);
            memcpy(out_host16, &d, 8);
        } break;
        }
        return AQG_OK;
    });
}

// corr(x, y) over two columns sharded alike (aggregations.h:383-407): the five 128-bit sums of every shard, folded exactly
int aqg_corr_sharded(aqg_comm* comm, int tx, const void* x, int ty, const void* y, uint32_t n, double* out_host) {
    aqg_ctx* ctx = aqg_comm_ctx(comm);
    if (!comm || !out_host || ((!x || !y) && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_corr_sharded: bad argument");
    void* mom = nullptr;
    int lrc = AQG_OK;
    if (n) {
        lrc = aqg_malloc(ctx, 96, &mom);
        if (lrc == AQG_OK) lrc = aqg_corr_sums_dev(ctx, tx, x, ty, y, n, mom);
    } else if (dt_is_fp(tx) || dt_is_fp(ty)) lrc = aqg_fail(ctx, AQG_ERR_DTYPE, "corr: integer columns");
    Gathered g;
    const int rc = exchange_records(comm, lrc, mom, 80, x, 1, n, 0, 0, &g);
    if (mom) aqg_free(ctx, mom);
    AQG_TRY(rc);
    unsigned __int128 s[5] = {0, 0, 0, 0, 0};                                // sx, sy, sxy, sx2, sy2
    for (int r = 0; r < g.world; ++r) if (g.rows(r)) for (int k = 0; k < 5; ++k) s[k] += (unsigned __int128)i128_of(g.moments(r) + 2 * k);
    const uint64_t len = g.rows_total();
    auto mulw = [](__int128 a, __int128 b) { return (__int128)((unsigned __int128)a * (unsigned __int128)b); };
    const __int128 sx = (__int128)s[0], sy = (__int128)s[1], sxy = (__int128)s[2], sx2 = (__int128)s[3], sy2 = (__int128)s[4];
    *out_host = ((double)mulw((__int128)len, sxy) - (double)mulw(sx, sy)) /
                sqrt(((double)mulw((__int128)len, sx2) - (double)mulw(sx, sx)) * ((double)mulw((__int128)len, sy2) - (double)mulw(sy, sy)));
    return AQG_OK;
}

// aqg_scan over a column sharded by row range: out = this rank's rows of the scan of the WHOLE column
int aqg_scan_sharded(aqg_comm* comm, int op, int t, const void* x, uint32_t n, uint32_t w, void* out) {
    aqg_ctx* ctx = aqg_comm_ctx(comm);
    if (!comm || (!x && n) || (!out && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan_sharded: bad argument");
    if (op < 0 || op > AQG_SCAN_STDDEVW) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan_sharded: bad op");
    if (w == 0 && (op == AQG_SCAN_SUMW || op == AQG_SCAN_AVGW || op == AQG_SCAN_VARW || op == AQG_SCAN_STDDEVW))
        return aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan_sharded: window 0 is undefined for sumw/avgw/varw");
    if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "scan: the column dtype is not numeric");
    if (op == AQG_SCAN_VARS || op == AQG_SCAN_STDDEVS) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_scan_sharded: vars / stddevs are not offered over shards (parity unpinned in the reference: D9)");
    const uint32_t esz = (uint32_t)esz_of(t);
    const int world = aqg_comm_world(comm);
    const bool prefix_sum = op == AQG_SCAN_SUMS || op == AQG_SCAN_AVGS;
    const bool prefix_mm = op == AQG_SCAN_MINS || op == AQG_SCAN_MAXS;
    const bool window = op == AQG_SCAN_SUMW || op == AQG_SCAN_AVGW || op == AQG_SCAN_VARW || op == AQG_SCAN_STDDEVW || op == AQG_SCAN_MINW || op == AQG_SCAN_MAXW || op == AQG_SCAN_RATIOW;
    // the halo a shard's successors need: w rows for the windows (w - 1 reach back, ratiow's divisor lies w back), 1 for the shifts
    uint32_t tail_cap = window ? w : (prefix_sum || prefix_mm) ? 0u : 1u;
    const bool running_mm = (op == AQG_SCAN_MINW || op == AQG_SCAN_MAXW) && (w == 0 || (uint64_t)tail_cap * esz * world > HALO_MAX_BYTES_TOTAL);
    if (running_mm) tail_cap = 0;                                              // (only right when w >= the whole column: checked below)
    if ((uint64_t)tail_cap * esz * world > HALO_MAX_BYTES_TOTAL) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan_sharded: the window's halo exceeds 1 GiB over the ranks");
    // ---- local moments the exchange ships ---------------------------------------------------------------------------------------
    void* mom = nullptr;
    int lrc = AQG_OK;
    const int flags = prefix_sum ? 1 : (prefix_mm || running_mm) ? 4 : 0;
    // the running max a window degrades to carries no numeric_limits seed, but the moments' max does (aggregations.h:73: for floating
    // columns the smallest POSITIVE value): such a shard's true maximum is the last element of its own running max
    const bool true_max = running_mm && op == AQG_SCAN_MAXW && dt_is_fp(t);
    if (flags && n) {
        lrc = aqg_malloc(ctx, 64, &mom);
        if (lrc == AQG_OK) lrc = aqg_stats_dev(ctx, t, x, n, flags, mom);
        if (lrc == AQG_OK && true_max) {
            lrc = aqg_scan_minmax_seeded(ctx, op, t, x, n, nullptr, out);
            if (lrc == AQG_OK) lrc = aqg_memset(ctx, static_cast<char*>(mom) + 40, 0, 8);
            if (lrc == AQG_OK) lrc = aqg_d2d(ctx, static_cast<char*>(mom) + 40, static_cast<const char*>(out) + (size_t)(n - 1) * esz, esz);
        }
    }
    Gathered g;
    const uint32_t my_tail = n < tail_cap ? n : tail_cap;
    const int xrc = exchange_records(comm, lrc, mom, 48, x, esz, n, my_tail, tail_cap, &g);
    if (mom) aqg_free(ctx, mom);
    AQG_TRY(xrc);
    const uint64_t before = g.rows_before(), total = g.rows_total();
    if (running_mm && w != 0 && w < total) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan_sharded: the window's halo exceeds 1 GiB over the ranks");
    if (n == 0) return AQG_OK;
    // ---- prefix scans: carry of the earlier shards ---------------------------------------------------------------------------------
    if (prefix_sum) {
        unsigned char carry[16] = {0};
        if (dt_is_fp(t)) { double s = -0.0; bool any = false; for (int r = 0; r < g.rank; ++r) if (g.rows(r)) { double a; memcpy(&a, g.moments(r), 8); s = any ? s + a : a; any = true; } memcpy(carry, &s, 8); }
        else { unsigned __int128 s = 0; for (int r = 0; r < g.rank; ++r) if (g.rows(r)) s += (unsigned __int128)i128_of(g.moments(r)); memcpy(carry, &s, 16); }
        return aqg_scan_resume(ctx, op, t, x, n, before ? carry : nullptr, before, out);
    }
    if (prefix_mm || running_mm) {
        unsigned char seed[8] = {0};
        bool any = false;
        const int rc = aqg_dispatch_num(t, [&](auto tt) -> int {
            using T = typename decltype(tt)::type;
            const bool is_max = op == AQG_SCAN_MAXS || op == AQG_SCAN_MAXW;
            T acc{};
            for (int r = 0; r < g.rank; ++r) {
                if (!g.rows(r)) continue;
                const T v = from_word<T>(g.moments(r)[is_max ? 5 : 4]);
                acc = !any ? v : is_max ? (v > acc ? v : acc) : (v < acc ? v : acc);
                any = true;
            }
            memcpy(seed, &acc, sizeof(T));
            return AQG_OK;
        });
        AQG_TRY(rc);
        if (true_max) return any ? aqg_scan_minmax_seeded(ctx, op, t, out, n, seed, out) : AQG_OK;      // (out is the shard's own running max already: max with the seed, in place)
        return aqg_scan_minmax_seeded(ctx, op, t, x, n, any ? seed : nullptr, out);
    }
    // ---- windows and shifts: the single-GPU kernel over the shard, then the rows that see the halo ------------------------------------
    uint32_t ww = w;
    if (op == AQG_SCAN_RATIOW && total <= w) ww = w ? 1 : 0;                    // aggregations.h:172-175 with the WHOLE column's length
    AQG_TRY(aqg_scan(ctx, op, t, x, n, ww, out));
    const size_t osz = aqg_dtype_size(aqg_scan_out_dtype(op, t));
    if (op == AQG_SCAN_NEXT) {                                                  // the last row takes the first row of the next non-empty shard
        int nx = -1;
        for (int r = g.rank + 1; r < g.world; ++r) if (g.rows(r)) { nx = r; break; }
        if (nx < 0) return AQG_OK;
        const uint64_t first = g.first_bits(nx);
        unsigned char v[8];
        memcpy(v, &first, 8);
        void* st;
        AQG_TRY(aqg_host_stage(ctx, 16, &st));
        memcpy(st, v, esz);
        AQG_HIP(ctx, hipMemcpyAsync(static_cast<char*>(out) + (size_t)(n - 1) * esz, st, esz, hipMemcpyHostToDevice, ctx->stream));
        AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return AQG_OK;
    }
    const uint32_t reach = window ? ww : 1u;                                   // rows of halo that matter / rows of the shard that see it
    if (before == 0 || reach == 0) return AQG_OK;
    const uint32_t h = before < reach ? (uint32_t)before : reach;
    const uint32_t m = n < reach ? n : reach;
    // ratiow keeps its window only while the column is longer than it: pad the small column beyond w when the whole column is
    const uint32_t pad = (op == AQG_SCAN_RATIOW && total > ww && (uint64_t)h + m <= ww) ? ww + 1 - (h + m) : 0u;
    const size_t L = (size_t)h + m + pad;
    void *small = nullptr, *sout = nullptr;
    AQG_TRY(aqg_malloc(ctx, L * esz + 64, &small));
    int rc = aqg_malloc(ctx, L * osz + 64, &sout);
    if (rc == AQG_OK && pad) rc = aqg_memset(ctx, static_cast<char*>(small) + ((size_t)h + m) * esz, 1, (size_t)pad * esz);
    // the halo: the last rows of the predecessors, nearest first
    uint32_t need = h;
    for (int r = g.rank - 1; r >= 0 && need && rc == AQG_OK; --r) {
        const uint32_t have = g.tail(r);
        const uint32_t take = have < need ? have : need;
        if (!take) continue;
        const unsigned char* src = g.recv + (size_t)r * g.rec_bytes + HDR_BYTES + (size_t)(have - take) * esz;
        rc = aqg_d2d(ctx, static_cast<char*>(small) + (size_t)(need - take) * esz, src, (size_t)take * esz);
        need -= take;
    }
    if (rc == AQG_OK && need) rc = aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan_sharded: internal: the gathered tails do not cover the halo");
    if (rc == AQG_OK) rc = aqg_d2d(ctx, static_cast<char*>(small) + (size_t)h * esz, x, (size_t)m * esz);
    if (rc == AQG_OK) rc = aqg_scan(ctx, op, t, small, (uint32_t)L, ww, sout);
    if (rc == AQG_OK) rc = aqg_d2d(ctx, out, static_cast<char*>(sout) + (size_t)h * osz, (size_t)m * osz);
    if (rc == AQG_OK) rc = aqg_sync(ctx);
    aqg_free(ctx, small);
    if (sout) aqg_free(ctx, sout);
    return rc;
}

} // extern "C"
