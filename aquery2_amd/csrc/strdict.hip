// strdict.hip -- aqg_str_encode: dictionary codes of an astring_view key column (reference server/types.h:281-334: equality = the
// content of the NUL-terminated strings; hash = the string view, server/hasher.h:97-106; mem_opt.cpp:22 groups by a string column
// and h2o's id1 .. id3 are strings).  codes[i] = dense id of row i's string in FIRST-OCCURRENCE order: the uint32 column the group-by /
// join entry points take as a key.
//
// The strings live in host memory behind n pointers.  Small inputs take a host hash map.  From 2^16 rows on the dictionary is built on
// the DEVICE: the host only walks the strings once -- several threads, lengths then bytes into ONE concatenated buffer -- and uploads
// bytes + offsets; a kernel hashes every row's bytes to 64 bits, the ordinary group-by build over {hash, length} numbers the rows by
// first occurrence (its group ids ARE the codes), and a second kernel compares every row's bytes with its group's first row: a 64-bit
// hash collision between different strings of one length (never seen; ~n^2 / 2^65) sends the call back to the host map.
// 1e8 short strings: see DESIGN.md (the host map inserts ~1e7 rows/s on one core).
#include <string_view>
#include <thread>
#include <unordered_map>

#include "aqg_internal.hpp"
#include "dev_common.hpp"

namespace {

int encode_on_host(aqg_ctx* ctx, const char* const* strs_host, uint32_t n, uint32_t* codes_dev, uint32_t* ndistinct_host, std::vector<uint32_t>* first_rows = nullptr) {
    std::unordered_map<std::string_view, uint32_t> dict;
    dict.reserve(1024);
    std::vector<uint32_t> codes(n);
    if (first_rows) first_rows->clear();
    for (uint32_t i = 0; i < n; ++i) {
        const char* p = strs_host[i] ? strs_host[i] : "";
        auto ins = dict.try_emplace(std::string_view(p), (uint32_t)dict.size());
        if (ins.second && first_rows) first_rows->push_back(i);
        codes[i] = ins.first->second;
    }
    if (ndistinct_host) *ndistinct_host = (uint32_t)dict.size();
    if (n) { AQG_HIP(ctx, hipMemcpyAsync(codes_dev, codes.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream)); AQG_HIP(ctx, hipStreamSynchronize(ctx->stream)); }
    return AQG_OK;
}

// 64-bit hash of row i's bytes (a multiply-xorshift per 8-byte word, the tail zero-padded) and its length
__global__ void __launch_bounds__(256) str_hash_kernel(const unsigned char* __restrict__ bytes, const uint64_t* __restrict__ off, uint32_t n,
                                                       uint64_t* __restrict__ h64, uint32_t* __restrict__ len32) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t b = off[i], e = off[i + 1];
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (e - b);
        for (uint64_t p = b; p < e; p += 8) {
            uint64_t w = 0;
            const uint32_t m = e - p < 8 ? (uint32_t)(e - p) : 8u;
            for (uint32_t k = 0; k < m; ++k) w |= (uint64_t)bytes[p + k] << (8 * k);
            h = (h ^ w) * 0xD6E8FEB86659FD93ull;
            h ^= h >> 32;
        }
        h *= 0xD6E8FEB86659FD93ull;
        h ^= h >> 29;
        h64[i] = h;
        len32[i] = (uint32_t)(e - b);
    }
}
// rows whose bytes differ from their group's first row (same hash and length, other content)
__global__ void __launch_bounds__(256) str_verify_kernel(const unsigned char* __restrict__ bytes, const uint64_t* __restrict__ off, uint32_t n,
                                                         const uint32_t* __restrict__ gid, const uint32_t* __restrict__ first_rows, uint32_t* __restrict__ mismatches) {
    uint32_t bad = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t r = first_rows[gid[i]];
        if (r == i) continue;
        const uint64_t a = off[i], b = off[r], len = off[i + 1] - a;
        for (uint64_t k = 0; k < len; ++k) if (bytes[a + k] != bytes[b + k]) { bad = 1; break; }
    }
    if (bad) atomicAdd(mismatches, 1u);
}

__global__ void __launch_bounds__(256) remap_codes_kernel(uint32_t* __restrict__ codes, uint32_t n, const uint32_t* __restrict__ remap) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) codes[i] = remap[codes[i]];
}
int str_encode_impl(aqg_ctx* ctx, const char* const* strs_host, uint32_t n, uint32_t* codes_dev, uint32_t* ndistinct_host, std::vector<uint32_t>* first_rows);

} // namespace

extern "C" int aqg_str_encode(aqg_ctx* ctx, const char* const* strs_host, uint32_t n, uint32_t* codes_dev, uint32_t* ndistinct_host) {
    return str_encode_impl(ctx, strs_host, n, codes_dev, ndistinct_host, nullptr);
}

// astring_view keys of a table sharded by ROW RANGE: codes of ONE dictionary over all shards, in GLOBAL first-occurrence order (shards
// are contiguous row ranges in rank order, so walking the ranks' dictionaries in rank order, each in its own first-occurrence order, is
// the global first occurrence) -- the uint32 key column aqg_groupby_agg_sharded takes.  Every rank encodes its rows (aqg_str_encode), the
// ranks all-gather their dictionaries' strings (sizes first, then the padded bytes: two small collectives), build the merged dictionary on
// the host and remap their code columns on the device.
extern "C" int aqg_str_encode_sharded(aqg_comm* comm, const char* const* strs_host, uint32_t n, uint32_t* codes_dev, uint32_t* ndistinct_global_host) {
    aqg_ctx* ctx = aqg_comm_ctx(comm);
    if (!comm || (!strs_host && n) || (!codes_dev && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_str_encode_sharded: bad argument");
    const int world = aqg_comm_world(comm), rank = aqg_comm_rank(comm);
    std::vector<uint32_t> first;
    uint32_t nd = 0;
    int lrc = str_encode_impl(ctx, strs_host, n, codes_dev, &nd, &first);
    // ---- this rank's dictionary: {count, bytes} then the strings, NUL-separated ------------------------------------------------------
    std::vector<unsigned char> mine;
    if (lrc == AQG_OK) for (uint32_t c = 0; c < nd; ++c) { const char* p = strs_host[first[c]] ? strs_host[first[c]] : ""; mine.insert(mine.end(), p, p + strlen(p) + 1); }
    uint64_t hdr[4] = {(uint64_t)(uint32_t)lrc, nd, mine.size(), 0};
    void *send, *recv;
    AQG_TRY(aqg_comm_scratch(comm, 32, (size_t)32 * world, &send, &recv));
    AQG_TRY(aqg_h2d(ctx, send, hdr, 32));
    AQG_TRY(aqg_comm_allgather_internal(comm, send, recv, 32));
    std::vector<uint64_t> all((size_t)4 * world);
    AQG_TRY(aqg_d2h(ctx, all.data(), recv, (size_t)32 * world));
    size_t maxb = 8;
    for (int r = 0; r < world; ++r) {
        if ((uint32_t)all[4 * r]) { if (r != rank || ctx->err.empty()) ctx->err = "sharded call: a rank failed before the exchange"; return (int)(uint32_t)all[4 * r]; }
        maxb = all[4 * r + 2] > maxb ? all[4 * r + 2] : maxb;
    }
    maxb = (maxb + 7) & ~(size_t)7;
    AQG_TRY(aqg_comm_scratch(comm, maxb, maxb * world, &send, &recv));
    if (!mine.empty()) AQG_TRY(aqg_h2d(ctx, send, mine.data(), mine.size()));
    AQG_TRY(aqg_comm_allgather_internal(comm, send, recv, maxb));
    std::vector<unsigned char> dicts(maxb * world);
    AQG_TRY(aqg_d2h(ctx, dicts.data(), recv, maxb * world));
    // ---- the merged dictionary, rank order ------------------------------------------------------------------------------------------------
    std::unordered_map<std::string_view, uint32_t> global;
    std::vector<uint32_t> remap(nd ? nd : 1);
    for (int r = 0; r < world; ++r) {
        const char* p = reinterpret_cast<const char*>(dicts.data() + (size_t)r * maxb);
        for (uint64_t c = 0; c < all[4 * r + 1]; ++c) {
            const std::string_view sv(p);
            const uint32_t id = global.try_emplace(sv, (uint32_t)global.size()).first->second;
            if (r == rank) remap[c] = id;
            p += sv.size() + 1;
        }
    }
    if (ndistinct_global_host) *ndistinct_global_host = (uint32_t)global.size();
    if (n) {
        void* dremap = nullptr;
        AQG_TRY(aqg_malloc(ctx, (size_t)(nd ? nd : 1) * 4, &dremap));
        int rc = aqg_h2d(ctx, dremap, remap.data(), (size_t)nd * 4);
        if (rc == AQG_OK) {
            hipLaunchKernelGGL(remap_codes_kernel, dim3(aqg_grid(ctx, n, 256, 4, 16)), dim3(256), 0, ctx->stream, codes_dev, n, static_cast<const uint32_t*>(dremap));
            rc = aqg_check_launch(ctx, "remap_codes_kernel");
        }
        if (rc == AQG_OK) rc = aqg_sync(ctx);
        aqg_free(ctx, dremap);
        AQG_TRY(rc);
    }
    return AQG_OK;
}

namespace {
int str_encode_impl(aqg_ctx* ctx, const char* const* strs_host, uint32_t n, uint32_t* codes_dev, uint32_t* ndistinct_host, std::vector<uint32_t>* first_rows) {
    if (!ctx || (!strs_host && n) || (!codes_dev && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_str_encode: bad argument");
    static const bool host_only = getenv("AQG_STR_HOST") != nullptr;          // A/B measurements only
    static const uint32_t dev_min = getenv("AQG_STR_DEVICE_MIN") ? (uint32_t)atoi(getenv("AQG_STR_DEVICE_MIN")) : (1u << 16);
    if (host_only || n < dev_min) return encode_on_host(ctx, strs_host, n, codes_dev, ndistinct_host, first_rows);
    // ---- host: one walk over the strings, in parallel: lengths, offsets, bytes ---------------------------------------------------
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : nt > 16 ? 16 : nt;
    std::vector<uint64_t> off((size_t)n + 1);
    std::vector<uint32_t> len(n);
    std::vector<uint64_t> part(nt + 1, 0);
    auto span = [&](unsigned t, uint32_t& lo, uint32_t& hi) { lo = (uint32_t)((uint64_t)n * t / nt); hi = (uint32_t)((uint64_t)n * (t + 1) / nt); };
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] {
            uint32_t lo, hi; span(t, lo, hi);
            uint64_t s = 0;
            for (uint32_t i = lo; i < hi; ++i) { const size_t l = strs_host[i] ? strlen(strs_host[i]) : 0; len[i] = (uint32_t)l; s += l; }
            part[t + 1] = s;
        });
        for (auto& x : th) x.join();
    }
    for (unsigned t = 0; t < nt; ++t) part[t + 1] += part[t];
    const uint64_t total = part[nt];
    std::vector<unsigned char> bytes(total ? total : 1);
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] {
            uint32_t lo, hi; span(t, lo, hi);
            uint64_t o = part[t];
            for (uint32_t i = lo; i < hi; ++i) {
                const uint64_t l = len[i];
                off[i] = o;
                if (l) memcpy(&bytes[o], strs_host[i], l);
                o += l;
            }
        });
        for (auto& x : th) x.join();
    }
    // (every thread wrote off[lo .. hi): the starts; the last slot is the end)
    off[n] = total;
    // ---- device: hash, number by first occurrence, verify ---------------------------------------------------------------------------
    void *dbytes = nullptr, *doff = nullptr, *dh = nullptr, *dlen = nullptr, *dmis = nullptr;
    aqg_groupby* g = nullptr;
    int rc = aqg_malloc(ctx, total + 16, &dbytes);
    if (rc == AQG_OK) rc = aqg_malloc(ctx, ((size_t)n + 1) * 8, &doff);
    if (rc == AQG_OK) rc = aqg_malloc(ctx, (size_t)n * 8 + 16, &dh);
    if (rc == AQG_OK) rc = aqg_malloc(ctx, (size_t)n * 4 + 16, &dlen);
    if (rc == AQG_OK) rc = aqg_malloc(ctx, 16, &dmis);
    if (rc == AQG_OK) rc = aqg_h2d(ctx, dbytes, bytes.data(), total);
    if (rc == AQG_OK) rc = aqg_h2d(ctx, doff, off.data(), ((size_t)n + 1) * 8);
    if (rc == AQG_OK) rc = aqg_memset(ctx, dmis, 0, 16);
    uint32_t mism = 0, G = 0;
    if (rc == AQG_OK) {
        const unsigned grid = aqg_grid(ctx, n, 256, 2, 16);
        hipLaunchKernelGGL(str_hash_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const unsigned char*>(dbytes), static_cast<const uint64_t*>(doff), n,
                           static_cast<uint64_t*>(dh), static_cast<uint32_t*>(dlen));
        rc = aqg_check_launch(ctx, "str_hash_kernel");
        const int dts[2] = {AQG_UINT64, AQG_UINT32};
        const void* cols[2] = {dh, dlen};
        if (rc == AQG_OK) rc = aqg_groupby_build(ctx, 2, dts, cols, n, 0, &g);
        if (rc == AQG_OK) {
            G = aqg_groupby_ngroups(g);
            hipLaunchKernelGGL(str_verify_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const unsigned char*>(dbytes), static_cast<const uint64_t*>(doff), n,
                               aqg_groupby_reversemap(g), aqg_groupby_first_rows(g), static_cast<uint32_t*>(dmis));
            rc = aqg_check_launch(ctx, "str_verify_kernel");
        }
        if (rc == AQG_OK) rc = aqg_d2h(ctx, &mism, dmis, 4);
        if (rc == AQG_OK && !mism) rc = aqg_d2d(ctx, codes_dev, aqg_groupby_reversemap(g), (size_t)n * 4);
        if (rc == AQG_OK && !mism && first_rows) { first_rows->resize(G); rc = aqg_d2h(ctx, first_rows->data(), aqg_groupby_first_rows(g), (size_t)G * 4); }
        if (rc == AQG_OK) rc = aqg_sync(ctx);
    }
    if (g) aqg_groupby_destroy(g);
    for (void* p : {dbytes, doff, dh, dlen, dmis}) if (p) aqg_free(ctx, p);
    if (rc != AQG_OK) return rc;
    if (mism) return encode_on_host(ctx, strs_host, n, codes_dev, ndistinct_host, first_rows);      // different strings under one 64-bit hash and length
    if (ndistinct_host) *ndistinct_host = G;
    return AQG_OK;
}
} // namespace
