// partition.hip -- high-cardinality group-by without global atomics.
//
// Scattered device-scope atomics top out near 3e10 per second on MI355X (measured: h2o Q5, 1e9 rows, 1e7 groups:
// 140-170 ms through the HBM table), so groups that do not fit one workgroup's LDS table are handled by
// PARTITIONING the rows instead: records {packed key, row id, value columns} are radix-partitioned on hash bits
// (7 bits per pass, stable scatter with wavefront match-any ranks, runs of >= 128 B per bin and tile) until a
// partition holds about a thousand groups; each partition is then aggregated in LDS by one workgroup and its
// groups are appended to a compact record table.  The compact table feeds the same collect / rank / emit
// kernels as the hash path (first-occurrence order comes from the carried row ids).
// Traffic for Q5 (16 B/row algorithmic): key count pass 4 + two passes (read 20 + write 24 / read 24 + write 24)
// + aggregate 24 = ~120 B/row, all streaming.
#include "groupby_dev.hpp"

namespace {

constexpr int PB = 256;
constexpr int PROUNDS = 32;
constexpr int PT = PB * PROUNDS;   // rows per tile (8192): ~32 rows = 128 B per bin and tile at 256 bins
constexpr int MAXPAY = MAXACC + 2;

struct KeyIn {            // where a pass reads the packed key of row i from
    int from_cols;        // 1: pack from the user's key columns; 0: record array
    KeySpec ks;
    const void* rec;      // record keys
    int ksz;              // 4 or 8 bytes per record key
};
__device__ inline uint64_t read_key(const KeyIn& k, size_t i) {
    if (k.from_cols) return pack_key(k.ks, i);
    return k.ksz == 4 ? (uint64_t) static_cast<const uint32_t*>(k.rec)[i] : static_cast<const uint64_t*>(k.rec)[i];
}
struct Payload {          // columns carried along: [0] = key, [1] = row id, [2..] = value columns
    int ncols;
    const void* in[MAXPAY];
    void* out[MAXPAY];
    int esz[MAXPAY];
};
__device__ inline uint32_t part_hash(uint64_t key) { return hash64(key * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull); }

// ---- MSD partitioning, up to two levels of <= 8 hash bits ---------------------------------------------------------------
// A level splits every SEGMENT of the current record set (level 1: the whole input = one segment; level 2: each level-1 bin)
// into 2^bits bins.  Tiles of PT rows never straddle segments: tile t belongs to segment b = upper_bound(tile_prefix, t) - 1.
// hist layout [segment][bin][tile in segment]: one global exclusive scan of it yields absolute destinations, because the
// counts of a segment add up to its length.  Ranks inside a tile come from returning LDS atomics (order inside a bin is
// irrelevant for aggregation; row ids travel with the records).
struct Segs {
    const uint32_t* seg_start;    // [nseg+1] row range of each segment
    const uint32_t* tile_prefix;  // [nseg+1] first tile of each segment
    uint32_t nseg;
};
__device__ inline bool tile_range(const Segs& sg, uint32_t t, uint32_t& seg, uint32_t& tin, uint32_t& ntseg, uint32_t& rb, uint32_t& re) {
    if (t >= sg.tile_prefix[sg.nseg]) return false;
    uint32_t lo = 0, hi = sg.nseg;                       // largest seg with tile_prefix[seg] <= t
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (sg.tile_prefix[mid] <= t) lo = mid; else hi = mid; }
    seg = lo;
    tin = t - sg.tile_prefix[seg];
    ntseg = sg.tile_prefix[seg + 1] - sg.tile_prefix[seg];
    rb = sg.seg_start[seg] + tin * PT;
    re = rb + PT < sg.seg_start[seg + 1] ? rb + PT : sg.seg_start[seg + 1];
    return true;
}

__global__ void __launch_bounds__(PB) part_hist_kernel(KeyIn kin, Segs sg, uint32_t shift, uint32_t bits, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    const uint32_t nb = 1u << bits;
    uint32_t seg, tin, ntseg, rb, re;
    if (!tile_range(sg, blockIdx.x, seg, tin, ntseg, rb, re)) return;
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t p = rb + threadIdx.x; p < re; p += PB) atomicAdd(&h[(part_hash(read_key(kin, p)) >> shift) & (nb - 1)], 1u);
    __syncthreads();
    if (threadIdx.x < nb) hist[(size_t)sg.tile_prefix[seg] * nb + (size_t)threadIdx.x * ntseg + tin] = h[threadIdx.x];
}

// Scatter with LDS-staged, coalesced writes: a tile's rows are ranked inside their bins (returning LDS atomics), laid out
// bin-major in LDS one column at a time, and streamed out so that consecutive lanes write consecutive addresses of a bin's run
// (~32 rows = 128 B per bin and tile).  Scattering straight from registers costs 34-68 ms per level at 1e9 rows (partial-line
// writes); staged it is a streaming copy.
template <bool FIRST>
__global__ void __launch_bounds__(PB) part_scatter_kernel(KeyIn kin, Payload pay, Segs sg, uint32_t shift, uint32_t bits, const uint32_t* __restrict__ hist_scanned) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint64_t* stage = reinterpret_cast<uint64_t*>(smem_raw);                 // PT slots of 8 bytes
    uint8_t* binid = reinterpret_cast<uint8_t*>(smem_raw) + (size_t)PT * 8;   // bin of every staged position
    __shared__ uint32_t gbase[256], cnt[256], lbase[256], wsum[4];
    const uint32_t nb = 1u << bits;
    uint32_t seg, tin, ntseg, rb, re;
    if (!tile_range(sg, blockIdx.x, seg, tin, ntseg, rb, re)) return;
    const uint32_t nrows = re - rb;
    gbase[threadIdx.x] = threadIdx.x < nb ? hist_scanned[(size_t)sg.tile_prefix[seg] * nb + (size_t)threadIdx.x * ntseg + tin] : 0;
    cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t pd[PROUNDS];                                                     // (digit << 16) | rank, later (digit << 16) | position
#pragma unroll
    for (int r = 0; r < PROUNDS; ++r) {
        const uint32_t j = r * PB + threadIdx.x;
        if (j < nrows) {
            const uint32_t d = (part_hash(read_key(kin, rb + j)) >> shift) & (nb - 1);
            pd[r] = (d << 16) | atomicAdd(&cnt[d], 1u);
        }
    }
    __syncthreads();
    {   // exclusive scan of the 256 bin counts
        const uint32_t c = cnt[threadIdx.x];
        const uint32_t incl = wave_scan_incl(c, OpAdd{}, lane_id());
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave_id(); ++w) base += wsum[w];
        lbase[threadIdx.x] = base + incl - c;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PROUNDS; ++r) {
        const uint32_t j = r * PB + threadIdx.x;
        if (j < nrows) {
            const uint32_t d = pd[r] >> 16, pos = lbase[d] + (pd[r] & 0xFFFF);
            pd[r] = (d << 16) | pos;
            binid[pos] = (uint8_t)d;
        }
    }
    for (int c = 0; c < pay.ncols; ++c) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PROUNDS; ++r) {
            const uint32_t j = r * PB + threadIdx.x;
            if (j < nrows) {
                const uint32_t p = rb + j;
                uint64_t v;
                if (c == 0) v = read_key(kin, p);
                else if (c == 1) v = FIRST ? (uint64_t)p : (uint64_t) static_cast<const uint32_t*>(pay.in[1])[p];
                else switch (pay.esz[c]) {
                    case 1: v = static_cast<const uint8_t*>(pay.in[c])[p]; break;
                    case 2: v = static_cast<const uint16_t*>(pay.in[c])[p]; break;
                    case 4: v = static_cast<const uint32_t*>(pay.in[c])[p]; break;
                    default: v = static_cast<const uint64_t*>(pay.in[c])[p]; break;
                }
                stage[pd[r] & 0xFFFF] = v;
            }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < nrows; j += PB) {
            const uint32_t d = binid[j];
            const size_t dst = (size_t)gbase[d] + (j - lbase[d]);
            const uint64_t v = stage[j];
            switch (pay.esz[c]) {
            case 1: static_cast<uint8_t*>(pay.out[c])[dst] = (uint8_t)v; break;
            case 2: static_cast<uint16_t*>(pay.out[c])[dst] = (uint16_t)v; break;
            case 4: static_cast<uint32_t*>(pay.out[c])[dst] = (uint32_t)v; break;
            default: static_cast<uint64_t*>(pay.out[c])[dst] = v; break;
            }
        }
    }
}

// after a level: start row of every bin of every segment = scanned count of its first tile (or the segment's end when empty);
// these become the next level's segments (or the final partitions).  One thread per (segment, bin).
__global__ void __launch_bounds__(256) bins_to_segments_kernel(Segs sg, uint32_t bits, const uint32_t* __restrict__ hist_scanned, uint32_t n,
                                                               uint32_t* __restrict__ out_start /* [nseg << bits | +1] */) {
    const uint32_t nb = 1u << bits, total = sg.nseg << bits;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= total; i += gridDim.x * blockDim.x) {
        if (i == total) { out_start[i] = n; continue; }
        uint32_t seg = i >> bits, d = i & (nb - 1);
        uint32_t ntseg = sg.tile_prefix[seg + 1] - sg.tile_prefix[seg];
        // empty segment: every bin starts (and ends) at the segment's start
        out_start[i] = ntseg ? hist_scanned[(size_t)sg.tile_prefix[seg] * nb + (size_t)d * ntseg] : sg.seg_start[seg];
    }
}
// tile_prefix[s] = number of tiles of the segments before s (single workgroup; nseg <= 65536)
__global__ void __launch_bounds__(1024) tile_prefix_kernel(const uint32_t* __restrict__ seg_start, uint32_t nseg, uint32_t* __restrict__ tile_prefix) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base <= nseg; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < nseg ? (seg_start[i + 1] - seg_start[i] + PT - 1) / PT : 0;
        uint32_t incl = wave_scan_incl(v, OpAdd{}, lane_id());
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        uint32_t wbase = carry;
        for (int w = 0; w < wave_id(); ++w) wbase += wsum[w];
        if (i <= nseg) tile_prefix[i] = wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = wbase + incl;
        __syncthreads();
    }
}

// one workgroup per partition (grid-stride): LDS open addressing {key64, first_row, count, acc...}; groups are appended to `out`
template <int NACC>
__global__ void __launch_bounds__(PB) part_agg_kernel(const void* __restrict__ rkeys, int ksz, const uint32_t* __restrict__ rrows, AccSpec as,
                                                      const uint32_t* __restrict__ pstart, uint32_t nparts, uint32_t lcap, int need_count,
                                                      GTable out, uint32_t out_cap) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const uint32_t LT = lcap + 1;                 // slot lcap: the group whose packed key equals the empty mark
    uint64_t* lkey = reinterpret_cast<uint64_t*>(smem_raw);
    uint64_t* lacc = lkey + LT;
    uint32_t* lfirst = reinterpret_cast<uint32_t*>(lacc + (size_t)NACC * LT);
    uint32_t* lcount = lfirst + LT;
    __shared__ uint32_t lused;
    const uint32_t lmask = lcap - 1, llimit = lcap - (lcap >> 3);
    for (uint32_t part = blockIdx.x; part < nparts; part += gridDim.x) {
        const uint32_t b = pstart[part], e = pstart[part + 1];
        if (b == e) continue;
        for (uint32_t s = threadIdx.x; s < LT; s += blockDim.x) {
            lkey[s] = EMPTY64; lfirst[s] = NOROW; lcount[s] = 0;
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) lacc[(size_t)a * LT + s] = acc_init(as.kind[a]);
        }
        if (threadIdx.x == 0) lused = 0;
        __syncthreads();
        for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
            const uint64_t key = ksz == 4 ? (uint64_t) static_cast<const uint32_t*>(rkeys)[i] : static_cast<const uint64_t*>(rkeys)[i];
            uint32_t s = hash64(key) & lmask, found = FAIL;
            if (key == EMPTY64) found = lcap;
            else for (uint32_t p = 0; p <= lmask; ++p) {
                uint64_t cur = lkey[s];
                if (cur == key) { found = s; break; }
                if (cur == EMPTY64) {
                    if (lused >= llimit) break;
                    unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&lkey[s]), EMPTY64, key);
                    if (old == EMPTY64) { atomicAdd(&lused, 1u); found = s; break; }
                    if (old == key) { found = s; break; }
                }
                s = (s + 1) & lmask;
            }
            if (found == FAIL) { out.flags[0] = 1; continue; }      // partition holds more groups than the table: the host re-plans
            atomicMin(&lfirst[found], rrows[i]);
            if (need_count) atomicAdd(&lcount[found], 1u);
            _Pragma("unroll") for (int a = 0; a < NACC; ++a)
                acc_apply(&lacc[(size_t)a * LT + found], as.kind[a], val_operand(as.dt[a], as.col[a], i, as.kind[a], as.square[a], as.part[a]));
        }
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < LT; s += blockDim.x) {
            if (lfirst[s] == NOROW) continue;
            uint32_t g = atomicAdd(&out.flags[1], 1u);
            if (g >= out_cap) { out.flags[0] = 1; continue; }
            *out.key_p(g) = lkey[s];
            *out.first_p(g) = lfirst[s];
            *out.count_p(g) = lcount[s];
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) *out.acc_p(a, g) = lacc[(size_t)a * LT + s];
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) iota_kernel(uint32_t* __restrict__ p, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}

} // namespace

// Partitioned aggregation of (ks, as) over n rows into the compact record table `out` (AoS records, `out_cap` slots,
// flags[1] = number of groups written, flags[0] = overflow).  Needs packed (<= 8 byte) keys.
// Workspace is taken from the context arena (caller has reset it and reserved `aqg_partition_ws_bytes`).
size_t aqg_partition_ws_bytes(uint32_t n, int ksz, const AccSpec& as, uint32_t pbits) {
    size_t per_row = (size_t)ksz + 4;
    for (int a = 0; a < as.nacc; ++a) {
        bool dup = false;
        for (int b = 0; b < a; ++b) dup |= as.col[b] == as.col[a];
        if (!dup && as.dt[a] != AQG_NONE) per_row += aqg_dtype_size(as.dt[a]);
    }
    const size_t max_tiles = (size_t)n / PT + 258;
    return 2 * ((size_t)n + 64) * per_row + max_tiles * 256 * 4 + (max_tiles * 256 / 2048 + 64) * 4 + ((size_t)(1u << pbits) + 600) * 16 + 65536;
}

int aqg_partition_aggregate(aqg_ctx* ctx, const KeySpec& ks, const AccSpec& as_in, uint32_t n, uint32_t pbits, uint32_t lcap, int need_count,
                            GTable out, uint32_t out_cap) {
    if (pbits > 16) return aqg_fail(ctx, AQG_ERR_OVERFLOW, "partitioned group-by: more than 65536 partitions needed");
    const int ksz = ks.total_bytes <= 4 ? 4 : 8;
    const uint32_t nparts = 1u << pbits;
    const uint32_t bits1 = pbits > 8 ? pbits - 8 : pbits;     // level 1: the high bits of the partition id
    const uint32_t bits2 = pbits - bits1;                     // level 2: up to 8 more
    const size_t max_tiles = (size_t)n / PT + 258;

    AccSpec as = as_in;
    int ucols = 0;
    const void* ucol[MAXACC]; int udt[MAXACC]; int acc_ucol[MAXACC];
    for (int a = 0; a < as.nacc; ++a) {
        acc_ucol[a] = -1;
        if (as.dt[a] == AQG_NONE) continue;
        for (int u = 0; u < ucols; ++u) if (ucol[u] == as.col[a]) acc_ucol[a] = u;
        if (acc_ucol[a] < 0) { ucol[ucols] = as.col[a]; udt[ucols] = as.dt[a]; acc_ucol[a] = ucols++; }
    }
    void* bufs[2][MAXPAY];
    for (int set = 0; set < 2; ++set) {
        AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * ksz, &bufs[set][0]));
        AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * 4, &bufs[set][1]));
        for (int u = 0; u < ucols; ++u) AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * aqg_dtype_size(udt[u]), &bufs[set][2 + u]));
    }
    uint32_t *hist, *bsum, *seg0, *tp0, *seg1, *tp1, *pstart;
    AQG_TRY(aqg_ws_get(ctx, max_tiles * 256, &hist));
    AQG_TRY(aqg_ws_get(ctx, max_tiles * 256 / 2048 + 64, &bsum));
    AQG_TRY(aqg_ws_get(ctx, 4, &seg0));
    AQG_TRY(aqg_ws_get(ctx, 4, &tp0));
    AQG_TRY(aqg_ws_get(ctx, 260, &seg1));
    AQG_TRY(aqg_ws_get(ctx, 260, &tp1));
    AQG_TRY(aqg_ws_get(ctx, (size_t)nparts + 4, &pstart));

    // level-1 segment table: one segment [0, n)
    uint32_t h0[2] = {0, n};
    void* st = nullptr;
    AQG_TRY(aqg_host_stage(ctx, 16, &st));
    memcpy(st, h0, 8);
    AQG_HIP(ctx, hipMemcpyAsync(seg0, st, 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(tile_prefix_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)seg0, 1u, tp0);

    KeyIn kin;
    kin.from_cols = 1; kin.ks = ks; kin.rec = nullptr; kin.ksz = ksz;
    auto payload = [&](int src, int dst) {
        Payload pay;
        pay.ncols = 2 + ucols;
        pay.esz[0] = ksz; pay.esz[1] = 4;
        pay.in[0] = src < 0 ? nullptr : bufs[src][0]; pay.in[1] = src < 0 ? nullptr : bufs[src][1];
        pay.out[0] = bufs[dst][0]; pay.out[1] = bufs[dst][1];
        for (int u = 0; u < ucols; ++u) {
            pay.esz[2 + u] = (int)aqg_dtype_size(udt[u]);
            pay.in[2 + u] = src < 0 ? ucol[u] : bufs[src][2 + u];
            pay.out[2 + u] = bufs[dst][2 + u];
        }
        return pay;
    };
    const unsigned grid_tiles = (unsigned)max_tiles;
    const size_t scatter_lds = (size_t)PT * 9;   // 8-byte stage slot + 1-byte bin id per row of a tile
    // ---- level 1 -------------------------------------------------------------------------------------------------------
    Segs sg0{seg0, tp0, 1u};
    {
        const uint32_t shift = bits2;                        // high bits first
        const uint64_t hcount = ((uint64_t)n / PT + 2) * (1u << bits1);
        hipLaunchKernelGGL(part_hist_kernel, dim3(grid_tiles), dim3(PB), 0, ctx->stream, kin, sg0, shift, bits1, hist);
        AQG_TRY(aqg_exclusive_scan_u32(ctx, hist, hcount, bsum));
        AQG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&part_scatter_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds));
        hipLaunchKernelGGL((part_scatter_kernel<true>), dim3(grid_tiles), dim3(PB), scatter_lds, ctx->stream, kin, payload(-1, 0), sg0, shift, bits1, (const uint32_t*)hist);
        uint32_t* dst_start = bits2 ? seg1 : pstart;
        hipLaunchKernelGGL(bins_to_segments_kernel, dim3(1), dim3(256), 0, ctx->stream, sg0, bits1, (const uint32_t*)hist, n, dst_start);
        AQG_TRY(aqg_check_launch(ctx, "partition level 1"));
    }
    int cur = 0;
    // ---- level 2 -------------------------------------------------------------------------------------------------------
    if (bits2) {
        const uint32_t nseg = 1u << bits1;
        hipLaunchKernelGGL(tile_prefix_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)seg1, nseg, tp1);
        Segs sg1{seg1, tp1, nseg};
        kin.from_cols = 0; kin.rec = bufs[0][0];
        const uint64_t hcount = (uint64_t)max_tiles * (1u << bits2);
        AQG_HIP(ctx, hipMemsetAsync(hist, 0, hcount * 4, ctx->stream));   // unused tail tiles must read as zero in the scan
        hipLaunchKernelGGL(part_hist_kernel, dim3(grid_tiles), dim3(PB), 0, ctx->stream, kin, sg1, 0u, bits2, hist);
        AQG_TRY(aqg_exclusive_scan_u32(ctx, hist, hcount, bsum));
        AQG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&part_scatter_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds));
        hipLaunchKernelGGL((part_scatter_kernel<false>), dim3(grid_tiles), dim3(PB), scatter_lds, ctx->stream, kin, payload(0, 1), sg1, 0u, bits2, (const uint32_t*)hist);
        hipLaunchKernelGGL(bins_to_segments_kernel, dim3(aqg_grid(ctx, nparts, 256, 1, 4)), dim3(256), 0, ctx->stream, sg1, bits2, (const uint32_t*)hist, n, pstart);
        AQG_TRY(aqg_check_launch(ctx, "partition level 2"));
        cur = 1;
    }
    // ---- aggregate each partition in LDS -------------------------------------------------------------------------------
    for (int a = 0; a < as.nacc; ++a) if (acc_ucol[a] >= 0) as.col[a] = bufs[cur][2 + acc_ucol[a]];
    for (int a = 0; a < as.nacc; ++a) if (as.dt[a] == AQG_NONE) { as.dt[a] = AQG_UINT32; as.col[a] = bufs[cur][1]; }   // row-index operands
    const size_t lds = ((size_t)lcap + 1) * (8 + 8 * (size_t)as.nacc + 4 + 4);
    unsigned grid = nparts < (unsigned)ctx->num_cu * 4 ? nparts : (unsigned)ctx->num_cu * 4;
    auto launch = [&](auto kern) -> int {
        AQG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        aqg_kernel_timer_begin(ctx);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PB), lds, ctx->stream, (const void*)bufs[cur][0], ksz, (const uint32_t*)bufs[cur][1], as, (const uint32_t*)pstart, nparts,
                           lcap, need_count, out, out_cap);
        aqg_kernel_timer_end(ctx);
        return aqg_check_launch(ctx, "part_agg_kernel");
    };
    switch (as.nacc) {
    case 0: return launch(&part_agg_kernel<0>);
    case 1: return launch(&part_agg_kernel<1>);
    case 2: return launch(&part_agg_kernel<2>);
    case 3: return launch(&part_agg_kernel<3>);
    case 4: return launch(&part_agg_kernel<4>);
    case 5: return launch(&part_agg_kernel<5>);
    case 6: return launch(&part_agg_kernel<6>);
    case 7: return launch(&part_agg_kernel<7>);
    default: return launch(&part_agg_kernel<8>);
    }
}
