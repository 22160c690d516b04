// postproc.hip -- aqg_groupby_postproc: the device form of AQHashTable::ht_postproc
// (reference server/hasher.h:181-198): a counting sort of row ids by group id that leaves the rows of
// every group in DESCENDING order, plus the group start offsets (ht_base after postproc).
//
// Stable LSD radix sort, 8 bits per pass, over the sequence p -> (key = reversemap[n-1-p], value = n-1-p):
// a stable sort of that reversed sequence yields descending row ids inside each group.  Per pass:
//   digit histogram per 4096-row tile -> exclusive scan of the (digit-major, tile-minor) counts ->
//   stable scatter: 1024 lanes x 4 rows, all loaded up front; wavefront match-any by 8 ballots gives the rank among equal
//   digits inside a wave, an LDS (round, wavefront, digit) count matrix orders the cells; the tile is staged digit-major
//   in LDS and streamed out in runs.
// One pass when G <= 256 (h2o Q1): 12 B/row (4 B histogram read + 4 B read + 4 B write).
#include "aqg_internal.hpp"
#include "dev_common.hpp"
#include "groupby_handle.hpp"

namespace {

constexpr int RB = 1024;           // lanes per workgroup
constexpr int ROUNDS = 4;          // rows per lane and tile, all loaded before first use
constexpr int RT = RB * ROUNDS;    // rows per tile
constexpr int NW = RB / 64;        // wavefronts per workgroup

// ---- exclusive scan of a uint32 array (in place), three kernels ----------------------------------
__global__ void __launch_bounds__(256) u32_block_sum_kernel(const uint32_t* __restrict__ d, uint64_t count, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t ws[4];
    uint64_t base = (uint64_t)blockIdx.x * 2048 + threadIdx.x * 8;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (base + j < count) c += d[base + j];
    c = wave_reduce(c, OpAdd{});
    if (lane_id() == 0) ws[wave_id()] = c;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ void __launch_bounds__(1024) u32_scan_small_kernel(uint32_t* __restrict__ d, uint32_t count) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < count; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < count ? d[i] : 0;
        uint32_t incl = wave_scan_incl(v, OpAdd{}, lane_id());
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        uint32_t wbase = carry;
        for (int w = 0; w < wave_id(); ++w) wbase += wsum[w];
        if (i < count) d[i] = wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = wbase + incl;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) u32_block_scan_kernel(uint32_t* __restrict__ d, uint64_t count, const uint32_t* __restrict__ bsum_excl) {
    __shared__ uint32_t ws[4];
    uint64_t base = (uint64_t)blockIdx.x * 2048 + threadIdx.x * 8;
    uint32_t v[8], c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = base + j < count ? d[base + j] : 0; c += v[j]; }
    uint32_t incl = wave_scan_incl(c, OpAdd{}, lane_id());
    if (lane_id() == 63) ws[wave_id()] = incl;
    __syncthreads();
    uint32_t run = bsum_excl[blockIdx.x] + incl - c;
    for (int w = 0; w < wave_id(); ++w) run += ws[w];
#pragma unroll
    for (int j = 0; j < 8; ++j) { if (base + j < count) d[base + j] = run; run += v[j]; }
}
} // namespace
// bsum must hold ceil(count/2048) words (declared in aqg_internal.hpp; also used by join.hip)
int aqg_exclusive_scan_u32(aqg_ctx* ctx, uint32_t* d, uint64_t count, uint32_t* bsum) {
    if (count == 0) return AQG_OK;
    uint32_t nb = (uint32_t)((count + 2047) / 2048);
    hipLaunchKernelGGL(u32_block_sum_kernel, dim3(nb), dim3(256), 0, ctx->stream, d, count, bsum);
    hipLaunchKernelGGL(u32_scan_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, bsum, nb);
    hipLaunchKernelGGL(u32_block_scan_kernel, dim3(nb), dim3(256), 0, ctx->stream, d, count, bsum);
    return aqg_check_launch(ctx, "exclusive_scan_u32");
}
namespace {

// element p of the pass input: FIRST pass reads the group-id column backwards and synthesises the row id.
// Rows beyond n read a clamped index (every load is issued, none sits behind a branch) and are masked by the caller.
template <bool FIRST> __device__ inline void load_pair(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n, uint32_t p,
                                                       uint32_t& k, uint32_t& v) {
    const uint32_t q = p < n ? p : n - 1;
    if constexpr (FIRST) { v = n - 1 - q; k = keys[v]; } else { k = keys[q]; v = vals[q]; }
}

// Tile of a workgroup.  Workgroups go to the eight XCDs round-robin; with tile = blockIdx consecutive tiles sit on different XCDs,
// and the pieces they write next to each other -- the end of one tile's run of a digit and the start of the next tile's, the
// [digit][tile] histogram words -- become partial-line writes out of eight different L2s.  Here XCD x walks the tiles
// [x * per, (x + 1) * per) in order, so neighbouring pieces meet in one L2 before they are written back.  (grid = 8 * per)
__device__ inline uint32_t xcd_tile(uint32_t ntiles) {
    const uint32_t per = (ntiles + 7) / 8;
    return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

template <bool FIRST>
__global__ void __launch_bounds__(RB) radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t dmask, uint32_t ntiles,
                                                        uint32_t* __restrict__ hist /* [digits][ntiles] */) {
    __shared__ uint32_t h[256];
    const uint32_t tile = xcd_tile(ntiles);
    if (tile >= ntiles) return;
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tbase = tile * RT;
    uint32_t k[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t p = tbase + r * RB + threadIdx.x, q = p < n ? p : n - 1;
        k[r] = FIRST ? keys[n - 1 - q] : keys[q];
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) if (tbase + r * RB + threadIdx.x < n) atomicAdd(&h[(k[r] >> shift) & dmask], 1u);
    __syncthreads();
    if (threadIdx.x <= dmask) hist[(size_t)threadIdx.x * ntiles + tile] = h[threadIdx.x];
}

// Stable scatter of one tile.  Rank of a row among the tile's rows of its digit = rows of that digit in earlier
// (round, wavefront) cells + its rank inside its own wavefront (match-any by 8 ballots).  The tile is then laid out digit-major
// in LDS and streamed out, so consecutive lanes write consecutive addresses of a digit's run (direct 4-byte scatters from
// registers ran at 15 % of the HBM roofline: h2o Q1 groups, 1e9 rows, 10.2 ms).
// PAY: the value that travels with a row.  0: its row id (ht_postproc).  1: the row's element of a value column `x` (aqg_grouped_flatten:
// the column in the flat row-list layout, x[vecs[g][i]] for every group) -- read coalesced in the FIRST pass (the rows are walked
// backwards, like the group ids), carried as one dword plane (two for 8-byte elements: `vals2`), written with its own element
// size in the LAST pass.  Carrying the value costs what carrying the row id costs, and saves the gather through the row ids
// afterwards: 64 B of line per row for a few hundred groups (a group's rows lie ~G rows apart).
struct PayIO { const void* x; void* xout; int esz; const uint32_t* vals2; uint32_t* vals2_out; };
template <bool FIRST, bool LAST, int PAY>
__global__ void __launch_bounds__(RB) radix_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n, uint32_t shift,
                                                           uint32_t ntiles, const uint32_t* __restrict__ hist_scanned,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int nbits /* significant bits of this pass's digit */,
                                                           uint32_t dmask /* 2^(digit width) - 1 */, PayIO io) {
    constexpr int CELLS = ROUNDS * NW;                 // (round, wavefront) cells in rank order
    __shared__ uint16_t cell[CELLS][256];              // rows of each digit per cell, then their exclusive prefix over the cells
    __shared__ uint32_t gbase[256], lbase[256], wsum[4];
    __shared__ uint32_t stage[RT], delta[RT];
    const uint32_t tile = xcd_tile(ntiles);
    if (tile >= ntiles) return;
    const uint32_t tbase = tile * RT;
    const uint32_t nrows = n - tbase < (uint32_t)RT ? n - tbase : (uint32_t)RT;
    const int lane = lane_id(), wid = wave_id();
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (uint32_t i = threadIdx.x; i < CELLS * 256 / 2; i += RB) reinterpret_cast<uint32_t*>(&cell[0][0])[i] = 0;
    if (threadIdx.x < 256) gbase[threadIdx.x] = threadIdx.x <= dmask ? hist_scanned[(size_t)threadIdx.x * ntiles + tile] : 0u;
    uint32_t k[ROUNDS], v[ROUNDS], v2[ROUNDS], d[ROUNDS], rank[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        load_pair<FIRST>(keys, vals, n, tbase + r * RB + threadIdx.x, k[r], v[r]);
        v2[r] = 0;
        if constexpr (PAY != 0) {
            const uint32_t p = tbase + r * RB + threadIdx.x, q = p < n ? p : n - 1;
            if constexpr (FIRST) {                      // v[r] is the row id here: fetch the row's element (wave-uniform size switch)
                const uint32_t row = v[r];
                if (io.esz == 4) v[r] = static_cast<const uint32_t*>(io.x)[row];
                else if (io.esz == 8) { const uint2 t = static_cast<const uint2*>(io.x)[row]; v[r] = t.x; v2[r] = t.y; }
                else if (io.esz == 2) v[r] = static_cast<const uint16_t*>(io.x)[row];
                else v[r] = static_cast<const uint8_t*>(io.x)[row];
            } else if (io.esz == 8) v2[r] = io.vals2[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const bool live = r * RB + threadIdx.x < nrows;
        d[r] = (k[r] >> shift) & dmask;
        uint64_t peers = __ballot(live);                // lanes of this wave holding the same digit in this round
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            if (bit >= nbits) break;                    // (uniform: the digit's higher bits are zero in every row -- 100 groups: seven ballots)
            uint64_t bal = __ballot((d[r] >> bit) & 1);
            peers &= ((d[r] >> bit) & 1) ? bal : ~bal;
        }
        rank[r] = __popcll(peers & lt_mask);
        if (live && rank[r] == 0) cell[r * NW + wid][d[r]] = (uint16_t)__popcll(peers);
    }
    __syncthreads();
    uint32_t total = 0, incl = 0;
    if (threadIdx.x < 256) {                            // one lane per digit: exclusive prefix over the cells, 16 cells at a time
        for (int c0 = 0; c0 < CELLS; c0 += 16) {
            uint32_t t[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) t[c] = cell[c0 + c][threadIdx.x];
#pragma unroll
            for (int c = 0; c < 16; ++c) { cell[c0 + c][threadIdx.x] = (uint16_t)total; total += t[c]; }
        }
        incl = wave_scan_incl(total, OpAdd{}, lane);
        if (lane == 63) wsum[wid] = incl;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t base = 0;
        for (int w = 0; w < wid; ++w) base += wsum[w];
        lbase[threadIdx.x] = base + incl - total;
    }
    __syncthreads();
    uint32_t pos[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r * RB + threadIdx.x < nrows) {
            const uint32_t lb = lbase[d[r]];
            pos[r] = lb + cell[r * NW + wid][d[r]] + rank[r];
            delta[pos[r]] = gbase[d[r]] - lb;
            stage[pos[r]] = v[r];
        }
    }
    __syncthreads();
    if constexpr (PAY != 0 && LAST) {                   // the flat column itself, in its own element size
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t j = r * RB + threadIdx.x;
            if (j < nrows) {
                const size_t o = (size_t)j + delta[j];
                if (io.esz == 4) static_cast<uint32_t*>(io.xout)[o] = stage[j];
                else if (io.esz == 8) static_cast<uint32_t*>(io.xout)[2 * o] = stage[j];
                else if (io.esz == 2) static_cast<uint16_t*>(io.xout)[o] = (uint16_t)stage[j];
                else static_cast<uint8_t*>(io.xout)[o] = (uint8_t)stage[j];
            }
        }
        if (io.esz == 8) {                              // (uniform) the high halves take the staging buffer next
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) if (r * RB + threadIdx.x < nrows) stage[pos[r]] = v2[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const uint32_t j = r * RB + threadIdx.x;
                if (j < nrows) static_cast<uint32_t*>(io.xout)[2 * ((size_t)j + delta[j]) + 1] = stage[j];
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t j = r * RB + threadIdx.x;
        if (j < nrows) vals_out[j + delta[j]] = stage[j];
    }
    if constexpr (PAY != 0) {
        if (io.esz == 8) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) if (r * RB + threadIdx.x < nrows) stage[pos[r]] = v2[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const uint32_t j = r * RB + threadIdx.x;
                if (j < nrows) io.vals2_out[j + delta[j]] = stage[j];
            }
        }
    }
    if constexpr (!LAST) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) if (r * RB + threadIdx.x < nrows) stage[pos[r]] = k[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t j = r * RB + threadIdx.x;
            if (j < nrows) keys_out[j + delta[j]] = stage[j];
        }
    }
}

__global__ void __launch_bounds__(256) copy_counts_kernel(const uint32_t* __restrict__ counts, uint32_t G, uint32_t* __restrict__ offsets) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g <= G; g += gridDim.x * blockDim.x) offsets[g] = g < G ? counts[g] : 0;
}

} // namespace

// The radix passes over the group-id column of a build.  `x == nullptr`: row ids -> row_ids_dev (ht_postproc); else the value column
// `x` of `esz`-byte elements -> xout in the flat layout.  `ws_managed`: the caller has reset and sized the workspace (aqg_postproc_ws_bytes).
size_t aqg_postproc_ws_bytes(uint32_t n, uint32_t G, int esz) {
    const uint32_t ntiles = aqg_ceil_div(n, RT);
    uint32_t bits = 0;
    while (bits < 32 && (1ull << bits) < G) ++bits;
    const uint32_t passes = bits == 0 ? 1 : (bits + 7) / 8;
    // digits of EQUAL width: 17 bits (1e5 groups) are three digits of 6 / 6 / 5 bits, not 8 / 8 / 1 -- the run a tile writes per digit is
    // four times as long (64 rows: two whole lines) and a rank takes six ballots instead of eight
    const uint32_t width = bits == 0 ? 1 : (bits + passes - 1) / passes, dmask = (1u << width) - 1;
    const uint64_t hcount = (uint64_t)(dmask + 1) * ntiles;
    size_t need = hcount * 4 + ((hcount + 2047) / 2048 + (G + 2048) / 2048 + 16) * 4 + 8192;
    if (passes > 1) need += (size_t)n * (esz == 8 ? 24 : 16) + 8192;
    return need;
}
int aqg_radix_by_group(aqg_ctx* ctx, aqg_groupby* g, uint32_t* row_ids_dev, const void* x, int esz, void* xout, bool ws_managed) {
    const uint32_t n = g->n, G = g->ngroups;
    if (n == 0) return AQG_OK;
    const uint32_t ntiles = aqg_ceil_div(n, RT);
    uint32_t bits = 0;
    while (bits < 32 && (1ull << bits) < G) ++bits;
    const uint32_t passes = bits == 0 ? 1 : (bits + 7) / 8;
    // digits of EQUAL width: 17 bits (1e5 groups) are three digits of 6 / 6 / 5 bits, not 8 / 8 / 1 -- the run a tile writes per digit is
    // four times as long (64 rows: two whole lines) and a rank takes six ballots instead of eight
    const uint32_t width = bits == 0 ? 1 : (bits + passes - 1) / passes, dmask = (1u << width) - 1;
    const uint64_t hcount = (uint64_t)(dmask + 1) * ntiles;
    const uint32_t grid8 = (ntiles + 7) / 8 * 8;                 // xcd_tile: eight interleaved walks over the tiles
    if (!ws_managed) { AQG_TRY(aqg_ws_reset(ctx)); AQG_TRY(aqg_ws_ensure(ctx, aqg_postproc_ws_bytes(n, G, esz))); }
    uint32_t *hist, *bsum, *k0 = nullptr, *v0 = nullptr, *k1 = nullptr, *v1 = nullptr, *w0 = nullptr, *w1 = nullptr;
    AQG_TRY(aqg_ws_get(ctx, hcount ? hcount : 1, &hist));
    AQG_TRY(aqg_ws_get(ctx, (hcount + 2047) / 2048 + (G + 2048) / 2048 + 16, &bsum));
    if (passes > 1) {
        AQG_TRY(aqg_ws_get(ctx, n, &k0)); AQG_TRY(aqg_ws_get(ctx, n, &v0));
        AQG_TRY(aqg_ws_get(ctx, n, &k1)); AQG_TRY(aqg_ws_get(ctx, n, &v1));
        if (x && esz == 8) { AQG_TRY(aqg_ws_get(ctx, n, &w0)); AQG_TRY(aqg_ws_get(ctx, n, &w1)); }
    }
    const uint32_t* kin = g->reversemap;
    const uint32_t* vin = nullptr;
    const uint32_t* win = nullptr;
    for (uint32_t pass = 0; pass < passes; ++pass) {
        const bool first = pass == 0, last = pass + 1 == passes;
        const uint32_t shift = pass * width;
        const int nbits = bits <= shift ? 1 : (int)(bits - shift < width ? bits - shift : width);
        uint32_t* kout = last ? nullptr : ((pass & 1) ? k1 : k0);
        uint32_t* vout = last ? row_ids_dev : ((pass & 1) ? v1 : v0);
        uint32_t* wout = last ? nullptr : ((pass & 1) ? w1 : w0);
        PayIO io{x, xout, esz, win, wout};
        if (first) hipLaunchKernelGGL((radix_hist_kernel<true>), dim3(grid8), dim3(RB), 0, ctx->stream, kin, n, shift, dmask, ntiles, hist);
        else hipLaunchKernelGGL((radix_hist_kernel<false>), dim3(grid8), dim3(RB), 0, ctx->stream, kin, n, shift, dmask, ntiles, hist);
        AQG_TRY(aqg_exclusive_scan_u32(ctx, hist, hcount, bsum));
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid8), dim3(RB), 0, ctx->stream, kin, vin, n, shift, ntiles, hist, kout, vout, nbits, dmask, io); };
        if (first && last) aqg_kernel_timer_begin(ctx);
        if (x) {
            if (first && last) go(&radix_scatter_kernel<true, true, 1>);
            else if (first) go(&radix_scatter_kernel<true, false, 1>);
            else if (last) go(&radix_scatter_kernel<false, true, 1>);
            else go(&radix_scatter_kernel<false, false, 1>);
        } else {
            if (first && last) go(&radix_scatter_kernel<true, true, 0>);
            else if (first) go(&radix_scatter_kernel<true, false, 0>);
            else if (last) go(&radix_scatter_kernel<false, true, 0>);
            else go(&radix_scatter_kernel<false, false, 0>);
        }
        if (first && last) aqg_kernel_timer_end(ctx);
        AQG_TRY(aqg_check_launch(ctx, "radix pass"));
        kin = kout; vin = vout; win = wout;
    }
    return AQG_OK;
}
// offsets[G+1] = exclusive scan of the group sizes (ht_base after ht_postproc, hasher.h:186-190); bsum: (G + 2048) / 2048 + 16 words
int aqg_group_offsets(aqg_ctx* ctx, const aqg_groupby* g, uint32_t* offsets_dev, uint32_t* bsum) {
    const uint32_t G = g->ngroups;
    hipLaunchKernelGGL(copy_counts_kernel, dim3(aqg_grid(ctx, G + 1, 256, 1, 8)), dim3(256), 0, ctx->stream, g->counts, G, offsets_dev);
    return aqg_exclusive_scan_u32(ctx, offsets_dev, (uint64_t)G + 1, bsum);
}

extern "C" int aqg_groupby_postproc(aqg_groupby* g, uint32_t* offsets_dev, uint32_t* row_ids_dev) {
    if (!g || !offsets_dev || (!row_ids_dev && g->n)) return AQG_ERR_ARG;
    aqg_ctx* ctx = g->ctx;
    if (!g->has_reversemap || !g->has_counts) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_postproc: handle was not made by aqg_groupby_build");
    const uint32_t n = g->n, G = g->ngroups;
    AQG_TRY(aqg_ws_reset(ctx));
    AQG_TRY(aqg_ws_ensure(ctx, aqg_postproc_ws_bytes(n, G, 4) + ((size_t)(G + 2048) / 2048 + 16) * 4 + 256));
    uint32_t* bsum0;
    AQG_TRY(aqg_ws_get(ctx, (G + 2048) / 2048 + 16, &bsum0));
    AQG_TRY(aqg_group_offsets(ctx, g, offsets_dev, bsum0));      // offsets = exclusive scan of counts, offsets[G] = n
    return aqg_radix_by_group(ctx, g, row_ids_dev, nullptr, 4, nullptr, /*ws_managed=*/true);
}
