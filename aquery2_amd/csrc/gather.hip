// gather.hip -- index gather and boolean-mask stream compaction.
//   gather:  ColRef::operator[](vector_type<uint32_t>&)   reference server/table.h:184-189
//   compact: ColRef::operator[](const std::vector<bool>&)  reference server/table.h:190-198, as a true
//            compaction (the reference's result carries `size` junk slots in front: defect D11)
// Compaction = per-tile popcount, one-workgroup scan of the tile counts, then ballot / prefix-popcount
// ranks inside each wavefront -- output order is ascending row id.
#include "aqg_internal.hpp"
#include "dev_common.hpp"

namespace {

constexpr int CB = 256, CIT = 8, CTS = CB * CIT;

// four consecutive outputs per lane: one 16-byte index load, four gathers in flight, one vector store
template <class W> __global__ void __launch_bounds__(256) gather_kernel(const W* __restrict__ x, const uint32_t* __restrict__ idx, uint32_t m, W* __restrict__ out) {
    const uint32_t nchunk = m >> 2;
    const bool aligned = ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (aligned) {
        uint32_t c_lo, c_hi;
        wg_span(nchunk, c_lo, c_hi);
        for (uint32_t c = c_lo + threadIdx.x; c < c_hi; c += blockDim.x) {
            const pack<uint32_t, 4> i4 = *reinterpret_cast<const pack<uint32_t, 4>*>(idx + (size_t)c * 4);
            pack<W, 4> o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o.v[j] = x[i4.v[j]];
            *reinterpret_cast<pack<W, 4>*>(out + (size_t)c * 4) = o;
        }
        for (uint32_t i = (nchunk << 2) + blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) out[i] = x[idx[i]];
    } else {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) out[i] = x[idx[i]];
    }
}

__global__ void __launch_bounds__(CB) mask_count_kernel(const uint8_t* __restrict__ mask, uint32_t n, uint32_t* __restrict__ tile_cnt) {
    __shared__ uint32_t ws[4];
    uint32_t base = blockIdx.x * CTS + threadIdx.x * CIT, c = 0;
    if (base + CIT <= n && (((uintptr_t)(mask + base)) & 7) == 0) {
        uint64_t w = *reinterpret_cast<const uint64_t*>(mask + base);
#pragma unroll
        for (int j = 0; j < CIT; ++j) c += ((w >> (8 * j)) & 0xFF) != 0;
    } else {
        for (int j = 0; j < CIT; ++j) if (base + j < n) c += mask[base + j] != 0;
    }
    c = wave_reduce(c, OpAdd{});
    if (lane_id() == 0) ws[wave_id()] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// exclusive scan of tile counts in place; total -> tile_cnt[ntiles]
__global__ void __launch_bounds__(1024) count_scan_kernel(uint32_t* __restrict__ tile_cnt, uint32_t ntiles) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < ntiles ? tile_cnt[i] : 0;
        uint32_t incl = wave_scan_incl(v, OpAdd{}, lane_id());
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        uint32_t wbase = carry;
        for (int w = 0; w < wave_id(); ++w) wbase += wsum[w];
        if (i < ntiles) tile_cnt[i] = wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_cnt[ntiles] = carry;
}

struct alignas(16) w128 { uint64_t a, b; };

// W = element word (uint8/16/32/64 or 16-byte struct); INDEX: write the row id instead of a value
template <class W, bool INDEX>
__global__ void __launch_bounds__(CB) compact_kernel(const W* __restrict__ x, const uint8_t* __restrict__ mask, uint32_t n,
                                                     const uint32_t* __restrict__ tile_off, W* __restrict__ out, uint32_t* __restrict__ idx_out) {
    __shared__ uint32_t ws[4];
    const uint32_t base = blockIdx.x * CTS + threadIdx.x * CIT;
    bool keep[CIT];
    W v[CIT];
    uint32_t c = 0;
    const bool full = base + CIT <= n;
    if (full && ((reinterpret_cast<uintptr_t>(mask + base)) & 7) == 0) {      // eight mask bytes with one load
        const uint64_t w = *reinterpret_cast<const uint64_t*>(mask + base);
#pragma unroll
        for (int j = 0; j < CIT; ++j) keep[j] = ((w >> (8 * j)) & 0xFF) != 0;
    } else {
#pragma unroll
        for (int j = 0; j < CIT; ++j) keep[j] = base + j < n && mask[base + j] != 0;
    }
    if constexpr (!INDEX) {                                                    // every value is loaded (no load behind a branch)
        if (full && (reinterpret_cast<uintptr_t>(x + base) & (sizeof(W) * 4 > 16 ? 15 : sizeof(W) * 4 - 1)) == 0) {
            const pack<W, 4> a = *reinterpret_cast<const pack<W, 4>*>(x + base), b = *reinterpret_cast<const pack<W, 4>*>(x + base + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a.v[j]; v[4 + j] = b.v[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < CIT; ++j) v[j] = x[base + j < n ? base + j : n - 1];
        }
    }
#pragma unroll
    for (int j = 0; j < CIT; ++j) c += keep[j];
    uint32_t incl = wave_scan_incl(c, OpAdd{}, lane_id());
    if (lane_id() == 63) ws[wave_id()] = incl;
    __syncthreads();
    // kept elements are packed in LDS first and leave as whole-wavefront runs of consecutive addresses (a lane storing its own
    // survivors straight to memory touched ~5 elements of stride per lane: 1.85 ms per 1e9 rows at 60 % kept)
    using S = std::conditional_t<INDEX, uint32_t, W>;
    __shared__ S stage[CTS];
    uint32_t pos = incl - c;
    for (int w = 0; w < wave_id(); ++w) pos += ws[w];
#pragma unroll
    for (int j = 0; j < CIT; ++j) {
        if (keep[j]) {
            if constexpr (INDEX) stage[pos] = base + j; else stage[pos] = v[j];
            ++pos;
        }
    }
    __syncthreads();
    const uint32_t total = ws[0] + ws[1] + ws[2] + ws[3];
    const uint32_t off = tile_off[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < total; i += CB) {
        if constexpr (INDEX) idx_out[off + i] = stage[i]; else out[off + i] = stage[i];
    }
}

// (A single-pass form -- links of eight tiles chained with the look-back of chain_dev.hpp, mask and values read once -- measured
// 2.49 ms per 1e9 rows against 1.66 ms for the two passes below: holding eight tiles of values costs the registers that let
// eight independent small workgroups per CU hide each other's latency here.)
template <bool INDEX>
int run_compact(aqg_ctx* ctx, int t, const void* x, const uint8_t* mask, uint32_t n, void* out, uint32_t* idx_out, uint32_t* m_host) {
    *m_host = 0;
    if (n == 0) return AQG_OK;
    uint32_t ntiles = aqg_ceil_div(n, CTS);
    AQG_TRY(aqg_ws_reset(ctx));
    AQG_TRY(aqg_ws_ensure(ctx, (size_t)(ntiles + 1) * 4 + ((size_t)ntiles / 2048 + 4) * 4 + 8192));
    uint32_t* tcnt;
    AQG_TRY(aqg_ws_get(ctx, ntiles + 1, &tcnt));
    hipLaunchKernelGGL(mask_count_kernel, dim3(ntiles), dim3(CB), 0, ctx->stream, mask, n, tcnt);
    if (ntiles > 8192) {     // large inputs: the three-kernel scan (the single-workgroup one took 0.56 ms at 1e9 rows)
        uint32_t* bsum;
        AQG_TRY(aqg_ws_get(ctx, (size_t)ntiles / 2048 + 4, &bsum));
        AQG_HIP(ctx, hipMemsetAsync(tcnt + ntiles, 0, 4, ctx->stream));
        AQG_TRY(aqg_exclusive_scan_u32(ctx, tcnt, (uint64_t)ntiles + 1, bsum));
    } else {
        hipLaunchKernelGGL(count_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, tcnt, ntiles);
    }
    if constexpr (INDEX) {
        hipLaunchKernelGGL((compact_kernel<uint32_t, true>), dim3(ntiles), dim3(CB), 0, ctx->stream, (const uint32_t*)nullptr, mask, n, tcnt, (uint32_t*)nullptr, idx_out);
    } else {
        switch (aqg_dtype_size(t)) {
        case 1: hipLaunchKernelGGL((compact_kernel<uint8_t, false>), dim3(ntiles), dim3(CB), 0, ctx->stream, (const uint8_t*)x, mask, n, tcnt, (uint8_t*)out, (uint32_t*)nullptr); break;
        case 2: hipLaunchKernelGGL((compact_kernel<uint16_t, false>), dim3(ntiles), dim3(CB), 0, ctx->stream, (const uint16_t*)x, mask, n, tcnt, (uint16_t*)out, (uint32_t*)nullptr); break;
        case 4: hipLaunchKernelGGL((compact_kernel<uint32_t, false>), dim3(ntiles), dim3(CB), 0, ctx->stream, (const uint32_t*)x, mask, n, tcnt, (uint32_t*)out, (uint32_t*)nullptr); break;
        case 8: hipLaunchKernelGGL((compact_kernel<uint64_t, false>), dim3(ntiles), dim3(CB), 0, ctx->stream, (const uint64_t*)x, mask, n, tcnt, (uint64_t*)out, (uint32_t*)nullptr); break;
        case 16: hipLaunchKernelGGL((compact_kernel<w128, false>), dim3(ntiles), dim3(CB), 0, ctx->stream, (const w128*)x, mask, n, tcnt, (w128*)out, (uint32_t*)nullptr); break;
        default: return aqg_fail(ctx, AQG_ERR_DTYPE, "compact: dtype");
        }
    }
    AQG_TRY(aqg_check_launch(ctx, "compact"));
    return aqg_d2h(ctx, m_host, tcnt + ntiles, 4);
}

} // namespace

extern "C" {

int aqg_gather(aqg_ctx* ctx, int t, const void* x, const uint32_t* idx, uint32_t m, void* out) {
    if (!ctx || ((!x || !idx || !out) && m)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_gather: bad argument");
    if (m == 0) return AQG_OK;
    AQG_CHECK_ROWS(ctx, m, "aqg_gather");
    unsigned grid = aqg_grid(ctx, m, 256, 4, 16);
    switch (aqg_dtype_size(t)) {
    case 1: hipLaunchKernelGGL((gather_kernel<uint8_t>), dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t*)x, idx, m, (uint8_t*)out); break;
    case 2: hipLaunchKernelGGL((gather_kernel<uint16_t>), dim3(grid), dim3(256), 0, ctx->stream, (const uint16_t*)x, idx, m, (uint16_t*)out); break;
    case 4: hipLaunchKernelGGL((gather_kernel<uint32_t>), dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*)x, idx, m, (uint32_t*)out); break;
    case 8: hipLaunchKernelGGL((gather_kernel<uint64_t>), dim3(grid), dim3(256), 0, ctx->stream, (const uint64_t*)x, idx, m, (uint64_t*)out); break;
    case 16: hipLaunchKernelGGL((gather_kernel<w128>), dim3(grid), dim3(256), 0, ctx->stream, (const w128*)x, idx, m, (w128*)out); break;
    default: return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_gather: dtype");
    }
    return aqg_check_launch(ctx, "gather_kernel");
}

int aqg_compact(aqg_ctx* ctx, int t, const void* x, const uint8_t* mask, uint32_t n, void* out, uint32_t* m_host) {
    if (!ctx || !m_host || ((!x || !mask || !out) && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_compact: bad argument");
    AQG_CHECK_ROWS(ctx, n, "aqg_compact");
    return run_compact<false>(ctx, t, x, mask, n, out, nullptr, m_host);
}

int aqg_mask_to_index(aqg_ctx* ctx, const uint8_t* mask, uint32_t n, uint32_t* idx_out, uint32_t* m_host) {
    if (!ctx || !m_host || ((!mask || !idx_out) && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_mask_to_index: bad argument");
    AQG_CHECK_ROWS(ctx, n, "aqg_mask_to_index");
    return run_compact<true>(ctx, AQG_UINT32, nullptr, mask, n, nullptr, idx_out, m_host);
}

} // extern "C"
