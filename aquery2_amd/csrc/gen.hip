// gen.hip -- counter-based synthetic h2o / time-series columns, generated in HBM.
// Row i of a column depends only on (seed, column, row_base + i): shards reproduce the same
// table whatever the GPU count.  MUST stay identical to oracle/aq_oracle.c: aqo_gen_column.
#include "aqg_internal.hpp"
#include "dev_common.hpp"

__device__ static inline uint64_t gen_rnd(uint64_t seed, int col, uint64_t row) {
    return splitmix64(splitmix64(seed * 256 + (uint64_t)col) ^ row);
}
__device__ static inline uint32_t gen_uniform(uint64_t r, uint32_t range) {
    return (uint32_t)(((r >> 32) * (uint64_t)range) >> 32);
}
__device__ static inline int32_t tri_wave(uint64_t i, uint32_t period, int32_t amp) {
    uint32_t ph = (uint32_t)(i % period);
    uint32_t half = period / 2;
    int64_t up = ph < half ? (int64_t)ph : (int64_t)(period - ph);
    return (int32_t)(2 * (int64_t)amp * up / (int64_t)half) - amp;
}

__device__ static inline uint32_t gen_value(int col, uint64_t seed, uint64_t row, uint32_t K, uint32_t big) {
    uint64_t r = gen_rnd(seed, col, row);
    switch (col) {
    case AQG_GEN_ID1: case AQG_GEN_ID2: case AQG_GEN_ID4: case AQG_GEN_ID5: return 1u + gen_uniform(r, K);
    case AQG_GEN_ID3: case AQG_GEN_ID6: return 1u + gen_uniform(r, big);
    case AQG_GEN_V1: return 1u + gen_uniform(r, 5);
    case AQG_GEN_V2: return 1u + gen_uniform(r, 15);
    case AQG_GEN_V3: {
        uint64_t micro = ((r >> 32) * 100000000ull) >> 32;
        return __float_as_uint((float)((double)micro / 1e6));
    }
    case AQG_GEN_TIMESTAMP: return (uint32_t)(int32_t)(row + 1);
    default: { // AQG_GEN_PRICE
        int32_t p = 275 + tri_wave(row, 1009, 100) + tri_wave(row, 104729, 100) + (int32_t)gen_uniform(r, 51) - 25;
        p = p < 50 ? 50 : (p > 500 ? 500 : p);
        return (uint32_t)p;
    }
    }
}

// every generated column is 4 bytes wide: each lane writes one dwordx4 per step
__global__ void __launch_bounds__(256) gen_kernel(int col, uint64_t seed, uint64_t row_base, uint32_t n, uint32_t K,
                                                  uint32_t big, uint32_t* __restrict__ out) {
    uint32_t nvec = n >> 2;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
        uint4 o;
        uint64_t row = row_base + (uint64_t)v * 4;
        o.x = gen_value(col, seed, row, K, big);
        o.y = gen_value(col, seed, row + 1, K, big);
        o.z = gen_value(col, seed, row + 2, K, big);
        o.w = gen_value(col, seed, row + 3, K, big);
        reinterpret_cast<uint4*>(out)[v] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
        uint32_t i = (nvec << 2) + threadIdx.x;
        out[i] = gen_value(col, seed, row_base + i, K, big);
    }
}

extern "C" int aqg_gen_column(aqg_ctx* ctx, int col, uint64_t seed, uint64_t row_base, uint32_t n, uint64_t n_total,
                              uint32_t K, void* out_dev) {
    if (!ctx || (!out_dev && n) || K == 0 || col < 0 || col > AQG_GEN_PRICE) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_gen_column: bad argument");
    if (n == 0) return AQG_OK;
    AQG_CHECK_ROWS(ctx, n, "aqg_gen_column");
    if ((uintptr_t)out_dev & 15) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_gen_column: output must be 16-byte aligned");
    uint64_t big = n_total / K;
    if (big < 1) big = 1;
    if (big > 0x7FFFFFFFull) big = 0x7FFFFFFFull;
    unsigned grid = aqg_grid(ctx, n, 256, 4, 16);
    hipLaunchKernelGGL(gen_kernel, dim3(grid), dim3(256), 0, ctx->stream, col, seed, row_base, n, K, (uint32_t)big,
                       static_cast<uint32_t*>(out_dev));
    return aqg_check_launch(ctx, "gen_kernel");
}
