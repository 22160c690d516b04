// reduced from unpack_kernel as it was before commit 13f94ce (aquery2_amd/csrc/groupby.hip): the sized key store inlined in the copy loop
#include <hip/hip_runtime.h>
#include <cstdint>
__host__ __device__ inline int dtype_size(int dt) {
    switch (dt) { case 10: case 15: case 16: return 1; case 7: case 14: return 2; case 0: case 11: case 1: return 4; default: return 8; }
}
extern "C" __global__ void __launch_bounds__(256) unpack_repro(const long long* __restrict__ gathered, uint32_t world, uint32_t gmax, int key_dt,
                                                                void* __restrict__ keys, long long* __restrict__ vals) {
    __shared__ uint32_t off[65];
    if (threadIdx.x == 0) { uint32_t s = 0; for (uint32_t r = 0; r < world; ++r) { off[r] = s; s += (uint32_t)gathered[(size_t)r * (gmax + 1) * 2]; } off[world] = s; }
    __syncthreads();
    if (off[world] > world * gmax) return;
    for (uint32_t r = blockIdx.x; r < world; r += gridDim.x) {
        const long long* src = gathered + (size_t)r * (gmax + 1) * 2;
        const uint32_t cnt = off[r + 1] - off[r];
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
            const long long k = src[2 + 2 * i];
            const uint32_t d = off[r] + i;
            switch (dtype_size(key_dt)) {
            case 1: static_cast<uint8_t*>(keys)[d] = (uint8_t)k; break;
            case 2: static_cast<uint16_t*>(keys)[d] = (uint16_t)k; break;
            case 4: static_cast<uint32_t*>(keys)[d] = (uint32_t)k; break;
            default: static_cast<long long*>(keys)[d] = k; break;
            }
            vals[d] = src[3 + 2 * i];
        }
    }
}
