#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned char sm[];
__global__ void k(unsigned* out, unsigned n) { for (unsigned i = threadIdx.x; i < n / 4; i += blockDim.x) ((unsigned*)sm)[i] = i; __syncthreads(); if (threadIdx.x == 0) out[blockIdx.x] = ((unsigned*)sm)[n / 4 - 1]; }
int main() {
    int v; hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0); printf("MaxSharedMemoryPerBlock %d\n", v);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("sharedMemPerBlock %zu optin %zu perMP %zu\n", p.sharedMemPerBlock, p.sharedMemPerBlockOptin, p.sharedMemPerMultiprocessor);
    unsigned* d; hipMalloc(&d, 4096);
    for (int kb : {48, 64, 72, 80, 96, 128, 160}) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        hipLaunchKernelGGL(k, dim3(4), dim3(256), kb * 1024, 0, d, (unsigned)kb * 1024);
        hipError_t e2 = hipDeviceSynchronize(); hipError_t e3 = hipGetLastError();
        unsigned h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, kb * 1024);
        printf("%d KB: setattr %d sync %d last %d val %u (want %u) occ %d\n", kb, e, e2, e3, h, kb * 256 - 1, nb);
    }
}
