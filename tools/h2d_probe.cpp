// scratch: host -> HBM ingest rates on the GPU box (pageable hipMemcpy, hipHostRegister + async copy, staged pinned chunks with threads)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 4e9);
    char* h = (char*)aligned_alloc(4096, bytes);
    memset(h, 1, bytes);
    void* d; CK(hipMalloc(&d, bytes));
    hipStream_t s; CK(hipStreamCreate(&s));
    double t0 = now(); CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); double t1 = now();
    printf("pageable hipMemcpy          %.3f s  %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e9);
    t0 = now(); CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); t1 = now();
    printf("pageable hipMemcpy (2nd)    %.3f s  %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e9);
    t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); t1 = now();
    printf("hipHostRegister             %.3f s  %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e9);
    t0 = now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t1 = now();
    printf("registered async copy       %.3f s  %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e9);
    t0 = now(); CK(hipHostUnregister(h)); t1 = now();
    printf("hipHostUnregister           %.3f s\n", t1 - t0);
    // chunked registration pipelined with the copies
    for (size_t chunk : {(size_t)64 << 20, (size_t)256 << 20, (size_t)1 << 30}) {
        t0 = now();
        for (size_t o = 0; o < bytes; o += chunk) {
            size_t c = bytes - o < chunk ? bytes - o : chunk;
            CK(hipHostRegister(h + o, c, hipHostRegisterDefault));
            CK(hipMemcpyAsync((char*)d + o, h + o, c, hipMemcpyHostToDevice, s));
        }
        CK(hipStreamSynchronize(s));
        t1 = now();
        printf("register+copy, %4zu MB chunks %.3f s  %.1f GB/s\n", chunk >> 20, t1 - t0, bytes / (t1 - t0) / 1e9);
        for (size_t o = 0; o < bytes; o += chunk) CK(hipHostUnregister(h + o));
    }
    // staged: T threads memcpy into pinned chunks, async copies behind them
    for (int T : {1, 4, 8, 16}) {
        const size_t chunk = (size_t)32 << 20; const int NB = 2 * T;
        std::vector<char*> pin(NB); std::vector<hipEvent_t> ev(NB);
        for (int i = 0; i < NB; ++i) { CK(hipHostMalloc((void**)&pin[i], chunk, hipHostMallocDefault)); CK(hipEventCreate(&ev[i])); }
        t0 = now();
        size_t nchunks = (bytes + chunk - 1) / chunk;
        std::vector<std::thread> th;
        std::vector<hipStream_t> ss(T);
        for (int t = 0; t < T; ++t) CK(hipStreamCreate(&ss[t]));
        for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
            int slot = 0;
            for (size_t k = t; k < nchunks; k += T, slot ^= 1) {
                int b = 2 * t + slot;
                hipEventSynchronize(ev[b]);
                size_t o = k * chunk, c = bytes - o < chunk ? bytes - o : chunk;
                memcpy(pin[b], h + o, c);
                hipMemcpyAsync((char*)d + o, pin[b], c, hipMemcpyHostToDevice, ss[t]);
                hipEventRecord(ev[b], ss[t]);
            }
            hipStreamSynchronize(ss[t]);
        });
        for (auto& x : th) x.join();
        t1 = now();
        printf("staged, %2d threads           %.3f s  %.1f GB/s\n", T, t1 - t0, bytes / (t1 - t0) / 1e9);
        for (int i = 0; i < NB; ++i) { hipHostFree(pin[i]); hipEventDestroy(ev[i]); }
        for (int t = 0; t < T; ++t) hipStreamDestroy(ss[t]);
    }
    return 0;
}
