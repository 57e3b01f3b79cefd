// Microbenchmark: service rate of device-scope atomicAdd from a chip-filling grid, by address pattern.
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip ; run: ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// every wave does `reps` atomics (one lane), on counter (wave_id % k) * stride_words
__global__ void k_atomics(unsigned* ctr, int k, int stride_words, int reps, unsigned* sink) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned acc = 0;
    if ((threadIdx.x & 63) == 0) {
        unsigned* p = ctr + (size_t)(wave % (unsigned)k) * stride_words;
        for (int r = 0; r < reps; r++) acc += atomicAdd(p, 1u);  // (returning: the next one waits for it, like a queue draw)
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}

int main() {
    unsigned *ctr, *sink;
    CHK(hipMalloc(&ctr, 1 << 20));
    CHK(hipMalloc(&sink, 4));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    const int blocks = 1792, threads = 256, reps = 16;  // 7 168 waves, as the traversal kernels' grid
    const double total = (double)blocks * (threads / 64) * reps;
    struct { int k, stride; const char* what; } cases[] = {
        {1, 0, "one address"},          {2, 1, "2 addresses, same 64 B line"},   {8, 1, "8 addresses, same 64 B line"},
        {8, 16, "8 addresses, 64 B apart"}, {8, 32, "8 addresses, 128 B apart"}, {8, 64, "8 addresses, 256 B apart"},
        {8, 1024, "8 addresses, 4 KiB apart"}, {64, 64, "64 addresses, 256 B apart"}, {7168, 1, "one address per wave, packed"},
    };
    for (auto& c : cases) {
        float best = 1e30f;
        for (int it = 0; it < 4; it++) {
            CHK(hipMemset(ctr, 0, 1 << 20));
            CHK(hipEventRecord(a));
            hipLaunchKernelGGL(k_atomics, dim3(blocks), dim3(threads), 0, 0, ctr, c.k, c.stride, reps, sink);
            CHK(hipEventRecord(b));
            CHK(hipEventSynchronize(b));
            float ms;
            CHK(hipEventElapsedTime(&ms, a, b));
            if (it && ms < best) best = ms;
        }
        printf("%-36s %8.3f ms  %7.2f ns per atomic  %8.1f M atomics/s\n", c.what, best, best * 1e6 / total, total / best / 1e3);
    }
    return 0;
}
