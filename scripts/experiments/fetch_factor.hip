// Microbenchmark: what rocprofv3's FETCH_SIZE reports for the traversal kernels' access shape.
// The guide's x2 correction for gfx950 is calibrated on wide coalesced streaming reads only
// (/opt/skills/guides/MI355X_MICROARCH.md, "HBM"); the traversal kernels gather 64-B records (one record per lane,
// four global_load_dwordx4 to consecutive addresses), half a 128-B line each. This program reads a KNOWN number of
// bytes in exactly that shape, from a table the size of the 1 M-triangle scene's fast tree (63 MB: Infinity-Cache
// resident) and from a 1 GiB table (HBM), plus the streaming shape as the control, one kernel name per case so that
//     rocprofv3 --pmc FETCH_SIZE -- ./fetch_factor        (and a pass with --kernel-trace --stats for the times)
// gives FETCH_SIZE per case; factor = bytes the kernel asked for / (FETCH_SIZE x 1024).
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_factor fetch_factor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Rec { float4 a, b, c, d; };  // 64 B, like PairRec / TriRec

// every lane reads ONE whole 64-B record chosen by `perm` (a permutation of the table's records: each record is
// read exactly once per launch, both halves of every 128-B line at unrelated times)
template <int TAG>
__global__ void __launch_bounds__(256) k_gather64(const Rec* __restrict__ tab, const uint32_t* __restrict__ perm, uint32_t n,
                                                  float* __restrict__ sink) {
    float acc = 0.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Rec r = tab[perm[i]];
        acc += (r.a.x + r.b.y) + (r.c.z + r.d.w);
    }
    if (acc == 12345.678f) *sink = acc;
}
// the same number of records, only the first 32 B of each (a k_verify NodeRec: two dwordx4)
template <int TAG>
__global__ void __launch_bounds__(256) k_gather32(const Rec* __restrict__ tab, const uint32_t* __restrict__ perm, uint32_t n,
                                                  float* __restrict__ sink) {
    float acc = 0.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Rec* r = tab + perm[i];
        const float4 a = r->a, b = r->b;
        acc += a.x + b.y;
    }
    if (acc == 12345.678f) *sink = acc;
}
// whole 128-B lines: every lane reads TWO adjacent records (a 128-B aligned pair), n / 2 lanes' worth
template <int TAG>
__global__ void __launch_bounds__(256) k_gather128(const Rec* __restrict__ tab, const uint32_t* __restrict__ perm, uint32_t n,
                                                   float* __restrict__ sink) {
    float acc = 0.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n / 2; i += gridDim.x * blockDim.x) {
        const Rec* p = tab + (size_t)(perm[i] & ~1u);
        const Rec r = p[0], q = p[1];
        acc += (r.a.x + r.b.y) + (r.c.z + r.d.w) + (q.a.x + q.b.y) + (q.c.z + q.d.w);
    }
    if (acc == 12345.678f) *sink = acc;
}
// control: the guide's calibrated shape, 16 B per lane, consecutive lanes consecutive addresses
template <int TAG>
__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ tab, size_t n16, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = tab[i];
        acc += v.x + v.w;
    }
    if (acc == 12345.678f) *sink = acc;
}

template <int TAG>
static void run_case(const char* what, size_t table_bytes, float* sink) {
    const uint32_t n = (uint32_t)(table_bytes / sizeof(Rec));
    Rec* tab;
    uint32_t* perm;
    CHK(hipMalloc(&tab, (size_t)n * sizeof(Rec)));
    CHK(hipMalloc(&perm, (size_t)n * 4));
    CHK(hipMemset(tab, 0, (size_t)n * sizeof(Rec)));
    std::vector<uint32_t> h(n);
    for (uint32_t i = 0; i < n; i++) h[i] = i;
    uint64_t s = 0x9E3779B97F4A7C15ull + TAG;  // Fisher-Yates with splitmix64
    auto next = [&]() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    for (uint32_t i = n - 1; i > 0; i--) { const uint32_t j = (uint32_t)(next() % (i + 1)); std::swap(h[i], h[j]); }
    CHK(hipMemcpy(perm, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    const dim3 grid(256 * 8), block(256);
    auto timed = [&](auto launch, const char* shape, double bytes) {
        float best = 1e30f;
        for (int it = 0; it < 4; it++) {
            CHK(hipEventRecord(a));
            launch();
            CHK(hipEventRecord(b));
            CHK(hipEventSynchronize(b));
            float ms;
            CHK(hipEventElapsedTime(&ms, a, b));
            if (it && ms < best) best = ms;
        }
        printf("%-10s %-34s asked %12.0f B (+ %10.0f B of indices)  %8.3f ms  %7.1f GB/s\n", what, shape, bytes,
               shape[0] == 's' ? 0.0 : (double)n * 4, best, bytes / best / 1e6);
    };
    timed([&] { hipLaunchKernelGGL((k_gather64<TAG>), grid, block, 0, 0, tab, perm, n, sink); }, "gather 64 B records (4 x dwordx4)", (double)n * 64);
    timed([&] { hipLaunchKernelGGL((k_gather32<TAG>), grid, block, 0, 0, tab, perm, n, sink); }, "gather 32 B of each record", (double)n * 32);
    timed([&] { hipLaunchKernelGGL((k_gather128<TAG>), grid, block, 0, 0, tab, perm, n, sink); }, "gather n/2 aligned 128 B pairs", (double)(n / 2) * 128);
    timed([&] { hipLaunchKernelGGL((k_stream<TAG>), grid, block, 0, 0, (const float4*)tab, (size_t)n * 4, sink); }, "stream, 16 B per lane", (double)n * 64);
    CHK(hipFree(tab));
    CHK(hipFree(perm));
}

int main() {
    float* sink;
    CHK(hipMalloc(&sink, 4));
    run_case<0>("63 MB", (size_t)63 << 20, sink);   // kernels k_gather64<0> ...: the fast tree's size, Infinity-Cache resident
    run_case<1>("1 GiB", (size_t)1 << 30, sink);    // k_gather64<1> ...: beyond the Infinity Cache
    return 0;
}
