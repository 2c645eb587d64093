// tools/xcd_affinity.hip — does a 1-GiB slice of a device buffer have an affinity to an XCD?
// Workgroups are handed to the 8 XCDs round-robin (blockIdx % 8, observed); a launch in which only the blocks of ONE
// residue class do any work therefore runs on ONE XCD.  For every (XCD x, slice j) pair the active blocks stream-read
// (mode r), stream-write (mode w) or copy slice j -> slice j of a second buffer (mode c); the table printed is GB/s.
// Build: hipcc --offload-arch=gfx950 -O3 tools/xcd_affinity.hip -o tools/bin/xcd_affinity
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e__ = (x);                                                                   \
        if (e__ != hipSuccess) {                                                                \
            std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e__));                \
            std::exit(1);                                                                       \
        }                                                                                       \
    } while (0)

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

template <int MODE>  // 0 read, 1 write, 2 copy
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t nvec, int xcd,
                                             unsigned long long* sink)
{
    if ((int)(blockIdx.x & 7u) != xcd)
        return;
    const size_t nactive = gridDim.x / 8;
    const size_t me = (blockIdx.x >> 3) * 256 + threadIdx.x, stride = nactive * 256;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (size_t i = me; i < nvec; i += stride) {
        if (MODE == 0) {
            const u32x4 v = __builtin_nontemporal_load(&src[i]);
            acc ^= v;
        } else if (MODE == 1) {
            __builtin_nontemporal_store(u32x4{(uint32_t)i, 1u, 2u, 3u}, &dst[i]);
        } else {
            __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
        }
    }
    if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u)
        atomicAdd(sink, 1ull);
}

int main(int argc, char** argv)
{
    const int nslices = argc > 1 ? std::atoi(argv[1]) : 8;
    const size_t slice = (size_t)1 << 30;
    void *a = nullptr, *b = nullptr;
    unsigned long long* sink = nullptr;
    CK(hipMalloc(&a, slice * nslices));
    CK(hipMalloc(&b, slice * nslices));
    CK(hipMalloc((void**)&sink, 8));
    CK(hipMemset(a, 1, slice * nslices));
    CK(hipMemset(b, 2, slice * nslices));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t nvec = slice / 16;
    const unsigned grid = 8 * 4096;  // 4096 active blocks on the one XCD
    for (int mode = 0; mode < 3; mode++) {
        std::printf("mode %s: rows = XCD (blockIdx %% 8), columns = 1-GiB slice; GB/s of the one active XCD\n",
                    mode == 0 ? "read" : (mode == 1 ? "write" : "copy (slice j of A -> slice j of B)"));
        for (int x = 0; x < 8; x++) {
            std::printf("xcd %d:", x);
            for (int j = 0; j < nslices; j++) {
                const u32x4* s = (const u32x4*)((char*)a + slice * j);
                u32x4* d = (u32x4*)((char*)b + slice * j);
                float best = 1e30f;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0, 0));
                    if (mode == 0)
                        hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, s, d, nvec, x, sink);
                    else if (mode == 1)
                        hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, s, d, nvec, x, sink);
                    else
                        hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, s, d, nvec, x, sink);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best)
                        best = ms;
                }
                const double bytes = (mode == 2 ? 2.0 : 1.0) * (double)slice;
                std::printf(" %6.0f", bytes / (best * 1e-3) / 1e9);
            }
            std::printf("\n");
            std::fflush(stdout);
        }
    }
    CK(hipFree(a));
    CK(hipFree(b));
    return 0;
}
