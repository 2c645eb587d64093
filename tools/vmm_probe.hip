// tools/vmm_probe.hip — is the "pool placement" effect a property of PHYSICAL chunks?
//
// Physical memory is created in 1-GiB chunks with the HIP virtual-memory API (hipMemCreate), every chunk is mapped
// on its own and used as the output of the 4K Gaussian over a fixed 32-frame input; the per-chunk rates are
// printed (three sweeps, to see whether a chunk's rate is stable).  Then the 8 fastest and the 8 slowest chunks
// are mapped back to back into two 8-GiB ranges and the 256-frame launch bench.py times is run on both, and on a
// plain hipMalloc pool.
//
//   hipcc --offload-arch=gfx950 -O2 -I include tools/vmm_probe.hip -L <pkg>/lib -lmi355_imgfilter -Wl,-rpath,<pkg>/lib
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mi355_imgfilter.h"

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            std::printf("HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                               \
        }                                                                           \
    } while (0)
#define MK(x)                                                   \
    do {                                                        \
        int r_ = (x);                                           \
        if (r_ != MI355_OK) {                                   \
            std::printf("mi355 error %d at %s:%d\n", r_, __FILE__, __LINE__); \
            return 1;                                           \
        }                                                       \
    } while (0)

static const int W = 3840, H = 2160;
static const size_t FRAME = (size_t)W * H * 4;

static int time_launch(mi355_ctx* ctx, const void* in, void* out, int frames, int warm, int reps, float* ms)
{
    for (int i = 0; i < warm; i++)
        MK(mi355_filter_dev(ctx, MI355_FILTER_GAUSS, in, out, W, H, frames, 5, 1.5f));
    MK(mi355_timer_begin(ctx));
    for (int i = 0; i < reps; i++)
        MK(mi355_filter_dev(ctx, MI355_FILTER_GAUSS, in, out, W, H, frames, 5, 1.5f));
    MK(mi355_timer_end(ctx, ms));
    *ms /= reps;
    return 0;
}

int main(int argc, char** argv)
{
    const int nchunks = argc > 1 ? std::atoi(argv[1]) : 40;
    mi355_ctx* ctx = nullptr;
    MK(mi355_ctx_create(0, &ctx));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran_min = 0, gran_rec = 0;
    CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
    std::printf("granularity: min %zu, recommended %zu\n", gran_min, gran_rec);
    const size_t CH = (size_t)1 << 30;
    const int FR = 32;  // 32 x 4K RGBA = 1.0617e9 B <= 1 GiB

    void* d_in = nullptr;
    CK(hipMalloc(&d_in, FRAME * 256));
    MK(mi355_synth_rgba8_dev(ctx, d_in, W, H, 256, 0, 0x5EED, 0));
    MK(mi355_sync(ctx));

    std::vector<hipMemGenericAllocationHandle_t> hnd(nchunks);
    hipDeviceptr_t va = nullptr;
    CK(hipMemAddressReserve(&va, CH * nchunks, 0, nullptr, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int i = 0; i < nchunks; i++) {
        CK(hipMemCreate(&hnd[i], CH, &prop, 0));
        CK(hipMemMap((char*)va + CH * i, CH, 0, hnd[i], 0));
    }
    CK(hipMemSetAccess(va, CH * nchunks, &acc, 1));

    // clocks up
    float ms = 0;
    if (time_launch(ctx, d_in, va, FR, 30, 10, &ms))
        return 1;
    std::vector<double> rate(nchunks, 0.0);
    for (int sweep = 0; sweep < 3; sweep++) {
        std::printf("sweep %d (GB/s per 1-GiB chunk, 32-frame launches):", sweep);
        for (int i = 0; i < nchunks; i++) {
            // a different 32-frame slice of the input per chunk, so reads come from HBM
            const void* in = (const char*)d_in + FRAME * FR * (i % 8);
            if (time_launch(ctx, in, (char*)va + CH * i, FR, 2, 8, &ms))
                return 1;
            const double gbs = 2.0 * FRAME * FR / (ms * 1e-3) / 1e9;
            rate[i] += gbs / 3.0;
            std::printf(" %.0f", gbs);
        }
        std::printf("\n");
    }
    std::vector<int> order(nchunks);
    for (int i = 0; i < nchunks; i++)
        order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return rate[a] > rate[b]; });
    std::printf("chunks by mean rate:");
    for (int i : order)
        std::printf(" %d:%.0f", i, rate[i]);
    std::printf("\n");

    // composite 8-GiB pools from the 8 fastest / 8 slowest chunks (the same physical handle may be mapped twice)
    hipDeviceptr_t va_fast = nullptr, va_slow = nullptr;
    CK(hipMemAddressReserve(&va_fast, CH * 8, 0, nullptr, 0));
    CK(hipMemAddressReserve(&va_slow, CH * 8, 0, nullptr, 0));
    for (int j = 0; j < 8; j++) {
        CK(hipMemMap((char*)va_fast + CH * j, CH, 0, hnd[order[j]], 0));
        CK(hipMemMap((char*)va_slow + CH * j, CH, 0, hnd[order[nchunks - 1 - j]], 0));
    }
    CK(hipMemSetAccess(va_fast, CH * 8, &acc, 1));
    CK(hipMemSetAccess(va_slow, CH * 8, &acc, 1));
    void* plain = nullptr;
    CK(hipMalloc(&plain, FRAME * 256));
    for (int round = 0; round < 3; round++) {
        float a = 0, b = 0, c = 0, d = 0;
        if (time_launch(ctx, d_in, va_fast, 256, 4, 12, &a) || time_launch(ctx, d_in, va_slow, 256, 4, 12, &b) ||
            time_launch(ctx, d_in, plain, 256, 4, 12, &c) || time_launch(ctx, d_in, va, 256, 4, 12, &d))
            return 1;
        const double by = 2.0 * FRAME * 256 / 1e9;
        std::printf("round %d, 256 frames: fast-8 %.0f GB/s | slow-8 %.0f | plain hipMalloc %.0f | first 8 chunks in creation order %.0f\n",
                    round, by / (a * 1e-3), by / (b * 1e-3), by / (c * 1e-3), by / (d * 1e-3));
    }
    MK(mi355_sync(ctx));
    mi355_ctx_destroy(ctx);
    return 0;
}
