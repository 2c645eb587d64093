// gray.hip — luminance of RGBA8 frames on gfx950.
//
// Replaces kernel `grayscale` (RT/kernel/grayscale_base.cl:1-19) + the host pass
// Controller::ConvertToUChar (RT/src/Controller.cpp:76-85): u8 in, u8 out, no float4 round trip
// (the reference writes 16 B/px and converts on one host thread).  Gray value = the CPU path's
// double-precision formula (src/Grayscale/grayscale.cpp:237), bit-exact.
//
// Pure streaming, 4 px (16 B) per lane per access, 4 independent accesses in flight per lane.  Two shapes:
// gray_strip_kernel (rows of 4-pixel multiples: a wave walks a 1-KiB-wide strip of a 4-row band, as the
// sliding-window kernels do) and gray_vec_kernel (anything else: grid-stride over the flat pixel array; frames
// are tightly packed, so a batch is one array).
// Algorithmic bytes: 8 B/px (RGBA out) or 5 B/px (1-channel out).  Bound: HBM.
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"
#include "slide_common.hpp"

namespace mi355 {

namespace {

constexpr int kGrayThreads = 256;
constexpr int kGrayIlp = 4;

__device__ __forceinline__ uint32_t luma_fast_u32(uint32_t px) { return (uint32_t)luma_px_fast(px); }

__device__ __forceinline__ uint32_t pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    return a | (b << 8) | (c << 16) | (d << 24);
}

template <bool ONE_CH>
__global__ __launch_bounds__(kGrayThreads) void gray_vec_kernel(const u32x4* __restrict__ in,
                                                                void* __restrict__ out,
                                                                size_t nquads)
{
    const size_t stride = (size_t)gridDim.x * kGrayThreads;
    size_t i = (size_t)blockIdx.x * kGrayThreads + threadIdx.x;
    for (; i + (kGrayIlp - 1) * stride < nquads; i += kGrayIlp * stride) {
        u32x4 p[kGrayIlp];
#pragma unroll
        for (int u = 0; u < kGrayIlp; u++)
            p[u] = __builtin_nontemporal_load(&in[i + u * stride]);
#pragma unroll
        for (int u = 0; u < kGrayIlp; u++) {
            const uint32_t g0 = luma_fast_u32(p[u].x), g1 = luma_fast_u32(p[u].y);
            const uint32_t g2 = luma_fast_u32(p[u].z), g3 = luma_fast_u32(p[u].w);
            if constexpr (ONE_CH) {
                __builtin_nontemporal_store(pack4(g0, g1, g2, g3),
                                            &reinterpret_cast<uint32_t*>(out)[i + u * stride]);
            } else {
                u32x4 o;
                o.x = gray_to_rgba(g0);
                o.y = gray_to_rgba(g1);
                o.z = gray_to_rgba(g2);
                o.w = gray_to_rgba(g3);
                __builtin_nontemporal_store(o, &reinterpret_cast<u32x4*>(out)[i + u * stride]);
            }
        }
    }
    for (; i < nquads; i += stride) {
        const u32x4 p = in[i];
        const uint32_t g0 = luma_fast_u32(p.x), g1 = luma_fast_u32(p.y), g2 = luma_fast_u32(p.z), g3 = luma_fast_u32(p.w);
        if constexpr (ONE_CH) {
            reinterpret_cast<uint32_t*>(out)[i] = pack4(g0, g1, g2, g3);
        } else {
            u32x4 o;
            o.x = gray_to_rgba(g0);
            o.y = gray_to_rgba(g1);
            o.z = gray_to_rgba(g2);
            o.w = gray_to_rgba(g3);
            reinterpret_cast<u32x4*>(out)[i] = o;
        }
    }
}

// Same arithmetic, the access shape of the sliding-window kernels: a wave owns a 1-KiB-wide column strip (64 lanes
// x 4 px) of a band of rows and walks down it, 4 rows in flight.  Work items are numbered strip-fastest, so what
// the chip has in flight is one compact window of the batch.
template <bool ONE_CH>
__global__ __launch_bounds__(kSlideWavesPerBlock * 64) void gray_strip_kernel(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int quads, int h, int nstrips, BandPlan plan)
{
    SlideItem it;
    if (!slide_item(plan, nstrips, h, &it))
        return;
    const int lane = threadIdx.x & 63;
    const int q = it.strip * 64 + lane;
    if (q >= quads)
        return;
    const size_t row_in = (size_t)quads * 16, row_out = ONE_CH ? (size_t)quads * 4 : row_in;
    const auto fin = uniform_ptr(in + it.frame * row_in * h);
    const auto fout = uniform_ptr(out + it.frame * row_out * h);
    uint32_t in_off = (uint32_t)q * 16u, out_off = (uint32_t)q * (ONE_CH ? 4u : 16u);
    constexpr int U = 4;
    for (int r0 = 0; r0 < it.nout; r0 += U) {
        u32x4 p[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int y = it.y0 + min(r0 + u, it.nout - 1);
            lane_offset_here(in_off);
            p[u] = __builtin_nontemporal_load((global_ptr<const u32x4>)(fin + (size_t)y * row_in + in_off));
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (r0 + u < it.nout) {  // wave-uniform
                float gf[4];
                luma_quad_fast(p[u], gf);
                const uint32_t g0 = (uint32_t)gf[0], g1 = (uint32_t)gf[1], g2 = (uint32_t)gf[2], g3 = (uint32_t)gf[3];
                const auto rowp = fout + (size_t)(it.y0 + r0 + u) * row_out;
                lane_offset_here(out_off);
                if constexpr (ONE_CH) {
                    gstore_nt<uint32_t>(rowp + out_off, pack4(g0, g1, g2, g3));
                } else {
                    u32x4 o;
                    o.x = gray_to_rgba(g0);
                    o.y = gray_to_rgba(g1);
                    o.z = gray_to_rgba(g2);
                    o.w = gray_to_rgba(g3);
                    gstore_nt<u32x4>(rowp + out_off, o);
                }
            }
        }
    }
}

// one pixel per thread: tails and buffers that are not 16-byte aligned
template <bool ONE_CH>
__global__ __launch_bounds__(kGrayThreads) void gray_px_kernel(const uint8_t* __restrict__ in,
                                                               uint8_t* __restrict__ out,
                                                               size_t first, size_t npx)
{
    const size_t stride = (size_t)gridDim.x * kGrayThreads;
    for (size_t i = first + (size_t)blockIdx.x * kGrayThreads + threadIdx.x; i < npx; i += stride) {
        const uint32_t g = luma_rgb(in[4 * i], in[4 * i + 1], in[4 * i + 2]);
        if constexpr (ONE_CH) {
            out[i] = (uint8_t)g;
        } else {
            out[4 * i] = (uint8_t)g;
            out[4 * i + 1] = (uint8_t)g;
            out[4 * i + 2] = (uint8_t)g;
            out[4 * i + 3] = 255;
        }
    }
}

inline unsigned grid_for(size_t items, unsigned per_block, unsigned cap)
{
    size_t b = (items + per_block - 1) / per_block;
    if (b < 1)
        b = 1;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace

hipError_t launch_gray(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                       int nframes, bool one_channel)
{
    const size_t npx = (size_t)w * h * nframes;
    const bool aligned = ((reinterpret_cast<uintptr_t>(d_in) & 15u) == 0) &&
                         ((reinterpret_cast<uintptr_t>(d_out) & (one_channel ? 3u : 15u)) == 0);
    const size_t nquads = aligned ? npx / 4 : 0;
    // grid-stride; measured on MI355X (tools/membench.hip): a flat 16 B/lane stream runs 5.3 TB/s with
    // 2,048 blocks, 6.2-6.4 TB/s with >= 8k blocks and non-temporal loads + stores; this kernel: 4.8 / 5.6 / 5.9 / 6.3 TB/s at 2k / 8k / 64k / 256k+ blocks
    static const unsigned kCap = [] {
        const char* e = tune_env("MI355_TUNE_GRAY_BLOCKS");  // tuning sweeps only
        return (e && atoi(e) > 0) ? (unsigned)atoi(e) : (1u << 20);
    }();
    // Rows of 4-pixel multiples take the strip-walk kernel with 4-row bands: measured on 256 x 4K frames, same
    // box, flat kernel 6.36 TB/s, strips of 4 / 8 / 16 / 32 rows 6.61 / 6.49 / 6.39 / 6.27 (1-channel output: 6.08
    // flat, 6.33 with 4 rows).  MI355_TUNE_GRAY_STRIP (tuning sweeps only): band height, 0 = flat kernel.
    static const bool kStripForced = tune_env("MI355_TUNE_GRAY_STRIP") != nullptr;  // tests: also on small shapes
    static const int kStripRows = [] {
        const char* e = tune_env("MI355_TUNE_GRAY_STRIP");
        return e ? atoi(e) : 4;
    }();
    // ... and only where it pays: big batches (8 x 4K frames: -2.5 %, 256: +2 %) of rows that fill the 64-lane
    // strips (64-pixel-wide frames would leave 48 lanes idle: -23 %)
    const int strip_quads = w / 4, strip_n = (strip_quads + 63) / 64;
    const bool strip_ok = (w & 3) == 0 && (kStripForced || (strip_quads * 100 >= strip_n * 64 * 93 && npx >= ((size_t)1 << 28)));
    if (nquads && kStripRows > 0 && strip_ok) {
        const int quads = strip_quads, nstrips = strip_n;
        BandPlan plan;
        if (!make_band_plan(h, nstrips, nframes, 8, kStripRows, kStripRows, kStripRows, 0.0, 0, &plan))
            return hipErrorInvalidValue;
        const dim3 grid(plan.nblocks_a + plan.nblocks_b), block(kSlideWavesPerBlock * 64);
        if (one_channel)
            hipLaunchKernelGGL(gray_strip_kernel<true>, grid, block, 0, stream, d_in, d_out, quads, h, nstrips, plan);
        else
            hipLaunchKernelGGL(gray_strip_kernel<false>, grid, block, 0, stream, d_in, d_out, quads, h, nstrips, plan);
        return hipGetLastError();
    }
    if (nquads) {
        const unsigned grid = grid_for(nquads, kGrayThreads * kGrayIlp, kCap);
        if (one_channel)
            hipLaunchKernelGGL(gray_vec_kernel<true>, dim3(grid), dim3(kGrayThreads), 0, stream,
                               reinterpret_cast<const u32x4*>(d_in), (void*)d_out, nquads);
        else
            hipLaunchKernelGGL(gray_vec_kernel<false>, dim3(grid), dim3(kGrayThreads), 0, stream,
                               reinterpret_cast<const u32x4*>(d_in), (void*)d_out, nquads);
    }
    const size_t done = nquads * 4;
    if (done < npx) {
        const unsigned grid = grid_for(npx - done, kGrayThreads, kCap);
        if (one_channel)
            hipLaunchKernelGGL(gray_px_kernel<true>, dim3(grid), dim3(kGrayThreads), 0, stream, d_in,
                               d_out, done, npx);
        else
            hipLaunchKernelGGL(gray_px_kernel<false>, dim3(grid), dim3(kGrayThreads), 0, stream, d_in,
                               d_out, done, npx);
    }
    return hipGetLastError();
}

}  // namespace mi355
