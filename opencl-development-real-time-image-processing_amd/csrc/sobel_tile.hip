// sobel_tile.hip — Sobel edge magnitude and the fused gray->Gaussian->Sobel pipeline, LDS-tiled,
// for any width/height and any odd k <= 63.
//
// Sobel replaces kernel `sobel_edge_detection` (RT/kernel/edge_base.cl:1-57) + ConvertToUChar
// (RT/src/Controller.cpp:76-85, :605) with the semantics of the reference CPU path
// (src/EdgeDetection/EdgeDetection.cpp:219-240: filter2D x2 (correlation, BORDER_REFLECT_101),
// magnitude, convertTo(CV_8UC1) = round-half-even + saturate) applied to
// gray = src/Grayscale/grayscale.cpp:237 of every RGBA pixel.  Every output pixel is written
// (the OpenCL kernel skips the border, edge_base.cl:12).  4 B in, 1 B out per pixel.
//
// The luminance is computed ONCE per pixel into an LDS tile with a 1-px halo (the reference kernel
// recomputes it for all 9 taps of every pixel, edge_base.cl:40-41); gx, gy are integers.
//
// Pipeline = exact composition of the three API calls (SURVEY.md §8a "a-pipe"):
//   g = luma(R,G,B); b = Gaussian_k(g) with clamp-to-edge taps and truncation (the single-channel
//   image of what the RGBA blur does to (g,g,g,255)); l = luma(b,b,b) RE-APPLIED; Sobel(l) with
//   reflect-101.  FAST / EXACT select the Gaussian arithmetic exactly as in gauss_tile.hip, so
//   pipeline(FAST) == sobel(gauss_FAST(gray(x))) bit for bit, and pipeline(EXACT) == the CPU chain.
#include "common.hpp"
#include "kernels.hpp"
#include "slide_common.hpp"

namespace mi355 {

namespace {

constexpr int kTW = 64;
constexpr int kTH = 16;
constexpr int kThreads = 256;
constexpr int kLW = kTW + 2;  // luma tile with 1-px halo
constexpr int kLH = kTH + 2;

// Sobel of the luma tile L (kLH x kLW ints in LDS) -> 4 consecutive pixels per thread.
__device__ __forceinline__ void sobel_from_tile(const int* L, uint8_t* fout, int w, int h, int x0,
                                                int y0, int tid, bool vec_store)
{
    const int ly = tid >> 4;          // 16 threads per row
    const int lx = (tid & 15) << 2;   // 4 px each
    const int gy_img = y0 + ly;
    if (gy_img >= h)
        return;
    const int* r0 = L + ly * kLW + lx;
    const int* r1 = r0 + kLW;
    const int* r2 = r1 + kLW;
    uint32_t packed = 0;
    uint32_t res[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int a = r0[i], b = r0[i + 1], c = r0[i + 2];
        const int d = r1[i], f = r1[i + 2];
        const int g = r2[i], hh = r2[i + 1], ii = r2[i + 2];
        const int sx = (c - a) + 2 * (f - d) + (ii - g);
        const int sy = (g + 2 * hh + ii) - (a + 2 * b + c);
        res[i] = sobel_mag_u8(sx, sy);
        packed |= res[i] << (8 * i);
    }
    const int gx_img = x0 + lx;
    uint8_t* o = fout + (size_t)gy_img * w + gx_img;
    if (vec_store && gx_img + 3 < w) {
        *reinterpret_cast<uint32_t*>(o) = packed;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (gx_img + i < w)
                o[i] = (uint8_t)res[i];
    }
}

__global__ __launch_bounds__(kThreads) void sobel_tile_kernel(const uint32_t* __restrict__ in,
                                                              uint8_t* __restrict__ out, int w, int h,
                                                              int tiles_x, int tiles_y,
                                                              uint32_t ntiles, int vec_store)
{
    __shared__ int L[kLH * kLW];
    const uint32_t tile = xcd_remap(blockIdx.x, ntiles);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const size_t frame = tile / ((uint32_t)tiles_x * tiles_y);
    const uint32_t* fin = in + frame * (size_t)w * h;
    uint8_t* fout = out + frame * (size_t)w * h;
    const int x0 = tx * kTW, y0 = ty * kTH;
    const int tid = threadIdx.x;

    for (int i = tid; i < kLH * kLW; i += kThreads) {
        const int ly = i / kLW, lx = i - ly * kLW;
        // positions beyond (w, h) are never consumed by a stored pixel: fold them onto w / h first
        const int py = min(y0 - 1 + ly, h), px = min(x0 - 1 + lx, w);
        const int gy = reflect101(py, h), gx = reflect101(px, w);
        L[i] = (int)luma_px_fast(fin[(size_t)gy * w + gx]);
    }
    __syncthreads();
    sobel_from_tile(L, fout, w, h, x0, y0, tid, vec_store != 0);
}

template <bool EXACT>
__global__ __launch_bounds__(kThreads) void pipeline_tile_kernel(const uint32_t* __restrict__ in,
                                                                 uint8_t* __restrict__ out, int w,
                                                                 int h, int tiles_x, int tiles_y,
                                                                 int k, const float* __restrict__ d_wt,
                                                                 uint32_t ntiles, int vec_store)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = k >> 1;
    const int GW = kTW + 2 * R + 2, GH = kTH + 2 * R + 2;
    // carve: [G float GH*GW] [V float kLH*GW (FAST)] [B int kLH*kLW] [weights]
    float* G = reinterpret_cast<float*>(smem);
    float* V = G + GH * GW;
    int* B = reinterpret_cast<int*>(V + (EXACT ? 0 : kLH * GW));
    float* wt = reinterpret_cast<float*>(B + kLH * kLW);

    const uint32_t tile = xcd_remap(blockIdx.x, ntiles);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const size_t frame = tile / ((uint32_t)tiles_x * tiles_y);
    const uint32_t* fin = in + frame * (size_t)w * h;
    uint8_t* fout = out + frame * (size_t)w * h;
    const int x0 = tx * kTW, y0 = ty * kTH;
    const int tid = threadIdx.x;

    const int nwt = EXACT ? k * k : k;
    for (int i = tid; i < nwt; i += kThreads)
        wt[i] = d_wt[i];
    // gray tile, clamp-to-edge: G(ly,lx) <-> image (y0-1-R+ly, x0-1-R+lx)
    for (int i = tid; i < GH * GW; i += kThreads) {
        const int ly = i / GW, lx = i - ly * GW;
        const int gy = clampi(y0 - 1 - R + ly, 0, h - 1);
        const int gx = clampi(x0 - 1 - R + lx, 0, w - 1);
        G[i] = luma_px_fast(fin[(size_t)gy * w + gx]);
    }
    __syncthreads();

    // blurred-and-regrayed tile: B(by,bx) <-> image (y0-1+by, x0-1+bx)
    if constexpr (EXACT) {
        for (int i = tid; i < kLH * kLW; i += kThreads) {
            const int by = i / kLW, bx = i - by * kLW;
            float s = 0.0f;
            for (int ky = 0; ky < k; ky++) {
                const float* row = G + (by + ky) * GW + bx;
                const float* wrow = wt + ky * k;
                for (int kx = 0; kx < k; kx++)
                    s += row[kx] * wrow[kx];
            }
            const uint32_t b = f2u8(s);
            B[i] = (int)luma_rgb(b, b, b);
        }
    } else {
        for (int i = tid; i < kLH * GW; i += kThreads) {
            const int by = i / GW, cx = i - by * GW;
            const float* col = G + by * GW + cx;
            float v = wt[0] * col[0];
            for (int j = 1; j < k; j++)
                v = __builtin_fmaf(wt[j], col[j * GW], v);
            V[i] = v;
        }
        __syncthreads();
        for (int i = tid; i < kLH * kLW; i += kThreads) {
            const int by = i / kLW, bx = i - by * kLW;
            const float* vr = V + by * GW + bx;
            float o = wt[0] * vr[0];
            for (int t = 1; t < k; t++)
                o = __builtin_fmaf(wt[t], vr[t], o);
            const uint32_t b = f2u8(o);
            B[i] = (int)luma_rgb(b, b, b);
        }
    }
    __syncthreads();
    // Sobel reads the blurred image with BORDER_REFLECT_101: halo positions that fall outside the
    // image (x = -1, x = w, y = -1, y = h) take the value of their mirror, which is an in-image
    // position of this same tile.  Only out-of-image entries are written, only in-image ones read.
    for (int i = tid; i < kLH * kLW; i += kThreads) {
        const int by = i / kLW, bx = i - by * kLW;
        const int iy = y0 - 1 + by, ix = x0 - 1 + bx;
        const bool oy = (iy < 0 || iy >= h), ox = (ix < 0 || ix >= w);
        if ((oy || ox) && iy <= h && ix <= w) {
            const int sy = reflect101(iy, h) - (y0 - 1);
            const int sx = reflect101(ix, w) - (x0 - 1);
            B[i] = B[sy * kLW + sx];
        }
    }
    __syncthreads();
    sobel_from_tile(B, fout, w, h, x0, y0, tid, vec_store != 0);
}

}  // namespace

hipError_t launch_sobel(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                        int nframes, int impl)
{
    if (impl != 1 && sobel_slide_supported(d_in, d_out, w, h))
        return launch_sobel_slide(stream, d_in, d_out, w, h, nframes);
    const int tiles_x = (w + kTW - 1) / kTW, tiles_y = (h + kTH - 1) / kTH;
    const size_t ntiles = (size_t)tiles_x * tiles_y * nframes;
    if (ntiles > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    const int vec = ((w & 3) == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 3u) == 0);
    hipLaunchKernelGGL(sobel_tile_kernel, dim3((unsigned)ntiles), dim3(kThreads), 0, stream,
                       reinterpret_cast<const uint32_t*>(d_in), d_out, w, h, tiles_x, tiles_y,
                       (uint32_t)ntiles, vec);
    return hipGetLastError();
}

hipError_t launch_pipeline(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                           int nframes, const GaussCoef& coef, bool exact, int impl)
{
    // the sliding-window kernel is bit-exact with the CPU chain ("exact by exception", pipe_slide.hip), so it serves
    // both Gaussian modes; tables it cannot take (non-separable, asymmetric factor) arrive here with exact = true
    {
        // 8 pixels per lane (pipe_slide8.hip, the same bits) pays on big launches of wide rows at k = 5: same box,
        // 4K frames, 4 / 8 pixels per lane (tools/pipe8_ab.sh): 256 frames 4.57 / 4.82 TB/s (another box 4.67 / 4.73),
        // 128 frames 4.53 / 4.71, 64 frames 4.31 / 4.30, 8 frames 4.02 / 3.87, 1 frame 2.26 / 1.77; 640 x 512 x 4096
        // 4.17 / 3.66 (80 octets = 2 strips of 40 lanes); k = 3: 4.84 / 4.91 on one box, 4.93 / 4.82 on another.
        const int octs = w / 8, strips8 = (octs + kSlideLanesOutMax - 1) / kSlideLanesOutMax;
        const int lanes8 = strips8 > 0 ? (octs + strips8 - 1) / strips8 : 0;
        bool want8 = coef.k == 5 && lanes8 >= 56 && (size_t)w * h * nframes >= 1000000000ull;
        if (const char* e = tune_env("MI355_PIPE8"))  // tuning build: 1 forces the 8-pixel kernel, 0 forbids it
            want8 = atoi(e) != 0;
        if (impl != 1 && want8 && pipe_slide8_supported(d_in, d_out, w, h, coef))
            return launch_pipe_slide8(stream, d_in, d_out, w, h, nframes, coef);
    }
    if (impl != 1 && pipe_slide_supported(d_in, d_out, w, h, coef))
        return launch_pipe_slide(stream, d_in, d_out, w, h, nframes, coef);
    const int k = coef.k, R = k / 2;
    const int tiles_x = (w + kTW - 1) / kTW, tiles_y = (h + kTH - 1) / kTH;
    const size_t ntiles = (size_t)tiles_x * tiles_y * nframes;
    if (ntiles > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    const int GW = kTW + 2 * R + 2, GH = kTH + 2 * R + 2;
    size_t lds = (size_t)GH * GW * 4 + (size_t)kLH * kLW * 4 + (size_t)(exact ? k * k : k) * 4;
    if (!exact)
        lds += (size_t)kLH * GW * 4;
    const int vec = ((w & 3) == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 3u) == 0);
    hipError_t e;
    if (exact) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(pipeline_tile_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL(pipeline_tile_kernel<true>, dim3((unsigned)ntiles), dim3(kThreads), lds,
                           stream, reinterpret_cast<const uint32_t*>(d_in), d_out, w, h, tiles_x,
                           tiles_y, k, coef.d_w2d, (uint32_t)ntiles, vec);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(pipeline_tile_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL(pipeline_tile_kernel<false>, dim3((unsigned)ntiles), dim3(kThreads), lds,
                           stream, reinterpret_cast<const uint32_t*>(d_in), d_out, w, h, tiles_x,
                           tiles_y, k, coef.d_w1d, (uint32_t)ntiles, vec);
    }
    return hipGetLastError();
}

}  // namespace mi355
