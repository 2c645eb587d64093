// gauss_tile.hip — general Gaussian blur (any odd k <= 63, any width/height), LDS-tiled.
//
// Replaces kernel `gaussian_blur` (RT/kernel/gaussian_base.cl:1-50) with the semantics of the
// reference CPU path (src/GaussianBlur/GaussianBlur.cpp:234-261): clamp-to-edge taps, all four
// channels, no division by the accumulated weight, truncation.
//
// One workgroup (256 threads = 4 waves) produces a 64x16 output tile.  The RGBA tile plus its
// k/2 halo is staged once in LDS (one dword per pixel, clamped addresses at the image border), so
// every input pixel is fetched from L2/HBM once per tile instead of k*k times per pixel as in the
// reference kernel.
//   FAST  : separable.  Pass V (vertical, k taps) turns the staged tile into float4 rows in LDS;
//           pass H (horizontal, k taps) reads them back.  Canonical op order, shared with the
//           register-resident kernel in gauss_slide.hip so both give identical bits:
//             v = w1[0]*r[0];  v = fma(w1[j], r[j], v)  j = 1..k-1      (top to bottom)
//             o = w1[0]*v[0];  o = fma(w1[t], v[t], o)  t = 1..k-1      (left to right)
//   EXACT : the CPU path's own arithmetic — k*k taps, ky outer / kx inner, separate multiply and
//           add (the library is built with -ffp-contract=off), so the result is bit-identical.
// Bound: HBM at small k (8 B/px algorithmic); FP32 VALU at large k (2k FMA per channel).
#include "common.hpp"
#include "kernels.hpp"

namespace mi355 {

namespace {

constexpr int kTW = 64;
constexpr int kTH = 16;
constexpr int kThreads = 256;

template <bool EXACT>
__global__ __launch_bounds__(kThreads) void gauss_tile_kernel(const uint32_t* __restrict__ in,
                                                              uint32_t* __restrict__ out, int w,
                                                              int h, int tiles_x, int tiles_y, int k,
                                                              const float* __restrict__ d_wt,
                                                              uint32_t ntiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = k >> 1;
    const int RW = kTW + 2 * R;
    const int RH = kTH + 2 * R;
    // carve: [V float4 TH*RW (FAST only)] [raw u32 RH*RW] [weights]
    f32x4* V = reinterpret_cast<f32x4*>(smem);
    uint32_t* raw = reinterpret_cast<uint32_t*>(smem + (EXACT ? 0 : (size_t)kTH * RW * 16));
    float* wt = reinterpret_cast<float*>(raw + RH * RW);

    const uint32_t tile = xcd_remap(blockIdx.x, ntiles);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const size_t frame = tile / ((uint32_t)tiles_x * tiles_y);
    const uint32_t* fin = in + frame * (size_t)w * h;
    uint32_t* fout = out + frame * (size_t)w * h;
    const int x0 = tx * kTW, y0 = ty * kTH;
    const int tid = threadIdx.x;

    const int nwt = EXACT ? k * k : k;
    for (int i = tid; i < nwt; i += kThreads)
        wt[i] = d_wt[i];
    for (int i = tid; i < RH * RW; i += kThreads) {
        const int ly = i / RW, lx = i - ly * RW;
        const int gy = clampi(y0 - R + ly, 0, h - 1);
        const int gx = clampi(x0 - R + lx, 0, w - 1);
        raw[i] = fin[(size_t)gy * w + gx];
    }
    __syncthreads();

    if constexpr (EXACT) {
        const int lx = tid & (kTW - 1);
        for (int ly = tid / kTW; ly < kTH; ly += kThreads / kTW) {
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            for (int ky = 0; ky < k; ky++) {
                const uint32_t* row = raw + (ly + ky) * RW + lx;
                const float* wrow = wt + ky * k;
                for (int kx = 0; kx < k; kx++) {
                    const uint32_t p = row[kx];
                    const float wv = wrow[kx];
                    // sum += pixel[c] * weight : u8 -> int -> float, multiply, then add
                    s0 += (float)(p & 0xFFu) * wv;
                    s1 += (float)((p >> 8) & 0xFFu) * wv;
                    s2 += (float)((p >> 16) & 0xFFu) * wv;
                    s3 += (float)(p >> 24) * wv;
                }
            }
            const int gx = x0 + lx, gy = y0 + ly;
            if (gx < w && gy < h)
                fout[(size_t)gy * w + gx] =
                    f2u8(s0) | (f2u8(s1) << 8) | (f2u8(s2) << 16) | (f2u8(s3) << 24);
        }
    } else {
        // pass V: every column of the staged tile, output rows only
        for (int i = tid; i < kTH * RW; i += kThreads) {
            const int ly = i / RW, cx = i - ly * RW;
            const uint32_t* col = raw + ly * RW + cx;
            uint32_t p = col[0];
            float wv = wt[0];
            float v0 = wv * (float)(p & 0xFFu);
            float v1 = wv * (float)((p >> 8) & 0xFFu);
            float v2 = wv * (float)((p >> 16) & 0xFFu);
            float v3 = wv * (float)(p >> 24);
            for (int j = 1; j < k; j++) {
                p = col[j * RW];
                wv = wt[j];
                v0 = __builtin_fmaf(wv, (float)(p & 0xFFu), v0);
                v1 = __builtin_fmaf(wv, (float)((p >> 8) & 0xFFu), v1);
                v2 = __builtin_fmaf(wv, (float)((p >> 16) & 0xFFu), v2);
                v3 = __builtin_fmaf(wv, (float)(p >> 24), v3);
            }
            f32x4 v;
            v.x = v0;
            v.y = v1;
            v.z = v2;
            v.w = v3;
            V[i] = v;
        }
        __syncthreads();
        // pass H
        const int lx = tid & (kTW - 1);
        for (int ly = tid / kTW; ly < kTH; ly += kThreads / kTW) {
            const f32x4* vr = V + ly * RW + lx;
            f32x4 a = vr[0];
            float wv = wt[0];
            float o0 = wv * a.x, o1 = wv * a.y, o2 = wv * a.z, o3 = wv * a.w;
            for (int t = 1; t < k; t++) {
                a = vr[t];
                wv = wt[t];
                o0 = __builtin_fmaf(wv, a.x, o0);
                o1 = __builtin_fmaf(wv, a.y, o1);
                o2 = __builtin_fmaf(wv, a.z, o2);
                o3 = __builtin_fmaf(wv, a.w, o3);
            }
            const int gx = x0 + lx, gy = y0 + ly;
            if (gx < w && gy < h)
                fout[(size_t)gy * w + gx] =
                    f2u8(o0) | (f2u8(o1) << 8) | (f2u8(o2) << 16) | (f2u8(o3) << 24);
        }
    }
}

}  // namespace

hipError_t launch_gauss_tile(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                             int nframes, const GaussCoef& coef, bool exact)
{
    const int k = coef.k, R = k / 2;
    const int tiles_x = (w + kTW - 1) / kTW, tiles_y = (h + kTH - 1) / kTH;
    const size_t ntiles = (size_t)tiles_x * tiles_y * nframes;
    if (ntiles > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    const int RW = kTW + 2 * R, RH = kTH + 2 * R;
    size_t lds = (size_t)RH * RW * 4 + (size_t)(exact ? k * k : k) * 4;
    if (!exact)
        lds += (size_t)kTH * RW * 16;
    hipError_t e;
    if (exact) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_tile_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL(gauss_tile_kernel<true>, dim3((unsigned)ntiles), dim3(kThreads), lds, stream,
                           reinterpret_cast<const uint32_t*>(d_in), reinterpret_cast<uint32_t*>(d_out),
                           w, h, tiles_x, tiles_y, k, coef.d_w2d, (uint32_t)ntiles);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_tile_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL(gauss_tile_kernel<false>, dim3((unsigned)ntiles), dim3(kThreads), lds, stream,
                           reinterpret_cast<const uint32_t*>(d_in), reinterpret_cast<uint32_t*>(d_out),
                           w, h, tiles_x, tiles_y, k, coef.d_w1d, (uint32_t)ntiles);
    }
    return hipGetLastError();
}

}  // namespace mi355
