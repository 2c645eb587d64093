// pipe_slide.hip — fused gray -> Gaussian -> Sobel on RGBA8 frames, register-resident sliding window,
// FAST Gaussian arithmetic, k in {3,5,7}, width >= 4, height >= 2 (RAGGED instantiation when width % 4 != 0
// or the pointers are not 16-byte aligned).  gfx950 only.
//
// Definition (SURVEY.md §8a "a-pipe", oracle_pipeline_rgba): exactly the composition of the three API calls
//   g = luma(R,G,B)                                   src/Grayscale/grayscale.cpp:237
//   b = trunc(clamp(Gaussian_k(g)))  clamp-to-edge     src/GaussianBlur/GaussianBlur.cpp:234-261 on (g,g,g,255)
//   l = luma(b,b,b)                  RE-APPLIED        (l != b for 65 byte values)
//   out = Sobel(l)                   reflect-101       src/EdgeDetection/EdgeDetection.cpp:219-240
// and, because the Gaussian uses the canonical FAST op order of gauss_slide.hip / gauss_tile.hip, the output
// is bit-identical to mi355_sobel(mi355_gauss(mi355_gray(x))) of this library (tests/test_gpu_parity.py).
//
// One wave per (frame, band, strip of <= 62 lanes + 1 halo lane per side), a lane owns 4 pixels:
//   row in -> luma (4 floats) -> K vertical accumulators (4 floats each) -> finished vertical sum ->
//   horizontal taps (neighbour lanes through DPP) -> trunc -> l = LUT[b] (256-byte table in LDS: luma(b,b,b)
//   always sits on the ambiguous S % 1000 == 0 case, so it is tabulated once per workgroup with the FP64
//   formula) -> 3-row ring of l -> Sobel row in fp32 -> 4 bytes stored.
// 4 B read + 1 B written per pixel; nothing intermediate touches memory (the three separate calls move
// 8 + 8 + 5 B/px).  Border rules: gray columns/rows clamp (replicated halo lane / clamped row index); the
// blurred image reflects: column x=-1 takes x=1 and x=w takes x=w-2 by a DPP fix-up in the two edge strips,
// row -1 takes row 1 and row h takes row h-2 by swapping ring slots at the first / last image row.
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"
#include "slide_common.hpp"

namespace mi355 {

namespace {

constexpr int kWavesPerBlock = kSlideWavesPerBlock;

template <int K>
struct PWeights {
    float w[K];
};

__device__ __forceinline__ float dppl(float v)  // lane l <- lane l-1
{
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}

__device__ __forceinline__ float dppr(float v)  // lane l <- lane l+1
{
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

// RAGGED = width % 4 != 0 or unaligned buffers (see gauss_slide.hip / sobel_slide.hip): unaligned 16-byte row
// accesses in interior strips, per-pixel clamped loads and per-byte stores in the two edge strips, and the
// reflected column x = w of the blurred image may sit anywhere inside a lane.
template <int R, bool CLAMP, bool RAGGED>
__global__ __launch_bounds__(kWavesPerBlock * 64) void pipe_slide_kernel(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int w, int h, int nstrips,
    int lanes_out, BandPlan plan, PWeights<2 * R + 1> wts)
{
    constexpr int K = 2 * R + 1;
    __shared__ uint8_t lut[256];  // lut[b] = luma(b, b, b), the reference double-precision formula
    lut[threadIdx.x] = (uint8_t)luma_rgb(threadIdx.x, threadIdx.x, threadIdx.x);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    SlideItem it;
    if (!slide_item(plan, nstrips, h, &it))
        return;  // (pipeline: after the only barrier)
    const int strip = it.strip, y0 = it.y0, nout = it.nout;
    const size_t frame = it.frame;

    const int q_lane = strip * lanes_out + lane - 1;
    const int quads = (w + 3) >> 2;
    const int q_load = clampi(q_lane, 0, min(quads - 1, (strip + 1) * lanes_out));  // idle lanes re-load the halo quad
    const bool left_of_image = q_lane < 0, right_of_image = q_lane >= quads;
    const bool edge_strip = (strip == 0) || (4 * (strip * lanes_out + 63) > w);  // wave-uniform
    const int x_lane = 4 * q_lane;
    const int jw = w - x_lane;  // RAGGED: position of column x = w inside this lane, if 0 <= jw <= 3
    const int q_end = min((strip + 1) * lanes_out, quads);
    const bool stores = (lane >= 1) && (q_lane < q_end);

    // output rows y0 .. y0+nout-1 need blurred rows y0-1 .. y0+nout, which need gray rows y0-1-R .. y0+nout+R
    const int nin = nout + 2 + 2 * R;

    const size_t row_bytes = (size_t)w * 4;
    const auto fin = uniform_ptr(in + frame * row_bytes * h);
    const auto fout = uniform_ptr(out + frame * (size_t)w * h);
    uint32_t in_off = (uint32_t)q_load * 16u;
    uint32_t out_off = (uint32_t)(stores ? q_lane : 0) * 4u;
    uint32_t px_off[4];  // RAGGED edge strips: the gray image clamps
#pragma unroll
    for (int j = 0; j < 4; j++)
        px_off[j] = (uint32_t)clampi(x_lane + j, 0, w - 1) * 4u;

    float wv[K];
#pragma unroll
    for (int j = 0; j < K; j++)
        wv[j] = wts.w[j];

    auto load_row = [&](int i) -> u32x4 {
        const int y = clampi(y0 - 1 - R + min(i, nin - 1), 0, h - 1);  // gray rows: clamp-to-edge
        const auto rowp = fin + (size_t)y * row_bytes;  // SGPR pair; + 32-bit lane offset = saddr form
        lane_offset_here(in_off);
        if constexpr (RAGGED) {
            u32x4 r;
            if (edge_strip) {  // wave-uniform
#pragma unroll
                for (int j = 0; j < 4; j++)
                    r[j] = gload<uint32_t>(rowp + px_off[j]);
            } else {
                r = gload_a4<u32x4>(rowp + in_off);
            }
            return r;
        } else {
            return gload<u32x4>(rowp + in_off);
        }
    };

    constexpr int PF = 3;
    u32x4 q[K];
#pragma unroll
    for (int u = 0; u < PF; u++)
        q[u] = load_row(u);

    float acc[K][4] = {};
    float l[3][4] = {};  // l rows of the last three blurred rows; slot = input row index % 3

    // One trip = 3K input rows, so that both rings (K vertical accumulators, 3 Sobel rows) have static slots:
    // no register moves, no per-row selects.  K-row groups past the band's last input row are skipped.
    for (int base = 0; base < nin; base += 3 * K) {
#pragma unroll
        for (int u3 = 0; u3 < 3; u3++) {
            if (base + u3 * K < nin) {  // wave-uniform
#pragma unroll
                for (int u = 0; u < K; u++) {
                    const int i = base + u3 * K + u;
                    const int s3 = (u3 * K + u) % 3;  // static after unrolling
                    u32x4 p = q[u];
                    q[(u + PF) % K] = load_row(i + PF);
                    if constexpr (!RAGGED) {
                        if (edge_strip) {
                            if (left_of_image)
                                p = u32x4{p.x, p.x, p.x, p.x};  // gray image clamps: replicate column 0
                            if (right_of_image)
                                p = u32x4{p.w, p.w, p.w, p.w};  // replicate column w-1
                        }
                    }
                    float g[4];
                    luma_quad_fast(p, g);
                    // vertical pass (canonical order): gray row i is tap j of blurred row i - j
#pragma unroll
                    for (int j = 0; j < K; j++) {
                        const int s = (u - j + K) % K;
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            acc[s][e] = (j == 0) ? wv[0] * g[e] : __builtin_fmaf(wv[j], g[e], acc[s][e]);
                    }
                    // Stage gating with scalar branches (i, y0, nout live in SGPRs, EXEC stays full for the DPP
                    // reads): the first 2R rows of a band only feed the vertical accumulators, the next two only
                    // fill the 3-row ring, and rows past the band's last output are never stored.
                    if (i >= 2 * R) {
                        const float* v = acc[(u + 1) % K];  // vertical sum of the blurred row that just completed
                        // horizontal pass (canonical order), neighbour-lane taps through DPP
                        float* lb = l[s3];
#pragma unroll
                        for (int px = 0; px < 4; px++) {
                            float sum = 0.0f;
#pragma unroll
                            for (int t = 0; t < K; t++) {
                                const int s = px - R + t;
                                const float src = (s < 0) ? dppl(v[4 + s]) : ((s > 3) ? dppr(v[s - 4]) : v[s]);
                                sum = (t == 0) ? wv[0] * src : __builtin_fmaf(wv[t], src, sum);
                            }
                            if constexpr (CLAMP)
                                sum = fminf(sum, 255.0f);
                            const uint32_t bq = (uint32_t)sum;  // truncation, as the Gaussian call stores it
                            lb[px] = (float)lut[bq];            // luma(b,b,b) re-applied
                        }
                        if (edge_strip) {
                            // the blurred image reflects (BORDER_REFLECT_101): x = -1 <- x = 1, x = w <- x = w-2
                            const float from_right = dppr(lb[1]);  // lane+1's pixel 1
                            const float from_left = dppl(lb[2]);   // lane-1's pixel 2
                            if (left_of_image)
                                lb[3] = from_right;
                            if constexpr (!RAGGED) {
                                if (right_of_image)
                                    lb[0] = from_left;
                            } else {
                                // column x = w is pixel jw of this lane; its mirror x = w-2 is pixel jw-2 of this
                                // lane or pixel jw+2 of the lane to the left (w >= 4 here)
                                const float from_left3 = dppl(lb[3]);
                                const float l0 = lb[0], l1 = lb[1];
                                if (jw == 0)
                                    lb[0] = from_left;
                                if (jw == 1)
                                    lb[1] = from_left3;
                                if (jw == 2)
                                    lb[2] = l0;
                                if (jw == 3)
                                    lb[3] = l1;
                            }
                        }
                        // blurred row just finished: image row yb = y0 - 1 + (i - 2R); Sobel output row m = yb - 1
                        const int m = y0 - 2 + i - 2 * R;
                        if (m >= y0 && m < y0 + nout) {
                            const float* lm = l[(s3 + 2) % 3];  // blurred row m
                            const float* lt = l[(s3 + 1) % 3];  // blurred row m - 1
                            // rows reflect too: at m = 0 the top row (-1) is row 1 = the bottom row; at m = h-1 the
                            // bottom row (h) is row h-2 = the top row.  Wave-uniform and rare: a branch, not selects.
                            const float* top = lt;
                            const float* bot = lb;
                            float cs[4], cd[4];
                            if (__builtin_expect(m == 0 || m == h - 1, 0)) {
                                // keeps this a real (never-taken) branch: hipcc otherwise if-converts both arms into
                                // 8 v_cndmask per row on the common path
                                asm volatile("; first / last image row");
#pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    const float tv = (m == 0) ? lb[j] : lt[j];
                                    const float bv = (m == h - 1) ? lt[j] : lb[j];
                                    cs[j] = __builtin_fmaf(2.0f, lm[j], tv) + bv;
                                    cd[j] = bv - tv;
                                }
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    cs[j] = __builtin_fmaf(2.0f, lm[j], top[j]) + bot[j];
                                    cd[j] = bot[j] - top[j];
                                }
                            }
                            const float csl = dppl(cs[3]), csr = dppr(cs[0]);
                            const float cdl = dppl(cd[3]), cdr = dppr(cd[0]);
                            const float gx0 = cs[1] - csl, gx1 = cs[2] - cs[0], gx2 = cs[3] - cs[1], gx3 = csr - cs[2];
                            const float gy0 = __builtin_fmaf(2.0f, cd[0], cdl) + cd[1];
                            const float gy1 = __builtin_fmaf(2.0f, cd[1], cd[0]) + cd[2];
                            const float gy2 = __builtin_fmaf(2.0f, cd[2], cd[1]) + cd[3];
                            const float gy3 = __builtin_fmaf(2.0f, cd[3], cd[2]) + cdr;
                            const float gxs[4] = {gx0, gx1, gx2, gx3}, gys[4] = {gy0, gy1, gy2, gy3};
                    const uint32_t r = sobel_mag_quad(gxs, gys);
                            if (stores) {
                                const auto rowp = fout + (size_t)m * w;
                                lane_offset_here(out_off);
                                if constexpr (RAGGED) {
                                    if (edge_strip) {
#pragma unroll
                                        for (int j = 0; j < 4; j++)
                                            if (x_lane + j < w)
                                                rowp[out_off + j] = (uint8_t)(r >> (8 * j));
                                    } else {
                                        gstore_a1<uint32_t>(rowp + out_off, r);
                                    }
                                } else {
                                    gstore_nt<uint32_t>(rowp + out_off, r);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int R>
hipError_t launch_r(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                    const GaussCoef& coef)
{
    constexpr int K = 2 * R + 1;
    const StripPlan sp = make_strip_plan(w);
    // ~64 VGPRs -> 8 waves/SIMD.  Short bands win although every band spends 2R+2 warm-up rows: measured on
    // 256 x 4K frames, k = 5 (two kinds of MI355X box): 24 rows 4.6-4.8 TB/s, 48 rows 4.4-4.6,
    // 108 rows (the former adaptive plan) 4.1-4.4.  The kernel is not VALU-bound (removing 15 % of its VALU
    // instructions changed nothing), short bands keep the rows that are in flight close together in memory.
    BandPlan plan;
    constexpr int kRows = (R == 1) ? 16 : (R == 2 ? 24 : 40);
    if (!make_band_plan(h, sp.nstrips, nframes, 8, kRows, kRows, kRows, 0.0, kRows / 2, &plan))
        return hipErrorInvalidValue;
    PWeights<K> wts;
    double wsum = 0.0;
    for (int j = 0; j < K; j++) {
        wts.w[j] = coef.h_w1d[j];
        wsum += (double)coef.h_w1d[j];
    }
    const bool clamp = !(255.0 * wsum * wsum * 1.0001 < 256.0);
    const bool ragged = (w & 3) != 0 || (reinterpret_cast<uintptr_t>(d_in) & 15u) != 0 ||
                        (reinterpret_cast<uintptr_t>(d_out) & 3u) != 0;
    const dim3 grid(plan.nblocks_a + plan.nblocks_b), block(kWavesPerBlock * 64);
#define MI355_LAUNCH(CL, RG)                                                                                  \
    hipLaunchKernelGGL((pipe_slide_kernel<R, CL, RG>), grid, block, 0, stream, d_in, d_out, w, h, sp.nstrips,  \
                       sp.lanes_out, plan, wts)
    if (clamp && ragged)
        MI355_LAUNCH(true, true);
    else if (clamp)
        MI355_LAUNCH(true, false);
    else if (ragged)
        MI355_LAUNCH(false, true);
    else
        MI355_LAUNCH(false, false);
#undef MI355_LAUNCH
    return hipGetLastError();
}

}  // namespace

bool pipe_slide_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, int k)
{
    if (k != 3 && k != 5 && k != 7)
        return false;
    if (w < 4 || h < 2)
        return false;
    return (reinterpret_cast<uintptr_t>(d_in) & 3u) == 0;
}

hipError_t launch_pipe_slide(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                             const GaussCoef& coef)
{
    switch (coef.k) {
    case 3: return launch_r<1>(stream, d_in, d_out, w, h, nframes, coef);
    case 5: return launch_r<2>(stream, d_in, d_out, w, h, nframes, coef);
    case 7: return launch_r<3>(stream, d_in, d_out, w, h, nframes, coef);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mi355
