// pipe_slide8.hip — the fused gray -> Gaussian -> Sobel kernel of pipe_slide.hip with EIGHT pixels per lane, for
// k in {3, 5}, width % 8 == 0, aligned buffers.  gfx950 only.  Same definition (SURVEY.md §8a "a-pipe",
// oracle_pipeline_rgba), same arithmetic ("exact by exception", exact_common.hpp), the same bits as pipe_slide.hip.
//
// Why: pipe_slide.hip is bound by the work of its waves (its rate is proportional to the lanes a strip uses:
// tools/lanes_sweep_pipeline.sh), ~160 instructions per wave-row of 240 pixels.  With 8 pixels per lane a wave-row
// covers 480 pixels and everything that is per ROW rather than per pixel is paid half as often — row control and
// address arithmetic, the flag test's tree / ballot / branch, the neighbour-lane taps of both stencils (4 + 4 DPP reads per
// row either way), the two halo lanes — and the 1-byte output leaves as 8 bytes per lane (480-byte spans = whole
// 32-byte sectors).  The price is registers: two 8-float rings instead of 4-float ones, 4-5 waves per SIMD instead of 8.
//
// Structure as in pipe_slide.hip: one wave per (frame, band, strip of <= 62 lanes + 1 halo lane per side);
//   row in (2 x 16 B per lane) -> luma (8 floats) -> ring of the last K gray rows -> vertical sums, symmetric pair form ->
//   horizontal taps (DPP at the lane's ends) -> S + delta -> flag test -> [exact chain] -> trunc -> l = LUT[b] ->
//   3-row ring of l -> Sobel row -> 8 bytes stored.  Odd bands walk upward.
#include <cmath>
#include <cstdlib>

#include "common.hpp"
#include "exact_common.hpp"
#include "kernels.hpp"
#include "slide_common.hpp"

namespace mi355 {

namespace {

constexpr int kWavesPerBlock = kSlideWavesPerBlock;
constexpr int PX = 8;

template <int K>
struct P8Tables {
    float w1[K];
    float w2[K * K];
    float delta;
};

struct Row8 {
    u32x4 a, b;  // pixels 0..3, 4..7
};

template <int R, bool CLAMP>
__global__ __launch_bounds__(kWavesPerBlock * 64, 4) void pipe_slide8_kernel(const uint8_t* __restrict__ in,
                                                                         uint8_t* __restrict__ out, int w, int h,
                                                                         int nstrips, int lanes_out, BandPlan plan,
                                                                         P8Tables<2 * R + 1> tab)
{
    constexpr int K = 2 * R + 1;
    __shared__ uint8_t lut[256];  // lut[b] = luma(b, b, b), the reference double-precision formula
    __shared__ float flat[256];   // flat[c] = the CPU path's chain over a window that is c everywhere (exact_common.hpp)
    lut[threadIdx.x] = (uint8_t)luma_rgb(threadIdx.x, threadIdx.x, threadIdx.x);
    flat[threadIdx.x] = flat_chain<K>((float)threadIdx.x, tab.w2);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    SlideItem it;
    if (!slide_item(plan, nstrips, h, &it))
        return;
    const int strip = it.strip, y0 = it.y0, nout = it.nout;
    const size_t frame = it.frame;
    const bool up = (it.band & 1) != 0;  // wave-uniform

    const int o_lane = strip * lanes_out + lane - 1;  // octet (8 pixels) of this lane
    const int octs = w >> 3;
    const int o_load = clampi(o_lane, 0, min(octs - 1, (strip + 1) * lanes_out));  // idle lanes re-load the halo octet
    const bool left_of_image = o_lane < 0, right_of_image = o_lane >= octs;
    const bool edge_strip = (strip == 0) || (8 * (strip * lanes_out + 63) > w);  // wave-uniform
    const int o_end = min((strip + 1) * lanes_out, octs);
    const bool stores = (lane >= 1) && (o_lane < o_end);
    const int keep_px = (lane == 0) ? PX - 1 : ((o_lane == o_end) ? 0 : -1);  // the pixel a halo lane's neighbour reads

    const int nin = nout + 2 + 2 * R;
    const int y_first = up ? y0 + nout + R : y0 - 1 - R;
    const int y_step = up ? -1 : 1;

    const size_t row_bytes = (size_t)w * 4;
    const auto fin = uniform_ptr(in + frame * row_bytes * h);
    const auto fout = uniform_ptr(out + frame * (size_t)w * h);
    uint32_t in_off = (uint32_t)o_load * 32u;
    uint32_t out_off = (uint32_t)(stores ? o_lane : 0) * 8u;

    float wv[R + 1];  // wv[d] = weight at distance d from the centre
#pragma unroll
    for (int d = 0; d <= R; d++)
        wv[d] = tab.w1[R - d];
    const float delta = tab.delta, two_delta = 2.0f * tab.delta;

    auto load_row = [&](int i) -> Row8 {
        const int y = clampi(y_first + y_step * min(i, nin - 1), 0, h - 1);  // gray rows: clamp-to-edge
        const auto rowp = fin + (size_t)y * row_bytes;
        lane_offset_here(in_off);
        Row8 r;
        r.a = gload<u32x4>(rowp + in_off);
        r.b = gload<u32x4>(rowp + in_off + 16);
        return r;
    };

    constexpr int PF = 3;
    Row8 q[K];
#pragma unroll
    for (int u = 0; u < PF; u++)
        q[u] = load_row(u);

    float g[K][PX];  // ring of the last K gray rows; slot = arrival index % K
    float l[3][PX];  // l rows of the last three blurred rows; slot as in pipe_slide.hip
#pragma unroll
    for (int s = 0; s < K; s++)
#pragma unroll
        for (int e = 0; e < PX; e++)
            g[s][e] = 0.0f;
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int e = 0; e < PX; e++)
            l[s][e] = 0.0f;

    for (int base = 0; base < nin; base += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            const int i = base + u;
            const int s3 = (u + 2) % 3;  // static after unrolling
            Row8 p = q[u];
            q[(u + PF) % K] = load_row(i + PF);
            if (edge_strip) {
                if (left_of_image) {  // gray image clamps: replicate column 0
                    p.a = u32x4{p.a.x, p.a.x, p.a.x, p.a.x};
                    p.b = p.a;
                }
                if (right_of_image) {  // replicate column w-1
                    p.b = u32x4{p.b.w, p.b.w, p.b.w, p.b.w};
                    p.a = p.b;
                }
            }
            luma_quad_int(p.a, &g[u][0], lut);
            luma_quad_int(p.b, &g[u][4], lut);
            if (i >= 2 * R) {
                // vertical pass, symmetric pair form; window = arrival rows i-2R .. i = slots (u+1+t) % K
                float v[PX];
#pragma unroll
                for (int e = 0; e < PX; e++) {
                    float acc = wv[0] * g[(u + 1 + R) % K][e];
#pragma unroll
                    for (int d = 1; d <= R; d++)
                        acc = __builtin_fmaf(wv[d], g[(u + 1 + R - d) % K][e] + g[(u + 1 + R + d) % K][e], acc);
                    v[e] = acc;
                }
                // horizontal pass; S' = S + delta rides on the centre tap (one-sided integer test)
                float S[PX], t[PX];
#pragma unroll
                for (int px = 0; px < PX; px++) {
                    float acc = __builtin_fmaf(wv[0], v[px], delta);
#pragma unroll
                    for (int d = 1; d <= R; d++) {
                        const int a = px - d, b = px + d;
                        const float va = (a < 0) ? dppl(v[PX + a]) : v[a];
                        const float vb = (b > PX - 1) ? dppr(v[b - PX]) : v[b];
                        acc = __builtin_fmaf(wv[d], va + vb, acc);
                    }
                    S[px] = acc;
                    t[px] = __builtin_amdgcn_fractf(acc);
                }
                const float tmin = fminf(fminf(fminf(t[0], t[1]), fminf(t[2], t[3])), fminf(fminf(t[4], t[5]), fminf(t[6], t[7])));
                const uint64_t flagged = __builtin_amdgcn_ballot_w64(tmin < two_delta);
                if (__builtin_expect(flagged != 0, 0)) {
                    if (dense_flags(flagged)) {  // flat content: constant windows take a table read, not the chain
                        if (!stores) {
                            // a halo lane owes its neighbour ONE blurred pixel (the column next to the strip), idle lanes none
#pragma unroll
                            for (int J = 0; J < PX; J++)
                                if (J != keep_px)
                                    t[J] = 1.0f;
                        }
                        flat_windows<K, PX>(g, S, t, two_delta, flat);
                    }
#define MI355_EXACT_PX(J)                                                          \
    if (__builtin_amdgcn_ballot_w64(t[J] < two_delta) != 0) {                        \
        if (up)                                                                    \
            S[J] = exact_sum<K, J, true, PX>(g, u, tab.w2);                        \
        else                                                                       \
            S[J] = exact_sum<K, J, false, PX>(g, u, tab.w2);                       \
    }
                    MI355_EXACT_PX(0)
                    MI355_EXACT_PX(1)
                    MI355_EXACT_PX(2)
                    MI355_EXACT_PX(3)
                    MI355_EXACT_PX(4)
                    MI355_EXACT_PX(5)
                    MI355_EXACT_PX(6)
                    MI355_EXACT_PX(7)
#undef MI355_EXACT_PX
                }
                float* lb = l[s3];
#pragma unroll
                for (int px = 0; px < PX; px++) {
                    float sum = S[px];
                    if constexpr (CLAMP)
                        sum = fminf(sum, 255.0f);
                    const uint32_t bq = (uint32_t)sum;  // truncation, as the Gaussian call stores it
                    lb[px] = (float)lut[bq];            // luma(b,b,b) re-applied
                }
                if (edge_strip) {
                    // the blurred image reflects (BORDER_REFLECT_101): x = -1 <- x = 1, x = w <- x = w-2
                    const float from_right = dppr(lb[1]);  // lane+1's pixel 1
                    const float from_left = dppl(lb[6]);   // lane-1's pixel 6
                    if (left_of_image)
                        lb[7] = from_right;
                    if (right_of_image)
                        lb[0] = from_left;
                }
                const int c = i - 2 * R;
                const int m = up ? y0 + nout - c + 1 : y0 - 2 + c;
                if (c >= 2 && m >= y0 && m < y0 + nout) {
                    const float* lm = l[(s3 + 2) % 3];  // blurred row m
                    const float* lo = l[(s3 + 1) % 3];  // the neighbour row that arrived first
                    float cs[PX], cd[PX];
                    if (__builtin_expect(m == 0 || m == h - 1, 0)) {
                        asm volatile("; first / last image row");  // keeps this a real (never-taken) branch
                        const int y_new = up ? m - 1 : m + 1;  // image row of the newest blurred row (lb)
                        const bool new_outside = y_new < 0 || y_new >= h;
#pragma unroll
                        for (int j = 0; j < PX; j++) {
                            const float nb = new_outside ? lo[j] : lb[j];  // (h >= 2: exactly one is outside)
                            cs[j] = __builtin_fmaf(2.0f, lm[j], nb) + nb;
                            cd[j] = 0.0f;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < PX; j++) {
                            cs[j] = __builtin_fmaf(2.0f, lm[j], lo[j]) + lb[j];
                            cd[j] = lb[j] - lo[j];  // sign depends on the walking direction; only gy^2 is used
                        }
                    }
                    const float csl = dppl(cs[PX - 1]), csr = dppr(cs[0]);
                    const float cdl = dppl(cd[PX - 1]), cdr = dppr(cd[0]);
                    float gxs[PX], gys[PX];
#pragma unroll
                    for (int j = 0; j < PX; j++) {
                        const float sl = (j == 0) ? csl : cs[j - 1], sr = (j == PX - 1) ? csr : cs[j + 1];
                        const float dl = (j == 0) ? cdl : cd[j - 1], dr = (j == PX - 1) ? cdr : cd[j + 1];
                        gxs[j] = sr - sl;
                        gys[j] = __builtin_fmaf(2.0f, cd[j], dl) + dr;
                    }
                    uint32_t r0 = sobel_mag_quad(&gxs[0], &gys[0]), r1 = sobel_mag_quad(&gxs[4], &gys[4]);
                    asm volatile("" : "+v"(r0), "+v"(r1));  // all 64 lanes, before the store's lane mask (DPP folds)
                    if (stores) {
                        const auto rowp = fout + (size_t)m * w;
                        lane_offset_here(out_off);
                        gstore_nt<u32x2>(rowp + out_off, u32x2{r0, r1});
                    }
                }
            }
        }
        // the last two rows written sit in slots K % 3 (older) and (K + 1) % 3 (newer): bring them to 0 and 1
        if constexpr (K % 3 == 2) {  // K = 5: older in 2, newer in 0
#pragma unroll
            for (int e = 0; e < PX; e++) {
                l[1][e] = l[0][e];
                l[0][e] = l[2][e];
            }
        }
    }
}

template <int R>
hipError_t launch_r8(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                     const GaussCoef& coef)
{
    constexpr int K = 2 * R + 1;
    const int octs = w / 8;
    const int nstrips = (octs + kSlideLanesOutMax - 1) / kSlideLanesOutMax;
    const int lanes_out = (octs + nstrips - 1) / nstrips;  // 4K: 480 octets = 8 strips x 60 lanes
    BandPlan plan;
    // Tall bands: this kernel's waves are few and long (4-5 per SIMD), and each band pays 2R + 2 warm-up rows.  Same
    // box, 256 x 4K frames (tools/pipe8_sweep.sh): k = 5: 24 rows 4.44 TB/s, 48: 4.70, 72: 4.73, 96: 4.82, 144: 4.80,
    // 216: 4.70 (pipe_slide.hip: 4.67); k = 3: 16 rows 4.91, 32: 5.14, 48: 5.24, 72: 5.33 (pipe_slide.hip: 4.94).
    // Smaller launches get shorter bands (make_band_plan).
    constexpr int kRowsMin = (R == 1) ? 16 : 24, kRowsMax = (R == 1) ? 72 : 96;
    int rows_min = kRowsMin, rows_max = kRowsMax;
    if (const char* e = tune_env("MI355_TUNE_PIPE8_ROWS"))
        rows_min = rows_max = atoi(e);
    if (!make_band_plan(h, nstrips, nframes, (R == 1) ? 5 : 4, rows_min, rows_max, rows_min, 0.0, kRowsMin / 2, &plan))
        return hipErrorInvalidValue;
    P8Tables<K> tab;
    double wsum = 0.0;
    for (int j = 0; j < K; j++) {
        tab.w1[j] = coef.h_w1d[j];
        wsum += (double)coef.h_w1d[j];
    }
    for (int j = 0; j < K * K; j++)
        tab.w2[j] = coef.h_w2d[j];
    tab.delta = (float)delta_bound<K>(tab.w1, tab.w2);
    const bool clamp = !(255.0 * wsum * wsum * 1.0001 < 256.0);
    const dim3 grid(plan.nblocks_a + plan.nblocks_b), block(kWavesPerBlock * 64);
    if (clamp)
        hipLaunchKernelGGL((pipe_slide8_kernel<R, true>), grid, block, 0, stream, d_in, d_out, w, h, nstrips, lanes_out, plan, tab);
    else
        hipLaunchKernelGGL((pipe_slide8_kernel<R, false>), grid, block, 0, stream, d_in, d_out, w, h, nstrips, lanes_out, plan, tab);
    return hipGetLastError();
}

}  // namespace

// what pipe_slide.hip needs, and: k in {3, 5}, width a multiple of 8 (>= 16), 16-byte aligned input, 8-byte aligned
// output
bool pipe_slide8_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef)
{
    if (!pipe_slide_supported(d_in, d_out, w, h, coef))
        return false;
    if (coef.k != 3 && coef.k != 5)
        return false;
    if ((w & 7) != 0 || w < 16)
        return false;
    return (reinterpret_cast<uintptr_t>(d_in) & 15u) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 7u) == 0;
}

hipError_t launch_pipe_slide8(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                              const GaussCoef& coef)
{
    switch (coef.k) {
    case 3: return launch_r8<1>(stream, d_in, d_out, w, h, nframes, coef);
    case 5: return launch_r8<2>(stream, d_in, d_out, w, h, nframes, coef);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mi355
