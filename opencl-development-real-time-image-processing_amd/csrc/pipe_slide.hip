// pipe_slide.hip — fused gray -> Gaussian -> Sobel on RGBA8 frames, register-resident sliding window,
// k in {3,5,7}, width >= 4, height >= 2 (RAGGED instantiation when width % 4 != 0 or the pointers are not
// 16-byte aligned).  gfx950 only.  BIT-EXACT with the reference's CPU chain in both Gaussian modes.
//
// Definition (SURVEY.md §8a "a-pipe", oracle_pipeline_rgba): exactly the composition of the three API calls
//   g = luma(R,G,B)                                   src/Grayscale/grayscale.cpp:237
//   b = trunc(clamp(Gaussian_k(g)))  clamp-to-edge     src/GaussianBlur/GaussianBlur.cpp:234-261 on (g,g,g,255)
//   l = luma(b,b,b)                  RE-APPLIED        (l != b for 65 byte values)
//   out = Sobel(l)                   reflect-101       src/EdgeDetection/EdgeDetection.cpp:219-240
//
// The Gaussian stage is "exact by exception" (the same idea as the luminance, common.hpp): the CPU path's value
// is trunc(S_cpu), S_cpu = the k*k-term float sum in its own order (ky outer, kx inner, separate multiply and
// add).  A separable fp32 evaluation S (2k multiply-adds instead of 2k*k operations) differs from S_cpu by at
// most delta, a bound the host derives rigorously from the two tables (launch_r: rounding errors of both
// evaluations + the mismatch of w1 (x) w1 against the 2-D table; ~3e-4 at k = 5).  So wherever S is further
// than delta from an integer, trunc(S) IS the CPU path's byte; the few pixels within delta (6e-4 of them on
// noise-like images) are recomputed with the CPU path's own operation sequence from the ring of gray rows the
// wave holds anyway.  Flat regions (S sits 5e-6 below an integer) take the exact chain for every pixel: correct,
// and ~2.5x slower there.
//
// One wave per (frame, band, strip of <= 62 lanes + 1 halo lane per side), a lane owns 4 pixels:
//   row in -> luma (4 floats) -> ring of the last K gray rows -> vertical sums of the row whose window is complete
//   (symmetric pair form: w[R] g_c + sum_d w[R-d] (g_{c-d} + g_{c+d}); the pair sums are exact integers, and the
//   form reads the same whichever way the band walks) -> horizontal taps (neighbour lanes through DPP) -> S ->
//   flag test -> [exact chain] -> trunc -> l = LUT[b] (256-byte table in LDS: luma(b,b,b) always sits on the
//   ambiguous S % 1000 == 0 case, so it is tabulated once per workgroup with the FP64 formula) -> 3-row ring of l
//   -> Sobel row in fp32 -> 4 bytes stored.
// 4 B read + 1 B written per pixel; nothing intermediate touches memory (the three separate calls move
// 8 + 8 + 5 B/px).  Odd bands walk UPWARD: a band and its lower neighbour then reach their 2R+2 shared boundary
// rows at the same moment (the end of both walks, or the start), and the second reader hits L2 instead of HBM.
// Border rules: gray columns/rows clamp (replicated halo lane / clamped row index); the blurred image reflects:
// column x=-1 takes x=1 and x=w takes x=w-2 by a DPP fix-up in the two edge strips, and a blurred row outside the
// image is replaced by its mirror, which is the OTHER neighbour row of the Sobel stencil.
#include <cmath>
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"
#include "slide_common.hpp"
#include "exact_common.hpp"

namespace mi355 {

namespace {

constexpr int kWavesPerBlock = kSlideWavesPerBlock;

template <int K>
struct PTables {
    float w1[K];      // separable factor (symmetric: w1[j] == w1[K-1-j])
    float w2[K * K];  // the reference's 2-D table, row-major [ky][kx] — read only by the exact chain
    float delta;      // |S - S_cpu| bound
};

// RAGGED = width % 4 != 0 or unaligned buffers (see gauss_slide.hip / sobel_slide.hip): unaligned 16-byte row
// accesses in interior strips, per-pixel clamped loads and per-byte stores in the two edge strips, and the
// reflected column x = w of the blurred image may sit anywhere inside a lane.
template <int R, bool CLAMP, bool RAGGED>
__global__ __launch_bounds__(kWavesPerBlock * 64) void pipe_slide_kernel(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int w, int h, int nstrips,
    int lanes_out, BandPlan plan, PTables<2 * R + 1> tab)
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
        return;  // (pipeline: after the only barrier)
    const int strip = it.strip, y0 = it.y0, nout = it.nout;
    const size_t frame = it.frame;
    const bool up = (it.band & 1) != 0;  // wave-uniform

    const int q_lane = strip * lanes_out + lane - 1;
    const int quads = (w + 3) >> 2;
    const int q_load = clampi(q_lane, 0, min(quads - 1, (strip + 1) * lanes_out));  // idle lanes re-load the halo quad
    const bool left_of_image = q_lane < 0, right_of_image = q_lane >= quads;
    const bool edge_strip = (strip == 0) || (4 * (strip * lanes_out + 63) > w);  // wave-uniform
    const int x_lane = 4 * q_lane;
    const int jw = w - x_lane;  // RAGGED: position of column x = w inside this lane, if 0 <= jw <= 3
    const int q_end = min((strip + 1) * lanes_out, quads);
    const bool stores = (lane >= 1) && (q_lane < q_end);
    const int keep_px = (lane == 0) ? 3 : ((q_lane == q_end) ? 0 : -1);  // the pixel a halo lane's neighbour reads

    // output rows y0 .. y0+nout-1 need blurred rows y0-1 .. y0+nout, which need gray rows y0-1-R .. y0+nout+R;
    // "arrival index" i = 0 .. nin-1 counts them in walking order (top down, or bottom up for odd bands)
    const int nin = nout + 2 + 2 * R;
    const int y_first = up ? y0 + nout + R : y0 - 1 - R;
    const int y_step = up ? -1 : 1;

    const size_t row_bytes = (size_t)w * 4;
    const auto fin = uniform_ptr(in + frame * row_bytes * h);
    const auto fout = uniform_ptr(out + frame * (size_t)w * h);
    uint32_t in_off = (uint32_t)q_load * 16u;
    uint32_t out_off = (uint32_t)(stores ? q_lane : 0) * 4u;
    uint32_t px_off[4];  // RAGGED edge strips: the gray image clamps
#pragma unroll
    for (int j = 0; j < 4; j++)
        px_off[j] = (uint32_t)clampi(x_lane + j, 0, w - 1) * 4u;

    float wv[R + 1];  // wv[d] = weight at distance d from the centre
#pragma unroll
    for (int d = 0; d <= R; d++)
        wv[d] = tab.w1[R - d];
    const float delta = tab.delta, two_delta = 2.0f * tab.delta;

    auto load_row = [&](int i) -> u32x4 {
        const int y = clampi(y_first + y_step * min(i, nin - 1), 0, h - 1);  // gray rows: clamp-to-edge
        const auto rowp = fin + (size_t)y * row_bytes;  // SGPR pair; + 32-bit lane offset = saddr form
        lane_offset_here(in_off);
        if constexpr (RAGGED) {
            u32x4 r;
            // edge strips: only the lanes that overlap the row's ends address their pixels one by one; the others take
            // the unaligned 16-byte access of the interior strips (gauss_slide.hip: +9 % at width 1023, +55 % at 427)
            if (edge_strip && !(x_lane >= 0 && x_lane + 3 < w)) {
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

    float g[K][4];  // ring of the last K gray rows; slot = arrival index % K
    float l[3][4];  // l rows of the last three blurred rows; slot = arrival index % 3
#pragma unroll
    for (int s = 0; s < K; s++)
#pragma unroll
        for (int e = 0; e < 4; e++)
            g[s][e] = 0.0f;
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int e = 0; e < 4; e++)
            l[s][e] = 0.0f;

    // One trip = K input rows: the gray ring's slots are static (slot = u).  The 3-row ring of l starts every trip
    // in the same phase — the previous trip's last two rows sit in slots 0 (older) and 1 (newer), row u writes slot
    // (u + 2) % 3 — which costs 8 register moves per trip when K % 3 != 0 and keeps its slots static as well.
    // (A 3K-row trip needs no moves, but with the exact chains in the body it is past what hipcc unrolls: it kept
    // the ring in LDS with computed slots instead.)
    for (int base = 0; base < nin; base += K) {
        {
            {
#pragma unroll
                for (int u = 0; u < K; u++) {
                    const int i = base + u;
                    const int s3 = (u + 2) % 3;  // static after unrolling
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
                    luma_quad_int(p, g[u], lut);
                    // Stage gating with scalar branches (i, y0, nout live in SGPRs, EXEC stays full for the DPP
                    // reads): the first 2R rows of a band only fill the gray ring, the next two only fill the
                    // 3-row ring, and rows past the band's last output are never stored.
                    if (i >= 2 * R) {
                        // window of the blurred row that just completed: arrival rows i-2R .. i = slots (u+1+t) % K
                        // vertical pass, symmetric pair form (the pair sums are exact integers <= 510)
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            float acc = wv[0] * g[(u + 1 + R) % K][e];
#pragma unroll
                            for (int d = 1; d <= R; d++)
                                acc = __builtin_fmaf(wv[d], g[(u + 1 + R - d) % K][e] + g[(u + 1 + R + d) % K][e], acc);
                            v[e] = acc;
                        }
                        // horizontal pass, same form; neighbour-lane taps through DPP
                        // S' = S + delta rides on the centre tap, so "S within delta of an integer n" reads
                        // "fract(S') < 2 delta", one-sided, and floor(S') = floor(S) everywhere else
                        float S[4];
#pragma unroll
                        for (int px = 0; px < 4; px++) {
                            float acc = __builtin_fmaf(wv[0], v[px], delta);
#pragma unroll
                            for (int d = 1; d <= R; d++) {
                                const int a = px - d, b = px + d;
                                const float va = (a < 0) ? dppl(v[4 + a]) : v[a];
                                const float vb = (b > 3) ? dppr(v[b - 4]) : v[b];
                                acc = __builtin_fmaf(wv[d], va + vb, acc);
                            }
                            S[px] = acc;
                        }
                        float t[4];  // fract is exact; S' > 0
#pragma unroll
                        for (int px = 0; px < 4; px++)
                            t[px] = __builtin_amdgcn_fractf(S[px]);
                        const float tmin = fminf(fminf(t[0], t[1]), fminf(t[2], t[3]));
                        const uint64_t flagged = __builtin_amdgcn_ballot_w64(tmin < two_delta);
                        if (__builtin_expect(flagged != 0, 0)) {
                            if (dense_flags(flagged)) {  // flat content: constant windows take a table read
                                if (!stores) {
                                    // a halo lane owes its neighbour one blurred pixel, idle lanes none
#pragma unroll
                                    for (int J = 0; J < 4; J++)
                                        if (J != keep_px)
                                            t[J] = 1.0f;
                                }
                                flat_windows<K, 4>(g, S, t, two_delta, flat);
                            }
                            // one wave-uniform branch per pixel position: only positions some lane flagged pay
#define MI355_EXACT_PX(J)                                                                                  \
    if (__builtin_amdgcn_ballot_w64(t[J] < two_delta) != 0) {                                              \
        if (up)                                                                                            \
            S[J] = exact_sum<K, J, true>(g, u, tab.w2);                                                    \
        else                                                                                               \
            S[J] = exact_sum<K, J, false>(g, u, tab.w2);                                                   \
    }
                            MI355_EXACT_PX(0)
                            MI355_EXACT_PX(1)
                            MI355_EXACT_PX(2)
                            MI355_EXACT_PX(3)
#undef MI355_EXACT_PX
                        }
                        float* lb = l[s3];
#pragma unroll
                        for (int px = 0; px < 4; px++) {
                            float sum = S[px];
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
                        // blurred arrival row c = i - 2R sits at image row yb(c); the Sobel row between the last
                        // three blurred rows is m = yb(c - 1)
                        const int c = i - 2 * R;
                        const int m = up ? y0 + nout - c + 1 : y0 - 2 + c;
                        if (c >= 2 && m >= y0 && m < y0 + nout) {
                            const float* lm = l[(s3 + 2) % 3];  // blurred row m
                            const float* lo = l[(s3 + 1) % 3];  // the neighbour row that arrived first
                            float cs[4], cd[4];
                            // A neighbour row outside the image is replaced by its mirror, which is the other
                            // neighbour (reflect-101): only at m = 0 and m = h-1.  Wave-uniform and rare: a branch.
                            if (__builtin_expect(m == 0 || m == h - 1, 0)) {
                                // keeps this a real (never-taken) branch: hipcc otherwise if-converts both arms into
                                // 8 v_cndmask per row on the common path
                                asm volatile("; first / last image row");
                                const int y_new = up ? m - 1 : m + 1;  // image row of the newest blurred row (lb)
                                const bool new_outside = y_new < 0 || y_new >= h;
#pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    const float nb = new_outside ? lo[j] : lb[j];  // (h >= 2: exactly one is outside)
                                    cs[j] = __builtin_fmaf(2.0f, lm[j], nb) + nb;
                                    cd[j] = 0.0f;
                                }
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    cs[j] = __builtin_fmaf(2.0f, lm[j], lo[j]) + lb[j];
                                    cd[j] = lb[j] - lo[j];  // sign depends on the walking direction; only gy^2 is used
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
                            uint32_t r = sobel_mag_quad(gxs, gys);
                            // computed by all 64 lanes, BEFORE the store's lane mask: hipcc otherwise sinks the stencil
                            // into the masked region and keeps the four lane shifts outside it as separate
                            // v_mov_b32_dpp (a DPP read under a partial EXEC sees zeros); here three of them fold into
                            // their consumers.  (3 of 160 instructions per row: +0.2 %, within noise.)
                            asm volatile("" : "+v"(r));
                            if (stores) {
                                const auto rowp = fout + (size_t)m * w;
                                lane_offset_here(out_off);
                                if constexpr (RAGGED) {
                                    if (edge_strip && x_lane + 3 >= w) {  // the last quad of a row may be partial
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
        // the last two rows written sit in slots K % 3 (older) and (K + 1) % 3 (newer): bring them to 0 and 1
        if constexpr (K % 3 == 2) {  // K = 5: older in 2, newer in 0
#pragma unroll
            for (int e = 0; e < 4; e++) {
                l[1][e] = l[0][e];
                l[0][e] = l[2][e];
            }
        } else if constexpr (K % 3 == 1) {  // K = 7: older in 1, newer in 2
#pragma unroll
            for (int e = 0; e < 4; e++) {
                l[0][e] = l[1][e];
                l[1][e] = l[2][e];
            }
        }
    }
}

template <int R>
hipError_t launch_r(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                    const GaussCoef& coef, const float* h_w2d)
{
    constexpr int K = 2 * R + 1;
    const StripPlan sp = make_strip_plan(w);
    BandPlan plan;
    constexpr int kRows = (R == 1) ? 16 : (R == 2 ? 24 : 40);
    if (!make_band_plan(h, sp.nstrips, nframes, 8, kRows, kRows, kRows, 0.0, kRows / 2, &plan))
        return hipErrorInvalidValue;
    PTables<K> tab;
    double wsum = 0.0;
    for (int j = 0; j < K; j++) {
        tab.w1[j] = coef.h_w1d[j];
        wsum += (double)coef.h_w1d[j];
    }
    for (int j = 0; j < K * K; j++)
        tab.w2[j] = h_w2d[j];
    tab.delta = (float)delta_bound<K>(tab.w1, tab.w2);
    const bool clamp = !(255.0 * wsum * wsum * 1.0001 < 256.0);
    const bool ragged = (w & 3) != 0 || (reinterpret_cast<uintptr_t>(d_in) & 15u) != 0 ||
                        (reinterpret_cast<uintptr_t>(d_out) & 3u) != 0;
    const dim3 grid(plan.nblocks_a + plan.nblocks_b), block(kWavesPerBlock * 64);
#define MI355_LAUNCH(CL, RG)                                                                                  \
    hipLaunchKernelGGL((pipe_slide_kernel<R, CL, RG>), grid, block, 0, stream, d_in, d_out, w, h, sp.nstrips,  \
                       sp.lanes_out, plan, tab)
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

// The kernel needs a separable table whose factor is symmetric (the pair form) and a useful error bound; anything
// else (only reachable through mi355_ctx_set_gauss_weights) goes to the tiled kernel.
bool pipe_slide_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef)
{
    (void)d_out;
    const int k = coef.k;
    if (k != 3 && k != 5 && k != 7)
        return false;
    if (w < 4 || h < 2 || !coef.separable || !coef.h_w2d)
        return false;
    for (int j = 0; j < k / 2; j++)
        if (coef.h_w1d[j] != coef.h_w1d[k - 1 - j])
            return false;
    const double delta = (k == 3) ? delta_bound<3>(coef.h_w1d, coef.h_w2d)
                                  : (k == 5 ? delta_bound<5>(coef.h_w1d, coef.h_w2d) : delta_bound<7>(coef.h_w1d, coef.h_w2d));
    if (!(delta < 0.01))
        return false;
    return (reinterpret_cast<uintptr_t>(d_in) & 3u) == 0;
}

hipError_t launch_pipe_slide(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                             const GaussCoef& coef)
{
    switch (coef.k) {
    case 3: return launch_r<1>(stream, d_in, d_out, w, h, nframes, coef, coef.h_w2d);
    case 5: return launch_r<2>(stream, d_in, d_out, w, h, nframes, coef, coef.h_w2d);
    case 7: return launch_r<3>(stream, d_in, d_out, w, h, nframes, coef, coef.h_w2d);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mi355
