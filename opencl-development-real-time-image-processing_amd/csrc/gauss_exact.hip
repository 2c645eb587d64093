// gauss_exact.hip — the EXACT-mode Gaussian (bit-identical to the reference CPU path,
// src/GaussianBlur/GaussianBlur.cpp:234-261) at sliding-window speed, k in {3, 5, 7}.  gfx950 only.
// (k = 7: four channels x 49-tap chains x two walking directions per row is more code than LLVM's default budget for
// `#pragma unroll` — the loop stayed rolled and the ring went to scratch memory; the Makefile raises the budget for
// this file.)
//
// MI355_GAUSS_EXACT used to mean the LDS-tiled kernel evaluating the CPU path's own k*k-term chain for every value
// (2 k^2 operations per channel: 1.0 TB/s at k = 5 on 64 x 4K frames; this kernel: 4.7 TB/s, same box).  This kernel is "exact by exception" (exact_common.hpp, the
// idea of pipe_slide.hip applied to four channels): a separable fp32 evaluation S (2k operations) is within
// delta_bound() of the CPU sum, so trunc(S) is the CPU byte unless S sits within that bound of an integer; only
// those values (7.5e-4 of them at k = 5 on noise-like frames) are recomputed with the CPU path's chain, from the ring
// of input rows the wave keeps in registers anyway.  Flat regions take the chain for every value: still exact,
// slower there.
//
// Structure: one wave per (frame, band, strip of <= 62 lanes + 1 halo lane per side), a lane owns 4 pixels; the ring
// holds the last K rows as floats, per channel; vertical sums in the symmetric pair form (reads the same whichever way
// the band walks: odd bands walk upward, shared boundary rows hit L2), horizontal taps through DPP.
// Frames whose alpha is constant (255 everywhere: every frame after cvtColor(BGR2RGBA); round 3: any value A, named by the
// band's first pixel) run a 3-channel pass; the alpha of an all-A window is what the CPU chain gives for it (flat[A], the
// table the constant-window path uses anyway).  The pass tests every loaded row (halo lanes included) and at the first
// other alpha abandons — the rows stored so far are correct — and the band is redone with four channels, as in
// gauss_slide.hip.  (A constant alpha channel is flat content: in the 4-channel walk every one of its values took the
// constant-window search — alpha = 128 everywhere: 3.41 TB/s, now 4.73, the opaque rate.)
#include <type_traits>

#include "common.hpp"
#include "exact_common.hpp"
#include "kernels.hpp"
#include "slide_common.hpp"

namespace mi355 {

namespace {

constexpr int kWavesPerBlock = kSlideWavesPerBlock;

template <int K>
struct ETables {
    float w1[K];
    float w2[K * K];
    float delta;
    uint32_t alpha255;  // the CPU chain's byte for an all-255 window, already shifted to bits 31:24
};

template <int R, bool CLAMP>
__global__ __launch_bounds__(kWavesPerBlock * 64) void gauss_exact_kernel(const uint8_t* __restrict__ in,
                                                                         uint8_t* __restrict__ out, int w, int h,
                                                                         int nstrips, int lanes_out, BandPlan plan,
                                                                         ETables<2 * R + 1> tab)
{
    constexpr int K = 2 * R + 1;
    __shared__ float flat[256];  // flat[c] = the CPU path's chain over a window that is c everywhere (exact_common.hpp)
    flat[threadIdx.x] = flat_chain<K>((float)threadIdx.x, tab.w2);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    SlideItem it;
    if (!slide_item(plan, nstrips, h, &it))
        return;  // after the only barrier
    const int strip = it.strip, y0 = it.y0, nout = it.nout;
    const size_t frame = it.frame;
    const bool up = (it.band & 1) != 0;  // wave-uniform

    const int q_lane = strip * lanes_out + lane - 1;
    const int quads = w >> 2;  // w % 4 == 0 (gauss_exact_supported)
    const int q_load = clampi(q_lane, 0, min(quads - 1, (strip + 1) * lanes_out));  // idle lanes re-load the halo quad
    const bool left_of_image = q_lane < 0, right_of_image = q_lane >= quads;
    const bool edge_strip = (strip == 0) || (4 * (strip * lanes_out + 63) > w);  // wave-uniform
    const int q_end = min((strip + 1) * lanes_out, quads);
    const bool stores = (lane >= 1) && (q_lane < q_end);

    // output rows y0 .. y0+nout-1 need input rows y0-R .. y0+nout-1+R; arrival index i counts them in walking order
    const int nin = nout + 2 * R;
    const int y_first = up ? y0 + nout - 1 + R : y0 - R;
    const int y_step = up ? -1 : 1;

    const size_t row_bytes = (size_t)w * 4;
    const auto fin = uniform_ptr(in + frame * row_bytes * h);
    const auto fout = uniform_ptr(out + frame * row_bytes * h);
    uint32_t in_off = (uint32_t)q_load * 16u;
    uint32_t out_off = (uint32_t)(stores ? q_lane : 0) * 16u;

    float wv[R + 1];  // wv[d] = weight at distance d from the centre
#pragma unroll
    for (int d = 0; d <= R; d++)
        wv[d] = tab.w1[R - d];
    const float delta = tab.delta, two_delta = 2.0f * tab.delta;

    auto load_row = [&](int i) -> u32x4 {
        const int y = clampi(y_first + y_step * min(i, nin - 1), 0, h - 1);  // clamp-to-edge rows (GaussianBlur.cpp:241)
        const auto rowp = fin + (size_t)y * row_bytes;
        lane_offset_here(in_off);
        return gload<u32x4>(rowp + in_off);
    };

    // One walk over the band with NCH channels.  Returns false when a 3-channel walk met a non-opaque pixel.
    auto walk = [&](auto nch_tag) -> bool {
        constexpr int NCH = decltype(nch_tag)::value;
        constexpr int PF = 3;
        u32x4 q[K];
#pragma unroll
        for (int u = 0; u < PF; u++)
            q[u] = load_row(u);
        uint32_t n_alpha = 0u, alpha_hi = tab.alpha255;  // 3-channel walk: (A ^ 0xFF) << 24 and the byte of an all-A window
        float g[NCH][K][4];  // ring of the last K rows, per channel; slot = arrival index % K
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int s = 0; s < K; s++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    g[c][s][e] = 0.0f;

        for (int base = 0; base < nin; base += K) {
#pragma unroll
            for (int u = 0; u < K; u++) {
                const int i = base + u;
                u32x4 p = q[u];
                q[(u + PF) % K] = load_row(i + PF);
                if constexpr (NCH == 3) {
                    // The band's first pixel names the alpha value A the 3-channel walk bets on (255 for every frame that
                    // went through cvtColor; round 3: any constant).  Every loaded row, all 64 lanes: the first pixel with
                    // another alpha ends the walk BEFORE the first output row whose window contains it is computed.  The
                    // alpha of an all-A window is what the CPU chain gives for it: flat[A], clamped and truncated.
                    if (i == 0) {  // wave-uniform
                        const uint32_t a0 = ((uint32_t)__builtin_amdgcn_readfirstlane((int)p.x) >> 24) & 0xFFu;
                        n_alpha = (a0 ^ 0xFFu) << 24;
                        float ca = flat[a0];
                        ca = ca < 0.0f ? 0.0f : (ca > 255.0f ? 255.0f : ca);
                        alpha_hi = __builtin_amdgcn_readfirstlane((uint32_t)ca << 24);
                    }
                    const uint32_t a = ((p.x ^ n_alpha) & (p.y ^ n_alpha)) & ((p.z ^ n_alpha) & (p.w ^ n_alpha));
                    if (__builtin_amdgcn_ballot_w64(a < 0xFF000000u) != 0 && i < nin)
                        return false;
                }
                if (edge_strip) {
                    if (left_of_image)
                        p = u32x4{p.x, p.x, p.x, p.x};  // clamp-to-edge columns: replicate column 0
                    if (right_of_image)
                        p = u32x4{p.w, p.w, p.w, p.w};  // replicate column w-1
                }
#pragma unroll
                for (int c = 0; c < NCH; c++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        g[c][u][e] = (float)((p[e] >> (8 * c)) & 0xFFu);  // v_cvt_f32_ubyteN
                if (i >= 2 * R) {
                    uint32_t px[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int c = 0; c < NCH; c++) {
                        // window = arrival rows i-2R .. i = slots (u+1+t) % K; vertical pass, symmetric pair form
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            float acc = wv[0] * g[c][(u + 1 + R) % K][e];
#pragma unroll
                            for (int d = 1; d <= R; d++)
                                acc = __builtin_fmaf(wv[d], g[c][(u + 1 + R - d) % K][e] + g[c][(u + 1 + R + d) % K][e], acc);
                            v[e] = acc;
                        }
                        // horizontal pass; S' = S + delta rides on the centre tap (one-sided integer test)
                        float S[4], t[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            float acc = __builtin_fmaf(wv[0], v[e], delta);
#pragma unroll
                            for (int d = 1; d <= R; d++) {
                                const int a = e - d, b = e + d;
                                const float va = (a < 0) ? dppl(v[4 + a]) : v[a];
                                const float vb = (b > 3) ? dppr(v[b - 4]) : v[b];
                                acc = __builtin_fmaf(wv[d], va + vb, acc);
                            }
                            S[e] = acc;
                            t[e] = __builtin_amdgcn_fractf(acc);
                        }
                        const float tmin = fminf(fminf(t[0], t[1]), fminf(t[2], t[3]));
                        const uint64_t flagged = __builtin_amdgcn_ballot_w64(tmin < two_delta);
                        if (__builtin_expect(flagged != 0, 0)) {
                            // (k = 7 is left out: with the ring of 4 x 7 rows the extra live values push the kernel
                            // from 198 VGPRs into scratch memory)
                            if (R <= 2 && dense_flags(flagged)) {  // flat content: constant windows take a table read
                                if (!stores) {           // halo and idle lanes store nothing: no exceptions on their behalf
#pragma unroll
                                    for (int J = 0; J < 4; J++)
                                        t[J] = 1.0f;
                                }
                                flat_windows<K, 4>(g[c], S, t, two_delta, flat);
                            }
#define MI355_EXACT_PX(J)                                                         \
    if (__builtin_amdgcn_ballot_w64(t[J] < two_delta) != 0) {                       \
        if (up)                                                                   \
            S[J] = exact_sum<K, J, true>(g[c], u, tab.w2);                        \
        else                                                                      \
            S[J] = exact_sum<K, J, false>(g[c], u, tab.w2);                       \
    }
                            MI355_EXACT_PX(0)
                            MI355_EXACT_PX(1)
                            MI355_EXACT_PX(2)
                            MI355_EXACT_PX(3)
#undef MI355_EXACT_PX
                        }
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            float sum = S[e];
                            if constexpr (CLAMP)
                                sum = fminf(sum, 255.0f);
                            px[e] |= (uint32_t)sum << (8 * c);  // uchar(clamp(sum, 0, 255)): truncation
                        }
                    }
                    if constexpr (NCH == 3) {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            px[e] |= alpha_hi;
                    }
                    const int cidx = i - 2 * R;  // completed output row in arrival order
                    const int m = up ? y0 + nout - 1 - cidx : y0 + cidx;
                    if (stores && cidx < nout) {
                        const auto rowp = fout + (size_t)m * row_bytes;
                        lane_offset_here(out_off);
                        gstore_nt<u32x4>(rowp + out_off, u32x4{px[0], px[1], px[2], px[3]});
                    }
                }
            }
        }
        return true;
    };
    if (!walk(std::integral_constant<int, 3>{}))
        walk(std::integral_constant<int, 4>{});
}

template <int R>
hipError_t launch_r(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                    const GaussCoef& coef)
{
    constexpr int K = 2 * R + 1;
    const StripPlan sp = make_strip_plan(w);
    BandPlan plan;
    constexpr int kRows = (R == 1) ? 16 : (R == 2 ? 24 : 40);
    constexpr int kWavesPerSimd = (R == 1) ? 5 : (R == 2 ? 3 : 2);
    if (!make_band_plan(h, sp.nstrips, nframes, kWavesPerSimd, kRows, kRows, kRows, 0.0, kRows / 2, &plan))
        return hipErrorInvalidValue;
    ETables<K> tab;
    double wsum = 0.0;
    for (int j = 0; j < K; j++) {
        tab.w1[j] = coef.h_w1d[j];
        wsum += (double)coef.h_w1d[j];
    }
    for (int j = 0; j < K * K; j++)
        tab.w2[j] = coef.h_w2d[j];
    tab.delta = (float)delta_bound<K>(tab.w1, tab.w2);
    // the CPU path's value for an all-255 window: its own chain (this translation unit is built with
    // -ffp-contract=off: one float multiply and one float add per tap), clamped and truncated
    float chain = 0.0f;
    for (int j = 0; j < K * K; j++)
        chain += 255.0f * coef.h_w2d[j];
    chain = chain < 0.0f ? 0.0f : (chain > 255.0f ? 255.0f : chain);
    tab.alpha255 = (uint32_t)chain << 24;
    const bool clamp = !(255.0 * wsum * wsum * 1.0001 + 0.01 < 256.0);
    const dim3 grid(plan.nblocks_a + plan.nblocks_b), block(kWavesPerBlock * 64);
    if (clamp)
        hipLaunchKernelGGL((gauss_exact_kernel<R, true>), grid, block, 0, stream, d_in, d_out, w, h, sp.nstrips, sp.lanes_out,
                           plan, tab);
    else
        hipLaunchKernelGGL((gauss_exact_kernel<R, false>), grid, block, 0, stream, d_in, d_out, w, h, sp.nstrips, sp.lanes_out,
                           plan, tab);
    return hipGetLastError();
}

}  // namespace

// k in {3, 5, 7}, width a multiple of 4, 16-byte aligned buffers, a separable table with a symmetric factor and a
// useful error bound (everything mi355_gauss_weights generates qualifies)
bool gauss_exact_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef)
{
    (void)h;
    const int k = coef.k;
    if ((k != 3 && k != 5 && k != 7) || !coef.separable || !coef.h_w2d)
        return false;
    if ((w & 3) != 0 || ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15u) != 0)
        return false;
    for (int j = 0; j < k / 2; j++)
        if (coef.h_w1d[j] != coef.h_w1d[k - 1 - j])
            return false;
    const double delta = (k == 3) ? delta_bound<3>(coef.h_w1d, coef.h_w2d)
                                  : (k == 5 ? delta_bound<5>(coef.h_w1d, coef.h_w2d) : delta_bound<7>(coef.h_w1d, coef.h_w2d));
    return delta < 0.01;
}

hipError_t launch_gauss_exact(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
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
