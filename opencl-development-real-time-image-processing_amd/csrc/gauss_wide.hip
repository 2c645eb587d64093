// gauss_wide.hip — separable Gaussian blur for the wider kernels k = 11, 13, 15, 17 (the reference
// ProgramHandler's default is k = 17, sigma = 6: include/ProgramHandler.hpp:9), register-resident sliding
// window.  Same semantics, same canonical FAST arithmetic (bit-identical to gauss_tile.hip) as gauss_slide.hip;
// different shape, because K accumulator rows of 4 pixels would need 16 K VGPRs:
//  * a lane owns 2 consecutive pixels (one 8-byte load); K running accumulators x 8 floats per lane;
//  * a wave owns a strip of 64 - 2H lanes plus H = ceil(R/2) halo lanes per side;
//  * horizontal taps reach up to R pixels = H lanes away, beyond one DPP shift: the finished vertical sums of
//    a row go through a wave-private 2-KiB LDS row (two ds_write_b128 per lane, K+1 ds_read_b128 back); LDS
//    operations of one wave execute in order, so no barrier is involved.
// The inner loop is unrolled K times (static accumulator slots): ~2 KB of code per row, 34 KB at k = 17.
// Bound: FP32 VALU (2K FMA per channel-pixel); algorithmic bytes 8 B/px.
#include <cmath>

#include "common.hpp"
#include "kernels.hpp"
#include "slide_common.hpp"

namespace mi355 {

namespace {

constexpr int kWavesPerBlock = kSlideWavesPerBlock;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

template <int K>
struct WWeights {
    float w[K];
};

struct WideLane {
    const uint8_t* fin;
    uint8_t* fout;
    f32x4* vr;  // this wave's 128-entry LDS row
    size_t row_bytes;
    uint32_t in_off, out_off;
    int y0, nout, nin, h, lane;
    bool left_of_image, right_of_image, edge_strip, stores;
};

// One pass over the band with NCH channels per pixel: 4 = general; 3 = opaque fast path (see gauss_slide.hip):
// alpha not computed, constant output byte alpha_hi, returns false at the first loaded row with an alpha != 255.
template <int R, int NCH>
__device__ __forceinline__ bool gauss_wide_band(const WideLane& L, const float (&wv)[2 * R + 1], uint32_t alpha_hi)
{
    constexpr int K = 2 * R + 1;
    auto load_row = [&](int i) -> u32x2 {
        const int y = clampi(L.y0 - R + min(i, L.nin - 1), 0, L.h - 1);
        return *reinterpret_cast<const u32x2*>(L.fin + (size_t)y * L.row_bytes + L.in_off);
    };
    constexpr int PF = 3;
    u32x2 q[K];
#pragma unroll
    for (int u = 0; u < PF; u++)
        q[u] = load_row(u);

    float acc[K][2 * NCH] = {};
    // window base for the horizontal pass: output pixel e of this lane reads pixels 2*lane + e - R + t
    const int win0 = 2 * L.lane - R;

    for (int base = 0; base < L.nin; base += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            const int i = base + u;  // rows of a last partial trip run with their store masked off
            u32x2 p = q[u];
            q[(u + PF) % K] = load_row(i + PF);
            if (L.edge_strip) {
                if (L.left_of_image)
                    p = u32x2{p.x, p.x};  // clamp-to-edge columns: replicate pixel 0
                if (L.right_of_image)
                    p = u32x2{p.y, p.y};  // replicate pixel w-1
            }
            if constexpr (NCH == 3) {
                if (__builtin_amdgcn_ballot_w64(((p.x & p.y) >> 24) != 0xFFu) != 0)  // wave-uniform
                    return false;
            }
#pragma unroll
            for (int px = 0; px < 2; px++) {
                float f[NCH];
#pragma unroll
                for (int c = 0; c < NCH; c++)
                    f[c] = (float)((p[px] >> (8 * c)) & 0xFFu);
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const int s = (u - j + K) % K;
#pragma unroll
                    for (int c = 0; c < NCH; c++)
                        acc[s][px * NCH + c] =
                            (j == 0) ? wv[0] * f[c] : __builtin_fmaf(wv[j], f[c], acc[s][px * NCH + c]);
                }
            }
            const int m = i - 2 * R;
            if (m >= 0) {  // wave-uniform: the first 2R rows of a band finish no output row
                const float* v = acc[(u + 1) % K];
                L.vr[2 * L.lane] = f32x4{v[0], v[1], v[2], NCH == 4 ? v[NCH - 1] : 0.0f};
                L.vr[2 * L.lane + 1] = f32x4{v[NCH], v[NCH + 1], v[NCH + 2], NCH == 4 ? v[2 * NCH - 1] : 0.0f};
                // horizontal pass: K+1 window values feed the two outputs (tap t of output e is value t+e)
                float o0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, o1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t <= K; t++) {
                    const f32x4 a = L.vr[clampi(win0 + t, 0, 127)];
#pragma unroll
                    for (int c = 0; c < NCH; c++) {
                        if (t < K)
                            o0[c] = (t == 0) ? wv[0] * a[c] : __builtin_fmaf(wv[t], a[c], o0[c]);
                        if (t >= 1)
                            o1[c] = (t == 1) ? wv[0] * a[c] : __builtin_fmaf(wv[t - 1], a[c], o1[c]);
                    }
                }
                if (L.stores && m < L.nout) {
                    u32x2 r;
                    r.x = f2u8(o0[0]) | (f2u8(o0[1]) << 8) | (f2u8(o0[2]) << 16);
                    r.y = f2u8(o1[0]) | (f2u8(o1[1]) << 8) | (f2u8(o1[2]) << 16);
                    if constexpr (NCH == 4) {
                        r.x |= f2u8(o0[3]) << 24;
                        r.y |= f2u8(o1[3]) << 24;
                    } else {
                        r.x |= alpha_hi;
                        r.y |= alpha_hi;
                    }
                    __builtin_nontemporal_store(
                        r, reinterpret_cast<u32x2*>(L.fout + (size_t)(L.y0 + m) * L.row_bytes + L.out_off));
                }
            }
        }
    }
    return true;
}

template <int R>
__global__ __launch_bounds__(kWavesPerBlock * 64) void gauss_wide_kernel(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int pairs /* w/2 */, int h, int nstrips,
    int lanes_out, BandPlan plan, WWeights<2 * R + 1> wts, uint32_t alpha_hi)
{
    constexpr int K = 2 * R + 1;
    constexpr int H = (R + 1) / 2;  // halo lanes per side (2 px each)
    __shared__ f32x4 vrow[kWavesPerBlock][128];
    SlideItem it;
    if (!slide_item(plan, nstrips, h, &it))
        return;
    WideLane L;
    L.vr = vrow[threadIdx.x >> 6];
    L.lane = threadIdx.x & 63;
    const int strip = it.strip;
    const int q_lane = strip * lanes_out + L.lane - H;  // this lane's pixel-pair column
    const int q_load = clampi(q_lane, 0, pairs - 1);
    const int q_end = min((strip + 1) * lanes_out, pairs);
    L.left_of_image = q_lane < 0;
    L.right_of_image = q_lane >= pairs;
    L.edge_strip = (strip == 0) || (strip * lanes_out + 64 - H > pairs);  // wave-uniform
    L.stores = (L.lane >= H) && (q_lane < q_end);
    L.y0 = it.y0;
    L.nout = it.nout;
    L.nin = it.nout + 2 * R;
    L.h = h;
    L.row_bytes = (size_t)pairs * 8;
    L.fin = in + it.frame * L.row_bytes * h;
    L.fout = out + it.frame * L.row_bytes * h;
    L.in_off = (uint32_t)q_load * 8u;
    L.out_off = (uint32_t)(L.stores ? q_lane : 0) * 8u;

    float wv[K];
#pragma unroll
    for (int j = 0; j < K; j++)
        wv[j] = wts.w[j];

    // opaque fast path first, full redo of the band if any alpha != 255 shows up (gauss_slide.hip)
    if (!gauss_wide_band<R, 3>(L, wv, alpha_hi))
        gauss_wide_band<R, 4>(L, wv, alpha_hi);
}

template <int R>
hipError_t launch_r(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                    const GaussCoef& coef)
{
    constexpr int K = 2 * R + 1;
    constexpr int H = (R + 1) / 2;
    const int pairs = w / 2;
    const int lanes_max = 64 - 2 * H;
    const int nstrips = (pairs + lanes_max - 1) / lanes_max;
    const int lanes_out = (pairs + nstrips - 1) / nstrips;
    BandPlan plan;
    if (!make_band_plan(h, nstrips, nframes, 2, 128, 360, 48, 0.1, &plan))
        return hipErrorInvalidValue;
    WWeights<K> wts;
    for (int j = 0; j < K; j++)
        wts.w[j] = coef.h_w1d[j];
    // constant alpha byte of the opaque fast path: the canonical chains on an all-255 channel, in float
    float vc = wts.w[0] * 255.0f;
    for (int t = 1; t < K; t++)
        vc = std::fmaf(wts.w[t], 255.0f, vc);
    float hc = wts.w[0] * vc;
    for (int t = 1; t < K; t++)
        hc = std::fmaf(wts.w[t], vc, hc);
    hc = hc < 0.0f ? 0.0f : (hc > 255.0f ? 255.0f : hc);
    const uint32_t alpha_hi = (uint32_t)hc << 24;
    hipLaunchKernelGGL(gauss_wide_kernel<R>, dim3(plan.nblocks_a + plan.nblocks_b), dim3(kWavesPerBlock * 64), 0,
                       stream, d_in, d_out, pairs, h, nstrips, lanes_out, plan, wts, alpha_hi);
    return hipGetLastError();
}

}  // namespace

bool gauss_wide_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, int k)
{
    (void)h;
    if (k != 11 && k != 13 && k != 15 && k != 17)
        return false;
    if ((w & 1) != 0)
        return false;
    return ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 7u) == 0;
}

hipError_t launch_gauss_wide(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                             const GaussCoef& coef)
{
    switch (coef.k) {
    case 11: return launch_r<5>(stream, d_in, d_out, w, h, nframes, coef);
    case 13: return launch_r<6>(stream, d_in, d_out, w, h, nframes, coef);
    case 15: return launch_r<7>(stream, d_in, d_out, w, h, nframes, coef);
    case 17: return launch_r<8>(stream, d_in, d_out, w, h, nframes, coef);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mi355
