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

template <int R>
__global__ __launch_bounds__(kWavesPerBlock * 64) void gauss_wide_kernel(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int pairs /* w/2 */, int h, int nstrips,
    int lanes_out, BandPlan plan, WWeights<2 * R + 1> wts)
{
    constexpr int K = 2 * R + 1;
    constexpr int H = (R + 1) / 2;  // halo lanes per side (2 px each)
    __shared__ f32x4 vrow[kWavesPerBlock][128];
    f32x4* vr = vrow[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    SlideItem it;
    if (!slide_item(plan, nstrips, h, &it))
        return;
    const int strip = it.strip, y0 = it.y0, nout = it.nout;
    const size_t frame = it.frame;

    const int q_lane = strip * lanes_out + lane - H;  // this lane's pixel-pair column
    const int q_load = clampi(q_lane, 0, pairs - 1);
    const bool left_of_image = q_lane < 0, right_of_image = q_lane >= pairs;
    const bool edge_strip = (strip == 0) || (strip * lanes_out + 64 - H > pairs);  // wave-uniform
    const int q_end = min((strip + 1) * lanes_out, pairs);
    const bool stores = (lane >= H) && (q_lane < q_end);
    const int nin = nout + 2 * R;

    const size_t row_bytes = (size_t)pairs * 8;
    const uint8_t* fin = in + frame * row_bytes * h;
    uint8_t* fout = out + frame * row_bytes * h;
    const uint32_t in_off = (uint32_t)q_load * 8u;
    const uint32_t out_off = (uint32_t)(stores ? q_lane : 0) * 8u;

    float wv[K];
#pragma unroll
    for (int j = 0; j < K; j++)
        wv[j] = wts.w[j];

    auto load_row = [&](int i) -> u32x2 {
        const int y = clampi(y0 - R + min(i, nin - 1), 0, h - 1);
        return *reinterpret_cast<const u32x2*>(fin + (size_t)y * row_bytes + in_off);
    };

    constexpr int PF = 3;
    u32x2 q[K];
#pragma unroll
    for (int u = 0; u < PF; u++)
        q[u] = load_row(u);

    float acc[K][8] = {};
    // window base for the horizontal pass: output pixel e of this lane reads pixels 2*lane + e - R + t
    const int win0 = 2 * lane - R;

    for (int base = 0; base < nin; base += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            const int i = base + u;  // rows of a last partial trip run with their store masked off
            u32x2 p = q[u];
            q[(u + PF) % K] = load_row(i + PF);
            if (edge_strip) {
                if (left_of_image)
                    p = u32x2{p.x, p.x};  // clamp-to-edge columns: replicate pixel 0
                if (right_of_image)
                    p = u32x2{p.y, p.y};  // replicate pixel w-1
            }
#pragma unroll
            for (int px = 0; px < 2; px++) {
                float f[4];
#pragma unroll
                for (int c = 0; c < 4; c++)
                    f[c] = (float)((p[px] >> (8 * c)) & 0xFFu);
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const int s = (u - j + K) % K;
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        acc[s][px * 4 + c] =
                            (j == 0) ? wv[0] * f[c] : __builtin_fmaf(wv[j], f[c], acc[s][px * 4 + c]);
                }
            }
            const int m = i - 2 * R;
            if (m >= 0) {  // wave-uniform: the first 2R rows of a band finish no output row
                const float* v = acc[(u + 1) % K];
                vr[2 * lane] = f32x4{v[0], v[1], v[2], v[3]};
                vr[2 * lane + 1] = f32x4{v[4], v[5], v[6], v[7]};
                // horizontal pass: K+1 window values feed the two outputs (tap t of output e is value t+e)
                f32x4 o0, o1;
#pragma unroll
                for (int t = 0; t <= K; t++) {
                    const f32x4 a = vr[clampi(win0 + t, 0, 127)];
                    if (t < K) {
                        if (t == 0) {
                            o0 = a * wv[0];
                        } else {
                            o0.x = __builtin_fmaf(wv[t], a.x, o0.x);
                            o0.y = __builtin_fmaf(wv[t], a.y, o0.y);
                            o0.z = __builtin_fmaf(wv[t], a.z, o0.z);
                            o0.w = __builtin_fmaf(wv[t], a.w, o0.w);
                        }
                    }
                    if (t >= 1) {
                        if (t == 1) {
                            o1 = a * wv[0];
                        } else {
                            o1.x = __builtin_fmaf(wv[t - 1], a.x, o1.x);
                            o1.y = __builtin_fmaf(wv[t - 1], a.y, o1.y);
                            o1.z = __builtin_fmaf(wv[t - 1], a.z, o1.z);
                            o1.w = __builtin_fmaf(wv[t - 1], a.w, o1.w);
                        }
                    }
                }
                if (stores && m < nout) {
                    u32x2 r;
                    r.x = f2u8(o0.x) | (f2u8(o0.y) << 8) | (f2u8(o0.z) << 16) | (f2u8(o0.w) << 24);
                    r.y = f2u8(o1.x) | (f2u8(o1.y) << 8) | (f2u8(o1.z) << 16) | (f2u8(o1.w) << 24);
                    __builtin_nontemporal_store(
                        r, reinterpret_cast<u32x2*>(fout + (size_t)(y0 + m) * row_bytes + out_off));
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
    hipLaunchKernelGGL(gauss_wide_kernel<R>, dim3(plan.nblocks_a + plan.nblocks_b), dim3(kWavesPerBlock * 64), 0,
                       stream, d_in, d_out, pairs, h, nstrips, lanes_out, plan, wts);
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
