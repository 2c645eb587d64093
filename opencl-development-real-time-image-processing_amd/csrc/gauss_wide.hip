// gauss_wide.hip — separable Gaussian blur for the wider kernels k = 11, 13, 15, 17 (the reference
// ProgramHandler's default is k = 17, sigma = 6: include/ProgramHandler.hpp:9), register-resident sliding
// window.  Same semantics, same canonical FAST arithmetic (bit-identical to gauss_tile.hip) as gauss_slide.hip;
// different shape, because K accumulator rows of 4 pixels would need 16 K VGPRs:
//  * a lane owns 2 consecutive pixels (one 8-byte load); K running accumulators x 8 floats per lane;
//  * a wave owns a strip of 64 - 2H lanes plus H = ceil(R/2) halo lanes per side;
//  * horizontal taps reach up to R pixels = H lanes away, beyond one DPP shift: the finished vertical sums of
//    a row go through a wave-private LDS row (two ds_write_b128 per lane, K+1 ds_read_b128 back at immediate
//    offsets from one per-lane base; the row is padded by R entries per side so that the halo lanes' windows
//    need no clamping — they read padding, and halo lanes never store); LDS operations of one wave execute in
//    order, so no barrier is involved.
//  * TWO kernels, because registers are allocated per kernel: the opaque pass (3 channels, see gauss_slide.hip)
//    needs 6 K accumulators, the general one 8 K — at k = 17 that is the difference between 2 waves per SIMD
//    and 1.  Kernel A runs the opaque pass and leaves one flag per work item (0 = band done, 1 = an alpha != 255
//    turned up); kernel B, launched right behind it on the same stream, redoes the flagged bands with 4 channels
//    and exits at once everywhere else.
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
    f32x4* vr;  // this wave's LDS row: entry R + p holds pixel p of the strip (p = 0..127), R padding entries per
                // side; even entries first, then odd entries (see gauss_wide_band)
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

    // Ring of the last K input rows as packed halves (a byte value is exact in fp16): {pixel 0, pixel 1} of one
    // channel per VGPR, K x NCH registers instead of the K x 2 x NCH fp32 accumulators of the first version —
    // at k = 17 that is 51 instead of 102: 139 VGPRs instead of 223, 3 waves per SIMD instead of 2.  The vertical sums of an
    // output row are formed when its window is complete, by v_fma_mix_f32 (fp32 multiply-add with an fp16 source,
    // converted exactly): the same canonical chain, the same bits.  v_fma_mix issues at ~3 cycles instead of 2
    // (tools/probe_valu.hip), which the doubled occupancy more than pays for.
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 ring[K][NCH];
    // horizontal pass: output pixel e of this lane reads strip pixels 2*lane + e - R + t = row entries
    // 2*lane + e + t, t = 0..K-1: one base pointer per lane, immediate offsets per tap
    // The row is stored as two planes, even and odd entries (entry e lives in plane e & 1 at index e >> 1), so
    // that the lanes of one ds_read_b128 touch consecutive 16-byte slots: with the entries of a lane's pixel
    // pair side by side, lanes l and l + 8 met in the same banks (SQ_LDS_BANK_CONFLICT = half the LDS cycles).
    constexpr int kPlane = (128 + 2 * R + 2) / 2;
    const f32x4* win = L.vr + L.lane;  // entry 2*lane + t  ->  plane t & 1, index lane + (t >> 1)

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
            for (int c = 0; c < NCH; c++)
                ring[u][c] = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz((float)((p.x >> (8 * c)) & 0xFFu),
                                                                           (float)((p.y >> (8 * c)) & 0xFFu)));
            const int m = i - 2 * R;
            if (m >= 0) {  // wave-uniform: the first 2R rows of a band finish no output row
                // canonical vertical chain, top tap first: the oldest row of the ring is slot u + 1.  (The first
                // step is written as fma(w0, x, +0) = w0 * x so that it too takes the fp16 source directly.)
                float v[2 * NCH];
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    float a0 = __builtin_fmaf(wv[0], (float)ring[(u + 1) % K][c].x, 0.0f);
                    float a1 = __builtin_fmaf(wv[0], (float)ring[(u + 1) % K][c].y, 0.0f);
#pragma unroll
                    for (int j = 1; j < K; j++) {
                        a0 = __builtin_fmaf(wv[j], (float)ring[(u + 1 + j) % K][c].x, a0);
                        a1 = __builtin_fmaf(wv[j], (float)ring[(u + 1 + j) % K][c].y, a1);
                    }
                    v[c] = a0;
                    v[NCH + c] = a1;
                }
                // this lane's pixels are entries R + 2*lane and R + 2*lane + 1
                L.vr[(R & 1) * kPlane + (R >> 1) + L.lane] = f32x4{v[0], v[1], v[2], NCH == 4 ? v[NCH - 1] : 0.0f};
                L.vr[((R + 1) & 1) * kPlane + ((R + 1) >> 1) + L.lane] =
                    f32x4{v[NCH], v[NCH + 1], v[NCH + 2], NCH == 4 ? v[2 * NCH - 1] : 0.0f};
                // The other 63 lanes read these entries below.  The hardware executes a wave's LDS operations in
                // order, but the compiler knows nothing of lanes: without a release/acquire pair at wavefront
                // scope it may treat the stores as thread-private (it deleted 20 of the 22 in one build of this
                // kernel, forwarding the values to this lane's own two reads).  The fences emit no instruction.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // horizontal pass: K+1 window values feed the two outputs (tap t of output e is value t+e)
                float o0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, o1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t <= K; t++) {
                    // entries R and R+1 of the window are this lane's own two pixels: no LDS read for them
                    const f32x4 a = (t == R)       ? f32x4{v[0], v[1], v[2], NCH == 4 ? v[NCH - 1] : 0.0f}
                                    : (t == R + 1) ? f32x4{v[NCH], v[NCH + 1], v[NCH + 2], NCH == 4 ? v[2 * NCH - 1] : 0.0f}
                                                   : win[(t & 1) * kPlane + (t >> 1)];
#pragma unroll
                    for (int c = 0; c < NCH; c++) {
                        if (t < K)
                            o0[c] = (t == 0) ? wv[0] * a[c] : __builtin_fmaf(wv[t], a[c], o0[c]);
                        if (t >= 1)
                            o1[c] = (t == 1) ? wv[0] * a[c] : __builtin_fmaf(wv[t - 1], a[c], o1[c]);
                    }
                }
                // (and the next row's stores must stay behind this row's reads)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (L.stores && m < L.nout) {
                    // uchar(clamp(v, 0, 255)) = floor, then v_cvt_pk_u8_f32 (exact on integers, saturates both ways,
                    // inserts the byte): 2 ops per channel instead of max + min + cvt + shift + or
                    u32x2 r;
                    r.x = (NCH == 4) ? 0u : alpha_hi;
                    r.y = r.x;
#pragma unroll
                    for (int c = 0; c < NCH; c++) {
                        r.x = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(o0[c]), (uint32_t)c, r.x);
                        r.y = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(o1[c]), (uint32_t)c, r.y);
                    }
                    __builtin_nontemporal_store(
                        r, reinterpret_cast<u32x2*>(L.fout + (size_t)(L.y0 + m) * L.row_bytes + L.out_off));
                }
            }
        }
    }
    return true;
}

// NCH = 3: kernel A (opaque pass, writes flags[work]); NCH = 4: kernel B (redoes the bands kernel A flagged)
template <int R, int NCH>
__global__ __launch_bounds__(kWavesPerBlock * 64, (NCH == 3 && R <= 6) ? 3 : 2) void gauss_wide_kernel(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int pairs /* w/2 */, int h, int nstrips,
    int lanes_out, BandPlan plan, WWeights<2 * R + 1> wts, uint32_t alpha_hi, uint32_t* __restrict__ flags)
{
    constexpr int K = 2 * R + 1;
    constexpr int H = (R + 1) / 2;  // halo lanes per side (2 px each)
    __shared__ f32x4 vrow[kWavesPerBlock][128 + 2 * R + 2];  // two planes of (128 + 2R + 2) / 2 entries
    SlideItem it;
    if (!slide_item(plan, nstrips, h, &it))
        return;
    if constexpr (NCH == 4) {
        if (flags[it.work] == 0)  // wave-uniform: kernel A finished this band
            return;
    }
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

    const bool done = gauss_wide_band<R, NCH>(L, wv, alpha_hi);
    if constexpr (NCH == 3) {
        if (L.lane == 0)
            flags[it.work] = done ? 0u : 1u;
    }
}

template <int R>
bool wide_plan(int w, int h, int nframes, int& pairs, int& nstrips, int& lanes_out, BandPlan* plan)
{
    constexpr int H = (R + 1) / 2;
    pairs = w / 2;
    const int lanes_max = 64 - 2 * H;
    nstrips = (pairs + lanes_max - 1) / lanes_max;
    lanes_out = (pairs + nstrips - 1) / nstrips;
    return make_band_plan(h, nstrips, nframes, 2, 128, 360, 48, 0.1, 4 * R, plan);
}

template <int R>
hipError_t launch_r(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                    const GaussCoef& coef, uint32_t* d_flags)
{
    constexpr int K = 2 * R + 1;
    int pairs, nstrips, lanes_out;
    BandPlan plan;
    if (!d_flags || !wide_plan<R>(w, h, nframes, pairs, nstrips, lanes_out, &plan))
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
    const dim3 grid(plan.nblocks_a + plan.nblocks_b), block(kWavesPerBlock * 64);
    hipLaunchKernelGGL((gauss_wide_kernel<R, 3>), grid, block, 0, stream, d_in, d_out, pairs, h, nstrips, lanes_out,
                       plan, wts, alpha_hi, d_flags);
    hipLaunchKernelGGL((gauss_wide_kernel<R, 4>), grid, block, 0, stream, d_in, d_out, pairs, h, nstrips, lanes_out,
                       plan, wts, alpha_hi, d_flags);
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

size_t gauss_wide_flag_items(int w, int h, int nframes, int k)
{
    int pairs, nstrips, lanes_out;
    BandPlan plan;
    bool ok = false;
    switch (k) {
    case 11: ok = wide_plan<5>(w, h, nframes, pairs, nstrips, lanes_out, &plan); break;
    case 13: ok = wide_plan<6>(w, h, nframes, pairs, nstrips, lanes_out, &plan); break;
    case 15: ok = wide_plan<7>(w, h, nframes, pairs, nstrips, lanes_out, &plan); break;
    case 17: ok = wide_plan<8>(w, h, nframes, pairs, nstrips, lanes_out, &plan); break;
    default: break;
    }
    return ok ? (size_t)plan.nwork_a + plan.nwork_b : 0;
}

hipError_t launch_gauss_wide(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                             const GaussCoef& coef, uint32_t* d_flags)
{
    switch (coef.k) {
    case 11: return launch_r<5>(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    case 13: return launch_r<6>(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    case 15: return launch_r<7>(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    case 17: return launch_r<8>(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mi355
