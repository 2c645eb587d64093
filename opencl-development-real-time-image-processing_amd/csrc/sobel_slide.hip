// sobel_slide.hip — Sobel edge magnitude of RGBA8 frames, register-resident sliding window, one
// wavefront per image strip (the shape of gauss_slide.hip).  gfx950 only; any width (RAGGED
// instantiation when width % 4 != 0 or the pointers are not 16-byte aligned).
//
// Replaces kernel `sobel_edge_detection` (RT/kernel/edge_base.cl:1-57) + ConvertToUChar
// (RT/src/Controller.cpp:76-85,605) with the semantics of the reference CPU path
// (src/EdgeDetection/EdgeDetection.cpp:219-240, BORDER_REFLECT_101, round + saturate) on
// gray = src/Grayscale/grayscale.cpp:237 of every pixel.  Bit-exact with the oracle.
//
//  * a lane owns 4 consecutive pixels (one global_load_dwordx4), a wave a strip of up to 62 lanes plus one
//    halo lane per side, walking down a band of rows; each input row is loaded once, 3 rows in flight.
//  * luminance once per pixel (integer fast path, FP64 only for the 0.1 % ambiguous colours), kept for
//    three rows in registers; per row the column sums t+2m+b and differences b-t, then
//    gx = cs[x+1]-cs[x-1], gy = cd[x-1]+2cd[x]+cd[x+1]; the x-1 / x+1 columns of the edge pixels come from
//    the neighbouring lane through DPP (wave_shr:1 / wave_shl:1) — no LDS, no barrier.
//  * reflect-101: rows by reflecting the wave-uniform row index; columns by loading the mirror pixel into
//    the halo lane (x = -1 takes x = 1, x = w takes x = w-2), only in the two edge strips.
//  * out = min(255, round(sqrt(gx^2+gy^2))) through the fix-up-free form in common.hpp; 4 results packed
//    into one dword store per lane.
// Algorithmic bytes: 5 B/px (4 read, 1 written).  Bound: HBM.
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"
#include "slide_common.hpp"

namespace mi355 {

namespace {

constexpr int kWavesPerBlock = kSlideWavesPerBlock;

__device__ __forceinline__ float dpp_left(float v)  // lane l <- lane l-1
{
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}

__device__ __forceinline__ float dpp_right(float v)  // lane l <- lane l+1
{
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

// RAGGED = width % 4 != 0 or unaligned buffers: interior strips use unaligned 16-byte row accesses, the two
// edge strips address their pixels one by one through reflect-101 (which also covers the partial last quad),
// and the output row (1 byte per pixel, width bytes long) is written byte by byte in the edge strips.
// LOCKSTEP (aligned instantiation, mid-size launches): one s_barrier per row keeps the four waves of a workgroup —
// adjacent strips, or the end of one row-band and the start of the next — to the same row, so their 1-KiB loads and
// 240-byte stores reach the memory system together.  Same box and buffers, 60-lane strips: 4K x 10 / 16 / 20 / 32 frames
// +3 / +7.5 / +8.7 / +7.2 %, 1080p x 48-128 +4 to +7 %, 720p, 1440p, 1600 x 1200, 640 x 480 (3-11 strips per row: no multiple
// of four needed here) +5 to +7 %; launches of 1-4 frames lose 5-7 % with it and ragged widths 0-5 %, so those keep
// LOCKSTEP = false; from 2^28 pixels the aligned-strip kernel below takes over (profiles/r03_sobel_lockstep_ab.txt).
// A wave that runs out of rows simply ends: s_barrier waits for the waves of the group that have not terminated.
template <bool RAGGED, bool LOCKSTEP>
__global__ __launch_bounds__(kWavesPerBlock * 64) void sobel_slide_kernel(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int w, int h, int nstrips, int lanes_out,
    BandPlan plan)
{
    constexpr int K = 3;
    const int quads = (w + 3) >> 2;
    const int lane = threadIdx.x & 63;
    SlideItem it;
    __shared__ uint8_t gray_lut[256];  // luma(v, v, v): the ambiguous case of gray pixels without FP64 (common.hpp)
    fill_gray_lut(gray_lut);
    __syncthreads();
    if (!slide_item(plan, nstrips, h, &it))
        return;  // after the only barrier
    const int strip = it.strip, y0 = it.y0, nout = it.nout;
    const size_t frame = it.frame;

    const int q_lane = strip * lanes_out + lane - 1;
    // lanes right of the strip's right halo lane are never read by a storing lane: they re-load the halo quad
    // (same address = same cache line) instead of the next strip's data
    const int q_load = clampi(q_lane, 0, min(quads - 1, (strip + 1) * lanes_out));
    const bool left_of_image = q_lane < 0, right_of_image = q_lane >= quads;
    const bool edge_strip = (strip == 0) || (4 * (strip * lanes_out + 63) > w);  // wave-uniform
    const int q_end = min((strip + 1) * lanes_out, quads);
    const bool stores = (lane >= 1) && (q_lane < q_end);
    const int x_lane = 4 * q_lane;

    const int nin = nout + 2;

    const size_t row_bytes = (size_t)w * 4;
    const auto fin = uniform_ptr(in + frame * row_bytes * h);
    const auto fout = uniform_ptr(out + frame * (size_t)w * h);  // one byte per pixel
    uint32_t in_off = (uint32_t)q_load * 16u;
    uint32_t out_off = (uint32_t)(stores ? q_lane : 0) * 4u;
    uint32_t px_off[4];  // RAGGED edge strips: BORDER_REFLECT_101 of each pixel column (x <= w is all that is read)
#pragma unroll
    for (int j = 0; j < 4; j++)
        px_off[j] = (uint32_t)reflect101(clampi(x_lane + j, -1, w), w) * 4u;

    // Odd bands walk UP (the stencil is symmetric under a vertical flip: gy only changes sign, and it is squared;
    // all arithmetic is exact, so the bits do not change).  A down-walking band and the up-walking band below it
    // then read their two shared boundary rows at the same moment — the end of both walks — and the second reader
    // hits L2 instead of HBM; likewise the up-walking band and the down-walking one below it at their start.
    const bool up = (it.band & 1) != 0;
    auto load_row = [&](int i) -> u32x4 {
        const int ii = min(i, nin - 1);
        const int y = reflect101(up ? y0 + nout - ii : y0 - 1 + ii, h);
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
    bool gray_run = false;  // wave-uniform hint for the luminance (common.hpp: gray_row)
    u32x4 q[K];
#pragma unroll
    for (int u = 0; u < PF; u++)
        q[u] = load_row(u);

    // luminance of the last three rows (integer-valued floats: the whole stencil runs in fp32, where
    // these small integers are exact and add/fma are the cheapest VALU ops), slot = input row index % 3
    float L[K][4] = {};

    for (int base = 0; base < nin; base += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            const int i = base + u;
            u32x4 p = q[u];
            if constexpr (LOCKSTEP)
                __builtin_amdgcn_s_barrier();
            q[(u + PF) % K] = load_row(i + PF);
            if constexpr (!RAGGED) {
                if (edge_strip) {
                    // BORDER_REFLECT_101 columns: the only halo pixel ever read is the one next to the image
                    if (left_of_image)
                        p.w = p.y;  // x = -1  <-  x = 1   (lane holds pixels 0..3)
                    if (right_of_image)
                        p.x = p.z;  // x = w   <-  x = w-2 (lane holds pixels w-4..w-1)
                }
            }
            luma_quad_fast(p, L[u], gray_lut, gray_run);

            // rows i-2 (top), i-1 (middle), i (bottom) -> output row m = i - 2
            const float* t = L[(u + 1) % K];
            const float* md = L[(u + 2) % K];
            const float* b = L[u];
            float cs[4], cd[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cs[j] = __builtin_fmaf(2.0f, md[j], t[j]) + b[j];
                cd[j] = b[j] - t[j];
            }
            const float csl = dpp_left(cs[3]), csr = dpp_right(cs[0]);
            const float cdl = dpp_left(cd[3]), cdr = dpp_right(cd[0]);
            const float gx0 = cs[1] - csl, gx1 = cs[2] - cs[0], gx2 = cs[3] - cs[1], gx3 = csr - cs[2];
            const float gy0 = __builtin_fmaf(2.0f, cd[0], cdl) + cd[1];
            const float gy1 = __builtin_fmaf(2.0f, cd[1], cd[0]) + cd[2];
            const float gy2 = __builtin_fmaf(2.0f, cd[2], cd[1]) + cd[3];
            const float gy3 = __builtin_fmaf(2.0f, cd[3], cd[2]) + cdr;
            const float gxs[4] = {gx0, gx1, gx2, gx3}, gys[4] = {gy0, gy1, gy2, gy3};
            const uint32_t r = sobel_mag_quad(gxs, gys);
            const int m = i - 2;
            if (stores && m >= 0 && m < nout) {
                const auto rowp = fout + (size_t)(up ? y0 + nout - 1 - m : y0 + m) * w;
                lane_offset_here(out_off);
                if constexpr (RAGGED) {
                    if (edge_strip && x_lane + 3 >= w) {  // the last quad of a row may be partial
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (x_lane + j < w)
                                rowp[out_off + j] = (uint8_t)(r >> (8 * j));
                    } else {
                        gstore_a1<uint32_t>(rowp + out_off, r);  // rows of w bytes: any byte alignment
                    }
                } else {
                    gstore_nt<uint32_t>(rowp + out_off, r);
                }
            }
        }
    }
}


// lane l <- lane l-1 / l+1, and the lane that has no such neighbour (0 / 63) takes `edge` instead (a DPP move
// without bound_ctrl leaves the destination, pre-loaded with `edge`, untouched there)
__device__ __forceinline__ float dpp_left_or(float v, float edge)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}

__device__ __forceinline__ float dpp_right_or(float v, float edge)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}

// Luminance of ONE wave-uniform pixel on the scalar unit (the two halo pixels of the strip kernel below): S, the quotient
// S / 1000 and the exception test of luma_quad_int (common.hpp) in s_bfe / s_mul / s_mul_hi — the VALU, which this kernel
// keeps 74 % busy, sees none of it.  The exception (S a multiple of 1000: 0.1 % of colours, every gray one) takes the
// table / FP64 form under a wave-uniform branch.  px must be wave-uniform (v_readfirstlane'd by the caller).
__device__ __forceinline__ uint32_t luma_px_uniform(uint32_t px, const uint8_t* gray_lut)
{
    const uint32_t r = px & 0xFFu, g = (px >> 8) & 0xFFu, b = (px >> 16) & 0xFFu;
    const uint32_t S = 299u * r + 587u * g + 114u * b;  // <= 255000
    constexpr uint32_t M = 4294968u;                    // ceil(2^32 / 1000)
    uint32_t q = (uint32_t)(((uint64_t)S * M) >> 32);
    const uint32_t low = S * M;                         // low half: < 1,000,000 exactly when S % 1000 == 0
    if (low < 1000000u)                                 // wave-uniform
        q = __builtin_amdgcn_readfirstlane(luma_px_ambiguous(px, gray_lut));
    return q;
}

// The aligned shape (width % 4 == 0, 16-byte-aligned input, 4-byte-aligned output): a wave owns exactly 64 pixel
// quads — all 64 lanes produce output, the row accesses are whole aligned 1-KiB loads and 256-byte stores like
// gray.hip's strip kernel — and the one pixel it needs on either side of its strip is fetched separately: a
// wave-uniform address (reflect-101 applied to it at the image border, so the arithmetic has no border cases),
// i.e. a scalar load.  sobel_slide_kernel above spends a halo LANE per side (62 output lanes, every access shifted
// by 16 bytes, 240-byte store spans) and remains the kernel for everything else.
__global__ __launch_bounds__(kWavesPerBlock * 64) void sobel_strip_kernel(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int quads, int h, int nstrips, BandPlan plan)
{
    constexpr int K = 3;
    const int lane = threadIdx.x & 63;
    SlideItem it;
    __shared__ uint8_t gray_lut[256];  // luma(v, v, v): the ambiguous case of gray pixels without FP64 (common.hpp)
    fill_gray_lut(gray_lut);
    __syncthreads();
    if (!slide_item(plan, nstrips, h, &it))
        return;  // after the only barrier
    const int w = 4 * quads, y0 = it.y0, nout = it.nout, nin = nout + 2;
    const int q = it.strip * 64 + lane;
    const int q_last = min((it.strip + 1) * 64, quads) - 1;  // last quad of this strip (wave-uniform)
    const bool active = q <= q_last;
    const bool partial = q_last - it.strip * 64 < 63;  // wave-uniform: only the last strip of a row can be partial
    const int x_left = reflect101(4 * it.strip * 64 - 1, w);   // x = -1 -> 1
    const int x_right = reflect101(4 * (q_last + 1), w);        // x = w  -> w-2

    const size_t row_bytes = (size_t)w * 4;
    const auto fin = uniform_ptr(in + it.frame * row_bytes * h);
    const auto fout = uniform_ptr(out + it.frame * (size_t)w * h);
    uint32_t in_off = (uint32_t)min(q, q_last) * 16u;
    uint32_t out_off = (uint32_t)min(q, q_last) * 4u;
    const bool up = (it.band & 1) != 0;  // see sobel_slide_kernel

    struct Row {
        u32x4 p;
        uint32_t hl, hr;
    };
    auto load_row = [&](int i) -> Row {
        const int ii = min(i, nin - 1);
        const int y = reflect101(up ? y0 + nout - ii : y0 - 1 + ii, h);
        const auto rowp = fin + (size_t)y * row_bytes;
        lane_offset_here(in_off);
        Row r;
        r.p = gload<u32x4>(rowp + in_off);
        r.hl = gload<uint32_t>(rowp + (uint32_t)x_left * 4u);   // uniform addresses
        r.hr = gload<uint32_t>(rowp + (uint32_t)x_right * 4u);
        return r;
    };

    constexpr int PF = 3;
    bool gray_run = false;  // wave-uniform hint for the luminance (common.hpp: gray_row)
    Row qr[K];
#pragma unroll
    for (int u = 0; u < PF; u++)
        qr[u] = load_row(u);

    float L[K][4] = {};
    // luminance of the two halo pixels, same ring — wave-uniform integers in SGPRs: the halo columns' share of the
    // stencil (cs = 2 m + t + b, cd = b - t) is scalar arithmetic too, and only its four results are converted for the
    // lanes at the strip's ends (round 2 computed both luminances and both column sums on the VALU in every lane: ~24
    // of 125 instructions per wave-row)
    int HL[K] = {}, HR[K] = {};

    for (int base = 0; base < nin; base += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            const int i = base + u;
            const Row r = qr[u];
            qr[(u + PF) % K] = load_row(i + PF);
            luma_quad_fast(r.p, L[u], gray_lut, gray_run);
            HL[u] = (int)luma_px_uniform((uint32_t)__builtin_amdgcn_readfirstlane((int)r.hl), gray_lut);
            HR[u] = (int)luma_px_uniform((uint32_t)__builtin_amdgcn_readfirstlane((int)r.hr), gray_lut);

            const int st = (u + 1) % K, sm = (u + 2) % K, sb = u;  // oldest, middle, newest row
            float cs[4], cd[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cs[j] = __builtin_fmaf(2.0f, L[sm][j], L[st][j]) + L[sb][j];
                cd[j] = L[sb][j] - L[st][j];
            }
            const float csL = (float)(2 * HL[sm] + HL[st] + HL[sb]), cdL = (float)(HL[sb] - HL[st]);
            const float csR = (float)(2 * HR[sm] + HR[st] + HR[sb]), cdR = (float)(HR[sb] - HR[st]);
            const float csl = dpp_left_or(cs[3], csL), cdl = dpp_left_or(cd[3], cdL);
            float csr = dpp_right_or(cs[0], csR), cdr = dpp_right_or(cd[0], cdR);
            if (partial) {  // the strip ends inside the wave: its last lane takes the halo column too
                if (q == q_last) {
                    csr = csR;
                    cdr = cdR;
                }
            }
            const float gxs[4] = {cs[1] - csl, cs[2] - cs[0], cs[3] - cs[1], csr - cs[2]};
            const float gys[4] = {__builtin_fmaf(2.0f, cd[0], cdl) + cd[1], __builtin_fmaf(2.0f, cd[1], cd[0]) + cd[2],
                                  __builtin_fmaf(2.0f, cd[2], cd[1]) + cd[3], __builtin_fmaf(2.0f, cd[3], cd[2]) + cdr};
            const uint32_t res = sobel_mag_quad(gxs, gys);
            const int m = i - 2;
            if (active && m >= 0 && m < nout) {
                const auto rowp = fout + (size_t)(up ? y0 + nout - 1 - m : y0 + m) * w;
                lane_offset_here(out_off);
                gstore_nt<uint32_t>(rowp + out_off, res);
            }
        }
    }
}

}  // namespace

bool sobel_slide_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h)
{
    (void)h;
    (void)w;
    return (reinterpret_cast<uintptr_t>(d_in) & 3u) == 0;  // pixels are dwords; everything else is handled
}

hipError_t launch_sobel_slide(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes)
{
    static const int kStripMode = [] {
        // tuning sweeps and tests only: 0 = never use the aligned-strip kernel, 2 = whenever the rows allow it
        const char* e = tune_env("MI355_TUNE_SOBEL_STRIP");
        return e ? atoi(e) : 1;
    }();
    const bool kStripOff = kStripMode == 0;
    // big batches of aligned rows: the aligned-strip kernel (256 x 4K frames, same box: 5.81-5.99 TB/s against
    // 5.67-5.72 for the halo-lane kernel; 640x512 x 8000: 5.48 / 5.24; but 8 x 4K frames: 5.45 / 5.86)
    const bool big = (size_t)w * h * nframes >= ((size_t)1 << 28) || kStripMode == 2;
    if (!kStripOff && big && (w & 3) == 0 && (reinterpret_cast<uintptr_t>(d_in) & 15u) == 0 &&
        (reinterpret_cast<uintptr_t>(d_out) & 3u) == 0) {
        const int quads = w / 4, nstrips = (quads + 63) / 64;
        BandPlan plan;
        if (!make_band_plan(h, nstrips, nframes, 8, 16, 16, 16, 0.0, 8, &plan))
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(sobel_strip_kernel, dim3(plan.nblocks_a + plan.nblocks_b), dim3(kWavesPerBlock * 64), 0,
                           stream, d_in, d_out, quads, h, nstrips, plan);
        return hipGetLastError();
    }
    // big batches: 48-lane strips (192-byte store spans, see slide_common.hpp: +5 % on 256 x 4K frames); small
    // ones keep the fewest, widest strips (8 x 4K frames: 48 lanes would cost 8 %)
    const StripPlan sp = make_strip_plan(w, big ? 48 : 0);
    // 50 VGPRs -> 8 waves/SIMD, and only 2 warm-up rows per band (loads that hit L2): this light kernel
    // wants many short work items — measured at steady clocks on 256 x 4K frames: 16-row bands 5.2 TB/s,
    // 32 rows 5.0, 64 rows 4.5, 128 rows 4.2
    BandPlan plan;
    if (!make_band_plan(h, sp.nstrips, nframes, 8, 16, 16, 16, 0.0, 8, &plan))
        return hipErrorInvalidValue;
    const bool ragged = (w & 3) != 0 || (reinterpret_cast<uintptr_t>(d_in) & 15u) != 0 ||
                        (reinterpret_cast<uintptr_t>(d_out) & 3u) != 0;
    // mid-size launches of aligned rows (8 x 10^7 pixels ~ ten 4K frames, up to where the strip kernel takes over): rows in
    // lock-step, see the kernel
    const bool lockstep = !ragged && !big && (size_t)w * h * nframes >= 80000000ull;
    const dim3 grid(plan.nblocks_a + plan.nblocks_b), block(kWavesPerBlock * 64);
    if (ragged)
        hipLaunchKernelGGL((sobel_slide_kernel<true, false>), grid, block, 0, stream, d_in, d_out, w, h, sp.nstrips,
                           sp.lanes_out, plan);
    else if (lockstep)
        hipLaunchKernelGGL((sobel_slide_kernel<false, true>), grid, block, 0, stream, d_in, d_out, w, h, sp.nstrips,
                           sp.lanes_out, plan);
    else
        hipLaunchKernelGGL((sobel_slide_kernel<false, false>), grid, block, 0, stream, d_in, d_out, w, h, sp.nstrips,
                           sp.lanes_out, plan);
    return hipGetLastError();
}

}  // namespace mi355
