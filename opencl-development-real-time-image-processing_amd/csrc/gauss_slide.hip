// gauss_slide.hip — the headline kernel: separable Gaussian blur of RGBA8 frames for k = 3, 5, 7, 9,
// register-resident sliding window, one wavefront per image strip.  gfx950 only.
//
// Replaces kernel `gaussian_blur` (RT/kernel/gaussian_base.cl:1-50), semantics of the reference CPU
// path src/GaussianBlur/GaussianBlur.cpp:234-261 (clamp-to-edge, 4 channels, truncation), FAST
// arithmetic (within 1 LSB per channel; canonical op order shared with gauss_tile.hip, so the two
// kernels agree bit for bit).
//
// Mapping (DESIGN.md "Gaussian, sliding window"):
//  * a lane owns 4 consecutive pixels (one 16-byte global_load_dwordx4 / global_store_dwordx4);
//    a wave owns a vertical strip of up to 62 such lanes plus one halo lane on each side, and walks
//    down a band of rows.  Every input row of the band is loaded exactly once by exactly one
//    coalesced 1-KiB wave access; 3 rows are kept in flight per wave (prefetch ring in VGPRs).
//  * vertical pass first, in registers: K running accumulators per lane (one per pending output
//    row), each new row is converted once (v_cvt_f32_ubyteN) and folded into all K of them;
//    the accumulator that just received its last tap is the finished vertical sum `v`.
//  * horizontal pass on `v`: taps that fall in the neighbouring lane are read through DPP
//    (wave_shr:1 / wave_shl:1) — no LDS, no barrier, no shuffle instruction.
//  * clamp-to-edge: rows by clamping the (wave-uniform) row index; columns by replicating the edge
//    pixel into the halo lane at load time, so the arithmetic itself has no border cases.
//  * constant alpha (every A = 255 is what cv::cvtColor produces; mattes and overlays are piecewise constant): a
//    band is first run on 3 channels; while the K rows of a window carry ONE alpha value A over the whole strip, the
//    blurred alpha is the constant byte alpha_tab[A] (the canonical chains on an all-A window, evaluated by the
//    host).  A row with mixed alphas, or a window that spans two values, aborts the pass and the band is redone on
//    4 channels, so the output never depends on which pass produced it.
//  * RAGGED instantiation: any width and any 4-byte-aligned pointer (unaligned 16-byte interior loads,
//    per-pixel clamped loads and predicated narrow stores in the two edge strips).
// Algorithmic bytes: 8 B/px.  Extra traffic: halo lanes (2/62 of the loads, L2/MALL hits) and 2R
// warm-up rows per band.  VALU: 2*K FMA per channel + 1 cvt in + 1 cvt/pack out.  Bound: HBM.
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"
#include "slide_common.hpp"

namespace mi355 {

namespace {

constexpr int kWavesPerBlock = kSlideWavesPerBlock;

template <int K>
struct Weights {
    float w[K];
};

__device__ __forceinline__ float dpp_from_left(float v)
{
    // lane l <- lane l-1 (lane 0 gets 0: it is a halo lane, its result is never stored)
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}

__device__ __forceinline__ float dpp_from_right(float v)
{
    // lane l <- lane l+1 (lane 63 gets 0: never an output lane)
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

__device__ __forceinline__ float ubyte_f32(uint32_t p, int c)
{
    return (float)((p >> (8 * c)) & 0xFFu);  // v_cvt_f32_ubyte{c}
}

// float -> u8 as uchar(std::clamp(v, 0.f, 255.f)).  The sums here are >= 0 and NaN-free (non-negative
// weights times u8), so only the upper clamp can matter, and it cannot when 255 * (sum of weights)^2
// < 256: the launcher selects CLAMP = false for such tables (every normalised Gaussian).
// Truncate 4 pixels x 4 channels of floats in [0, 256) to bytes and merge each pixel into one dword in
// 16 instructions: v_cvt_u32_f32 for byte 0, then SDWA conversions that write byte 1..3 of the same
// register and preserve the rest.  (Measured on gfx950: every non-FMA VALU op costs ~3 cycles at 4
// waves/SIMD, so the compiler's cvt + shift/or sequence, 7 ops per pixel, is worth replacing.)
// gfx940-class "dst_sel forwarding" hazard: a VALU op that reads a VGPR right after a sub-dword (SDWA)
// write of it needs a wait state, and hipcc does not look inside an asm statement — so the four pixels
// are interleaved (three instructions between two partial writes of one register) and the block ends
// with an s_nop before anything else can read the results.  A back-to-back version loses a byte.
__device__ __forceinline__ void cvt_pack4x4(const float (&h)[4][4] /* [channel][pixel] */, u32x4& o)
{
    uint32_t o0, o1, o2, o3;
    asm("v_cvt_u32_f32_e32 %0, %4\n\t"
        "v_cvt_u32_f32_e32 %1, %5\n\t"
        "v_cvt_u32_f32_e32 %2, %6\n\t"
        "v_cvt_u32_f32_e32 %3, %7\n\t"
        "v_cvt_u32_f32_sdwa %0, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %1, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %2, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %3, %11 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %0, %12 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %1, %13 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %2, %14 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %3, %15 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %0, %16 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %1, %17 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %2, %18 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %3, %19 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "s_nop 1"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
        : "v"(h[0][0]), "v"(h[0][1]), "v"(h[0][2]), "v"(h[0][3]), "v"(h[1][0]), "v"(h[1][1]), "v"(h[1][2]),
          "v"(h[1][3]), "v"(h[2][0]), "v"(h[2][1]), "v"(h[2][2]), "v"(h[2][3]), "v"(h[3][0]), "v"(h[3][1]),
          "v"(h[3][2]), "v"(h[3][3]));
    o = u32x4{o0, o1, o2, o3};
}

// Same for the opaque fast path: three channels converted, the alpha byte is a wave-uniform constant
// (already shifted to bits 31:24).  The v_or that merges it reads its register three instructions after the
// last partial write of it.
__device__ __forceinline__ void cvt_pack3x4(const float (&h)[4][4] /* [channel][pixel] */, uint32_t alpha_hi,
                                            u32x4& o)
{
    uint32_t o0, o1, o2, o3;
    asm("v_cvt_u32_f32_e32 %0, %4\n\t"
        "v_cvt_u32_f32_e32 %1, %5\n\t"
        "v_cvt_u32_f32_e32 %2, %6\n\t"
        "v_cvt_u32_f32_e32 %3, %7\n\t"
        "v_cvt_u32_f32_sdwa %0, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %1, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %2, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %3, %11 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %0, %12 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %1, %13 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %2, %14 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_u32_f32_sdwa %3, %15 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_or_b32_e32 %0, %16, %0\n\t"
        "v_or_b32_e32 %1, %16, %1\n\t"
        "v_or_b32_e32 %2, %16, %2\n\t"
        "v_or_b32_e32 %3, %16, %3"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
        : "v"(h[0][0]), "v"(h[0][1]), "v"(h[0][2]), "v"(h[0][3]), "v"(h[1][0]), "v"(h[1][1]), "v"(h[1][2]),
          "v"(h[1][3]), "v"(h[2][0]), "v"(h[2][1]), "v"(h[2][2]), "v"(h[2][3]), "s"(alpha_hi));
    o = u32x4{o0, o1, o2, o3};
}

template <bool CLAMP>
__device__ __forceinline__ uint32_t to_u8(float a)
{
    if constexpr (CLAMP)
        a = fminf(a, 255.0f);
    return (uint32_t)a;
}

template <bool CLAMP>
__device__ __forceinline__ uint32_t pack_px(float a, float b, float c, float d)
{
    if constexpr (CLAMP) {
        a = fminf(a, 255.0f);
        b = fminf(b, 255.0f);
        c = fminf(c, 255.0f);
        d = fminf(d, 255.0f);
    }
    return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
}

// Horizontal 5-tap pass of ONE channel of a lane's 4 pixels, taps from the neighbouring lanes fused
// into the multiply-adds as DPP operands (v_mul_f32_dpp / v_fmac_f32_dpp, wave_shr:1 = value of lane-1,
// wave_shl:1 = value of lane+1).  hipcc fuses the DPP move into v_mul but not into v_fmac, and the
// separate v_mov_b32_dpp cost 16 VALU + 16 VGPRs per row, hence this block.  Same op order as the generic
// code path (o = w0*s0; o = fma(w_t, s_t, o)), so results are bit-identical.
//  * `s_nop 1` first: a VGPR written by VALU needs 2 wait states before a DPP read of it, and the
//    hazard recogniser does not look inside an asm statement.  Inside the block no DPP source is written.
//    (Round 3 tried opening the block with its two plain multiplies instead of the s_nop — same bits, the hazard
//    distance kept: no change on opaque frames, 3 % SLOWER on the 4-channel pass, same box; put back.)
//  * volatile: the block must execute with the full EXEC mask of the wave (a DPP read of a disabled lane
//    returns 0), so it must not be sunk into the store-predicated region.
__device__ __forceinline__ void hpass5_dpp(float v0, float v1, float v2, float v3, float w0, float w1, float w2,
                                           float w3, float w4, float& o0, float& o1, float& o2, float& o3)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_mul_f32_dpp %0, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %1, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_e32 %2, %8, %4\n\t"
        "v_mul_f32_e32 %3, %8, %5\n\t"
        "v_fmac_f32_dpp %0, %7, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_e32 %1, %9, %4\n\t"
        "v_fmac_f32_e32 %2, %9, %5\n\t"
        "v_fmac_f32_e32 %3, %9, %6\n\t"
        "v_fmac_f32_e32 %0, %10, %4\n\t"
        "v_fmac_f32_e32 %1, %10, %5\n\t"
        "v_fmac_f32_e32 %2, %10, %6\n\t"
        "v_fmac_f32_e32 %3, %10, %7\n\t"
        "v_fmac_f32_e32 %0, %11, %5\n\t"
        "v_fmac_f32_e32 %1, %11, %6\n\t"
        "v_fmac_f32_e32 %2, %11, %7\n\t"
        "v_fmac_f32_dpp %3, %4, %11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_e32 %0, %12, %6\n\t"
        "v_fmac_f32_e32 %1, %12, %7\n\t"
        "v_fmac_f32_dpp %2, %4, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %5, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4));
}

// What a wave needs to know about its work item, computed once in the kernel
struct SlideLane {
    global_ptr<const uint8_t> fin;  // wave-uniform frame bases, pinned in SGPR pairs (slide_common.hpp)
    global_ptr<uint8_t> fout;
    size_t row_bytes;
    uint32_t in_off, out_off;
    int y0, nout, nin, h;
    bool left_of_image, right_of_image, edge_strip, stores;
    // RAGGED variant (width % 4 != 0 or buffers not 16-byte aligned): lanes of an edge strip that overlap the row's right
    // end load the row's last four pixels and shift them into place (slide_common.hpp: RaggedEdge; six v_cndmask per row,
    // edge strips only, no divergent branch).  Images narrower than 4 pixels keep the per-pixel form (px_off).
    uint32_t px_off[4];   // w < 4: byte offsets of the 4 (clamped) pixels inside a row
    int x_lane, w;        // first pixel of the lane (may be < 0 or >= w), image width
    RaggedEdge edge;
};

// "Do this lane's four pixels all carry alpha A?" as one AND chain: with nA = (A ^ 0xFF) << 24 the top byte of
// (p ^ nA) is 0xFF exactly where the pixel's alpha is A, so the AND over the four pixels keeps 0xFF there iff all four
// are (two v_xor_b32 + two three-input v_bitop3_b32 + a compare).  For A = 255 (NA0: nA = 0 at compile time) it is the
// v_and_b32 + v_bitop3_b32 + compare of the round-1 test.
template <bool NA0>
__device__ __forceinline__ bool alpha_row_differs(const u32x4& p, uint32_t nA)
{
    uint32_t t;
    if constexpr (NA0)
        t = (p.x & p.y) & (p.z & p.w);
    else
        t = ((p.x ^ nA) & (p.y ^ nA)) & ((p.z ^ nA) & (p.w ^ nA));
    return __builtin_amdgcn_ballot_w64(t < 0xFF000000u) != 0;
}

// Return codes of a pass over a band (3-channel passes only stop early; the rows stored so far are correct)
constexpr int kBandDone = 0;          // every output row stored
constexpr int kBandAbortUniform = 1;  // AMODE 1: a row whose alpha is uniform over the strip but not 255 — worth AMODE 2
constexpr int kBandAbort = 2;         // a row with mixed alphas, or (AMODE 2) a window that spans two values: 4 channels
constexpr int kBandAbortUniformFirst = 3;  // kBandAbortUniform at the band's first row: q3[0 .. PF-1] still hold the first rows

// AMODE of a 3-channel pass: 1 = alpha 255 only — the hot loop of opaque frames, exactly round 2's instruction stream —;
// 2 = any constant, piecewise (value tracking, table loads, window check).  The kernel tries 1, then 2 only where 1
// stopped at a uniform row, then the 4-channel pass.  (With both in ONE loop — a wave-uniform branch per row between the
// two tests, a run counter — the opaque path ran 0.7-1.1 % slower than round 2's on boxes where it is not memory-bound.)

// One pass over the band with NCH channels computed per pixel.  NCH = 4: the general path.  NCH = 3: the
// constant-alpha fast path — alpha is not computed; while the last K rows carry one alpha value A over all 64 lanes
// (halo included) every output gets the constant byte alpha_tab[A] (alpha_row_differs: three or four instructions per
// row; the value changes through a scalar branch and one s_load).  A row with mixed alphas, or an output row whose window
// spans two values, ends the pass before that row has stored anything: the rows stored so far are correct, and
// the caller redoes the band with NCH = 4.
template <int R, bool CLAMP, int NCH, bool RAGGED, bool UP, int AMODE, bool LOCKSTEP>
__device__ __forceinline__ int gauss_slide_band(const SlideLane& L, const float (&wv)[2 * R + 1], uint32_t alpha_hi,
                                                const uint32_t* __restrict__ alpha_tab, u32x4 (&q3)[2 * R + 1],
                                                bool preloaded)
{
    constexpr int K = 2 * R + 1;
    static_assert((NCH == 4) == (AMODE == 0), "AMODE 1 / 2 belong to the 3-channel pass");
    uint32_t in_off = L.in_off, out_off = L.out_off;
    // AMODE 2: (A ^ 0xFF) << 24 for the alpha value A the rows seen last carry — alpha_hi = alpha_tab[A] — and how many
    // consecutive rows, the current one included, carry it.  The first row sets them (run = 0 forces the first test to
    // take the "value changes" branch only if the row is not 255; 255 is where alpha_hi starts).
    uint32_t cur_nA = 0u;
    int run = 0;
    auto load_row = [&](int i) -> u32x4 {
        // rows past the band's last input re-read that last row (an L1/L2 hit, never consumed); an UP band
        // walks from its bottom-most input row to its top-most one
        const int ii = min(i, L.nin - 1);
        const int y = clampi(UP ? L.y0 + L.nout - 1 + R - ii : L.y0 - R + ii, 0, L.h - 1);
        const auto rowp = L.fin + (size_t)y * L.row_bytes;  // SGPR pair; + 32-bit lane offset = saddr form
        if constexpr (R <= 2)  // (k = 9: the asm statements keep hipcc from unrolling the trip; k = 7: -4 %)
            lane_offset_here(in_off);
        if constexpr (RAGGED) {
            if (L.w < 4) {  // wave-uniform, tiny images only
                u32x4 r;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    r[j] = gload<uint32_t>(rowp + L.px_off[j]);
                return r;
            }
            // every lane: one 16-byte access inside the row (in_off is clamped to pixel w - 4); the row is only 4-byte aligned
            return gload_a4<u32x4>(rowp + in_off);
        } else {
            // plain (cached) load: the halo lanes' lines are read again by the neighbouring strip
            return gload<u32x4>(rowp + in_off);
        }
    };

    // prefetch ring: row i lives in slot i % K; its load is issued PF rows before it is consumed, so
    // only PF of the K slots are live at a time (PF x 1 KiB in flight per wave)
    constexpr int PF = (K < 3) ? K : 3;
    // The 3-channel passes share the kernel's ring q3: when the alpha = 255 pass stops at the band's FIRST row the
    // constant-alpha pass starts from the rows already in flight (AMODE 2, preloaded) instead of a second start-up per
    // band.  The 4-channel pass keeps a ring of its own (sharing it too cost 12 VGPRs).
    u32x4 q_own[K];
    u32x4 (&q)[K] = (AMODE == 0) ? q_own : q3;
    if (AMODE != 2 || !preloaded) {
#pragma unroll
        for (int u = 0; u < PF; u++)
            q[u] = load_row(u);
    }

    // k <= 7: ring of the last K input rows, converted to float once (row i lives in slot i % K, static after
    // unrolling); the vertical sums of an output row are formed when its window is complete — in either walking
    // direction.  k = 9: K running accumulators instead (input row i is tap j of output row i - j), the form all
    // sizes used first: same multiply-add chains, same count, but 10 VGPRs fewer here, which is the third wave
    // per SIMD (168 vs 174); it can only walk down.
    constexpr bool kRing = R <= 3;
    static_assert(kRing || !UP, "the accumulator form walks down only");
    float rows[K][4 * NCH] = {};

    // One trip = K input rows.  Input row i completes the window of output row m = i - 2R; the first 2R rows of
    // a band and the rows of a last partial trip only convert (scalar branch below).
    for (int base = 0; base < L.nin; base += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            const int i = base + u;
            // the row is used in place where its slot is not the one the next load fills (k >= 5: PF < K) — a private copy
            // cost two v_mov_b64 per row on every strip for the sake of the edge strips' replication
            u32x4 p_copy = q[u];
            u32x4& p = *((PF < K) ? &q[u] : &p_copy);
            if constexpr (LOCKSTEP) {
                // Adjacent strips of a row-band in step: the four waves' 1-KiB accesses to one row reach memory together
                // (+1.3 % on opaque 4K and 1080p batches, +1.2 % at k = 3, +1.5 % on 8- and 64-frame launches, same box
                // and buffers; nothing on the issue-bound 4-channel pass or at k >= 7).  A wave that leaves the loop (end
                // of band, a pass that stops) just stops arriving: s_barrier waits for the waves of the group that have
                // not terminated, and every wave either reaches another s_barrier of this loop — in whichever pass — or
                // ends, so nobody waits for ever.  A template parameter, not a flag: the scalar test of a flag in this
                // loop cost launches of one or two frames 1.4 %.
                __builtin_amdgcn_s_barrier();
            }
            q[(u + PF) % K] = load_row(i + PF);
            if (L.edge_strip) {  // wave-uniform
                if constexpr (!RAGGED) {
                    // halo lanes outside the image replicate the edge pixel (clamp-to-edge columns)
                    if (L.left_of_image)
                        p = u32x4{p.x, p.x, p.x, p.x};
                    if (L.right_of_image)
                        p = u32x4{p.w, p.w, p.w, p.w};
                } else if (L.w >= 4) {
                    if (L.left_of_image)
                        p = u32x4{p.x, p.x, p.x, p.x};
                    ragged_shift_clamp(p, L.edge);  // lanes past the row's end: pixel w - 1 replicated
                }
            }
            if constexpr (AMODE == 1) {
                if (alpha_row_differs<true>(p, 0u)) {  // wave-uniform: the pass ends here
                    const uint32_t a_row = ((uint32_t)__builtin_amdgcn_readfirstlane((int)p.x) >> 24) & 0xFFu;
                    if (alpha_row_differs<false>(p, (a_row ^ 0xFFu) << 24))
                        return kBandAbort;
                    return (i == 0 && PF < K) ? kBandAbortUniformFirst : kBandAbortUniform;  // (k = 3: slot 0 is refilled already)
                }
            }
            if constexpr (AMODE == 2) {
                if (alpha_row_differs<false>(p, cur_nA)) {  // wave-uniform, rare: the alpha value changes here
                    const uint32_t a_new = ((uint32_t)__builtin_amdgcn_readfirstlane((int)p.x) >> 24) & 0xFFu;  // (int -> int builtin)
                    const uint32_t nA = (a_new ^ 0xFFu) << 24;
                    if (alpha_row_differs<false>(p, nA))
                        return kBandAbort;
                    cur_nA = nA;
                    alpha_hi = __builtin_amdgcn_readfirstlane(alpha_tab[a_new]);
                    run = 0;
                }
                run++;
            }
            if constexpr (kRing) {
#pragma unroll
                for (int px = 0; px < 4; px++)
#pragma unroll
                    for (int c = 0; c < NCH; c++)
                        rows[u][px * NCH + c] = ubyte_f32(p[px], c);
            } else {
                // rows[] holds accumulators: convert one pixel's channels, fold them into all K of them
#pragma unroll
                for (int px = 0; px < 4; px++) {
                    float f[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; c++)
                        f[c] = ubyte_f32(p[px], c);
#pragma unroll
                    for (int j = 0; j < K; j++) {
                        const int sl = (u - j + K) % K;
#pragma unroll
                        for (int c = 0; c < NCH; c++)
                            rows[sl][px * NCH + c] =
                                (j == 0) ? wv[0] * f[c] : __builtin_fmaf(wv[j], f[c], rows[sl][px * NCH + c]);
                    }
                }
            }
            const int m = i - 2 * R;  // output row whose window this input row completes
            // warm-up rows (m < 0) and the rows of a last partial trip (m >= nout) produce no output: skip
            // the horizontal pass and the store with a scalar branch (m and nout live in SGPRs, so EXEC
            // stays full inside, which the DPP reads of the horizontal pass require)
            if (m >= 0 && m < L.nout) {
                if constexpr (AMODE == 2) {
                    if (run < K)  // the window spans two alpha values: its blurred alpha is not a constant
                        return kBandAbort;
                }
                // vertical pass in the canonical order, top tap first: a DOWN band holds the window oldest row
                // = top row (slot u+1 ... slot u), an UP band newest row = top row (slot u, u-1, ...) — a static
                // permutation of the same multiply-add chain, so both directions give the same bits
                auto vsum = [&](int e) -> float {
                    if constexpr (!kRing)
                        return rows[(u + 1) % K][e];  // the accumulator that just received its last tap
                    float a = wv[0] * rows[UP ? u : (u + 1) % K][e];
#pragma unroll
                    for (int j = 1; j < K; j++)
                        a = __builtin_fmaf(wv[j], rows[UP ? (u - j + K) % K : (u + 1 + j) % K][e], a);
                    return a;
                };
                u32x4 o;
                if constexpr (R == 2) {
                    float hres[4][4];  // [channel][pixel]
#pragma unroll
                    for (int c = 0; c < NCH; c++)
                        hpass5_dpp(vsum(0 * NCH + c), vsum(1 * NCH + c), vsum(2 * NCH + c), vsum(3 * NCH + c), wv[0],
                                   wv[1], wv[2], wv[3], wv[4], hres[c][0], hres[c][1], hres[c][2], hres[c][3]);
                    if constexpr (CLAMP) {
#pragma unroll
                        for (int px = 0; px < 4; px++)
                            o[px] = (NCH == 4) ? pack_px<true>(hres[0][px], hres[1][px], hres[2][px], hres[3][px])
                                               : (pack_px<true>(hres[0][px], hres[1][px], hres[2][px], 0.0f) | alpha_hi);
                    } else if constexpr (NCH == 4) {
                        cvt_pack4x4(hres, o);
                    } else {
                        cvt_pack3x4(hres, alpha_hi, o);
                    }
                } else {
                    float v[4 * NCH];
#pragma unroll
                    for (int e = 0; e < 4 * NCH; e++)
                        v[e] = vsum(e);
#pragma unroll
                    for (int px = 0; px < 4; px++) {
                        float r4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int c = 0; c < NCH; c++) {
                            float sum = 0.0f;
#pragma unroll
                            for (int t = 0; t < K; t++) {
                                const int s = px - R + t;
                                float src;
                                if (s < 0)
                                    src = dpp_from_left(v[(4 + s) * NCH + c]);
                                else if (s > 3)
                                    src = dpp_from_right(v[(s - 4) * NCH + c]);
                                else
                                    src = v[s * NCH + c];
                                sum = (t == 0) ? wv[0] * src : __builtin_fmaf(wv[t], src, sum);
                            }
                            r4[c] = sum;
                        }
                        o[px] = pack_px<CLAMP>(r4[0], r4[1], r4[2], r4[3]);
                        if constexpr (NCH == 3)
                            o[px] |= alpha_hi;
                    }
                }
                if (L.stores) {
                    const auto rowp = L.fout + (size_t)(UP ? L.y0 + L.nout - 1 - m : L.y0 + m) * L.row_bytes;
                    if constexpr (R <= 2)
                        lane_offset_here(out_off);
                    if constexpr (RAGGED) {
                        if (L.edge_strip && L.x_lane + 3 >= L.w) {  // the last quad of a row may be partial
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                if (L.x_lane + j < L.w)
                                    gstore_a4<uint32_t>(rowp + out_off + 4 * j, o[j]);
                        } else {
                            gstore_a4<u32x4>(rowp + out_off, o);
                        }
                    } else {
                        gstore_nt<u32x4>(rowp + out_off, o);
                    }
                }
            }
        }
    }
    return kBandDone;
}

// MODE 0: one kernel, opaque pass then (if an alpha != 255 turns up) the general pass — k = 3, 5, where both fit
// in 4 waves/SIMD.  MODE 3 / MODE 4: the two passes as two kernels launched back to back (k = 7, 9): registers
// are allocated per kernel, and the opaque pass alone needs 139 instead of 173 VGPRs at k = 7 (3 waves/SIMD
// instead of 2).  Kernel MODE 3 leaves flags[work] = 0 (band done) or 1; kernel MODE 4 redoes the flagged bands
// and exits at once everywhere else (gauss_wide.hip does the same).
template <int R, bool CLAMP, bool RAGGED, int MODE, bool LOCKSTEP>
__global__ __launch_bounds__(kWavesPerBlock * 64, (R == 3 && MODE == 3) ? 3 : 1) void gauss_slide_kernel(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int w, int h, int nstrips, int lanes_out,
    BandPlan plan, Weights<2 * R + 1> wts, uint32_t alpha_hi, const uint32_t* __restrict__ alpha_tab,
    uint32_t* __restrict__ flags)
{
    const int quads = (w + 3) >> 2;  // RAGGED: the last quad of a row may hold fewer than 4 pixels
    constexpr int K = 2 * R + 1;
    const int lane = threadIdx.x & 63;
    SlideItem it;
    if (!slide_item(plan, nstrips, h, &it))
        return;
    if constexpr (MODE == 4) {
        if (flags[it.work] == 0)  // wave-uniform
            return;
    }
    const int strip = it.strip;

    const int q_lane = strip * lanes_out + lane - 1;  // this lane's pixel-quad column
    // replicated at the image border; lanes right of the strip's right halo lane (never read by a storing lane)
    // re-load the halo quad instead of the next strip's data
    const int q_load = clampi(q_lane, 0, min(quads - 1, (strip + 1) * lanes_out));
    const int q_end = min((strip + 1) * lanes_out, quads);
    SlideLane L;
    L.left_of_image = q_lane < 0;
    L.right_of_image = q_lane >= quads;
    // edge strip = a wave that touches pixels outside [0, w): wave-uniform
    L.edge_strip = (strip == 0) || (4 * (strip * lanes_out + 63) > w);
    L.stores = (lane >= 1) && (q_lane < q_end);
    L.y0 = it.y0;
    L.nout = it.nout;
    L.nin = it.nout + 2 * R;
    L.h = h;
    L.w = w;
    L.x_lane = 4 * q_lane;
    L.row_bytes = (size_t)w * 4;
    L.fin = uniform_ptr(in + it.frame * L.row_bytes * h);  // uniform base; lanes add a 32-bit offset
    L.fout = uniform_ptr(out + it.frame * L.row_bytes * h);
    L.in_off = (uint32_t)q_load * 16u;
    L.out_off = (uint32_t)(L.stores ? q_lane : 0) * 16u;
    L.edge = RaggedEdge{false, false, false, false};
    if constexpr (RAGGED) {
        if (w >= 4)
            L.in_off = ragged_edge_setup(q_lane, q_load, w, &L.edge);
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        L.px_off[j] = (uint32_t)clampi(4 * q_lane + j, 0, w - 1) * 4u;

    // weights live in VGPRs: the DPP forms (v_mul_f32_dpp / v_fmac_f32_dpp) take no SGPR operand
    float wv[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
        wv[j] = wts.w[j];
        asm volatile("" : "+v"(wv[j]));
    }

    // Opaque fast path first (frames from cv::cvtColor(BGR2RGBA) have A = 255 everywhere,
    // RT/src/ProgramHandler.cpp:127): with every alpha tap 255 the blurred alpha is one constant byte, which
    // the host computes with the same float chain (std::fmaf) — 25 % of the arithmetic gone, same bits.
    // A band that meets any other alpha value is redone in full.
    // Odd bands walk UP, so that a band and its lower neighbour read their shared boundary rows (the 2R rows
    // each needs from the other) at the same moment and the second reader hits L2 (sobel_slide.hip does the
    // same; measured there: HBM reads -12 %).  Wave-uniform: two instantiations of the band code.
    const bool up = (it.band & 1) != 0;
    u32x4 q3[K];  // the 3-channel passes' ring of input rows
    auto run = [&](auto amode, bool preloaded) -> int {
        constexpr int AMODE = decltype(amode)::value;
        constexpr int NCH = AMODE == 0 ? 4 : 3;
        if constexpr (R <= 3) {  // (the k = 9 form walks down only)
            if (up)
                return gauss_slide_band<R, CLAMP, NCH, RAGGED, true, AMODE, LOCKSTEP>(L, wv, alpha_hi, alpha_tab, q3, preloaded);
        }
        return gauss_slide_band<R, CLAMP, NCH, RAGGED, false, AMODE, LOCKSTEP>(L, wv, alpha_hi, alpha_tab, q3, preloaded);
    };
    // 3 channels while alpha is 255; where that stops at a row of another UNIFORM alpha, 3 channels with the value
    // tracked (constant and piecewise-constant alpha); 4 channels for what is left.  A pass that stops hands the band to
    // the next one, which starts it over: the rows an earlier pass stored are rewritten with the same bytes.  Only the
    // hand-over from the alpha = 255 pass to the constant-alpha pass at the band's FIRST row keeps the rows in flight
    // (q3; +0.8 % on constant-alpha frames).  Sharing the ring with the 4-channel pass as well cost 12 VGPRs — the ragged
    // instantiation's fourth wave — and a kernel-level "look at the first row, then pick the pass" loop around single
    // call sites sent the register allocator to 191-224 VGPRs on the ragged instantiations; both removed.
    auto three_channels = [&]() -> int {
        int code = run(std::integral_constant<int, 1>{}, false);
        if (code == kBandAbortUniform || code == kBandAbortUniformFirst)
            code = run(std::integral_constant<int, 2>{}, code == kBandAbortUniformFirst);
        return code;
    };
    if constexpr (MODE == 0) {
        if (three_channels() != kBandDone)
            run(std::integral_constant<int, 0>{}, false);
    } else if constexpr (MODE == 3) {
        const int code = three_channels();
        if (lane == 0)
            flags[it.work] = code == kBandDone ? 0u : 1u;
    } else {
        run(std::integral_constant<int, 0>{}, false);
    }
}

template <int R>
bool slide_plan(int w, int h, int nframes, StripPlan* sp, BandPlan* plan)
{
    constexpr int K = 2 * R + 1;
    *sp = make_strip_plan(w);
    // 122 VGPRs at k = 5 -> 4 waves/SIMD; opaque pass 3 waves at k = 7, 2 at k = 9.
    // Band height, measured on 256 x 4K frames on the two kinds of MI355X box met ("slow" boxes
    // copy at 4.4 TB/s instead of 5.5 and prefer short bands by up to 10 %, "fast" ones hardly care):
    //   k = 3: 12 rows (with up/down walking, slow box: 12 rows 5.72 TB/s, 16 rows 5.58, 8 rows 5.69; before it
    //          16 rows 6.23 fast / 5.50 slow against 6.10 / 5.01 for adaptive tall bands)
    //   k = 5: 24 rows (6.10 / 5.47; adaptive 6.10 / 5.05) — warm-up rows skip the horizontal pass, so short
    //          bands cost little arithmetic; the general 4-channel pass alone would prefer ~64 rows (-4 % here)
    //   k >= 7: VALU-bound, the 2R warm-up rows hurt: tall adaptive bands + short-band tail
    return (K == 3)   ? make_band_plan(h, sp->nstrips, nframes, 5, 12, 12, 12, 0.0, 6, plan)
           : (K == 5) ? make_band_plan(h, sp->nstrips, nframes, 4, 24, 24, 24, 0.0, 12, plan)
                      : make_band_plan(h, sp->nstrips, nframes, 3, 96, 270, 40, 0.1, 4 * R + 4, plan);
}

template <int R>
hipError_t launch_r(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                    const GaussCoef& coef, uint32_t* d_flags)
{
    constexpr int K = 2 * R + 1;
    constexpr bool kSplit = R >= 3;  // two kernels + flags
    StripPlan sp;
    BandPlan plan;
    if (!slide_plan<R>(w, h, nframes, &sp, &plan) || (kSplit && !d_flags))
        return hipErrorInvalidValue;
    const int nstrips = sp.nstrips, lanes_out = sp.lanes_out;
    const bool ragged = (w & 3) != 0 ||
                        (((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15u) != 0);
    Weights<K> wts;
    for (int j = 0; j < K; j++)
        wts.w[j] = coef.h_w1d[j];
    double wsum = 0.0;
    for (int j = 0; j < K; j++)
        wsum += (double)coef.h_w1d[j];
    const bool clamp = !(255.0 * wsum * wsum * 1.0001 < 256.0);  // externally installed tables may overflow
    // constant alpha bytes of the fast path: the canonical chains on an all-A channel, evaluated by the host when
    // the table was installed (kernels.hpp: gauss_const_alpha); A = 255, the usual case, travels by value
    if (!coef.d_alpha_tab)
        return hipErrorInvalidValue;
    const uint32_t alpha_hi = coef.h_alpha_tab[255];
    const uint32_t* alpha_tab = coef.d_alpha_tab;
    const dim3 grid(plan.nblocks_a + plan.nblocks_b), block(kWavesPerBlock * 64);
    // lock-step rows (see the row loop) where a workgroup is kWavesPerBlock adjacent strips of one band — with 5 strips
    // per row (width 1023) the coupled waves belong to different bands and the same barrier costs 4 %, on one-strip
    // frames 7 % — and the launch is several times what the chip holds at once (1-4 4K frames, cache-resident: -1.2 %)
    const bool lockstep = R <= 2 && !ragged && nstrips % kWavesPerBlock == 0 && plan.nwork_b == 0 && plan.nwork_a >= 8192u;
#define MI355_LAUNCH1(CL, RG, MD, LS)                                                                               \
    hipLaunchKernelGGL((gauss_slide_kernel<R, CL, RG, MD, LS>), grid, block, 0, stream, d_in, d_out, w, h, nstrips, \
                       lanes_out, plan, wts, alpha_hi, alpha_tab, d_flags)
#define MI355_LAUNCH(CL, RG)                  \
    do {                                      \
        if constexpr (kSplit) {               \
            MI355_LAUNCH1(CL, RG, 3, false);  \
            MI355_LAUNCH1(CL, RG, 4, false);  \
        } else if constexpr (!RG) {           \
            if (lockstep)                     \
                MI355_LAUNCH1(CL, RG, 0, true);  \
            else                              \
                MI355_LAUNCH1(CL, RG, 0, false); \
        } else {                              \
            MI355_LAUNCH1(CL, RG, 0, false);  \
        }                                     \
    } while (0)
    if (clamp && ragged)
        MI355_LAUNCH(true, true);
    else if (clamp)
        MI355_LAUNCH(true, false);
    else if (ragged)
        MI355_LAUNCH(false, true);
    else
        MI355_LAUNCH(false, false);
#undef MI355_LAUNCH
#undef MI355_LAUNCH1
    return hipGetLastError();
}

}  // namespace

bool gauss_slide_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, int k)
{
    (void)h;
    (void)w;
    if (k != 3 && k != 5 && k != 7 && k != 9)
        return false;
    // any width: rows that are not 16-byte aligned take the RAGGED variant; pixels are dwords
    return ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 3u) == 0;
}

size_t gauss_slide_flag_items(int w, int h, int nframes, int k)
{
    StripPlan sp;
    BandPlan plan;
    bool ok = false;
    switch (k) {  // only the two-kernel variants (k = 7, 9) use flags
    case 7: ok = slide_plan<3>(w, h, nframes, &sp, &plan); break;
    case 9: ok = slide_plan<4>(w, h, nframes, &sp, &plan); break;
    default: break;
    }
    return ok ? (size_t)plan.nwork_a + plan.nwork_b : 0;
}

hipError_t launch_gauss_slide(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                              int nframes, const GaussCoef& coef, uint32_t* d_flags)
{
    switch (coef.k) {
    case 3: return launch_r<1>(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    case 5: return launch_r<2>(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    case 7: return launch_r<3>(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    case 9: return launch_r<4>(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mi355
