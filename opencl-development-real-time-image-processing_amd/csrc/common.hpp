// common.hpp — device helpers shared by the gfx950 filter kernels.
//
// Arithmetic contracts (DESIGN.md "Arithmetic"):
//  * luminance: the reference CPU path src/Grayscale/grayscale.cpp:237,
//      uchar(0.299*r + 0.587*g + 0.114*b), evaluated in FP64, left to right, no FMA
//      contraction, truncation.  The whole library is compiled with -ffp-contract=off so that
//      `a*b + c` never fuses; fused multiply-adds are written explicitly (__builtin_fmaf).
//  * Sobel magnitude: integer gx, gy; out = min(255, round-half-even(sqrt(gx^2+gy^2))), which for
//      integer s = gx^2+gy^2 < 2^24 equals k + (s > k*k + k), k = floor(sqrt(s)) (no ties exist).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

namespace mi355 {

constexpr int kWave = 64;

// MI355_TUNE_* environment overrides (band heights, strip widths, kernel selection) exist for tuning sweeps
// and for the test that proves outputs do not depend on the work decomposition.  They are compiled in only
// with -DMI355_TUNE_ENV (lib/libmi355_imgfilter_tune.so, `make tune`); the product library reads no environment.
inline const char* tune_env(const char* name)
{
#ifdef MI355_TUNE_ENV
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// 16-byte vector of four RGBA pixels (one dword each): the unit of every coalesced access
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ uint32_t luma_rgb(uint32_t r, uint32_t g, uint32_t b)
{
    // ((0.299*r) + (0.587*g)) + (0.114*b) in double; -ffp-contract=off keeps mul and add apart
    double v = 0.299 * (double)r + 0.587 * (double)g + 0.114 * (double)b;
    return (uint32_t)v;  // C truncation; v is in [0, 255.0000000000001]
}

// px = R | G<<8 | B<<16 | A<<24 (little-endian load of an RGBA pixel)
__device__ __forceinline__ uint32_t luma_px(uint32_t px)
{
    return luma_rgb(px & 0xFFu, (px >> 8) & 0xFFu, (px >> 16) & 0xFFu);
}

// Same value as luma_px without FP64 on the common path (FP64 ops issue at ~5.7 cycles per wave64 on
// gfx950, fp32 add/fma at 2, everything else at ~3).  With S = 299r + 587g + 114b the double-precision sum
// truncates to floor(S/1000) for every colour except some with S % 1000 == 0, where it can land just below
// the integer (3,464 of the 2^24 colours; checked exhaustively on CPU and on GPU).  So: q = floor(S/1000)
// exactly, and only pixels with S == 1000q (0.1 % of colours) take the FP64 formula.
//   S       = 256 (r + 2g) + (43r + 75g + 114b)       two v_dot4_u32_u8 + one shift-add, exact integer
//   q       = floor(S * 0.001f + 0.0005f)              S/1000 = n + j/1000, so +0.0005 keeps the argument
//                                                      >= 0.0005 away from an integer; float error < 4e-5
// Returns q as a float (all callers continue in fp32, where these small integers are exact).
__device__ __forceinline__ float luma_px_fast(uint32_t px)
{
    const uint32_t hi = __builtin_amdgcn_udot4(px, 0x00000201u, 0u, false);
    const uint32_t lo = __builtin_amdgcn_udot4(px, 0x00724B2Bu, 0u, false);
    const float S = (float)((hi << 8) + lo);  // <= 255000 < 2^24: exact
    float q = __builtin_floorf(__builtin_fmaf(S, 0.001f, 0.0005f));
    if (__builtin_fmaf(q, -1000.0f, S) == 0.0f)  // exact test; rare: the reference formula picks q or q - 1
        q = (float)luma_px(px);
    return q;
}

// The ambiguous case of a GRAY pixel (r = g = b = v: S = 1000 v, always ambiguous) read from a table: gray_lut[v] =
// luma_rgb(v, v, v), 256 bytes in LDS that a kernel fills once per workgroup (fill_gray_lut).  On colour content one pixel
// in a thousand takes the FP64 formula; on gray content — monochrome cameras, documents, the reference's own Artemis
// photographs — EVERY pixel did (256 x 4K frames of gray noise: Sobel 4.2 TB/s against 5.9 on colour noise, fused
// pipeline 4.05 against 4.9).  Other ambiguous colours keep the FP64 formula.
__device__ __forceinline__ void fill_gray_lut(uint8_t* gray_lut)  // 256 threads; the caller's __syncthreads() follows
{
    gray_lut[threadIdx.x] = (uint8_t)luma_rgb(threadIdx.x, threadIdx.x, threadIdx.x);
}

__device__ __forceinline__ uint32_t luma_px_ambiguous(uint32_t px, const uint8_t* gray_lut)
{
    if (((px ^ (px >> 8)) & 0xFFFFu) == 0u)  // r == g and g == b
        return gray_lut[px & 0xFFu];
    return luma_px(px);
}

// Four pixels at once: the same values, but one exception test for the whole quad (the minimum of the four
// remainders; they are >= 0 because q never exceeds S/1000), i.e. one branch per lane-quad instead of four.
__device__ __forceinline__ void luma_quad_fast(const u32x4& p, float g[4])
{
    float rem[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t hi = __builtin_amdgcn_udot4(p[j], 0x00000201u, 0u, false);
        const uint32_t lo = __builtin_amdgcn_udot4(p[j], 0x00724B2Bu, 0u, false);
        const float S = (float)((hi << 8) + lo);
        g[j] = __builtin_floorf(__builtin_fmaf(S, 0.001f, 0.0005f));
        rem[j] = __builtin_fmaf(g[j], -1000.0f, S);  // exact: an integer in [0, 999]
    }
    if (fminf(fminf(rem[0], rem[1]), fminf(rem[2], rem[3])) == 0.0f) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (rem[j] == 0.0f)
                g[j] = (float)luma_px(p[j]);
    }
}

// gray_run: a wave-uniform hint the caller carries from row to row.  It goes up when most lanes of a row met the
// ambiguous case (gray content does that on every pixel) and then the NEXT row first asks whether all its 256 pixels
// are gray — one OR-tree and one ballot — and if so reads the four luminances straight from the table: no dot
// products, no quotient, no exception branch, so gray frames cost what colour frames cost.  Colour content never raises
// the hint and pays one scalar branch per row.
__device__ __forceinline__ bool gray_row(const u32x4& p, float g[4], const uint8_t* gray_lut)
{
    const uint32_t ng = ((p.x ^ (p.x >> 8)) | (p.y ^ (p.y >> 8)) | (p.z ^ (p.z >> 8)) | (p.w ^ (p.w >> 8))) & 0xFFFFu;
    if (__builtin_amdgcn_ballot_w64(ng != 0u) != 0)
        return false;
#pragma unroll
    for (int j = 0; j < 4; j++)
        g[j] = (float)gray_lut[p[j] & 0xFFu];
    return true;
}

__device__ __forceinline__ void luma_quad_fast(const u32x4& p, float g[4], const uint8_t* gray_lut, bool& gray_run)
{
    if (gray_run) {  // wave-uniform
        if (gray_row(p, g, gray_lut))
            return;
        gray_run = false;
    }
    float rem[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t hi = __builtin_amdgcn_udot4(p[j], 0x00000201u, 0u, false);
        const uint32_t lo = __builtin_amdgcn_udot4(p[j], 0x00724B2Bu, 0u, false);
        const float S = (float)((hi << 8) + lo);
        g[j] = __builtin_floorf(__builtin_fmaf(S, 0.001f, 0.0005f));
        rem[j] = __builtin_fmaf(g[j], -1000.0f, S);  // exact: an integer in [0, 999]
    }
    const bool amb = fminf(fminf(rem[0], rem[1]), fminf(rem[2], rem[3])) == 0.0f;
    gray_run = __builtin_popcountll(__builtin_amdgcn_ballot_w64(amb)) >= 48;
    if (amb) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (rem[j] == 0.0f)
                g[j] = (float)luma_px_ambiguous(p[j], gray_lut);
    }
}

// The same four values with the quotient taken in integer arithmetic.  With M = 4294968 = ceil(2^32 / 1000) the
// 48-bit product S * M = floor(S / 1000) * 2^32 + (S % 1000) * 4294967.3 + 0.704 * S, S <= 255000, so
//   q   = high 32 bits (v_mul_hi_u32_u24) = floor(S / 1000) exactly (the excess 0.704 S / 2^32 < 4.2e-5 never
//         carries across a multiple of 1/1000), and
//   low = low 32 bits (v_mul_u32_u24) <= 179,520 when S % 1000 == 0 and >= 4,294,967 otherwise,
// which is the exception test for free.  Six VALU instructions per pixel instead of seven and no fp32 rounding
// argument; used by the fused pipeline, which is VALU-bound.  mi355_selftest runs all 2^24 colours through it.
__device__ __forceinline__ void luma_quad_int(const u32x4& p, float g[4])
{
    constexpr uint32_t M = 4294968u;
    uint32_t q[4], low[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t hi = __builtin_amdgcn_udot4(p[j], 0x00000201u, 0u, false);
        const uint32_t lo = __builtin_amdgcn_udot4(p[j], 0x00724B2Bu, 0u, false);
        const uint32_t S = (hi << 8) + lo;
        __builtin_assume(S < (1u << 18));  // <= 255000: lets instruction selection take the 24-bit multiplies
        q[j] = (uint32_t)(((uint64_t)S * M) >> 32);
        low[j] = __umul24(S, M);
    }
    if (min(min(low[0], low[1]), min(low[2], low[3])) < 1000000u) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (low[j] < 1000000u)
                q[j] = luma_px(p[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        g[j] = (float)q[j];
}

// (No gray_run hint here: the fused pipeline's rows are bound by the instructions BEHIND the luminance, the all-gray row
// path gained it nothing on gray frames, and the extra branch between the loads and their first use cost its ragged-width
// variant 13 % — hipcc drains the memory counter at such joins.)
__device__ __forceinline__ void luma_quad_int(const u32x4& p, float g[4], const uint8_t* gray_lut)
{
    constexpr uint32_t M = 4294968u;
    uint32_t q[4], low[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t hi = __builtin_amdgcn_udot4(p[j], 0x00000201u, 0u, false);
        const uint32_t lo = __builtin_amdgcn_udot4(p[j], 0x00724B2Bu, 0u, false);
        const uint32_t S = (hi << 8) + lo;
        __builtin_assume(S < (1u << 18));  // <= 255000: lets instruction selection take the 24-bit multiplies
        q[j] = (uint32_t)(((uint64_t)S * M) >> 32);
        low[j] = __umul24(S, M);
    }
    if (min(min(low[0], low[1]), min(low[2], low[3])) < 1000000u) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (low[j] < 1000000u)
                q[j] = luma_px_ambiguous(p[j], gray_lut);
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        g[j] = (float)q[j];
}

// min(255, round-half-even(sqrt(gx^2 + gy^2))) for integer-valued floats |gx|, |gy| <= 1020 — what the
// reference computes as saturate_cast<uchar>(lrint(sqrt(s))) — in four VALU ops per pixel, packing included:
// s = gx^2 + gy^2 is exact in fp32 (< 2^21); v_cvt_pk_u8_f32 rounds to nearest-even, saturates at 255 and
// inserts the byte.  Rounding the 1-ulp v_sqrt_f32 instead of the exact root cannot change the result: for
// integer s the root is never within 1/(8n+4) of a half-integer n + 1/2 (the closest, s = n^2 + n, is
// 0.125/(n + 0.5) below it, >= 4.9e-4 for n <= 255, against an ulp of 1.5e-5), and everything above 255.5
// saturates.  mi355_selftest checks all 1021 x 1021 (|gx|, |gy|) pairs on the device against sobel_mag_u8.
__device__ __forceinline__ uint32_t sobel_mag_pack(float gx, float gy, uint32_t byte, uint32_t old)
{
    const float s = __builtin_fmaf(gx, gx, gy * gy);
    return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_sqrtf(s), byte, old);
}

__device__ __forceinline__ uint32_t sobel_mag_fast(float gx, float gy)
{
    return sobel_mag_pack(gx, gy, 0u, 0u);
}

__device__ __forceinline__ uint32_t sobel_mag_quad(const float gx[4], const float gy[4])
{
    uint32_t r = 0u;
#pragma unroll
    for (int j = 0; j < 4; j++)
        r = sobel_mag_pack(gx[j], gy[j], (uint32_t)j, r);
    return r;
}

__device__ __forceinline__ uint32_t gray_to_rgba(uint32_t g)
{
    return g * 0x00010101u | 0xFF000000u;
}

// min(255, round(sqrt(s))) for 0 <= s < 2^24, exact.
__device__ __forceinline__ uint32_t sobel_mag_u8(int gx, int gy)
{
    uint32_t s = (uint32_t)(gx * gx + gy * gy);
    if (s >= 65281u)  // sqrt(s) > 255.5
        return 255u;
    uint32_t k = (uint32_t)__builtin_amdgcn_sqrtf((float)s);  // within 1 of floor(sqrt(s))
    k -= (k * k > s) ? 1u : 0u;
    k += ((k + 1u) * (k + 1u) <= s) ? 1u : 0u;
    k += (s > k * k + k) ? 1u : 0u;
    return k;  // <= 255 because s < 65281
}

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while ((unsigned)p >= (unsigned)len)
        p = (p < 0) ? -p : 2 * (len - 1) - p;
    return p;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi)
{
    return v < lo ? lo : (v > hi ? hi : v);
}

// float -> u8 exactly as the CPU path: uchar(std::clamp(sum, 0.f, 255.f)) (truncation)
__device__ __forceinline__ uint32_t f2u8(float v)
{
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    return (uint32_t)v;
}

__device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

__device__ __forceinline__ uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xFF51AFD7ED558CCDull;
    k ^= k >> 33;
    k *= 0xC4CEB9FE1A85EC53ull;
    k ^= k >> 33;
    return k;
}

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (observed round-robin
// placement, speed only), so give each of the 8 residue classes one contiguous chunk of the
// logical tile list; neighbouring tiles (shared halo rows/columns) then meet in one L2.
__device__ __forceinline__ uint32_t xcd_remap_contiguous(uint32_t bid, uint32_t nblk)
{
    const uint32_t q = nblk >> 3, r = nblk & 7u;
    const uint32_t xcd = bid & 7u, idx = bid >> 3;
    const uint32_t base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + idx;
}

// kXcdRun > 0: the list is cut into groups of 8 runs of kXcdRun blocks and XCD x takes run x of every group, so the eight
// XCDs work on eight NEIGHBOURING runs at a time instead of eight regions an eighth of the batch (1 GiB) apart; the
// remainder of the list keeps the contiguous mapping.  Bijective.  0 = contiguous mapping everywhere.
constexpr uint32_t kXcdRun = 0;

__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk)
{
    if constexpr (kXcdRun == 0) {
        return xcd_remap_contiguous(bid, nblk);
    } else {
        const uint32_t full = nblk / (8u * kXcdRun) * (8u * kXcdRun);
        if (bid >= full)
            return full + xcd_remap_contiguous(bid - full, nblk - full);
        const uint32_t xcd = bid & 7u, idx = bid >> 3;
        return ((idx / kXcdRun) * 8u + xcd) * kXcdRun + idx % kXcdRun;
    }
}

}  // namespace mi355
