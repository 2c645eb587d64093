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

namespace mi355 {

constexpr int kWave = 64;

// 16-byte vector of four RGBA pixels (one dword each): the unit of every coalesced access
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
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

// min(255, round-half-even(sqrt(gx^2 + gy^2))) for integer-valued floats |gx|, |gy| <= 1020, without
// integer fix-ups: with s = gx^2 + gy^2 (exact in fp32), round(sqrt(s)) = floor(0.5 + 0.5 * sqrt(4s - 1))
// for s >= 1 — 4s-1 is never a perfect square and the argument of floor stays >= 1/511 away from an integer
// for results <= 255, far more than the error of v_sqrt_f32.  Brute-forced against the exact integer form
// for every s < 2^21, also with a +-2 ulp sqrt.  The float -> u32 conversion truncates, which is the floor.
__device__ __forceinline__ uint32_t sobel_mag_fast(float gx, float gy)
{
    const float s = __builtin_fmaf(gx, gx, gy * gy);
    const float t = fmaxf(__builtin_fmaf(4.0f, s, -1.0f), 0.0f);
    const float u = __builtin_amdgcn_sqrtf(t);
    return (uint32_t)fminf(__builtin_fmaf(u, 0.5f, 0.5f), 255.5f);
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
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk)
{
    const uint32_t q = nblk >> 3, r = nblk & 7u;
    const uint32_t xcd = bid & 7u, idx = bid >> 3;
    const uint32_t base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + idx;
}

}  // namespace mi355
