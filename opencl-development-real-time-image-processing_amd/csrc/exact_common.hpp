// exact_common.hpp — "exact by exception": what the fused pipeline (pipe_slide.hip) and the EXACT-mode Gaussian
// (gauss_exact.hip) share.  The CPU path's byte is trunc(S_cpu), S_cpu = its k*k-term float sum in its own order
// (src/GaussianBlur/GaussianBlur.cpp:243-256: ky outer, kx inner, separate multiply and add).  A separable fp32
// evaluation S differs from S_cpu by at most delta_bound(); wherever S is further than that from an integer,
// trunc(S) IS the CPU byte, and the few values within it are recomputed with exact_sum(), the CPU path's own chain.
#pragma once
#include <cmath>

#include "common.hpp"

namespace mi355 {

__device__ __forceinline__ float dppl(float v)  // lane l <- lane l-1
{
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}

__device__ __forceinline__ float dppr(float v)  // lane l <- lane l+1
{
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

// The CPU path's own sum for pixel J of every lane (GaussianBlur.cpp:243-256: ky outer, kx inner, float multiply
// then float add, starting from 0), from the ring of gray rows: the window's arrival rows sit in slots
// (u + 1 + t) % K, t = 0 .. K-1, and the rows are visited in IMAGE order, top to bottom — arrival order for a band
// walking down, the reverse for a band walking up (UP).  u is a constant after unrolling, so every register index
// is static.  Runs under a wave-uniform branch: EXEC is full, the DPP reads see every lane.
// PX = pixels per lane (4; 8 in pipe_slide8.hip).
template <int K, int J, bool UP, int PX = 4>
__device__ __forceinline__ float exact_sum(const float (&g)[K][PX], int u, const float* __restrict__ w2)
{
    constexpr int R = K / 2;
    float sum = 0.0f;
#pragma unroll
    for (int ky = 0; ky < K; ky++) {
        const float* r = g[(u + 1 + (UP ? K - 1 - ky : ky)) % K];
#pragma unroll
        for (int kx = 0; kx < K; kx++) {
            const int col = J - R + kx;
            const float val = (col < 0) ? dppl(r[PX + col]) : ((col > PX - 1) ? dppr(r[col - PX]) : r[col]);
            sum = sum + val * w2[ky * K + kx];  // -ffp-contract=off: v_mul_f32 then v_add_f32
        }
    }
    return sum;
}

// Constant windows.  Where the image is flat the separable value sits a few 1e-6 under an integer (the table sums to
// just under one), so EVERY pixel is flagged, and paying the 2 k^2-operation chain for all of them made flat content —
// letterbox bars, saturated regions, graphics, the blocks of a decoded JPEG sky — the worst case (256 x 4K frames of
// 64 x 64 flat patches: fused pipeline 2.6 TB/s against 4.9 on noise, EXACT Gaussian 1.4 against 4.8).  But the CPU path's
// chain over a window whose k^2 values all equal c is a function of c alone: flat_chain() tabulates it once per
// workgroup (256 floats in LDS), and flat_windows() replaces the chain by a table read for every flagged pixel whose
// window is constant: per row, column minima / maxima over the ring (2 (K - 1) operations per column, neighbour columns
// through DPP), per pixel a 2R + 1-column minimum / maximum and one comparison.  Called under a wave-uniform branch, and
// only when many lanes are flagged at once (dense_flags): noise-like frames never enter it.
template <int K>
__device__ __forceinline__ float flat_chain(float c, const float* __restrict__ w2)
{
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < K * K; i++)
        sum = sum + c * w2[i];  // -ffp-contract=off: the CPU path's multiply, then its add (GaussianBlur.cpp:249-252)
    return sum;
}

__device__ __forceinline__ bool dense_flags(uint64_t ballot)
{
    // wave-uniform.  On noise-like frames a flagged row has ONE flagged lane (a lane-row is flagged with probability
    // ~3e-3: three at once once in a thousand rows); a flat area a few lanes wide is worth the ~130-instruction search,
    // which costs about two chains.
    return __builtin_popcountll(ballot) >= 3;
}

// g = the ring exact_sum() reads (all K slots are the window's rows).  For every pixel J whose flag is up (t[J] <
// two_delta) and whose K x K window is constant: S[J] = flat[c], the flag goes down (t[J] = 1).
template <int K, int PX>
__device__ __forceinline__ void flat_windows(const float (&g)[K][PX], float (&S)[PX], float (&t)[PX], float two_delta,
                                             const float* flat)
{
    constexpr int R = K / 2;
    float mn[PX + 2 * R], mx[PX + 2 * R];
#pragma unroll
    for (int e = 0; e < PX; e++) {
        float a = g[0][e], b = g[0][e];
#pragma unroll
        for (int s = 1; s < K; s++) {
            a = fminf(a, g[s][e]);
            b = fmaxf(b, g[s][e]);
        }
        mn[R + e] = a;
        mx[R + e] = b;
    }
#pragma unroll
    for (int i = 0; i < R; i++) {
        mn[R - 1 - i] = dppl(mn[R + PX - 1 - i]);  // column -1 - i = the left lane's column PX - 1 - i
        mx[R - 1 - i] = dppl(mx[R + PX - 1 - i]);
        mn[R + PX + i] = dppr(mn[R + i]);          // column PX + i = the right lane's column i
        mx[R + PX + i] = dppr(mx[R + i]);
    }
#pragma unroll
    for (int J = 0; J < PX; J++) {
        float a = mn[J], b = mx[J];
#pragma unroll
        for (int c = 1; c <= 2 * R; c++) {
            a = fminf(a, mn[J + c]);
            b = fmaxf(b, mx[J + c]);
        }
        const bool is_flat = (a == b) && (t[J] < two_delta);
        if (__builtin_amdgcn_ballot_w64(is_flat) != 0) {
            const float val = flat[(uint32_t)a];  // a is a byte value held as a float
            S[J] = is_flat ? val : S[J];
            t[J] = is_flat ? 1.0f : t[J];
        }
    }
}

// u(x) = half an ulp of a float of magnitude <= x
inline double half_ulp(double x)
{
    if (!(x > 0.0))
        return 0.0;
    int e = 0;
    std::frexp(x, &e);  // x = m * 2^e, m in [0.5, 1)  ->  ulp = 2^(e - 24)
    return std::ldexp(1.0, e - 25);
}

// |S - S_cpu| <= delta for every window of bytes, where S is the kernel's separable pair-form evaluation with w1 and
// S_cpu the CPU path's k*k-term float sum with w2.  Everything is non-negative (checked by the caller), so partial
// sums never exceed the final ones.
template <int K>
double delta_bound(const float* w1, const float* w2)
{
    constexpr int R = K / 2;
    double sum2 = 0.0, max2 = 0.0, sum1 = 0.0, mismatch = 0.0;
    for (int i = 0; i < K * K; i++) {
        sum2 += (double)w2[i];
        max2 = std::fmax(max2, (double)w2[i]);
    }
    for (int i = 0; i < K; i++)
        sum1 += (double)w1[i];
    for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++)
            mismatch += std::fabs((double)w1[i] * (double)w1[j] - (double)w2[i * K + j]);
    // CPU path (GaussianBlur.cpp:243-256): term i (row-major, ky outer / kx inner) is a product of a byte and w2[i],
    // rounded (error <= half an ulp of 255 * w2[i]), added to the running sum, rounded again — and the running sum
    // after term i cannot exceed 255 * (w2[0] + ... + w2[i]) plus the error made so far (every term is non-negative),
    // so the early additions happen in low binades.  (Round 2, second half: the bound used to charge every product
    // with the largest weight's ulp and every addition with the final sum's: 2.1e-4 at k = 5, now 1.4e-4.)
    double e_cpu = 0.0, cum = 0.0;
    for (int i = 0; i < K * K; i++) {
        cum += (double)w2[i];
        e_cpu += half_ulp(255.0 * (double)w2[i]) + half_ulp(255.0 * cum + 1e-3);
    }
    (void)max2;
    // kernel, vertical: acc = w(0) g_c, then acc = fma(w(d), g_{c-d} + g_{c+d}, acc), d = 1 .. R (the integer pair sums
    // are exact): R + 1 roundings, the one after distance d of a value <= 255 * c_d, c_d = w(0) + 2 (w(1) + ... + w(d))
    const double tv = 255.0 * sum1;
    double c_d = (double)w1[R];
    double e_v = half_ulp(255.0 * c_d);
    for (int d = 1; d <= R; d++) {
        c_d += 2.0 * (double)w1[R - d];
        e_v += half_ulp(255.0 * c_d + 1e-3);
    }
    // kernel, horizontal: acc = fma(w(0), v_c, delta), then acc = fma(w(d), v_{c-d} + v_{c+d}, acc): the R pair sums are
    // rounded float additions of values <= tv (weighted by w(d) afterwards), the R + 1 fused operations round values
    // <= tv * c_d + delta (delta < 0.01, checked by the callers); the vertical error arrives through weights summing to sum1
    double e_pairs = 0.0;
    for (int d = 1; d <= R; d++)
        e_pairs += (double)w1[R - d] * half_ulp(2.0 * tv + 1e-3);
    c_d = (double)w1[R];
    double e_chain = half_ulp(tv * c_d + 0.011);
    for (int d = 1; d <= R; d++) {
        c_d += 2.0 * (double)w1[R - d];
        e_chain += half_ulp(tv * c_d + 0.011);
    }
    const double e_h = e_v * sum1 + e_pairs + e_chain;
    (void)sum2;
    // every term above is a worst case already; the margin only covers the double arithmetic of this function
    return 1.02 * (e_cpu + e_h + 255.0 * mismatch) + 1e-7;
}


}  // namespace mi355
