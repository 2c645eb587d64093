// gauss_mfma.hip — separable Gaussian blur of RGBA8 frames on the matrix cores, any odd k <= 17 (the reference
// application's own default is k = 17, sigma = 6: include/ProgramHandler.hpp:9), width % 4 == 0.  gfx950 only.
// The LDS-STAGED version: the first matrix-core kernel of this library.  gauss_mfma_reg.hip (operands loaded straight
// into the matrix layout, horizontal pass first) is 15-20 % faster and is what MI355_IMPL_MFMA / AUTO launch; this one
// stays as its A/B partner (tuning build, MI355_MFMA_LDS=1: tools/mfma_reg_ab.sh, tests/test_gpu_mfma.py) and serves
// frames of 2 GiB and more, which the other kernel's 32-bit row offsets exclude.
//
// Why matrix cores for a stencil: at k = 17 the separable blur costs 2 * 17 * 4 = 136 multiply-adds per pixel; the
// register-resident VALU kernel (gauss_wide.hip) is FP32-issue-bound at 2.1 TB/s (26 % of the HBM roofline).  Each
// pass is a product with a banded (Toeplitz) matrix, and v_mfma_f32_16x16x32_f16 has K = 32 = 16 outputs + 2 * 8
// halo taps: ONE matrix instruction blurs a 16 x 16 tile along one axis for any radius <= 8.
//
// Arithmetic (FAST contract: within 1 LSB per channel of the CPU path src/GaussianBlur/GaussianBlur.cpp:234-261):
//   * pixels are bytes: exact in fp16.  Weights are fp32: each is split hi + lo (two fp16, together 22 bits) and
//     both parts are accumulated by the matrix instruction in fp32, so pass 1 (vertical) carries ~1e-6 relative
//     error.  Its fp32 result is split the same way (hi + lo) and pass 2 (horizontal) sums hi*Whi + lo*Whi + hi*Wlo.
//     Both passes scale their weights by 256 so that every lo part stays a NORMAL fp16 (nothing depends on how the
//     matrix unit treats fp16 subnormals) and the intermediate 256 * V <= 65280 still fits fp16.
//   * the result is 65536 * sum in fp32 (< 2^24): truncation is the integer conversion followed by taking byte 2.
//   Measured against the oracle: max |d| = 1, mismatching bytes ~1e-4 (tests/test_gpu_mfma.py).
//
// Layout.  A workgroup (4 waves) owns a column strip of 64 output pixels and walks down a band of 16-row blocks.
// LDS holds the bytes of the current window as four fp16 planes (R, G, B, A), row-major, 80 columns (8 halo pixels
// per side), in a ring of three 16-row slabs; a block reads two slabs (32 rows = 16 outputs + 8 + 8) while the
// third is refilled.  Wave v owns the 16-pixel sub-strip v, all four channels:
//   pass 1  D1[x][y'] = sum_y X^T[x][y] * Tv^T[y][y']   A = X^T read with ds_read_b64_tr_b16 (hardware transpose of
//           the row-major plane), B = the banded weights (constant registers).  Two 16-column tiles (x0-8.. and
//           x0+8..) per channel.
//   pass 2  Z[x'][y'] = sum_x Th^T[x'][x] * D1[x][y']    the accumulator tile IS the next B operand (its row index
//           x lives in the registers, its column y' on the lanes): no LDS round trip, no lane movement.  The k order
//           that falls out of stacking two accumulator tiles is baked into the constant A operand.
//   Z has y' on the lanes and 4 consecutive x' in the registers: each lane packs RGBA for 4 adjacent pixels and
//   stores 16 bytes.
// The MFMA k index is only a summation index, so its mapping to window rows is chosen for the LDS banks: group h of
// 16 lanes reads window rows 4h..4h+3 and 16+4h..16+4h+3; with 160-byte rows (40 dwords) the eight rows that one
// 32-lane half touches start at banks 0, 40, 16, 56, 32, 8, 48, 24: conflict-free.
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace mi355 {

namespace {

#ifndef MI355_MFMA_TX
#define MI355_MFMA_TX 64
#endif
constexpr int kTX = MI355_MFMA_TX;                 // output pixels per strip (16 per wave)
constexpr int kWaves = kTX / 16;
constexpr int kQuadsPerRow = kTX / 4;
constexpr int kHalo = 8;                           // halo pixels per side (radius <= 8)
constexpr int kCols = kTX + 2 * kHalo;             // 80 staged columns
constexpr int kPitch = kCols;                      // fp16 elements per LDS row: 160 B
constexpr int kSlabRows = 16;
constexpr int kSlabElems = kSlabRows * kPitch;     // per plane
constexpr int kPlaneElems = 3 * kSlabElems;        // ring of three slabs
constexpr int kThreads = 64 * kWaves;              // = 16 rows x kQuadsPerRow quads: one main quad per thread per slab

typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
using lds_fp16x4 = __attribute__((address_space(3))) fp16x4;

struct MWeights {
    float w[17];  // w[8 + d], d = -8 .. 8: the separable factor centred, zero beyond the radius
};

struct MPlan {
    int nstrips, nbands, blocks_per_band, nblocks;  // nblocks = ceil(h / 16)
    uint32_t nwork;
};

constexpr int kOutPitch = kTX + 4;                  // dwords per row of the output tile: 256 B + 16 B of padding
constexpr int kTapPad = 40;  // taps are looked up at distances -39 .. +39: a zero-padded table, no branches

__device__ __forceinline__ void split16(float x, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// two floats -> two fp16 in one dword (v_cvt_pkrtz_f16_f32); bytes are exact under any rounding, and a
// round-toward-zero hi part only makes the lo part non-negative
__device__ __forceinline__ uint32_t pk16(float a, float b)
{
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}

// v - (float)half of hp, exact in fp32, in ONE instruction: v_fma_mix_f32 reads the fp16 operand as it is
// (op_sel_hi marks src0 as fp16, op_sel picks its high or low half).  hipcc emits v_cvt_f32_f16 + v_sub_f32.
template <int HALF>
__device__ __forceinline__ float residual(uint32_t hp, float v)
{
    float r;
    if constexpr (HALF == 0)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
    return r;
}

// a0 / a1: LDS byte addresses (one VGPR each per block); off: compile-time byte offset -> the instruction's offset field
__device__ __forceinline__ h8 tr_read2(uint32_t a0, uint32_t a1, int off)
{
    const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(uintptr_t)(a0 + off));
    const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(uintptr_t)(a1 + off));
    h8 r;
    __builtin_memcpy(&r, &a, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&r) + 8, &b, 8);
    return r;
}

template <bool CLAMP>
__global__ __launch_bounds__(kThreads) void gauss_mfma_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                             int w, int h, MPlan plan, MWeights W, float alpha_top,
                                                             float plane_bias)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[4 * kPlaneElems];  // 30,720 B
    __shared__ __attribute__((aligned(16))) uint32_t otile[2 * 16 * kOutPitch];  // two output tiles, 8,704 B
    __shared__ uint32_t clear_flag[3 * kWaves];  // [slab slot][wave]: something non-opaque was staged
    __shared__ float wtab[2 * kTapPad];  // wtab[kTapPad + d] = 256 * w(d), zero beyond the radius
    if (threadIdx.x < 2 * kTapPad) {
        const int d = (int)threadIdx.x - kTapPad;
        float v = 0.0f;
#pragma unroll
        for (int t = 0; t < 17; t++)  // kernarg reads with static indices: no scratch, no divergence
            v = (d == t - 8) ? W.w[t] * 256.0f : v;
        wtab[threadIdx.x] = v;
    }
    __syncthreads();

    const uint32_t work = xcd_remap(blockIdx.x, plan.nwork);
    const int strip = work % plan.nstrips;
    const int band = (work / plan.nstrips) % plan.nbands;
    const size_t frame = work / ((uint32_t)plan.nstrips * plan.nbands);
    const int x0 = strip * kTX;
    const int blk0 = band * plan.blocks_per_band;
    const int nb = min(plan.blocks_per_band, plan.nblocks - blk0);
    const int yb0 = blk0 * 16;

    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int hgrp = l >> 4, n = l & 15, q = (l >> 2) & 3, p = l & 3;

    // ---- constant operands ------------------------------------------------------------------------------------
    // pass 1, B[k][y']: k = 8 hgrp + j sums over window row rho(k) = (j < 4 ? 4 hgrp + j : 16 + 4 hgrp + j - 4);
    // window row rho is image row yb - 8 + rho, output row y' is yb + n
    // pass 2, A[x'][k]: k = 8 hgrp + j sums over x = xt - 8 + 16 (j >> 2) + 4 hgrp + (j & 3); x' = xt + n
    h8 b1hi, b1lo, a2hi, a2lo;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int rho = (j < 4) ? 4 * hgrp + j : 16 + 4 * hgrp + (j - 4);
        _Float16 hi, lo;
        split16(wtab[kTapPad + rho - 8 - n], hi, lo);
        b1hi[j] = hi;
        b1lo[j] = lo;
        const int dx = -8 + 16 * (j >> 2) + 4 * hgrp + (j & 3) - n;
        split16(wtab[kTapPad + dx], hi, lo);
        a2hi[j] = hi;
        a2lo[j] = lo;
    }

    // ---- roles.  The first half of the waves LOADS (and converts, and fills the LDS planes), the second half STORES
    // (reads the output tile, writes the rows); every wave computes.  Loads and stores share one in-order counter per
    // wave (vmcnt), and hipcc waits for that counter to reach zero in front of every step's loads: a wave that does
    // both waits for its previous store each step.  (Stores redirected to an L2-resident region: +1 %; stores
    // removed: +14 % — it is the waiting, not the DRAM traffic.)  A wave that only stores never waits at all.
    // Loader thread (r, qs): row r of the slab, quads qs and qs + kQuadsPerRow / 2, and one halo quad when qs < 4.
    constexpr int kLQ = kQuadsPerRow / 2;  // loader threads per row (kThreads / 2 = 16 rows x kLQ)
    const bool is_loader = wv < kWaves / 2;  // wave-uniform
    const int r = (tid / kLQ) & 15, qc = tid % kLQ;
    const int col_main = kHalo + 4 * qc;                             // LDS column of the first main quad
    const int col_halo = (qc < 2) ? 4 * qc : kCols - 8 + 4 * (qc - 2);  // columns 0, 4, 72, 76
    const bool has_halo = qc < 4;
    const uint32_t* fin = reinterpret_cast<const uint32_t*>(in) + frame * (size_t)w * h;
    uint32_t* fout = reinterpret_cast<uint32_t*>(out) + frame * (size_t)w * h;

    // Workgroup-uniform: every staged column of this strip lies inside the image (INTERIOR) or not.  The walk is
    // instantiated twice and entered through one scalar branch: with the clamped-load path as a sibling branch
    // INSIDE the loop, hipcc waits for all outstanding memory operations — the previous block's stores included —
    // before every load (the two paths share destination registers).
    auto walk = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        auto load_quad = [&](const uint32_t* rowp, int x) -> u32x4 {
            if constexpr (INTERIOR)
                return *reinterpret_cast<const u32x4*>(rowp + x);  // w % 4 == 0 and x % 4 == 0: 16-byte aligned
            u32x4 v;
    #pragma unroll
            for (int j = 0; j < 4; j++)
                v[j] = rowp[clampi(x + j, 0, w - 1)];  // clamp-to-edge columns (GaussianBlur.cpp:240)
            return v;
        };
        // Four px dwords -> four fp16 planes in two instructions per pair of values: a byte b placed under the
        // exponent byte 0x64 IS the fp16 number 1024 + b (v_perm_b32 gathers the two bytes, v_or_b32 adds the
        // exponents).  The constant 1024 each value carries comes out of pass 1 as 1024 * (sum of the pass-1 weights),
        // the same for every output, and is cancelled by the accumulator's initial value (plane_bias).
        // The ALPHA plane holds 1024 + (255 - A) (0x64FF64FF minus the gathered bytes, no borrow): an opaque window
        // then blurs to exactly 0 and the output alpha is a host-evaluated constant (see the epilogue), instead of
        // 255 * sum(w) rounding to either side of an integer.
        auto store_quad = [&](int slot, int col, const u32x4& v) {
    #pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t sel = 0x0C000C00u | (uint32_t)c | ((uint32_t)(4 + c) << 16);  // (0, S0.byte c, 0, S1.byte c)
                uint32_t lo = __builtin_amdgcn_perm(v[1], v[0], sel), hi = __builtin_amdgcn_perm(v[3], v[2], sel);
                if (c == 3) {
                    lo = 0x64FF64FFu - lo;
                    hi = 0x64FF64FFu - hi;
                } else {
                    lo |= 0x64006400u;
                    hi |= 0x64006400u;
                }
                *reinterpret_cast<uint2*>(&lds[slot * kSlabElems + r * kPitch + col + c * kPlaneElems]) = uint2{lo, hi};
            }
        };
        // Two register sets: slab s + 2 is written to LDS at the end of block s, slab s + 3 is already on its way —
        // loads are issued two blocks (2 x ~2.5 us) before their data is needed.  With one set (loads one block
        // ahead) the kernel ran at the same speed with its arithmetic removed: it was bound by load latency.
        struct Staged {
            u32x4 main, main2, halo;
        };
        Staged stA, stB;
        auto load_slab = [&](int s, Staged& st) {  // slab s = image rows yb0 - 8 + 16 s .. + 15
            if (!is_loader)
                return;
            const int y = clampi(yb0 - kHalo + 16 * s + r, 0, h - 1);  // clamp-to-edge rows (GaussianBlur.cpp:241)
            const uint32_t* rowp = fin + (size_t)y * w;
            st.main = load_quad(rowp, x0 + 4 * qc);
            st.main2 = load_quad(rowp, x0 + 4 * (qc + kLQ));
            if (has_halo)
                st.halo = load_quad(rowp, x0 - kHalo + col_halo);
        };
        // Each wave also records whether anything it staged into the slab was NOT opaque (alpha != 255): when both
        // slabs of a block are opaque in all four waves, the alpha plane is all zeros, its blur is exactly 0 and the
        // output alpha is the constant — the block skips the alpha channel, a quarter of its matrix and vector work.
        // (Every frame the reference hands its Controller went through cvtColor(BGR2RGBA): RT/src/ProgramHandler.cpp:127.)
        auto write_slab = [&](int s, const Staged& st) {
            if (!is_loader)
                return;
            const int slot = s % 3;
            store_quad(slot, col_main, st.main);
            store_quad(slot, col_main + 4 * kLQ, st.main2);
            uint32_t a = st.main[0] & st.main[1] & st.main[2] & st.main[3] & st.main2[0] & st.main2[1] & st.main2[2] & st.main2[3];
            if (has_halo) {
                store_quad(slot, col_halo, st.halo);
                a &= st.halo[0] & st.halo[1] & st.halo[2] & st.halo[3];
            }
            const bool wave_clear = __builtin_amdgcn_ballot_w64(a < 0xFF000000u) != 0;
            if (l == 0)
                clear_flag[slot * kWaves + wv] = wave_clear ? 1u : 0u;
        };

        load_slab(0, stA);
        write_slab(0, stA);
        load_slab(1, stB);
        write_slab(1, stB);
        load_slab(2, stA);  // (rows clamp: harmless when the band has a single block)
        __syncthreads();

        const uint32_t alpha_const = (uint32_t)alpha_top;  // byte 2 = the opaque alpha
        const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) _Float16*)lds;
        // per-lane offsets of the two transposed reads of a D1 tile (elements): row 4 hgrp + q, column 4 p
        const int tr_off = (4 * hgrp + q) * kPitch + 4 * p + 16 * wv;

        // one block; `cur` holds slab b + 2 (loaded during block b - 1), `nxt` receives slab b + 3
        // Output path.  The accumulator layout leaves lane (n, hgrp) of wave v with row n, pixels 16 v + 4 hgrp .. + 3:
        // stored from there, a wave instruction scatters 64-byte pieces over 16 rows, and with the arithmetic removed the
        // kernel moved its 17 GB at 4.4 TB/s, 8.5 GB of loads alone at the equivalent of 8.8: the stores were the slow
        // half.  So a block's pixels are transposed through an LDS tile (16 rows x 64 px, rows padded by 16 bytes: the
        // ds_write_b128 of 8 lanes lands on 32 different banks) and each wave stores 4 whole rows x 256 contiguous
        // bytes.  Two tiles: block b's tile is read at the top of step b + 1, after that step's loads are issued, and
        // is overwritten at the end of step b + 2 — two barriers later.
        auto out_tile = [&](int b) { return otile + (b & 1) * (16 * kOutPitch); };
        auto store_block = [&](int b) {
            if (is_loader)
                return;
    #pragma unroll
            for (int half = 0; half < 2; half++) {
                const int j = (tid - kThreads / 2) + half * (kThreads / 2);  // quad of the 16 x (kTX / 4) tile
                const int row = j / (kTX / 4), piece = j % (kTX / 4);
                const u32x4 v = *reinterpret_cast<const u32x4*>(out_tile(b) + row * kOutPitch + 4 * piece);
                const int yo = yb0 + 16 * b + row, xo = x0 + 4 * piece;
                u32x4* dst = reinterpret_cast<u32x4*>(fout + (size_t)yo * w + xo);
                if (yo < h && xo < w)  // w % 4 == 0 and xo % 4 == 0: the lane's four pixels are inside together
                    __builtin_nontemporal_store(v, dst);
            }
        };
        auto step = [&](int b, Staged& cur, Staged& nxt) {
            const bool more = b + 1 < nb;
            if (b + 2 < nb)
                load_slab(b + 3, nxt);
            if (b > 0)
                store_block(b - 1);
            // two base pointers per block; plane and tile are immediate offsets of the reads
            uint32_t w0 = lds_base + 2u * (uint32_t)((b % 3) * kSlabElems + tr_off);
            uint32_t w1 = lds_base + 2u * (uint32_t)(((b + 1) % 3) * kSlabElems + tr_off);
            asm volatile("" : "+v"(w0), "+v"(w1));  // keep them as the two bases: hipcc otherwise rebuilds every address
            uint32_t px[4] = {0u, 0u, 0u, 0u}, pz[4] = {0u, 0u, 0u, 0u};
            const int f0 = (b % 3) * kWaves, f1 = ((b + 1) % 3) * kWaves;
            uint32_t any_clear = 0u;
    #pragma unroll
            for (int v = 0; v < kWaves / 2; v++)  // the loader waves
                any_clear |= clear_flag[f0 + v] | clear_flag[f1 + v];
            const bool opaque = __builtin_amdgcn_readfirstlane(any_clear) == 0;
    #pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c == 3 && opaque) {  // wave-uniform
    #pragma unroll
                    for (int e = 0; e < 4; e++)
                        px[e] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(alpha_const, pz[e], 0x0C0C0602u), px[e], 0x05040100u);
                    break;
                }
                f4 d1[2];
    #pragma unroll
                for (int t = 0; t < 2; t++) {
                    const h8 a1 = tr_read2(w0, w1, 2 * (c * kPlaneElems + 16 * t));
                    f4 acc = {plane_bias, plane_bias, plane_bias, plane_bias};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1hi, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1lo, acc, 0, 0, 0);
                    d1[t] = acc;  // 256 * V[x = xt - 8 + 16 t + 4 hgrp + reg][y' = n]
                }
                // accumulator tiles -> the next B operand, split hi + lo: element 4 t + e = d1[t][e]
                uint32_t bh[4], bl[4];
    #pragma unroll
                for (int t = 0; t < 2; t++)
    #pragma unroll
                    for (int e2 = 0; e2 < 2; e2++) {
                        const float v0 = d1[t][2 * e2], v1 = d1[t][2 * e2 + 1];
                        const uint32_t hp = pk16(v0, v1);
                        bh[2 * t + e2] = hp;
                        bl[2 * t + e2] = pk16(residual<0>(hp, v0), residual<1>(hp, v1));
                    }
                const h8 b2hi = __builtin_bit_cast(h8, u32x4{bh[0], bh[1], bh[2], bh[3]});
                const h8 b2lo = __builtin_bit_cast(h8, u32x4{bl[0], bl[1], bl[2], bl[3]});
                f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2hi, b2hi, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2lo, b2hi, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2hi, b2lo, z, 0, 0, 0);
                // z = 65536 * blurred value of pixel (x' = xt + 4 hgrp + reg, y' = yb + n): the integer conversion
                // truncates, the byte wanted is byte 2.  Alpha: the plane held 255 - A, so the value is alpha_top - z.
    #pragma unroll
                for (int e = 0; e < 4; e++) {
                    float zz = z[e];
                    if (c == 3)
                        zz = alpha_top - zz;
                    uint32_t u = (uint32_t)zz;
                    if constexpr (CLAMP)
                        u = min(u, 0x00FFFFFFu);
                    if (c == 0)
                        px[e] = u;
                    else if (c == 1)
                        px[e] = __builtin_amdgcn_perm(u, px[e], 0x0C0C0602u);   // byte 0 = R.byte2, byte 1 = G.byte2
                    else if (c == 2)
                        pz[e] = u;
                    else
                        px[e] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(u, pz[e], 0x0C0C0602u), px[e], 0x05040100u);
                }
            }
            // refill first, store after: the refill waits for this block's loads, which were issued before anything
            // else in the block; waiting after the stores would wait for the stores too (one in-order counter)
            if (more)
                write_slab(b + 2, cur);
            *reinterpret_cast<u32x4*>(out_tile(b) + n * kOutPitch + 16 * wv + 4 * hgrp) = u32x4{px[0], px[1], px[2], px[3]};
            __syncthreads();
        };
        for (int b = 0; b < nb; b += 2) {
            step(b, stA, stB);
            if (b + 1 < nb)
                step(b + 1, stB, stA);
        }
        store_block(nb - 1);
    };
    if (x0 >= kHalo && x0 + kTX + kHalo <= w)
        walk(std::true_type{});
    else
        walk(std::false_type{});
}

}  // namespace

// k <= 17 odd, width a multiple of 4, 16-byte aligned buffers, a separable table whose factor keeps the scaled
// intermediate inside fp16 (256 * 255 * sum(w1) < 65504).
bool gauss_mfma_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef)
{
    (void)h;
    if (coef.k > 17 || coef.k < 3 || !coef.separable || !coef.h_w2d)
        return false;
    if ((w & 3) != 0 || ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15u) != 0)
        return false;
    double s = 0.0;
    for (int j = 0; j < coef.k; j++)
        s += (double)coef.h_w1d[j];
    return 256.0 * 255.0 * s < 65400.0;
}

hipError_t launch_gauss_mfma(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                             const GaussCoef& coef)
{
    const int R = coef.k / 2;
    MWeights W;
    double s = 0.0;
    for (int d = -8; d <= 8; d++) {
        W.w[8 + d] = (d >= -R && d <= R) ? coef.h_w1d[d + R] : 0.0f;
        s += (double)W.w[8 + d];
    }
    MPlan plan;
    plan.nstrips = (w + kTX - 1) / kTX;
    plan.nblocks = (h + 15) / 16;
    // ~15 blocks (240 rows) per band: 16 halo rows per band are 6.7 % extra reads; launches too small to fill the
    // chip (256 CUs x 4 workgroups) are cut finer
    int bpb = 15;
    while (bpb > 2 && (size_t)plan.nstrips * ((plan.nblocks + bpb - 1) / bpb) * nframes < 2048)
        bpb = (bpb + 1) / 2;
    plan.blocks_per_band = bpb;
    plan.nbands = (plan.nblocks + bpb - 1) / bpb;
    const size_t nwork = (size_t)plan.nstrips * plan.nbands * nframes;
    if (nwork > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    plan.nwork = (uint32_t)nwork;
    const bool clamp = !(255.0 * s * s * 1.0001 < 256.0);
    // Alpha of an opaque window: what the CPU path computes for 255 everywhere — its own k*k-term float chain
    // (GaussianBlur.cpp:243-256; capi.hip is built with -ffp-contract=off, so this is a float multiply and a float
    // add per tap) — e.g. 254 for the reference's tables, whose weights sum to just under 1.  The kernel blurs
    // 255 - A and subtracts from alpha_top, chosen so that a zero blur lands on that byte.
    float chain = 0.0f;
    for (int i = 0; i < coef.k * coef.k; i++)
        chain += 255.0f * coef.h_w2d[i];
    const int c255 = (int)(chain < 0.0f ? 0.0f : (chain > 255.0f ? 255.0f : chain));
    double top = 255.0 * s * s * 65536.0;
    const double lo_lim = c255 * 65536.0 + 1.0, hi_lim = (c255 + 1) * 65536.0 - 8.0;
    top = top < lo_lim ? lo_lim : (top > hi_lim ? hi_lim : top);
    const float alpha_top = (float)top;
    // -1024 * (sum over the 17 taps of the fp16 hi + lo parts of 256 w), as the kernel splits them
    double bsum = 0.0;
    for (int d = -8; d <= 8; d++) {
        const float x = W.w[8 + d] * 256.0f;
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        bsum += (double)(float)hi + (double)(float)lo;
    }
    const float plane_bias = (float)(-1024.0 * bsum);
    if (clamp)
        hipLaunchKernelGGL(gauss_mfma_kernel<true>, dim3(plan.nwork), dim3(kThreads), 0, stream, d_in, d_out, w, h, plan, W,
                           alpha_top, plane_bias);
    else
        hipLaunchKernelGGL(gauss_mfma_kernel<false>, dim3(plan.nwork), dim3(kThreads), 0, stream, d_in, d_out, w, h, plan, W,
                           alpha_top, plane_bias);
    return hipGetLastError();
}

}  // namespace mi355
