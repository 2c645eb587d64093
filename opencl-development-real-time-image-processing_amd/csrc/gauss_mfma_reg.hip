// gauss_mfma_reg.hip — separable Gaussian blur of RGBA8 frames on the matrix cores, every matrix operand born in the
// register layout the instruction wants (no LDS staging of the input).  Any odd k <= 17 (the reference application's
// own default is k = 17, sigma = 6: include/ProgramHandler.hpp:9), width % 4 == 0.  gfx950 only.  Same arithmetic and
// contract as gauss_mfma.hip (within 1 LSB per channel of the CPU path, src/GaussianBlur/GaussianBlur.cpp:234-261;
// fp16 hi + lo splits of weights and intermediate, everything scaled by 256, see there); a different skeleton.
//
// gauss_mfma.hip stages fp16 planes in LDS (vertical pass first, hardware-transposed reads) and is bound by its
// workgroup-synchronous chain — load, convert, LDS, barrier, transposed reads, matrix chain, output tile, barrier —
// at 4.0-4.36 TB/s whatever is removed from it.  Here the HORIZONTAL pass comes first:
//   pass 1  H[y][x'] = sum_x X[y][x] * Th[x][x']       A = X: lane (m, g) of v_mfma_f32_16x16x32_f16 supplies row m,
//           k = 8g .. 8g+7 — EIGHT CONSECUTIVE PIXELS OF ONE ROW, which is what a 32-byte global load returns.  The
//           lane splits its 8 RGBA pixels into four channel operands (pixel pairs deinterleaved once, then one
//           v_perm_b32 per operand dword: a byte under the exponent byte 0x64 is the fp16 number 1024 + b; the 1024s
//           are cancelled by the accumulator's initial value); B = banded weights, constant registers.
//   pass 2  Z[x'][y'] = sum_y H^T[x'][y] * Tv^T[y][y']  the accumulator of pass 1 has x' on the lanes and 4 rows in the
//           registers: two vertically adjacent 16-row tiles of H (this step's and the previous step's, kept in
//           registers as fp16 hi + lo) ARE the A operand of pass 2 (8 k-slots = rows 4g..4g+3 of either tile; the k
//           order is baked into the constant B operand).  Z leaves row y' on the lanes and 4 consecutive pixels in the
//           registers.
// A WAVE owns a column of 16 output pixels and walks down a band of 16-row tiles with its input two tiles ahead in
// registers; each H tile is computed once and used by two output blocks; the 32-pixel input window of a 16-pixel
// column means every input pixel is loaded by two waves (HBM traffic stays 1.02 x algorithmic — the halo tile of
// each band — the second read hits a cache).  What the memory system needed (same-box ladder, 256 x 4K frames, k = 17):
//   * loads issued in the matrix arrangement (adjacent lanes = adjacent ROWS): every lane its own cache line, 3.35 TB/s
//     -> issued coalesced (lane 4r + p reads piece p of row r) and moved to the matrix arrangement with
//     ds_bpermute_b32: 3.94;
//   * stores from the matrix arrangement are 64-byte pieces over 16 rows: 4.1 TB/s with them, 6.4 without, and the
//     matrix work was not the limit (4.0 with it removed) -> the four waves of a workgroup own adjacent columns, so a
//     block goes through an LDS tile and each wave stores 4 whole rows x 256 contiguous bytes; one barrier per block:
//     5.0 (the LDS-staged kernel on that box: 4.3);
//   * 68 blocks per band instead of 15: 5.2.
//   Tried and dropped: hand-issued loads with a hand-placed s_waitcnt vmcnt(2) (hipcc drains the counter at the top
//   of every step; counting by hand was 3 % SLOWER); one barrier per TWO blocks (32-row output tile: -3.5 %); 5 waves
//   per SIMD (96 VGPRs: weights re-read from LDS, input one tile ahead instead of two: -13 %).
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"
#include "slide_common.hpp"

namespace mi355 {

namespace {

constexpr int kWavesR = 4;
constexpr int kThreadsR = 64 * kWavesR;
constexpr int kStripR = 16 * kWavesR;  // output pixels per workgroup
constexpr int kTapPadR = 40;
constexpr int kOutPitchR = kStripR + 4;  // dwords per row of the output tile: 256 B + 16 B (ds_write_b128 of 8 rows: 32 banks)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct RWeights {
    float w[17];  // w[8 + d], d = -8 .. 8: the separable factor centred, zero beyond the radius
};

struct RPlan {
    int nstrips, nbands, blocks_per_band, nblocks;  // nblocks = ceil(h / 16)
    uint32_t nwork;
};

__device__ __forceinline__ void split16r(float x, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

__device__ __forceinline__ uint32_t pk16r(float a, float b)
{
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}

// v - (float)half of hp, exact, one instruction (v_fma_mix_f32 reads the fp16 operand in place)
template <int HALF>
__device__ __forceinline__ float residual_r(uint32_t hp, float v)
{
    float r;
    if constexpr (HALF == 0)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
    return r;
}

template <bool CLAMP>
__global__ __launch_bounds__(kThreadsR) void gauss_mfma_reg_kernel(const uint8_t* __restrict__ in,
                                                                  uint8_t* __restrict__ out, int w, int h, RPlan plan,
                                                                  RWeights W, float alpha_top, float plane_bias, float hsum,
                                                                  const uint32_t* __restrict__ alpha_cpu)
{
    __shared__ float wtab[2 * kTapPadR];  // wtab[kTapPadR + d] = 256 * w(d), zero beyond the radius
    __shared__ __attribute__((aligned(16))) uint32_t otile[2][16][kOutPitchR];  // two output tiles, 8,704 B
    if (threadIdx.x < 2 * kTapPadR) {
        const int d = (int)threadIdx.x - kTapPadR;
        float v = 0.0f;
#pragma unroll
        for (int t = 0; t < 17; t++)
            v = (d == t - 8) ? W.w[t] * 256.0f : v;
        wtab[threadIdx.x] = v;
    }
    __syncthreads();

    const uint32_t work = xcd_remap(blockIdx.x, plan.nwork);
    const int strip = work % plan.nstrips;
    const int band = (work / plan.nstrips) % plan.nbands;
    const size_t frame = work / ((uint32_t)plan.nstrips * plan.nbands);
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int n = l & 15, g = l >> 4;
    const int x0 = strip * kStripR + 16 * wv;  // this wave's 16 output columns (a wave whose columns lie beyond the
                                               // image keeps step: clamped loads, nothing of it is stored)
    const int blk0 = band * plan.blocks_per_band;
    const int nb = min(plan.blocks_per_band, plan.nblocks - blk0);
    const int yb0 = blk0 * 16;

    // ---- constant B operands ----------------------------------------------------------------------------------
    // pass 1, B[k][x']: k = 8g + j is pixel x0 - 8 + k, output x' = x0 + n               -> tap distance 8g + j - 8 - n
    // pass 2, B[k][y']: k-slot (g, j) is window row rho = (j < 4 ? 4g + j : 16 + 4g + j - 4) of the two stacked H tiles
    //                   (image row yb - 8 + rho), output row y' = yb + n                -> tap distance rho - 8 - n
    h8 b1hi, b1lo, b2hi, b2lo;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        _Float16 hi, lo;
        split16r(wtab[kTapPadR + 8 * g + j - 8 - n], hi, lo);
        b1hi[j] = hi;
        b1lo[j] = lo;
        const int rho = (j < 4) ? 4 * g + j : 16 + 4 * g + (j - 4);
        split16r(wtab[kTapPadR + rho - 8 - n], hi, lo);
        b2hi[j] = hi;
        b2lo[j] = lo;
    }

    const uint32_t row_bytes = (uint32_t)w * 4u;  // a frame is < 2 GiB (gauss_mfma_reg_supported): 32-bit offsets
    const auto fin = uniform_ptr(in + frame * (size_t)row_bytes * h);
    const auto fout = uniform_ptr(out + frame * (size_t)row_bytes * h);
    const uint32_t alpha_const = (uint32_t)alpha_top;  // byte 2 = the alpha of an opaque window

    struct Staged {
        u32x4 lo, hi;  // pixels 0..3 and 4..7 of this lane's piece
    };
    // 256 * H of one 16-row tile as fp16 hi + lo; channel c = elements 2 (c & 1), 2 (c & 1) + 1 (rows 4g, 4g+1 | rows
    // 4g+2, 4g+3) of vector c >> 1.  (Vectors, not arrays: hipcc left a struct of arrays in scratch memory.)
    struct HTile {
        u32x4 hi[2], lo[2];
        int alpha;  // wave-uniform: the alpha value every staged pixel of the tile carries (0..255), or -1 (mixed)
    };

    // Per wave: every pixel of the 32-pixel window inside the image (INTERIOR) or not; two instantiations entered
    // through one scalar branch (see gauss_mfma.hip: as sibling branches inside the loop the two load paths make hipcc
    // drain the memory counter before every load).
    auto walk = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        // Loads are issued in the COALESCED arrangement — lane 4r + p reads piece p of row r, four adjacent lanes one
        // 128-byte line — and moved to the matrix arrangement (lane r + 16 p) by ds_bpermute_b32 when the tile is
        // consumed.
        const int lr = l >> 2, lp = l & 3;
        auto load_tile = [&](int j, Staged& st) __attribute__((always_inline)) {  // tile j = image rows yb0 - 8 + 16 j .. + 15
            const int y = clampi(yb0 - 8 + 16 * j + lr, 0, h - 1);  // clamp-to-edge rows (GaussianBlur.cpp:241)
            const uint32_t row_off = (uint32_t)y * row_bytes;
            const int xp = x0 - 8 + 8 * lp;
            if constexpr (INTERIOR) {
                uint32_t off = row_off + (uint32_t)xp * 4u;  // 32-byte aligned
                lane_offset_here(off);
                st.lo = gload<u32x4>(fin + off);
                st.hi = gload<u32x4>(fin + off + 16);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {  // clamp-to-edge columns (GaussianBlur.cpp:240)
                    st.lo[i] = gload<uint32_t>(fin + (row_off + (uint32_t)clampi(xp + i, 0, w - 1) * 4u));
                    st.hi[i] = gload<uint32_t>(fin + (row_off + (uint32_t)clampi(xp + 4 + i, 0, w - 1) * 4u));
                }
            }
        };
        const int src_in = 4 * (4 * n + g);         // byte address of the lane that loaded row n, piece g
        auto to_matrix_lanes = [&](Staged& st) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                st.lo[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src_in, (int)st.lo[i]);
                st.hi[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src_in, (int)st.hi[i]);
            }
        };
        // pass 1 of one tile.  The alpha operand is 255 - A (0x64FF64FF minus the gathered bytes, no borrow): an opaque
        // window blurs to exactly 0 and its alpha is the host-evaluated constant (launch_gauss_mfma_reg).
        // probe (wave-uniform): whether a tile that is not opaque is examined for ONE other alpha value.  Frames whose alpha
        // varies from pixel to pixel would pay the examination (an OR chain, two compares, a ballot) on every tile for
        // nothing (-1.9 % at k = 17 when every fourth tile was examined), so a wave examines the first tile of its band and
        // every tile that follows a constant one: a band that meets mixed alpha stays on the general path to its end.
        auto pass1 = [&](Staged& st, HTile& t, bool probe) __attribute__((always_inline)) {
            to_matrix_lanes(st);
            const uint32_t a = st.lo[0] & st.lo[1] & st.lo[2] & st.lo[3] & st.hi[0] & st.hi[1] & st.hi[2] & st.hi[3];
            t.alpha = 255;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(a < 0xFF000000u) != 0, 0)) {  // wave-uniform: not opaque — one other value, or mixed?
                t.alpha = -1;
                if (probe) {
                    const uint32_t o = st.lo[0] | st.lo[1] | st.lo[2] | st.lo[3] | st.hi[0] | st.hi[1] | st.hi[2] | st.hi[3];
                    const uint32_t a0 = ((uint32_t)__builtin_amdgcn_readfirstlane((int)a) >> 24) & 0xFFu;
                    const bool differs = ((a >> 24) != a0) | ((o >> 24) != a0);  // AND and OR of the alphas both a0 <=> all are
                    if (__builtin_amdgcn_ballot_w64(differs) == 0)
                        t.alpha = (int)a0;
                }
            }
            // pixel pairs deinterleaved once: rb[i] = (R0, R1, B0, B1), ga[i] = (G0, G1, A0, A1) of pixels 2i, 2i + 1
            uint32_t rb[4], ga[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t p0 = (i < 2) ? st.lo[2 * i] : st.hi[2 * i - 4], p1 = (i < 2) ? st.lo[2 * i + 1] : st.hi[2 * i - 3];
                rb[i] = __builtin_amdgcn_perm(p1, p0, 0x06020400u);
                ga[i] = __builtin_amdgcn_perm(p1, p0, 0x07030501u);
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c == 3 && t.alpha == 255) {  // wave-uniform: the plane 255 - A of an opaque tile is 0
                    t.hi[1][2] = t.hi[1][3] = t.lo[1][2] = t.lo[1][3] = 0u;
                    continue;
                }
                if (c == 3 && __builtin_expect(t.alpha >= 0, 0)) {  // wave-uniform: ... and that of a constant-alpha tile blurs to a constant,
                    // 256 * (255 - A) * sum of the 17 split weights; only a block whose OTHER tile is mixed ever reads it
                    const float hc = (float)(255 - t.alpha) * hsum;
                    const uint32_t hp = pk16r(hc, hc);
                    const uint32_t lp = pk16r(residual_r<0>(hp, hc), residual_r<1>(hp, hc));
                    t.hi[1][2] = t.hi[1][3] = hp;
                    t.lo[1][2] = t.lo[1][3] = lp;
                    continue;
                }
                // channel c of pixel pair i sits in bytes (c >> 1) * 2, (c >> 1) * 2 + 1 of rb[i] (c even) / ga[i] (c odd);
                // the exponent byte comes from the constant: one v_perm_b32 per operand dword
                const uint32_t sel = (c >> 1) ? 0x04030402u : 0x04010400u;  // (S1.b, 0x64, S1.b', 0x64), S0 = 0x64646464
                uint32_t d[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    d[i] = __builtin_amdgcn_perm(0x64646464u, (c & 1) ? ga[i] : rb[i], sel);
                    if (c == 3)
                        d[i] ^= 0x00FF00FFu;  // 255 - A
                }
                const h8 a1 = __builtin_bit_cast(h8, u32x4{d[0], d[1], d[2], d[3]});
                f4 acc = {plane_bias, plane_bias, plane_bias, plane_bias};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1hi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1lo, acc, 0, 0, 0);
                // acc[r] = 256 * H[row 4g + r][x' = n]
#pragma unroll
                for (int e2 = 0; e2 < 2; e2++) {
                    const float v0 = acc[2 * e2], v1 = acc[2 * e2 + 1];
                    const uint32_t hp = pk16r(v0, v1);
                    t.hi[c >> 1][2 * (c & 1) + e2] = hp;
                    t.lo[c >> 1][2 * (c & 1) + e2] = pk16r(residual_r<0>(hp, v0), residual_r<1>(hp, v1));
                }
            }
        };
        // pass 2 of output block b (rows yb0 + 16 b ..) from the H tiles b (up) and b + 1 (dn), and its store
        auto pass2 = [&](const HTile& up, const HTile& dn, int b) __attribute__((always_inline)) {
            // wave-uniform: both tiles carry ONE alpha value -> the block's alpha is the CPU chain's byte for it (table)
            const bool const_alpha = up.alpha >= 0 && up.alpha == dn.alpha;
            uint32_t alpha_out = alpha_const;
            if (__builtin_expect(const_alpha && up.alpha != 255, 0))
                alpha_out = __builtin_amdgcn_readfirstlane(alpha_cpu[up.alpha]);
            uint32_t u[4][4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c == 3 && const_alpha) {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        u[3][e] = alpha_out;
                    continue;
                }
                const int vi = c >> 1, e0 = 2 * (c & 1);
                const h8 ahi = __builtin_bit_cast(h8, u32x4{up.hi[vi][e0], up.hi[vi][e0 + 1], dn.hi[vi][e0], dn.hi[vi][e0 + 1]});
                const h8 alo = __builtin_bit_cast(h8, u32x4{up.lo[vi][e0], up.lo[vi][e0 + 1], dn.lo[vi][e0], dn.lo[vi][e0 + 1]});
                f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, b2hi, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, b2hi, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, b2lo, z, 0, 0, 0);
                // z[r] = 65536 * blurred value of pixel (x0 + 4g + r, yb + n): the integer conversion truncates, the
                // byte wanted is byte 2.  Alpha: the operand held 255 - A, so the value is alpha_top - z.
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float zz = z[e];
                    if (c == 3)
                        zz = alpha_top - zz;
                    uint32_t v = (uint32_t)zz;
                    if constexpr (CLAMP)
                        v = min(v, 0x00FFFFFFu);
                    u[c][e] = v;
                }
            }
            u32x4 px;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t rg = __builtin_amdgcn_perm(u[1][e], u[0][e], 0x0C0C0602u);  // byte 0 = R, byte 1 = G
                const uint32_t ba = __builtin_amdgcn_perm(u[3][e], u[2][e], 0x0C0C0602u);  // byte 0 = B, byte 1 = A
                px[e] = __builtin_amdgcn_perm(ba, rg, 0x05040100u);
            }
            // Output path.  Stored from the matrix arrangement a wave instruction scatters 64-byte pieces over 16 rows:
            // 4.1 TB/s, 6.4 with the stores removed.  The four waves' pieces of a row are adjacent, so the block goes
            // through an LDS tile (rows padded by 16 bytes: the ds_write_b128 of 8 rows lands on 32 different banks) and
            // each wave stores 4 whole rows x 256 contiguous bytes.  Two tiles, ONE barrier per block: tile b & 1 is
            // rewritten at block b + 2, after the barrier of block b + 1, which every wave reaches only after its reads
            // of block b.
            *reinterpret_cast<u32x4*>(&otile[b & 1][n][16 * wv + 4 * g]) = px;
            __syncthreads();
            const int row = 4 * wv + (l >> 4), chunk = l & 15;
            const u32x4 v = *reinterpret_cast<const u32x4*>(&otile[b & 1][row][4 * chunk]);
            const int yo = yb0 + 16 * b + row, xo = strip * kStripR + 4 * chunk;
            if (yo < h && xo < w) {  // w % 4 == 0: the lane's four pixels are inside together
                uint32_t off = (uint32_t)yo * row_bytes + (uint32_t)xo * 4u;
                lane_offset_here(off);
                gstore_nt<u32x4>(fout + off, v);
            }
        };

        Staged stA, stB;
        HTile hA, hB;
        hA.alpha = hB.alpha = -1;
        load_tile(0, stA);
        load_tile(1, stB);
        // step j: tile j arrives in `st` (loaded two steps ago), its registers are refilled with tile j + 2 as soon as
        // the operands are built; output block j - 1 = tiles j - 1 (`up`) and j (`cur`)
        auto step = [&](int j, Staged& st, HTile& cur, const HTile& up) __attribute__((always_inline)) {
            pass1(st, cur, j == 0 || up.alpha >= 0);
            if (j + 2 <= nb)
                load_tile(j + 2, st);
            if (j >= 1)
                pass2(up, cur, j - 1);
        };
        for (int j = 0; j <= nb; j += 2) {
            step(j, stA, hA, hB);
            if (j + 1 <= nb)
                step(j + 1, stB, hB, hA);
        }
    };
    if (x0 >= 8 && x0 + 24 <= w)  // per wave; both instantiations execute one barrier per block
        walk(std::true_type{});
    else
        walk(std::false_type{});
}

}  // namespace

// k <= 17 odd, width a multiple of 4, 16-byte aligned buffers, frames below 2 GiB, a separable table whose factor
// keeps the scaled intermediate inside fp16 (256 * 255 * sum(w1) < 65504)
bool gauss_mfma_reg_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef)
{
    if (coef.k > 17 || coef.k < 3 || !coef.separable || !coef.h_w2d)
        return false;
    if ((w & 3) != 0 || ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15u) != 0)
        return false;
    if ((size_t)w * h * 4 >= 0x7FFFFFFFull)
        return false;
    double s = 0.0;
    for (int j = 0; j < coef.k; j++)
        s += (double)coef.h_w1d[j];
    return 256.0 * 255.0 * s < 65400.0;
}

hipError_t launch_gauss_mfma_reg(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                                 const GaussCoef& coef)
{
    const int R = coef.k / 2;
    RWeights W;
    double s = 0.0;
    for (int d = -8; d <= 8; d++) {
        W.w[8 + d] = (d >= -R && d <= R) ? coef.h_w1d[d + R] : 0.0f;
        s += (double)W.w[8 + d];
    }
    RPlan plan;
    plan.nstrips = (w + kStripR - 1) / kStripR;
    plan.nblocks = (h + 15) / 16;
    // Tall bands: a band pays one extra 16-row tile of reads and pass-1 work.  Same box, 256 x 4K frames, k = 17:
    // 15 blocks per band 4.91 TB/s, 30: 5.05, 45: 5.13, 68: 5.19, 135 (the whole height): 5.20; 64 frames and 1080p
    // peak at 68.  Launches too small to fill the chip are cut finer.
    int bpb = 68;
    if (const char* e = tune_env("MI355_MFMA_BPB"))
        bpb = atoi(e);
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
    // (GaussianBlur.cpp:243-256; built with -ffp-contract=off) — e.g. 254 for the reference's tables.  The kernel blurs
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
    const float hsum = (float)bsum;  // a constant plane v blurs (pass 1) to v * hsum
    if (!coef.d_alpha_cpu)
        return hipErrorInvalidValue;
    if (clamp)
        hipLaunchKernelGGL(gauss_mfma_reg_kernel<true>, dim3(plan.nwork), dim3(kThreadsR), 0, stream, d_in, d_out, w, h,
                           plan, W, alpha_top, plane_bias, hsum, coef.d_alpha_cpu);
    else
        hipLaunchKernelGGL(gauss_mfma_reg_kernel<false>, dim3(plan.nwork), dim3(kThreadsR), 0, stream, d_in, d_out, w, h,
                           plan, W, alpha_top, plane_bias, hsum, coef.d_alpha_cpu);
    return hipGetLastError();
}

}  // namespace mi355
