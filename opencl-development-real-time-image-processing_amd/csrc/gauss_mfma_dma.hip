// gauss_mfma_dma.hip — gauss_mfma_reg.hip with its input tiles loaded by LDS-DMA (global_load_lds_dwordx4: memory ->
// LDS, no registers in between) instead of global_load -> VGPR -> ds_bpermute.  Same arithmetic, same bits, same
// decomposition (a wave owns a 16-pixel column and walks down a band of 16-row tiles; four adjacent columns per
// workgroup; horizontal pass first; see gauss_mfma_reg.hip for the matrix-operand layout).  gfx950 only.
//
// RESULT: same bits, same speed as gauss_mfma_reg.hip (-1 .. +2 % over five launch shapes on one box; 2 or 4 stages, 4
// or 5 waves per SIMD: within noise or worse) — the register kernel stays the default (gauss.hip), this one is selected by
// MI355_MFMA_DMA=1 in the tuning build.  The measurement that motivated it was misread: few reads in flight at low
// latency means the memory system is idle because the waves are busy issuing, not because loads are issued too late.
//
// Why it was built.  The register kernel ran at 5.15 TB/s with either memory stream alone good for 6.3-6.6 (profiles/
// r02_mfma_ablations.txt), and the L2's memory-side counters say what it lacks: 15,000 reads in flight over the chip at an
// average latency of 1,220 cycles, against 18,700 at 1,530 cycles for the k = 5 VALU kernel and 33,000 at 2,780 for the
// grayscale strip walk (TCC_EA0_RDREQ_LEVEL / busy cycles, / TCC_EA0_RDREQ: tools/ea_level.sh) — the memory system is not
// saturated, the kernel does not keep enough loads in flight, and every staged tile costs it 8 VGPRs per 16 rows of a
// budget that decides its occupancy (a third staging register set: 132 VGPRs, -5 %).  LDS-DMA stages tiles at no
// register cost: NSTAGE tiles per wave in flight in a wave-private ring (no barrier: a wave's LDS traffic is its own),
// counted with s_waitcnt vmcnt(2 (NSTAGE - 1)) rather than drained.  The lane that loads (row r, 16-byte piece) is chosen
// freely — the DMA writes lane-linear, 16 bytes per lane — so four adjacent lanes still cover one 128-byte line of a row
// and the matrix-arrangement reads (two ds_read_b128 per lane instead of eight ds_bpermute_b32) are bank-conflict-free
// (pieces of rows 4..7 and 12..15 swapped pairwise).  Edge columns (clamped pixels) load through registers into the same
// LDS layout, unprefetched: 2 strips of 60 at 4K.
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"
#include "slide_common.hpp"

namespace mi355 {

namespace {

constexpr int kWavesR = 4;
constexpr int kStagesR = 3;  // input tiles in flight per wave (2 KiB of LDS each)
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
__global__ __launch_bounds__(kThreadsR) void gauss_mfma_dma_kernel(const uint8_t* __restrict__ in,
                                                                  uint8_t* __restrict__ out, int w, int h, RPlan plan,
                                                                  RWeights W, float alpha_top, float plane_bias)
{
    __shared__ float wtab[2 * kTapPadR];  // wtab[kTapPadR + d] = 256 * w(d), zero beyond the radius
    __shared__ __attribute__((aligned(16))) uint32_t otile[2][16][kOutPitchR];  // two output tiles, 8,704 B
    // input tiles, wave-private: stage[wave][slot][half][16 B x 64 lanes], half 0 = pixels 0..3 of a lane's piece, 1 = 4..7
    __shared__ __attribute__((aligned(16))) uint8_t stage[kWavesR][kStagesR][2][1024];
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

    const uint32_t row_bytes = (uint32_t)w * 4u;  // a frame is < 2 GiB (gauss_mfma_dma_supported): 32-bit offsets
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
        bool clear;  // wave-uniform: some staged pixel was not opaque
    };

    // Per wave: every pixel of the 32-pixel window inside the image (INTERIOR) or not; two instantiations entered
    // through one scalar branch (see gauss_mfma.hip: as sibling branches inside the loop the two load paths make hipcc
    // drain the memory counter before every load).
    auto walk = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        // Loader lane 4q + i fetches row q, piece i ^ ((q >> 2) & 1) (pieces = 8 pixels = 32 bytes, as two 16-byte halves);
        // the DMA writes lane-linear, so reader (row n, piece g) finds its halves at 16 * (4 n + (g ^ ((n >> 2) & 1))).
        const int lr = l >> 2, lp = (l & 3) ^ ((l >> 4) & 1);
        const uint32_t lds_wave = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&stage[wv][0][0][0];
        auto issue_tile = [&](int j, int slot) __attribute__((always_inline)) {  // tile j = image rows yb0 - 8 + 16 j .. + 15
            const int y = clampi(yb0 - 8 + 16 * j + lr, 0, h - 1);  // clamp-to-edge rows (GaussianBlur.cpp:241)
            const uint32_t row_off = (uint32_t)y * row_bytes;
            const int xp = x0 - 8 + 8 * lp;
            if constexpr (INTERIOR) {
                const uint32_t off = row_off + (uint32_t)xp * 4u;  // 32-byte aligned
                const uint32_t m_lo = __builtin_amdgcn_readfirstlane(lds_wave + (uint32_t)slot * 2048u);
                const uint32_t m_hi = m_lo + 1024u - 16u;  // the instruction offset (16) moves the LDS address too
                // lgkmcnt(0): the reads of the tile that occupied this slot have returned (write-after-read)
                asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                             "s_mov_b32 m0, %2\n\t"
                             "s_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, %1\n\t"
                             "s_mov_b32 m0, %3\n\t"
                             "s_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:16"
                             :
                             : "v"(off), "s"((uint64_t)fin), "s"(m_lo), "s"(m_hi)
                             : "memory");  // M0 is written here and nowhere else in this kernel (gfx9 LDS instructions do not read it)
            } else {
                u32x4 lo, hi;
#pragma unroll
                for (int i = 0; i < 4; i++) {  // clamp-to-edge columns (GaussianBlur.cpp:240)
                    lo[i] = gload<uint32_t>(fin + (row_off + (uint32_t)clampi(xp + i, 0, w - 1) * 4u));
                    hi[i] = gload<uint32_t>(fin + (row_off + (uint32_t)clampi(xp + 4 + i, 0, w - 1) * 4u));
                }
                *reinterpret_cast<u32x4*>(&stage[wv][slot][0][16 * l]) = lo;
                *reinterpret_cast<u32x4*>(&stage[wv][slot][1][16 * l]) = hi;
            }
        };
        const int pos_in = 16 * (4 * n + (g ^ ((n >> 2) & 1)));  // this lane's operand in the stage: row n, piece g
        // later = tiles issued after tile j (their loads may still be in flight: loads complete in order, a store in
        // between only makes the count more conservative)
        auto fetch_tile = [&](int slot, bool later_full, Staged& st) __attribute__((always_inline)) {
            if constexpr (INTERIOR) {
                if (later_full)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (kStagesR - 1)) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            st.lo = *reinterpret_cast<const u32x4*>(&stage[wv][slot][0][pos_in]);
            st.hi = *reinterpret_cast<const u32x4*>(&stage[wv][slot][1][pos_in]);
        };
        // pass 1 of one tile.  The alpha operand is 255 - A (0x64FF64FF minus the gathered bytes, no borrow): an opaque
        // window blurs to exactly 0 and its alpha is the host-evaluated constant (launch_gauss_mfma_dma).
        auto pass1 = [&](Staged& st, HTile& t) __attribute__((always_inline)) {
            const uint32_t a = st.lo[0] & st.lo[1] & st.lo[2] & st.lo[3] & st.hi[0] & st.hi[1] & st.hi[2] & st.hi[3];
            t.clear = __builtin_amdgcn_ballot_w64(a < 0xFF000000u) != 0;
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
                if (c == 3 && !t.clear) {  // wave-uniform
                    t.hi[1][2] = t.hi[1][3] = t.lo[1][2] = t.lo[1][3] = 0u;
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
            const bool opaque = !up.clear && !dn.clear;  // wave-uniform
            uint32_t u[4][4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c == 3 && opaque) {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        u[3][e] = alpha_const;
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

        HTile hA, hB;
        int slot = 0;  // slot of tile j = j % kStagesR
        if constexpr (INTERIOR) {
#pragma unroll
            for (int j = 0; j < kStagesR; j++)
                if (j <= nb)
                    issue_tile(j, j);
        }
        // step j: tile j has landed in its slot (issued kStagesR steps ago); once its operands are in registers the slot
        // is refilled with tile j + kStagesR; output block j - 1 = tiles j - 1 (`up`) and j (`cur`)
        auto step = [&](int j, HTile& cur, const HTile& up) __attribute__((always_inline)) {
            Staged st;
            if constexpr (!INTERIOR)
                issue_tile(j, slot);  // edge columns: through registers, no prefetch
            fetch_tile(slot, j + kStagesR - 1 <= nb, st);
            pass1(st, cur);
            if constexpr (INTERIOR) {
                if (j + kStagesR <= nb)
                    issue_tile(j + kStagesR, slot);
            }
            slot = (slot + 1 == kStagesR) ? 0 : slot + 1;
            if (j >= 1)
                pass2(up, cur, j - 1);
        };
        for (int j = 0; j <= nb; j += 2) {
            step(j, hA, hB);
            if (j + 1 <= nb)
                step(j + 1, hB, hA);
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
bool gauss_mfma_dma_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef)
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

hipError_t launch_gauss_mfma_dma(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
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
    if (clamp)
        hipLaunchKernelGGL(gauss_mfma_dma_kernel<true>, dim3(plan.nwork), dim3(kThreadsR), 0, stream, d_in, d_out, w, h,
                           plan, W, alpha_top, plane_bias);
    else
        hipLaunchKernelGGL(gauss_mfma_dma_kernel<false>, dim3(plan.nwork), dim3(kThreadsR), 0, stream, d_in, d_out, w, h,
                           plan, W, alpha_top, plane_bias);
    return hipGetLastError();
}

}  // namespace mi355
