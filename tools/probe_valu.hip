// probe_valu.hip — issue cost (cycles per wave64 instruction) of the VALU opcodes the filter kernels
// lean on, on gfx950: N independent copies of one instruction in a loop, timed with s_memtime,
// with 1 and with 4 waves resident per SIMD.  Informs which conversions / packs / lane moves are cheap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP16(x) x x x x x x x x x x x x x x x x

#define PROBE(NAME, ASMSTR)                                                                      \
    __global__ __launch_bounds__(1024) void NAME(uint64_t* out, float seed)                         \
    {                                                                                            \
        float a = seed + threadIdx.x, b = seed * 2.f, c = seed * 3.f, d = seed * 0.5f;            \
        float e = a + 1.f, f = b + 1.f, g = c + 1.f, h = d + 1.f;                                  \
        uint32_t u = __builtin_bit_cast(uint32_t, a);                                              \
        uint64_t t0, t1;                                                                         \
        __syncthreads();                                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                \
        for (int it = 0; it < 64; it++) {                                                        \
            asm volatile(REP16(ASMSTR)                                                           \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(u)); \
        }                                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;         \
        if (a + b + c + d + e + f + g + h + (float)u == 12345.678f) out[0] = 0;                    \
    }

// each ASMSTR = 8 independent instructions
PROBE(k_fma, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n")
PROBE(k_fmac, "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %4\n v_fmac_f32 %3, %4, %5\n v_fmac_f32 %4, %5, %6\n v_fmac_f32 %5, %6, %7\n v_fmac_f32 %6, %7, %0\n v_fmac_f32 %7, %0, %1\n")
PROBE(k_fmac_dpp_wave, "v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %2, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %4, %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %5, %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %6, %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %7, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
PROBE(k_fmac_dpp_row, "v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %2, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %4, %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %5, %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %6, %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %7, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
PROBE(k_mov_dpp_wave, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
PROBE(k_cvt_ubyte, "v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n v_cvt_f32_ubyte0 %4, %8\n v_cvt_f32_ubyte1 %5, %8\n v_cvt_f32_ubyte2 %6, %8\n v_cvt_f32_ubyte3 %7, %8\n")
PROBE(k_cvt_u32, "v_cvt_u32_f32 %0, %1\n v_cvt_u32_f32 %1, %2\n v_cvt_u32_f32 %2, %3\n v_cvt_u32_f32 %3, %4\n v_cvt_u32_f32 %4, %5\n v_cvt_u32_f32 %5, %6\n v_cvt_u32_f32 %6, %7\n v_cvt_u32_f32 %7, %0\n")
PROBE(k_cvt_u32_sdwa, "v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %2, %3 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %3, %4 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %4, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %5, %6 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %6, %7 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %7, %0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n")
PROBE(k_cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %1, 0, %0\n v_cvt_pk_u8_f32 %1, %2, 1, %1\n v_cvt_pk_u8_f32 %2, %3, 2, %2\n v_cvt_pk_u8_f32 %3, %4, 3, %3\n v_cvt_pk_u8_f32 %4, %5, 0, %4\n v_cvt_pk_u8_f32 %5, %6, 1, %5\n v_cvt_pk_u8_f32 %6, %7, 2, %6\n v_cvt_pk_u8_f32 %7, %0, 3, %7\n")
PROBE(k_floor, "v_floor_f32 %0, %1\n v_floor_f32 %1, %2\n v_floor_f32 %2, %3\n v_floor_f32 %3, %4\n v_floor_f32 %4, %5\n v_floor_f32 %5, %6\n v_floor_f32 %6, %7\n v_floor_f32 %7, %0\n")
PROBE(k_lshl_or, "v_lshl_or_b32 %0, %1, 8, %2\n v_lshl_or_b32 %1, %2, 8, %3\n v_lshl_or_b32 %2, %3, 8, %4\n v_lshl_or_b32 %3, %4, 8, %5\n v_lshl_or_b32 %4, %5, 8, %6\n v_lshl_or_b32 %5, %6, 8, %7\n v_lshl_or_b32 %6, %7, 8, %0\n v_lshl_or_b32 %7, %0, 8, %1\n")
PROBE(k_perm, "v_perm_b32 %0, %1, %2, %8\n v_perm_b32 %1, %2, %3, %8\n v_perm_b32 %2, %3, %4, %8\n v_perm_b32 %3, %4, %5, %8\n v_perm_b32 %4, %5, %6, %8\n v_perm_b32 %5, %6, %7, %8\n v_perm_b32 %6, %7, %0, %8\n v_perm_b32 %7, %0, %1, %8\n")

PROBE(k_fma_mix_lo, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %2, %3, %1 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %2, %3, %4, %2 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %4, %5, %6, %4 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %5, %6, %7, %5 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %6, %7, %0, %6 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %7, %0, %1, %7 op_sel_hi:[0,1,0]\n")
PROBE(k_fma_mix_hi, "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %2, %3, %1 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %2, %3, %4, %2 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %4, %5, %6, %4 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %5, %6, %7, %5 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %6, %7, %0, %6 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %7, %0, %1, %7 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n")
PROBE(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_cvt_pkrtz_f16_f32 %1, %2, %3\n v_cvt_pkrtz_f16_f32 %2, %3, %4\n v_cvt_pkrtz_f16_f32 %3, %4, %5\n v_cvt_pkrtz_f16_f32 %4, %5, %6\n v_cvt_pkrtz_f16_f32 %5, %6, %7\n v_cvt_pkrtz_f16_f32 %6, %7, %0\n v_cvt_pkrtz_f16_f32 %7, %0, %1\n")

PROBE(k_pk_fma_f16, "v_pk_fma_f16 %0, %1, %2, %0\n v_pk_fma_f16 %1, %2, %3, %1\n v_pk_fma_f16 %2, %3, %4, %2\n v_pk_fma_f16 %3, %4, %5, %3\n v_pk_fma_f16 %4, %5, %6, %4\n v_pk_fma_f16 %5, %6, %7, %5\n v_pk_fma_f16 %6, %7, %0, %6\n v_pk_fma_f16 %7, %0, %1, %7\n")
PROBE(k_pk_add_f16, "v_pk_add_f16 %0, %1, %2\n v_pk_add_f16 %1, %2, %3\n v_pk_add_f16 %2, %3, %4\n v_pk_add_f16 %3, %4, %5\n v_pk_add_f16 %4, %5, %6\n v_pk_add_f16 %5, %6, %7\n v_pk_add_f16 %6, %7, %0\n v_pk_add_f16 %7, %0, %1\n")
PROBE(k_pk_mul_f16, "v_pk_mul_f16 %0, %1, %2\n v_pk_mul_f16 %1, %2, %3\n v_pk_mul_f16 %2, %3, %4\n v_pk_mul_f16 %3, %4, %5\n v_pk_mul_f16 %4, %5, %6\n v_pk_mul_f16 %5, %6, %7\n v_pk_mul_f16 %6, %7, %0\n v_pk_mul_f16 %7, %0, %1\n")
PROBE(k_dot2_f32_f16, "v_dot2_f32_f16 %0, %1, %2, %0\n v_dot2_f32_f16 %1, %2, %3, %1\n v_dot2_f32_f16 %2, %3, %4, %2\n v_dot2_f32_f16 %3, %4, %5, %3\n v_dot2_f32_f16 %4, %5, %6, %4\n v_dot2_f32_f16 %5, %6, %7, %5\n v_dot2_f32_f16 %6, %7, %0, %6\n v_dot2_f32_f16 %7, %0, %1, %7\n")
PROBE(k_dot2c_f32_f16, "v_dot2c_f32_f16 %0, %1, %2\n v_dot2c_f32_f16 %1, %2, %3\n v_dot2c_f32_f16 %2, %3, %4\n v_dot2c_f32_f16 %3, %4, %5\n v_dot2c_f32_f16 %4, %5, %6\n v_dot2c_f32_f16 %5, %6, %7\n v_dot2c_f32_f16 %6, %7, %0\n v_dot2c_f32_f16 %7, %0, %1\n")
PROBE(k_sqrt_f32, "v_sqrt_f32 %0, %1\n v_sqrt_f32 %1, %2\n v_sqrt_f32 %2, %3\n v_sqrt_f32 %3, %4\n v_sqrt_f32 %4, %5\n v_sqrt_f32 %5, %6\n v_sqrt_f32 %6, %7\n v_sqrt_f32 %7, %0\n")
PROBE(k_mul_hi_u24, "v_mul_hi_u32_u24 %0, %1, %2\n v_mul_hi_u32_u24 %1, %2, %3\n v_mul_hi_u32_u24 %2, %3, %4\n v_mul_hi_u32_u24 %3, %4, %5\n v_mul_hi_u32_u24 %4, %5, %6\n v_mul_hi_u32_u24 %5, %6, %7\n v_mul_hi_u32_u24 %6, %7, %0\n v_mul_hi_u32_u24 %7, %0, %1\n")
PROBE(k_mul_u24, "v_mul_u32_u24 %0, %1, %2\n v_mul_u32_u24 %1, %2, %3\n v_mul_u32_u24 %2, %3, %4\n v_mul_u32_u24 %3, %4, %5\n v_mul_u32_u24 %4, %5, %6\n v_mul_u32_u24 %5, %6, %7\n v_mul_u32_u24 %6, %7, %0\n v_mul_u32_u24 %7, %0, %1\n")
PROBE(k_dot4_u8, "v_dot4_u32_u8 %0, %1, %2, %0\n v_dot4_u32_u8 %1, %2, %3, %1\n v_dot4_u32_u8 %2, %3, %4, %2\n v_dot4_u32_u8 %3, %4, %5, %3\n v_dot4_u32_u8 %4, %5, %6, %4\n v_dot4_u32_u8 %5, %6, %7, %5\n v_dot4_u32_u8 %6, %7, %0, %6\n v_dot4_u32_u8 %7, %0, %1, %7\n")
PROBE(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %1\n v_cvt_f32_u32 %1, %2\n v_cvt_f32_u32 %2, %3\n v_cvt_f32_u32 %3, %4\n v_cvt_f32_u32 %4, %5\n v_cvt_f32_u32 %5, %6\n v_cvt_f32_u32 %6, %7\n v_cvt_f32_u32 %7, %0\n")
PROBE(k_min_f32, "v_min_f32 %0, %1, %2\n v_min_f32 %1, %2, %3\n v_min_f32 %2, %3, %4\n v_min_f32 %3, %4, %5\n v_min_f32 %4, %5, %6\n v_min_f32 %5, %6, %7\n v_min_f32 %6, %7, %0\n v_min_f32 %7, %0, %1\n")
PROBE(k_fract_f32, "v_fract_f32 %0, %1\n v_fract_f32 %1, %2\n v_fract_f32 %2, %3\n v_fract_f32 %3, %4\n v_fract_f32 %4, %5\n v_fract_f32 %5, %6\n v_fract_f32 %6, %7\n v_fract_f32 %7, %0\n")
PROBE(k_lshlrev, "v_lshlrev_b32 %0, 2, %1\n v_lshlrev_b32 %1, 2, %2\n v_lshlrev_b32 %2, 2, %3\n v_lshlrev_b32 %3, 2, %4\n v_lshlrev_b32 %4, 2, %5\n v_lshlrev_b32 %5, 2, %6\n v_lshlrev_b32 %6, 2, %7\n v_lshlrev_b32 %7, 2, %0\n")
PROBE(k_lshl_add, "v_lshl_add_u32 %0, %1, 8, %2\n v_lshl_add_u32 %1, %2, 8, %3\n v_lshl_add_u32 %2, %3, 8, %4\n v_lshl_add_u32 %3, %4, 8, %5\n v_lshl_add_u32 %4, %5, 8, %6\n v_lshl_add_u32 %5, %6, 8, %7\n v_lshl_add_u32 %6, %7, 8, %0\n v_lshl_add_u32 %7, %0, 8, %1\n")
PROBE(k_add_f32, "v_add_f32 %0, %1, %2\n v_add_f32 %1, %2, %3\n v_add_f32 %2, %3, %4\n v_add_f32 %3, %4, %5\n v_add_f32 %4, %5, %6\n v_add_f32 %5, %6, %7\n v_add_f32 %6, %7, %0\n v_add_f32 %7, %0, %1\n")
PROBE(k_cvt_f16_f32, "v_cvt_f16_f32 %0, %1\n v_cvt_f16_f32 %1, %2\n v_cvt_f16_f32 %2, %3\n v_cvt_f16_f32 %3, %4\n v_cvt_f16_f32 %4, %5\n v_cvt_f16_f32 %5, %6\n v_cvt_f16_f32 %6, %7\n v_cvt_f16_f32 %7, %0\n")
PROBE(k_pack_b32_f16, "v_pack_b32_f16 %0, %1, %2\n v_pack_b32_f16 %1, %2, %3\n v_pack_b32_f16 %2, %3, %4\n v_pack_b32_f16 %3, %4, %5\n v_pack_b32_f16 %4, %5, %6\n v_pack_b32_f16 %5, %6, %7\n v_pack_b32_f16 %6, %7, %0\n v_pack_b32_f16 %7, %0, %1\n")
PROBE(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0\n v_mad_u32_u24 %1, %2, %3, %1\n v_mad_u32_u24 %2, %3, %4, %2\n v_mad_u32_u24 %3, %4, %5, %3\n v_mad_u32_u24 %4, %5, %6, %4\n v_mad_u32_u24 %5, %6, %7, %5\n v_mad_u32_u24 %6, %7, %0, %6\n v_mad_u32_u24 %7, %0, %1, %7\n")
PROBE(k_pk_add_u16, "v_pk_add_u16 %0, %1, %2\n v_pk_add_u16 %1, %2, %3\n v_pk_add_u16 %2, %3, %4\n v_pk_add_u16 %3, %4, %5\n v_pk_add_u16 %4, %5, %6\n v_pk_add_u16 %5, %6, %7\n v_pk_add_u16 %6, %7, %0\n v_pk_add_u16 %7, %0, %1\n")
PROBE(k_pk_mad_u16, "v_pk_mad_u16 %0, %1, %2, %0\n v_pk_mad_u16 %1, %2, %3, %1\n v_pk_mad_u16 %2, %3, %4, %2\n v_pk_mad_u16 %3, %4, %5, %3\n v_pk_mad_u16 %4, %5, %6, %4\n v_pk_mad_u16 %5, %6, %7, %5\n v_pk_mad_u16 %6, %7, %0, %6\n v_pk_mad_u16 %7, %0, %1, %7\n")
PROBE(k_sqrt_f16, "v_sqrt_f16 %0, %1\n v_sqrt_f16 %1, %2\n v_sqrt_f16 %2, %3\n v_sqrt_f16 %3, %4\n v_sqrt_f16 %4, %5\n v_sqrt_f16 %5, %6\n v_sqrt_f16 %6, %7\n v_sqrt_f16 %7, %0\n")
PROBE(k_rsq_f32, "v_rsq_f32 %0, %1\n v_rsq_f32 %1, %2\n v_rsq_f32 %2, %3\n v_rsq_f32 %3, %4\n v_rsq_f32 %4, %5\n v_rsq_f32 %5, %6\n v_rsq_f32 %6, %7\n v_rsq_f32 %7, %0\n")

template <typename K> void run(const char* name, K kern, uint64_t* d)
{
    for (int waves_per_simd : {1, 2, 4}) {
        int threads = 64 * 4 * waves_per_simd;  // one block per CU -> waves spread over 4 SIMDs
        hipMemset(d, 0, 16 * 8 * 256);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(16 * 256);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int b = 0; b < 256; b++) for (int w = 0; w < threads / 64; w++) v.push_back((double)h[b * 16 + w]);
        std::sort(v.begin(), v.end());
        double med = v[v.size() / 2];
        double ninst = 64.0 * 16 * 8;
        // cycles per instruction per wave; with W waves sharing a SIMD the SIMD issues W instructions in that time
        printf("%-18s waves/SIMD=%d: %6.2f cyc per instr per wave -> %5.2f SIMD-cycles per instr\n", name, waves_per_simd, med / ninst, med / ninst / waves_per_simd);
    }
}

// 64-bit / packed variants use register pairs: separate kernels
__global__ __launch_bounds__(1024) void k_pk(uint64_t* out, float seed, int mode)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a = {seed + threadIdx.x, seed}, b = {seed * 2.f, seed}, c = {seed * 3.f, 1.f}, d = {seed, 2.f};
    double x = seed, y = seed * 1e-9;
    uint64_t t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 64 * 16; it++) {
        if (mode == 0)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else
            asm volatile("v_fma_f64 %0, %0, %1, %1\n v_mul_f64 %1, %1, %0\n v_add_f64 %0, %0, %1\n v_fma_f64 %1, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_mul_f64 %1, %1, %0\n v_add_f64 %0, %0, %1\n v_fma_f64 %1, %0, %1, %1\n" : "+v"(x), "+v"(y));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (a.x + b.x + c.x + d.x + (float)x + (float)y == 12345.678f) out[0] = 0;
}

int main()
{
    uint64_t* d; hipMalloc(&d, 16 * 8 * 256);
    run("v_fma_f32", k_fma, d);
    run("v_fmac_f32", k_fmac, d);
    run("v_fmac_dpp wave_shr", k_fmac_dpp_wave, d);
    run("v_fmac_dpp row_shr", k_fmac_dpp_row, d);
    run("v_mov_dpp wave_shr", k_mov_dpp_wave, d);
    run("v_cvt_f32_ubyteN", k_cvt_ubyte, d);
    run("v_cvt_u32_f32", k_cvt_u32, d);
    run("v_cvt_u32_f32_sdwa", k_cvt_u32_sdwa, d);
    run("v_cvt_pk_u8_f32", k_cvt_pk_u8, d);
    run("v_floor_f32", k_floor, d);
    run("v_lshl_or_b32", k_lshl_or, d);
    run("v_perm_b32", k_perm, d);
    run("v_fma_mix_f32 lo", k_fma_mix_lo, d);
    run("v_fma_mix_f32 hi", k_fma_mix_hi, d);
    run("v_cvt_pkrtz_f16", k_cvt_pkrtz, d);
    run("v_pk_fma_f16", k_pk_fma_f16, d);
    run("v_pk_add_f16", k_pk_add_f16, d);
    run("v_pk_mul_f16", k_pk_mul_f16, d);
    run("v_dot2_f32_f16", k_dot2_f32_f16, d);
    run("v_dot2c_f32_f16", k_dot2c_f32_f16, d);
    run("v_sqrt_f32", k_sqrt_f32, d);
    run("v_mul_hi_u32_u24", k_mul_hi_u24, d);
    run("v_mul_u32_u24", k_mul_u24, d);
    run("v_dot4_u32_u8", k_dot4_u8, d);
    run("v_cvt_f32_u32", k_cvt_f32_u32, d);
    run("v_min_f32", k_min_f32, d);
    run("v_fract_f32", k_fract_f32, d);
    run("v_lshlrev_b32", k_lshlrev, d);
    run("v_lshl_add_u32", k_lshl_add, d);
    run("v_add_f32", k_add_f32, d);
    run("v_cvt_f16_f32", k_cvt_f16_f32, d);
    run("v_pack_b32_f16", k_pack_b32_f16, d);
    run("v_mad_u32_u24", k_mad_u32_u24, d);
    run("v_pk_add_u16", k_pk_add_u16, d);
    run("v_pk_mad_u16", k_pk_mad_u16, d);
    run("v_sqrt_f16", k_sqrt_f16, d);
    run("v_rsq_f32", k_rsq_f32, d);
    for (int mode : {0, 1}) for (int w : {1, 2, 4}) {
        hipMemset(d, 0, 16 * 8 * 256);
        hipLaunchKernelGGL(k_pk, dim3(256), dim3(256 * w), 0, 0, d, 1.0f, mode); hipDeviceSynchronize();
        std::vector<uint64_t> h(16 * 256); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v; for (int b = 0; b < 256; b++) for (int k = 0; k < 4 * w; k++) v.push_back((double)h[b * 16 + k]);
        std::sort(v.begin(), v.end()); double med = v[v.size() / 2] / (64.0 * 16 * 8);
        printf("%-18s waves/SIMD=%d: %6.2f cyc per instr per wave -> %5.2f SIMD-cycles per instr\n", mode == 0 ? "v_pk_fma_f32" : "f64 fma/mul/add", w, med, med / w);
    }
    return 0;
}
