// probe_mfma_small.hip — do the small matrix instructions (v_mfma_f32_4x4x1_16b_f32: per lane 4 multiply-adds
// acc[i] += A(lane 4b+i) * B(own lane); v_mfma_i32_4x4x4_16b_i8: per lane 4 byte dot products) run BESIDE the
// vector ALU on gfx950?  Timed with s_memtime: fp32 FMAs alone, matrix instructions alone, both interleaved in one
// wave's stream, with 1, 2 and 4 waves per SIMD.  If the mix costs max(a, b) rather than a + b, scatter-form vertical
// stencil passes can move off the VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

#define FMA8 "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n"
#define CVT8 "v_cvt_u32_f32 %0, %1\n v_cvt_u32_f32 %1, %2\n v_cvt_u32_f32 %2, %3\n v_cvt_u32_f32 %3, %4\n v_cvt_u32_f32 %4, %5\n v_cvt_u32_f32 %5, %6\n v_cvt_u32_f32 %6, %7\n v_cvt_u32_f32 %7, %0\n"
#define MF(acc) "v_mfma_f32_4x4x1_16b_f32 " acc ", %12, %13, " acc "\n"
#define MI(acc) "v_mfma_i32_4x4x4_16b_i8 " acc ", %12, %13, " acc "\n"

// MODE: 0 = 8 FMA; 1 = 4 MFMA f32; 2 = 8 FMA + 4 MFMA f32 interleaved; 3 = 8 FMA + 2 MFMA f32; 4 = 4 MFMA i8;
// 5 = 8 FMA + 4 MFMA i8; 6 = 8 cvt (3-cycle class) ; 7 = 8 cvt + 4 MFMA f32; 8 = 8 FMA + 8 MFMA f32; 9 = 8 MFMA f32
template <int MODE>
__global__ __launch_bounds__(1024) void probe(uint64_t* out, float seed)
{
    float a = seed + threadIdx.x, b = seed * 2.f, c = seed * 3.f, d = seed * 0.5f;
    float e = a + 1.f, f = b + 1.f, g = c + 1.f, h = d + 1.f;
    f4 m0 = {a, b, c, d}, m1 = {b, c, d, a}, m2 = {c, d, a, b}, m3 = {d, a, b, c};
    float wa = seed * 0.25f, wb = seed * 0.125f;
    uint64_t t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 256; it++) {
#define OPS "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(wa), "v"(wb)
        if constexpr (MODE == 0) asm volatile(FMA8 FMA8 : OPS);
        if constexpr (MODE == 1) asm volatile(MF("%8") MF("%9") MF("%10") MF("%11") MF("%8") MF("%9") MF("%10") MF("%11") : OPS);
        if constexpr (MODE == 2) asm volatile(MF("%8") "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n" MF("%9") "v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n" MF("%10") "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n" MF("%11") "v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n"
                                              MF("%8") "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n" MF("%9") "v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n" MF("%10") "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n" MF("%11") "v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n" : OPS);
        if constexpr (MODE == 3) asm volatile(MF("%8") FMA8 MF("%9") FMA8 : OPS);
        if constexpr (MODE == 4) asm volatile(MI("%8") MI("%9") MI("%10") MI("%11") MI("%8") MI("%9") MI("%10") MI("%11") : OPS);
        if constexpr (MODE == 5) asm volatile(MI("%8") "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n" MI("%9") "v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n" MI("%10") "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n" MI("%11") "v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n"
                                              MI("%8") "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n" MI("%9") "v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n" MI("%10") "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n" MI("%11") "v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n" : OPS);
        if constexpr (MODE == 6) asm volatile(CVT8 CVT8 : OPS);
        if constexpr (MODE == 7) asm volatile(MF("%8") "v_cvt_u32_f32 %0, %1\n v_cvt_u32_f32 %1, %2\n" MF("%9") "v_cvt_u32_f32 %2, %3\n v_cvt_u32_f32 %3, %4\n" MF("%10") "v_cvt_u32_f32 %4, %5\n v_cvt_u32_f32 %5, %6\n" MF("%11") "v_cvt_u32_f32 %6, %7\n v_cvt_u32_f32 %7, %0\n"
                                              MF("%8") "v_cvt_u32_f32 %0, %1\n v_cvt_u32_f32 %1, %2\n" MF("%9") "v_cvt_u32_f32 %2, %3\n v_cvt_u32_f32 %3, %4\n" MF("%10") "v_cvt_u32_f32 %4, %5\n v_cvt_u32_f32 %5, %6\n" MF("%11") "v_cvt_u32_f32 %6, %7\n v_cvt_u32_f32 %7, %0\n" : OPS);
#undef OPS
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (a + b + c + d + e + f + g + h + m0.x + m1.y + m2.z + m3.w == 12345.678f) out[0] = 0;
}

template <int MODE> void run(const char* name, uint64_t* d)
{
    for (int waves_per_simd : {1, 2, 4}) {
        int threads = 64 * 4 * waves_per_simd;
        hipMemset(d, 0, 16 * 8 * 256);
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(16 * 256);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int b = 0; b < 256; b++) for (int w = 0; w < threads / 64; w++) v.push_back((double)h[b * 16 + w]);
        std::sort(v.begin(), v.end());
        const double med = v[v.size() / 2];
        // ticks per loop trip per wave, and per trip of the SIMD (W waves share it)
        printf("%-34s waves/SIMD=%d: %8.2f ticks per trip per wave -> %7.2f per trip of the SIMD\n", name, waves_per_simd, med / 256.0, med / 256.0 / waves_per_simd);
    }
}

int main()
{
    uint64_t* d; hipMalloc(&d, 16 * 8 * 256);
    run<0>("16 v_fma_f32", d);
    run<6>("16 v_cvt_u32_f32", d);
    run<1>("8 mfma_f32_4x4x1", d);
    run<4>("8 mfma_i32_4x4x4_i8", d);
    run<2>("16 fma + 8 mfma_f32 interleaved", d);
    run<3>("16 fma + 2 mfma_f32", d);
    run<5>("16 fma + 8 mfma_i8 interleaved", d);
    run<7>("16 cvt + 8 mfma_f32 interleaved", d);
    return 0;
}
