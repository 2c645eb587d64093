// probe_isa.hip — answers two gfx950 questions the sliding-window kernels depend on, by experiment:
//  (1) v_cvt_pk_u8_f32: rounding mode and saturation of the float -> u8 conversion;
//  (2) DPP wave_shr:1 / wave_shl:1: lane l receives lane l-1 / l+1 across the 16-lane row
//      boundaries of a wave64, and what lanes 0 / 63 receive with bound_ctrl.
// Build: hipcc --offload-arch=gfx950 -O2 probe_isa.hip -o probe_isa ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>

__global__ void k_cvt(const float* in, uint32_t* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
}

__global__ void k_dpp(int* shr, int* shl, float* fm)
{
    int l = threadIdx.x;
    int v = 100 + l;
    shr[l] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, true);   // wave_shr:1
    shl[l] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, true);   // wave_shl:1
    float f = (float)l;
    float nb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x138, 0xF, 0xF, true));
    fm[l] = __builtin_fmaf(nb, 2.0f, 1000.0f);
}

int main()
{
    std::vector<float> h;
    for (int n = -2; n <= 258; n++)
        for (float fr : {0.0f, 0.25f, 0.5f, 0.75f, 0.99f, 0.99999f})
            h.push_back((float)n + fr);
    h.push_back(1e9f); h.push_back(-1e9f); h.push_back(NAN);
    int n = (int)h.size();
    float* din; uint32_t* dout;
    hipMalloc(&din, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
    std::vector<uint32_t> o(n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int trunc_ok = 0, rne_ok = 0, sat_ok = 0;
    for (int i = 0; i < n; i++) {
        float x = h[i];
        if (std::isnan(x)) { printf("cvt_pk_u8(nan) = %u\n", o[i]); continue; }
        float c = fminf(fmaxf(x, 0.f), 255.f);
        trunc_ok += (o[i] == (uint32_t)c);
        rne_ok += (o[i] == (uint32_t)fminf(fmaxf(nearbyintf(x), 0.f), 255.f));
        if (x > 255.f || x < 0.f) sat_ok += (o[i] == (x < 0 ? 0u : 255u));
    }
    printf("cvt_pk_u8_f32: n=%d matches_trunc_sat=%d matches_rne_sat=%d sat_cases_ok=%d\n", n, trunc_ok, rne_ok, sat_ok);
    for (float x : {0.5f, 1.5f, 2.5f, 254.99997f, 255.00003f, 255.5f, 256.0f, -0.5f})
        for (int i = 0; i < n; i++) if (h[i] == x) { printf("  cvt_pk_u8(%g) = %u\n", x, o[i]); break; }

    int *dshr, *dshl; float* dfm;
    hipMalloc(&dshr, 256); hipMalloc(&dshl, 256); hipMalloc(&dfm, 256);
    hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, dshr, dshl, dfm);
    int shr[64], shl[64]; float fm[64];
    hipMemcpy(shr, dshr, 256, hipMemcpyDeviceToHost);
    hipMemcpy(shl, dshl, 256, hipMemcpyDeviceToHost);
    hipMemcpy(fm, dfm, 256, hipMemcpyDeviceToHost);
    int ok_shr = 0, ok_shl = 0, ok_fm = 0;
    for (int l = 1; l < 64; l++) ok_shr += (shr[l] == 100 + l - 1);
    for (int l = 0; l < 63; l++) ok_shl += (shl[l] == 100 + l + 1);
    for (int l = 1; l < 64; l++) ok_fm += (fm[l] == 1000.0f + 2.0f * (l - 1));
    printf("wave_shr: %d/63 lanes got lane-1; lane0=%d | wave_shl: %d/63 lanes got lane+1; lane63=%d | fmac_dpp %d/63, lane0=%g\n",
           ok_shr, shr[0], ok_shl, shl[63], ok_fm, fm[0]);
    return 0;
}
