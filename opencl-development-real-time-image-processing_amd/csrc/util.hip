// util.hip — synthetic-frame generator and the order-independent checksum (measurement and
// multi-GPU parity helpers; SURVEY.md §8d "Config 2-5", "Parity sampling").
// Both are defined bit-for-bit by their CPU twins in oracle/imgfilter_oracle.c.
#include "common.hpp"
#include "kernels.hpp"

namespace mi355 {

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t synth_hash(uint32_t seed, uint32_t frame, uint32_t y, uint32_t x)
{
    uint32_t hsh = seed ^ (frame * 0x9E3779B1u);
    hsh = fmix32(hsh ^ (y * 0x85EBCA77u));
    hsh = fmix32(hsh ^ (x * 0xC2B2AE3Du));
    return hsh;
}

__global__ __launch_bounds__(kThreads) void synth_kernel(uint32_t* __restrict__ out, int w, int h,
                                                         int nframes, int first_frame, uint32_t seed,
                                                         int mode)
{
    const size_t npx = (size_t)w * h * nframes;
    const size_t stride = (size_t)gridDim.x * kThreads;
    const size_t per_frame = (size_t)w * h;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < npx; i += stride) {
        const uint32_t f = (uint32_t)(i / per_frame);
        const uint32_t rem = (uint32_t)(i - (size_t)f * per_frame);
        const uint32_t y = rem / (uint32_t)w, x = rem - y * (uint32_t)w;
        // mode 2: flat 64 x 64 patches (one colour per patch): frames whose windows are constant almost everywhere
        const uint32_t hsh = (mode == 2) ? synth_hash(seed, (uint32_t)first_frame + f, y >> 6, x >> 6)
                                         : synth_hash(seed, (uint32_t)first_frame + f, y, x);
        uint32_t px;
        if (mode == 0 || mode == 2) {
            px = (hsh & 0x00FFFFFFu) | 0xFF000000u;
        } else if (mode == 3) {  // gray noise, r = g = b: every pixel sits on the luminance's ambiguous case S = 1000 v
            px = (hsh & 0xFFu) * 0x00010101u | 0xFF000000u;
        } else {
            const int gx = (int)((x * 255u) / (uint32_t)(w > 1 ? w - 1 : 1));
            const int gy = (int)((y * 255u) / (uint32_t)(h > 1 ? h - 1 : 1));
            const int r = clampi(gx + (int)(hsh & 15u) - 8, 0, 255);
            const int g = clampi(gy + (int)((hsh >> 8) & 15u) - 8, 0, 255);
            const int b = clampi(((gx + gy) >> 1) + (int)((hsh >> 16) & 15u) - 8, 0, 255);
            px = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16) | 0xFF000000u;
        }
        out[i] = px;
    }
}

__global__ __launch_bounds__(kThreads) void checksum_kernel(const uint8_t* __restrict__ buf,
                                                            size_t nbytes, uint64_t index_base,
                                                            unsigned long long* __restrict__ acc)
{
    const size_t nwords = nbytes >> 2;
    const size_t stride = (size_t)gridDim.x * kThreads;
    uint64_t sum = 0;
    const bool aligned = (reinterpret_cast<uintptr_t>(buf) & 3u) == 0;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nwords; i += stride) {
        uint32_t wv;
        if (aligned)
            wv = reinterpret_cast<const uint32_t*>(buf)[i];
        else
            wv = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) |
                 ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
        sum += fmix64((uint64_t)wv + ((index_base + i) << 32));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (nbytes & 3u)) {
        uint32_t wv = 0;
        for (size_t b = 0; b < (nbytes & 3u); b++)
            wv |= (uint32_t)buf[4 * nwords + b] << (8 * b);
        sum += fmix64((uint64_t)wv + ((index_base + nwords) << 32));
    }
    // wave reduction, one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        sum += __shfl_down(sum, off, 64);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(acc, (unsigned long long)sum);
}

// cv::cvtColor(BGR2RGBA): out = (R, G, B, 255) from packed (B, G, R).  4 pixels = 3 dwords in, 4 dwords out
// per thread when the pixel count allows (12-byte loads, 16-byte stores); 7 B/px of traffic.
__global__ __launch_bounds__(kThreads) void bgr_to_rgba_kernel(const uint8_t* __restrict__ in,
                                                               uint32_t* __restrict__ out, size_t npx, int aligned)
{
    const size_t nquads = aligned ? npx / 4 : 0;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nquads; i += stride) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(in) + 3 * i;
        const uint32_t a = p[0], b = p[1], c = p[2];  // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        u32x4 o;
        o.x = 0xFF000000u | ((a >> 16) & 0xFFu) | (a & 0xFF00u) | ((a & 0xFFu) << 16);
        o.y = 0xFF000000u | ((b >> 8) & 0xFFu) | ((b & 0xFFu) << 8) | ((a >> 24) << 16);
        o.z = 0xFF000000u | (c & 0xFFu) | ((b >> 24) << 8) | (((b >> 16) & 0xFFu) << 16);
        o.w = 0xFF000000u | (c >> 24) | (((c >> 16) & 0xFFu) << 8) | (((c >> 8) & 0xFFu) << 16);
        reinterpret_cast<u32x4*>(out)[i] = o;
    }
    for (size_t i = nquads * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; i < npx; i += stride) {
        const uint8_t* p = in + 3 * i;
        out[i] = 0xFF000000u | (uint32_t)p[2] | ((uint32_t)p[1] << 8) | ((uint32_t)p[0] << 16);
    }
}

// Exhaustive device-side check of the two fast arithmetic forms against their exact definitions:
//   low  32 bits of *acc: colours (all 2^24) where luma_px_fast / luma_quad_fast != the FP64 formula
//   high 32 bits of *acc: (|gx|, |gy|) pairs (all 1021^2) where the v_sqrt_f32 + v_cvt_pk_u8_f32 magnitude
//                         != the integer form sobel_mag_u8, in any of the four byte positions
__global__ __launch_bounds__(kThreads) void selftest_kernel(unsigned long long* __restrict__ acc)
{
    static_assert(kThreads == 256, "fill_gray_lut wants 256 threads");
    __shared__ uint8_t gray_lut[256];
    fill_gray_lut(gray_lut);
    __syncthreads();
    const uint32_t stride = gridDim.x * kThreads;
    uint32_t bad_luma = 0, bad_mag = 0;
    bool hint = false;  // the table forms carry the gray-row hint as the kernels do
    for (uint32_t c = (blockIdx.x * kThreads + threadIdx.x) * 4u; c < (1u << 24); c += stride * 4u) {
        const u32x4 p = {c | 0xFF000000u, (c + 1u) | 0x7F000000u, c + 2u, (c + 3u) | 0x01000000u};
        float g[4], gi[4], gt[4], git[4];
        luma_quad_fast(p, g);
        luma_quad_int(p, gi);
        luma_quad_fast(p, gt, gray_lut, hint);  // ambiguous gray pixels by table (common.hpp)
        luma_quad_int(p, git, gray_lut);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t want = luma_px(p[j]);
            bad_luma += ((uint32_t)g[j] != want) + ((uint32_t)luma_px_fast(p[j]) != want) + ((uint32_t)gi[j] != want) +
                        ((uint32_t)gt[j] != want) + ((uint32_t)git[j] != want);
        }
    }
    // all-gray wave rows: the first call raises the hint (every lane ambiguous), the second takes gray_row()
    for (uint32_t r = 0; r < 4; r++) {
        const uint32_t v0 = (threadIdx.x * 4u + r * 61u) & 0xFFu;
        const u32x4 p = {v0 * 0x010101u | 0xFF000000u, ((v0 + 1u) & 0xFFu) * 0x010101u, ((v0 + 2u) & 0xFFu) * 0x010101u | 0x80000000u,
                         ((v0 + 3u) & 0xFFu) * 0x010101u};
        float ga[4], gb[4];
        bool h2 = false;
        luma_quad_fast(p, ga, gray_lut, h2);
        const bool raised = h2;
        luma_quad_fast(p, gb, gray_lut, h2);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t want = luma_px(p[j]);
            bad_luma += ((uint32_t)ga[j] != want) + ((uint32_t)gb[j] != want);
        }
        bad_luma += raised ? 0u : 1u;
    }
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < 1021u * 1021u; i += stride) {
        const int gx = (int)(i / 1021u), gy = (int)(i % 1021u);
        const uint32_t want = sobel_mag_u8(gx, gy);
        const float fx[4] = {(float)gx, (float)-gx, (float)gy, (float)-gy};
        const float fy[4] = {(float)gy, (float)gy, (float)-gx, (float)gx};
        bad_mag += (sobel_mag_quad(fx, fy) != want * 0x01010101u) + (sobel_mag_fast(fx[0], fy[0]) != want);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bad_luma += __shfl_down(bad_luma, off, 64);
        bad_mag += __shfl_down(bad_mag, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && (bad_luma | bad_mag))
        atomicAdd(acc, (unsigned long long)bad_luma | ((unsigned long long)bad_mag << 32));
}

// Streaming copy, 16 B per lane, non-temporal loads and stores, 4 accesses in flight per lane: the on-box
// ceiling for "read N bytes + write N bytes" that bench.py reports beside the filter kernels (tools/membench.hip
// measured this form at 6.2-6.4 TB/s with >= 8k blocks; hipMemcpy D2D reaches only 4.4-4.7).
__global__ __launch_bounds__(kThreads) void stream_copy_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out,
                                                               size_t n)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
            v[u] = __builtin_nontemporal_load(&in[i + u * stride]);
#pragma unroll
        for (int u = 0; u < 4; u++)
            __builtin_nontemporal_store(v[u], &out[i + u * stride]);
    }
    for (; i < n; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(&in[i]), &out[i]);
}

__global__ __launch_bounds__(kThreads) void byte_copy_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                             size_t n)
{
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n)
        out[i] = in[i];
}

}  // namespace

hipError_t launch_selftest(hipStream_t stream, unsigned long long* d_acc)
{
    hipLaunchKernelGGL(selftest_kernel, dim3(256 * 8), dim3(kThreads), 0, stream, d_acc);
    return hipGetLastError();
}

hipError_t launch_bgr_to_rgba(hipStream_t stream, const uint8_t* d_bgr, uint8_t* d_rgba, size_t npx)
{
    size_t blocks = (npx / 4 + kThreads - 1) / kThreads;
    if (blocks > (1u << 20))
        blocks = 1u << 20;
    if (blocks < 1)
        blocks = 1;
    const int aligned = ((reinterpret_cast<uintptr_t>(d_bgr) & 3u) == 0) && ((reinterpret_cast<uintptr_t>(d_rgba) & 15u) == 0);
    hipLaunchKernelGGL(bgr_to_rgba_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, d_bgr,
                       reinterpret_cast<uint32_t*>(d_rgba), npx, aligned);
    return hipGetLastError();
}

hipError_t launch_synth(hipStream_t stream, uint8_t* d_out, int w, int h, int nframes, int first_frame,
                        uint32_t seed, int mode)
{
    const size_t npx = (size_t)w * h * nframes;
    size_t blocks = (npx + kThreads - 1) / kThreads;
    if (blocks > 256 * 8)
        blocks = 256 * 8;
    if (blocks < 1)
        blocks = 1;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream,
                       reinterpret_cast<uint32_t*>(d_out), w, h, nframes, first_frame, seed, mode);
    return hipGetLastError();
}

hipError_t launch_stream_copy(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, size_t nbytes)
{
    if (nbytes == 0)
        return hipSuccess;
    const bool aligned = ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst)) & 15u) == 0;
    const size_t nvec = aligned ? nbytes / 16 : 0;
    if (nvec) {
        size_t blocks = (nvec + (size_t)kThreads * 4 - 1) / ((size_t)kThreads * 4);
        if (blocks > (1u << 16))
            blocks = 1u << 16;
        hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream,
                           reinterpret_cast<const u32x4*>(d_src), reinterpret_cast<u32x4*>(d_dst), nvec);
    }
    const size_t done = nvec * 16, rest = nbytes - done;
    if (rest) {
        const size_t blocks = (rest + kThreads - 1) / kThreads;
        if (blocks > 0x7FFFFFFFull)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(byte_copy_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, d_src + done,
                           d_dst + done, rest);
    }
    return hipGetLastError();
}

hipError_t launch_checksum(hipStream_t stream, const uint8_t* d_buf, size_t nbytes, uint64_t index_base,
                           unsigned long long* d_acc)
{
    size_t blocks = ((nbytes >> 2) + kThreads - 1) / kThreads;
    if (blocks > 256 * 8)
        blocks = 256 * 8;
    if (blocks < 1)
        blocks = 1;
    hipLaunchKernelGGL(checksum_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, d_buf, nbytes,
                       index_base, d_acc);
    return hipGetLastError();
}

}  // namespace mi355
