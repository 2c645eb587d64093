// membench.hip — on-box streaming ceilings, to know what "HBM-bound" can mean for these kernels.
//   copy   : 16 B/lane grid-stride copy (read 1 + write 1), plain / nt loads / nt stores
//   rows   : the sliding-window access shape: each wave reads 1-KiB pieces at a 15,360-B row pitch
//            (4K RGBA), walking down 128 rows, K loads in flight; writes likewise
//   read   : read-only sum (4 B/px side of the traffic)
// Build: hipcc --offload-arch=gfx950 -O3 membench.hip -o membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int LDNT, int STNT, int ILP>
__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n)
{
    size_t stride = (size_t)gridDim.x * 256, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (ILP - 1) * stride < n; i += ILP * stride) {
        u32x4 v[ILP];
#pragma unroll
        for (int u = 0; u < ILP; u++) v[u] = LDNT ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < ILP; u++) { if (STNT) __builtin_nontemporal_store(v[u], &out[i + u * stride]); else out[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) out[i] = in[i];
}

// contiguous-chunk copy: each block owns one contiguous chunk (better DRAM page locality?)
template <int LDNT, int STNT>
__global__ __launch_bounds__(256) void copy_chunk_k(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n, size_t per_block)
{
    size_t b0 = (size_t)blockIdx.x * per_block, b1 = b0 + per_block; if (b1 > n) b1 = n;
    for (size_t i = b0 + threadIdx.x; i < b1; i += 1024) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { size_t j = i + u * 256; if (j < b1) v[u] = LDNT ? __builtin_nontemporal_load(&in[j]) : in[j]; }
#pragma unroll
        for (int u = 0; u < 4; u++) { size_t j = i + u * 256; if (j < b1) { if (STNT) __builtin_nontemporal_store(v[u], &out[j]); else out[j] = v[u]; } }
    }
}

// sliding-window shape: wave = strip of 64 lanes x 16 B, walks `rows` rows at pitch `quads` (in 16-B units)
template <int K, int LDNT, int STNT, int LANES, int HALO>
__global__ __launch_bounds__(256) void rows_k(const u32x4* __restrict__ in, u32x4* __restrict__ out, int quads, int h, int nstrips, int band_rows, int nbands, uint32_t nwork)
{
    uint32_t work = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (work >= nwork) return;
    int lane = threadIdx.x & 63;
    int strip = work % nstrips, band = (work / nstrips) % nbands; size_t frame = work / (nstrips * nbands);
    int q = strip * LANES + lane - (LANES < 64 ? 1 : 0); if (q >= quads) q = quads - 1; if (q < 0) q = 0;
    int qh = strip * LANES + (lane == 0 ? -1 : LANES); if (qh < 0) qh = 0; if (qh >= quads) qh = quads - 1;
    const u32x4* hin = in + frame * (size_t)quads * h + qh;
    const u32x4* fin = in + frame * (size_t)quads * h + q; u32x4* fout = out + frame * (size_t)quads * h + q;
    int y0 = band * band_rows; int n = band_rows; if (y0 + n > h) n = h - y0;
    u32x4 r[K];
#pragma unroll
    for (int u = 0; u < K; u++) { int y = y0 + (u < n ? u : n - 1); r[u] = LDNT ? __builtin_nontemporal_load(fin + (size_t)y * quads) : fin[(size_t)y * quads]; }
    for (int base = 0; base < n; base += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            int i = base + u; u32x4 p = r[u];
            int yn = y0 + (i + K < n ? i + K : n - 1);
            r[u] = LDNT ? __builtin_nontemporal_load(fin + (size_t)yn * quads) : fin[(size_t)yn * quads];
            if (HALO) { if (lane == 0 || lane == 63) { u32x4 hv = hin[(size_t)(y0 + (i < n ? i : n - 1)) * quads]; p.y ^= hv.x; } }
            p.x ^= 0x01010101u;
            if (i < n && (LANES == 64 || (lane >= 1 && lane <= LANES))) { if (STNT) __builtin_nontemporal_store(p, fout + (size_t)(y0 + i) * quads); else fout[(size_t)(y0 + i) * quads] = p; }
        }
    }
}

// read-only stream (optionally with a 1/4-size dword write per 16 B read, the Sobel traffic shape)
template <int WRITE4, int LDNT>
__global__ __launch_bounds__(256) void read_k(const u32x4* __restrict__ in, uint32_t* __restrict__ out, size_t n)
{
    size_t stride = (size_t)gridDim.x * 256, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = LDNT ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint32_t x = v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
            if (WRITE4) __builtin_nontemporal_store(x, &out[i + u * stride]); else acc ^= x;
        }
    }
    if (!WRITE4 && acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <typename F> float timeit(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main(int argc, char** argv)
{
    const int w = 3840, h = 2160, frames = 64; const int quads = w / 4;
    size_t bytes = (size_t)w * h * 4 * frames, n = bytes / 16;
    u32x4 *in, *out; hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMemset(in, 1, bytes); hipMemset(out, 0, bytes);
    auto gbs = [&](float ms) { return 2.0 * bytes / (ms * 1e-3) / 1e9; };
    printf("buffer %.2f GB in + same out\n", bytes / 1e9);
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        float a = timeit([&] { hipLaunchKernelGGL((copy_k<0, 0, 4>), dim3(blocks), dim3(256), 0, 0, in, out, n); }, 10);
        float b = timeit([&] { hipLaunchKernelGGL((copy_k<1, 1, 4>), dim3(blocks), dim3(256), 0, 0, in, out, n); }, 10);
        float c = timeit([&] { hipLaunchKernelGGL((copy_k<0, 1, 4>), dim3(blocks), dim3(256), 0, 0, in, out, n); }, 10);
        float d = timeit([&] { hipLaunchKernelGGL((copy_k<0, 0, 8>), dim3(blocks), dim3(256), 0, 0, in, out, n); }, 10);
        float e = timeit([&] { hipLaunchKernelGGL((copy_k<0, 0, 1>), dim3(blocks), dim3(256), 0, 0, in, out, n); }, 10);
        printf("copy grid-stride blocks=%5d: plain ilp4 %7.1f | nt/nt %7.1f | ld plain st nt %7.1f | plain ilp8 %7.1f | plain ilp1 %7.1f GB/s\n", blocks, gbs(a), gbs(b), gbs(c), gbs(d), gbs(e));
    }
    for (size_t per_block : {(size_t)4096, (size_t)16384, (size_t)65536}) {
        unsigned blocks = (unsigned)((n + per_block - 1) / per_block);
        float a = timeit([&] { hipLaunchKernelGGL((copy_chunk_k<0, 0>), dim3(blocks), dim3(256), 0, 0, in, out, n, per_block); }, 10);
        float b = timeit([&] { hipLaunchKernelGGL((copy_chunk_k<1, 1>), dim3(blocks), dim3(256), 0, 0, in, out, n, per_block); }, 10);
        printf("copy chunked per_block=%6zu x16B (%u blocks): plain %7.1f | nt %7.1f GB/s\n", per_block, blocks, gbs(a), gbs(b));
    }
    {
        int band_rows = 128, nbands = (h + band_rows - 1) / band_rows;
        {
            int nstrips = 16; uint32_t nwork = nstrips * nbands * frames; unsigned blocks = (nwork + 3) / 4;
            float a = timeit([&] { hipLaunchKernelGGL((rows_k<5, 0, 0, 60, 0>), dim3(blocks), dim3(256), 0, 0, in, out, quads, h, nstrips, band_rows, nbands, nwork); }, 10);
            float c = timeit([&] { hipLaunchKernelGGL((rows_k<5, 0, 1, 60, 0>), dim3(blocks), dim3(256), 0, 0, in, out, quads, h, nstrips, band_rows, nbands, nwork); }, 10);
            float b = timeit([&] { hipLaunchKernelGGL((rows_k<5, 1, 1, 60, 0>), dim3(blocks), dim3(256), 0, 0, in, out, quads, h, nstrips, band_rows, nbands, nwork); }, 10);
            printf("rows 60-lane strips + halo lanes: plain %7.1f | st nt %7.1f | nt/nt %7.1f GB/s\n", gbs(a), gbs(c), gbs(b));
        }
        {
            int nstrips = 15; uint32_t nwork = nstrips * nbands * frames; unsigned blocks = (nwork + 3) / 4;
            float a = timeit([&] { hipLaunchKernelGGL((rows_k<5, 0, 0, 64, 0>), dim3(blocks), dim3(256), 0, 0, in, out, quads, h, nstrips, band_rows, nbands, nwork); }, 10);
            float c = timeit([&] { hipLaunchKernelGGL((rows_k<5, 0, 1, 64, 0>), dim3(blocks), dim3(256), 0, 0, in, out, quads, h, nstrips, band_rows, nbands, nwork); }, 10);
            float b = timeit([&] { hipLaunchKernelGGL((rows_k<5, 1, 1, 64, 0>), dim3(blocks), dim3(256), 0, 0, in, out, quads, h, nstrips, band_rows, nbands, nwork); }, 10);
            float d = timeit([&] { hipLaunchKernelGGL((rows_k<5, 0, 1, 64, 1>), dim3(blocks), dim3(256), 0, 0, in, out, quads, h, nstrips, band_rows, nbands, nwork); }, 10);
            float e = timeit([&] { hipLaunchKernelGGL((rows_k<8, 0, 1, 64, 1>), dim3(blocks), dim3(256), 0, 0, in, out, quads, h, nstrips, band_rows, nbands, nwork); }, 10);
            printf("rows 64-lane aligned strips: plain %7.1f | st nt %7.1f | nt/nt %7.1f | st nt + 2-lane halo loads %7.1f | same K8 %7.1f GB/s\n", gbs(a), gbs(c), gbs(b), gbs(d), gbs(e));
        }
    }
    for (int blocks : {2048, 8192, 16384}) {
        float a = timeit([&] { hipLaunchKernelGGL((read_k<0, 0>), dim3(blocks), dim3(256), 0, 0, in, (uint32_t*)out, n); }, 10);
        float b = timeit([&] { hipLaunchKernelGGL((read_k<0, 1>), dim3(blocks), dim3(256), 0, 0, in, (uint32_t*)out, n); }, 10);
        float c = timeit([&] { hipLaunchKernelGGL((read_k<1, 0>), dim3(blocks), dim3(256), 0, 0, in, (uint32_t*)out, n); }, 10);
        float d = timeit([&] { hipLaunchKernelGGL((read_k<1, 1>), dim3(blocks), dim3(256), 0, 0, in, (uint32_t*)out, n); }, 10);
        printf("read-only blocks=%5d: plain %7.1f | nt %7.1f GB/s read ;  read + dword write (5 B/px shape): plain %7.1f | nt %7.1f GB/s total\n", blocks,
               bytes / (a * 1e-3) / 1e9, bytes / (b * 1e-3) / 1e9, 1.25 * bytes / (c * 1e-3) / 1e9, 1.25 * bytes / (d * 1e-3) / 1e9);
    }
    float m = timeit([&] { hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0); }, 5);
    printf("hipMemcpy D2D: %7.1f GB/s\n", gbs(m));
    return 0;
}
