// kernels.hpp — host-callable launchers of the gfx950 kernels (one .hip file each).
// All launchers enqueue on `stream` and return the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355 {

// Coefficients of one (k, sigma), device-resident, owned by the context.
//   w2d : k*k floats, the reference's table (RT/src/Controller.cpp:352-372), row-major.
//   w1d : k floats, the separable factor used by the FAST arithmetic:
//         w1d[i] = rowsum_i(w2d) / sqrt(sum(w2d)) evaluated in double, rounded to float.
struct GaussCoef {
    int k;
    const float* d_w2d;
    const float* d_w1d;
    float h_w1d[64];  // host copy, passed by value to the register-resident kernels
    const float* h_w2d;  // host copy of w2d (owned by the context), passed by value to the fused pipeline kernel
    // true when w2d == w1d (x) w1d up to float rounding, with non-negative entries: the FAST (separable) kernels
    // may stand in for the 2-D table.  Externally installed tables that are not (mi355_ctx_set_gauss_weights) are
    // applied tap by tap by the tiled kernels, as the reference kernel applies them (RT/kernel/gaussian_base.cl:23-44).
    bool separable;
    // alpha_tab[A] = (the FAST kernels' output byte for a channel whose whole window holds A) << 24: the canonical
    // separable chains on an all-A window, evaluated on the host with the same float operations
    // (gauss_const_alpha).  The constant-alpha fast path of the sliding-window kernels reads it.
    const uint32_t* d_alpha_tab;
    uint32_t h_alpha_tab[256];
    // alpha_cpu[A] = (the CPU path's byte for a channel whose whole k x k window holds A — its own k*k-term float chain,
    // src/GaussianBlur/GaussianBlur.cpp:243-256) << 16: what the matrix-core kernel stores for constant-alpha windows
    // (its alpha = 255 constant since round 2, now for every A).
    const uint32_t* d_alpha_cpu;
    uint32_t h_alpha_cpu[256];
};

// The FAST arithmetic's result for one channel of a pixel whose k x k window holds the value `a` everywhere:
// v = w[0]*a; v = fma(w[j], a, v) down the column, o = w[0]*v; o = fma(w[t], v, o) along the row,
// uchar(std::clamp(o, 0, 255)) — the kernels' own chain, so the byte is the one they would have computed.
uint32_t gauss_const_alpha(const float* w1d, int k, uint32_t a);

hipError_t launch_gray(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                       int nframes, bool one_channel);

// impl: 0 = choose, 1 = force the LDS-tiled kernel, 2 = the matrix-core kernel wherever it applies.
// d_flags: device scratch of gauss_flag_items(...) uint32 (one per work item of the two-kernel opaque/fallback
// scheme, see gauss_wide.hip); may be null when that returns 0.
size_t gauss_flag_items(const uint8_t* d_in, const uint8_t* d_out, int w, int h, int nframes, const GaussCoef& coef,
                        bool exact, int impl);
hipError_t launch_gauss(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                        int nframes, const GaussCoef& coef, bool exact, int impl, uint32_t* d_flags);

// the two implementations launch_gauss chooses between
hipError_t launch_gauss_tile(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                             int nframes, const GaussCoef& coef, bool exact);

// register-resident sliding-window kernel (k = 3,5,7,9; width % 4 == 0; 16-byte aligned buffers)
bool gauss_slide_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, int k);
size_t gauss_slide_flag_items(int w, int h, int nframes, int k);
hipError_t launch_gauss_slide(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                              int nframes, const GaussCoef& coef, uint32_t* d_flags);

// register-resident kernel for k = 11..17 (2 px per lane, LDS row exchange); width % 2 == 0, 8-byte aligned
bool gauss_wide_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, int k);
size_t gauss_wide_flag_items(int w, int h, int nframes, int k);
hipError_t launch_gauss_wide(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                             const GaussCoef& coef, uint32_t* d_flags);

// EXACT-mode sliding-window kernel (gauss_exact.hip): k in {3,5,7}, width % 4 == 0, 16-byte aligned buffers;
// bit-identical to the CPU path ("exact by exception")
bool gauss_exact_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef);
hipError_t launch_gauss_exact(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                              const GaussCoef& coef);

// matrix-core kernel (gauss_mfma.hip): any odd k <= 17, width % 4 == 0, 16-byte aligned buffers, FAST arithmetic
bool gauss_mfma_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef);
bool gauss_mfma_reg_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef);
// gauss_mfma_dma.hip: the same kernel with LDS-DMA input staging (same conditions as gauss_mfma_reg_supported)
bool gauss_mfma_dma_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef);
hipError_t launch_gauss_mfma_dma(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                                 const GaussCoef& coef);
hipError_t launch_gauss_mfma_reg(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                                 const GaussCoef& coef);
hipError_t launch_gauss_mfma(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                             const GaussCoef& coef);

// impl: 0 = choose (sliding window when width % 4 == 0 and buffers aligned), 1 = force the LDS-tiled kernel
hipError_t launch_sobel(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                        int nframes, int impl);
bool sobel_slide_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h);
hipError_t launch_sobel_slide(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes);

hipError_t launch_pipeline(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                           int nframes, const GaussCoef& coef, bool exact, int impl);
bool pipe_slide_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef);
// 8 pixels per lane (pipe_slide8.hip): k in {3,5}, width % 8 == 0, aligned buffers; the same bits
bool pipe_slide8_supported(const uint8_t* d_in, const uint8_t* d_out, int w, int h, const GaussCoef& coef);
hipError_t launch_pipe_slide8(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                              const GaussCoef& coef);
hipError_t launch_pipe_slide(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                             const GaussCoef& coef);

// image2d_t-mode semantics of the reference (image2d.hip): filter 0 gray / 2 gauss / 3 sobel
hipError_t launch_image2d(hipStream_t stream, int filter, const uint8_t* d_in, uint8_t* d_out, int w, int h, int nframes,
                          int k, const float* d_table);

hipError_t launch_synth(hipStream_t stream, uint8_t* d_out, int w, int h, int nframes,
                        int first_frame, uint32_t seed, int mode);

// packed BGR (3 B/px) -> RGBA (A = 255), cv::cvtColor(BGR2RGBA)
hipError_t launch_bgr_to_rgba(hipStream_t stream, const uint8_t* d_bgr, uint8_t* d_rgba, size_t npx);

// streaming device-to-device copy (non-temporal, 16 B/lane): the on-box "read N + write N bytes" ceiling
hipError_t launch_stream_copy(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, size_t nbytes);

// *d_acc += order-independent checksum of nbytes at d_buf (see include/mi355_imgfilter.h)
hipError_t launch_selftest(hipStream_t stream, unsigned long long* d_acc);
hipError_t launch_checksum(hipStream_t stream, const uint8_t* d_buf, size_t nbytes,
                           uint64_t index_base, unsigned long long* d_acc);

}  // namespace mi355
