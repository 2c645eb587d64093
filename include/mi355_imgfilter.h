/*
 * mi355_imgfilter.h — C-ABI of libmi355_imgfilter.so, the MI355X (gfx950) image-filter hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Every entry point
 * names the reference interface it replaces (paths relative to the reference repository root;
 * RT/ = src/RealtimeImageProcessing/).  The C++ classes in host/ (Controller, ProgramHandler)
 * forward to these functions; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Pixel layout everywhere: interleaved 8-bit RGBA, row-major, tightly packed (stride = 4*width),
 * as produced by cv::cvtColor(BGR2RGBA) at RT/src/ProgramHandler.cpp:127.
 *
 * Conventions: every function returns MI355_OK (0) or a negative MI355_ERR_* code; nothing throws,
 * nothing calls exit().  A context is bound to one GPU and one HIP stream and must be used from one
 * host thread at a time (the reference Controller is single-threaded too: include/Controller.hpp:50).
 * There is no CPU fallback: without a usable gfx950 device mi355_ctx_create fails.
 */
#ifndef MI355_IMGFILTER_H
#define MI355_IMGFILTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK 0
#define MI355_ERR_BAD_ARG (-1)     /* null pointer, non-positive size, even/oversized kernel_size */
#define MI355_ERR_HIP (-2)         /* a HIP runtime call failed; see mi355_last_hip_error */
#define MI355_ERR_NO_DEVICE (-3)   /* no GPU / device index out of range */
#define MI355_ERR_UNSUPPORTED (-4) /* request outside what the kernels implement */
#define MI355_ERR_NOMEM (-5)

/* Largest Gaussian kernel_size accepted (odd).  The reference's own default is 17
 * (include/ProgramHandler.hpp:9). */
#define MI355_MAX_GAUSS_K 63

/* Gaussian arithmetic selection (mi355_ctx_set_gauss_mode):
 *   FAST  — separable, FMA, vertical pass then horizontal pass; within +-1 LSB per channel of the
 *           reference CPU path (src/GaussianBlur/GaussianBlur.cpp:234-261).  Default.
 *   EXACT — the reference CPU path's own arithmetic (k*k taps, ky outer / kx inner, separate float
 *           multiply and add, truncation): bit-identical output, several times slower. */
#define MI355_GAUSS_FAST 0
#define MI355_GAUSS_EXACT 1

typedef struct mi355_ctx mi355_ctx;

/* ---- context -------------------------------------------------------------------------------
 * Replaces the OpenCL bootstrap of Controller (RT/src/Controller.cpp:13-197: GetPlatforms,
 * GetDevices, CreateContext, CreateCommandQueue w/ CL_QUEUE_PROFILING_ENABLE :118, CreateProgram,
 * CreateKernel).  One context = {HIP device, stream, timing events, pooled device/pinned staging
 * buffers, cached Gaussian coefficient table}.  The reference allocates and frees device buffers on
 * every call (:234-244, :515-516); the pool here grows to the largest frame seen and is reused. */
int mi355_device_count(int* count);
int mi355_ctx_create(int device, mi355_ctx** out);
/* Same, but launches on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)
 * so the caller's stream-ordered allocator and events see the kernels. */
int mi355_ctx_create_on_stream(int device, void* hip_stream, mi355_ctx** out);
int mi355_ctx_destroy(mi355_ctx* ctx);
int mi355_ctx_device_name(mi355_ctx* ctx, char* buf, size_t buflen);
int mi355_sync(mi355_ctx* ctx);
int mi355_last_hip_error(mi355_ctx* ctx);
const char* mi355_strerror(int code);

int mi355_ctx_set_gauss_mode(mi355_ctx* ctx, int mode);

/* Kernel selection (a tuning / test knob):
 *   AUTO — the fastest kernel that applies.  Gaussian, FAST mode: the register-resident sliding-window kernels for
 *          k in {3,5} (any width); the matrix-core kernel for odd 7 <= k <= 17 when width % 4 == 0, width >= 64,
 *          the launch has >= 2^16 pixels and the device buffers are 16-byte aligned, the VALU kernels otherwise
 *          (k in {7,9} any width; k in {11,13,15,17} with an even width); the LDS-tiled kernel for everything else.
 *          EXACT mode: the exact-by-exception sliding kernel for k in {3,5,7} and width % 4 == 0, the tiled kernel
 *          otherwise.  Pipeline: k in {3,5,7}, w >= 4, h >= 2 sliding (8 pixels per lane for k = 5 launches of
 *          >= 10^9 pixels with width % 8 == 0), tiled otherwise.
 *   TILE — always the LDS-tiled kernels.
 *   VALU — as AUTO but never the matrix cores.  TILE and VALU give identical bits. */
#define MI355_IMPL_AUTO 0
#define MI355_IMPL_TILE 1
/*   MFMA — the Gaussian on the matrix cores (csrc/gauss_mfma_reg.hip) wherever it applies (FAST mode, odd k <= 17,
 *          width % 4 == 0, 16-byte aligned device buffers); AUTO elsewhere.  Within 1 LSB of the CPU path like every
 *          FAST kernel, but not bit-identical to the VALU kernels (other rounding). */
#define MI355_IMPL_MFMA 2
#define MI355_IMPL_VALU 3
int mi355_ctx_set_impl(mi355_ctx* ctx, int impl);

/* Input pixel format of the HOST-buffer calls (mi355_*_rgba8, mi355_filter_batched, mi355_filter_stream):
 *   RGBA — 4 bytes per pixel, what the reference hands its Controller after cv::cvtColor(BGR2RGBA)
 *          (RT/src/ProgramHandler.cpp:127).  Default.
 *   BGR  — 3 bytes per pixel, tightly packed, as cv::imread / the camera deliver frames
 *          (RT/src/ProgramHandler.cpp:116, RT/RealtimeImageProcessing.cpp:325) and as the reference's CPU
 *          grayscale reads them (src/Grayscale/grayscale.cpp:234-236).  The frame crosses PCIe as 3 B/px and a
 *          device kernel performs the BGR2RGBA expansion (A = 255) ahead of the filter: 25 % less H2D traffic
 *          on a path that is PCIe-bound, and no host-side cvtColor.  Results equal the RGBA path on the
 *          converted frame.  Device-resident calls always take RGBA; mi355_bgr_to_rgba8_dev is the converter. */
#define MI355_INPUT_RGBA 0
#define MI355_INPUT_BGR 1
int mi355_ctx_set_input_format(mi355_ctx* ctx, int format);
int mi355_bgr_to_rgba8_dev(mi355_ctx* ctx, const void* d_bgr, void* d_rgba, int w, int h, int nframes);

/* ---- Gaussian coefficients -----------------------------------------------------------------
 * mi355_gauss_weights replaces Controller::_GenerateGaussianKernelBuffers
 * (RT/src/Controller.cpp:352-372 = src/GaussianBlur/src/Controller.cpp:342-362): k*k floats,
 * row-major [(y+k/2)*k + (x+k/2)], same promotion chain (float argument, exp in double, divide by
 * the double 2*M_PI*sigma^2, store float, float running sum, divide by it).  Pure host function.
 * The reference regenerates and re-uploads the table on every frame (:667,:674, leaking the
 * cl_mem); here the table is cached per (k, sigma) inside the context.
 *
 * mi355_ctx_set_gauss_weights installs an externally supplied k*k table (multi-GPU mode: rank 0
 * generates it, RCCL broadcasts it, every rank installs the same bytes) for the given (k, sigma)
 * key; later calls with that (k, sigma) use it instead of regenerating.  The table is applied as given: when it
 * is not w (x) w for one non-negative vector w up to float rounding (a non-separable or asymmetric table, negative
 * lobes), the separable FAST kernels do not apply and every call with that key runs the tap-by-tap (EXACT
 * arithmetic) kernel, as the reference kernel would (RT/kernel/gaussian_base.cl:23-44).  Non-finite entries are
 * rejected (MI355_ERR_BAD_ARG).  Installed tables are never evicted (at most 64 per context, one more is
 * MI355_ERR_BAD_ARG); of the tables the library generates itself a context keeps the 16 most recently used.  The
 * first call with a new (k, sigma) key — and mi355_ctx_set_gauss_weights itself — synchronises the context's stream
 * (allocation + blocking upload of the table); calls with a cached key do not. */
int mi355_gauss_weights(int k, float sigma, float* out_k2);
int mi355_ctx_set_gauss_weights(mi355_ctx* ctx, int k, float sigma, const float* w_k2);

/* ---- host-buffer calls: what Controller::PerformCL* forwards to ----------------------------
 * Each call = H2D, one kernel, D2H, synchronous on return, exactly like the reference
 * (RT/src/Controller.cpp:429-519, :521-613, :615-744).  prof_ns (may be NULL) receives the six
 * timestamps the reference appends to profiling_events via _profileEvent (:66-74, :471,:489,:513):
 * write-start, write-end, kernel-start, kernel-end, read-start, read-end, in ns on one clock.
 *
 * mi355_gray_rgba8     replaces PerformCLImageGrayscaling (buffer mode): out = w*h*4 bytes,
 *                      (g,g,g,255) per pixel.  g follows the CPU path src/Grayscale/grayscale.cpp:237
 *                      (double arithmetic, truncation) bit-exactly — not the float kernel
 *                      RT/kernel/grayscale_base.cl:14.
 * mi355_gray1_rgba8    same gray value, one byte per pixel (the CPU path's output shape,
 *                      grayscale.cpp:216).
 * mi355_gauss_rgba8    replaces PerformCLGaussianBlur: out = w*h*4 bytes; clamp-to-edge taps, all
 *                      four channels, truncation; semantics of the CPU path GaussianBlur.cpp:234-261
 *                      (no division by the accumulated weight, unlike gaussian_base.cl:48).
 * mi355_sobel_rgba8    replaces PerformCLImageEdgeDetection: out = w*h bytes.  Semantics of the CPU
 *                      path src/EdgeDetection/EdgeDetection.cpp:219-240 applied to
 *                      gray = grayscale.cpp:237 of each pixel: 3x3 correlation, BORDER_REFLECT_101,
 *                      float magnitude, round-half-even, saturate — every pixel written (the OpenCL
 *                      kernel edge_base.cl:12 leaves the border unwritten).
 * mi355_pipeline_rgba8 fused gray -> Gaussian -> Sobel, defined as the exact composition of the three
 *                      calls above (SURVEY.md §8a "a-pipe"): out = w*h bytes. */
int mi355_gray_rgba8(mi355_ctx* ctx, const uint8_t* rgba, uint8_t* out_rgba, int w, int h,
                     uint64_t prof_ns[6]);
int mi355_gray1_rgba8(mi355_ctx* ctx, const uint8_t* rgba, uint8_t* out_gray, int w, int h,
                      uint64_t prof_ns[6]);
int mi355_gauss_rgba8(mi355_ctx* ctx, const uint8_t* rgba, uint8_t* out_rgba, int w, int h, int k,
                      float sigma, uint64_t prof_ns[6]);
int mi355_sobel_rgba8(mi355_ctx* ctx, const uint8_t* rgba, uint8_t* out_gray, int w, int h,
                      uint64_t prof_ns[6]);
int mi355_pipeline_rgba8(mi355_ctx* ctx, const uint8_t* rgba, uint8_t* out_gray, int w, int h, int k,
                         float sigma, uint64_t prof_ns[6]);

/* ---- image2d_t mode ---------------------------------------------------------
 * What the reference computes when image support is NOT bypassed (BYPASS_IMAGE_SUPPORT = false; no shipped
 * application does that): its *_images.cl kernels and the host code around them, which differ from the buffer path
 * in WHAT they compute —
 *   MI355_FILTER_GRAY   RT/kernel/grayscale_images.cl:15-22 + Controller.cpp:256-258,76-85: fp32 luminance of the
 *                       normalised texel, R/FLOAT image, host truncation of f * 255       -> out = w*h bytes
 *   MI355_FILTER_GAUSS  RT/kernel/gaussian_images.cl:1-36 + Controller.cpp:374-403: taps outside the image read the
 *                       border colour 0 (CLK_ADDRESS_CLAMP), no renormalisation, the image-mode table (its last row
 *                       and column are 0), round-to-nearest into UNORM_INT8              -> out = w*h*4 bytes
 *   MI355_FILTER_SOBEL  RT/kernel/edge_images.cl:3-47: RED channel only, interior pixels only (border 0), magnitude
 *                       clamped to [0, 1], host truncation of f * 255                     -> out = w*h bytes
 * Same call shape and profiling contract as the buffer-mode calls.  mi355_gauss_weights_image2d is the image-mode
 * table generator (Controller::_GenerateGaussianKernelImage2D), bit-identical to the reference's. */
int mi355_image2d_rgba8(mi355_ctx* ctx, int filter, const uint8_t* rgba, uint8_t* out, int w, int h, int k,
                        float sigma, uint64_t prof_ns[6]);
int mi355_gauss_weights_image2d(int k, float sigma, float* out_k2);

/* Batched host-buffer form: nframes frames back to back in `rgba` and in `out`; one H2D, one launch
 * (grid.z-style frame index), one D2H.  filter: MI355_FILTER_*. */
#define MI355_FILTER_GRAY 0     /* RGBA -> RGBA (g,g,g,255) */
#define MI355_FILTER_GRAY1 1    /* RGBA -> 1 byte            */
#define MI355_FILTER_GAUSS 2    /* RGBA -> RGBA              */
#define MI355_FILTER_SOBEL 3    /* RGBA -> 1 byte            */
#define MI355_FILTER_PIPELINE 4 /* RGBA -> 1 byte            */
int mi355_filter_batched(mi355_ctx* ctx, int filter, const uint8_t* rgba, uint8_t* out, int w, int h,
                         int nframes, int k, float sigma, uint64_t prof_ns[6]);
/* bytes per output pixel of a filter (4 or 1), or MI355_ERR_BAD_ARG */
int mi355_filter_out_bpp(int filter);

/* Streamed host-buffer form: nframes frames in `rgba` -> `out`, both on the host, moved
 * in chunks of chunk_frames (0 = choose) through three device slots with three stages in flight on three HIP
 * streams — H2D of chunk i+1, the kernel of chunk i, D2H of chunk i-1 — instead of the reference's
 * write / wait / kernel / wait / read / wait per frame (RT/src/Controller.cpp:470,488,512).  Results are
 * identical to mi355_filter_batched.  Any host memory works; the copies run at PCIe DMA rate and overlap in
 * both directions only when `rgba` and `out` are pinned (mi355_host_alloc).  elapsed_ms (may be NULL)
 * receives the wall time of the whole call, PCIe included. */
int mi355_filter_stream(mi355_ctx* ctx, int filter, const uint8_t* rgba, uint8_t* out, int w, int h,
                        int nframes, int chunk_frames, int k, float sigma, double* elapsed_ms);
/* Pinned (page-locked) host memory for frames: hipHostMalloc / hipHostFree. */
int mi355_host_alloc(mi355_ctx* ctx, size_t nbytes, void** h_ptr);
int mi355_host_free(mi355_ctx* ctx, void* h_ptr);

/* ---- device-resident calls -----------------------------------------------------------------
 * d_in / d_out are device pointers on the context's GPU holding nframes tightly packed frames;
 * the call enqueues the kernel(s) on the context's stream and returns without synchronising.
 * [d_in, d_in + 4*w*h*nframes) and the output range must not overlap: the stencil filters (Gaussian, Sobel,
 * pipeline) read neighbouring rows and halo pixels that another wave may already have overwritten, and the
 * grayscale kernels, although pointwise, are compiled with non-aliasing (__restrict__) pointers and non-temporal
 * accesses — so an in-place or overlapping call is rejected with MI355_ERR_BAD_ARG, for every filter, instead of
 * returning corrupted pixels.  (The reference never aliases them either: two clCreateBuffer objects per call.)
 * These are what a caller that already owns device memory (torch, a capture pipeline) binds, and
 * what the roofline measurement times (no PCIe in the timed region).
 * Stream capture: after the first call with a given (k, sigma) and frame size — which installs the weight table and
 * sizes the scratch buffers, synchronising the stream — these calls allocate nothing and wait for nothing, so they may
 * be issued while the context's stream is being captured into a hipGraph and replayed later
 * (tests/test_gpu_configs.py: test_device_resident_calls_can_be_captured_into_a_hip_graph; tools/graph_probe.py). */
int mi355_gray_rgba8_dev(mi355_ctx* ctx, const void* d_in, void* d_out, int w, int h, int nframes);
int mi355_gray1_rgba8_dev(mi355_ctx* ctx, const void* d_in, void* d_out, int w, int h, int nframes);
int mi355_gauss_rgba8_dev(mi355_ctx* ctx, const void* d_in, void* d_out, int w, int h, int nframes,
                          int k, float sigma);
int mi355_sobel_rgba8_dev(mi355_ctx* ctx, const void* d_in, void* d_out, int w, int h, int nframes);
int mi355_pipeline_rgba8_dev(mi355_ctx* ctx, const void* d_in, void* d_out, int w, int h, int nframes,
                             int k, float sigma);
int mi355_filter_dev(mi355_ctx* ctx, int filter, const void* d_in, void* d_out, int w, int h,
                     int nframes, int k, float sigma);

/* Synthetic frames: px = hash(seed, first_frame + f, y, x), A = 255; mode 1 = smooth
 * gradient + 4-bit noise, mode 2 = flat 64 x 64 patches (constant windows: the content that sends the
 * exact-by-exception kernels down their exception path), mode 3 = gray noise (r = g = b: every pixel on the
 * luminance's ambiguous case).  Bit-identical to oracle_synth_rgba. */
int mi355_synth_rgba8_dev(mi355_ctx* ctx, void* d_out, int w, int h, int nframes, int first_frame,
                          uint32_t seed, int mode);
/* Order-independent 64-bit checksum of nbytes at d_buf (sum of per-word hashes mod 2^64; word i is
 * hashed with index index_base + i), written to *out after synchronising the stream.  Sharding a
 * batch over GPUs and adding the per-rank values gives the single-GPU value. */
int mi355_checksum_dev(mi355_ctx* ctx, const void* d_buf, size_t nbytes, uint64_t index_base,
                       uint64_t* out);

/* Streaming device-to-device copy of nbytes (non-overlapping) on the context's stream, no synchronisation: 16 B per
 * lane, non-temporal loads and stores.  A measurement helper: bench.py times it on the same buffers as the filter
 * to report the box's own "read N + write N bytes" ceiling beside the 8 TB/s spec peak (hipMemcpy D2D is ~25 %
 * slower than this form on MI355X and is not a ceiling).  Replaces nothing in the reference. */
int mi355_stream_copy_dev(mi355_ctx* ctx, void* d_dst, const void* d_src, size_t nbytes);

/* Device self-test of the two fast arithmetic forms the kernels use in place of the reference's FP64 luminance
 * (src/Grayscale/grayscale.cpp:237) and sqrt + round + saturate (src/EdgeDetection/EdgeDetection.cpp:236-240):
 * every one of the 2^24 colours and every (|gx|, |gy|) <= 1020 pair is compared with the exact definition on
 * the GPU the context is bound to.  Both counts are 0 on a correct device / build.  ~1 ms. */
int mi355_selftest(mi355_ctx* ctx, uint32_t* bad_luma, uint32_t* bad_mag);

/* ---- device memory + timing helpers for hosts that do not bring their own ------------------- */
int mi355_dev_alloc(mi355_ctx* ctx, size_t nbytes, void** d_ptr);
/* Frame pools with a placement search.  On MI355X the PHYSICAL placement of a streaming kernel's input and output
 * buffers decides up to 8 % of its rate (the same command reads 5.5 or 6.0 TB/s from one process to the next), and
 * an allocation cannot be steered, only re-drawn.  This call allocates the input pool
 * (nframes x w x h RGBA, filled with 0xFF) and up to `tries` (<= 16) candidate output pools side by side — all
 * alive at once, hence all in different places; fewer if memory runs short — runs a few launches of `filter` on
 * each, keeps the fastest and frees the others.  tries <= 1: plain allocation, no probing.  probe_ms (optional,
 * `tries` floats): average launch time per candidate, -1 for candidates that were not allocated.  Synchronises
 * the context's stream.  Release with mi355_pool_free.  (Replaces nothing in the reference, which allocates and
 * frees its buffers per frame, RT/src/Controller.cpp:646-652,742-744.) */
int mi355_pool_alloc(mi355_ctx* ctx, int filter, int w, int h, int nframes, int k, float sigma, int tries,
                     void** d_in, void** d_out, float* probe_ms);
int mi355_pool_free(mi355_ctx* ctx, void* d_in, void* d_out);
int mi355_dev_free(mi355_ctx* ctx, void* d_ptr);
int mi355_copy_h2d(mi355_ctx* ctx, void* d_dst, const void* h_src, size_t nbytes);
int mi355_copy_d2h(mi355_ctx* ctx, void* h_dst, const void* d_src, size_t nbytes);
/* hipEvent pair on the context's stream: begin records, end records + synchronises and returns the
 * elapsed milliseconds.  This is the HIP-event timing the roofline figure is computed from. */
int mi355_timer_begin(mi355_ctx* ctx);
int mi355_timer_end(mi355_ctx* ctx, float* elapsed_ms);

/* ---- device group: one batch sharded over several GPUs (SURVEY.md §8b "Threading", §8e) -----------------------------
 * The reference owns one queue on one device (RT/src/ProgramHandler.cpp:108) and has nothing to match; this is the
 * product form of what BASELINE.json's north_star calls the batched-frame mode: independent frames, contiguous
 * frame ranges per GPU, no pixel ever crosses GPUs, no collective.
 *
 * A group = one mi355_ctx + one host worker thread per member.  `devices` lists the HIP device ordinal of each
 * member (NULL = 0 .. ndev-1); an ordinal may repeat (two members then share that GPU on two streams — how the
 * one-GPU tests run it).  Member m of an n-member group owns the frames mi355_group_shard(m, n, nframes) names:
 * base = nframes / n, the first nframes % n members take one extra, ranges contiguous in member order (the same
 * split bench.py's shard_range makes over ranks).
 *
 *   mi355_group_filter_batched  host buffers, nframes frames back to back in `rgba` / `out`: every member streams its
 *                               range through its own mi355_filter_stream (H2D / kernel / D2H overlapped), all
 *                               members at once; returns when all are done.  elapsed_ms (may be NULL): wall time.
 *   mi355_group_filter_dev      device-resident: d_in[m] / d_out[m] are pointers on member m's GPU holding
 *                               nframes[m] frames (0 = member idle); every member launches on its own stream and
 *                               synchronises it; returns when all are done.
 * Gaussian coefficients: a table for (k, sigma) is generated ONCE on the host (mi355_gauss_weights) and the same bytes
 * are installed on every member before the first call that uses the key; mi355_group_set_gauss_weights installs a
 * caller's table on every member.  Mode / impl / input-format setters apply to every member.
 * Status: MI355_OK, or the first failing member's code in member order (mi355_group_member_status gives each
 * member's code of the last call).  One call at a time per group (calls are serialised internally). */
typedef struct mi355_group mi355_group;
int mi355_group_create(int ndev, const int* devices, mi355_group** out);
int mi355_group_destroy(mi355_group* g);
int mi355_group_size(mi355_group* g, int* ndev);
int mi355_group_member_ctx(mi355_group* g, int member, mi355_ctx** ctx); /* borrowed; for alloc / copies / checksums */
int mi355_group_member_status(mi355_group* g, int member);
int mi355_group_shard(int member, int nmembers, int nframes, int* first_frame, int* count); /* pure host function */
int mi355_group_set_gauss_mode(mi355_group* g, int mode);
int mi355_group_set_impl(mi355_group* g, int impl);
int mi355_group_set_input_format(mi355_group* g, int format);
int mi355_group_set_gauss_weights(mi355_group* g, int k, float sigma, const float* w_k2);
int mi355_group_filter_batched(mi355_group* g, int filter, const uint8_t* rgba, uint8_t* out, int w, int h,
                               int nframes, int k, float sigma, double* elapsed_ms);
int mi355_group_filter_dev(mi355_group* g, int filter, const void* const* d_in, void* const* d_out, int w, int h,
                           const int* nframes, int k, float sigma);

/* Library build info: "gfx950;<git-or-date>" */
const char* mi355_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355_IMGFILTER_H */
