// capi.hip — the C-ABI of libmi355_imgfilter.so (include/mi355_imgfilter.h): context, pooled
// buffers, coefficient cache, HIP-event profiling, and the host-buffer / device-resident entry
// points.  No CPU fallback anywhere: every filter call ends in a gfx950 kernel launch or an error.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>
#include <vector>

#include "../../include/mi355_imgfilter.h"
#include "kernels.hpp"

using namespace mi355;

namespace {

struct CoefEntry {
    int k;
    uint32_t sigma_bits;
    GaussCoef coef;
    float* d_buf;  // [k*k w2d][k w1d][256 constant-alpha bytes << 24, as uint32][256 CPU-chain bytes << 16]
    std::vector<float> h_tab;  // host copy of the same block (its heap storage stays put when entries move)
    uint64_t last_use;
    bool installed;  // came through mi355_ctx_set_gauss_weights: never evicted (it cannot be regenerated)
};

// Distinct GENERATED (k, sigma) tables kept per context; the least recently used one goes.  Tables a caller installed
// (mi355_ctx_set_gauss_weights: a box filter, a non-separable table, the bytes rank 0 broadcast) are kept outside
// this cap and are never evicted — a miss would silently regenerate the default Gaussian under the same key; at most
// kMaxInstalledCoefs of them per context, one more is MI355_ERR_BAD_ARG.
constexpr size_t kMaxCoefEntries = 16;
constexpr size_t kMaxInstalledCoefs = 64;

}  // namespace

struct mi355_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[6] = {};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    void* d_in = nullptr;
    size_t d_in_cap = 0;
    void* d_out = nullptr;
    size_t d_out_cap = 0;
    unsigned long long* d_acc = nullptr;
    float* d_img_table = nullptr;  // image2d-mode Gaussian table (k*k floats, MI355_MAX_GAUSS_K^2 capacity)
    void* d_flags = nullptr;  // per-work-item flags of the two-kernel Gaussian (gauss_wide.hip), pooled
    size_t d_flags_cap = 0;
    // streamed path: copy streams, per-slot events and device slots (created on first use)
    static constexpr int kSlots = 3;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_h2d[kSlots] = {}, ev_k[kSlots] = {}, ev_d2h[kSlots] = {};
    void* slot_in[kSlots] = {};
    void* slot_out[kSlots] = {};
    size_t slot_in_cap = 0, slot_out_cap = 0;
    int input_format = MI355_INPUT_RGBA;
    void* d_raw = nullptr;  // BGR staging of the host-buffer calls
    size_t d_raw_cap = 0;
    void* slot_raw[3] = {};
    size_t slot_raw_cap = 0;
    int gauss_mode = MI355_GAUSS_FAST;
    int impl = MI355_IMPL_AUTO;
    int last_hip = 0;
    char name[256] = {};
    std::vector<CoefEntry> coefs;
    uint64_t coef_clock = 0;
    std::vector<void*> pinned;  // mi355_host_alloc blocks still outstanding (freed at destroy)
};

namespace {

#define HIP_TRY(ctx, expr)                     \
    do {                                       \
        hipError_t e__ = (expr);               \
        if (e__ != hipSuccess) {               \
            if (ctx)                           \
                (ctx)->last_hip = (int)e__;    \
            return MI355_ERR_HIP;              \
        }                                      \
    } while (0)

uint64_t now_ns()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

uint32_t fbits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

bool valid_k(int k) { return k >= 1 && k <= MI355_MAX_GAUSS_K && (k & 1) == 1; }
bool valid_sigma(float s) { return std::isfinite(s) && s > 0.0f; }

// Controller::_GenerateGaussianKernelBuffers, RT/src/Controller.cpp:352-372.
// `exp` there binds to ::exp(double); the float argument is widened.
void gen_weights(int k, float sigma, float* out)
{
    const int half = k / 2;
    float sum = 0.0f;
    for (int y = -half; y <= half; y++) {
        for (int x = -half; x <= half; x++) {
            const float arg = -(x * x + y * y) / (2 * sigma * sigma);
            const float value = (float)(std::exp((double)arg) / (2 * M_PI * sigma * sigma));
            out[(y + half) * k + (x + half)] = value;
            sum += value;
        }
    }
    for (int i = 0; i < k * k; i++)
        out[i] /= sum;
}

// Controller::_GenerateGaussianKernelImage2D, RT/src/Controller.cpp:374-403: the image-mode twin.  Its loops run
// x, y < half (one short), so the last row and the last column of the table stay 0, and the normalising sum covers
// only the (k-1) x (k-1) entries that were written; x is the OUTER index.  Same promotion chain as gen_weights.
void gen_weights_image2d(int k, float sigma, float* out)
{
    const int half = k / 2;
    for (int i = 0; i < k * k; i++)
        out[i] = 0.0f;
    float sum = 0.0f;
    for (int x = -half; x < half; x++) {
        for (int y = -half; y < half; y++) {
            const float arg = -((x * x + y * y) / (2 * sigma * sigma));
            const float value = (float)(std::exp((double)arg) / (2 * M_PI * sigma * sigma));
            out[(x + half) * k + (y + half)] = value;
            sum += value;
        }
    }
    for (int i = 0; i < k * k; i++)
        out[i] /= sum;
}

// Separable factor of the table for the FAST arithmetic: rowsum / sqrt(total), in double.  Returns whether the
// factor really reproduces the table: every entry non-negative, total > 0 and |w2d[i][j] - w1d[i] * w1d[j]| within
// float rounding of the largest entry (the reference's own tables deviate by <= 1.8e-7 of it for every odd k <= 63;
// the bound used is 1e-6).  A table that fails — non-separable, asymmetric, negative lobes — must not go through
// the separable kernels: the caller routes it to the tap-by-tap tiled kernel.
bool separable_factor(int k, const float* w2d, float* w1d)
{
    double tot = 0.0, wmax = 0.0;
    bool nonneg = true;
    std::vector<double> rs(k, 0.0);
    for (int i = 0; i < k; i++) {
        for (int j = 0; j < k; j++) {
            const double v = (double)w2d[i * k + j];
            rs[i] += v;
            nonneg = nonneg && v >= 0.0;
            wmax = std::fmax(wmax, std::fabs(v));
        }
        tot += rs[i];
    }
    if (!nonneg || !(tot > 0.0)) {
        for (int i = 0; i < k; i++)
            w1d[i] = 0.0f;
        return false;
    }
    const double s = std::sqrt(tot);
    for (int i = 0; i < k; i++)
        w1d[i] = (float)(rs[i] / s);
    double dev = 0.0;
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++)
            dev = std::fmax(dev, std::fabs((double)w2d[i * k + j] - (double)w1d[i] * (double)w1d[j]));
    return dev <= 1.0e-6 * wmax;
}

// First use of a (k, sigma) key — and a call that evicts — synchronises the context's stream and makes blocking
// allocations / copies; every later call with a cached key does not (include/mi355_imgfilter.h says so).
int install_coef(mi355_ctx* ctx, int k, float sigma, const float* w2d, bool installed, const GaussCoef** out)
{
    CoefEntry* slot = nullptr;
    size_t n_generated = 0, n_installed = 0;
    for (auto& e : ctx->coefs) {
        if (e.k == k && e.sigma_bits == fbits(sigma))
            slot = &e;
        (e.installed ? n_installed : n_generated)++;
    }
    if (!slot) {
        if (installed && n_installed >= kMaxInstalledCoefs)
            return MI355_ERR_BAD_ARG;
        if (!installed && n_generated >= kMaxCoefEntries) {
            // evict the least recently used GENERATED table; a kernel still in flight may be reading it
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            size_t lru = ctx->coefs.size();
            for (size_t i = 0; i < ctx->coefs.size(); i++)
                if (!ctx->coefs[i].installed && (lru == ctx->coefs.size() || ctx->coefs[i].last_use < ctx->coefs[lru].last_use))
                    lru = i;
            (void)hipFree(ctx->coefs[lru].d_buf);
            ctx->coefs.erase(ctx->coefs.begin() + (long)lru);
        }
        CoefEntry e{};
        e.k = k;
        e.sigma_bits = fbits(sigma);
        HIP_TRY(ctx, hipMalloc((void**)&e.d_buf, sizeof(float) * (size_t)(k * k + k + 512)));
        ctx->coefs.push_back(e);
        slot = &ctx->coefs.back();
    }
    slot->last_use = ++ctx->coef_clock;
    slot->installed = slot->installed || installed;
    std::vector<float> host((size_t)k * k + k + 512);
    std::memcpy(host.data(), w2d, sizeof(float) * (size_t)k * k);
    const bool separable = separable_factor(k, w2d, host.data() + (size_t)k * k);
    uint32_t alpha_tab[256];
    for (uint32_t a = 0; a < 256; a++)
        alpha_tab[a] = gauss_const_alpha(host.data() + (size_t)k * k, k, a) << 24;
    std::memcpy(host.data() + (size_t)k * k + k, alpha_tab, sizeof(alpha_tab));
    uint32_t alpha_cpu[256];
    for (uint32_t a = 0; a < 256; a++) {
        // the CPU path's own chain over a window that is `a` everywhere (one float multiply, one float add per tap;
        // this translation unit is built with -ffp-contract=off), clamped and truncated
        float chain = 0.0f;
        for (int i = 0; i < k * k; i++)
            chain += (float)a * w2d[i];
        chain = chain < 0.0f ? 0.0f : (chain > 255.0f ? 255.0f : chain);
        alpha_cpu[a] = (uint32_t)chain << 16;
    }
    std::memcpy(host.data() + (size_t)k * k + k + 256, alpha_cpu, sizeof(alpha_cpu));
    // blocking copy from pageable memory: the table is live on the device when this returns
    const hipError_t ce = hipMemcpy(slot->d_buf, host.data(), sizeof(float) * host.size(), hipMemcpyHostToDevice);
    if (ce != hipSuccess) {
        // never leave a half-installed table behind: a later call with this (k, sigma) would find it
        ctx->last_hip = (int)ce;
        (void)hipFree(slot->d_buf);
        ctx->coefs.erase(ctx->coefs.begin() + (slot - ctx->coefs.data()));
        return MI355_ERR_HIP;
    }
    slot->coef.k = k;
    slot->coef.separable = separable;
    slot->coef.d_w2d = slot->d_buf;
    slot->coef.d_w1d = slot->d_buf + (size_t)k * k;
    slot->coef.d_alpha_tab = reinterpret_cast<const uint32_t*>(slot->d_buf + (size_t)k * k + k);
    std::memcpy(slot->coef.h_alpha_tab, alpha_tab, sizeof(alpha_tab));
    slot->coef.d_alpha_cpu = reinterpret_cast<const uint32_t*>(slot->d_buf + (size_t)k * k + k + 256);
    std::memcpy(slot->coef.h_alpha_cpu, alpha_cpu, sizeof(alpha_cpu));
    std::memset(slot->coef.h_w1d, 0, sizeof(slot->coef.h_w1d));
    std::memcpy(slot->coef.h_w1d, host.data() + (size_t)k * k, sizeof(float) * k);
    slot->h_tab = host;
    slot->coef.h_w2d = slot->h_tab.data();
    if (out)
        *out = &slot->coef;
    return MI355_OK;
}

int get_coef(mi355_ctx* ctx, int k, float sigma, const GaussCoef** out)
{
    for (auto& e : ctx->coefs)
        if (e.k == k && e.sigma_bits == fbits(sigma)) {
            e.last_use = ++ctx->coef_clock;
            *out = &e.coef;
            return MI355_OK;
        }
    std::vector<float> w((size_t)k * k);
    gen_weights(k, sigma, w.data());
    return install_coef(ctx, k, sigma, w.data(), false, out);
}

int ensure(mi355_ctx* ctx, void** p, size_t* cap, size_t need)
{
    if (*cap >= need)
        return MI355_OK;
    if (*p) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    // grow geometrically so a stream of slightly different frame sizes does not reallocate each time
    size_t want = need + need / 4;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {
        e = hipMalloc(p, need);
        want = need;
    }
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        *p = nullptr;
        return MI355_ERR_NOMEM;
    }
    *cap = want;
    return MI355_OK;
}

bool filter_needs_gauss(int f) { return f == MI355_FILTER_GAUSS || f == MI355_FILTER_PIPELINE; }

int check_frames(const void* in, const void* out, int w, int h, int nframes)
{
    if (!in || !out || w <= 0 || h <= 0 || nframes <= 0)
        return MI355_ERR_BAD_ARG;
    // 2^31 tiles / 2^40 bytes is far beyond 288 GB of HBM; reject before any size_t arithmetic wraps
    const double px = (double)w * (double)h * (double)nframes;
    if (px > 6.0e10)
        return MI355_ERR_BAD_ARG;
    return MI355_OK;
}

int dispatch_dev(mi355_ctx* ctx, int filter, const void* d_in, void* d_out, int w, int h, int nframes,
                 int k, float sigma)
{
    if (!ctx)
        return MI355_ERR_BAD_ARG;
    int rc = check_frames(d_in, d_out, w, h, nframes);
    if (rc != MI355_OK)
        return rc;
    if (reinterpret_cast<uintptr_t>(d_in) & 3u)
        return MI355_ERR_BAD_ARG;  // RGBA pixels are accessed as dwords
    const bool rgba_out = (filter == MI355_FILTER_GRAY || filter == MI355_FILTER_GAUSS);
    if (rgba_out && (reinterpret_cast<uintptr_t>(d_out) & 3u))
        return MI355_ERR_BAD_ARG;
    {
        // the stencil kernels read neighbouring rows / halo pixels of what another wave may already have overwritten,
        // and the (pointwise) grayscale kernels are compiled for non-aliasing pointers (__restrict__, non-temporal
        // accesses): in-place and overlapping calls are rejected, not run, for every filter (the reference never
        // aliases them either: two clCreateBuffer objects per call, RT/src/Controller.cpp:234-244)
        const int bpp = mi355_filter_out_bpp(filter);
        if (bpp < 0)
            return MI355_ERR_BAD_ARG;
        const size_t npx = (size_t)w * h * nframes;
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(d_in), a1 = a0 + npx * 4;
        const uintptr_t b0 = reinterpret_cast<uintptr_t>(d_out), b1 = b0 + npx * (size_t)bpp;
        if (a0 < b1 && b0 < a1)
            return MI355_ERR_BAD_ARG;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const GaussCoef* coef = nullptr;
    if (filter_needs_gauss(filter)) {
        if (!valid_k(k) || !valid_sigma(sigma))
            return MI355_ERR_BAD_ARG;
        rc = get_coef(ctx, k, sigma, &coef);
        if (rc != MI355_OK)
            return rc;
    }
    const uint8_t* in = static_cast<const uint8_t*>(d_in);
    uint8_t* out = static_cast<uint8_t*>(d_out);
    // a table the separable factor does not reproduce is applied tap by tap, whatever the mode
    const bool exact = ctx->gauss_mode == MI355_GAUSS_EXACT || (coef && !coef->separable);
    hipError_t e;
    switch (filter) {
    case MI355_FILTER_GRAY:
        e = launch_gray(ctx->stream, in, out, w, h, nframes, false);
        break;
    case MI355_FILTER_GRAY1:
        e = launch_gray(ctx->stream, in, out, w, h, nframes, true);
        break;
    case MI355_FILTER_GAUSS: {
        const size_t nflags = gauss_flag_items(in, out, w, h, nframes, *coef, exact, ctx->impl);
        if (nflags) {
            rc = ensure(ctx, &ctx->d_flags, &ctx->d_flags_cap, nflags * sizeof(uint32_t));
            if (rc != MI355_OK)
                return rc;
        }
        e = launch_gauss(ctx->stream, in, out, w, h, nframes, *coef, exact, ctx->impl,
                         static_cast<uint32_t*>(ctx->d_flags));
        break;
    }
    case MI355_FILTER_SOBEL:
        e = launch_sobel(ctx->stream, in, out, w, h, nframes, ctx->impl);
        break;
    case MI355_FILTER_PIPELINE:
        e = launch_pipeline(ctx->stream, in, out, w, h, nframes, *coef, exact, ctx->impl);
        break;
    default:
        return MI355_ERR_BAD_ARG;
    }
    HIP_TRY(ctx, e);
    return MI355_OK;
}

// The six timestamps of a write / kernel / read call from the four events recorded around the three operations on the
// in-order stream (ev[0] before the write, ev[1] between write and kernel, ev[3] between kernel and read, ev[5] after
// the read): write-end IS kernel-start, kernel-end IS read-start.
int fill_prof(mi355_ctx* ctx, uint64_t host0, uint64_t prof_ns[6])
{
    static const int kEv[6] = {0, 1, 1, 3, 3, 5};
    float ms1 = 0.0f, ms3 = 0.0f, ms5 = 0.0f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms1, ctx->ev[0], ctx->ev[1]));
    HIP_TRY(ctx, hipEventElapsedTime(&ms3, ctx->ev[0], ctx->ev[3]));
    HIP_TRY(ctx, hipEventElapsedTime(&ms5, ctx->ev[0], ctx->ev[5]));
    prof_ns[0] = host0;
    for (int i = 1; i < 6; i++) {
        const float ms = kEv[i] == 1 ? ms1 : (kEv[i] == 3 ? ms3 : ms5);
        prof_ns[i] = host0 + (uint64_t)((double)ms * 1.0e6);
    }
    return MI355_OK;
}

// H2D, kernel, D2H with the reference's six profiling timestamps (RT/src/Controller.cpp:66-74)
int run_host(mi355_ctx* ctx, int filter, const uint8_t* in, uint8_t* out, int w, int h, int nframes,
             int k, float sigma, uint64_t prof_ns[6])
{
    if (!ctx)
        return MI355_ERR_BAD_ARG;
    const int bpp = mi355_filter_out_bpp(filter);
    if (bpp < 0)
        return MI355_ERR_BAD_ARG;
    int rc = check_frames(in, out, w, h, nframes);
    if (rc != MI355_OK)
        return rc;
    if (filter_needs_gauss(filter) && (!valid_k(k) || !valid_sigma(sigma)))
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npx = (size_t)w * h * nframes;
    const bool bgr = ctx->input_format == MI355_INPUT_BGR;
    const size_t in_bytes = npx * 4, out_bytes = npx * (size_t)bpp;
    rc = ensure(ctx, &ctx->d_in, &ctx->d_in_cap, in_bytes);
    if (rc != MI355_OK)
        return rc;
    if (bgr) {
        rc = ensure(ctx, &ctx->d_raw, &ctx->d_raw_cap, npx * 3);
        if (rc != MI355_OK)
            return rc;
    }
    rc = ensure(ctx, &ctx->d_out, &ctx->d_out_cap, out_bytes);
    if (rc != MI355_OK)
        return rc;
    if (filter_needs_gauss(filter)) {
        // build/upload the coefficient table outside the timed write/kernel/read window
        const GaussCoef* coef = nullptr;
        rc = get_coef(ctx, k, sigma, &coef);
        if (rc != MI355_OK)
            return rc;
    }
    // Four events on the in-order stream give the six timestamps: write-end IS kernel-start and kernel-end IS
    // read-start (round 2 recorded six events and made five elapsed-time queries per call; at 75 x 75 the API calls
    // around the three operations were a third of the call).  No events at all when the caller wants no timestamps.
    hipStream_t s = ctx->stream;
    const uint64_t host0 = now_ns();
    if (prof_ns)
        HIP_TRY(ctx, hipEventRecord(ctx->ev[0], s));
    if (bgr) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_raw, in, npx * 3, hipMemcpyHostToDevice, s));
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_in, in, in_bytes, hipMemcpyHostToDevice, s));
    }
    if (prof_ns)
        HIP_TRY(ctx, hipEventRecord(ctx->ev[1], s));
    if (bgr)  // the BGR2RGBA expansion counts as kernel time
        HIP_TRY(ctx, launch_bgr_to_rgba(s, static_cast<const uint8_t*>(ctx->d_raw), static_cast<uint8_t*>(ctx->d_in), npx));
    rc = dispatch_dev(ctx, filter, ctx->d_in, ctx->d_out, w, h, nframes, k, sigma);
    if (rc != MI355_OK)
        return rc;
    if (prof_ns)
        HIP_TRY(ctx, hipEventRecord(ctx->ev[3], s));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    if (prof_ns)
        HIP_TRY(ctx, hipEventRecord(ctx->ev[5], s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (prof_ns)
        return fill_prof(ctx, host0, prof_ns);
    return MI355_OK;
}

int create_common(int device, hipStream_t stream, bool own, mi355_ctx** out)
{
    if (!out)
        return MI355_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return MI355_ERR_NO_DEVICE;
    if (device < 0 || device >= n)
        return MI355_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return MI355_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return MI355_ERR_UNSUPPORTED;  // the code objects in this library are gfx950 only
    mi355_ctx* ctx = new (std::nothrow) mi355_ctx();
    if (!ctx)
        return MI355_ERR_NOMEM;
    ctx->device = device;
    std::snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
    bool ok = hipSetDevice(device) == hipSuccess;
    if (ok && own)
        ok = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess;
    ctx->stream = stream;
    ctx->own_stream = own && ok;
    for (int i = 0; ok && i < 6; i++)
        ok = hipEventCreate(&ctx->ev[i]) == hipSuccess;
    ok = ok && hipEventCreate(&ctx->t0) == hipSuccess && hipEventCreate(&ctx->t1) == hipSuccess;
    ok = ok && hipMalloc((void**)&ctx->d_acc, sizeof(unsigned long long)) == hipSuccess;
    if (!ok) {
        mi355_ctx_destroy(ctx);
        return MI355_ERR_HIP;
    }
    *out = ctx;
    return MI355_OK;
}

}  // namespace

extern "C" {

#define MI355_API __attribute__((visibility("default")))

MI355_API int mi355_device_count(int* count)
{
    if (!count)
        return MI355_ERR_BAD_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        n = 0;
    *count = n;
    return MI355_OK;
}

MI355_API int mi355_ctx_create(int device, mi355_ctx** out)
{
    return create_common(device, nullptr, true, out);
}

MI355_API int mi355_ctx_create_on_stream(int device, void* hip_stream, mi355_ctx** out)
{
    return create_common(device, static_cast<hipStream_t>(hip_stream), false, out);
}

MI355_API int mi355_ctx_destroy(mi355_ctx* ctx)
{
    if (!ctx)
        return MI355_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& e : ctx->coefs)
        (void)hipFree(e.d_buf);
    for (void* p : ctx->pinned)
        (void)hipHostFree(p);
    if (ctx->d_in)
        (void)hipFree(ctx->d_in);
    if (ctx->d_raw)
        (void)hipFree(ctx->d_raw);
    for (void* p : ctx->slot_raw)
        if (p)
            (void)hipFree(p);
    if (ctx->d_out)
        (void)hipFree(ctx->d_out);
    if (ctx->d_acc)
        (void)hipFree(ctx->d_acc);
    if (ctx->d_flags)
        (void)hipFree(ctx->d_flags);
    if (ctx->d_img_table)
        (void)hipFree(ctx->d_img_table);
    for (int i = 0; i < mi355_ctx::kSlots; i++) {
        if (ctx->slot_in[i])
            (void)hipFree(ctx->slot_in[i]);
        if (ctx->slot_out[i])
            (void)hipFree(ctx->slot_out[i]);
        if (ctx->ev_h2d[i])
            (void)hipEventDestroy(ctx->ev_h2d[i]);
        if (ctx->ev_k[i])
            (void)hipEventDestroy(ctx->ev_k[i]);
        if (ctx->ev_d2h[i])
            (void)hipEventDestroy(ctx->ev_d2h[i]);
    }
    if (ctx->s_h2d)
        (void)hipStreamDestroy(ctx->s_h2d);
    if (ctx->s_d2h)
        (void)hipStreamDestroy(ctx->s_d2h);
    for (auto& ev : ctx->ev)
        if (ev)
            (void)hipEventDestroy(ev);
    if (ctx->t0)
        (void)hipEventDestroy(ctx->t0);
    if (ctx->t1)
        (void)hipEventDestroy(ctx->t1);
    if (ctx->own_stream && ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MI355_OK;
}

MI355_API int mi355_ctx_device_name(mi355_ctx* ctx, char* buf, size_t buflen)
{
    if (!ctx || !buf || buflen == 0)
        return MI355_ERR_BAD_ARG;
    std::snprintf(buf, buflen, "%s", ctx->name);
    return MI355_OK;
}

MI355_API int mi355_sync(mi355_ctx* ctx)
{
    if (!ctx)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

MI355_API int mi355_last_hip_error(mi355_ctx* ctx) { return ctx ? ctx->last_hip : 0; }

MI355_API const char* mi355_strerror(int code)
{
    switch (code) {
    case MI355_OK: return "ok";
    case MI355_ERR_BAD_ARG: return "bad argument";
    case MI355_ERR_HIP: return "HIP runtime error";
    case MI355_ERR_NO_DEVICE: return "no such GPU";
    case MI355_ERR_UNSUPPORTED: return "unsupported (library is built for gfx950 only)";
    case MI355_ERR_NOMEM: return "out of memory";
    default: return "unknown error";
    }
}

MI355_API int mi355_ctx_set_gauss_mode(mi355_ctx* ctx, int mode)
{
    if (!ctx || (mode != MI355_GAUSS_FAST && mode != MI355_GAUSS_EXACT))
        return MI355_ERR_BAD_ARG;
    ctx->gauss_mode = mode;
    return MI355_OK;
}

MI355_API int mi355_ctx_set_impl(mi355_ctx* ctx, int impl)
{
    if (!ctx || impl < MI355_IMPL_AUTO || impl > MI355_IMPL_VALU)
        return MI355_ERR_BAD_ARG;
    ctx->impl = impl;
    return MI355_OK;
}

MI355_API int mi355_ctx_set_input_format(mi355_ctx* ctx, int format)
{
    if (!ctx || (format != MI355_INPUT_RGBA && format != MI355_INPUT_BGR))
        return MI355_ERR_BAD_ARG;
    ctx->input_format = format;
    return MI355_OK;
}

MI355_API int mi355_bgr_to_rgba8_dev(mi355_ctx* ctx, const void* d_bgr, void* d_rgba, int w, int h, int nframes)
{
    if (!ctx)
        return MI355_ERR_BAD_ARG;
    int rc = check_frames(d_bgr, d_rgba, w, h, nframes);
    if (rc != MI355_OK)
        return rc;
    if (reinterpret_cast<uintptr_t>(d_rgba) & 3u)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_bgr_to_rgba(ctx->stream, static_cast<const uint8_t*>(d_bgr), static_cast<uint8_t*>(d_rgba),
                                    (size_t)w * h * nframes));
    return MI355_OK;
}

MI355_API int mi355_gauss_weights(int k, float sigma, float* out_k2)
{
    if (!out_k2 || !valid_k(k) || !valid_sigma(sigma))
        return MI355_ERR_BAD_ARG;
    gen_weights(k, sigma, out_k2);
    return MI355_OK;
}

MI355_API int mi355_gauss_weights_image2d(int k, float sigma, float* out_k2)
{
    if (!out_k2 || !valid_k(k) || !valid_sigma(sigma))
        return MI355_ERR_BAD_ARG;
    gen_weights_image2d(k, sigma, out_k2);
    return MI355_OK;
}

MI355_API int mi355_image2d_rgba8(mi355_ctx* ctx, int filter, const uint8_t* rgba, uint8_t* out, int w, int h, int k,
                                  float sigma, uint64_t prof_ns[6])
{
    if (!ctx || (filter != MI355_FILTER_GRAY && filter != MI355_FILTER_GAUSS && filter != MI355_FILTER_SOBEL))
        return MI355_ERR_BAD_ARG;
    int rc = check_frames(rgba, out, w, h, 1);
    if (rc != MI355_OK)
        return rc;
    const bool gauss = filter == MI355_FILTER_GAUSS;
    if (gauss && (!valid_k(k) || !valid_sigma(sigma)))
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npx = (size_t)w * h, in_bytes = npx * 4, out_bytes = npx * (gauss ? 4 : 1);
    rc = ensure(ctx, &ctx->d_in, &ctx->d_in_cap, in_bytes);
    if (rc != MI355_OK)
        return rc;
    rc = ensure(ctx, &ctx->d_out, &ctx->d_out_cap, out_bytes);
    if (rc != MI355_OK)
        return rc;
    if (gauss) {
        if (!ctx->d_img_table)
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_img_table, sizeof(float) * MI355_MAX_GAUSS_K * MI355_MAX_GAUSS_K));
        std::vector<float> table((size_t)k * k);
        gen_weights_image2d(k, sigma, table.data());
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // a kernel in flight may still read the previous table
        HIP_TRY(ctx, hipMemcpy(ctx->d_img_table, table.data(), sizeof(float) * table.size(), hipMemcpyHostToDevice));
    }
    hipStream_t s = ctx->stream;
    const uint64_t host0 = now_ns();
    if (prof_ns)
        HIP_TRY(ctx, hipEventRecord(ctx->ev[0], s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_in, rgba, in_bytes, hipMemcpyHostToDevice, s));
    if (prof_ns)
        HIP_TRY(ctx, hipEventRecord(ctx->ev[1], s));
    HIP_TRY(ctx, launch_image2d(s, filter, static_cast<const uint8_t*>(ctx->d_in), static_cast<uint8_t*>(ctx->d_out), w, h, 1,
                                k, ctx->d_img_table));
    if (prof_ns)
        HIP_TRY(ctx, hipEventRecord(ctx->ev[3], s));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    if (prof_ns)
        HIP_TRY(ctx, hipEventRecord(ctx->ev[5], s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (prof_ns)
        return fill_prof(ctx, host0, prof_ns);
    return MI355_OK;
}

MI355_API int mi355_ctx_set_gauss_weights(mi355_ctx* ctx, int k, float sigma, const float* w_k2)
{
    if (!ctx || !w_k2 || !valid_k(k) || !valid_sigma(sigma))
        return MI355_ERR_BAD_ARG;
    for (int i = 0; i < k * k; i++)
        if (!std::isfinite(w_k2[i]))
            return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // a kernel still in flight may be reading the previous table of this key
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return install_coef(ctx, k, sigma, w_k2, true, nullptr);
}

MI355_API int mi355_filter_out_bpp(int filter)
{
    switch (filter) {
    case MI355_FILTER_GRAY:
    case MI355_FILTER_GAUSS: return 4;
    case MI355_FILTER_GRAY1:
    case MI355_FILTER_SOBEL:
    case MI355_FILTER_PIPELINE: return 1;
    default: return MI355_ERR_BAD_ARG;
    }
}

MI355_API int mi355_gray_rgba8(mi355_ctx* ctx, const uint8_t* rgba, uint8_t* out, int w, int h,
                               uint64_t prof_ns[6])
{
    return run_host(ctx, MI355_FILTER_GRAY, rgba, out, w, h, 1, 0, 0.0f, prof_ns);
}

MI355_API int mi355_gray1_rgba8(mi355_ctx* ctx, const uint8_t* rgba, uint8_t* out, int w, int h,
                                uint64_t prof_ns[6])
{
    return run_host(ctx, MI355_FILTER_GRAY1, rgba, out, w, h, 1, 0, 0.0f, prof_ns);
}

MI355_API int mi355_gauss_rgba8(mi355_ctx* ctx, const uint8_t* rgba, uint8_t* out, int w, int h, int k,
                                float sigma, uint64_t prof_ns[6])
{
    return run_host(ctx, MI355_FILTER_GAUSS, rgba, out, w, h, 1, k, sigma, prof_ns);
}

MI355_API int mi355_sobel_rgba8(mi355_ctx* ctx, const uint8_t* rgba, uint8_t* out, int w, int h,
                                uint64_t prof_ns[6])
{
    return run_host(ctx, MI355_FILTER_SOBEL, rgba, out, w, h, 1, 0, 0.0f, prof_ns);
}

MI355_API int mi355_pipeline_rgba8(mi355_ctx* ctx, const uint8_t* rgba, uint8_t* out, int w, int h,
                                   int k, float sigma, uint64_t prof_ns[6])
{
    return run_host(ctx, MI355_FILTER_PIPELINE, rgba, out, w, h, 1, k, sigma, prof_ns);
}

MI355_API int mi355_filter_batched(mi355_ctx* ctx, int filter, const uint8_t* rgba, uint8_t* out, int w,
                                   int h, int nframes, int k, float sigma, uint64_t prof_ns[6])
{
    return run_host(ctx, filter, rgba, out, w, h, nframes, k, sigma, prof_ns);
}

MI355_API int mi355_host_alloc(mi355_ctx* ctx, size_t nbytes, void** h_ptr)
{
    if (!ctx || !h_ptr || nbytes == 0)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(h_ptr, nbytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        *h_ptr = nullptr;
        return MI355_ERR_NOMEM;
    }
    ctx->pinned.push_back(*h_ptr);
    return MI355_OK;
}

MI355_API int mi355_host_free(mi355_ctx* ctx, void* h_ptr)
{
    if (!ctx)
        return MI355_ERR_BAD_ARG;
    if (!h_ptr)
        return MI355_OK;
    for (size_t i = 0; i < ctx->pinned.size(); i++)
        if (ctx->pinned[i] == h_ptr) {
            ctx->pinned.erase(ctx->pinned.begin() + i);
            HIP_TRY(ctx, hipHostFree(h_ptr));
            return MI355_OK;
        }
    return MI355_ERR_BAD_ARG;  // not a block of this context
}

MI355_API int mi355_filter_stream(mi355_ctx* ctx, int filter, const uint8_t* rgba, uint8_t* out, int w, int h,
                                  int nframes, int chunk_frames, int k, float sigma, double* elapsed_ms)
{
    if (!ctx)
        return MI355_ERR_BAD_ARG;
    const int bpp = mi355_filter_out_bpp(filter);
    if (bpp < 0 || chunk_frames < 0)
        return MI355_ERR_BAD_ARG;
    int rc = check_frames(rgba, out, w, h, nframes);
    if (rc != MI355_OK)
        return rc;
    if (filter_needs_gauss(filter) && (!valid_k(k) || !valid_sigma(sigma)))
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t fpx = (size_t)w * h;
    if (chunk_frames == 0) {
        // ~64 MB of input per chunk: long enough DMA transfers to run at link rate, short enough to overlap
        chunk_frames = (int)((64u << 20) / (fpx * 4));
        if (chunk_frames < 1)
            chunk_frames = 1;
    }
    if (chunk_frames > nframes)
        chunk_frames = nframes;
    constexpr int NS = mi355_ctx::kSlots;
    if (!ctx->s_h2d) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->s_h2d, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->s_d2h, hipStreamNonBlocking));
        for (int i = 0; i < NS; i++) {
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_h2d[i], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_k[i], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_d2h[i], hipEventDisableTiming));
        }
    }
    const bool bgr = ctx->input_format == MI355_INPUT_BGR;
    const size_t in_bpp = bgr ? 3 : 4;
    const size_t in_chunk = fpx * 4 * chunk_frames, out_chunk = fpx * (size_t)bpp * chunk_frames;
    if (bgr && ctx->slot_raw_cap < fpx * 3 * chunk_frames) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        for (int i = 0; i < NS; i++) {
            if (ctx->slot_raw[i])
                (void)hipFree(ctx->slot_raw[i]);
            ctx->slot_raw[i] = nullptr;
        }
        ctx->slot_raw_cap = 0;
        for (int i = 0; i < NS; i++)
            if (hipMalloc(&ctx->slot_raw[i], fpx * 3 * chunk_frames) != hipSuccess)
                return MI355_ERR_NOMEM;
        ctx->slot_raw_cap = fpx * 3 * chunk_frames;
    }
    if (ctx->slot_in_cap < in_chunk || ctx->slot_out_cap < out_chunk) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        for (int i = 0; i < NS; i++) {
            if (ctx->slot_in[i])
                (void)hipFree(ctx->slot_in[i]);
            if (ctx->slot_out[i])
                (void)hipFree(ctx->slot_out[i]);
            ctx->slot_in[i] = ctx->slot_out[i] = nullptr;
        }
        ctx->slot_in_cap = ctx->slot_out_cap = 0;
        for (int i = 0; i < NS; i++) {
            if (hipMalloc(&ctx->slot_in[i], in_chunk) != hipSuccess ||
                hipMalloc(&ctx->slot_out[i], out_chunk) != hipSuccess)
                return MI355_ERR_NOMEM;
        }
        ctx->slot_in_cap = in_chunk;
        ctx->slot_out_cap = out_chunk;
    }
    if (filter_needs_gauss(filter)) {
        const GaussCoef* coef = nullptr;
        rc = get_coef(ctx, k, sigma, &coef);
        if (rc != MI355_OK)
            return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t t0 = now_ns();
    const int nchunks = (nframes + chunk_frames - 1) / chunk_frames;
    for (int c = 0; c < nchunks; c++) {
        const int slot = c % NS;
        const int f0 = c * chunk_frames;
        const int nf = (nframes - f0 < chunk_frames) ? nframes - f0 : chunk_frames;
        // input slot is free once the kernel of chunk c-NS has read it
        if (c >= NS)
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_h2d, ctx->ev_k[slot], 0));
        HIP_TRY(ctx, hipMemcpyAsync(bgr ? ctx->slot_raw[slot] : ctx->slot_in[slot], rgba + (size_t)f0 * fpx * in_bpp,
                                    fpx * in_bpp * nf, hipMemcpyHostToDevice, ctx->s_h2d));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_h2d[slot], ctx->s_h2d));
        // kernel: needs its input, and its output slot drained by the D2H of chunk c-NS
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_h2d[slot], 0));
        if (c >= NS)
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_d2h[slot], 0));
        if (bgr)
            HIP_TRY(ctx, launch_bgr_to_rgba(ctx->stream, static_cast<const uint8_t*>(ctx->slot_raw[slot]),
                                            static_cast<uint8_t*>(ctx->slot_in[slot]), fpx * nf));
        rc = dispatch_dev(ctx, filter, ctx->slot_in[slot], ctx->slot_out[slot], w, h, nf, k, sigma);
        if (rc != MI355_OK) {
            (void)hipDeviceSynchronize();
            return rc;
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k[slot], ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_d2h, ctx->ev_k[slot], 0));
        HIP_TRY(ctx, hipMemcpyAsync(out + (size_t)f0 * fpx * bpp, ctx->slot_out[slot], fpx * (size_t)bpp * nf,
                                    hipMemcpyDeviceToHost, ctx->s_d2h));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_d2h[slot], ctx->s_d2h));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_d2h));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_h2d));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (elapsed_ms)
        *elapsed_ms = (double)(now_ns() - t0) * 1e-6;
    return MI355_OK;
}

MI355_API int mi355_filter_dev(mi355_ctx* ctx, int filter, const void* d_in, void* d_out, int w, int h,
                               int nframes, int k, float sigma)
{
    if (mi355_filter_out_bpp(filter) < 0)
        return MI355_ERR_BAD_ARG;
    return dispatch_dev(ctx, filter, d_in, d_out, w, h, nframes, k, sigma);
}

MI355_API int mi355_gray_rgba8_dev(mi355_ctx* ctx, const void* d_in, void* d_out, int w, int h,
                                   int nframes)
{
    return dispatch_dev(ctx, MI355_FILTER_GRAY, d_in, d_out, w, h, nframes, 0, 0.0f);
}

MI355_API int mi355_gray1_rgba8_dev(mi355_ctx* ctx, const void* d_in, void* d_out, int w, int h,
                                    int nframes)
{
    return dispatch_dev(ctx, MI355_FILTER_GRAY1, d_in, d_out, w, h, nframes, 0, 0.0f);
}

MI355_API int mi355_gauss_rgba8_dev(mi355_ctx* ctx, const void* d_in, void* d_out, int w, int h,
                                    int nframes, int k, float sigma)
{
    return dispatch_dev(ctx, MI355_FILTER_GAUSS, d_in, d_out, w, h, nframes, k, sigma);
}

MI355_API int mi355_sobel_rgba8_dev(mi355_ctx* ctx, const void* d_in, void* d_out, int w, int h,
                                    int nframes)
{
    return dispatch_dev(ctx, MI355_FILTER_SOBEL, d_in, d_out, w, h, nframes, 0, 0.0f);
}

MI355_API int mi355_pipeline_rgba8_dev(mi355_ctx* ctx, const void* d_in, void* d_out, int w, int h,
                                       int nframes, int k, float sigma)
{
    return dispatch_dev(ctx, MI355_FILTER_PIPELINE, d_in, d_out, w, h, nframes, k, sigma);
}

MI355_API int mi355_synth_rgba8_dev(mi355_ctx* ctx, void* d_out, int w, int h, int nframes,
                                    int first_frame, uint32_t seed, int mode)
{
    if (!ctx || mode < 0 || mode > 3)
        return MI355_ERR_BAD_ARG;
    int rc = check_frames(d_out, d_out, w, h, nframes);
    if (rc != MI355_OK)
        return rc;
    if (reinterpret_cast<uintptr_t>(d_out) & 3u)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_synth(ctx->stream, static_cast<uint8_t*>(d_out), w, h, nframes, first_frame, seed,
                              mode));
    return MI355_OK;
}

MI355_API int mi355_checksum_dev(mi355_ctx* ctx, const void* d_buf, size_t nbytes, uint64_t index_base,
                                 uint64_t* out)
{
    if (!ctx || !d_buf || !out)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_acc, 0, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, launch_checksum(ctx->stream, static_cast<const uint8_t*>(d_buf), nbytes, index_base,
                                 ctx->d_acc));
    unsigned long long v = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&v, ctx->d_acc, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out = (uint64_t)v;
    return MI355_OK;
}

MI355_API int mi355_stream_copy_dev(mi355_ctx* ctx, void* d_dst, const void* d_src, size_t nbytes)
{
    if (!ctx || !d_dst || !d_src)
        return MI355_ERR_BAD_ARG;
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(d_src), b0 = reinterpret_cast<uintptr_t>(d_dst);
    if (a0 < b0 + nbytes && b0 < a0 + nbytes)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_stream_copy(ctx->stream, static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst),
                                    nbytes));
    return MI355_OK;
}

MI355_API int mi355_selftest(mi355_ctx* ctx, uint32_t* bad_luma, uint32_t* bad_mag)
{
    if (!ctx || !bad_luma || !bad_mag)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_acc, 0, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, launch_selftest(ctx->stream, ctx->d_acc));
    unsigned long long v = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&v, ctx->d_acc, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *bad_luma = (uint32_t)v;
    *bad_mag = (uint32_t)(v >> 32);
    return MI355_OK;
}

MI355_API int mi355_dev_alloc(mi355_ctx* ctx, size_t nbytes, void** d_ptr)
{
    if (!ctx || !d_ptr || nbytes == 0)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(d_ptr, nbytes);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        *d_ptr = nullptr;
        return MI355_ERR_NOMEM;
    }
    return MI355_OK;
}

MI355_API int mi355_pool_alloc(mi355_ctx* ctx, int filter, int w, int h, int nframes, int k, float sigma, int tries,
                               void** d_in, void** d_out, float* probe_ms)
{
    if (!ctx || !d_in || !d_out)
        return MI355_ERR_BAD_ARG;
    *d_in = *d_out = nullptr;
    const int bpp = mi355_filter_out_bpp(filter);
    if (bpp < 0 || w <= 0 || h <= 0 || nframes <= 0 || (double)w * h * nframes > 6.0e10)
        return MI355_ERR_BAD_ARG;
    if (filter_needs_gauss(filter) && (!valid_k(k) || !valid_sigma(sigma)))
        return MI355_ERR_BAD_ARG;
    constexpr int kMaxCand = 16;
    for (int i = 0; probe_ms && i < tries; i++)
        probe_ms[i] = -1.0f;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npx = (size_t)w * h * nframes, in_bytes = npx * 4, out_bytes = npx * (size_t)bpp;
    void* in = nullptr;
    if (hipMalloc(&in, in_bytes) != hipSuccess)
        return MI355_ERR_NOMEM;
    // defined, opaque input (A = 255): the placement probe then times the path real frames take
    hipError_t e = hipMemsetAsync(in, 0xFF, in_bytes, ctx->stream);
    int rc = (e == hipSuccess) ? MI355_OK : MI355_ERR_HIP;
    if (e != hipSuccess)
        ctx->last_hip = (int)e;

    // all candidates alive at once, hence all in different places; stop early when memory runs short
    void* cand[kMaxCand] = {};
    int ncand = 0;
    const int want = tries < 1 ? 1 : (tries > kMaxCand ? kMaxCand : tries);
    while (rc == MI355_OK && ncand < want) {
        size_t free_b = 0, total_b = 0;
        if (ncand > 0 && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < out_bytes + ((size_t)8 << 30)))
            break;
        if (hipMalloc(&cand[ncand], out_bytes) != hipSuccess) {
            cand[ncand] = nullptr;
            break;
        }
        ncand++;
    }
    if (rc == MI355_OK && ncand == 0)
        rc = MI355_ERR_NOMEM;
    int keep = 0;
    float best_ms = 0.0f;
    for (int i = 0; rc == MI355_OK && ncand > 1 && i < ncand; i++) {
        float ms = 0.0f;
        for (int l = 0; rc == MI355_OK && l < 4 + 8; l++) {
            if (l == 4 && (e = hipEventRecord(ctx->t0, ctx->stream)) != hipSuccess) {
                ctx->last_hip = (int)e;
                rc = MI355_ERR_HIP;
                break;
            }
            rc = dispatch_dev(ctx, filter, in, cand[i], w, h, nframes, k, sigma);
        }
        if (rc != MI355_OK)
            break;
        e = hipEventRecord(ctx->t1, ctx->stream);
        if (e == hipSuccess)
            e = hipEventSynchronize(ctx->t1);
        if (e == hipSuccess)
            e = hipEventElapsedTime(&ms, ctx->t0, ctx->t1);
        if (e != hipSuccess) {
            ctx->last_hip = (int)e;
            rc = MI355_ERR_HIP;
            break;
        }
        ms /= 8.0f;
        if (probe_ms)
            probe_ms[i] = ms;
        if (i == 0 || ms < best_ms) {
            best_ms = ms;
            keep = i;
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < ncand; i++)
        if (rc != MI355_OK || i != keep)
            (void)hipFree(cand[i]);
    if (rc != MI355_OK) {
        (void)hipFree(in);
        return rc;
    }
    *d_in = in;
    *d_out = cand[keep];
    return MI355_OK;
}

MI355_API int mi355_pool_free(mi355_ctx* ctx, void* d_in, void* d_out)
{
    if (!ctx)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (d_in)
        HIP_TRY(ctx, hipFree(d_in));
    if (d_out)
        HIP_TRY(ctx, hipFree(d_out));
    return MI355_OK;
}

MI355_API int mi355_dev_free(mi355_ctx* ctx, void* d_ptr)
{
    if (!ctx)
        return MI355_ERR_BAD_ARG;
    if (!d_ptr)
        return MI355_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(d_ptr));
    return MI355_OK;
}

MI355_API int mi355_copy_h2d(mi355_ctx* ctx, void* d_dst, const void* h_src, size_t nbytes)
{
    if (!ctx || !d_dst || !h_src)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

MI355_API int mi355_copy_d2h(mi355_ctx* ctx, void* h_dst, const void* d_src, size_t nbytes)
{
    if (!ctx || !h_dst || !d_src)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

MI355_API int mi355_timer_begin(mi355_ctx* ctx)
{
    if (!ctx)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->t0, ctx->stream));
    return MI355_OK;
}

MI355_API int mi355_timer_end(mi355_ctx* ctx, float* elapsed_ms)
{
    if (!ctx || !elapsed_ms)
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->t1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->t1));
    HIP_TRY(ctx, hipEventElapsedTime(elapsed_ms, ctx->t0, ctx->t1));
    return MI355_OK;
}

}  // extern "C"

// group.hip installs the table it generated once on every member through this (hidden) hook: same path as a table
// the context generated itself — evictable, outside the cap on caller-installed tables.
int mi355_internal_install_generated(mi355_ctx* ctx, int k, float sigma, const float* w_k2)
{
    if (!ctx || !w_k2 || !valid_k(k) || !valid_sigma(sigma))
        return MI355_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (auto& e : ctx->coefs)
        if (e.k == k && e.sigma_bits == fbits(sigma))
            return MI355_OK;  // already there (generated or installed by the caller: the caller's table wins)
    return install_coef(ctx, k, sigma, w_k2, false, nullptr);
}

extern "C" {

MI355_API const char* mi355_build_info(void) { return "gfx950;" __DATE__ " " __TIME__; }

}  // extern "C"
