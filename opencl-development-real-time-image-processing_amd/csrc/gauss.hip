// gauss.hip — Gaussian blur dispatch: picks the kernel for (k, width, mode, impl).
#include "../../include/mi355_imgfilter.h"
#include "common.hpp"
#include "kernels.hpp"

#include <cmath>

namespace mi355 {

uint32_t gauss_const_alpha(const float* w1d, int k, uint32_t a)
{
    const float av = (float)a;
    float vc = w1d[0] * av;
    for (int t = 1; t < k; t++)
        vc = std::fmaf(w1d[t], av, vc);  // finished vertical sum
    float hc = w1d[0] * vc;
    for (int t = 1; t < k; t++)
        hc = std::fmaf(w1d[t], vc, hc);
    hc = hc < 0.0f ? 0.0f : (hc > 255.0f ? 255.0f : hc);  // uchar(std::clamp(.)) of the CPU path
    return (uint32_t)hc;
}

namespace {

enum class GaussKernel { Tile, Slide, Wide, Mfma, Exact };

// AUTO: which k the matrix-core kernel takes over from the register-resident VALU kernels.  Its cost does not
// depend on k (one K = 32 matrix instruction covers any radius <= 8), theirs grows with k.  Same box, 256 x 4K frames,
// VALU / matrix cores (profiles/r02_kernel_table.txt, tools/k79_ab.sh): k = 5 opaque 5.91 / 5.07 TB/s, alpha noise
// 5.16 / 4.5; k = 7 opaque 4.96 / 5.21 (5.05 / 4.98 on another box), alpha noise 4.20 / 4.60; k = 9 opaque 4.32 / 5.16,
// alpha noise 3.42 / 4.61, one frame 2.22 / 3.35; k = 11 2.83 / 5.05; k = 17 2.01 / 5.04.
constexpr int kMfmaAutoMinK = 7;

GaussKernel choose(const uint8_t* d_in, const uint8_t* d_out, int w, int h, int nframes, const GaussCoef& coef,
                   bool exact, int impl)
{
    // EXACT arithmetic: the sliding-window exact-by-exception kernel where it applies (a table it cannot take —
    // non-separable, asymmetric factor — arrives here with exact = true as well), the tiled kernel otherwise
    if (exact && impl != MI355_IMPL_TILE && gauss_exact_supported(d_in, d_out, w, h, coef))
        return GaussKernel::Exact;
    if (exact || impl == MI355_IMPL_TILE)
        return GaussKernel::Tile;
    // (frames of 2 GiB and more — 23,000 x 23,000 pixels — stay on the VALU kernels: the LDS-staged matrix-core kernel
    // that used to take them is an A/B partner in the tuning build only)
    const bool mfma_ok = gauss_mfma_reg_supported(d_in, d_out, w, h, coef);
    if (impl == MI355_IMPL_MFMA && mfma_ok)
        return GaussKernel::Mfma;
    // a launch must be worth a 64-pixel-wide, 16-row-blocked decomposition
    if (impl == MI355_IMPL_AUTO && mfma_ok && coef.k >= kMfmaAutoMinK && w >= 64 && (size_t)w * h * nframes >= (1u << 16))
        return GaussKernel::Mfma;
    if (gauss_slide_supported(d_in, d_out, w, h, coef.k))
        return GaussKernel::Slide;
    if (gauss_wide_supported(d_in, d_out, w, h, coef.k))
        return GaussKernel::Wide;
    return GaussKernel::Tile;
}

}  // namespace

size_t gauss_flag_items(const uint8_t* d_in, const uint8_t* d_out, int w, int h, int nframes, const GaussCoef& coef,
                        bool exact, int impl)
{
    switch (choose(d_in, d_out, w, h, nframes, coef, exact, impl)) {
    case GaussKernel::Slide: return gauss_slide_flag_items(w, h, nframes, coef.k);
    case GaussKernel::Wide: return gauss_wide_flag_items(w, h, nframes, coef.k);
    default: return 0;
    }
}

hipError_t launch_gauss(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                        int nframes, const GaussCoef& coef, bool exact, int impl, uint32_t* d_flags)
{
    switch (choose(d_in, d_out, w, h, nframes, coef, exact, impl)) {
    case GaussKernel::Mfma:
#ifdef MI355_TUNE_ENV
        // A/B partners, tuning build only (csrc/Makefile TSRCS): the LDS-staged first version (MI355_MFMA_LDS=1) and the
        // LDS-DMA input staging (MI355_MFMA_DMA=1) — same bits as gauss_mfma_reg.hip, not faster (-1 .. +2 % over five
        // launch shapes, profiles/r02_mfma_ablations.txt)
        if (tune_env("MI355_MFMA_LDS") && gauss_mfma_supported(d_in, d_out, w, h, coef))
            return launch_gauss_mfma(stream, d_in, d_out, w, h, nframes, coef);
        if (tune_env("MI355_MFMA_DMA") && gauss_mfma_dma_supported(d_in, d_out, w, h, coef))
            return launch_gauss_mfma_dma(stream, d_in, d_out, w, h, nframes, coef);
#endif
        return launch_gauss_mfma_reg(stream, d_in, d_out, w, h, nframes, coef);
    case GaussKernel::Exact: return launch_gauss_exact(stream, d_in, d_out, w, h, nframes, coef);
    case GaussKernel::Slide: return launch_gauss_slide(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    case GaussKernel::Wide: return launch_gauss_wide(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    default: return launch_gauss_tile(stream, d_in, d_out, w, h, nframes, coef, exact);
    }
}

}  // namespace mi355
