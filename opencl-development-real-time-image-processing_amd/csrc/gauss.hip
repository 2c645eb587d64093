// gauss.hip — Gaussian blur dispatch: picks the kernel for (k, width, mode).
#include "common.hpp"
#include "kernels.hpp"

namespace mi355 {

size_t gauss_flag_items(const uint8_t* d_in, const uint8_t* d_out, int w, int h, int nframes, int k, bool exact,
                        int impl)
{
    if (exact || impl == 1)
        return 0;
    if (gauss_slide_supported(d_in, d_out, w, h, k))
        return gauss_slide_flag_items(w, h, nframes, k);
    if (gauss_wide_supported(d_in, d_out, w, h, k))
        return gauss_wide_flag_items(w, h, nframes, k);
    return 0;
}

hipError_t launch_gauss(hipStream_t stream, const uint8_t* d_in, uint8_t* d_out, int w, int h,
                        int nframes, const GaussCoef& coef, bool exact, int impl, uint32_t* d_flags)
{
    // EXACT arithmetic exists only in the tiled kernel; the sliding-window kernel covers the FAST
    // arithmetic for the small kernels and 4-pixel-aligned rows (every config in BASELINE.json).
    if (!exact && impl != 1 && gauss_slide_supported(d_in, d_out, w, h, coef.k))
        return launch_gauss_slide(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    if (!exact && impl != 1 && gauss_wide_supported(d_in, d_out, w, h, coef.k))
        return launch_gauss_wide(stream, d_in, d_out, w, h, nframes, coef, d_flags);
    return launch_gauss_tile(stream, d_in, d_out, w, h, nframes, coef, exact);
}

}  // namespace mi355
