"""GPU suite, part 3 (-m gpu): the HIP path on the reference's own test images, against the reference's own published
numbers.

tests/test_published_mae.py pins the CPU restatement (the oracle) with the `Error_MAE` values the reference's benchmark
applications published (tests/golden/published_mae.json).  Here the PRODUCT takes the oracle's seat: the same
comparisons with the CPU operand replaced by what the HIP kernels return through the C-ABI.  Where the product is
bit-exact with the CPU path (grayscale, Sobel, EXACT-mode Gaussian) it must reproduce the published numbers itself.
Runs on the committed decoded pixels (tests/golden/ref_images); /root/reference is not needed.
"""
import numpy as np
import pytest

from test_published_mae import CL, NAMES, _committed, mae, printed, published, rgba_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref_images():
    out = {n: _committed(n) for n in NAMES if _committed(n) is not None}
    assert len(out) == 6
    return out


def test_hip_paths_equal_the_oracle_on_the_reference_images(ctx, pkg, oracle, ref_images):
    """Every filter, every committed reference image (odd sizes included: 75x75, 427x640), whole image."""
    for n, (rgb, y) in ref_images.items():
        rgba = rgba_of(rgb)
        assert np.array_equal(ctx.gray1(rgba), oracle.gray_rgba_1ch(rgba)), n
        assert np.array_equal(ctx.gray(rgba), oracle.gray_rgba(rgba)), n
        assert np.array_equal(ctx.sobel(rgba), oracle.sobel_rgba(rgba)), n
        ref = oracle.gauss_rgba(rgba, 5, 1.5)
        d = np.abs(ctx.gauss(rgba, 5, 1.5).astype(int) - ref.astype(int))
        assert d.max() <= 1, n
        ctx.set_gauss_mode(pkg.GAUSS_EXACT)
        try:
            assert np.array_equal(ctx.gauss(rgba, 5, 1.5), ref), n
        finally:
            ctx.set_gauss_mode(pkg.GAUSS_FAST)
        assert np.array_equal(ctx.pipeline(rgba, 5, 1.5), oracle.pipeline_rgba(rgba, 5, 1.5)), n
        # the decoder's luma plane (the reference's CPU Sobel input) as a gray RGBA frame: luma(y, y, y) is not the
        # identity for 65 byte values, so feed Sobel the re-grayed plane on both sides
        yy = np.ascontiguousarray(np.dstack([y, y, y, np.full_like(y, 255)]))
        assert np.array_equal(ctx.sobel(yy), oracle.sobel_gray(oracle.gray_rgba_1ch(yy))), n


def test_hip_grayscale_reproduces_the_published_numbers(ctx, ref_images):
    """src/Grayscale/results/Linux_100_*_sorted_results.csv, Error_MAE: the HIP grayscale in the CPU path's seat."""
    for n, (rgb, _) in ref_images.items():
        rgba = rgba_of(rgb)
        assert printed(mae(ctx.gray1(rgba), CL.cl_gray(rgba))) == published("gray", n), n


def test_hip_exact_gaussian_reproduces_the_published_numbers(ctx, pkg, oracle, ref_images):
    """src/GaussianBlur/results/Linux_100_*_sorted_results.csv (blue plane, k = 5, sigma = 1.5).  EXACT mode is
    bit-identical to the CPU path and reproduces them; FAST mode (<= 1 LSB by contract) is reported beside it."""
    wt = oracle.gauss_weights(5, 1.5)
    for n, (rgb, _) in ref_images.items():
        rgba = rgba_of(rgb)
        ocl = CL.cl_gauss(rgba, 5, wt)
        ctx.set_gauss_mode(pkg.GAUSS_EXACT)
        try:
            exact = ctx.gauss(rgba, 5, 1.5)
        finally:
            ctx.set_gauss_mode(pkg.GAUSS_FAST)
        assert printed(mae(exact[..., 2], ocl[..., 2])) == published("gauss", n), n
        fast = ctx.gauss(rgba, 5, 1.5)
        assert mae(fast[..., 2], ocl[..., 2]) < 0.01  # FAST differs from either path in < 1 % of the bytes, by 1


def test_hip_sobel_on_the_decoder_luma_plane(ctx, oracle, ref_images):
    """src/EdgeDetection/results/Linux_100_*_sorted_results.csv compare the OpenCL kernel with OpenCV's Sobel of the
    JPEG decoder's luma plane y (EdgeDetection.cpp:202).  The product's Sobel takes RGBA and applies the reference's
    luminance first, and luma(v, v, v) = v - 1 for 65 byte values, so a (y, y, y, 255) frame reaches the stencil as y
    only where no such value sits in the 3x3 window: there the HIP output IS the CPU path's value (the operand the
    published numbers were computed with); elsewhere it is the CPU path's value for the re-grayed plane."""
    for n, (_, y) in ref_images.items():
        yy = np.ascontiguousarray(np.dstack([y, y, y, np.full_like(y, 255)]))
        regray = ctx.gray1(yy)
        hip = ctx.sobel(yy)
        assert np.array_equal(hip, oracle.sobel_gray(regray)), n
        same = np.pad(regray == y, 1, mode="reflect")
        h, w = y.shape
        ok = np.ones((h, w), bool)
        for dy in range(3):
            for dx in range(3):
                ok &= same[dy:dy + h, dx:dx + w]
        assert np.array_equal(hip[ok], oracle.sobel_gray(np.ascontiguousarray(y))[ok]), n
        assert ok.mean() > 0.05, (n, ok.mean())


def test_hip_image2d_mode_reproduces_the_published_windows_numbers(ctx, pkg, oracle, ref_images):
    """The reference's Windows runs went through its image2d_t kernels (test_published_mae.py).  The product's
    image2d-mode entry (mi355_image2d_rgba8) in the OpenCL path's seat, the HIP buffer-mode grayscale /
    the oracle's CPU Sobel in the CPU path's: grayscale to the last printed digit, Sobel within 2e-5."""
    for n, (rgb, y) in ref_images.items():
        rgba = rgba_of(rgb)
        img_gray, _ = ctx.image2d(pkg.FILTER_GRAY, rgba)
        assert np.array_equal(img_gray, oracle.image2d_gray(rgba).reshape(img_gray.shape)), n
        assert printed(mae(ctx.gray1(rgba), img_gray)) == published("gray", n, "Windows"), n
        img_sobel, _ = ctx.image2d(pkg.FILTER_SOBEL, rgba)
        assert np.array_equal(img_sobel, oracle.image2d_sobel(rgba).reshape(img_sobel.shape)), n
        pub = published("sobel", n, "Windows")
        assert abs(mae(oracle.sobel_gray(np.ascontiguousarray(y)), img_sobel) - pub) / pub < 2e-5, n
