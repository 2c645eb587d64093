"""GPU suite (-m gpu): the HIP path, called through the C-ABI, against the oracle on the same seeded
inputs.  Bars: bit-exact for grayscale, Sobel, the EXACT Gaussian and the EXACT pipeline; |d| <= 1 LSB
per channel for the FAST (separable, FMA) Gaussian — the tolerance BASELINE.json's north_star states."""
import numpy as np
import pytest

from conftest import rand_rgba

pytestmark = pytest.mark.gpu

# the reference's own image sizes (images/*.jpg: 75x75, 240x?, 640x512, 1023x819 — widths that are not
# multiples of 4 or 64) plus degenerate and tile-boundary shapes.  (h, w)
SIZES = [(1, 1), (1, 7), (9, 1), (2, 2), (3, 5), (16, 64), (17, 65), (75, 75), (33, 248), (40, 252),
         (31, 256), (64, 500), (130, 1023)]


def test_context_reports_gfx950(ctx):
    assert "gfx950" in ctx.device_name


@pytest.mark.parametrize("h,w", SIZES)
def test_gray_bit_exact(ctx, oracle, h, w):
    img = rand_rgba(h, w, seed=h * 1000 + w, alpha=None)
    assert np.array_equal(ctx.gray(img), oracle.gray_rgba(img))
    assert np.array_equal(ctx.gray1(img), oracle.gray_rgba_1ch(img))


def test_gray_all_16m_colours(ctx, oracle):
    r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")
    rgba = np.stack([r, g, b, np.full_like(r, 255)], -1).astype(np.uint8).reshape(4096, 4096, 4)
    got = ctx.gray1(rgba)
    assert np.array_equal(got, oracle.gray_rgba_1ch(rgba))
    assert got.reshape(256, 256, 256)[0, 72, 24] == 44


def test_gray_fixture_config1(ctx, oracle, fixture_rgba):
    assert np.array_equal(ctx.gray1(fixture_rgba), oracle.gray_rgba_1ch(fixture_rgba))


@pytest.fixture(autouse=True)
def _reset_kernel_selection(request):
    """Tests that pin a kernel family (IMPL_TILE / IMPL_VALU) leave the session's context on AUTO."""
    yield
    if "ctx" in request.fixturenames:
        c, p = request.getfixturevalue("ctx"), request.getfixturevalue("pkg")
        c.set_impl(p.IMPL_AUTO)
        c.set_gauss_mode(p.GAUSS_FAST)


def _mfma_takes(h, w, n):
    """AUTO hands k >= 7 to the matrix-core kernel for launches worth its decomposition (csrc/gauss.hip)."""
    return w % 4 == 0 and w >= 64 and h * w * n >= (1 << 16)


def _gauss_check(ctx, pkg, oracle, img, k, sigma):
    ref = oracle.gauss_rgba(img, k, sigma)
    ctx.set_gauss_mode(pkg.GAUSS_EXACT)
    exact = ctx.gauss(img, k, sigma)
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    fast = ctx.gauss(img, k, sigma)
    assert np.array_equal(exact, ref), "EXACT mode must be bit-identical to the CPU path"
    d = np.abs(fast.astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= 1, "FAST mode tolerance is 1 LSB per channel (north_star)"
    return float((d != 0).mean())


@pytest.mark.parametrize("h,w", SIZES)
def test_gauss_5x5(ctx, pkg, oracle, h, w):
    img = rand_rgba(h, w, seed=h * 77 + w, alpha=None)
    frac = _gauss_check(ctx, pkg, oracle, img, 5, 1.5)
    if img.size >= 4096:  # a rate is meaningless on a handful of values
        assert frac < 0.01   # off-by-one only where the sum sits within float rounding of an integer


@pytest.mark.parametrize("k,sigma", [(1, 1.0), (3, 0.8), (7, 2.0), (9, 2.5), (17, 6.0), (31, 10.0)])
def test_gauss_other_kernels(ctx, pkg, oracle, k, sigma):
    img = rand_rgba(45, 83, seed=k, alpha=None)
    _gauss_check(ctx, pkg, oracle, img, k, sigma)


@pytest.mark.parametrize("k,sigma", [(3, 0.8), (5, 1.5), (7, 2.0), (9, 2.5)])
@pytest.mark.parametrize("h,w", [(1, 4), (3, 8), (40, 252), (33, 248), (7, 256), (131, 500), (300, 1920), (5, 3840)])
def test_gauss_sliding_window_kernel_equals_tiled_kernel(ctx, pkg, oracle, k, sigma, h, w):
    """The register-resident kernel (gauss_slide.hip; width % 4 == 0, k <= 9) and the LDS-tiled kernel
    implement one canonical FAST arithmetic: identical bits.  Both are within 1 LSB of the CPU path.  (IMPL_VALU pins
    the kernel: under AUTO, k >= 7 launches big enough for it go to the matrix cores — tests/test_gpu_mfma.py.)"""
    img = rand_rgba(h, w, seed=h * 7 + w + k, alpha=None)
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    ctx.set_impl(pkg.IMPL_TILE)
    tiled = ctx.gauss(img, k, sigma)
    ctx.set_impl(pkg.IMPL_VALU)
    slide = ctx.gauss(img, k, sigma)
    assert np.array_equal(slide, tiled)
    ref = oracle.gauss_rgba(img, k, sigma)
    assert np.abs(slide.astype(np.int16) - ref.astype(np.int16)).max() <= 1


@pytest.mark.parametrize("k,sigma", [(11, 3.0), (13, 3.3), (15, 4.0), (17, 6.0)])
@pytest.mark.parametrize("h,w", [(1, 2), (3, 6), (30, 100), (61, 112), (75, 114), (40, 252), (200, 640), (9, 3840)])
def test_gauss_wide_kernel_equals_tiled_kernel(ctx, pkg, oracle, k, sigma, h, w):
    """k = 11..17 (17/6 is the reference ProgramHandler's default): every FAST kernel is within 1 LSB of the CPU path.
    gauss_wide.hip (width % 2 == 0; IMPL_VALU pins it) additionally gives the LDS-tiled kernel's bits; under AUTO big
    launches of 4-pixel-multiple rows go to the matrix-core kernel, which rounds differently (tests/test_gpu_mfma.py)."""
    img = rand_rgba(h, w, seed=h * 7 + w + k, alpha=None)
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    ctx.set_impl(pkg.IMPL_TILE)
    tiled = ctx.gauss(img, k, sigma)
    ctx.set_impl(pkg.IMPL_VALU)
    wide = ctx.gauss(img, k, sigma)
    assert np.array_equal(wide, tiled)
    ref = oracle.gauss_rgba(img, k, sigma, threads=8)
    assert np.abs(wide.astype(np.int16) - ref.astype(np.int16)).max() <= 1
    ctx.set_impl(pkg.IMPL_AUTO)
    auto = ctx.gauss(img, k, sigma)
    assert np.abs(auto.astype(np.int16) - ref.astype(np.int16)).max() <= 1
    if not _mfma_takes(h, w, 1):
        assert np.array_equal(auto, tiled)


def test_gauss_wide_kernel_opaque_fast_path_and_fallback(ctx, pkg, oracle):
    h, w = 200, 300
    base = oracle.synth_rgba(w, h, 1, first_frame=3, mode=1)[0]   # A = 255
    for pos in (None, (0, 0), (199, 299), (100, 111), (100, 112), (137, 5), (64, 223)):
        img = base.copy()
        if pos is not None:
            img[pos[0], pos[1], 3] = 200
        for k, sigma in ((11, 3.0), (17, 6.0)):
            ctx.set_impl(pkg.IMPL_TILE)
            tiled = ctx.gauss(img, k, sigma)
            ctx.set_impl(pkg.IMPL_VALU)
            assert np.array_equal(ctx.gauss(img, k, sigma), tiled), (pos, k)


@pytest.mark.parametrize("k,sigma", [(5, 1.5), (7, 2.0), (9, 2.5), (11, 3.0), (17, 6.0)])
def test_gauss_batch_with_opaque_and_non_opaque_frames(ctx, pkg, oracle, k, sigma):
    """k >= 7 runs the opaque pass and the general pass as two kernels that talk through one flag per work item
    (frame, band, strip): a batch in which only some frames / bands carry a non-opaque pixel must come out as
    the tiled kernel computes it, frame by frame."""
    frames = oracle.synth_rgba(602, 330, 5, first_frame=k, mode=1).copy()   # A = 255; width % 4 != 0: VALU kernels
    frames[1, 17, 33, 3] = 0
    frames[3, 329, 599, 3] = 254
    frames[3, 150, 300, 3] = 1
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    ctx.set_impl(pkg.IMPL_TILE)
    tiled = ctx.gauss(frames, k, sigma)
    ctx.set_impl(pkg.IMPL_VALU)
    got = ctx.gauss(frames, k, sigma)
    assert np.array_equal(got, tiled)
    # the untouched frames equal what they give alone (no flag leaks across frames)
    assert np.array_equal(got[0], ctx.gauss(frames[0], k, sigma))
    assert np.array_equal(got[4], ctx.gauss(frames[4], k, sigma))


def test_gauss_wide_kernel_batched_multi_band(ctx, pkg, oracle):
    frames = oracle.synth_rgba(1002, 420, 3, first_frame=1, mode=1)   # width % 4 != 0: gauss_wide.hip
    ctx.set_impl(pkg.IMPL_TILE)
    tiled = ctx.gauss(frames, 17, 6.0)
    ctx.set_impl(pkg.IMPL_VALU)
    assert np.array_equal(ctx.gauss(frames, 17, 6.0), tiled)
    frames = oracle.synth_rgba(1000, 420, 3, first_frame=1, mode=1)   # width % 4 == 0: AUTO = matrix cores
    ctx.set_impl(pkg.IMPL_AUTO)
    got = ctx.gauss(frames, 17, 6.0)
    for f in range(3):
        assert np.abs(got[f].astype(np.int16) - oracle.gauss_rgba(frames[f], 17, 6.0, threads=8).astype(np.int16)).max() <= 1


@pytest.mark.parametrize("k,sigma", [(3, 0.8), (5, 1.5), (7, 2.0), (9, 2.5)])
def test_gauss_opaque_fast_path_and_its_fallback(ctx, pkg, oracle, k, sigma):
    """The sliding-window kernel skips the alpha arithmetic while every alpha byte a wave sees is 255 and redoes
    the band in full otherwise.  Opaque frames, frames with ONE non-opaque pixel (at band / strip / halo
    boundaries and in the image corners), and a non-opaque frame must all equal the tiled kernel bit for bit
    and the CPU path within 1 LSB (alpha included: a blurred all-255 channel is 254 or 255)."""
    h, w = 300, 512   # several bands (tall and tail), 3 strips
    base = oracle.synth_rgba(w, h, 1, first_frame=k, mode=1)[0]   # A = 255 everywhere
    cases = [None, (0, 0), (0, w - 1), (h - 1, 0), (h - 1, w - 1), (1, 239), (150, 240), (151, 243), (127, 100),
             (128, 100), (269, 300), (270, 511), (299, 256)]
    ref_opaque = None
    for pos in cases:
        img = base.copy()
        if pos is not None:
            img[pos[0], pos[1], 3] = 7
        ctx.set_gauss_mode(pkg.GAUSS_FAST)
        ctx.set_impl(pkg.IMPL_TILE)
        tiled = ctx.gauss(img, k, sigma)
        ctx.set_impl(pkg.IMPL_VALU)
        slide = ctx.gauss(img, k, sigma)
        assert np.array_equal(slide, tiled), pos
        if pos is None:
            ref_opaque = slide
            assert set(np.unique(slide[..., 3]).tolist()) <= {254, 255}
            ref = oracle.gauss_rgba(img, k, sigma)
            assert np.abs(slide.astype(np.int16) - ref.astype(np.int16)).max() <= 1
        else:
            # far from the transparent pixel nothing changes
            far = np.ones((h, w), bool)
            far[max(0, pos[0] - k):pos[0] + k + 1, max(0, pos[1] - k):pos[1] + k + 1] = False
            assert np.array_equal(slide[far], ref_opaque[far]), pos
    noisy = rand_rgba(h, w, seed=k, alpha=None)
    ctx.set_impl(pkg.IMPL_TILE)
    tiled = ctx.gauss(noisy, k, sigma)
    ctx.set_impl(pkg.IMPL_VALU)
    assert np.array_equal(ctx.gauss(noisy, k, sigma), tiled)


@pytest.mark.parametrize("k,sigma", [(3, 0.8), (5, 1.5), (7, 2.0), (9, 2.5)])
def test_gauss_constant_and_piecewise_constant_alpha(ctx, pkg, oracle, k, sigma):
    """The 3-channel pass also serves alpha that is constant but not 255, and alpha that is piecewise constant (a
    matte: regions of different values): while the K rows of a window carry one value A over the whole strip the
    output alpha is the host-evaluated byte of an all-A window; rows with mixed alphas and windows that span two values
    send the band to the 4-channel pass.  Whatever the route, the bytes are the tiled kernel's (which computes alpha
    like any other channel), and within 1 LSB of the CPU path."""
    h, w = 300, 512
    base = oracle.synth_rgba(w, h, 1, first_frame=k + 40, mode=0)[0]
    cases = {}
    for a in (0, 1, 128, 200, 254):
        img = base.copy()
        img[..., 3] = a
        cases["const %d" % a] = img
    img = base.copy()
    img[h // 2:, :, 3] = 128                 # horizontal edge: changes inside a band, between bands, at a band boundary
    cases["top 255 / bottom 128"] = img
    img = base.copy()
    img[:24, :, 3] = 7                       # exactly the first k=5 band
    img[24:25, :, 3] = 9                     # one row
    img[200:, :, 3] = 0
    cases["stripes"] = img
    img = base.copy()
    img[:, :240, 3] = 33                     # vertical edge on a strip boundary (60 lanes x 4 px)
    cases["left 33 / right 255 at x=240"] = img
    img = base.copy()
    img[:, :243, 3] = 33                     # vertical edge inside a strip
    cases["left 33 / right 255 at x=243"] = img
    img = base.copy()
    img[100:200, 100:300, 3] = 64            # a block
    img[120:130, 120:130, 3] = 255           # with a hole
    cases["block"] = img
    img = base.copy()
    img[..., 3] = (np.arange(h)[:, None] // 3 + np.zeros((1, w), int)).astype(np.uint8)   # a new value every 3 rows
    cases["ramp by rows"] = img
    for name, img in cases.items():
        ctx.set_gauss_mode(pkg.GAUSS_FAST)
        ctx.set_impl(pkg.IMPL_TILE)
        tiled = ctx.gauss(img, k, sigma)
        ctx.set_impl(pkg.IMPL_VALU)
        slide = ctx.gauss(img, k, sigma)
        assert np.array_equal(slide, tiled), name
        ref = oracle.gauss_rgba(img, k, sigma)
        assert np.abs(slide.astype(np.int16) - ref.astype(np.int16)).max() <= 1, name
    # a batch: frames with different constants, one mixed
    frames = np.stack([cases["const 128"], cases["const 0"], cases["block"], base])
    ctx.set_impl(pkg.IMPL_TILE)
    tiled = ctx.gauss(frames, k, sigma)
    ctx.set_impl(pkg.IMPL_VALU)
    assert np.array_equal(ctx.gauss(frames, k, sigma), tiled)
    ctx.set_impl(pkg.IMPL_AUTO)


@pytest.mark.parametrize("k,sigma", [(3, 0.8), (5, 1.5), (9, 2.5)])
@pytest.mark.parametrize("h,w", [(1, 1), (2, 3), (7, 5), (75, 75), (40, 249), (33, 251), (64, 253), (131, 501),
                                 (300, 1023), (6, 3841)])
def test_gauss_sliding_window_ragged_widths(ctx, pkg, oracle, k, sigma, h, w):
    """Widths that are not multiples of 4 (the reference's own images are 75x75 and 1023x819): the RAGGED
    variant of the sliding-window kernel — unaligned rows, partial last quad — equals the tiled kernel."""
    for alpha in (255, None):
        img = rand_rgba(h, w, seed=h + w + k, alpha=alpha)
        ctx.set_gauss_mode(pkg.GAUSS_FAST)
        ctx.set_impl(pkg.IMPL_TILE)
        tiled = ctx.gauss(img, k, sigma)
        ctx.set_impl(pkg.IMPL_VALU)
        slide = ctx.gauss(img, k, sigma)
        assert np.array_equal(slide, tiled)
    if h * w <= 40000:
        ref = oracle.gauss_rgba(img, k, sigma)
        assert np.abs(slide.astype(np.int16) - ref.astype(np.int16)).max() <= 1


def test_gauss_ragged_batch_of_odd_frames(ctx, pkg, oracle):
    """Frame stride not a multiple of 16 bytes: frames 1.. start unaligned."""
    frames = oracle.synth_rgba(1023, 131, 3, first_frame=4, mode=1)
    ctx.set_impl(pkg.IMPL_TILE)
    tiled = ctx.gauss(frames, 5, 1.5)
    ctx.set_impl(pkg.IMPL_VALU)
    assert np.array_equal(ctx.gauss(frames, 5, 1.5), tiled)


def test_gauss_sliding_window_batched_multi_band(ctx, pkg, oracle):
    """Several frames, several bands per frame (h > 128), several strips (w > 248), edge strips."""
    frames = oracle.synth_rgba(1000, 300, 3, first_frame=1, mode=1)
    ctx.set_impl(pkg.IMPL_TILE)
    tiled = ctx.gauss(frames, 5, 1.5)
    ctx.set_impl(pkg.IMPL_VALU)
    slide = ctx.gauss(frames, 5, 1.5)
    assert np.array_equal(slide, tiled)
    ref = oracle.gauss_rgba(frames[2], 5, 1.5)
    d = np.abs(slide[2].astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 0.01


def test_gauss_known_answers(ctx, pkg):
    ctx.set_gauss_mode(pkg.GAUSS_EXACT)
    assert np.unique(ctx.gauss(np.full((40, 70, 4), 255, np.uint8), 5, 1.5)).tolist() == [254]
    assert np.unique(ctx.gauss(np.full((40, 70, 4), 200, np.uint8), 5, 1.5)).tolist() == [200]
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    assert set(np.unique(ctx.gauss(np.full((40, 70, 4), 255, np.uint8), 5, 1.5)).tolist()) <= {254, 255}


def test_gauss_photographic_fixture(ctx, pkg, oracle, fixture_rgba):
    frac = _gauss_check(ctx, pkg, oracle, fixture_rgba, 5, 1.5)
    assert frac < 0.01


def test_gauss_external_weight_table(ctx, pkg, oracle):
    """Multi-GPU mode installs a broadcast table instead of generating it."""
    img = rand_rgba(30, 50, seed=9)
    table = oracle.gauss_weights(5, 1.5)
    ctx.set_gauss_weights(5, 1.5, table)
    ctx.set_gauss_mode(pkg.GAUSS_EXACT)
    assert np.array_equal(ctx.gauss(img, 5, 1.5), oracle.gauss_rgba(img, 5, weights=table))
    ctx.set_gauss_mode(pkg.GAUSS_FAST)


@pytest.mark.parametrize("h,w", SIZES)
def test_sobel_bit_exact(ctx, oracle, h, w):
    img = rand_rgba(h, w, seed=h * 31 + w)
    assert np.array_equal(ctx.sobel(img), oracle.sobel_rgba(img))


@pytest.mark.parametrize("h,w", [(1, 4), (2, 8), (3, 12), (40, 252), (33, 248), (7, 256), (131, 500), (300, 1920),
                                 (5, 3840), (1, 1), (1, 2), (2, 3), (9, 5), (75, 75), (40, 249), (33, 251),
                                 (64, 253), (131, 501), (300, 1023), (4, 3841)])
def test_sobel_sliding_window_kernel(ctx, pkg, oracle, h, w):
    """sobel_slide.hip (any width; RAGGED variant when width % 4 != 0) against the oracle and against the
    LDS-tiled kernel."""
    for seed, mode in ((h + w, None), (7, 1)):
        img = rand_rgba(h, w, seed=seed) if mode is None else oracle.synth_rgba(w, h, 1, first_frame=seed, mode=1)[0]
        ctx.set_impl(pkg.IMPL_TILE)
        tiled = ctx.sobel(img)
        ctx.set_impl(pkg.IMPL_AUTO)
        slide = ctx.sobel(img)
        assert np.array_equal(slide, tiled)
        assert np.array_equal(slide, oracle.sobel_rgba(img))


def test_sobel_all_colours_through_the_integer_luma_path(ctx, oracle):
    """The sliding kernel's luminance is integer arithmetic + an FP64 path for ambiguous colours: run all
    2^24 colours through it (a 4096x4096 frame whose pixel (i, j) has colour index i*4096+j)."""
    r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")
    rgba = np.stack([r, g, b, np.full_like(r, 255)], -1).astype(np.uint8).reshape(4096, 4096, 4)
    assert np.array_equal(ctx.sobel(rgba), oracle.sobel_rgba(rgba))


def test_sobel_known_answers(ctx):
    flat = np.full((20, 70, 4), 90, np.uint8)
    assert ctx.sobel(flat).max() == 0
    step = np.zeros((20, 70, 4), np.uint8)
    step[:, 35:, :3] = 255
    out = ctx.sobel(step)
    assert (out[:, [34, 35]] == 255).all() and out[:, :34].max() == 0 and out[:, 36:].max() == 0


def test_sobel_smooth_frames(ctx, oracle):
    img = oracle.synth_rgba(333, 97, 1, mode=1)[0]
    assert np.array_equal(ctx.sobel(img), oracle.sobel_rgba(img))


@pytest.mark.parametrize("h,w", SIZES)
def test_pipeline(ctx, pkg, oracle, h, w):
    img = rand_rgba(h, w, seed=h * 13 + w)
    k, sigma = 5, 1.5
    ctx.set_gauss_mode(pkg.GAUSS_EXACT)
    assert np.array_equal(ctx.pipeline(img, k, sigma), oracle.pipeline_rgba(img, k, sigma))
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    fused = ctx.pipeline(img, k, sigma)
    if w >= 4 and h >= 2:
        # the sliding-window kernel is exact by exception: the CPU chain's bytes in FAST mode too
        assert np.array_equal(fused, oracle.pipeline_rgba(img, k, sigma))
    # the three FAST calls chained carry the FAST Gaussian's <= 1 LSB through Sobel: <= 6 grey levels
    chained = ctx.sobel(ctx.gauss(ctx.gray(img), k, sigma))
    assert np.abs(fused.astype(int) - chained.astype(int)).max() <= 6


@pytest.mark.parametrize("k,sigma", [(3, 0.8), (5, 1.5), (7, 2.0)])
@pytest.mark.parametrize("h,w", [(2, 4), (3, 8), (5, 12), (40, 252), (70, 248), (67, 256), (131, 500), (200, 1920),
                                 (9, 3840), (2, 5), (3, 6), (5, 7), (75, 75), (40, 249), (33, 250), (64, 251),
                                 (131, 501), (200, 1023), (4, 3841)])
def test_pipeline_sliding_window_kernel(ctx, pkg, oracle, k, sigma, h, w):
    """pipe_slide.hip (k <= 7, any width >= 4, h >= 2; both Gaussian modes) is bit-exact with the CPU chain; the
    LDS-tiled fused kernel is bit-exact in EXACT mode and, in FAST mode, equal to the three FAST calls chained and
    within the FAST tolerance of the CPU chain (a 1-LSB difference in one blurred pixel moves gx, gy by at most 4
    each: |d magnitude| <= 6)."""
    img = oracle.synth_rgba(w, h, 1, first_frame=h + k, mode=(h + w) & 1)[0]
    ref = oracle.pipeline_rgba(img, k, sigma)
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    ctx.set_impl(pkg.IMPL_TILE)
    tiled = ctx.pipeline(img, k, sigma)
    chained_tiled = ctx.sobel(ctx.gauss(ctx.gray(img), k, sigma))
    ctx.set_gauss_mode(pkg.GAUSS_EXACT)
    tiled_exact = ctx.pipeline(img, k, sigma)
    ctx.set_impl(pkg.IMPL_AUTO)
    slide_exact_mode = ctx.pipeline(img, k, sigma)
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    slide = ctx.pipeline(img, k, sigma)
    assert np.array_equal(slide, ref)
    assert np.array_equal(slide_exact_mode, ref)
    assert np.array_equal(tiled_exact, ref)
    assert np.array_equal(tiled, chained_tiled)
    assert np.abs(tiled.astype(int) - ref.astype(int)).max() <= 6
    assert (tiled != ref).mean() < 0.05


def test_pipeline_sliding_window_batched_multi_band(ctx, pkg, oracle):
    frames = oracle.synth_rgba(1000, 300, 3, first_frame=2, mode=1)
    got = ctx.pipeline(frames, 5, 1.5)
    for f in range(3):
        assert np.array_equal(got[f], oracle.pipeline_rgba(frames[f], 5, 1.5))
    # flat regions take the exact chain for every pixel: constant and two-level frames
    flat = np.full((2, 97, 260, 4), 255, np.uint8)
    flat[1, :, 130:, :3] = 17
    got = ctx.pipeline(flat, 5, 1.5)
    for f in range(2):
        assert np.array_equal(got[f], oracle.pipeline_rgba(flat[f], 5, 1.5))


def test_pipeline_other_kernels(ctx, pkg, oracle):
    img = oracle.synth_rgba(150, 70, 1, mode=1)[0]
    for k, sigma in ((3, 0.8), (9, 2.5), (17, 6.0)):
        ctx.set_gauss_mode(pkg.GAUSS_EXACT)
        assert np.array_equal(ctx.pipeline(img, k, sigma), oracle.pipeline_rgba(img, k, sigma))
        ctx.set_gauss_mode(pkg.GAUSS_FAST)
        if k <= 7:   # sliding-window kernel: the CPU chain's bytes
            assert np.array_equal(ctx.pipeline(img, k, sigma), oracle.pipeline_rgba(img, k, sigma))
        else:        # tiled FAST kernel: the three FAST calls chained
            assert np.array_equal(ctx.pipeline(img, k, sigma), ctx.sobel(ctx.gauss(ctx.gray(img), k, sigma)))


def test_batched_equals_per_frame(ctx, oracle):
    frames = oracle.synth_rgba(120, 45, 5, first_frame=2)
    for name in ("gray", "sobel"):
        batched = getattr(ctx, name)(frames)
        for f in range(5):
            assert np.array_equal(batched[f], getattr(ctx, name)(frames[f]))
    g = ctx.gauss(frames, 5, 1.5)
    p = ctx.pipeline(frames, 5, 1.5)
    for f in range(5):
        assert np.array_equal(g[f], ctx.gauss(frames[f], 5, 1.5))
        assert np.array_equal(p[f], ctx.pipeline(frames[f], 5, 1.5))


def test_streamed_host_path_equals_batched(ctx, pkg, oracle):
    """mi355_filter_stream (pinned or pageable host memory, several chunks, ragged last chunk)."""
    frames = oracle.synth_rgba(320, 90, 11, first_frame=1, mode=1)
    for filt, name, args in ((pkg.FILTER_GAUSS, "gauss", (5, 1.5)), (pkg.FILTER_SOBEL, "sobel", ()),
                             (pkg.FILTER_PIPELINE, "pipeline", (5, 1.5)), (pkg.FILTER_GRAY, "gray", ())):
        want = getattr(ctx, name)(frames, *args)
        k, sigma = args if args else (0, 0.0)
        for chunk in (1, 4, 0):
            got, ms = ctx.stream(filt, frames, k=k, sigma=sigma, chunk_frames=chunk)
            assert np.array_equal(got, want) and ms > 0
    pin_in = ctx.pinned_empty(frames.shape)
    pin_out = ctx.pinned_empty(frames.shape)
    pin_in[...] = frames
    got, _ = ctx.stream(pkg.FILTER_GAUSS, pin_in, out=pin_out, k=5, sigma=1.5, chunk_frames=3)
    assert np.array_equal(got, ctx.gauss(frames, 5, 1.5))
    ctx.pinned_free(pin_in)
    ctx.pinned_free(pin_out)


def test_bgr_input_format(ctx, pkg, oracle, fixture_rgb):
    """3-byte BGR frames (what cv::imread hands the reference): the device performs cvtColor(BGR2RGBA)."""
    rng = np.random.default_rng(3)
    for (h, w) in ((1, 1), (5, 7), (33, 250), (64, 256)):
        bgr = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        rgba = np.concatenate([bgr[..., ::-1], np.full((2, h, w, 1), 255, np.uint8)], axis=-1)
        ctx.set_input_format(pkg.INPUT_BGR)
        try:
            g1, gs, ga, pp = ctx.gray1(bgr), ctx.sobel(bgr), ctx.gauss(bgr, 5, 1.5), ctx.pipeline(bgr, 5, 1.5)
            st, _ = ctx.stream(pkg.FILTER_GAUSS, bgr, k=5, sigma=1.5, chunk_frames=1)
        finally:
            ctx.set_input_format(pkg.INPUT_RGBA)
        for f in range(2):
            assert np.array_equal(g1[f], oracle.gray_bgr(bgr[f]))        # the reference CPU path's own layout
        assert np.array_equal(gs, ctx.sobel(rgba))
        assert np.array_equal(ga, ctx.gauss(rgba, 5, 1.5))
        assert np.array_equal(pp, ctx.pipeline(rgba, 5, 1.5))
        assert np.array_equal(st, ga)
    # BASELINE.json config 1 through the GPU: grayscale of the Tulips fixture in its native BGR layout
    bgr = np.ascontiguousarray(fixture_rgb[..., ::-1])
    ctx.set_input_format(pkg.INPUT_BGR)
    try:
        got = ctx.gray1(bgr)
    finally:
        ctx.set_input_format(pkg.INPUT_RGBA)
    assert np.array_equal(got, oracle.gray_bgr(bgr))


def test_per_frame_entry_points_and_profiling_contract(ctx, oracle):
    """mi355_*_rgba8: six timestamps, write/kernel/read, non-decreasing (Controller.cpp:66-74)."""
    img = rand_rgba(64, 96, seed=4)
    out, prof = ctx.single("gray", img)
    assert np.array_equal(out, oracle.gray_rgba(img))
    assert len(prof) == 6 and all(b >= a for a, b in zip(prof, prof[1:])) and prof[5] > prof[0]
    out, prof = ctx.single("sobel", img)
    assert np.array_equal(out, oracle.sobel_rgba(img))
    out, prof = ctx.single("gauss", img, 5, 1.5)
    assert np.abs(out.astype(int) - oracle.gauss_rgba(img, 5, 1.5).astype(int)).max() <= 1
    out, _ = ctx.single("gray1", img)
    assert np.array_equal(out, oracle.gray_rgba_1ch(img))


def test_synth_and_checksum_match_cpu_twins(ctx, oracle):
    w, h, n = 251, 67, 3
    for mode in (0, 1):
        d = ctx.alloc(w * h * n * 4)
        ctx.synth_dev(d, w, h, n, first_frame=3, seed=0x5EED, mode=mode)
        got = np.empty((n, h, w, 4), np.uint8)
        ctx.d2h(got, d)
        ref = oracle.synth_rgba(w, h, n, first_frame=3, seed=0x5EED, mode=mode)
        assert np.array_equal(got, ref)
        assert ctx.checksum_dev(d, got.nbytes) == oracle.checksum(ref)
        per = w * h
        parts = sum(ctx.checksum_dev(d + f * per * 4, per * 4, index_base=f * per) for f in range(n))
        assert parts % (1 << 64) == oracle.checksum(ref)
        ctx.free(d)


def test_device_resident_api(ctx, pkg, oracle):
    w, h, n = 256, 40, 3
    frames = oracle.synth_rgba(w, h, n)
    d_in = ctx.alloc(frames.nbytes)
    d_out = ctx.alloc(frames.nbytes)
    ctx.h2d(d_in, frames)
    ctx.filter_dev(pkg.FILTER_SOBEL, d_in, d_out, w, h, n)
    got = np.empty((n, h, w), np.uint8)
    ctx.d2h(got, d_out)
    for f in range(n):
        assert np.array_equal(got[f], oracle.sobel_rgba(frames[f]))
    ctx.free(d_in)
    ctx.free(d_out)


def test_bad_arguments_are_rejected_not_launched(ctx, pkg):
    img = rand_rgba(8, 8, seed=1)
    for k, sigma in ((4, 1.0), (0, 1.0), (65, 1.0), (5, 0.0), (5, float("inf"))):
        with pytest.raises(pkg.Mi355Error):
            ctx.gauss(img, k, sigma)
    with pytest.raises(pkg.Mi355Error):
        ctx.filter_dev(pkg.FILTER_GRAY, 0, 0, 8, 8, 1)
    with pytest.raises(pkg.Mi355Error):
        ctx.filter_dev(42, 16, 16, 8, 8, 1)
    assert np.array_equal(ctx.gray(img)[..., 3], np.full((8, 8), 255, np.uint8))   # still usable


def test_random_shapes_all_kernels_agree(ctx, pkg, oracle):
    """Seeded random (frames, h, w, k) shapes — band/strip/tail boundaries of the sliding-window work
    decomposition are data dependent — every kernel pair must agree and match the oracle."""
    rng = np.random.default_rng(20261004)
    for case in range(36):
        n = int(rng.integers(1, 4))
        w = int(rng.integers(1, 130)) * 4 if case % 3 else int(rng.integers(1, 520))
        h = int(rng.integers(1, 700)) if case % 4 else int(rng.integers(1, 40))
        k = int(rng.choice([3, 5, 7, 9]))
        sigma = float(rng.uniform(0.6, 3.0))
        frames = oracle.synth_rgba(w, h, n, first_frame=case, seed=case + 1, mode=case & 1)
        ctx.set_gauss_mode(pkg.GAUSS_FAST)
        ctx.set_impl(pkg.IMPL_TILE)
        g_t, s_t, p_t = ctx.gauss(frames, k, sigma), ctx.sobel(frames), ctx.pipeline(frames, k, sigma)
        ctx.set_impl(pkg.IMPL_VALU)
        g_v = ctx.gauss(frames, k, sigma)
        ctx.set_impl(pkg.IMPL_AUTO)
        g_a, s_a, p_a = ctx.gauss(frames, k, sigma), ctx.sobel(frames), ctx.pipeline(frames, k, sigma)
        tag = (case, n, h, w, k)
        assert np.array_equal(g_v, g_t), tag
        if k >= 7 and _mfma_takes(h, w, n):   # AUTO = matrix cores: other rounding, same contract
            assert np.abs(g_a.astype(np.int16) - g_t.astype(np.int16)).max() <= 2, tag
        else:
            assert np.array_equal(g_a, g_t), tag
        assert np.array_equal(s_a, s_t), tag
        f = int(rng.integers(0, n))
        if k <= 7 and w >= 4 and h >= 2:
            assert np.array_equal(p_a[f], oracle.pipeline_rgba(frames[f], k, sigma)), tag
            assert np.abs(p_a.astype(int) - p_t.astype(int)).max() <= 6, tag
        else:
            assert np.array_equal(p_a, p_t), tag
        assert np.array_equal(s_a[f], oracle.sobel_rgba(frames[f])), tag
        assert np.array_equal(ctx.gray(frames)[f], oracle.gray_rgba(frames[f])), tag
        if h * w <= 120000:
            ref = oracle.gauss_rgba(frames[f], k, sigma)
            assert np.abs(g_a[f].astype(np.int16) - ref.astype(np.int16)).max() <= 1, tag


def test_8k_frame_many_strips(ctx, pkg, oracle):
    """One 7680x4320 frame (33 strips, 45+ bands): whole-frame oracle comparison for gray and Sobel, the
    Gaussian on three horizontal slabs (the CPU path needs ~1 s per 30 Mpixel)."""
    w, h = 7680, 4320
    frame = oracle.synth_rgba(w, h, 1, first_frame=11, mode=1)[0]
    assert np.array_equal(ctx.gray1(frame), oracle.gray_rgba_1ch(frame))
    assert np.array_equal(ctx.sobel(frame), oracle.sobel_rgba(frame))
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    g = ctx.gauss(frame, 5, 1.5)
    for y0, y1 in ((0, 40), (2100, 2140), (4280, 4320)):
        a, b = max(0, y0 - 2), min(h, y1 + 2)
        ref = oracle.gauss_rgba(frame[a:b], 5, 1.5)[y0 - a:y1 - a]
        assert np.abs(g[y0:y1].astype(np.int16) - ref.astype(np.int16)).max() <= 1
    assert np.array_equal(ctx.pipeline(frame, 5, 1.5), oracle.pipeline_rgba(frame, 5, 1.5))


# ---- BASELINE.json full sizes: size-independent properties ------------------------------------
def test_full_size_4k_properties(ctx, pkg, oracle):
    """4K frames, properties that hold at any size (the whole-frame oracle comparisons of the 4K and 1080p
    configurations are in tests/test_gpu_configs.py): fused == chained, a constant frame stays constant, shifting
    the content shifts the interior of the result."""
    w, h = 3840, 2160
    frame = oracle.synth_rgba(w, h, 1, first_frame=7, mode=1)[0]
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    gauss = ctx.gauss(frame, 5, 1.5)
    const = np.full((h, w, 4), 200, np.uint8)
    flat = np.unique(ctx.gauss(const, 5, 1.5)).tolist()
    assert len(flat) == 1 and flat[0] in (199, 200)   # FAST: one value everywhere, within 1 LSB of 200
    ctx.set_gauss_mode(pkg.GAUSS_EXACT)
    assert np.unique(ctx.gauss(const[:256], 5, 1.5)).tolist() == [200]   # EXACT: the CPU path's value
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    assert ctx.sobel(const).max() == 0
    assert ctx.pipeline(const, 5, 1.5).max() == 0
    shifted = np.roll(frame, 64, axis=1)
    g2 = ctx.gauss(shifted, 5, 1.5)
    assert np.array_equal(g2[:, 64 + 2:-2], gauss[:, 2:-64 - 2])
    s1, s2 = ctx.sobel(frame), ctx.sobel(shifted)
    assert np.array_equal(s2[:, 64 + 1:-1], s1[:, 1:-64 - 1])


@pytest.mark.gpu
def test_device_selftest_fast_arithmetic(ctx):
    """All 2^24 colours (fast luminance == FP64 formula, src/Grayscale/grayscale.cpp:237) and all 1021^2 gradient
    pairs (v_sqrt_f32 + v_cvt_pk_u8_f32 == round-half-even + saturate, src/EdgeDetection/EdgeDetection.cpp:236-240),
    checked on the device itself."""
    assert ctx.selftest() == (0, 0)


_BAND_PLAN_SCRIPT = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as e
pkg = e.load_package()
rng = np.random.default_rng(99)
h = hashlib.sha256()
with pkg.Context(0) as ctx:
    for (n, hh, ww) in [(3, 131, 500), (2, 97, 1023), (1, 300, 252), (2, 70, 4), (1, 33, 260), (2, 40, 1024)]:
        x = rng.integers(0, 256, (n, hh, ww, 4), dtype=np.uint8)
        y = x.copy(); y[..., 3] = 255
        for img in (x, y):
            for k, s in [(3, 0.8), (5, 1.5), (7, 2.0), (9, 2.5)]:
                h.update(ctx.gauss(img, k, s).tobytes())
            for k, s in [(3, 0.8), (5, 1.5), (7, 2.0)]:
                h.update(ctx.pipeline(img, k, s).tobytes())
            h.update(ctx.sobel(img).tobytes())
            h.update(ctx.gray1(img).tobytes())
            h.update(ctx.gray(img).tobytes())
print(h.hexdigest())
"""


def test_results_do_not_depend_on_the_band_plan():
    """The sliding-window kernels cut frames into bands whose height is a tuning choice (slide_common.hpp) and
    the Sobel kernel walks odd bands upward: outputs must be the same bytes for every plan.  The plan is fixed per
    process (MI355_TUNE_* are read once), hence one short child process per plan.  Only the tune build of the
    library (csrc/Makefile `make tune`, -DMI355_TUNE_ENV) reads those variables; the first digest comes from the
    product library itself, so the two builds are compared too."""
    import os
    import subprocess
    import sys
    import __graft_entry__ as entry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tune_lib = os.path.join(entry.ROOT, "tools", "lib", "libmi355_imgfilter_tune.so")
    assert os.path.exists(tune_lib), "run __graft_entry__.build()"
    digests = {}
    for plan in ({}, {"MI355_TUNE_BAND_ROWS": "1"}, {"MI355_TUNE_BAND_ROWS": "7"},
                 {"MI355_TUNE_BAND_ROWS": "50", "MI355_TUNE_TAIL_ROWS": "3", "MI355_TUNE_TAIL_FRAC": "0.3"},
                 # kernels that normally serve big batches only: Sobel aligned strips, gray strips
                 {"MI355_TUNE_SOBEL_STRIP": "2", "MI355_TUNE_BAND_ROWS": "5"}, {"MI355_TUNE_SOBEL_STRIP": "2"},
                 {"MI355_TUNE_SOBEL_STRIP": "0", "MI355_TUNE_LANES_OUT": "48"}, {"MI355_TUNE_GRAY_STRIP": "3"},
                 {"MI355_TUNE_GRAY_STRIP": "0"}):
        env = dict(os.environ, **plan)
        env.pop("MI355_IMGFILTER_LIB", None)
        if plan:
            env["MI355_IMGFILTER_LIB"] = tune_lib
        out = subprocess.run([sys.executable, "-c", _BAND_PLAN_SCRIPT, root], env=env, capture_output=True,
                             text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests[tuple(sorted(plan.items()))] = out.stdout.strip().splitlines()[-1]
    assert len(set(digests.values())) == 1, digests


def test_big_batch_kernels(ctx, pkg, oracle):
    """Batches of >= 2^28 pixels of 4-pixel-multiple rows take gray.hip's strip-walk kernel and sobel_slide.hip's
    aligned-strip kernel, smaller ones the flat / halo-lane kernels: the same 36 x 4K frames as one call and as 9
    calls of 4 frames must give the same bytes (checksums add up over the word index), and the first and last
    frame must equal the oracle."""
    w, h, n = 3840, 2160, 36
    per = w * h
    d_in = ctx.alloc(per * n * 4)
    ctx.synth_dev(d_in, w, h, n, first_frame=0, seed=0x5EED, mode=0)
    for filt, bpp in ((pkg.FILTER_GRAY, 4), (pkg.FILTER_GRAY1, 1), (pkg.FILTER_SOBEL, 1)):
        d_out = ctx.alloc(per * n * bpp)
        ctx.filter_dev(filt, d_in, d_out, w, h, n)
        whole = ctx.checksum_dev(d_out, per * n * bpp)
        got_first = np.empty((h, w, bpp), np.uint8)
        got_last = np.empty((h, w, bpp), np.uint8)
        ctx.d2h(got_first, d_out)
        ctx.d2h(got_last, d_out + (n - 1) * per * bpp)
        d_part = ctx.alloc(per * 4 * bpp)
        parts = 0
        for f in range(0, n, 4):
            ctx.filter_dev(filt, d_in + f * per * 4, d_part, w, h, 4)
            parts += ctx.checksum_dev(d_part, per * 4 * bpp, index_base=f * per * bpp // 4)
        assert parts % (1 << 64) == whole
        for f, got in ((0, got_first), (n - 1, got_last)):
            frame = oracle.synth_rgba(w, h, 1, first_frame=f, seed=0x5EED, mode=0)[0]
            ref = (oracle.sobel_rgba(frame) if filt == pkg.FILTER_SOBEL
                   else oracle.gray_rgba(frame) if bpp == 4 else oracle.gray_rgba_1ch(frame))
            assert np.array_equal(got, ref.reshape(h, w, bpp))
        ctx.free(d_part)
        ctx.free(d_out)
    ctx.free(d_in)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("k,sigma", [(3, 0.8), (5, 1.5)])
def test_gauss_lockstep_rows_with_strips_that_leave_early(ctx, pkg, oracle, k, sigma):
    """Big launches of frames whose strips come in fours (gauss_slide.hip: LOCKSTEP — a workgroup is four adjacent
    strips of one band, one s_barrier per row): 200 frames of 960 x 264 = 4 strips x 11 (k = 5) or 22 (k = 3) bands, >=
    8192 work items.  Strips that stop their 3-channel pass early (alpha noise in ONE strip's columns, a constant from
    some row on, one stray pixel, noise everywhere) sit in one workgroup with strips that do not: nobody may wait for
    ever, and the bytes are the tiled kernel's over the whole batch and within 1 LSB of the CPU path."""
    w, h, n = 960, 264, 200
    per = w * h * 4
    assert (w // 4) % 60 == 0 and ((w // 4) // 60) % 4 == 0 and 4 * (-(-h // 24)) * n >= 8192
    d_in = ctx.alloc(per * n)
    ctx.synth_dev(d_in, w, h, n, first_frame=0, seed=0x5EED, mode=0)
    frames = np.empty((n, h, w, 4), np.uint8)
    ctx.d2h(frames, d_in)
    rng = np.random.default_rng(k)
    touched = [3, 5, 7, 9, 11, 64, 199]
    frames[3, :, 240:480, 3] = rng.integers(0, 256, (h, 240), dtype=np.uint8)   # strip 1 alone leaves at its first row
    frames[5, 100:, 480:720, 3] = 128                                           # strip 2: a constant from row 100 on
    frames[7, :, :, 3] = rng.integers(0, 256, (h, w), dtype=np.uint8)           # every strip leaves
    frames[9, 131, 500, 3] = 0                                                  # one pixel
    frames[11, :, :, 3] = 77                                                    # the constant-alpha pass everywhere
    frames[64, 24:48, :240, 3] = 200                                            # exactly one band of strip 0
    frames[199, h - 1, w - 1, 3] = 1                                            # the very last pixel of the launch
    ctx.h2d(d_in, frames)
    outs = {}
    for impl in (pkg.IMPL_TILE, pkg.IMPL_VALU):
        ctx.set_gauss_mode(pkg.GAUSS_FAST)
        ctx.set_impl(impl)
        d_out = ctx.alloc(per * n)
        ctx.filter_dev(pkg.FILTER_GAUSS, d_in, d_out, w, h, n, k, sigma)
        got = np.empty((n, h, w, 4), np.uint8)
        ctx.d2h(got, d_out)
        ctx.free(d_out)
        outs[impl] = got
    ctx.set_impl(pkg.IMPL_AUTO)
    ctx.free(d_in)
    assert np.array_equal(outs[pkg.IMPL_VALU], outs[pkg.IMPL_TILE])
    for f in touched + [0]:
        ref = oracle.gauss_rgba(frames[f], k, sigma)
        assert np.abs(outs[pkg.IMPL_VALU][f].astype(np.int16) - ref.astype(np.int16)).max() <= 1, f


@pytest.mark.timeout(600)
@pytest.mark.parametrize("w,h,n", [(1920, 1080, 40), (1280, 725, 87)])
def test_sobel_lockstep_rows_on_mid_size_launches(ctx, pkg, oracle, w, h, n):
    """Launches of 8 x 10^7 pixels and more (below the 2^28 where the aligned-strip kernel takes over) run the halo-lane
    Sobel kernel with one s_barrier per row (sobel_slide.hip: LOCKSTEP).  1920 x 1080: 8 strips per row, a workgroup is
    four adjacent strips; 1280 x 725: 6 strips, 46 bands of which the last has 5 rows — workgroups straddle rows and bands
    of different length, waves run out of rows at different times.  The whole launch must equal the same frames run four
    at a time (no barrier there: checksums add up over the word index), first and last frame must equal the oracle."""
    per = w * h
    assert 80_000_000 <= per * n < (1 << 28)
    d_in = ctx.alloc(per * n * 4)
    ctx.synth_dev(d_in, w, h, n, first_frame=0, seed=0x5EED, mode=0)
    d_out = ctx.alloc(per * n)
    ctx.filter_dev(pkg.FILTER_SOBEL, d_in, d_out, w, h, n)
    whole = ctx.checksum_dev(d_out, per * n)
    got_first = np.empty((h, w), np.uint8)
    got_last = np.empty((h, w), np.uint8)
    ctx.d2h(got_first, d_out)
    ctx.d2h(got_last, d_out + (n - 1) * per)
    step = 4 if (per * 4) % 4 == 0 else 1
    d_part = ctx.alloc(per * step)
    parts = 0
    for f in range(0, n, step):
        m = min(step, n - f)
        ctx.filter_dev(pkg.FILTER_SOBEL, d_in + f * per * 4, d_part, w, h, m)
        parts += ctx.checksum_dev(d_part, per * m, index_base=f * per // 4)
    assert parts % (1 << 64) == whole
    for f, got in ((0, got_first), (n - 1, got_last)):
        frame = oracle.synth_rgba(w, h, 1, first_frame=f, seed=0x5EED, mode=0)[0]
        assert np.array_equal(got, oracle.sobel_rgba(frame).reshape(h, w))
    for d in (d_part, d_out, d_in):
        ctx.free(d)


def test_gray_content_takes_the_table_path_and_stays_bit_exact(ctx, pkg, oracle):
    """On gray content (r = g = b: monochrome cameras, documents, the reference's Artemis photographs) every pixel sits on
    the luminance's ambiguous case S = 1000 v; gray pixels then read luma(v, v, v) from a 256-byte table instead of
    evaluating the FP64 formula (common.hpp: luma_px_ambiguous).  Gray noise, gray with colour specks (both branches in one
    wave), and the 36-frame 4K batch that takes sobel_slide.hip's aligned-strip kernel: bit-identical to the CPU path."""
    for (h, w, n) in [(9, 64, 1), (70, 1023, 2), (131, 512, 1), (300, 1920, 1)]:
        x = oracle.synth_rgba(w, h, n, first_frame=1, mode=3)
        speck = x.copy()
        rng = np.random.default_rng(h * w)
        ys, xs = rng.integers(0, h, 200), rng.integers(0, w, 200)
        speck[0, ys, xs, :3] = rng.integers(0, 256, (200, 3), dtype=np.uint8)
        speck[0, ys[:50], xs[:50], :3] = np.array([[0, 72, 24]], np.uint8)  # an ambiguous NON-gray colour: S = 45,000
        for y in (x, speck):
            sob, g1, pipe = ctx.sobel(y), ctx.gray1(y), ctx.pipeline(y, 5, 1.5)
            for f in range(n):
                assert np.array_equal(sob[f], oracle.sobel_rgba(y[f])), (h, w, f)
                assert np.array_equal(g1[f], oracle.gray_rgba_1ch(y[f])), (h, w, f)
                assert np.array_equal(pipe[f], oracle.pipeline_rgba(y[f], 5, 1.5)), (h, w, f)
    w, h, n = 3840, 2160, 36
    per = w * h
    d_in, d_out = ctx.alloc(per * n * 4), ctx.alloc(per * n)
    ctx.synth_dev(d_in, w, h, n, first_frame=0, seed=0x5EED, mode=3)
    for filt, ref_fn in ((pkg.FILTER_SOBEL, oracle.sobel_rgba), (pkg.FILTER_PIPELINE, lambda fr: oracle.pipeline_rgba(fr, 5, 1.5))):
        ctx.filter_dev(filt, d_in, d_out, w, h, n, 5, 1.5)
        got = np.empty((h, w), np.uint8)
        for f in (0, n - 1):
            ctx.d2h(got, d_out + f * per)
            assert np.array_equal(got, ref_fn(oracle.synth_rgba(w, h, 1, first_frame=f, mode=3)[0]))
    ctx.free(d_out)
    ctx.free(d_in)


def test_sobel_stays_close_to_the_reference_opencl_kernel(ctx, oracle):
    """Secondary, tolerance-only check: the reference's own GPU kernel (RT/kernel/edge_base.cl:12-56
    + Controller::ConvertToUChar, RT/src/Controller.cpp:76-85) computes a float luminance / 255, float Sobel, clamps
    the magnitude to [0, 1], leaves the border unwritten and truncates * 255; the CPU path this library follows rounds
    an integer Sobel of the truncated luminance and saturates.  On interior pixels the two differ by rounding only
    (each truncated luminance is < 1 below the float one and the stencil weights sum to 8 in magnitude): a few grey
    levels at most, under one on average (the reference's own published GPU-vs-CPU MAE for this
    filter is 2.0-7.5, src/EdgeDetection/results/*_Tulips_sorted_results.csv)."""
    img = oracle.synth_rgba(320, 200, 1, first_frame=11, mode=1)[0]
    got = ctx.sobel(img).astype(np.float64)
    f = img.astype(np.float32)
    gray = (np.float32(0.299) * f[..., 0] + np.float32(0.587) * f[..., 1] + np.float32(0.114) * f[..., 2]) / np.float32(255)
    gx = (gray[:-2, 2:] + 2 * gray[1:-1, 2:] + gray[2:, 2:]) - (gray[:-2, :-2] + 2 * gray[1:-1, :-2] + gray[2:, :-2])
    gy = (gray[2:, :-2] + 2 * gray[2:, 1:-1] + gray[2:, 2:]) - (gray[:-2, :-2] + 2 * gray[:-2, 1:-1] + gray[:-2, 2:])
    mag = np.clip(np.sqrt(gx * gx + gy * gy), 0.0, 1.0)
    cl_kernel = np.floor(mag.astype(np.float32) * np.float32(255)).astype(np.float64)
    d = np.abs(got[1:-1, 1:-1] - cl_kernel)
    assert d.max() <= 6 and d.mean() < 1.0, (d.max(), d.mean())


def test_pool_alloc_with_placement_search(ctx, pkg, oracle):
    """mi355_pool_alloc: the pools it hands out are ordinary device buffers (the filter run on them equals the
    oracle), every probed candidate reports a positive time, and tries = 1 is a plain allocation."""
    w, h, n = 500, 300, 3
    frames = rand_rgba(h, w, seed=77, n=n)
    for tries in (1, 3):
        d_in, d_out, ms = ctx.pool_alloc(pkg.FILTER_SOBEL, w, h, n, tries=tries)
        assert d_in and d_out and len(ms) == tries
        if tries > 1:
            assert ms[0] > 0 and all(m > 0 or m == -1.0 for m in ms)
        ctx.h2d(d_in, frames)
        ctx.filter_dev(pkg.FILTER_SOBEL, d_in, d_out, w, h, n)
        got = np.empty((n, h, w), np.uint8)
        ctx.d2h(got, d_out)
        assert np.array_equal(got, np.stack([oracle.sobel_rgba(f) for f in frames]))
        ctx.pool_free(d_in, d_out)
    with pytest.raises(pkg.Mi355Error):
        ctx.pool_alloc(pkg.FILTER_GAUSS, w, h, n, k=4, sigma=1.0)


def test_two_contexts_on_two_host_threads(pkg, oracle):
    """A context is single-threaded, but two contexts (own streams, own pools, own coefficient caches) must be usable
    from two host threads at once — the library keeps no mutable global state."""
    import threading
    frames = [rand_rgba(97, 252, seed=s) for s in (1, 2)]
    want = [(oracle.gauss_rgba(f, 5, 1.5), oracle.sobel_rgba(f), oracle.gray_rgba_1ch(f), oracle.pipeline_rgba(f, 5, 1.5))
            for f in frames]
    errors = []

    def worker(idx):
        try:
            with pkg.Context(0) as c:
                for it in range(25):
                    g = c.gauss(frames[idx], 5, 1.5)
                    assert np.abs(g.astype(np.int16) - want[idx][0].astype(np.int16)).max() <= 1
                    assert np.array_equal(c.sobel(frames[idx]), want[idx][1])
                    assert np.array_equal(c.gray1(frames[idx]), want[idx][2])
                    if it % 5 == 0:
                        assert np.array_equal(c.pipeline(frames[idx], 5, 1.5), want[idx][3])
        except Exception as e:  # noqa: BLE001
            errors.append((idx, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("h,w", [(130, 1024), (300, 500), (257, 252)])
def test_sobel_commutes_with_flips_and_transposition(ctx, h, w):
    """The Sobel magnitude of a mirrored / transposed frame is the mirrored / transposed magnitude (exact integer
    arithmetic, symmetric stencil, reflect-101 borders) — which sends the same pixels through other strips, other
    bands and the other walking direction of the kernels."""
    img = rand_rgba(h, w, seed=h + w)
    ref = ctx.sobel(img)
    assert np.array_equal(ctx.sobel(np.ascontiguousarray(img[::-1])), ref[::-1])
    assert np.array_equal(ctx.sobel(np.ascontiguousarray(img[:, ::-1])), ref[:, ::-1])
    assert np.array_equal(ctx.sobel(np.ascontiguousarray(img.transpose(1, 0, 2))), ref.T)
