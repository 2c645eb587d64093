"""CPU suite: the oracle against the reference-built golden vectors, hand-derivable known answers
 and an independent numpy restatement.  No GPU."""
import hashlib

import numpy as np
import pytest

from conftest import rand_rgba


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---- weights: PINNED by the reference's own output -------------------------------------------
def _parse(key):
    k, s, layout = key.split(",")
    return int(k[2:]), float(s[6:]), layout


def test_weights_match_reference_golden_vectors(oracle, golden_weights):
    """gauss_weights_ref.json holds the output of the reference's Controller::_GenerateGausianKernel
    (compiled from /root/reference, see tests/golden/make_golden.py)."""
    n = 0
    for key, bits in golden_weights.items():
        k, sigma, layout = _parse(key)
        if layout != "buffers":
            continue
        got = oracle.gauss_weights(k, sigma).reshape(-1).view(np.uint32)
        assert got.tolist() == bits, key
        n += 1
    assert n >= 15


def test_float_exp_variant_is_rejected_by_the_pin(oracle, golden_weights):
    """The comparison discriminates: evaluating exp in float (a plausible misreading of
    Controller.cpp:360) fails against the reference vectors."""
    bad = 0
    for key, bits in golden_weights.items():
        k, sigma, layout = _parse(key)
        if layout == "buffers" and k >= 3:
            got = oracle.gauss_weights(k, sigma, variant="_fexp").reshape(-1).view(np.uint32)
            bad += got.tolist() != bits
    assert bad >= 8


def test_weights_against_live_reference_build(oracle):
    """Where oracle/_ref exists (built from /root/reference by oracle/Makefile), compare live."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built here")
    for k in (1, 3, 5, 7, 9, 17, 25):
        for sigma in (0.4, 1.0, 1.5, 2.2, 6.0, 11.0):
            a = oracle.gauss_weights(k, sigma).view(np.uint32)
            r = oracle.ref_gauss_weights(k, sigma).view(np.uint32)
            assert np.array_equal(a, r), (k, sigma)


def test_weights_known_values(oracle):
    w = oracle.gauss_weights(5, 1.5)
    assert abs(w[2, 2] - 0.08531173) < 1e-8
    assert abs(w[2, 1] - 0.06831229) < 1e-8
    assert abs(w[0, 0] - 0.01441882) < 1e-8
    assert np.array_equal(w, w.T) and np.array_equal(w, w[::-1, ::-1])
    assert w.sum(dtype=np.float64) < 1.0  # the float table sums to just under one
    assert oracle.gauss_weights(1, 2.0).tolist() == [[1.0]]
    with pytest.raises(ValueError):
        oracle.gauss_weights(4, 1.0)


# ---- grayscale: known answers + independent restatement (pinned by the reference's published numbers in
#      test_published_mae.py) -----------------------------------------------------------------
def test_gray_known_answers(oracle):
    assert oracle.gray_px(0, 72, 24) == 44      # exact rational value is 45.0; the double sum is below
    assert oracle.gray_px(255, 255, 255) == 255
    assert oracle.gray_px(0, 0, 0) == 0
    assert oracle.gray_px(255, 0, 0) == 76 and oracle.gray_px(0, 255, 0) == 149 and oracle.gray_px(0, 0, 255) == 29
    assert sum(oracle.gray_px(v, v, v) != v for v in range(256)) == 65   # not idempotent


def test_gray_all_colours_vs_numpy_and_exact_floor(oracle):
    r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")
    rgba = np.stack([r, g, b, np.full_like(r, 255)], -1).astype(np.uint8).reshape(4096, 4096, 4)
    got = oracle.gray_rgba_1ch(rgba)
    R, G, B = (rgba[..., i].astype(np.float64) for i in range(3))
    assert np.array_equal(got, ((0.299 * R + 0.587 * G) + 0.114 * B).astype(np.uint8))
    Ri, Gi, Bi = (rgba[..., i].astype(np.int64) for i in range(3))
    assert int((got != ((299 * Ri + 587 * Gi + 114 * Bi) // 1000)).sum()) == 3464
    full = oracle.gray_rgba(rgba[:64])
    assert np.array_equal(full[..., 0], got[:64]) and np.array_equal(full[..., 1], got[:64])
    assert np.array_equal(full[..., 2], got[:64]) and (full[..., 3] == 255).all()


def test_gray_config1_fixture(oracle, fixture_rgb, regression):
    """BASELINE.json config 1: grayscale of images/Tulips_medium640.jpg through the CPU path."""
    assert sha(fixture_rgb) == regression["fixture_rgb_sha256"]
    assert fixture_rgb[0, 0].tolist() == [175, 150, 5]
    bgr = fixture_rgb[..., ::-1]
    out = oracle.gray_bgr(bgr)
    assert out.shape == (512, 640)
    assert out[0, 0] == int(0.299 * 175 + 0.587 * 150 + 0.114 * 5)
    assert sha(out) == regression["gray_bgr"]
    rgba = np.dstack([fixture_rgb, np.full((512, 640), 255, np.uint8)])
    assert np.array_equal(oracle.gray_rgba_1ch(rgba), out)   # layout does not change the value


# ---- Gaussian: known answers (pinned by the reference's published numbers in test_published_mae.py) ---
def test_gauss_constant_images(oracle):
    for v, expect in ((255, 254), (128, 128), (200, 200), (0, 0)):
        out = oracle.gauss_rgba(np.full((9, 11, 4), v, np.uint8), 5, 1.5)
        assert np.unique(out).tolist() == [expect], v


def test_gauss_impulse_is_truncated_weight_table(oracle):
    img = np.zeros((11, 13, 4), np.uint8)
    img[5, 6] = (255, 200, 100, 50)
    w = oracle.gauss_weights(5, 1.5)
    out = oracle.gauss_rgba(img, 5, 1.5)
    for c, v in enumerate((255, 200, 100, 50)):
        expect = (np.float32(v) * w[::-1, ::-1]).astype(np.uint8)
        assert np.array_equal(out[3:8, 4:9, c], expect)
    assert out[:3].max() == 0 and out[8:].max() == 0


def test_gauss_corner_impulse_clamps(oracle):
    img = np.zeros((7, 7, 4), np.uint8)
    img[0, 0] = 255
    w = oracle.gauss_weights(5, 1.5).astype(np.float32)
    out = oracle.gauss_rgba(img, 5, 1.5)
    # pixel (0,0) sees the corner through every tap with ky<=0 and kx<=0, summed sequentially in float
    s = np.float32(0)
    for ky in range(5):
        for kx in range(5):
            if ky <= 2 and kx <= 2:
                s = np.float32(s + np.float32(255) * w[ky, kx])
    assert out[0, 0, 0] == int(s)


def test_gauss_vs_numpy_restatement(oracle):
    img = rand_rgba(13, 17, 7, alpha=None)
    k, sigma = 5, 1.5
    w = oracle.gauss_weights(k, sigma)
    h, wd, _ = img.shape
    acc = np.zeros((h, wd, 4), np.float32)
    for ky in range(-2, 3):
        for kx in range(-2, 3):
            ys = np.clip(np.arange(h) + ky, 0, h - 1)
            xs = np.clip(np.arange(wd) + kx, 0, wd - 1)
            acc = (acc + img[ys][:, xs].astype(np.float32) * w[ky + 2, kx + 2]).astype(np.float32)
    assert np.array_equal(oracle.gauss_rgba(img, k, sigma), np.clip(acc, 0, 255).astype(np.uint8))


def test_gauss_tiny_and_ragged_sizes(oracle):
    for (h, w) in ((1, 1), (1, 9), (9, 1), (2, 2), (3, 5)):
        img = rand_rgba(h, w, h * 100 + w)
        out = oracle.gauss_rgba(img, 5, 1.5)
        assert out.shape == img.shape
        if (h, w) == (1, 1):
            # all 25 taps hit the one pixel
            assert abs(int(out[0, 0, 0]) - int(img[0, 0, 0])) <= 1
    assert np.array_equal(oracle.gauss_rgba(rand_rgba(4, 6, 1), 1, 1.0), rand_rgba(4, 6, 1))  # k=1: identity
    img = rand_rgba(20, 31, 3)
    assert np.array_equal(oracle.gauss_rgba(img, 5, 1.5), oracle.gauss_rgba(img, 5, 1.5, threads=4))


# ---- Sobel: restated OpenCV semantics; known answers (published numbers: test_published_mae.py) -------
def test_sobel_known_answers(oracle):
    assert oracle.sobel_gray(np.full((5, 7), 77, np.uint8)).max() == 0         # flat -> 0, border too
    step = np.zeros((8, 10), np.uint8)
    step[:, 5:] = 255
    out = oracle.sobel_gray(step)
    assert (out[:, [4, 5]] == 255).all() and out[:, :4].max() == 0 and out[:, 6:].max() == 0
    ramp = np.tile(np.arange(10, dtype=np.uint8) * 3, (6, 1))
    out = oracle.sobel_gray(ramp)
    assert (out[:, 1:-1] == 24).all()         # gx = 8 * slope
    assert (out[:, 0] == 0).all() and (out[:, -1] == 0).all()   # reflect-101: column -1 == column 1
    one = np.array([[9]], np.uint8)
    assert oracle.sobel_gray(one).tolist() == [[0]]
    assert oracle.sobel_gray(np.array([[0, 255]], np.uint8)).tolist() == [[0, 0]]   # 1xN reflect


def test_sobel_rounding_is_half_even_free_and_saturates(oracle):
    # gx = 3, gy = 4 -> 5 exactly; gx = 1, gy = 1 -> sqrt(2) = 1.41 -> 1; large -> 255
    img = np.zeros((3, 3), np.uint8)
    img[0, 2] = 1
    assert oracle.sobel_gray(img)[1, 1] == 1          # gx=1, gy=-1
    img = np.zeros((3, 3), np.uint8)
    img[1, 2] = 60
    assert oracle.sobel_gray(img)[1, 1] == 120
    img[1, 2] = 128
    assert oracle.sobel_gray(img)[1, 1] == 255        # 256 saturates
    # integer closed form used by the GPU kernel agrees everywhere on random data
    g = np.random.default_rng(5).integers(0, 256, (37, 41), dtype=np.uint8)
    out = oracle.sobel_gray(g)
    p = np.pad(g.astype(np.int64), 1, mode="reflect")
    gx = (p[:-2, 2:] - p[:-2, :-2]) + 2 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2])
    gy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    s = gx * gx + gy * gy
    kk = np.floor(np.sqrt(s.astype(np.float64))).astype(np.int64)
    kk -= (kk * kk > s)
    kk += ((kk + 1) * (kk + 1) <= s)
    assert np.array_equal(out, np.minimum(255, kk + (s > kk * kk + kk)).astype(np.uint8))


def test_sobel_rgba_is_gray_then_sobel(oracle):
    img = rand_rgba(19, 23, 11)
    assert np.array_equal(oracle.sobel_rgba(img), oracle.sobel_gray(oracle.gray_rgba_1ch(img)))


# ---- pipeline: literal composition -------------------------------------------------------------
def test_pipeline_is_the_composition_of_the_three_calls(oracle):
    img = rand_rgba(21, 29, 13)
    k, sigma = 5, 1.5
    chained = oracle.sobel_rgba(oracle.gauss_rgba(oracle.gray_rgba(img), k, sigma))
    assert np.array_equal(oracle.pipeline_rgba(img, k, sigma), chained)
    # the luminance IS re-applied to the blurred gray: dropping it changes the result somewhere
    b = oracle.gauss_rgba(oracle.gray_rgba(img), k, sigma)[..., 0]
    assert not np.array_equal(oracle.gray_rgba_1ch(np.dstack([b, b, b, b])), b)


# ---- regression pins of the oracle itself (self-generated; detect accidental edits only) ------
def test_oracle_regression_hashes(oracle, fixture_rgba, regression):
    assert sha(oracle.gray_rgba(fixture_rgba)) == regression["gray_rgba"]
    assert sha(oracle.gauss_rgba(fixture_rgba, 5, 1.5)) == regression["gauss_k5_s1.5"]
    assert sha(oracle.gauss_rgba(fixture_rgba[:64, :96], 17, 6.0)) == regression["gauss_k17_s6"]
    assert sha(oracle.sobel_rgba(fixture_rgba)) == regression["sobel"]
    assert sha(oracle.pipeline_rgba(fixture_rgba, 5, 1.5)) == regression["pipeline_k5_s1.5"]
    s0 = oracle.synth_rgba(251, 67, 2, first_frame=3, seed=0x5EED, mode=0)
    assert sha(s0) == regression["synth_mode0"]
    assert sha(oracle.synth_rgba(251, 67, 1, first_frame=0, seed=0x5EED, mode=1)) == regression["synth_mode1"]
    assert "%016x" % oracle.checksum(s0) == regression["synth_mode0_checksum"]


def test_checksum_is_shardable(oracle):
    s = oracle.synth_rgba(40, 30, 4)
    whole = oracle.checksum(s)
    words_per_frame = 40 * 30
    parts = sum(oracle.checksum(s[f], index_base=f * words_per_frame) for f in range(4)) % (1 << 64)
    assert parts == whole
    assert oracle.synth_rgba(40, 30, 1, first_frame=2).tobytes() == s[2].tobytes()


# ---- image2d_t mode ------------------------------------------------------------------------------
def test_image2d_weight_generator_matches_reference_vectors(oracle, golden_weights):
    """oracle_gauss_weights_image2d against the reference's own Controller::_GenerateGaussianKernelImage2D
    (RT/src/Controller.cpp:374-403), built from /root/reference by oracle/Makefile: PINNED."""
    n = 0
    for key, bits in golden_weights.items():
        k, s, layout = key.split(",")
        if layout != "image2d":
            continue
        got = oracle.gauss_weights_image2d(int(k[2:]), float(s[6:]))
        assert got.reshape(-1).view(np.uint32).tolist() == bits, key
        kk = int(k[2:])
        if kk > 1:   # the generator's loops stop one short: last row and last column stay zero
            assert not got[-1, :].any() and not got[:, -1].any()
        n += 1
    assert n >= 15


def test_image2d_known_answers(oracle):
    """Hand-derivable values of the restated *_images.cl semantics."""
    white = np.full((6, 9, 4), 255, np.uint8)
    # gray of white: 0.299 + 0.587 + 0.114 in fp32, times 255, truncated
    g = np.float32(0.299) * np.float32(1) + np.float32(0.587) * np.float32(1)
    g = np.float32(g) + np.float32(0.114) * np.float32(1)
    assert np.unique(oracle.image2d_gray(white)).tolist() == [int(np.float32(g) * np.float32(255))]
    assert oracle.image2d_gray(np.zeros((3, 3, 4), np.uint8)).max() == 0
    # Sobel: border is never written (0); a vertical red step saturates the clamp: 255 on the two step columns
    img = np.zeros((8, 12, 4), np.uint8)
    img[:, 6:, 0] = 255
    img[:, :, 1] = 77          # green and blue are ignored: the kernel reads .x only
    s = oracle.image2d_sobel(img)
    assert not s[0].any() and not s[-1].any() and not s[:, 0].any() and not s[:, -1].any()
    assert (s[1:-1, 5:7] == 255).all() and not s[1:-1, 1:5].any() and not s[1:-1, 7:-1].any()
    # Gaussian: taps outside the image contribute 0 and nothing is renormalised: a white image darkens at the border,
    # and even in the interior the image-mode table (last row / column zero, weights still summing to 1) applies
    out = oracle.image2d_gauss(white, 5, 1.5)
    assert out[3, 4].tolist() == [255, 255, 255, 255] or out[3, 4, 0] >= 254
    assert out[0, 0, 0] < out[3, 4, 0] and out[0, 0, 0] < 200
