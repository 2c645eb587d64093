"""GPU suite (-m gpu): the matrix-core Gaussian (csrc/gauss_mfma_reg.hip, MI355_IMPL_MFMA; its LDS-staged A/B
partner csrc/gauss_mfma.hip through the tuning build) against the oracle.

Contract: FAST arithmetic, |d| <= 1 LSB per channel against the CPU path (src/GaussianBlur/GaussianBlur.cpp:234-261)
on every shape; the fp16 hi + lo splits keep the error of the sums near 1e-4, so the share of bytes that differ at
all stays small (asserted: < 0.5 %)."""
import numpy as np
import pytest

from conftest import rand_rgba

pytestmark = pytest.mark.gpu


@pytest.fixture()
def mfma(ctx, pkg):
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    ctx.set_impl(pkg.IMPL_MFMA)
    yield ctx
    ctx.set_impl(pkg.IMPL_AUTO)


def _check(got, ref, rate=0.005):
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= 1, int(d.max())
    if d.size >= 4096:
        assert (d != 0).mean() < rate, float((d != 0).mean())


def test_impulse_lands_where_the_table_says(mfma, oracle):
    """One bright pixel per channel at different places: geometry of both passes, every tap of a k = 17 table."""
    h, w = 80, 160
    img = np.zeros((h, w, 4), np.uint8)
    img[40, 70, 0] = 255
    img[17, 5, 1] = 255      # near the left edge: clamp-to-edge columns
    img[2, 150, 2] = 200     # near the top edge
    img[79, 159, 3] = 255    # corner
    for k, sigma in ((17, 6.0), (5, 1.5), (3, 0.8), (9, 2.5)):
        _check(mfma.gauss(img, k, sigma), oracle.gauss_rgba(img, k, sigma), rate=1.0)


@pytest.mark.parametrize("k,sigma", [(3, 0.8), (5, 1.5), (7, 2.0), (9, 2.5), (11, 3.0), (13, 3.3), (15, 4.0), (17, 6.0)])
@pytest.mark.parametrize("h,w", [(1, 4), (3, 8), (16, 64), (17, 68), (40, 252), (61, 112), (131, 500), (200, 640),
                                 (300, 1920), (9, 3840)])
def test_mfma_gaussian_shapes(mfma, oracle, k, sigma, h, w):
    img = rand_rgba(h, w, seed=h * 7 + w + k, alpha=None)
    _check(mfma.gauss(img, k, sigma), oracle.gauss_rgba(img, k, sigma, threads=8))


def test_mfma_gaussian_batches_bands_and_smooth_frames(mfma, oracle):
    frames = oracle.synth_rgba(1000, 520, 3, first_frame=1, mode=1)     # 3 bands of 240 rows, 16 strips, ragged last strip
    got = mfma.gauss(frames, 17, 6.0)
    for f in range(3):
        _check(got[f], oracle.gauss_rgba(frames[f], 17, 6.0, threads=8))
    flat = np.full((100, 128, 4), 255, np.uint8)
    out = mfma.gauss(flat, 17, 6.0)
    assert set(np.unique(out).tolist()) <= {254, 255}
    assert np.array_equal(mfma.gauss(np.zeros((50, 64, 4), np.uint8), 17, 6.0), np.zeros((50, 64, 4), np.uint8))


def test_mfma_full_frames(mfma, oracle):
    for (w, h, k, sigma) in ((1920, 1080, 17, 6.0), (3840, 2160, 17, 6.0), (3840, 2160, 5, 1.5), (3840, 2160, 11, 3.0)):
        frame = oracle.synth_rgba(w, h, 1, first_frame=k, mode=1)[0]
        _check(mfma.gauss(frame, k, sigma), oracle.gauss_rgba(frame, k, sigma, threads=16))
    noisy = rand_rgba(1080, 1920, seed=3, alpha=None)
    _check(mfma.gauss(noisy, 17, 6.0), oracle.gauss_rgba(noisy, 17, 6.0, threads=16))


@pytest.mark.parametrize("k,sigma", [(7, 2.0), (17, 6.0)])
def test_mfma_constant_and_piecewise_constant_alpha(mfma, oracle, k, sigma):
    """The matrix-core kernel recognises tiles whose alpha is ONE value (not only 255): a block between two such tiles of
    the same value stores the CPU chain's byte for an all-A window (no alpha plane, no matrix work for it), and a
    constant tile next to a mixed one contributes its — constant — alpha plane.  Constants, a horizontal and a vertical
    edge on and off the 16-row / 16-pixel tile boundaries, a block with a hole, one stray pixel: within 1 LSB of the
    CPU path everywhere, alpha included."""
    h, w = 200, 320
    base = oracle.synth_rgba(w, h, 1, first_frame=k, mode=0)[0]
    cases = {}
    for a in (0, 1, 128, 254):
        img = base.copy()
        img[..., 3] = a
        cases["const %d" % a] = img
    img = base.copy()
    img[96:, :, 3] = 128          # edge on a tile boundary
    cases["rows 96"] = img
    img = base.copy()
    img[101:, :, 3] = 7           # edge inside a tile
    cases["rows 101"] = img
    img = base.copy()
    img[:, :160, 3] = 33          # vertical edge on a 16-pixel column boundary
    cases["cols 160"] = img
    img = base.copy()
    img[:, :165, 3] = 33
    cases["cols 165"] = img
    img = base.copy()
    img[..., 3] = 200
    img[50:120, 60:200, 3] = 64
    img[70:80, 100:110, 3] = 255
    cases["block"] = img
    img = base.copy()
    img[..., 3] = 90
    img[131, 47, 3] = 91          # one stray pixel in a constant frame
    cases["stray"] = img
    for name, img in cases.items():
        got = mfma.gauss(img, k, sigma)
        ref = oracle.gauss_rgba(img, k, sigma)
        d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
        assert d.max() <= 1, (name, int(d.max()))
        # the alpha of a constant frame is the CPU path's own byte, exactly
        if name.startswith("const"):
            assert np.array_equal(got[..., 3], ref[..., 3]), name
    frames = np.stack([cases["const 128"], cases["block"], base, cases["rows 101"]])
    got = mfma.gauss(frames, k, sigma)
    for f in range(4):
        _check(got[f], oracle.gauss_rgba(frames[f], k, sigma), rate=1.0)


def test_mfma_falls_back_where_it_does_not_apply(mfma, pkg, oracle):
    """Widths that are not multiples of 4, k > 17, EXACT mode: IMPL_MFMA behaves like AUTO."""
    img = rand_rgba(33, 251, seed=5, alpha=None)
    _check(mfma.gauss(img, 17, 6.0), oracle.gauss_rgba(img, 17, 6.0))
    img = rand_rgba(40, 64, seed=6, alpha=None)
    _check(mfma.gauss(img, 31, 10.0), oracle.gauss_rgba(img, 31, 10.0))
    mfma.set_gauss_mode(pkg.GAUSS_EXACT)
    assert np.array_equal(mfma.gauss(img, 17, 6.0), oracle.gauss_rgba(img, 17, 6.0))
    mfma.set_gauss_mode(pkg.GAUSS_FAST)


_PARTNER_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import __graft_entry__ as entry
from conftest import rand_rgba
pkg = entry.load_package(); oracle = entry.load_oracle()
worst = 0
with pkg.Context(0) as ctx:
    ctx.set_impl(pkg.IMPL_MFMA)
    for (h, w, k, s, a) in [(16, 64, 17, 6.0, None), (33, 68, 5, 1.5, None), (200, 640, 11, 3.0, 255), (75, 252, 9, 2.5, None),
                            (1100, 320, 17, 6.0, 255), (2300, 128, 13, 3.3, None)]:
        img = rand_rgba(h, w, seed=h + w + k, alpha=a)
        if a == 255:
            img[h // 2, w // 3, 3] = 7
        d = np.abs(ctx.gauss(img, k, s).astype(np.int16) - oracle.gauss_rgba(img, k, s, threads=8).astype(np.int16))
        worst = max(worst, int(d.max()))
        assert (d != 0).mean() < 0.005 or d.size < 4096, (h, w, k)
print(worst)
"""


def test_lds_staged_partner_and_band_heights():
    """The tuning build keeps two knobs of the matrix-core path reachable: MI355_MFMA_LDS=1 runs the LDS-staged kernel
    (gauss_mfma.hip, round 2's first version, the A/B partner of tools/mfma_reg_ab.sh), MI355_MFMA_BPB sets the band
    height of gauss_mfma_reg.hip (1 block per band: every block pays a halo tile; 3: ragged last band).  Each within
    1 LSB of the oracle on shapes with edges, several bands (2300 rows > 68 blocks), opaque frames with one hole."""
    import os
    import subprocess
    import sys
    import __graft_entry__ as entry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tune_lib = os.path.join(entry.ROOT, "tools", "lib", "libmi355_imgfilter_tune.so")
    assert os.path.exists(tune_lib), "run __graft_entry__.build()"
    # MI355_MFMA_DMA=1: gauss_mfma_dma.hip, the register kernel with its input tiles staged by LDS-DMA (global_load_lds_dwordx4)
    for knobs in ({"MI355_MFMA_LDS": "1"}, {"MI355_MFMA_BPB": "1"}, {"MI355_MFMA_BPB": "3"}, {"MI355_MFMA_DMA": "1"},
                  {"MI355_MFMA_DMA": "1", "MI355_MFMA_BPB": "2"}):
        env = dict(os.environ, MI355_IMGFILTER_LIB=tune_lib, **knobs)
        out = subprocess.run([sys.executable, "-c", _PARTNER_SCRIPT, root], env=env, capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert int(out.stdout.strip().splitlines()[-1]) <= 1, (knobs, out.stdout)
