"""ctypes door to oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module (see the header of oracle/imgfilter_oracle.c).  The product never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libref_gauss_weights.so")

_u8p = ctypes.POINTER(ctypes.c_uint8)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(ref=True):
    """Compile the restatement (and, when /root/reference exists, oracle/_ref)."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", _HERE] + targets, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _load():
    if not os.path.exists(_LIB_PATH):
        build(ref=False)
    lib = ctypes.CDLL(_LIB_PATH)
    lib.oracle_gray_px.restype = ctypes.c_uint8
    lib.oracle_gray_px.argtypes = [ctypes.c_int] * 3
    lib.oracle_gray_bgr.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int]
    lib.oracle_gray_rgba_1ch.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int]
    lib.oracle_gray_rgba.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int]
    for name in ("oracle_gauss_weights", "oracle_gauss_weights_fexp"):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int, ctypes.c_float, _f32p]
    lib.oracle_gauss_rgba.restype = ctypes.c_int
    lib.oracle_gauss_rgba.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]
    lib.oracle_gauss_rgba_mt.restype = ctypes.c_int
    lib.oracle_gauss_rgba_mt.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p,
                                         ctypes.c_int]
    lib.oracle_max_threads.restype = ctypes.c_int
    lib.oracle_sobel_gray.restype = ctypes.c_int
    lib.oracle_sobel_gray.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int]
    lib.oracle_sobel_rgba.restype = ctypes.c_int
    lib.oracle_sobel_rgba.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int]
    lib.oracle_pipeline_rgba.restype = ctypes.c_int
    lib.oracle_pipeline_rgba.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]
    lib.oracle_image2d_gray.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int]
    lib.oracle_gauss_weights_image2d.restype = ctypes.c_int
    lib.oracle_gauss_weights_image2d.argtypes = [ctypes.c_int, ctypes.c_float, _f32p]
    lib.oracle_image2d_gauss.restype = ctypes.c_int
    lib.oracle_image2d_gauss.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]
    lib.oracle_image2d_sobel.restype = ctypes.c_int
    lib.oracle_image2d_sobel.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int]
    lib.oracle_synth_rgba.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_uint32, ctypes.c_int]
    lib.oracle_checksum.restype = ctypes.c_uint64
    lib.oracle_checksum.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_uint64]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p8(a):
    return a.ctypes.data_as(_u8p)


def _pf(a):
    return a.ctypes.data_as(_f32p)


def _c(a, dtype=np.uint8):
    return np.ascontiguousarray(a, dtype=dtype)


def gray_px(r, g, b):
    return int(lib().oracle_gray_px(int(r), int(g), int(b)))


def gray_bgr(bgr):
    bgr = _c(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().oracle_gray_bgr(_p8(bgr), _p8(out), w, h)
    return out


def gray_rgba_1ch(rgba):
    rgba = _c(rgba)
    h, w, _ = rgba.shape
    out = np.empty((h, w), np.uint8)
    lib().oracle_gray_rgba_1ch(_p8(rgba), _p8(out), w, h)
    return out


def gray_rgba(rgba):
    rgba = _c(rgba)
    h, w, _ = rgba.shape
    out = np.empty((h, w, 4), np.uint8)
    lib().oracle_gray_rgba(_p8(rgba), _p8(out), w, h)
    return out


def gauss_weights(k, sigma, variant=""):
    out = np.empty((k, k), np.float32)
    fn = getattr(lib(), "oracle_gauss_weights" + variant)
    rc = fn(int(k), float(sigma), _pf(out))
    if rc != 0:
        raise ValueError("oracle_gauss_weights: bad (k, sigma) = (%r, %r)" % (k, sigma))
    return out


def gauss_rgba(rgba, k, sigma=None, weights=None, threads=1):
    rgba = _c(rgba)
    h, w, _ = rgba.shape
    if weights is None:
        weights = gauss_weights(k, sigma)
    weights = _c(weights, np.float32)
    out = np.empty_like(rgba)
    if threads == 1:
        rc = lib().oracle_gauss_rgba(_p8(rgba), _p8(out), w, h, int(k), _pf(weights))
    else:
        rc = lib().oracle_gauss_rgba_mt(_p8(rgba), _p8(out), w, h, int(k), _pf(weights), int(threads))
    if rc != 0:
        raise ValueError("oracle_gauss_rgba rc=%d" % rc)
    return out


def max_threads():
    return int(lib().oracle_max_threads())


def sobel_gray(gray):
    gray = _c(gray)
    h, w = gray.shape
    out = np.empty((h, w), np.uint8)
    rc = lib().oracle_sobel_gray(_p8(gray), _p8(out), w, h)
    if rc != 0:
        raise ValueError("oracle_sobel_gray rc=%d" % rc)
    return out


def sobel_rgba(rgba):
    rgba = _c(rgba)
    h, w, _ = rgba.shape
    out = np.empty((h, w), np.uint8)
    rc = lib().oracle_sobel_rgba(_p8(rgba), _p8(out), w, h)
    if rc != 0:
        raise ValueError("oracle_sobel_rgba rc=%d" % rc)
    return out


def pipeline_rgba(rgba, k, sigma=None, weights=None):
    rgba = _c(rgba)
    h, w, _ = rgba.shape
    if weights is None:
        weights = gauss_weights(k, sigma)
    weights = _c(weights, np.float32)
    out = np.empty((h, w), np.uint8)
    rc = lib().oracle_pipeline_rgba(_p8(rgba), _p8(out), w, h, int(k), _pf(weights))
    if rc != 0:
        raise ValueError("oracle_pipeline_rgba rc=%d" % rc)
    return out


# --- image2d_t mode: restated OpenCL-C semantics of the reference's *_images.cl kernels ---------
def image2d_gray(rgba):
    rgba = _c(rgba)
    h, w, _ = rgba.shape
    out = np.empty((h, w), np.uint8)
    lib().oracle_image2d_gray(_p8(rgba), _p8(out), w, h)
    return out


def gauss_weights_image2d(k, sigma):
    out = np.empty((k, k), np.float32)
    if lib().oracle_gauss_weights_image2d(int(k), float(sigma), _pf(out)) != 0:
        raise ValueError("oracle_gauss_weights_image2d: bad (k, sigma)")
    return out


def image2d_gauss(rgba, k, sigma):
    rgba = _c(rgba)
    h, w, _ = rgba.shape
    table = gauss_weights_image2d(k, sigma)
    out = np.empty_like(rgba)
    if lib().oracle_image2d_gauss(_p8(rgba), _p8(out), w, h, int(k), _pf(table)) != 0:
        raise ValueError("oracle_image2d_gauss")
    return out


def image2d_sobel(rgba):
    rgba = _c(rgba)
    h, w, _ = rgba.shape
    out = np.empty((h, w), np.uint8)
    if lib().oracle_image2d_sobel(_p8(rgba), _p8(out), w, h) != 0:
        raise ValueError("oracle_image2d_sobel")
    return out


def synth_rgba(w, h, nframes=1, first_frame=0, seed=0x5EED, mode=0):
    out = np.empty((nframes, h, w, 4), np.uint8)
    lib().oracle_synth_rgba(_p8(out), w, h, nframes, first_frame, ctypes.c_uint32(seed), mode)
    return out


def checksum(buf, index_base=0):
    buf = _c(buf).reshape(-1)
    return int(lib().oracle_checksum(_p8(buf), buf.size, ctypes.c_uint64(index_base)))


# --- the reference build (oracle/_ref), present only where /root/reference was --------------
def have_ref():
    return os.path.exists(_REF_PATH)


def ref_gauss_weights(k, sigma, image_support=False):
    r = ctypes.CDLL(_REF_PATH)
    r.ref_gauss_weights.restype = ctypes.c_int
    r.ref_gauss_weights.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, _f32p]
    out = np.empty((k, k), np.float32)
    n = r.ref_gauss_weights(int(k), float(sigma), int(bool(image_support)), _pf(out))
    assert n == k * k
    return out
