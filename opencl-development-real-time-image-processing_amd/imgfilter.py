"""ctypes binding of include/mi355_imgfilter.h.

Mirrors, for Python hosts, what host/Controller.cpp does for C++ hosts: the three Perform* calls of
the reference `Controller` (include/Controller.hpp:37-47 in the reference) plus the fused pipeline,
in a host-buffer form (numpy in / numpy out, with the six profiling timestamps) and a
device-resident form (raw device pointers, e.g. torch tensors' data_ptr()).

There is no fallback: if the shared library is missing or cannot be loaded this module raises.
"""
import ctypes
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB = os.path.join(_HERE, "lib", "libmi355_imgfilter.so")
_HEADER = os.path.join(_ROOT, "include", "mi355_imgfilter.h")

FILTER_GRAY, FILTER_GRAY1, FILTER_GAUSS, FILTER_SOBEL, FILTER_PIPELINE = 0, 1, 2, 3, 4
GAUSS_FAST, GAUSS_EXACT = 0, 1
INPUT_RGBA, INPUT_BGR = 0, 1
IMPL_AUTO, IMPL_TILE, IMPL_MFMA, IMPL_VALU = 0, 1, 2, 3
OUT_BPP = {FILTER_GRAY: 4, FILTER_GRAY1: 1, FILTER_GAUSS: 4, FILTER_SOBEL: 1, FILTER_PIPELINE: 1}

_u8p = ctypes.POINTER(ctypes.c_uint8)
_f32p = ctypes.POINTER(ctypes.c_float)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p
_ci = ctypes.c_int


class Mi355Error(RuntimeError):
    def __init__(self, fn, code, detail=""):
        self.fn, self.code = fn, code
        super().__init__("%s failed: %d (%s)%s" % (fn, code, detail, ""))


def library_path():
    return _LIB


def build_library(jobs=8):
    """Compile csrc/*.hip for gfx950 into lib/libmi355_imgfilter.so (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-s", "-j%d" % jobs, "-C", os.path.join(_HERE, "csrc"), "all", "tune"], check=True)
    return _LIB


def declared_symbols():
    """Every function name include/mi355_imgfilter.h declares."""
    text = open(_HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", text)))


_lib = None


def load_library(path=None):
    """The product library (cached).  `path`: another build of the same library, bound the same way and NOT cached —
    several builds can live in one process (tools/abx.py: same-process A/B timing on the same buffers)."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    other = path is not None
    # MI355_IMGFILTER_LIB: another build of the same library (A/B timing of kernel changes, tools/ab.sh)
    path = path or os.environ.get("MI355_IMGFILTER_LIB", _LIB)
    if not os.path.exists(path):
        raise Mi355Error("load_library", -1, "%s is missing: run __graft_entry__.build()" % path)
    lib = ctypes.CDLL(path)
    sig = {
        "mi355_device_count": [ctypes.POINTER(_ci)],
        "mi355_ctx_create": [_ci, ctypes.POINTER(_vp)],
        "mi355_ctx_create_on_stream": [_ci, _vp, ctypes.POINTER(_vp)],
        "mi355_ctx_destroy": [_vp],
        "mi355_ctx_device_name": [_vp, ctypes.c_char_p, ctypes.c_size_t],
        "mi355_sync": [_vp],
        "mi355_last_hip_error": [_vp],
        "mi355_ctx_set_gauss_mode": [_vp, _ci],
        "mi355_ctx_set_input_format": [_vp, _ci],
        "mi355_bgr_to_rgba8_dev": [_vp, _vp, _vp, _ci, _ci, _ci],
        "mi355_ctx_set_impl": [_vp, _ci],
        "mi355_gauss_weights": [_ci, ctypes.c_float, _f32p],
        "mi355_ctx_set_gauss_weights": [_vp, _ci, ctypes.c_float, _f32p],
        "mi355_gauss_weights_image2d": [_ci, ctypes.c_float, _f32p],
        "mi355_image2d_rgba8": [_vp, _ci, _u8p, _u8p, _ci, _ci, _ci, ctypes.c_float, _u64p],
        "mi355_gray_rgba8": [_vp, _u8p, _u8p, _ci, _ci, _u64p],
        "mi355_gray1_rgba8": [_vp, _u8p, _u8p, _ci, _ci, _u64p],
        "mi355_gauss_rgba8": [_vp, _u8p, _u8p, _ci, _ci, _ci, ctypes.c_float, _u64p],
        "mi355_sobel_rgba8": [_vp, _u8p, _u8p, _ci, _ci, _u64p],
        "mi355_pipeline_rgba8": [_vp, _u8p, _u8p, _ci, _ci, _ci, ctypes.c_float, _u64p],
        "mi355_filter_batched": [_vp, _ci, _u8p, _u8p, _ci, _ci, _ci, _ci, ctypes.c_float, _u64p],
        "mi355_filter_out_bpp": [_ci],
        "mi355_filter_stream": [_vp, _ci, _u8p, _u8p, _ci, _ci, _ci, _ci, _ci, ctypes.c_float,
                                ctypes.POINTER(ctypes.c_double)],
        "mi355_host_alloc": [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)],
        "mi355_host_free": [_vp, _vp],
        "mi355_gray_rgba8_dev": [_vp, _vp, _vp, _ci, _ci, _ci],
        "mi355_gray1_rgba8_dev": [_vp, _vp, _vp, _ci, _ci, _ci],
        "mi355_gauss_rgba8_dev": [_vp, _vp, _vp, _ci, _ci, _ci, _ci, ctypes.c_float],
        "mi355_sobel_rgba8_dev": [_vp, _vp, _vp, _ci, _ci, _ci],
        "mi355_pipeline_rgba8_dev": [_vp, _vp, _vp, _ci, _ci, _ci, _ci, ctypes.c_float],
        "mi355_filter_dev": [_vp, _ci, _vp, _vp, _ci, _ci, _ci, _ci, ctypes.c_float],
        "mi355_synth_rgba8_dev": [_vp, _vp, _ci, _ci, _ci, _ci, ctypes.c_uint32, _ci],
        "mi355_checksum_dev": [_vp, _vp, ctypes.c_size_t, ctypes.c_uint64, _u64p],
        "mi355_stream_copy_dev": [_vp, _vp, _vp, ctypes.c_size_t],
        "mi355_selftest": [_vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)],
        "mi355_dev_alloc": [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)],
        "mi355_pool_alloc": [_vp, _ci, _ci, _ci, _ci, _ci, ctypes.c_float, _ci, ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                             _f32p],
        "mi355_pool_free": [_vp, _vp, _vp],
        "mi355_dev_free": [_vp, _vp],
        "mi355_copy_h2d": [_vp, _vp, _vp, ctypes.c_size_t],
        "mi355_copy_d2h": [_vp, _vp, _vp, ctypes.c_size_t],
        "mi355_group_create": [_ci, ctypes.POINTER(_ci), ctypes.POINTER(_vp)],
        "mi355_group_destroy": [_vp],
        "mi355_group_size": [_vp, ctypes.POINTER(_ci)],
        "mi355_group_member_ctx": [_vp, _ci, ctypes.POINTER(_vp)],
        "mi355_group_member_status": [_vp, _ci],
        "mi355_group_shard": [_ci, _ci, _ci, ctypes.POINTER(_ci), ctypes.POINTER(_ci)],
        "mi355_group_set_gauss_mode": [_vp, _ci],
        "mi355_group_set_impl": [_vp, _ci],
        "mi355_group_set_input_format": [_vp, _ci],
        "mi355_group_set_gauss_weights": [_vp, _ci, ctypes.c_float, _f32p],
        "mi355_group_filter_batched": [_vp, _ci, _u8p, _u8p, _ci, _ci, _ci, _ci, ctypes.c_float,
                                       ctypes.POINTER(ctypes.c_double)],
        "mi355_group_filter_dev": [_vp, _ci, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _ci, _ci, ctypes.POINTER(_ci), _ci,
                                   ctypes.c_float],
        "mi355_timer_begin": [_vp],
        "mi355_timer_end": [_vp, _f32p],
    }
    for name, args in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if path == _LIB:
                raise
            continue  # an older build under MI355_IMGFILTER_LIB
        fn.argtypes = args
        fn.restype = _ci
    lib.mi355_strerror.argtypes = [_ci]
    lib.mi355_strerror.restype = ctypes.c_char_p
    lib.mi355_build_info.argtypes = []
    lib.mi355_build_info.restype = ctypes.c_char_p
    if not other:
        _lib = lib
    return lib


def _check(fn, rc, ctx=None):
    if rc != 0:
        lib = load_library()
        detail = lib.mi355_strerror(rc).decode()
        if ctx is not None and rc == -2:
            detail += ", hipError=%d" % lib.mi355_last_hip_error(ctx)
        raise Mi355Error(fn, rc, detail)


def gauss_weights(k, sigma):
    """Host coefficient table — replaces Controller::_GenerateGaussianKernelBuffers."""
    out = np.empty((k, k), np.float32) if k > 0 else np.empty((1, 1), np.float32)
    rc = load_library().mi355_gauss_weights(int(k), float(sigma), out.ctypes.data_as(_f32p))
    _check("mi355_gauss_weights", rc)
    return out


def gauss_weights_image2d(k, sigma):
    """The image-mode table — replaces Controller::_GenerateGaussianKernelImage2D (last row / column zero)."""
    out = np.empty((k, k), np.float32) if k > 0 else np.empty((1, 1), np.float32)
    rc = load_library().mi355_gauss_weights_image2d(int(k), float(sigma), out.ctypes.data_as(_f32p))
    _check("mi355_gauss_weights_image2d", rc)
    return out


def device_count():
    n = _ci(0)
    _check("mi355_device_count", load_library().mi355_device_count(ctypes.byref(n)))
    return n.value


class Context:
    """One GPU + one HIP stream + pooled buffers + cached coefficient tables."""

    def __init__(self, device=0, stream=None, lib=None):
        self._lib = lib if lib is not None else load_library()
        self._h = _vp()
        if stream is None:
            rc = self._lib.mi355_ctx_create(int(device), ctypes.byref(self._h))
            fn = "mi355_ctx_create"
        else:
            rc = self._lib.mi355_ctx_create_on_stream(int(device), _vp(int(stream)), ctypes.byref(self._h))
            fn = "mi355_ctx_create_on_stream"
        _check(fn, rc)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if self._h:
            self._lib.mi355_ctx_destroy(self._h)
            self._h = _vp()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        _check("mi355_ctx_device_name", self._lib.mi355_ctx_device_name(self._h, buf, 256), self._h)
        return buf.value.decode()

    def sync(self):
        _check("mi355_sync", self._lib.mi355_sync(self._h), self._h)

    def set_gauss_mode(self, mode):
        _check("mi355_ctx_set_gauss_mode", self._lib.mi355_ctx_set_gauss_mode(self._h, int(mode)), self._h)

    def set_input_format(self, fmt):
        """INPUT_RGBA (default) or INPUT_BGR: host-buffer calls then take (h, w, 3) / (n, h, w, 3) BGR frames."""
        _check("mi355_ctx_set_input_format", self._lib.mi355_ctx_set_input_format(self._h, int(fmt)), self._h)
        self._in_ch = 3 if fmt == INPUT_BGR else 4

    def set_impl(self, impl):
        _check("mi355_ctx_set_impl", self._lib.mi355_ctx_set_impl(self._h, int(impl)), self._h)

    def set_gauss_weights(self, k, sigma, table):
        table = np.ascontiguousarray(table, np.float32)
        if table.size != k * k:
            raise Mi355Error("mi355_ctx_set_gauss_weights", -1, "table must hold k*k floats")
        rc = self._lib.mi355_ctx_set_gauss_weights(self._h, int(k), float(sigma), table.ctypes.data_as(_f32p))
        _check("mi355_ctx_set_gauss_weights", rc, self._h)

    # -- host-buffer calls (numpy) ------------------------------------------------------------
    def _host(self, filt, rgba, k=0, sigma=0.0, profile=False):
        rgba = np.ascontiguousarray(rgba, np.uint8)
        if rgba.ndim == 3:
            frames = rgba[None]
        elif rgba.ndim == 4:
            frames = rgba
        else:
            raise Mi355Error("filter", -1, "expected (h, w, 4) or (n, h, w, 4) uint8")
        n, h, w, c = frames.shape
        if c != getattr(self, "_in_ch", 4):
            raise Mi355Error("filter", -1, "expected %d channels per pixel" % getattr(self, "_in_ch", 4))
        bpp = OUT_BPP[filt]
        out = np.empty((n, h, w, 4) if bpp == 4 else (n, h, w), np.uint8)
        prof = (ctypes.c_uint64 * 6)()
        rc = self._lib.mi355_filter_batched(self._h, filt, frames.ctypes.data_as(_u8p),
                                            out.ctypes.data_as(_u8p), w, h, n, int(k), float(sigma), prof)
        _check("mi355_filter_batched", rc, self._h)
        if rgba.ndim == 3:
            out = out[0]
        return (out, list(prof)) if profile else out

    def gray(self, rgba, profile=False):
        """Controller::PerformCLImageGrayscaling (buffer mode): (g,g,g,255) per pixel."""
        return self._host(FILTER_GRAY, rgba, profile=profile)

    def gray1(self, rgba, profile=False):
        return self._host(FILTER_GRAY1, rgba, profile=profile)

    def gauss(self, rgba, k, sigma, profile=False):
        """Controller::PerformCLGaussianBlur."""
        return self._host(FILTER_GAUSS, rgba, k, sigma, profile=profile)

    def sobel(self, rgba, profile=False):
        """Controller::PerformCLImageEdgeDetection: one byte per pixel."""
        return self._host(FILTER_SOBEL, rgba, profile=profile)

    def pipeline(self, rgba, k, sigma, profile=False):
        return self._host(FILTER_PIPELINE, rgba, k, sigma, profile=profile)

    def image2d(self, filt, rgba, k=0, sigma=0.0):
        """mi355_image2d_rgba8: the reference's image2d_t-mode semantics.  Returns (out, six timestamps)."""
        rgba = np.ascontiguousarray(rgba, np.uint8)
        h, w, _ = rgba.shape
        out = np.empty((h, w, 4) if filt == FILTER_GAUSS else (h, w), np.uint8)
        prof = (ctypes.c_uint64 * 6)()
        rc = self._lib.mi355_image2d_rgba8(self._h, int(filt), rgba.ctypes.data_as(_u8p), out.ctypes.data_as(_u8p),
                                           w, h, int(k), float(sigma), prof)
        _check("mi355_image2d_rgba8", rc, self._h)
        return out, list(prof)

    def pinned_empty(self, shape, dtype=np.uint8):
        """numpy array over pinned host memory (mi355_host_alloc); release with pinned_free(arr)."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = _vp()
        _check("mi355_host_alloc", self._lib.mi355_host_alloc(self._h, nbytes, ctypes.byref(p)), self._h)
        buf = (ctypes.c_uint8 * nbytes).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def pinned_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            _check("mi355_host_free", self._lib.mi355_host_free(self._h, _vp(p)), self._h)

    def stream(self, filt, frames, out=None, k=0, sigma=0.0, chunk_frames=0):
        """mi355_filter_stream: overlapped H2D / kernel / D2H over a host batch.  Returns (out, elapsed_ms)."""
        assert frames.flags["C_CONTIGUOUS"] and frames.dtype == np.uint8 and frames.ndim == 4
        n, h, w, c = frames.shape
        assert c == getattr(self, "_in_ch", 4)
        bpp = OUT_BPP[filt]
        if out is None:
            out = np.empty((n, h, w, 4) if bpp == 4 else (n, h, w), np.uint8)
        ms = ctypes.c_double(0)
        rc = self._lib.mi355_filter_stream(self._h, int(filt), frames.ctypes.data_as(_u8p), out.ctypes.data_as(_u8p),
                                           w, h, n, int(chunk_frames), int(k), float(sigma), ctypes.byref(ms))
        _check("mi355_filter_stream", rc, self._h)
        return out, ms.value

    def single(self, name, rgba, *args):
        """The per-frame C entry points themselves (mi355_gray_rgba8, ...), one frame."""
        rgba = np.ascontiguousarray(rgba, np.uint8)
        h, w, _ = rgba.shape
        prof = (ctypes.c_uint64 * 6)()
        one = name in ("gray1", "sobel", "pipeline")
        out = np.empty((h, w) if one else (h, w, 4), np.uint8)
        fn = getattr(self._lib, "mi355_%s_rgba8" % name)
        extra = [int(args[0]), float(args[1])] if name in ("gauss", "pipeline") else []
        rc = fn(self._h, rgba.ctypes.data_as(_u8p), out.ctypes.data_as(_u8p), w, h, *extra, prof)
        _check("mi355_%s_rgba8" % name, rc, self._h)
        return out, list(prof)

    # -- device-resident calls (raw pointers) -------------------------------------------------
    def filter_dev(self, filt, d_in, d_out, w, h, nframes, k=0, sigma=0.0):
        rc = self._lib.mi355_filter_dev(self._h, int(filt), _vp(int(d_in)), _vp(int(d_out)), int(w), int(h),
                                        int(nframes), int(k), float(sigma))
        _check("mi355_filter_dev", rc, self._h)

    def synth_dev(self, d_out, w, h, nframes, first_frame=0, seed=0x5EED, mode=0):
        rc = self._lib.mi355_synth_rgba8_dev(self._h, _vp(int(d_out)), int(w), int(h), int(nframes),
                                             int(first_frame), ctypes.c_uint32(seed), int(mode))
        _check("mi355_synth_rgba8_dev", rc, self._h)

    def checksum_dev(self, d_buf, nbytes, index_base=0):
        out = ctypes.c_uint64(0)
        rc = self._lib.mi355_checksum_dev(self._h, _vp(int(d_buf)), int(nbytes), ctypes.c_uint64(index_base),
                                          ctypes.byref(out))
        _check("mi355_checksum_dev", rc, self._h)
        return out.value

    def stream_copy_dev(self, d_dst, d_src, nbytes):
        rc = self._lib.mi355_stream_copy_dev(self._h, _vp(int(d_dst)), _vp(int(d_src)), int(nbytes))
        _check("mi355_stream_copy_dev", rc, self._h)

    def selftest(self):
        """(bad_luma, bad_mag): exhaustive on-device check of the fast luminance / magnitude forms; (0, 0) = good."""
        a, b = ctypes.c_uint32(0), ctypes.c_uint32(0)
        _check("mi355_selftest", self._lib.mi355_selftest(self._h, ctypes.byref(a), ctypes.byref(b)), self._h)
        return a.value, b.value

    def pool_alloc(self, filt, w, h, nframes, k=5, sigma=1.5, tries=4):
        """(d_in, d_out, probe_ms): input (0xFF-filled) and output frame pools, the output pool placed by a short
        search over physical placements (see mi355_pool_alloc); release with pool_free."""
        a, b = _vp(), _vp()
        ms = (ctypes.c_float * max(1, tries))()
        rc = self._lib.mi355_pool_alloc(self._h, int(filt), int(w), int(h), int(nframes), int(k), float(sigma),
                                        int(tries), ctypes.byref(a), ctypes.byref(b), ms)
        _check("mi355_pool_alloc", rc, self._h)
        return a.value, b.value, [float(x) for x in ms][:max(1, tries)]

    def pool_free(self, d_in, d_out):
        _check("mi355_pool_free", self._lib.mi355_pool_free(self._h, _vp(int(d_in or 0)), _vp(int(d_out or 0))),
               self._h)

    def alloc(self, nbytes):
        p = _vp()
        _check("mi355_dev_alloc", self._lib.mi355_dev_alloc(self._h, int(nbytes), ctypes.byref(p)), self._h)
        return p.value

    def free(self, d_ptr):
        _check("mi355_dev_free", self._lib.mi355_dev_free(self._h, _vp(int(d_ptr))), self._h)

    def h2d(self, d_dst, arr):
        arr = np.ascontiguousarray(arr)
        rc = self._lib.mi355_copy_h2d(self._h, _vp(int(d_dst)), _vp(arr.ctypes.data), arr.nbytes)
        _check("mi355_copy_h2d", rc, self._h)

    def d2h(self, arr, d_src):
        assert arr.flags["C_CONTIGUOUS"]
        rc = self._lib.mi355_copy_d2h(self._h, _vp(arr.ctypes.data), _vp(int(d_src)), arr.nbytes)
        _check("mi355_copy_d2h", rc, self._h)

    def timer_begin(self):
        _check("mi355_timer_begin", self._lib.mi355_timer_begin(self._h), self._h)

    def timer_end(self):
        ms = ctypes.c_float(0)
        _check("mi355_timer_end", self._lib.mi355_timer_end(self._h, ctypes.byref(ms)), self._h)
        return ms.value


def group_shard(member, nmembers, nframes):
    """(first_frame, count) of one member: mi355_group_shard, a pure host function (= bench.shard_range's strong split)."""
    a, b = _ci(0), _ci(0)
    _check("mi355_group_shard", load_library().mi355_group_shard(int(member), int(nmembers), int(nframes),
                                                                  ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


class Group:
    """mi355_group_*: one batch of frames sharded over several GPUs, one context + one host thread per member.
    `devices`: HIP ordinals, one per member (an ordinal may repeat: members then share that GPU)."""

    def __init__(self, devices):
        self._lib = load_library()
        self._h = _vp()
        devs = (_ci * len(devices))(*[int(d) for d in devices])
        _check("mi355_group_create", self._lib.mi355_group_create(len(devices), devs, ctypes.byref(self._h)))
        self.size = len(devices)

    def close(self):
        if self._h:
            self._lib.mi355_group_destroy(self._h)
            self._h = _vp()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def member(self, i):
        """A borrowed Context of member i (device memory, copies, checksums); do not close it."""
        h = _vp()
        _check("mi355_group_member_ctx", self._lib.mi355_group_member_ctx(self._h, int(i), ctypes.byref(h)))
        c = Context.__new__(Context)
        c._lib, c._h = self._lib, h
        c.close = lambda: None
        return c

    def member_status(self, i):
        return self._lib.mi355_group_member_status(self._h, int(i))

    def set_gauss_mode(self, mode):
        _check("mi355_group_set_gauss_mode", self._lib.mi355_group_set_gauss_mode(self._h, int(mode)))

    def set_impl(self, impl):
        _check("mi355_group_set_impl", self._lib.mi355_group_set_impl(self._h, int(impl)))

    def set_input_format(self, fmt):
        _check("mi355_group_set_input_format", self._lib.mi355_group_set_input_format(self._h, int(fmt)))
        self._in_ch = 3 if fmt == INPUT_BGR else 4

    def set_gauss_weights(self, k, sigma, table):
        table = np.ascontiguousarray(table, np.float32)
        rc = self._lib.mi355_group_set_gauss_weights(self._h, int(k), float(sigma), table.ctypes.data_as(_f32p))
        _check("mi355_group_set_gauss_weights", rc)

    def filter_batched(self, filt, frames, k=0, sigma=0.0, out=None):
        """Host frames (n, h, w, c) -> (out, elapsed_ms): every member streams its contiguous range."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w, c = frames.shape
        assert c == getattr(self, "_in_ch", 4)
        bpp = OUT_BPP[filt]
        if out is None:
            out = np.empty((n, h, w, 4) if bpp == 4 else (n, h, w), np.uint8)
        ms = ctypes.c_double(0)
        rc = self._lib.mi355_group_filter_batched(self._h, int(filt), frames.ctypes.data_as(_u8p), out.ctypes.data_as(_u8p),
                                                  w, h, n, int(k), float(sigma), ctypes.byref(ms))
        _check("mi355_group_filter_batched", rc)
        return out, ms.value

    def filter_dev(self, filt, d_in, d_out, w, h, nframes, k=0, sigma=0.0):
        """Device-resident: d_in[m] / d_out[m] / nframes[m] per member; launches everywhere, returns when all are done."""
        m = self.size
        a = (_vp * m)(*[_vp(int(x or 0)) for x in d_in])
        b = (_vp * m)(*[_vp(int(x or 0)) for x in d_out])
        nf = (_ci * m)(*[int(x) for x in nframes])
        rc = self._lib.mi355_group_filter_dev(self._h, int(filt), a, b, int(w), int(h), nf, int(k), float(sigma))
        _check("mi355_group_filter_dev", rc)
