"""numpy restatement of the reference's OpenCL BUFFER-MODE kernels — TEST INFRASTRUCTURE ONLY.

Why this exists: the reference ships no golden outputs, but it does ship the OUTPUT OF ITS OWN RUNS:
`src/{Grayscale,GaussianBlur,EdgeDetection}/results/Linux_100_{Tulips,Artemis}_sorted_results.csv`, column
`Error_MAE` = mean |CPU path - OpenCL path| per image, printed with 6 significant digits.  Both operands of that
difference can be restated: the CPU path is `oracle/imgfilter_oracle.c`, the OpenCL path is this file.  If the two
restatements reproduce the published numbers on the reference's own images, the CPU restatement is pinned by the
reference's own outputs (a wrong loop order, rounding rule, border rule or luminance formula moves the counts; the
negative controls in tests/test_published_mae.py show by how much).  tests/test_published_mae.py does exactly that.

What is restated (file:line in the reference):
  cl_gray    src/Grayscale/kernel/grayscale_base.cl:14 (= RT/kernel/grayscale_base.cl), read back as float4 and
             converted on the host by `uchar(f * 255.0f)` (src/Grayscale/grayscale.cpp:196, RT/src/Controller.cpp:76-85)
  cl_gauss   src/GaussianBlur/kernel/gaussian_base.cl:23-49 (`sum += convert_float4(pixel) * weight`, `sum /=
             total_weight`, `convert_uchar4` = truncation)
  cl_sobel   src/EdgeDetection/kernel/edge_base.cl:12-56 (interior pixels only; per-tap luminance / 255; sqrt; clamp to
             [0, 1]); the host converts with `uchar(f * 255.0f)`; border elements of the device buffer are never
             written (a fresh buffer reads back 0)

OpenCL C leaves two things to the device compiler that change low-order bits: contraction of a*b+c into fma, and the
accuracy of `/` and `sqrt` (2.5 / 3 ulp allowed).  The publisher's Linux device is identified by its results: of the
seven possible contraction patterns of the luminance expression x two division forms, exactly ONE reproduces all
eight published grayscale numbers to every printed digit (`fma(0.114f, b, fma(0.299f, r, 0.587f * g)) * (1 / 255.0f)`);
the Gaussian numbers need `sum = fma(pixel, weight, sum)` (all eight exact; without fma all eight would be 0).  Those are
the defaults here; the other patterns stay selectable so the test can show that they do NOT reproduce the numbers.
"""
import numpy as np

f32 = np.float32
f64 = np.float64

C_R, C_G, C_B = f32(0.299), f32(0.587), f32(0.114)
RCP255 = f32(1.0) / f32(255.0)


def fma32(a, b, c):
    """fl32(a * b + c) with ONE rounding.  a, b, c: float32 arrays / scalars.  The product of two float32 is exact in
    float64; the sum is rounded to odd in float64 (TwoSum tells whether it was inexact) so that the final rounding to
    float32 is the correct single rounding."""
    p = np.asarray(a, dtype=f64) * np.asarray(b, dtype=f64)
    c = np.asarray(c, dtype=f64)
    s = p + c
    bb = s - p
    e = (p - (s - bb)) + (c - bb)
    bits = np.ascontiguousarray(s).view(np.int64)
    need = (e != 0) & ((bits & 1) == 0)
    adj = np.where((e > 0) == (s > 0), 1, -1)
    bits = np.where(need, bits + adj, bits)
    return bits.view(f64).astype(f32)


def _luma_sum(r, g, b, contraction):
    """0.299f * r + 0.587f * g + 0.114f * b in float32, left to right, under one contraction pattern."""
    if contraction == "none":
        return (C_R * r + C_G * g) + C_B * b
    if contraction == "published":  # t = fma(0.299, r, 0.587 * g); s = fma(0.114, b, t)
        return fma32(C_B, b, fma32(C_R, r, C_G * g))
    if contraction == "fma_chain":  # t = fma(0.587, g, 0.299 * r); s = fma(0.114, b, t)
        return fma32(C_B, b, fma32(C_G, g, C_R * r))
    if contraction == "fma_last":
        return fma32(C_B, b, C_R * r + C_G * g)
    if contraction == "fma_first":
        return fma32(C_G, g, C_R * r) + C_B * b
    raise ValueError(contraction)


def _div255(s, division):
    if division == "rcp":  # x / 255.0f compiled as x * (1 / 255.0f)
        return s * RCP255
    if division == "ieee":
        return s / f32(255.0)
    raise ValueError(division)


def cl_gray(rgba, contraction="published", division="rcp"):
    """(h, w, 4) u8 -> (h, w) u8: the gray byte of the (g, g, g, 255) pixel the buffer-mode call returns."""
    r, g, b = (rgba[..., i].astype(f32) for i in range(3))
    gray = _div255(_luma_sum(r, g, b, contraction), division)
    return (gray * f32(255.0)).astype(np.uint8)  # C truncation; values are in [0, 255]


def cl_gauss(rgba, k, weights, fma=True):
    """(h, w, 4) u8 -> (h, w, 4) u8.  weights: k*k float32, the buffers-mode table (oracle.gauss_weights)."""
    wt = np.asarray(weights, dtype=f32).reshape(-1)
    R = k // 2
    h, w, _ = rgba.shape
    pad = np.pad(rgba, ((R, R), (R, R), (0, 0)), mode="edge").astype(f32)  # clamp(neighbor, 0, size - 1)
    s = np.zeros((h, w, 4), f32)
    tw = f32(0.0)
    for ky in range(k):
        for kx in range(k):
            wgt = wt[ky * k + kx]
            px = pad[ky:ky + h, kx:kx + w, :]
            s = fma32(px, wgt, s) if fma else (s + px * wgt).astype(f32)
            tw = f32(tw + wgt)
    s = (s / tw).astype(f32)
    return np.clip(s, 0, 255).astype(np.uint8)  # convert_uchar4: round toward zero


_SX = ((-1, 0, 1), (-2, 0, 2), (-1, 0, 1))
_SY = ((-1, -2, -1), (0, 0, 0), (1, 2, 1))


def cl_sobel(rgba, contraction="published", division="rcp", sqrt_ulps=0, border=0):
    """(h, w, 4) u8 -> (h, w) u8.  sqrt_ulps: the device's sqrt may be off by a few ulp (OpenCL allows 3); 0 = correctly
    rounded, +1 = one ulp above, to bracket what a device can return."""
    h, w, _ = rgba.shape
    out = np.full((h, w), border, np.uint8)
    if h < 3 or w < 3:
        return out
    r, g, b = (rgba[..., i].astype(f32) for i in range(3))
    gray = _div255(_luma_sum(r, g, b, contraction), division)
    gx = np.zeros((h - 2, w - 2), f32)
    gy = np.zeros((h - 2, w - 2), f32)
    for ky in range(3):
        for kx in range(3):
            t = gray[ky:ky + h - 2, kx:kx + w - 2]
            gx = (gx + t * f32(_SX[ky][kx])).astype(f32)  # products by 0, +-1, +-2 are exact: fma or not is the same
            gy = (gy + t * f32(_SY[ky][kx])).astype(f32)
    arg = (gx * gx + gy * gy).astype(f32)
    m = np.sqrt(arg)
    if sqrt_ulps:
        m = np.where(arg > 0, (np.ascontiguousarray(m).view(np.int32) + sqrt_ulps).view(f32), m)
    m = np.clip(m, f32(0.0), f32(1.0))
    out[1:-1, 1:-1] = (m * f32(255.0)).astype(np.uint8)
    return out


def cl_sobel_image2d(rgba, sqrt_ulps=0):
    """src/EdgeDetection/kernel/edge_images.cl:9-46 (= RT/kernel/edge_images.cl): red channel of the normalised texel
    (byte / 255.0f), interior pixels only, sqrt, clamp to [0, 1]; read back as float and converted by uchar(f * 255.0f).
    No luminance expression, hence no contraction freedom: only the device's sqrt is open.  With sqrt_ulps = 0 this is
    oracle_image2d_sobel (imgfilter_oracle.c) bit for bit."""
    h, w, _ = rgba.shape
    out = np.zeros((h, w), np.uint8)
    if h < 3 or w < 3:
        return out
    px = rgba[..., 0].astype(f32) / f32(255.0)
    gx = np.zeros((h - 2, w - 2), f32)
    gy = np.zeros((h - 2, w - 2), f32)
    for ky in range(3):
        for kx in range(3):
            t = px[ky:ky + h - 2, kx:kx + w - 2]
            gx = (gx + t * f32(_SX[ky][kx])).astype(f32)
            gy = (gy + t * f32(_SY[ky][kx])).astype(f32)
    arg = (gx * gx + gy * gy).astype(f32)
    m = np.sqrt(arg)
    if sqrt_ulps:
        m = np.where(arg > 0, (np.ascontiguousarray(m).view(np.int32) + sqrt_ulps).view(f32), m)
    m = np.clip(m, f32(0.0), f32(1.0))
    out[1:-1, 1:-1] = (m * f32(255.0)).astype(np.uint8)
    return out
