#!/usr/bin/env python3
"""Regenerates tests/golden/*.  Run in the build container (needs /root/reference and oracle/_ref).

What is committed and where it comes from:
  tulips_medium640_rgb.png   the reference's own test image images/Tulips_medium640.jpg (BASELINE.json
                             config 1), decoded here with PIL (libjpeg-turbo) and stored losslessly.
                             JPEG decoding is outside the hot path: parity is defined on decoded pixels.
  gauss_weights_ref.json     OUTPUT OF THE REFERENCE ITSELF: Controller::_GenerateGausianKernel compiled
                             from /root/reference (oracle/Makefile target `ref`), float bit patterns.
                             These are the golden vectors that pin the weight generator.
  oracle_regression.json     sha256 of the oracle's outputs on the fixture image and on synthetic frames.
                             Produced by OUR restatement, so they pin nothing about the reference: they
                             only detect an accidental change of the oracle ("parity unpinned" paths).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402

REF_IMG = "/root/reference/images/Tulips_medium640.jpg"

WEIGHT_CASES = [(1, 1.0), (3, 0.5), (3, 0.8), (3, 1.0), (5, 1.0), (5, 1.5), (5, 2.0), (7, 1.5), (7, 2.0),
                (9, 2.5), (11, 3.0), (13, 3.3), (15, 4.0), (17, 6.0), (21, 5.0), (31, 10.0), (63, 20.0)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    from PIL import Image

    O.build(ref=True)
    img = np.asarray(Image.open(REF_IMG).convert("RGB"))
    assert img.shape == (512, 640, 3), img.shape
    Image.fromarray(img).save(os.path.join(HERE, "tulips_medium640_rgb.png"), optimize=True)

    weights = {}
    for k, s in WEIGHT_CASES:
        for image_support in (False, True):
            w = O.ref_gauss_weights(k, s, image_support)
            key = "k=%d,sigma=%r,%s" % (k, s, "image2d" if image_support else "buffers")
            weights[key] = [int(v) for v in w.reshape(-1).view(np.uint32)]
    json.dump(weights, open(os.path.join(HERE, "gauss_weights_ref.json"), "w"), indent=0)

    rgba = np.dstack([img, np.full(img.shape[:2], 255, np.uint8)])
    bgr = img[..., ::-1]
    synth0 = O.synth_rgba(251, 67, 2, first_frame=3, seed=0x5EED, mode=0)
    synth1 = O.synth_rgba(251, 67, 1, first_frame=0, seed=0x5EED, mode=1)
    reg = {
        "fixture_rgb_sha256": sha(img),
        "fixture_pixel00": [int(v) for v in img[0, 0]],
        "gray_bgr": sha(O.gray_bgr(bgr)),
        "gray_rgba": sha(O.gray_rgba(rgba)),
        "gauss_k5_s1.5": sha(O.gauss_rgba(rgba, 5, 1.5)),
        "gauss_k17_s6": sha(O.gauss_rgba(rgba[:64, :96], 17, 6.0)),
        "sobel": sha(O.sobel_rgba(rgba)),
        "pipeline_k5_s1.5": sha(O.pipeline_rgba(rgba, 5, 1.5)),
        "synth_mode0": sha(synth0),
        "synth_mode1": sha(synth1),
        "synth_mode0_checksum": "%016x" % O.checksum(synth0),
    }
    json.dump(reg, open(os.path.join(HERE, "oracle_regression.json"), "w"), indent=1)
    print(json.dumps(reg, indent=1))


if __name__ == "__main__":
    main()
