#!/usr/bin/env python3
"""Regenerates tests/golden/published_mae.json and tests/golden/ref_images/*.  Run in the build container
(needs /root/reference).  Nothing of the reference's source is copied: the JSON holds NUMBERS the reference
published about its own runs, the PNGs hold decoded PIXELS of its test images.

  published_mae.json   the `Error_MAE` column (and image, resolution) of
                       src/{Grayscale,GaussianBlur,EdgeDetection}/results/{Linux,Windows}_100_{Tulips,Artemis}_sorted_results.csv
                       = mean |CPU path - OpenCL path| as the reference's three benchmark applications measured
                       it (grayscale.cpp:307-346, GaussianBlur.cpp:326-330, EdgeDetection.cpp:305-309).
  ref_images/<name>_rgb.png   images/<name>.jpg decoded as cv::imread(IMREAD_COLOR) decodes it (libjpeg, RGB order
                       here), stored losslessly.  Decoded with PIL; that PIL and OpenCV return the same pixels for
                       these files is not assumed but shown by the test that uses them (8 of 8 grayscale numbers
                       reproduced to the last printed digit).
  ref_images/<name>_y.png     the same file decoded as cv::imread(IMREAD_GRAYSCALE) decodes a JPEG: the decoder's
                       own luma plane (libjpeg out_color_space = JCS_GRAYSCALE; PIL: Image.draft('L', size)).
                       This is the input of the reference's CPU Sobel (EdgeDetection.cpp:202).
The two largest images (1023x819, 683x1023; 3.4 MB as PNG) are not committed: the test decodes them live where
/root/reference exists (this container) and skips them elsewhere.
"""
import csv
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
NAMES = ["Tulips_square75", "Tulips_small240", "Tulips_medium640", "Tulips_large1024",
         "Artemis_square75", "Artemis_small240", "Artemis_medium640", "Artemis_large1024"]
COMMITTED = [n for n in NAMES if "large" not in n]
APPS = {"gray": "Grayscale", "gauss": "GaussianBlur", "sobel": "EdgeDetection"}


def decode_rgb(path):
    return np.asarray(Image.open(path).convert("RGB"))


def decode_y(path):
    im = Image.open(path)
    im.draft("L", im.size)  # ask libjpeg for its grayscale output, as OpenCV does for IMREAD_GRAYSCALE
    assert im.mode == "L", im.mode
    return np.asarray(im)


def main():
    out = {}
    for key, app in APPS.items():
        for osname in ("Linux", "Windows"):
            for series in ("Tulips", "Artemis"):
                path = os.path.join(REF, "src", app, "results", "%s_100_%s_sorted_results.csv" % (osname, series))
                for row in csv.DictReader(open(path), skipinitialspace=True):
                    name = os.path.splitext(os.path.basename(row["Image"].strip()))[0]
                    entry = {"resolution": row["Resolution"].strip(), "Error_MAE": row["Error_MAE"].strip(),
                             "source": os.path.relpath(path, REF)}
                    for col in ("Num_Iterations", "avg_CPU_Time_ms", "avg_OpenCL_Time_ms", "avg_OpenCL_kernel_ms",
                                "avg_OpenCL_kernel_write_ms", "avg_OpenCL_kernel_read_ms", "avg_OpenCL_kernel_operation_ms"):
                        if col in row:  # the published timings, for tools/harness.py's side-by-side columns
                            entry[col] = row[col].strip()
                    out.setdefault(key, {}).setdefault(osname, {})[name] = entry
    json.dump(out, open(os.path.join(HERE, "published_mae.json"), "w"), indent=1, sort_keys=True)
    os.makedirs(os.path.join(HERE, "ref_images"), exist_ok=True)
    for n in COMMITTED:
        src = os.path.join(REF, "images", n + ".jpg")
        rgb, y = decode_rgb(src), decode_y(src)
        assert rgb.shape[:2] == y.shape
        if n != "Tulips_medium640":  # that one is tests/golden/tulips_medium640_rgb.png already
            Image.fromarray(rgb).save(os.path.join(HERE, "ref_images", n + "_rgb.png"), optimize=True)
        Image.fromarray(y).save(os.path.join(HERE, "ref_images", n + "_y.png"), optimize=True)
        print(n, rgb.shape)


if __name__ == "__main__":
    main()
