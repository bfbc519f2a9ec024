#!/usr/bin/env python3
"""Decode the reference's three example photographs into tests/golden/real_images.npz.

Tooling, not a test: it only runs where /root/reference exists (the build container).  What is committed is DATA — the
decoded pixels of example_images/flickr30kimage/{1009434119,480048562}.jpg and example_images/cocoimage/000000005586.jpg
(the files explain_image.py:321-371 / :267-318 point at; all three are 224 x 224 RGB, so `load_img(target_size=(224, 224))`,
models/preprocessors.py:38-53, does not resample them) — as one (3, 224, 224, 3) uint8 RGB array plus their names.
BASELINE configs[0] is "single Flickr30K image": tests/test_gpu_real_images.py and bench.py's `latency` block use image 0.

Usage:  python tests/golden/make_real_images.py [--ref /root/reference]
"""
import argparse
import os

import numpy as np
from PIL import Image

FILES = ["flickr30kimage/1009434119.jpg", "flickr30kimage/480048562.jpg", "cocoimage/000000005586.jpg"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.dirname(os.path.abspath(__file__)))
    a = ap.parse_args()
    imgs = []
    for f in FILES:
        im = Image.open(os.path.join(a.ref, "example_images", f)).convert("RGB")
        if im.size != (224, 224):                       # keras load_img(target_size=...) default interpolation
            im = im.resize((224, 224), Image.NEAREST)
        imgs.append(np.asarray(im, dtype=np.uint8))
    out = os.path.join(a.out, "real_images.npz")
    np.savez_compressed(out, rgb_u8=np.stack(imgs), names=np.array(FILES))
    print("wrote", out, np.stack(imgs).shape, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
