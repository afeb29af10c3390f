"""Generates tests/golden/mean_cnr_reference.json from the reference's committed CNR dumps.

The 11 files under /root/reference/test/mean_cnr/in/*.bmp are OUTPUTS of the reference pipeline
(debugProcess' cnr.bmp, 384 x 384 = level 3 of a 3072 x 3072 image) for the unaltered image and for
5 Gaussian-noise / 5 Poisson-noise alterations; their inputs are missing blobs. The reference's
test/mean_cnr/script.py:13-24 reduces each dump to  mean(pixels) / 2^8 * 256; this script applies the same
reduction (data only, no reference code is executed) and stores the 11 numbers.
Run here, where /root/reference exists:   python tests/golden/make_mean_cnr.py
"""
import json
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/test/mean_cnr/in"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mean_cnr_reference.json")

res = {}
for name in sorted(os.listdir(SRC)):
    with Image.open(os.path.join(SRC, name)) as img:
        a = np.array(img.convert("L"), dtype=np.uint8)
    res[name[:-4]] = {"mean_cnr": float(np.mean(a) / 2 ** 8 * 256), "width": int(a.shape[1]), "height": int(a.shape[0])}
json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
