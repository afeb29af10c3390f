"""Helpers shared by the golden-fixture tests: the stage dictionary of one executed back end and its digests."""
import hashlib

import numpy as np


def stages(get_image, x, levels, clahe, kinds):
    """The fixture's stage dictionary from any back end (oracle handle or MusicaProcessing)."""
    out = {"normalized": get_image(kinds.IMG_NORMALIZED, 0), "minmax": np.array(x.minmax(), dtype=np.float32)}
    for i in range(levels):
        out["downsampled_%d" % i] = get_image(kinds.IMG_DOWNSAMPLED, i)
        out["bandpass_%d" % i] = get_image(kinds.IMG_BANDPASS, i)
        out["expand_%d" % i] = get_image(kinds.IMG_EXPAND, i)
        out["contrast_curve_%d" % i] = x.contrast_curve(i)
    for i in range(4):
        out["sdev_%d" % i] = get_image(kinds.IMG_SDEV, i)
        out["noise_hist_%d" % i] = x.noise_hist(i)
        out["noise_hist_max_%d" % i] = np.array(x.noise_hist_max(i), dtype=np.uint32)
    out["cnr"] = get_image(kinds.IMG_CNR, 3)
    out["grad_hist"] = x.grad_hist()
    out["grad_hist_max"] = np.array(x.grad_hist_max(), dtype=np.uint32)
    gc, gw = x.grad_curve()
    out["grad_curve"] = gc
    out["grad_window"] = np.array(gw, dtype=np.float32)
    out["graded"] = get_image(kinds.IMG_GRADED, 0)
    out["out_pixels"] = x.out_pixels()
    if clahe:
        out["clahe_hist"] = x.clahe_hist()
        out["clahe_curves"] = x.clahe_curves()
        out["clahe_graded"] = get_image(kinds.IMG_CLAHE_GRADED, 0)
    return out


def digest(a):
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":   # one bit pattern per value: -0 -> +0, every NaN -> the default NaN
        a = np.where(np.isnan(a), np.float32(np.nan), a + np.float32(0.0)).astype(a.dtype)
    return hashlib.sha256(a.tobytes()).hexdigest()[:32]


def check_digest(got, want):
    assert set(want) == set(got)
    bad = [k for k in want if digest(got[k]) != want[k]]
    assert not bad, "stages differing from the committed digests: %s" % bad
