"""Metamorphic-testing harness for the MUSICA pipeline — the counterpart of the reference's
test/metamorphic_test/script.py, restated on numpy/scipy and driving libmusica_hip.so.

The reference perturbs each raw image (collimator shutters, translations, rotations, Gaussian and
Poisson noise), runs `maverick-standalone <raw> <bmp>` and records three similarities between the
processed altered image and the processed unaltered image: 1 - RMSE/255 (`mse_similarity`,
script.py:143-145), SSIM (:147-152) and histogram distances (:154-198), both "direct" and
"registration based" (the altered result cropped / rotated back onto the unaltered one, :442-456,
:484-508, :586-608). It only logs the numbers. This module keeps the alteration generators, their
parameter grids and the metric definitions, can drive the library in-process or through the drop-in CLI
exactly like `run_process` (:200-214), and exposes the relations as data so tests can assert them on
phantoms (the reference's raw_images/ are missing blobs).
"""
import csv
import math
import os
import subprocess

import numpy as np
from scipy import ndimage

from . import processing as mp

PROCESSING_MARGIN = 10  # script.py:24 == MUSICA_OUT_MARGIN

# alteration grids of the reference, for a 3072-pixel image (script.py:414, 459, 511, 562, 612, 636)
REF_IMAGE_SIZE = 3072
SHUTTERS = [200, 400, 600, 800, 1000]
TRANSLATIONS = [300, 600, 900, 1200, 1500]
ROTATIONS = [9, 18, 27, 36, 45]
GAUSS_SIGMAS = [4.0, 16.0, 64.0, 256.0, 1024.0]
POISSON_FACTORS = [0.1, 0.05, 0.025, 0.0125, 0.00625]


def scaled(values, image_size):
    """The reference's pixel-valued grids scaled from 3072 to `image_size`."""
    return [max(1, int(round(v * image_size / REF_IMAGE_SIZE))) for v in values]


# ---- alteration generators (script.py:49-141) ------------------------------------------------------

def apply_quantum_noise(image, scale_factor=1.0, rng=None):
    """script.py:49-58: Poisson noise at `scale_factor` of the dose."""
    rng = rng or np.random.default_rng()
    scaled_image = image.astype(np.float64) * scale_factor
    noisy = rng.poisson(scaled_image).astype(np.float32) / scale_factor
    return np.clip(noisy, 0, 65535).astype(np.uint16)


def add_gaussian_noise(image, mean, sigma, rng=None):
    """script.py:60-66: additive Gaussian noise, truncated to int before the add."""
    rng = rng or np.random.default_rng()
    noise = rng.normal(mean, sigma, image.shape).astype(np.int32)
    return np.clip(image.astype(np.int32) + noise, 0, 65535).astype(np.uint16)


def apply_collimator(image, shutter_h, shutter_v, rng=None):
    """script.py:75-95: outside the shutter rectangle the detector sees 1 % of the dose (+ Poisson noise)."""
    h, w = image.shape
    mask = np.zeros((h, w), dtype=bool)
    mask[shutter_v:h - shutter_v + 1, shutter_h:w - shutter_h + 1] = True   # PIL rectangles include both corners
    low = apply_quantum_noise((image / 100).astype(np.float64), 1, rng)
    return np.where(mask, image, low).astype(np.uint16)


def clamp_translation(image, x_shift, y_shift=0):
    """script.py:97-120: shift, filling the uncovered band with the 99th percentile of a 2-pixel strip."""
    h, w = image.shape
    bright, margin = 2, 10
    left = margin if x_shift > 0 else 0
    right = w - margin if x_shift < 0 else w
    top = margin if y_shift > 0 else 0
    bottom = h - margin if y_shift < 0 else h
    cropped = image[top:bottom, left:right]
    b_right = margin + bright if x_shift > 0 else w
    b_bottom = margin + bright if y_shift > 0 else h
    fill = int(np.percentile(image[top:b_bottom, left:b_right], 99))
    out = np.full((h, w), fill, dtype=np.uint16)
    ys, xs = max(y_shift, 0), max(x_shift, 0)
    hh, ww = min(cropped.shape[0], h - ys), min(cropped.shape[1], w - xs)
    out[ys:ys + hh, xs:xs + ww] = cropped[:hh, :ww]
    return out


def clamp_rotate(image, degree):
    """script.py:122-141: rotate the image minus a 100-pixel margin (nearest neighbour, counter-clockwise),
    filling with the 95th percentile."""
    h, w = image.shape
    margin = min(100, h // 8)
    cropped = image[margin:h - margin, margin:w - margin]
    fill = int(np.percentile(cropped, 95))
    rot = ndimage.rotate(cropped, degree, reshape=False, order=0, mode="constant", cval=fill)
    out = np.full((h, w), fill, dtype=np.uint16)
    out[margin:h - margin, margin:w - margin] = rot
    return out


# ---- similarity metrics (script.py:143-198) ----------------------------------------------------------

def mse_similarity(a, b):
    """1 - RMSE / 255 (script.py:143-145)."""
    e = np.abs(a.astype(np.float64) - b.astype(np.float64)) / 255
    return 1.0 - math.sqrt(np.mean(np.square(e)))


def ssim_similarity(a, b):
    """skimage.metrics.structural_similarity with its defaults for uint8 input (script.py:147-152):
    7 x 7 uniform window, K1 = 0.01, K2 = 0.03, data range 255, sample covariance, borders cropped."""
    x, y = a.astype(np.float64), b.astype(np.float64)
    win = 7
    npx = win * win
    cov_norm = npx / (npx - 1)
    ux, uy = ndimage.uniform_filter(x, win), ndimage.uniform_filter(y, win)
    uxx, uyy, uxy = ndimage.uniform_filter(x * x, win), ndimage.uniform_filter(y * y, win), ndimage.uniform_filter(x * y, win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())


def hist_similarity(a, b):
    """(intersection, normalised Euclidean distance, Bhattacharyya coefficient) of the 256-bin histograms
    (script.py:154-198; np.histogram(bins=256) spans [min, max] of each image, as there)."""
    ha, _ = np.histogram(a.ravel(), bins=256)
    hb, _ = np.histogram(b.ravel(), bins=256)
    inter = np.sum(np.minimum(ha, hb)) / min(np.sum(ha), np.sum(hb))
    na, nb = ha / np.sum(ha), hb / np.sum(hb)
    e_dist = math.sqrt(np.sum((na - nb) ** 2)) / math.sqrt(2)
    b_coef = float(np.sum(np.sqrt(na * nb)))
    return float(inter), float(e_dist), b_coef


def similarities(a, b):
    inter, e_dist, b_coef = hist_similarity(a, b)
    return {"mse": mse_similarity(a, b), "ssim": ssim_similarity(a, b), "hist_intersection": inter,
            "hist_distance": e_dist, "hist_bhattacharyya": b_coef}


# ---- registration of the altered result onto the unaltered one (script.py:442-456, 484-508, 586-608) ----

def register_collimator(alt, unalt, shutter):
    x = y = shutter + PROCESSING_MARGIN
    w = alt.shape[1] - (2 * shutter + 2 * PROCESSING_MARGIN)
    h = alt.shape[0] - (2 * shutter + 2 * PROCESSING_MARGIN)
    return alt[y:y + h, x:x + w], unalt[y:y + h, x:x + w]


def register_translation_x(alt, unalt, tx):
    return alt[:, tx:], unalt[:, PROCESSING_MARGIN:alt.shape[1] - tx + PROCESSING_MARGIN]


def register_translation_y(alt, unalt, ty):
    return alt[ty:, :], unalt[PROCESSING_MARGIN:alt.shape[0] - ty + PROCESSING_MARGIN, :]


def register_rotation(alt, unalt, degree):
    h, w = unalt.shape
    ang = math.radians(degree)
    new_w = w * abs(math.cos(ang)) + h * abs(math.sin(ang))
    new_h = h * abs(math.cos(ang)) + w * abs(math.sin(ang))
    inner_w = w * h / new_h if w < h else h * w / new_w
    inner_h = h * w / new_w if w < h else w * h / new_h
    left, top = int((w - inner_w) / 2), int((h - inner_h) / 2)
    right, bottom = int((w + inner_w) / 2), int((h + inner_h) / 2)
    rot = ndimage.rotate(unalt, degree, reshape=False, order=0, mode="constant", cval=0)
    return alt[top:bottom, left:right], rot[top:bottom, left:right]


# ---- running the pipeline -------------------------------------------------------------------------

class Runner:
    """Processes raw images to the 8-bit output the reference's saveOutImage writes (margin cropped)."""

    def __init__(self, image_size, levels=0, device=0, use_cli=False):
        self.n, self.levels, self.device, self.use_cli = image_size, levels, device, use_cli
        self.proc = None
        if not use_cli:
            self.proc = mp.MusicaProcessing(device=device)
            if not self.proc.init(image_size, levels=levels):
                raise RuntimeError("musica_create failed: " + mp.last_error())

    def run(self, raw, workdir=None):
        """raw: (N, N) uint16 -> (N-20, N-20) uint8."""
        if self.use_cli:
            return self._run_cli(raw, workdir or ".")
        if not self.proc.execute(raw):
            raise RuntimeError("musica_execute failed: " + mp.last_error())
        return self.proc.out_pixels()

    def mean_cnr(self):
        """mean(cnr image) * 256 of the last run — what test/mean_cnr/script.py prints for a cnr.bmp dump."""
        return self.proc.stats().mean_cnr

    def _run_cli(self, raw, workdir):
        """run_process of script.py:200-214: write the raw file, spawn the CLI, read the BMP back."""
        from .phantom import write_raw
        raw_path, out_path = os.path.join(workdir, "in.raw"), os.path.join(workdir, "out.bmp")
        write_raw(raw_path, raw)
        cmd = [mp.CLI_PATH, os.path.abspath(raw_path), os.path.abspath(out_path), "--size", str(self.n), "--device", str(self.device)]
        if self.levels:
            cmd += ["--levels", str(self.levels)]
        subprocess.run(cmd, check=True, capture_output=True)
        return read_bmp_gray(out_path)

    def close(self):
        if self.proc:
            self.proc.cleanup()


def read_bmp_gray(path):
    """Reads the 24-bpp bottom-up BMP saveOutImage writes; returns the gray channel top-down."""
    b = open(path, "rb").read()
    off = int.from_bytes(b[10:14], "little")
    w, h = int.from_bytes(b[18:22], "little"), int.from_bytes(b[22:26], "little")
    row = (w * 3 + 3) & ~3
    a = np.frombuffer(b, dtype=np.uint8, count=row * h, offset=off).reshape(h, row)[:, 0:w * 3:3]
    return a[::-1].copy()


def run_study(raw, runner, rng=None, shutters=None, translations=None, rotations=None, sigmas=None, factors=None):
    """The reference's per-image loop (script.py:383-657): returns a list of rows
    {alteration, direct: {...}, registered: {...} or None, mean_cnr}."""
    rng = rng or np.random.default_rng(0)
    n = raw.shape[0]
    shutters = scaled(SHUTTERS, n) if shutters is None else shutters
    translations = scaled(TRANSLATIONS, n) if translations is None else translations
    rotations = ROTATIONS if rotations is None else rotations
    sigmas = GAUSS_SIGMAS if sigmas is None else sigmas
    factors = POISSON_FACTORS if factors is None else factors
    unalt = runner.run(raw)
    rows = [{"alteration": "unaltered", "direct": similarities(unalt, unalt), "registered": None, "mean_cnr": runner.mean_cnr() if runner.proc else None}]

    def add(name, altered_raw, reg=None):
        alt = runner.run(altered_raw)
        row = {"alteration": name, "direct": similarities(alt, unalt), "registered": None,
               "mean_cnr": runner.mean_cnr() if runner.proc else None}
        if reg is not None:
            a, u = reg(alt, unalt)
            if a.size and a.shape == u.shape and min(a.shape) >= 8:
                row["registered"] = similarities(a, u)
        rows.append(row)

    for s in shutters:
        add("c_sh_%d" % s, apply_collimator(raw, s, s, rng), lambda a, u, s=s: register_collimator(a, u, s))
    for t in translations:
        add("t_x_%d" % t, clamp_translation(raw, t, 0), lambda a, u, t=t: register_translation_x(a, u, t))
    for t in translations:
        add("t_y_%d" % t, clamp_translation(raw, 0, t), lambda a, u, t=t: register_translation_y(a, u, t))
    for d in rotations:
        add("r_%d" % d, clamp_rotate(raw, d), lambda a, u, d=d: register_rotation(a, u, d))
    for sg in sigmas:
        add("gn_%s" % sg, add_gaussian_noise(raw, 0.0, sg, rng))
    for f in factors:
        add("pn_%s" % f, apply_quantum_noise(raw, f, rng))
    return rows


# ---- command line: the reference's three CSV files (script.py:223-330) -------------------------
CSV_HEADER = ['raw file', 'alteration', 'altered vs unaltered mse', 'altered vs unaltered ssim', 'altered vs unaltered histogram distance',
              'altered vs reference mse', 'altered vs reference ssim', 'altered vs reference histogram distance',
              'normalized altered vs reference mse', 'normalized altered vs reference ssim',
              'normalized altered vs reference histogram distance']


def write_study_csvs(rows, out_dir, raw_name, mean_cnr=True):
    """direct_robustness.csv / reg_based_robustness.csv with the reference's column layout. The six "vs reference"
    columns compare with the vendor-processed image, which the reference tree does not ship (missing blobs): they
    stay empty. mean_cnr.csv adds what test/mean_cnr/script.py reports per alteration."""
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "direct_robustness.csv"), "w", newline="") as fd, \
            open(os.path.join(out_dir, "reg_based_robustness.csv"), "w", newline="") as fr:
        wd, wr = csv.writer(fd), csv.writer(fr)
        wd.writerow(CSV_HEADER)
        wr.writerow(CSV_HEADER)
        for r in rows:
            if r["alteration"] == "unaltered":
                continue
            d = r["direct"]
            wd.writerow([raw_name, r["alteration"], d["mse"], d["ssim"], d["hist_distance"]] + [""] * 6)
            if r["registered"] is not None:
                g = r["registered"]
                wr.writerow([raw_name, r["alteration"], g["mse"], g["ssim"], g["hist_distance"]] + [""] * 6)
    if mean_cnr:
        with open(os.path.join(out_dir, "mean_cnr.csv"), "w", newline="") as fc:
            wc = csv.writer(fc)
            wc.writerow(["raw file", "alteration", "mean cnr"])
            for r in rows:
                wc.writerow([raw_name, r["alteration"], r["mean_cnr"]])


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Metamorphic study of one raw image (or a seeded phantom) on the HIP MUSICA path")
    ap.add_argument("--raw", help="raw file: 256-byte header + N*N little-endian uint16 (test/standalone/main.cpp:54-75)")
    ap.add_argument("--phantom-seed", type=int, default=1, help="seed of the synthetic phantom used when --raw is absent")
    ap.add_argument("--size", type=int, default=3072, help="image side N (the reference's CLI fixes 3072)")
    ap.add_argument("--levels", type=int, default=0)
    ap.add_argument("--out", default="out", help="directory of the CSV files")
    ap.add_argument("--cli", action="store_true", help="run every image through the musica-standalone process (run_process, script.py:200-214)")
    args = ap.parse_args(argv)
    if args.raw:
        from .processing import read_raw
        raw = read_raw(args.raw, args.size)
        name = os.path.basename(args.raw)
    else:
        from .phantom import phantom
        raw = phantom(args.size, args.phantom_seed, noise=4.0)
        name = "phantom_%d_seed%d" % (args.size, args.phantom_seed)
    runner = Runner(args.size, args.levels, use_cli=args.cli)
    rows = run_study(raw, runner, rng=np.random.default_rng(0))
    runner.close()
    write_study_csvs(rows, args.out, name, mean_cnr=not args.cli)
    print("wrote %d alterations to %s" % (len(rows) - 1, args.out))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
