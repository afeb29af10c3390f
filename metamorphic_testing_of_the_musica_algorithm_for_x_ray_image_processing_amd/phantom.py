"""Seeded synthetic X-ray phantoms.

The reference's inputs (raw_images/{foot,hand,head,knee,pelvis,thorax}/image.raw,
3072 x 3072 little-endian uint16 behind a 256-byte header) are missing blobs
(.MISSING_LARGE_BLOBS), so every test and benchmark runs on phantoms of the
same shape: detector counts = I0 * exp(-attenuation) with smooth soft-tissue
blobs, sharp-edged "bone" ellipses and rectangles, a direct-exposure
background, and signal-dependent (Poisson-like) noise. No pixel is ever 0
(zeros trigger the early exits of noise_hist.comp:29 and
gradation_histogram.comp:24), and the structure extends well inside the
100-pixel border img_relevant.comp:21 ignores.
"""
import os

import numpy as np


def phantom(image_size, seed, bits=16, noise=1.0):
    """Return an (N, N) uint16 phantom. `bits` = 16 -> counts in [2000, 60000]; 12 -> [150, 4000]."""
    n = int(image_size)
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(-1.0, 1.0, n, dtype=np.float32),
                         np.linspace(-1.0, 1.0, n, dtype=np.float32), indexing="ij")
    att = np.zeros((n, n), dtype=np.float32)
    # soft tissue: broad Gaussian blobs
    for _ in range(6):
        cx, cy = rng.uniform(-0.6, 0.6, size=2)
        sx, sy = rng.uniform(0.15, 0.5, size=2)
        amp = rng.uniform(0.3, 1.2)
        att += amp * np.exp(-(((xx - cx) / sx) ** 2 + ((yy - cy) / sy) ** 2)).astype(np.float32)
    # bones: rotated ellipses with sharp edges and a denser cortex
    for _ in range(5):
        cx, cy = rng.uniform(-0.5, 0.5, size=2)
        a, b = rng.uniform(0.05, 0.35), rng.uniform(0.02, 0.12)
        th = rng.uniform(0, np.pi)
        xr = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        yr = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        r = (xr / a) ** 2 + (yr / b) ** 2
        att += np.where(r < 1.0, 0.6, 0.0).astype(np.float32)
        att += np.where((r < 1.0) & (r > 0.7), 0.5, 0.0).astype(np.float32)
    # an implant-like rectangle and a step wedge
    x0, y0 = rng.uniform(-0.7, 0.3, size=2)
    att += np.where((xx > x0) & (xx < x0 + 0.25) & (yy > y0) & (yy < y0 + 0.08), 1.0, 0.0).astype(np.float32)
    for k in range(5):
        att += np.where((xx > -0.9 + 0.1 * k) & (xx < -0.8 + 0.1 * k) & (yy > 0.7) & (yy < 0.85), 0.25 * (k + 1), 0.0).astype(np.float32)
    # fine trabecular-like texture
    tex = rng.standard_normal((n // 8 + 2, n // 8 + 2)).astype(np.float32)
    tex = np.kron(tex, np.ones((8, 8), dtype=np.float32))[:n, :n]
    att += 0.03 * tex * (att > 0.5)
    if bits == 16:
        lo, hi = 2000.0, 60000.0
    else:
        lo, hi = 150.0, 4000.0
    inten = hi * np.exp(-att)
    inten = np.maximum(inten, lo).astype(np.float32)
    # signal-dependent noise (normal approximation of Poisson counts)
    inten = inten + noise * np.sqrt(inten) * rng.standard_normal((n, n)).astype(np.float32)
    top = 65535.0 if bits == 16 else 4095.0
    return np.clip(np.rint(inten), 1.0, top).astype(np.uint16)


def phantom_batch(image_size, seeds, bits=16, noise=1.0):
    return np.stack([phantom(image_size, s, bits=bits, noise=noise) for s in seeds])


def write_raw(path, pixels):
    """Write the reference's raw format: 256 zero bytes + N*N little-endian uint16
    (test/metamorphic_test/script.py:39-47 writes the same header)."""
    px = np.ascontiguousarray(pixels, dtype="<u2")
    with open(path, "wb") as f:
        f.write(b"\x00" * 256)
        f.write(px.tobytes())
    return os.path.getsize(path)
