"""MUSICA_FLAG_REFERENCE_ORDER: the GPU in the shaders' LITERAL arithmetic order (SURVEY 8b's `reference-order` flag).

img_smooth.comp:32-45, img_smooth_upsampled.comp:32-45 (with `* 4.0` per tap) and img_sdev.comp:17-30 accumulate their 25
taps m (x) outer, n (y) inner, from 0. A context created with the flag runs one thread per texel in exactly that order and
must be BIT-IDENTICAL to the oracle's MUSICA_ORDER_REFERENCE — every f32 image, histogram, argmax, curve point, window
scalar, 8-bit pixel and the BMP file — on every BASELINE configuration at its full size and on the reference's own
3072 x 3072 / L = 12, in-process and through `musica-standalone --reference-order`.

The second half states what the DEFAULT (separable) order is worth against the literal one: both are now GPU contexts, the
literal one proven equal to the oracle above, so the distribution of the differences is measured at full size on every
configuration and the test asserts the bounds DESIGN.md section 2 quotes (and prints the measured numbers).
PARITY UNPINNED: the oracle is the build's restatement of the shaders; the reference holds no vectors for this path.
"""
import json
import os
import subprocess

import numpy as np
import pytest

from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom, write_raw
from test_gpu_parity import _compare_all, _proc, _same

pytestmark = pytest.mark.gpu

# (name, N, levels, seed, bits, flags): BASELINE configs[0], [1], [2], [4] and the reference's own configuration
# (configs[3], the 8-image shard, is its own test below)
CONFIGS = [
    ("configs0_512_L4", 512, 4, 1, 16, 0),
    ("configs1_2048_L6", 2048, 6, 2, 16, 0),
    ("configs2_4096_L8_clahe", 4096, 8, 3, 16, mp.FLAG_CLAHE),
    ("configs4_8192_L10_12bit", 8192, 10, 5, 12, 0),
    ("reference_3072_L12", 3072, 0, 31, 16, 0),
]


def _oracle_flags(ob, flags):
    return ob.FLAG_CLAHE if flags & mp.FLAG_CLAHE else 0


@pytest.mark.parametrize("name,n,levels,seed,bits,flags", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_reference_order_bit_identical_to_literal_oracle(ob, name, n, levels, seed, bits, flags):
    px = phantom(n, seed, bits=bits)
    o = ob.Oracle(n, levels, ob.ORDER_REFERENCE, _oracle_flags(ob, flags)).execute(px)
    p = _proc(n, levels, flags=flags | mp.FLAG_REFERENCE_ORDER)
    assert p.execute(px), mp.last_error()
    assert p.pyramidLevels == o.levels
    _compare_all(p, o, ob, tag=name + " (reference order): ")
    if flags & mp.FLAG_CLAHE:
        assert np.array_equal(p.clahe_hist(), o.clahe_hist())
        a, b = p.clahe_curves(), o.clahe_curves()
        assert ((a == b) | (np.isnan(a) & np.isnan(b))).all()
        _same(p.image(mp.IMG_CLAHE_GRADED), o.image(ob.IMG_CLAHE_GRADED), name + ": clahe graded")
    p.cleanup()


def test_reference_order_configs3_shard_in_one_batch(ob):
    n, levels, b = 2048, 6, 8
    px = np.stack([phantom(n, 100 + k) for k in range(b)])
    p = _proc(n, levels, batch=b, flags=mp.FLAG_REFERENCE_ORDER)
    p.upload(px)
    for _ in range(2):                                       # capture, then replay
        assert p.execute_device(), mp.last_error()
    p.sync()
    for k in range(b):
        o = ob.Oracle(n, levels, ob.ORDER_REFERENCE).execute(px[k])
        _compare_all(p, o, ob, idx=k, tag="configs[3] image %d (reference order): " % k)
    p.cleanup()


def test_reference_order_small_and_odd_sides(ob):
    # the 1 .. 7-pixel tail of the reference's level rule, sides that are not multiples of 8 / 4 / 2
    for n, levels, seed in [(200, 5, 4), (333, 0, 6), (256, 0, 3), (1000, 6, 5)]:
        px = phantom(n, seed)
        o = ob.Oracle(n, levels, ob.ORDER_REFERENCE).execute(px)
        p = _proc(n, levels, flags=mp.FLAG_REFERENCE_ORDER)
        assert p.execute(px), mp.last_error()
        _compare_all(p, o, ob, tag="%d/L%d (reference order): " % (n, levels))
        p.cleanup()


def test_cli_reference_order_3072(ob, tmp_path):
    n, seed = 3072, 31
    px = phantom(n, seed)
    raw, out, want = tmp_path / "image.raw", tmp_path / "out.bmp", tmp_path / "oracle.bmp"
    write_raw(str(raw), px)
    r = subprocess.run([mp.CLI_PATH, str(raw), str(out), "--reference-order"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ob.Oracle(n, 0, ob.ORDER_REFERENCE).execute(px).save_out_image(str(want))
    assert out.read_bytes() == want.read_bytes()
    # and the default order writes a different file only in a handful of pixels (measured below); both are valid BMPs of one size
    r = subprocess.run([mp.CLI_PATH, str(raw), str(tmp_path / "fast.bmp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert len((tmp_path / "fast.bmp").read_bytes()) == len(out.read_bytes())


# ---- what the default order is worth against the literal one ---------------------------------------------------------
def _distribution(p, q, levels):
    """default-order context p against reference-order context q (same input): the numbers DESIGN.md section 2 quotes."""
    d = {}
    d["downsampled_max"] = max(float(np.abs(p.image(mp.IMG_DOWNSAMPLED, i) - q.image(mp.IMG_DOWNSAMPLED, i)).max()) for i in range(levels))
    d["bandpass_max"] = max(float(np.abs(p.image(mp.IMG_BANDPASS, i) - q.image(mp.IMG_BANDPASS, i)).max()) for i in range(levels))
    d["sdev_max"] = max(float(np.abs(p.image(mp.IMG_SDEV, i) - q.image(mp.IMG_SDEV, i)).max()) for i in range(4))
    d["argmax_shift"] = [abs(int(p.noise_hist_max(i)[1]) - int(q.noise_hist_max(i)[1])) for i in range(4)]
    d["noise_hist_moved_fraction"] = max(
        float(np.abs(p.noise_hist(i).astype(np.int64) - q.noise_hist(i).astype(np.int64)).sum()) / max(1.0, float(q.noise_hist(i).sum())) for i in range(4))
    rec = np.abs(p.image(mp.IMG_EXPAND, 0) - q.image(mp.IMG_EXPAND, 0))
    d["recon_max"] = float(rec.max())
    d["recon_p9998"] = float(np.quantile(rec, 0.9998))
    d["recon_frac_above_4e-6"] = float((rec > 4e-6).mean())
    d["recon_frac_above_2e-3"] = float((rec > 2e-3).mean())
    g = np.abs(p.image(mp.IMG_GRADED) - q.image(mp.IMG_GRADED))
    d["graded_max"] = float(np.nanmax(g))
    d8 = np.abs(p.out_pixels().astype(np.int32) - q.out_pixels().astype(np.int32))
    d["out8_frac_differ"] = float((d8 != 0).mean())
    d["out8_frac_differ_by_more_than_1"] = float((d8 > 1).mean())
    d["out8_max"] = int(d8.max())
    gw_p, gw_q = p.grad_curve()[1], q.grad_curve()[1]
    d["grad_window_default"] = [float(v) for v in gw_p]
    d["grad_window_literal"] = [float(v) for v in gw_q]
    return d


@pytest.mark.parametrize("name,n,levels,seed,bits,flags", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_default_order_against_literal_order_distribution(name, n, levels, seed, bits, flags):
    """Bounds asserted (and quoted in DESIGN.md section 2 / bench.py's `parity` string):
    every stencil output within 6e-7 (downsampled: 7 ulp at 1.0 measured at 4096^2) / 1e-6 (band-pass, sdev); every noise-histogram argmax within one bin and
    at most 1 % of a histogram's counts in a different bin; reconstruction: max 5e-2 (a block under a cnr texel that sits
    within rounding distance of the noise-reduction thresholds 3 / 9, noise_reduction.comp:24-31: measured 2.1e-2 at configs[4]); with all four argmax equal at most
    0.02 % of the texels above 4e-6 and at most 0.1 % of the 8-bit pixels different; with an argmax one bin apart (the
    curve abscissae move by 1 / 2048 * 0.1 * ...) at most 0.1 % of the texels above 2e-3 and at most 1 % of the 8-bit
    pixels off by more than one grey level."""
    px = phantom(n, seed, bits=bits)
    p = _proc(n, levels, flags=flags)
    q = _proc(n, levels, flags=flags | mp.FLAG_REFERENCE_ORDER)
    assert p.execute(px) and q.execute(px), mp.last_error()
    L = p.pyramidLevels
    _same(p.image(mp.IMG_NORMALIZED), q.image(mp.IMG_NORMALIZED), "normalized")    # no stencil before this image
    d = _distribution(p, q, L)
    print("LITERAL_ORDER_DISTRIBUTION %s %s" % (name, json.dumps(d)))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "literal_order_%s.json" % name), "w") as f:
            json.dump(d, f, indent=1)
    assert d["downsampled_max"] <= 6e-7 and d["bandpass_max"] <= 1e-6 and d["sdev_max"] <= 1e-6
    assert max(d["argmax_shift"]) <= 1 and d["noise_hist_moved_fraction"] <= 0.01
    assert d["recon_max"] <= 5e-2      # 0.3 x |contrast-enhanced band| under a cnr texel that crosses 3 or 9 (measured: 2.1e-2 at configs[4])
    if max(d["argmax_shift"]) == 0:
        assert d["recon_frac_above_4e-6"] <= 2e-4
        assert d["out8_frac_differ"] <= 1e-3
    else:
        assert d["recon_frac_above_2e-3"] <= 1e-3
        assert d["out8_frac_differ_by_more_than_1"] <= 1e-2
    p.cleanup()
    q.cleanup()
