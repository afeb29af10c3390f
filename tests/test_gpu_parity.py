"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (stated here so the numbers live next to the assertions):
  * HIP vs oracle in MUSICA_ORDER_FAST (the arithmetic order the kernels implement): every f32
    image BIT-IDENTICAL (compared as float arrays, so -0 == +0), every histogram / argmax / curve
    point / window scalar EXACTLY equal, the final 8-bit pixels identical;
  * HIP vs oracle in MUSICA_ORDER_REFERENCE (literal 25-tap accumulation of the shaders): pyramid
    images within 4e-6 absolute on [0, 1]-scaled data, noise-histogram argmax within 1 bin,
    final 8-bit image: at most 0.1 % of pixels differ (getY's x > 1 -> 0 discontinuity and bin
    flips at 1e-7-level differences are inherent to the reference's semantics).
PARITY UNPINNED note: the oracle itself is pinned only by the analytic KATs in test_oracle_kat.py
(no golden vectors exist in the reference for this path).
"""
import numpy as np
import pytest

from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom

pytestmark = pytest.mark.gpu


def _proc(n, levels=0, batch=1, flags=0):
    p = mp.MusicaProcessing()
    assert p.init(n, levels=levels, batch=batch, flags=flags), mp.last_error()
    return p


def _same(a, b, what):
    assert a.shape == b.shape, what
    eq = (a == b) | (np.isnan(a) & np.isnan(b))
    if not eq.all():
        bad = np.argwhere(~eq)
        y, x = bad[0][-2], bad[0][-1]
        raise AssertionError("%s: %d / %d texels differ, first at (x=%d, y=%d): hip=%r oracle=%r, max abs diff %g"
                             % (what, len(bad), a.size, x, y, a[tuple(bad[0])], b[tuple(bad[0])], np.nanmax(np.abs(a - b))))


def _compare_all(p, o, ob, idx=0, tag=""):
    L = o.levels
    _same(p.image(mp.IMG_NORMALIZED, 0, idx), o.image(ob.IMG_NORMALIZED), tag + "normalized")
    assert p.minmax(idx) == o.minmax()
    for i in range(L):
        _same(p.image(mp.IMG_DOWNSAMPLED, i, idx), o.image(ob.IMG_DOWNSAMPLED, i), tag + "downsampled[%d]" % i)
        _same(p.image(mp.IMG_BANDPASS, i, idx), o.image(ob.IMG_BANDPASS, i), tag + "bandpass[%d]" % i)
    for i in range(4):
        _same(p.image(mp.IMG_SDEV, i, idx), o.image(ob.IMG_SDEV, i), tag + "sdev[%d]" % i)
        assert np.array_equal(p.noise_hist(i, idx), o.noise_hist(i)), tag + "noise_hist[%d]" % i
        assert p.noise_hist_max(i, idx) == o.noise_hist_max(i), tag + "noise_hist_max[%d]" % i
    for i in range(L):
        assert np.array_equal(p.contrast_curve(i, idx), o.contrast_curve(i)), tag + "contrast_curve[%d]" % i
        assert p.contrast_params(i) == o.contrast_params(i)
    for i in range(3):
        assert p.nr_params(i) == o.nr_params(i)
    _same(p.image(mp.IMG_CNR, 3, idx), o.image(ob.IMG_CNR, 3), tag + "cnr")
    for i in reversed(range(L)):
        _same(p.image(mp.IMG_EXPAND, i, idx), o.image(ob.IMG_EXPAND, i), tag + "expand[%d]" % i)
    assert np.array_equal(p.grad_hist(idx), o.grad_hist()), tag + "grad_hist"
    assert p.grad_hist_max(idx) == o.grad_hist_max()
    gc, gw = p.grad_curve(idx)
    oc, ow = o.grad_curve()
    assert np.array_equal(gc, oc) and gw == ow, tag + "grad_curve"
    _same(p.image(mp.IMG_GRADED, 0, idx), o.image(ob.IMG_GRADED), tag + "graded")
    assert np.array_equal(p.out_pixels(idx), o.out_pixels()), tag + "out pixels"
    # on-demand (debugProcess) images
    _same(p.image(mp.IMG_RELEVANT, 0, idx), o.image(ob.IMG_RELEVANT), tag + "relevant")
    _same(p.image(mp.IMG_SQRT, 0, idx), o.image(ob.IMG_SQRT), tag + "sqrt")
    for i in (0, L - 1):
        _same(p.image(mp.IMG_LOWPASS, i, idx), o.image(ob.IMG_LOWPASS, i), tag + "lowpass[%d]" % i)
        _same(p.image(mp.IMG_EXP_BANDPASS, i, idx), o.image(ob.IMG_EXP_BANDPASS, i), tag + "exp_bandpass[%d]" % i)
    hs, os_ = p.stats(idx), o.stats()
    assert list(hs.noise_max_bin) == list(os_.noise_max_bin) and hs.grad_max_bin == os_.grad_max_bin
    assert (hs.t0, hs.ta, hs.t1) == (os_.t0, os_.ta, os_.t1)
    if np.isfinite(os_.mean_cnr):
        assert abs(hs.mean_cnr - os_.mean_cnr) <= 1e-5 * max(1.0, abs(os_.mean_cnr))   # f64 sums in a different order
    else:                                                                               # flat image: x / 0 everywhere
        assert (np.isnan(hs.mean_cnr) and np.isnan(os_.mean_cnr)) or hs.mean_cnr == os_.mean_cnr


# configs[0] of BASELINE.json (512, L = 4), the reference's level rule (L = ceil(log2 N)) down to
# 2x2 / 1x1 levels, sizes that are not multiples of 8 / 4 / 2, a power of 8 (exact min chain) and
# the 2048 / L6 case of configs[1].
CASES = [(512, 4, 1), (512, 0, 2), (256, 0, 3), (200, 5, 4), (1000, 6, 5), (333, 0, 6), (1024, 6, 7), (2048, 6, 8), (1792, 0, 9)]


@pytest.mark.parametrize("dispatch", ["default", "graph2"])
@pytest.mark.parametrize("n,levels,seed", CASES)
def test_pipeline_bit_exact_vs_fast_oracle(ob, n, levels, seed, dispatch, monkeypatch):
    # "default": what musica_create picks for a lone context of this size (one image up to 3072^2: eager launches on one
    # stream); "graph2": the captured two-stream graph that larger steps replay
    if dispatch == "graph2":
        monkeypatch.setenv("MUSICA_GRAPH", "1")
        monkeypatch.setenv("MUSICA_STREAMS", "2")
    px = phantom(n, seed)
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px)
    p = _proc(n, levels)
    assert p.execute(px), mp.last_error()
    assert p.pyramidLevels == o.levels
    _compare_all(p, o, ob)
    p.cleanup()


def test_twelve_bit_input_and_second_execute_is_idempotent(ob):
    n = 768
    px = phantom(n, 21, bits=12)
    assert px.max() <= 4095
    o = ob.Oracle(n, 6, ob.ORDER_FAST).execute(px)
    p = _proc(n, 6)
    assert p.execute(px)
    _compare_all(p, o, ob)
    first = p.graded().copy()
    assert p.execute(px)                     # histograms are cleared per execute (src/vk_processing.cpp:2153-2162)
    assert np.array_equal(first, p.graded())
    _compare_all(p, o, ob)
    p.cleanup()


def test_batch_of_independent_images(ob):
    n, levels, b = 512, 5, 3
    px = np.stack([phantom(n, 100 + k) for k in range(b)])
    p = _proc(n, levels, batch=b)
    assert p.execute(px)
    for k in range(b):
        o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px[k])
        _compare_all(p, o, ob, idx=k, tag="image %d: " % k)
    p.cleanup()


def test_resident_input_entry_point(ob):
    n = 512
    px = phantom(n, 33)
    p = _proc(n, 5)
    p.upload(px)
    assert p.execute_device()
    p.sync()
    o = ob.Oracle(n, 5, ob.ORDER_FAST).execute(px)
    _same(p.graded()[0], o.image(ob.IMG_GRADED), "graded (resident input)")
    p.cleanup()


def test_default_dispatch_per_workload(monkeypatch):
    """What musica_create picks when nothing overrides it: one image below 2048^2 and batches of up to 3072^2
    texels: one stream, eager; everything larger: two streams, graph replay (eager for one image with 11 or more levels); pipeline
    contexts (MUSICA_FLAG_LINEAR): one stream + graph; one-shot contexts (the CLI's flags): one stream, eager."""
    for v in ("MUSICA_STREAMS", "MUSICA_GRAPH"):
        monkeypatch.delenv(v, raising=False)
    cases = [((512, 4, 1, 0), (1, False)), ((1024, 5, 4, 0), (1, False)), ((2048, 6, 1, 0), (2, True)), ((2048, 0, 1, 0), (2, False)),
             ((2048, 6, 8, 0), (2, True)), ((2048, 6, 2, 0), (1, False)), ((2048, 0, 8, 0), (2, True)), ((2048, 6, 8, mp.FLAG_LINEAR), (1, True)), ((2048, 6, 1, mp.FLAG_ONE_SHOT), (1, False)), ((2048, 6, 1, mp.FLAG_NO_AUTOTUNE | mp.FLAG_NO_GRAPH), (2, False)),
             ((4096, 8, 1, mp.FLAG_CLAHE), (2, True))]
    for (n, levels, batch, flags), want in cases:
        p = _proc(n, levels, batch=batch, flags=flags)
        assert p.dispatch() == want, (n, levels, batch, flags, p.dispatch_text())
        p.cleanup()


@pytest.mark.parametrize("batch", [1, 3])
def test_dispatch_forms_give_the_same_bits(ob, batch, monkeypatch):
    """MUSICA_STREAMS = 1 (one in-order stream, the reference's order) and 2 (two streams: the analysis beside the
    reduce tail), graph replay and eager: every form against the oracle, twice in a row (histograms re-cleared, events re-armed),
    and musica_get_dispatch reports the form."""
    n, levels = 1032, 6
    px = np.stack([phantom(n, 800 + k) for k in range(batch)])
    want = [ob.Oracle(n, levels, ob.ORDER_FAST).execute(px[k]) for k in range(batch)]
    for dag in ("1", "2"):
        for flags, graph in ((0, "1"), (mp.FLAG_NO_GRAPH, "1"), (0, "0")):
            monkeypatch.setenv("MUSICA_STREAMS", dag)
            monkeypatch.setenv("MUSICA_GRAPH", graph)
            p = _proc(n, levels, batch=batch, flags=flags)
            assert p.dispatch() == (int(dag), flags == 0 and graph == "1")
            for rep in range(2):
                assert p.execute(px)
            for k in range(batch):
                _compare_all(p, want[k], ob, idx=k, tag="dag %s flags %d image %d: " % (dag, flags, k))
            p.cleanup()


@pytest.mark.parametrize("env", [{"MUSICA_FUSE_GH": "0"}, {"MUSICA_XCD_SWIZZLE": "0"}, {"MUSICA_GRAD_ONE_LAUNCH": "0"}, {"MUSICA_TINY_TAIL": "0"},
                                 {"MUSICA_STREAMS": "1", "MUSICA_PAIR_RB_SDEV": "1"}, {"MUSICA_STREAMS": "1", "MUSICA_PAIR_RB_SDEV": "1", "MUSICA_SDEV_IN_EXPAND": "1", "MUSICA_AUTOTUNE": "0", "MUSICA_SDEV_RUN": "0"},
                                 {"MUSICA_STREAMS": "1", "MUSICA_PAIR_RB_SDEV": "1", "MUSICA_TINY_TAIL": "0", "MUSICA_XCD_SWIZZLE": "0"},
                                 {"MUSICA_SDEV_ONE_LAUNCH": "1"}, {"MUSICA_SDEV_ONE_LAUNCH": "1", "MUSICA_AUTOTUNE": "0", "MUSICA_SDEV_RUN": "0", "MUSICA_SDEV_IN_EXPAND": "1"},
                                 {"MUSICA_SDEV_ONE_LAUNCH": "1", "MUSICA_XCD_SWIZZLE": "0", "MUSICA_AUTOTUNE": "0", "MUSICA_SDEV_RUN": "0"}, {"MUSICA_SDEV_ONE_LAUNCH": "0", "MUSICA_SDEV_IN_EXPAND": "1"},
                                 {"MUSICA_AUTOTUNE": "0", "MUSICA_SDEV_RUN": "0"}, {"MUSICA_AUTOTUNE": "0", "MUSICA_SDEV_RUN": "1"},
                                 {"MUSICA_AUTOTUNE": "0", "MUSICA_RB_ROWS": "1"}, {"MUSICA_AUTOTUNE": "0", "MUSICA_RB_ROWS": "3"}, {"MUSICA_AUTOTUNE": "0", "MUSICA_RB_ROWS": "4"}, {"MUSICA_AUTOTUNE": "0", "MUSICA_RB_ROWS": "64"},
                                 {"MUSICA_AUTOTUNE": "0", "MUSICA_RB_ROWS": "5", "MUSICA_EXPAND_ROWS": "2", "MUSICA_SDEV_ROWS": "16"},
                                 {"MUSICA_AUTOTUNE": "0", "MUSICA_RB_ROWS": "32", "MUSICA_EXPAND_ROWS": "16", "MUSICA_SDEV_ROWS": "64", "MUSICA_SDEV_RUN": "0"}],
                         ids=lambda e: ",".join("%s=%s" % (k[7:], v) for k, v in e.items()))
def test_kernel_variants_and_launch_geometries_give_the_same_bits(ob, env, monkeypatch):
    """The fallback forms the library keeps because some context needs them anyway (separate gradation histogram, plain tile mapping,
    one launch per tail level, both sdev forms) and extreme / odd rows-per-wavefront choices: all bit-identical to the oracle."""
    n, levels = 1024, 6
    px = phantom(n, 900)
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    p = _proc(n, levels)
    assert p.execute(px)
    _compare_all(p, o, ob, tag=str(env) + ": ")
    p.cleanup()


@pytest.mark.parametrize("dag,one_launch", [("1", "1"), ("2", "1"), ("1", "0")])
def test_exact_zeros_in_the_reconstruction_take_the_literal_histogram(ob, dag, one_launch, monkeypatch):
    """A collimated image (test/metamorphic_test/script.py's collimator alteration blacks out a frame): raw zeros give
    normalized 0, band 0 and — far enough inside — a reconstruction that is exactly 0, where the reference's histogram
    thread `return`s (gradation_histogram.comp:24) and the noise histogram `break`s (noise_hist.comp:29). The level-0
    expand kernel that bins on the fly must hand such an image to the literal kernel; an image of the same batch
    without zeros keeps the fused count. Frame edges are not multiples of 16, so areas with texels on both sides exist.
    The recount and the tone curve are one launch (k_grad_recount_curve: 64 workgroups recount, the one that draws the last ticket
    builds the curve; three executes in a row: the tickets re-arm); MUSICA_GRAD_ONE_LAUNCH=0 is the two-launch form."""
    monkeypatch.setenv("MUSICA_STREAMS", dag)
    monkeypatch.setenv("MUSICA_GRAD_ONE_LAUNCH", one_launch)
    n, levels = 1024, 4
    a = phantom(n, 61)
    a[:203, :] = 0
    a[:, :187] = 0
    a[-230:, :] = 0
    b = phantom(n, 62)
    px = np.stack([a, b])
    p = _proc(n, levels, batch=2)
    assert p.fuses_gradhist()
    for rep in range(2):
        assert p.execute(px)
    for k in range(2):
        o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px[k])
        if k == 0:
            assert (o.image(ob.IMG_EXPAND, 0) == 0.0).sum() > 1000
        _compare_all(p, o, ob, idx=k, tag="zeros image %d: " % k)
    assert p.execute(px[::-1].copy())           # the flag is per image and re-armed per execute
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px[1])
    _compare_all(p, o, ob, idx=0, tag="swapped: ")
    p.cleanup()


def test_streaming_execute_overlaps_copies_and_keeps_every_batch_exact(ob):
    """musica_execute_stream: a sequence of different batches through the two-buffer pipeline (pageable and pinned inputs,
    odd count so both device buffers and both cached graphs are reused): every image's stats row equals the oracle's, the
    context ends up holding the last batch, and a plain execute afterwards still works."""
    n, levels, b, count = 520, 5, 2, 5
    p = _proc(n, levels, batch=b)
    seq = [np.stack([phantom(n, 1000 + 10 * j + k) for k in range(b)]) for j in range(count)]
    pinned = p.host_alloc(seq[1].shape)
    pinned[...] = seq[1]
    ok, st = p.execute_stream([seq[0], pinned, seq[2], seq[3], seq[4]], want_stats=True)
    assert ok, mp.last_error()
    assert len(st) == count * b
    for j in range(count):
        for k in range(b):
            o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(seq[j][k])
            got, want = st[j * b + k], o.stats()
            assert got.image_id == j * b + k
            assert list(got.noise_max_bin) == list(want.noise_max_bin) and got.grad_max_bin == want.grad_max_bin and got.grad_max_value == want.grad_max_value
            assert (got.t0, got.ta, got.t1, got.min_sqrt, got.max_sqrt) == (want.t0, want.ta, want.t1, want.min_sqrt, want.max_sqrt)
            assert abs(got.mean_cnr - want.mean_cnr) <= 1e-5 * max(1.0, abs(want.mean_cnr))
            if j == count - 1:
                _compare_all(p, o, ob, idx=k, tag="stream, last batch, image %d: " % k)
    p.host_free(pinned)
    assert p.execute_stream([seq[2]])                      # a sequence of one
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(seq[2][1])
    _same(p.graded()[1], o.image(ob.IMG_GRADED), "graded after a one-batch stream")
    assert p.execute(seq[0])
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(seq[0][0])
    _same(p.graded()[0], o.image(ob.IMG_GRADED), "graded after execute")
    p.cleanup()


def _random_cases(count, seed):
    """(N, levels, phantom seed, bits, noise) drawn once from a fixed generator: odd sizes, sizes around the strip
    (512) and vector (8) boundaries, the smallest accepted sides, reference-rule and explicit level counts."""
    rng = np.random.default_rng(seed)
    pool = [16, 17, 23, 24, 31, 32, 40, 63, 64, 65, 72, 100, 127, 128, 136, 255, 256, 264, 504, 511, 512, 513, 520,
            528, 600, 776, 1000, 1016, 1024, 1032, 1096, 1536, 1544]
    cases = []
    for _ in range(count):
        n = int(rng.choice(pool))
        lref = int(np.ceil(np.log2(n)))
        levels = int(rng.choice([0, 4, min(5, lref), min(6, lref)]))
        cases.append((n, levels, int(rng.integers(1, 10_000)), int(rng.choice([12, 16])), float(rng.choice([0.0, 1.0, 4.0]))))
    return cases


@pytest.mark.parametrize("n,levels,seed,bits,noise", _random_cases(40, 2024))
def test_random_configurations_match_the_oracle(ob, n, levels, seed, bits, noise):
    px = phantom(n, seed, bits=bits, noise=noise)
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px)
    p = _proc(n, levels)
    assert p.execute(px), mp.last_error()
    _compare_all(p, o, ob, tag="N=%d L=%d seed=%d: " % (n, levels, seed))
    p.cleanup()


def test_largest_baseline_size_matches_the_oracle(ob):
    """BASELINE configs[4]: 8192 x 8192, 12-bit, 10-level pyramid — final pixels, histograms, curves and the level-0
    images against the oracle (which needs a few seconds and ~4 GB for it)."""
    n, levels = 8192, 10
    px = phantom(n, 5, bits=12)
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px)
    p = _proc(n, levels)
    assert p.execute(px)
    assert np.array_equal(p.out_pixels(0), o.out_pixels())
    _same(p.image(mp.IMG_GRADED, 0, 0), o.image(ob.IMG_GRADED), "graded 8192")
    _same(p.image(mp.IMG_BANDPASS, 0, 0), o.image(ob.IMG_BANDPASS, 0), "bandpass[0] 8192")
    _same(p.image(mp.IMG_EXPAND, 0, 0), o.image(ob.IMG_EXPAND, 0), "expand[0] 8192")
    for i in range(4):
        assert np.array_equal(p.noise_hist(i, 0), o.noise_hist(i))
    assert np.array_equal(p.grad_hist(0), o.grad_hist())
    gc, gw = p.grad_curve(0)
    oc, ow = o.grad_curve()
    assert np.array_equal(gc, oc) and gw == ow
    p.cleanup()


def test_maximum_size_runs_and_is_deterministic():
    """The largest accepted side (16384: a 1 GiB level-0 plane, 32-bit buffer offsets up to 2^30) — no oracle at this
    size in the test budget; properties instead: two executes give identical bits, the output is finite and spans
    the 8-bit range, the min / max scalars are those of the input; one more pixel of side is refused."""
    n = 16384
    rng = np.random.default_rng(3)
    base = phantom(2048, 9)
    px = np.tile(base, (8, 8))
    px = (px.astype(np.int64) + rng.integers(0, 8, size=(n, 1))).clip(1, 65535).astype(np.uint16)   # rows differ: no exact tiling
    p = _proc(n, 6)
    assert p.execute(px)
    a = p.out_pixels(0).copy()
    mn, mx = p.minmax(0)
    assert mx == np.floor(np.sqrt(float(px.max()))) and mn == 0.0          # 16384 is not a power of 8: min chain -> 0
    assert p.execute(px)
    assert np.array_equal(a, p.out_pixels(0))
    g = p.image(mp.IMG_GRADED, 0, 0)
    assert np.isfinite(g).all() and a.min() < 32 and a.max() > 200
    p.cleanup()
    q = mp.MusicaProcessing()
    assert not q.init(n + 8, levels=6)
    assert "out of range" in mp.last_error()


def test_exact_math_shortcuts_on_the_device():
    """csrc/exact_math.h: the rsq-based sqrt (single and 8-wide grouped, +0 mixed in) against sqrtf over all
    2^32 float patterns, and the shortcut normalisation against the literal one over every (pixel, min, max)."""
    p = _proc(64, 4)
    assert p.selftest_exact_math() == [0, 0, 0, 0]
    p.cleanup()


@pytest.mark.parametrize("lo,hi", [(0, 65535), (900, 65535), (0, 4095), (4096, 4160)])
def test_every_raw_value_normalises_like_the_oracle(ob, lo, hi):
    """A 256 x 256 image holding every value of [lo, hi] (all 65536 for the first case): pins the hardware sqrt
    of the u16-fused level-0 kernels for each possible pixel against the CPU's sqrtf, min chain both ways
    (256 is not a power of 8 -> min 0; 512 is -> min floor(sqrt(lo)))."""
    for n in (256, 512):
        vals = (lo + np.arange(n * n, dtype=np.int64) % (hi - lo + 1)).astype(np.uint16)
        px = np.random.default_rng(lo + n).permutation(vals).reshape(n, n)
        o = ob.Oracle(n, 4, ob.ORDER_FAST).execute(px)
        p = _proc(n, 4)
        assert p.execute(px)
        _compare_all(p, o, ob)
        p.cleanup()


def test_graph_replay_equals_eager_launches_and_follows_the_input_pointer(ob):
    """The captured hipGraph (default) and MUSICA_FLAG_NO_GRAPH give the same bits; a caller-owned input
    buffer at another address re-captures instead of replaying stale pointers; enabling per-kernel
    profiling drops to eager launches and back without changing the result."""
    n, levels, b = 520, 5, 2
    px1 = np.stack([phantom(n, 300 + k) for k in range(b)])
    px2 = np.stack([phantom(n, 400 + k) for k in range(b)])
    g, e = _proc(n, levels, batch=b), _proc(n, levels, batch=b, flags=mp.FLAG_NO_GRAPH)
    for px in (px1, px2, px1):               # replay on the library's own input buffer with changing contents
        assert g.execute(px) and e.execute(px)
        _same(g.graded(), e.graded(), "graded (graph vs eager)")
        for k in range(b):
            assert np.allclose(g.stats(k).as_row(), e.stats(k).as_row(), rtol=1e-5, atol=0)   # mean_cnr: f64 atomics
    want1, want2 = e.graded().copy(), None
    assert e.execute(px2)
    want2 = e.graded().copy()
    d1, d2 = g.device_alloc(px1.nbytes), g.device_alloc(px2.nbytes)
    g.h2d(d1, px1)
    g.h2d(d2, px2)
    for d, want in ((d1, want1), (d2, want2), (d1, want1), (None, None)):
        if d is None:
            g.upload(px2)
            want = want2
        assert g.execute_device(d)
        g.sync()
        _same(g.graded(), want, "graded (caller-owned input %s)" % d)
    g.profile_enable(True)
    assert g.execute(px1)
    _same(g.graded(), want1, "graded (profiling, eager)")
    assert g.profile()["reduce_l0"][1] == 1 and g.profile()["grad_apply"][0] > 0
    g.profile_enable(False)
    assert g.execute(px2)
    _same(g.graded(), want2, "graded (graph again)")
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px2[1])
    _compare_all(g, o, ob, idx=1, tag="graph: ")
    g.device_free(d1)
    g.device_free(d2)
    g.cleanup()
    e.cleanup()


@pytest.mark.parametrize("n,levels", [(512, 5), (1000, 6), (264, 0)])
def test_streaming_kernels_equal_generic_kernels(n, levels):
    px = phantom(n, 77)
    a, b = _proc(n, levels), _proc(n, levels, flags=mp.FLAG_GENERIC_KERNELS)
    assert a.execute(px) and b.execute(px)
    for i in range(a.pyramidLevels):
        _same(a.image(mp.IMG_DOWNSAMPLED, i), b.image(mp.IMG_DOWNSAMPLED, i), "downsampled[%d]" % i)
        _same(a.image(mp.IMG_BANDPASS, i), b.image(mp.IMG_BANDPASS, i), "bandpass[%d]" % i)
        _same(a.image(mp.IMG_EXPAND, i), b.image(mp.IMG_EXPAND, i), "expand[%d]" % i)
    _same(a.graded(), b.graded(), "graded")
    a.cleanup()
    b.cleanup()


@pytest.mark.parametrize("side", [8, 16, 24, 64, 100, 257, 512, 520, 1024, 1032, 2048, 3080, 3, 2, 1, 5, 7, 12])
def test_metric_kernel_vs_oracle(ob, side):
    rng = np.random.default_rng(side)
    img = rng.random((2, side, side), dtype=np.float32)
    p = _proc(64, 4)
    got = p.k_reduce_host(img)
    for k in range(2):
        expect = ob.k_downsample(ob.k_smooth(img[k], ob.ORDER_FAST))
        _same(got[k], expect, "smooth+downsample side %d image %d" % (side, k))
        literal = ob.k_downsample(ob.k_smooth(img[k], ob.ORDER_REFERENCE))
        assert np.abs(got[k] - literal).max() <= 4e-7     # 25 taps, values in [0, 1]
    p.cleanup()


@pytest.mark.parametrize("side", [4096, 8192])
def test_metric_kernel_properties_full_size(side):
    # size-independent properties at BASELINE's 4096 x 4096 and at the 8192 x 8192 bench.py also times: a constant stays
    # constant (sum w = 1), and the operator is linear: R(a + b) ~= R(a) + R(b)
    p = _proc(64, 4)
    const = np.full((1, side, side), 0.75, dtype=np.float32)
    out = p.k_reduce_host(const)
    assert out.shape == (1, side // 2, side // 2)
    assert np.abs(out - 0.75).max() <= 2e-7
    rng = np.random.default_rng(side)
    a = rng.random((1, side, side), dtype=np.float32)
    b = rng.random((1, side, side), dtype=np.float32)
    ra, rb, rab = p.k_reduce_host(a), p.k_reduce_host(b), p.k_reduce_host(a + b)
    assert np.abs(rab - (ra + rb)).max() <= 1e-6
    p.cleanup()


@pytest.mark.parametrize("n,levels,seed", [(512, 4, 1), (1024, 6, 7), (2048, 6, 100)])
def test_pipeline_close_to_literal_oracle(ob, n, levels, seed):
    """HIP (separable order) against the oracle in the shaders' LITERAL 25-tap order: this is the tolerance the
    build claims against the reference's arithmetic (parity with the reference itself is unpinned). Stencil
    outputs within 4e-7 / 1e-6; histogram argmax within one bin. The end of the pipeline is then bounded in BOTH
    situations: when every noise-histogram argmax agrees (the contrast curves are then identical) the
    reconstruction is within 4e-6 on all but 0.02 % of the texels and at most 0.1 % of the 8-bit pixels differ —
    the exceptions sit under cnr texels within rounding distance of the noise-reduction thresholds 3 and 9, where the
    reference's linearFunction (noise_reduction.comp:24-31, `m * c + lowFactor`) jumps by 3 m = 0.2 ... 0.3 of the band
    value, so a 1e-7 difference in cnr moves a whole 8 x 8 block by up to ~5e-3; when an argmax moved by one bin (a tie
    broken by 1e-7-level differences: p changes by 1 / 2048 * 0.1, the curve abscissae with it) the reconstruction is
    within 2e-3 on 99.9 % of the texels and the 8-bit image within one grey level on 99 % of the pixels."""
    px = phantom(n, seed)
    o = ob.Oracle(n, levels, ob.ORDER_REFERENCE).execute(px)
    p = _proc(n, levels)
    assert p.execute(px)
    _same(p.image(mp.IMG_NORMALIZED), o.image(ob.IMG_NORMALIZED), "normalized")
    for i in range(o.levels):
        assert np.abs(p.image(mp.IMG_DOWNSAMPLED, i) - o.image(ob.IMG_DOWNSAMPLED, i)).max() <= 4e-7
        assert np.abs(p.image(mp.IMG_BANDPASS, i) - o.image(ob.IMG_BANDPASS, i)).max() <= 1e-6
    shifts = []
    for i in range(4):
        assert np.abs(p.image(mp.IMG_SDEV, i) - o.image(ob.IMG_SDEV, i)).max() <= 1e-6
        shifts.append(abs(int(p.noise_hist_max(i)[1]) - int(o.noise_hist_max(i)[1])))
        assert shifts[-1] <= 1
        assert np.abs(p.noise_hist(i).astype(np.int64) - o.noise_hist(i).astype(np.int64)).sum() <= 0.01 * o.noise_hist(i).sum() + 4
    rec = np.abs(p.image(mp.IMG_EXPAND, 0) - o.image(ob.IMG_EXPAND, 0))
    d8 = np.abs(p.out_pixels().astype(np.int32) - o.out_pixels().astype(np.int32))
    assert rec.max() <= 2e-2
    if max(shifts) == 0:
        assert (rec > 4e-6).mean() <= 2e-4
        assert (d8 != 0).mean() <= 1e-3
    else:
        print("noise-histogram argmax shifted by one bin at levels", [i for i in range(4) if shifts[i]])
        assert (rec > 2e-3).mean() <= 1e-3
        assert (d8 > 1).mean() <= 1e-2
    p.cleanup()


def test_analysis_stage_exact_given_identical_band(ob):
    # "bit-exact for the histogram/index reductions ... given identical f32 input": feed the SAME band
    # images (from the literal-order oracle) to both sides and run only the analysis stage.
    n, levels = 1024, 6
    px = phantom(n, 3)
    lit = ob.Oracle(n, levels, ob.ORDER_REFERENCE).execute(px)
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px)
    p = _proc(n, levels)
    assert p.execute(px)
    for i in range(levels):
        band = lit.image(ob.IMG_BANDPASS, i)
        p.set_image(mp.IMG_BANDPASS, i, band)
        o.set_image(ob.IMG_BANDPASS, i, band)
    p.run_stage(mp.STAGE_ANALYSIS)
    o.run_stage(ob.STAGE_ANALYSIS)
    for i in range(4):
        _same(p.image(mp.IMG_SDEV, i), o.image(ob.IMG_SDEV, i), "sdev[%d]" % i)
        assert np.array_equal(p.noise_hist(i), o.noise_hist(i))
        assert p.noise_hist_max(i) == o.noise_hist_max(i)
        assert np.array_equal(p.contrast_curve(i), o.contrast_curve(i))
    _same(p.image(mp.IMG_CNR, 3), o.image(ob.IMG_CNR, 3), "cnr")
    p.cleanup()


def test_gradation_stage_edge_cases(ob):
    # zeros inside the reconstruction (`return` semantics of gradation_histogram.comp:24), values > 1
    # (bins dropped, getY -> 0), NaN, and a non-monotone tone curve (t1 < ts) that forces the literal
    # first-match scan instead of the binary search.
    n, levels = 512, 5
    px = phantom(n, 9)
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px)
    p = _proc(n, levels)
    assert p.execute(px)
    rec = o.image(ob.IMG_EXPAND, 0).copy()
    rng = np.random.default_rng(0)
    ys, xs = rng.integers(0, n, 200), rng.integers(0, n, 200)
    rec[ys[:120], xs[:120]] = 0.0
    rec[ys[120:160], xs[120:160]] = 1.5
    rec[ys[160:180], xs[160:180]] = np.nan
    rec[ys[180:], xs[180:]] = -0.25
    # squeeze the relevant part of the histogram into a narrow range so that t1 < ts
    rec[150:360, 150:360] = 0.3 + 0.02 * rng.random((210, 210), dtype=np.float32)
    p.set_image(mp.IMG_EXPAND, 0, rec)
    o.set_image(ob.IMG_EXPAND, 0, rec)
    p.run_stage(mp.STAGE_GRADATION)
    o.run_stage(ob.STAGE_GRADATION)
    assert np.array_equal(p.grad_hist(), o.grad_hist())
    gc, gw = p.grad_curve()
    oc, ow = o.grad_curve()
    assert np.array_equal(gc, oc) and gw == ow
    _same(p.image(mp.IMG_GRADED), o.image(ob.IMG_GRADED), "graded")


def test_clahe_gradation(ob):
    n, levels = 512, 5
    px = phantom(n, 12)
    o = ob.Oracle(n, levels, ob.ORDER_FAST, ob.FLAG_CLAHE).execute(px)
    p = _proc(n, levels, flags=mp.FLAG_CLAHE)
    assert p.execute(px)
    assert np.array_equal(p.clahe_hist(), o.clahe_hist())
    a, b = p.clahe_curves(), o.clahe_curves()
    assert ((a == b) | (np.isnan(a) & np.isnan(b))).all()
    _same(p.image(mp.IMG_CLAHE_GRADED), o.image(ob.IMG_CLAHE_GRADED), "clahe graded")
    _same(p.image(mp.IMG_GRADED), o.image(ob.IMG_GRADED), "graded")      # the old gradation still runs (.cpp:2491-2518)
    p.cleanup()


def test_save_out_image_and_debug_process(ob, tmp_path):
    n, levels = 512, 5
    px = phantom(n, 41)
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px)
    p = _proc(n, levels)
    assert p.execute(px)
    a, b = tmp_path / "hip.bmp", tmp_path / "oracle.bmp"
    assert p.saveOutImage(str(a))
    o.save_out_image(str(b))
    assert a.read_bytes() == b.read_bytes()
    assert len(a.read_bytes()) == 54 + (n - 20) * ((n - 20) * 3)
    # debugProcess (src/vk_processing.cpp:2661-2756): every image dump byte for byte against the oracle's restatement
    # of the same function (quantisation of src/vk_state.cpp:834, one-component stbi_write_bmp)
    d, e = tmp_path / "dump", tmp_path / "oracle_dump"
    d.mkdir()
    e.mkdir()
    assert p.debugProcess(str(d))
    o.debug_process(str(e))
    images = ["norm.bmp", "sdev.bmp", "cnr.bmp", "relevant.bmp", "graded.bmp", "noise_hist.bmp", "grad_hist.bmp"] + \
             ["red_bandpass_%d.bmp" % i for i in range(levels)] + ["red_lowpass_%d.bmp" % i for i in range(levels)] + \
             ["exp_bandpass_%d.bmp" % i for i in range(levels)] + ["exp_lowpass_%d.bmp" % i for i in range(levels)]
    assert sorted(f.name for f in e.iterdir()) == sorted(images)
    names = {f.name for f in d.iterdir()}
    for want in images + ["noise_hist.csv", "grad_hist.csv", "grad_curve.csv"]:
        assert want in names, want
    for name in images:
        assert (d / name).read_bytes() == (e / name).read_bytes(), name
    # the two RGBA plots (RENDER_HISTS): 512 x 128, four components, V4 header (stbi_write_bmp comp = 4)
    raw = (d / "noise_hist.bmp").read_bytes()
    assert len(raw) == 14 + 108 + 512 * 128 * 4 and int.from_bytes(raw[18:22], "little") == 512 and int.from_bytes(raw[22:26], "little") == 128
    assert np.array_equal(p.render_noise_hist(), o.render_noise_hist()) and np.array_equal(p.render_grad_hist(), o.render_grad_hist())
    # graded.bmp is the un-cropped 8-bit image; the level-3 dumps have side N / 8
    raw = (d / "graded.bmp").read_bytes()
    assert int.from_bytes(raw[18:22], "little") == n
    assert int.from_bytes((d / "cnr.bmp").read_bytes()[18:22], "little") == n // 8
    # exp_bandpass_i is the contrast-curve output of level L-1-i BEFORE noise reduction (expandBandpassImageStates, :1100)
    _same(p.image(mp.IMG_CONTRAST_BAND, 0), o.image(ob.IMG_CONTRAST_BAND, 0), "contrast band 0")
    assert not np.array_equal(p.image(mp.IMG_CONTRAST_BAND, 0), p.image(mp.IMG_EXP_BANDPASS, 0))
    # the CSVs hold the histograms / curve the getters return
    rows = (d / "grad_hist.csv").read_text().splitlines()
    assert rows[0] == "bin,weight" and [int(r.split(",")[1]) for r in rows[1:]] == [int(v) for v in o.grad_hist()]
    rows = (d / "noise_hist.csv").read_text().splitlines()[1:]
    for lvl in range(4):
        assert [int(r.split(",")[1 + lvl]) for r in rows] == [int(v) for v in o.noise_hist(lvl)]
    p.cleanup()


@pytest.mark.parametrize("n,levels,batch,flags", [(1032, 6, 1, 0), (1032, 6, 3, 0), (520, 4, 2, 0), (2048, 6, 1, 0), (1024, 5, 1, "clahe"), (2056, 7, 1, "linear"), (264, 4, 1, 0)])
def test_sdev_inside_the_expand_launches(ob, n, levels, batch, flags, monkeypatch):
    """MUSICA_SDEV_IN_EXPAND=1 (the default of byte-bound workloads: steps in flight from 2 x 2048^2 texels, a lone context from 8 x 2048^2):
    the expand launches of levels 0 .. 2 compute the 5 x 5 RMS of their band image in registers (k_expand_fast<.., SD>, a window of six band
    rows) and the sdev launches of those levels store nothing. Everything against the oracle, twice in a row, in both launch modes; the sdev
    images themselves come from the getters' on-demand launch. Sides with 1, 2 and 3 strips, segments that end inside the image, a level
    of 33 rows (264 / 8), batches, CLAHE (the instantiation that also counts the CLAHE histogram) and a one-stream context.
    Then the stage entry points on the same context: they use the stored images (an injected sdev image must reach the expand stage)."""
    f = {0: 0, "clahe": mp.FLAG_CLAHE, "linear": mp.FLAG_LINEAR}[flags]
    px = np.stack([phantom(n, 300 + k) for k in range(batch)])
    of = ob.FLAG_CLAHE if flags == "clahe" else 0
    want = [ob.Oracle(n, levels, ob.ORDER_FAST, of).execute(px[k]) for k in range(batch)]
    monkeypatch.setenv("MUSICA_SDEV_IN_EXPAND", "1")
    for graph in ("1", "0"):
        monkeypatch.setenv("MUSICA_GRAPH", graph)
        p = _proc(n, levels, batch=batch, flags=f)
        assert p.fuses_sdev()
        for rep in range(2):
            assert p.execute(px)
        for k in range(batch):
            _compare_all(p, want[k], ob, idx=k, tag="sdev in expand, graph %s, image %d: " % (graph, k))
        p.cleanup()
    monkeypatch.setenv("MUSICA_SDEV_IN_EXPAND", "0")
    p = _proc(n, levels, batch=batch, flags=f)
    assert not p.fuses_sdev()
    p.cleanup()
    # stage entry points after a whole-step execute: sdev level 1 replaced by a constant image -> the expand stage must see it
    monkeypatch.setenv("MUSICA_SDEV_IN_EXPAND", "1")
    p = _proc(n, levels, batch=batch, flags=f)
    assert p.execute(px)
    s1 = p.image(mp.IMG_SDEV, 1, 0)
    _same(s1, want[0].image(ob.IMG_SDEV, 1), "sdev[1] on demand")
    fake = np.full_like(s1, 0.004)
    p.set_image(mp.IMG_SDEV, 1, fake, 0)
    _same(p.image(mp.IMG_SDEV, 0, 0), want[0].image(ob.IMG_SDEV, 0), "sdev[0] kept")
    p.run_stage(mp.STAGE_EXPAND)
    o2 = ob.Oracle(n, levels, ob.ORDER_FAST, of).execute(px[0])
    o2.set_image(ob.IMG_SDEV, 1, fake)
    o2.run_stage(ob.STAGE_EXPAND)
    _same(p.image(mp.IMG_EXPAND, 0, 0), o2.image(ob.IMG_EXPAND, 0), "expand[0] from the injected sdev[1]")
    assert p.execute(px)   # and the next whole step computes its own again
    _same(p.image(mp.IMG_EXPAND, 0, 0), want[0].image(ob.IMG_EXPAND, 0), "expand[0] after the next step")
    p.cleanup()


@pytest.mark.parametrize("n", [64, 533, 534, 535])
def test_save_out_image_rows_built_on_the_device(ob, n, tmp_path, monkeypatch):
    """saveOutImage's file image — header by the host, 24-bpp bottom-up padded rows by k_out_bmp24 straight into page-locked memory, one
    write — against the former path (1 byte per pixel read back, rows expanded on the host: MUSICA_SAVE_ON_DEVICE=0) and the oracle's
    stbi_write_bmp restatement, for widths N - 20 with row padding 0, 1, 2 and 3; twice into the same context (the file image is reused)."""
    levels = 4
    px = phantom(n, 77)
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px)
    want = tmp_path / "oracle.bmp"
    o.save_out_image(str(want))
    assert (3 * (n - 20)) % 4 == {64: 0, 533: 3, 534: 2, 535: 1}[n]
    for on_device in ("1", "0"):
        monkeypatch.setenv("MUSICA_SAVE_ON_DEVICE", on_device)
        p = _proc(n, levels)
        for rep in range(2):
            assert p.execute(px)
            got = tmp_path / ("hip_%s_%d.bmp" % (on_device, rep))
            assert p.saveOutImage(str(got))
            assert got.read_bytes() == want.read_bytes(), (on_device, rep)
        p.cleanup()


def test_cli_drop_in(ob, tmp_path):
    import subprocess
    from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import write_raw
    n = 512
    px = phantom(n, 55)
    raw, out = tmp_path / "image.raw", tmp_path / "out.bmp"
    write_raw(str(raw), px)
    r = subprocess.run([mp.CLI_PATH, str(raw), str(out), "--size", str(n)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "raw file" in r.stdout and "tot:" in r.stdout
    o = ob.Oracle(n, 0, ob.ORDER_FAST).execute(px)
    ref = tmp_path / "oracle.bmp"
    o.save_out_image(str(ref))
    assert out.read_bytes() == ref.read_bytes()
    # --debug-dir: what a debug build of the reference's harness dumps after execute (main.cpp:81-84 -> debugProcess)
    dd, de = tmp_path / "cli_dump", tmp_path / "cli_dump_oracle"
    dd.mkdir()
    de.mkdir()
    r = subprocess.run([mp.CLI_PATH, str(raw), str(out), "--size", str(n), "--debug-dir", str(dd)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    o.debug_process(str(de))
    for f in sorted(de.iterdir()):
        assert (dd / f.name).read_bytes() == f.read_bytes(), f.name
    assert (dd / "noise_hist.bmp").exists() and (dd / "grad_hist.bmp").exists()
    # wrong file size -> MAIN ERROR, exit code 1 (main.cpp:57-60)
    r = subprocess.run([mp.CLI_PATH, str(raw), str(out), "--size", "256"], capture_output=True, text=True)
    assert r.returncode == 1 and "MAIN ERROR: the image data don't match the actual image size" in r.stderr


@pytest.mark.parametrize("sd", ["0", "1", "1+pairs"])
def test_steps_in_flight_on_three_contexts_are_the_lone_contexts_steps(ob, sd, monkeypatch):
    """musica_pipeline_*: steps alternate over three linear contexts without waiting for one another (what bench.py times);
    with sdev stored (sd = 0) and computed inside the expand launches (sd = 1: what the timed contexts of bench.py do).
    Each context holds DIFFERENT images, so a result that leaked between contexts would show; every image of every
    context is bit-identical to the oracle after 7 overlapping steps, and the strided image ids of a rank's stats
    rows (rank + index * world) come from the stats kernel itself."""
    from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import batch as mb
    n, levels, b, depth = 520, 5, 2, 3
    monkeypatch.setenv("MUSICA_SDEV_IN_EXPAND", sd[0])
    monkeypatch.setenv("MUSICA_PAIR_RB_SDEV", "1" if sd.endswith("pairs") else "0")   # the pairs of the one-stream script (k_rb_sdev): what byte-bound pipeline contexts run
    px = [np.stack([phantom(n, 1000 + 10 * c + k) for k in range(b)]) for c in range(depth)]
    pipe = mp.MusicaPipeline(n, levels=levels, batch=b, depth=depth)
    assert pipe.context(0).fuses_sdev() == (sd[0] == "1")
    pipe.upload(px[0])
    stale = pipe.context(0)
    pipe.prime()
    assert stale._h is None                    # wrappers borrowed before prime() are void afterwards (their context may be gone)
    with pytest.raises(ValueError):
        pipe.upload(px[0][:1])                 # one image where the batch is two
    ctx = [pipe.context(c) for c in range(depth)]
    for c in range(depth):
        ctx[c].upload(px[c])
    for _ in range(7):
        pipe.step()
    assert pipe.last()._h == ctx[0]._h
    d_rows = pipe.last().device_alloc(b * mb.STATS_WORDS * 4)
    pipe.last().stats_device(d_rows, image_id_base=3, image_id_stride=8)
    pipe.sync()
    rows = np.zeros((b, mb.STATS_WORDS), dtype=np.int32)
    pipe.last().d2h(rows, d_rows)
    assert [int(r[0]) for r in rows] == [3, 11]
    pipe.last().device_free(d_rows)
    for c in range(depth):
        for k in range(b):
            o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px[c][k])
            _compare_all(ctx[c], o, ob, idx=k, tag="context %d image %d: " % (c, k))
    pipe.cleanup()


@pytest.mark.parametrize("nbuf", [3, 6])
def test_rotating_caller_owned_device_buffers_without_syncing(ob, nbuf, monkeypatch):
    """A caller that rotates its own device buffers through ONE context (musica_execute_device never waits): every distinct
    input pointer gets a captured graph, a context keeps four of them, and with more the least recently used executable graph
    is only destroyed after the stream has drained — 3 buffers replay, 6 recapture on every step; either way every step's
    result is the oracle's."""
    monkeypatch.setenv("MUSICA_GRAPH", "1")    # a context this small would launch eagerly by default: the graph slots are what is tested
    n, levels = 520, 5
    imgs = [phantom(n, 4000 + k) for k in range(nbuf)]
    want = [ob.Oracle(n, levels, ob.ORDER_FAST).execute(im).image(ob.IMG_GRADED) for im in imgs]
    p = _proc(n, levels)
    bufs = []
    for im in imgs:
        d = p.device_alloc(im.nbytes)
        p.h2d(d, im)
        bufs.append(d)
    for rnd in range(3):
        for k in range(nbuf):                  # back to back, no sync in between
            assert p.execute_device(bufs[k]), mp.last_error()
        last = (rnd * nbuf + nbuf - 1) % nbuf
        _same(p.image(mp.IMG_GRADED), want[last], "graded after round %d (%d buffers)" % (rnd, nbuf))
    # and every buffer once more, each checked
    for k in range(nbuf):
        assert p.execute_device(bufs[k]), mp.last_error()
        _same(p.image(mp.IMG_GRADED), want[k], "graded of buffer %d" % k)
    for d in bufs:
        p.device_free(d)
    p.cleanup()


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_clahe_context_with_and_without_the_raw_pixel_relevant_image(ob, fuse, monkeypatch):
    """A CLAHE context normally takes `normalized <= 0.9` of its relevant image from the raw pixels (no stored normalized image,
    gradation histogram inside the level-0 expand launch); MUSICA_CLAHE_FUSE=0 is the stored-image form. Both equal the oracle
    bit for bit, including the CLAHE histograms, curves and the blended image."""
    monkeypatch.setenv("MUSICA_CLAHE_FUSE", fuse)
    n, levels = 1024, 6
    px = phantom(n, 77)
    o = ob.Oracle(n, levels, ob.ORDER_FAST, ob.FLAG_CLAHE).execute(px)
    p = _proc(n, levels, flags=mp.FLAG_CLAHE)
    assert p.fuses_gradhist() == (fuse == "1")
    for rep in range(2):
        assert p.execute(px), mp.last_error()
    _compare_all(p, o, ob, tag="clahe fuse=%s: " % fuse)
    assert np.array_equal(p.clahe_hist(), o.clahe_hist())
    a, b = p.clahe_curves(), o.clahe_curves()
    assert ((a == b) | (np.isnan(a) & np.isnan(b))).all()
    _same(p.image(mp.IMG_CLAHE_GRADED), o.image(ob.IMG_CLAHE_GRADED), "clahe graded")
    p.cleanup()


@pytest.mark.parametrize("n,levels,batch,tail,dag", [(3072, 0, 1, "1", None), (3072, 0, 1, "0", None), (1000, 0, 2, "1", None), (333, 0, 1, "1", None),
                                                     (2048, 0, 2, "1", "1"), (2048, 0, 1, "1", "2"), (4096, 0, 1, "1", None)])
def test_tiny_tail_of_a_full_depth_pyramid_in_one_launch(ob, n, levels, batch, tail, dag, monkeypatch):
    """levels = 0 is the reference's own call (full depth: 12 levels at 3072^2). The levels of side <= 32 — reduce, band and expand of
    each, 14 launches at 3072^2 — run as ONE launch of one workgroup per image (k_tiny_tail) in every dispatch form; MUSICA_TINY_TAIL=0
    is one launch per level and stage. Every image of every level equals the oracle either way."""
    monkeypatch.setenv("MUSICA_TINY_TAIL", tail)
    if dag is not None:
        monkeypatch.setenv("MUSICA_STREAMS", dag)
    px = np.stack([phantom(n, 40 + k) for k in range(batch)])
    p = _proc(n, levels, batch=batch)
    p.upload(px)
    for rep in range(2):
        assert p.execute_device(), mp.last_error()
    p.sync()
    for k in range(batch):
        o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px[k])
        assert p.pyramidLevels == o.levels
        _compare_all(p, o, ob, idx=k, tag="tiny tail=%s %d image %d: " % (tail, n, k))
    p.cleanup()


@pytest.mark.parametrize("n,levels,seed,batch,in_expand,one_apply", [(2048, 6, 21, 2, "1", "1"), (2056, 6, 22, 1, "1", "1"), (3072, 7, 23, 1, "1", "1"),
                                                                     (3072, 7, 23, 1, "0", "1"), (2048, 6, 21, 2, "1", "0"), (1024, 5, 24, 3, "1", "1")])
def test_clahe_histogram_in_the_level0_expand_launch_and_both_curves_in_one_apply(ob, n, levels, seed, batch, in_expand, one_apply, monkeypatch):
    """Sides whose CLAHE tile (N / 4) is at least a 512-column strip: the level-0 expand launch counts clahe_histogram.comp while the
    texels are in registers (k_expand_fast<.., CH>) and k_clahe_hist does not run; 2056 and 3072 put tile borders inside strips
    and inside workgroups (1024: tiles of 256 columns, the separate launch). MUSICA_CLAHE_IN_EXPAND=0 is the separate launch everywhere.
    The tone curve and the CLAHE curves are applied in one pass over the reconstruction (k_grad_clahe_apply4; MUSICA_CLAHE_ONE_APPLY=0:
    k_grad_apply and k_clahe_apply4). Histograms, curves and both graded images equal the oracle."""
    monkeypatch.setenv("MUSICA_CLAHE_IN_EXPAND", in_expand)
    monkeypatch.setenv("MUSICA_CLAHE_ONE_APPLY", one_apply)
    px = np.stack([phantom(n, seed + k) for k in range(batch)])
    p = _proc(n, levels, batch=batch, flags=mp.FLAG_CLAHE)
    assert p.fuses_gradhist()
    p.upload(px)
    for rep in range(2):
        assert p.execute_device(), mp.last_error()
    p.sync()
    for k in range(batch):
        o = ob.Oracle(n, levels, ob.ORDER_FAST, ob.FLAG_CLAHE).execute(px[k])
        _compare_all(p, o, ob, idx=k, tag="clahe in expand=%s image %d: " % (in_expand, k))
        assert np.array_equal(p.clahe_hist(k), o.clahe_hist())
        assert int(o.clahe_hist().sum()) > 0
        a, b = p.clahe_curves(k), o.clahe_curves()
        assert ((a == b) | (np.isnan(a) & np.isnan(b))).all()
        _same(p.image(mp.IMG_CLAHE_GRADED, 0, k), o.image(ob.IMG_CLAHE_GRADED), "clahe graded %d" % k)
    p.cleanup()


@pytest.mark.parametrize("n,levels,seed,batch", [(1024, 6, 5, 1), (520, 5, 12, 3), (2048, 6, 100, 1)])
def test_histogram_plots_equal_the_oracles(ob, n, levels, seed, batch):
    """The RENDER_HISTS plots (noise_hist_render.comp on the cnr level, gradation_curve_debug_render.comp): the HIP kernels'
    512 x 128 rgba8 images against the oracle's restatement, texel for texel, for every image of a batch; and they are
    not trivially empty (bars, the argmax colour, the three window columns and the curve are there)."""
    px = np.stack([phantom(n, seed + k) for k in range(batch)])
    p = _proc(n, levels, batch=batch)
    assert p.execute(px if batch > 1 else px[0]), mp.last_error()
    for k in range(batch):
        o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(px[k])
        a, b = p.render_noise_hist(k), o.render_noise_hist()
        assert np.array_equal(a, b), "noise plot of image %d" % k
        c, d = p.render_grad_hist(k), o.render_grad_hist()
        assert np.array_equal(c, d), "gradation plot of image %d" % k
        assert (a[..., 3] == 255).all() and (c[..., 3] == 255).all()
        assert (a[..., :3] == [0, 255, 0]).all(axis=-1).any() or o.noise_hist_max(3)[1] >= 512      # the argmax bar is green
        assert (c[..., :3] == [0, 0, 255]).all(axis=-1).sum() >= 256                                # the curve, one texel per column
        assert (c[..., :3] == [0, 255, 0]).all(axis=-1).sum() >= 127                                # the ta column
    p.cleanup()


def test_native_pipeline_of_the_c_abi(ob):
    """musica_pipeline_* (the C ABI's steps-in-flight object): create with depth 3 -> four one-stream contexts, prime() times the
    four windows of hardware queues and keeps three contexts, 8 overlapping steps; the context of the last step and the two
    beside it equal the oracle bit for bit, the step counter walks the kept contexts in order, and depth 1 is a plain context."""
    n, levels, b = 520, 5, 2
    px = np.stack([phantom(n, 2000 + k) for k in range(b)])
    want = [ob.Oracle(n, levels, ob.ORDER_FAST).execute(px[k]) for k in range(b)]
    pl = mp.MusicaPipeline(n, levels=levels, batch=b, depth=3)
    pl.upload(px)
    pl.prime(6)
    cal = pl.calibration()
    assert sorted(cal) == [0, 1, 2, 3] and all(0.0 < v < 50.0 for v in cal.values())
    handles = [pl.context(k)._h for k in range(3)]
    assert len(set(handles)) == 3
    with pytest.raises(IndexError):
        pl.context(3)
    for s in range(8):
        pl.step()
        assert pl.last()._h == handles[s % 3]
    pl.sync()
    for k in range(3):
        c = pl.context(k)
        assert c.batch == b and c.pyramidLevels == levels
        for i in range(b):
            _compare_all(c, want[i], ob, idx=i, tag="pipeline context %d image %d: " % (k, i))
    # a caller-owned device buffer as the step's input
    other = np.stack([phantom(n, 2100 + k) for k in range(b)])
    c0 = pl.context(0)
    d = c0.device_alloc(other.nbytes)
    c0.h2d(d, other)
    pl.step(d)                       # step 8 -> context 8 % 3 = 2
    pl.sync()
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(other[1])
    _same(pl.last().image(mp.IMG_GRADED, 0, 1), o.image(ob.IMG_GRADED), "graded (caller-owned input through the pipeline)")
    c0.device_free(d)
    pl.cleanup()
    one = mp.MusicaPipeline(n, levels=levels, batch=1, depth=1)
    one.upload(px[:1])
    one.prime()
    assert one.calibration() == {}
    one.step()
    one.sync()
    _same(one.last().image(mp.IMG_GRADED), want[0].image(ob.IMG_GRADED), "graded (depth 1)")
    one.cleanup()


@pytest.mark.parametrize("n,batch", [(16, 1), (333, 3), (1000, 2), (2048, 2)])
def test_minmax_two_stage_edge_cases(ob, n, batch):
    """k_minmax_u16 (two stages: per-block slots + the last ticket holder folds them; the step's clears in the same launch) on
    sizes with one block, with images that start at odd byte offsets inside the batch (odd N: the scalar path), with several
    trips per block, on extreme pixel values, and twice in a row (the ticket must be back at 0, the histograms cleared again)."""
    rng = np.random.default_rng(n)
    px = rng.integers(1000, 60000, size=(batch, n, n), dtype=np.uint16)
    px[0, n // 2, n // 3] = 65535
    px[batch - 1, n - 1, n - 1] = 7
    if batch > 1:
        px[1, 0, 0] = 0
    p = _proc(n, 4 if n < 64 else 5, batch=batch)
    for rep in range(2):
        assert p.execute(px if batch > 1 else px[0]), mp.last_error()
        for k in range(batch):
            o = ob.Oracle(n, p.pyramidLevels, ob.ORDER_FAST).execute(px[k])
            assert p.minmax(k) == o.minmax(), "image %d rep %d" % (k, rep)
            assert np.array_equal(p.noise_hist(0, k), o.noise_hist(0)) and np.array_equal(p.grad_hist(k), o.grad_hist())
    p.cleanup()


@pytest.mark.parametrize("over", [{"linear_low_contrast": 1}, {"linear_high_contrast": 1}, {"linear_low_contrast": 1, "linear_high_contrast": 1},
                                  {"nr_low_cnr": 2.0, "nr_high_cnr": 10.0, "nr_min_low_factor": 0.5, "nr_max_high_factor": 1.5},
                                  {"high_contrast_max_reduction": 0.5, "low_contrast_max_enhancement": 2.0}],
                         ids=lambda o: ",".join("%s=%s" % kv for kv in o.items()))
def test_runtime_tunables_against_the_oracle(ob, over):
    """musica_create_ex (ABI version 3): the reference's compile-time configuration — the LINEAR_* forms of src/vk_processing.cpp:262-293
    and the constants of include/vk_processing.h:39-49 — as runtime values; every variant bit-identical to the oracle created with the
    same musica_tunables, and different from the default configuration's result."""
    n, levels = 1024, 7
    px = phantom(n, 321)
    t = mp.default_tunables(**over)
    o = ob.Oracle(n, levels, ob.ORDER_FAST, tunables=t).execute(px)
    p = mp.MusicaProcessing()
    assert p.init(n, levels=levels, tunables=t), mp.last_error()
    got = p.tunables()
    assert all(getattr(got, k) == getattr(t, k) for k, _ in mp.Tunables._fields_)
    assert p.execute(px), mp.last_error()
    _compare_all(p, o, ob, tag=str(over) + ": ")
    base = _proc(n, levels)
    assert base.execute(px)
    assert not np.array_equal(base.graded(), p.graded())
    # a batch through the pipeline entry point with the same tunables
    p.cleanup()
    base.cleanup()


def test_create_ex_refuses_unusable_tunables():
    p = mp.MusicaProcessing()
    assert not p.init(512, levels=4, tunables=mp.default_tunables(nr_low_cnr=9.0))          # == nr_high_cnr: the slope divides by zero
    assert "nr_high_cnr == nr_low_cnr" in mp.last_error()
    assert not p.init(512, levels=4, tunables=mp.default_tunables(low_contrast_max_enhancement=float("nan")))


@pytest.mark.parametrize("one_launch", ["1", "0"])
def test_exact_zeros_recount_with_two_column_blocks(ob, one_launch, monkeypatch):
    """The literal recount of a collimated image where k_grad_recount_curve's grid is two column blocks wide (sides above 4096: 16
    wavefronts of 256 columns side by side cover 4096): columns beyond 4096 hold zeros AND data, three executes in a row (the last-ticket
    hand-off re-arms), both launch forms. 8192^2 is a BASELINE configuration; 4104^2 has the same grid shape at a fifth of the oracle's time."""
    monkeypatch.setenv("MUSICA_GRAD_ONE_LAUNCH", one_launch)
    n, levels = 4104, 4
    a = phantom(n, 71)
    a[:150, :] = 0
    a[:, :170] = 0
    a[-260:, :] = 0
    a[:, -190:] = 0          # zeros inside the second column block (columns 4096 .. 4103 are all zero; 3914 .. 4095 too)
    a[1000:1100, 4000:4104] = 0
    o = ob.Oracle(n, levels, ob.ORDER_FAST).execute(a)
    assert (o.image(ob.IMG_EXPAND, 0) == 0.0).sum() > 1000
    p = _proc(n, levels)
    assert p.fuses_gradhist()
    for rep in range(3):
        assert p.execute(a), mp.last_error()
        _compare_all(p, o, ob, tag="two column blocks, execute %d: " % rep)
    p.cleanup()
