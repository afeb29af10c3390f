"""ctypes binding of the CPU oracle (oracle/build/libmusica_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg. The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "build", "libmusica_oracle.so")
REF_BMP_PATH = os.path.join(_HERE, "_ref", "libref_bmp.so")

ORDER_REFERENCE = 0
ORDER_FAST = 1
FLAG_CLAHE = 1

MAX_POINTS = 256
NOISE_BINS = 2048
GRAD_BINS = 1024

# image kinds (musica.h musica_image_kind + oracle-only kinds)
IMG_NORMALIZED, IMG_DOWNSAMPLED, IMG_BANDPASS, IMG_SDEV, IMG_CNR, IMG_EXPAND = 0, 1, 2, 3, 4, 5
IMG_GRADED, IMG_RELEVANT, IMG_LOWPASS, IMG_EXP_BANDPASS, IMG_SQRT, IMG_CLAHE_GRADED = 6, 7, 8, 9, 10, 11
IMG_SMOOTH, IMG_UPSAMPLED, IMG_EXP_UPSAMPLED, IMG_EXP_LOWPASS, IMG_CONTRAST_BAND, IMG_NR_BAND = 100, 101, 102, 103, 104, 105

STAGE_NORM, STAGE_REDUCE, STAGE_ANALYSIS, STAGE_EXPAND, STAGE_GRADATION = 0, 1, 2, 3, 4


class HistMaxPoint(C.Structure):
    _fields_ = [("maxValue", C.c_uint32), ("maxBin", C.c_uint32)]


class ContrastParams(C.Structure):
    _fields_ = [("lowContrastFactor", C.c_float), ("highContrastFactor", C.c_float)]


class Point(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class ContrastCurve(C.Structure):
    _fields_ = [("points", Point * MAX_POINTS), ("pointsCount", C.c_uint32)]

    def as_array(self):
        n = self.pointsCount
        return np.array([(self.points[i].x, self.points[i].y) for i in range(n)], dtype=np.float32)


class NrParams(C.Structure):
    _fields_ = [("lowCnr", C.c_float), ("lowFactor", C.c_float), ("highCnr", C.c_float), ("highFactor", C.c_float)]


class GradCurve(C.Structure):
    _fields_ = [("points", Point * MAX_POINTS), ("pointsCount", C.c_uint32),
                ("t0", C.c_float), ("ta", C.c_float), ("t1", C.c_float)]

    def as_array(self):
        n = self.pointsCount
        return np.array([(self.points[i].x, self.points[i].y) for i in range(n)], dtype=np.float32)


class Stats(C.Structure):
    _fields_ = [("image_id", C.c_uint32), ("min_sqrt", C.c_float), ("max_sqrt", C.c_float),
                ("noise_max_bin", C.c_uint32 * 4), ("noise_max_value", C.c_uint32 * 4),
                ("grad_max_bin", C.c_uint32), ("grad_max_value", C.c_uint32),
                ("mean_cnr", C.c_float), ("t0", C.c_float), ("ta", C.c_float), ("t1", C.c_float)]


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "musica_oracle.c")):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return LIB_PATH


_lib = None


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    fp, u32p, u16p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint16), C.POINTER(C.c_uint8)
    vp = C.c_void_p
    L.musica_oracle_create.restype = vp
    L.musica_oracle_create.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]
    L.musica_oracle_create_ex.restype = vp
    L.musica_oracle_create_ex.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, vp]
    L.musica_oracle_tunables_default.restype = None
    L.musica_oracle_tunables_default.argtypes = [vp]
    L.musica_oracle_host_contrast_params_ex.restype = ContrastParams
    L.musica_oracle_host_contrast_params_ex.argtypes = [C.c_uint32, C.c_uint32, vp]
    L.musica_oracle_host_nr_params_ex.restype = NrParams
    L.musica_oracle_host_nr_params_ex.argtypes = [C.c_uint32, vp]
    L.musica_oracle_destroy.argtypes = [vp]
    L.musica_oracle_set_threads.argtypes = [C.c_int]
    L.musica_oracle_get_threads.restype = C.c_int
    L.musica_oracle_levels.restype = C.c_uint32
    L.musica_oracle_levels.argtypes = [vp]
    L.musica_oracle_level_size.restype = C.c_uint32
    L.musica_oracle_level_size.argtypes = [vp, C.c_uint32]
    L.musica_oracle_execute.restype = C.c_int
    L.musica_oracle_execute.argtypes = [vp, u16p]
    L.musica_oracle_run_stage.restype = C.c_int
    L.musica_oracle_run_stage.argtypes = [vp, C.c_int]
    L.musica_oracle_image.restype = fp
    L.musica_oracle_image.argtypes = [vp, C.c_int, C.c_uint32, u32p]
    L.musica_oracle_set_image.restype = C.c_int
    L.musica_oracle_set_image.argtypes = [vp, C.c_int, C.c_uint32, fp]
    L.musica_oracle_noise_hist.restype = u32p
    L.musica_oracle_noise_hist.argtypes = [vp, C.c_uint32]
    L.musica_oracle_grad_hist.restype = u32p
    L.musica_oracle_grad_hist.argtypes = [vp]
    L.musica_oracle_noise_hist_max.restype = HistMaxPoint
    L.musica_oracle_noise_hist_max.argtypes = [vp, C.c_uint32]
    L.musica_oracle_grad_hist_max.restype = HistMaxPoint
    L.musica_oracle_grad_hist_max.argtypes = [vp]
    L.musica_oracle_contrast_curve.restype = C.POINTER(ContrastCurve)
    L.musica_oracle_contrast_curve.argtypes = [vp, C.c_uint32]
    L.musica_oracle_grad_curve.restype = C.POINTER(GradCurve)
    L.musica_oracle_grad_curve.argtypes = [vp]
    L.musica_oracle_contrast_params.restype = ContrastParams
    L.musica_oracle_contrast_params.argtypes = [vp, C.c_uint32]
    L.musica_oracle_nr_params.restype = NrParams
    L.musica_oracle_nr_params.argtypes = [vp, C.c_uint32]
    L.musica_oracle_minmax.argtypes = [vp, fp, fp]
    L.musica_oracle_stats.argtypes = [vp, C.POINTER(Stats)]
    L.musica_oracle_clahe_hist.restype = u32p
    L.musica_oracle_clahe_hist.argtypes = [vp]
    L.musica_oracle_clahe_curves.restype = C.POINTER(Point)
    L.musica_oracle_clahe_curves.argtypes = [vp]
    L.musica_oracle_out_pixels.restype = C.c_int
    L.musica_oracle_out_pixels.argtypes = [vp, u8p]
    L.musica_oracle_save_out_image.restype = C.c_int
    L.musica_oracle_save_out_image.argtypes = [vp, C.c_char_p]
    L.musica_oracle_debug_process.restype = C.c_int
    L.musica_oracle_debug_process.argtypes = [vp, C.c_char_p]
    L.musica_oracle_write_bmp_gray.restype = C.c_int
    L.musica_oracle_write_bmp_gray.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, u8p]
    L.musica_oracle_write_bmp_rgba.restype = C.c_int
    L.musica_oracle_write_bmp_rgba.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, u8p]
    L.musica_oracle_render_noise_hist.restype = None
    L.musica_oracle_render_noise_hist.argtypes = [vp, u8p]
    L.musica_oracle_render_grad_hist.restype = None
    L.musica_oracle_render_grad_hist.argtypes = [vp, u8p]
    L.musica_oracle_read_raw.restype = C.c_int
    L.musica_oracle_read_raw.argtypes = [C.c_char_p, C.c_uint32, u16p]
    # single-shader entry points
    L.musica_oracle_k_sqrt.argtypes = [u16p, C.c_uint32, fp]
    L.musica_oracle_k_max_reduce.argtypes = [fp, C.c_uint32, fp]
    L.musica_oracle_k_min_reduce.argtypes = [fp, C.c_uint32, fp]
    L.musica_oracle_k_normalize.argtypes = [fp, C.c_uint32, C.c_float, C.c_float, fp]
    L.musica_oracle_k_smooth.argtypes = [fp, C.c_uint32, fp, C.c_int]
    L.musica_oracle_k_downsample.argtypes = [fp, C.c_uint32, fp]
    L.musica_oracle_k_upsample.argtypes = [fp, C.c_uint32, fp, C.c_uint32]
    L.musica_oracle_k_smooth_upsampled.argtypes = [fp, C.c_uint32, fp, C.c_int]
    L.musica_oracle_k_difference.argtypes = [fp, fp, C.c_uint32, fp]
    L.musica_oracle_k_addition.argtypes = [fp, fp, C.c_uint32, fp]
    L.musica_oracle_k_sdev.argtypes = [fp, C.c_uint32, fp, C.c_int]
    L.musica_oracle_k_noise_hist.argtypes = [fp, C.c_uint32, C.c_uint32, u32p]
    L.musica_oracle_k_histogram_max.argtypes = [u32p, C.c_uint32, C.POINTER(HistMaxPoint)]
    L.musica_oracle_k_contrast_curve_generate.argtypes = [HistMaxPoint, ContrastParams, C.POINTER(ContrastCurve)]
    L.musica_oracle_k_contrast_curve_apply.argtypes = [fp, fp, C.c_uint32, C.POINTER(ContrastCurve), fp]
    L.musica_oracle_k_cnr.argtypes = [fp, C.c_uint32, HistMaxPoint, fp]
    L.musica_oracle_k_noise_reduction.argtypes = [fp, C.c_uint32, fp, C.c_uint32, NrParams, fp]
    L.musica_oracle_k_relevant.argtypes = [fp, C.c_uint32, fp, C.c_uint32, fp]
    L.musica_oracle_k_gradation_histogram.argtypes = [fp, fp, C.c_uint32, C.c_uint32, u32p]
    L.musica_oracle_k_gradation_curve_generate.argtypes = [u32p, C.POINTER(GradCurve)]
    L.musica_oracle_k_apply_gradation_curve.argtypes = [fp, C.c_uint32, C.POINTER(GradCurve), fp]
    L.musica_oracle_get_y.restype = C.c_float
    L.musica_oracle_get_y.argtypes = [C.POINTER(Point), C.c_uint32, C.c_float]
    L.musica_oracle_k_clahe_histogram.argtypes = [fp, fp, C.c_uint32, u32p]
    L.musica_oracle_k_clahe_grad_curve.argtypes = [u32p, C.POINTER(Point)]
    L.musica_oracle_k_clahe_grad_curve_apply.argtypes = [fp, C.c_uint32, C.POINTER(Point), fp]
    L.musica_oracle_host_contrast_params.restype = ContrastParams
    L.musica_oracle_host_contrast_params.argtypes = [C.c_uint32, C.c_uint32]
    L.musica_oracle_host_nr_params.restype = NrParams
    L.musica_oracle_host_nr_params.argtypes = [C.c_uint32]
    _lib = L
    return L


def set_threads(n):
    lib().musica_oracle_set_threads(int(n))


class Oracle:
    """One image through VulkanProcessing::execute, restated on the CPU."""

    def __init__(self, image_size, levels=0, order=ORDER_FAST, flags=0, tunables=None):
        """tunables: a ctypes structure with musica_tunables' layout (include/musica.h; e.g. processing.Tunables or binding.Tunables); None = the reference's constants."""
        self.L = lib()
        self.h = self.L.musica_oracle_create_ex(image_size, levels, order, flags, C.cast(C.byref(tunables), C.c_void_p) if tunables is not None else None)
        if not self.h:
            raise ValueError("musica_oracle_create(%d, %d) failed" % (image_size, levels))
        self.N = image_size
        self.levels = self.L.musica_oracle_levels(self.h)
        self.order = order

    def close(self):
        if self.h:
            self.L.musica_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level_size(self, level):
        return self.L.musica_oracle_level_size(self.h, level)

    def execute(self, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.uint16)
        assert px.size == self.N * self.N
        ok = self.L.musica_oracle_execute(self.h, px.ctypes.data_as(C.POINTER(C.c_uint16)))
        assert ok == 1
        return self

    def run_stage(self, stage):
        assert self.L.musica_oracle_run_stage(self.h, stage) == 1

    def image(self, kind, level=0):
        side = C.c_uint32(0)
        p = self.L.musica_oracle_image(self.h, kind, level, C.byref(side))
        if not p:
            raise KeyError("no oracle image kind=%d level=%d" % (kind, level))
        s = side.value
        return np.ctypeslib.as_array(p, shape=(s, s)).copy()

    def set_image(self, kind, level, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        assert self.L.musica_oracle_set_image(self.h, kind, level, _f32p(a)) == 1

    def noise_hist(self, level):
        return np.ctypeslib.as_array(self.L.musica_oracle_noise_hist(self.h, level), shape=(NOISE_BINS,)).copy()

    def grad_hist(self):
        return np.ctypeslib.as_array(self.L.musica_oracle_grad_hist(self.h), shape=(GRAD_BINS,)).copy()

    def noise_hist_max(self, level):
        p = self.L.musica_oracle_noise_hist_max(self.h, level)
        return (p.maxValue, p.maxBin)

    def grad_hist_max(self):
        p = self.L.musica_oracle_grad_hist_max(self.h)
        return (p.maxValue, p.maxBin)

    def contrast_curve(self, level):
        return self.L.musica_oracle_contrast_curve(self.h, level).contents.as_array()

    def grad_curve(self):
        c = self.L.musica_oracle_grad_curve(self.h).contents
        return c.as_array(), (c.t0, c.ta, c.t1)

    def contrast_params(self, level):
        p = self.L.musica_oracle_contrast_params(self.h, level)
        return (p.lowContrastFactor, p.highContrastFactor)

    def nr_params(self, level):
        p = self.L.musica_oracle_nr_params(self.h, level)
        return (p.lowCnr, p.lowFactor, p.highCnr, p.highFactor)

    def minmax(self):
        a, b = C.c_float(), C.c_float()
        self.L.musica_oracle_minmax(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def stats(self):
        s = Stats()
        self.L.musica_oracle_stats(self.h, C.byref(s))
        return s

    def clahe_hist(self):
        return np.ctypeslib.as_array(self.L.musica_oracle_clahe_hist(self.h), shape=(4, 4, 256)).copy()

    def clahe_curves(self):
        p = C.cast(self.L.musica_oracle_clahe_curves(self.h), C.POINTER(C.c_float))
        return np.ctypeslib.as_array(p, shape=(4, 4, 256, 2)).copy()

    def out_pixels(self):
        n = self.N - 20
        out = np.empty((n, n), dtype=np.uint8)
        assert self.L.musica_oracle_out_pixels(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8))) == 1
        return out

    def render_noise_hist(self):
        out = np.empty((128, 512, 4), dtype=np.uint8)
        self.L.musica_oracle_render_noise_hist(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def render_grad_hist(self):
        out = np.empty((128, 512, 4), dtype=np.uint8)
        self.L.musica_oracle_render_grad_hist(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def debug_process(self, directory):
        assert self.L.musica_oracle_debug_process(self.h, os.fsencode(directory)) == 1

    def save_out_image(self, path):
        assert self.L.musica_oracle_save_out_image(self.h, os.fsencode(path)) == 1


# ---- single-shader helpers (numpy in, numpy out) ---------------------------

def _sq(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[0] == a.shape[1]
    return a


def k_sqrt(px):
    px = np.ascontiguousarray(px, dtype=np.uint16)
    out = np.empty(px.shape, dtype=np.float32)
    lib().musica_oracle_k_sqrt(px.ctypes.data_as(C.POINTER(C.c_uint16)), px.shape[0], _f32p(out))
    return out


def _reduce(fn, a):
    a = _sq(a)
    os_ = (a.shape[0] + 7) // 8
    out = np.zeros((os_, os_), dtype=np.float32)
    fn(_f32p(a), a.shape[0], _f32p(out))
    return out


def k_max_reduce(a):
    return _reduce(lib().musica_oracle_k_max_reduce, a)


def k_min_reduce(a):
    return _reduce(lib().musica_oracle_k_min_reduce, a)


def chain(fn, a):
    while a.shape[0] > 1:
        a = fn(a)
    return float(a[0, 0])


def k_normalize(a, minv, maxv):
    a = _sq(a)
    out = np.empty_like(a)
    lib().musica_oracle_k_normalize(_f32p(a), a.shape[0], minv, maxv, _f32p(out))
    return out


def k_smooth(a, order=ORDER_REFERENCE):
    a = _sq(a)
    out = np.empty_like(a)
    lib().musica_oracle_k_smooth(_f32p(a), a.shape[0], _f32p(out), order)
    return out


def k_smooth_upsampled(a, order=ORDER_REFERENCE):
    a = _sq(a)
    out = np.empty_like(a)
    lib().musica_oracle_k_smooth_upsampled(_f32p(a), a.shape[0], _f32p(out), order)
    return out


def k_downsample(a):
    a = _sq(a)
    os_ = (a.shape[0] + 1) // 2
    out = np.empty((os_, os_), dtype=np.float32)
    lib().musica_oracle_k_downsample(_f32p(a), a.shape[0], _f32p(out))
    return out


def k_upsample(a, out_side):
    a = _sq(a)
    out = np.zeros((out_side, out_side), dtype=np.float32)
    lib().musica_oracle_k_upsample(_f32p(a), a.shape[0], _f32p(out), out_side)
    return out


def k_sdev(a, order=ORDER_REFERENCE):
    a = _sq(a)
    out = np.empty_like(a)
    lib().musica_oracle_k_sdev(_f32p(a), a.shape[0], _f32p(out), order)
    return out


def k_noise_hist(sdev, groups):
    a = _sq(sdev)
    h = np.zeros(NOISE_BINS, dtype=np.uint32)
    lib().musica_oracle_k_noise_hist(_f32p(a), a.shape[0], groups, h.ctypes.data_as(C.POINTER(C.c_uint32)))
    return h


def k_histogram_max(hist):
    h = np.ascontiguousarray(hist, dtype=np.uint32)
    p = HistMaxPoint()
    lib().musica_oracle_k_histogram_max(h.ctypes.data_as(C.POINTER(C.c_uint32)), h.size, C.byref(p))
    return (p.maxValue, p.maxBin)


def k_contrast_curve_generate(max_bin, low, high):
    c = ContrastCurve()
    lib().musica_oracle_k_contrast_curve_generate(HistMaxPoint(0, max_bin), ContrastParams(low, high), C.byref(c))
    return c


def k_contrast_curve_apply(band, sdev, curve):
    b = _sq(band)
    s = _sq(sdev)
    out = np.empty_like(b)
    lib().musica_oracle_k_contrast_curve_apply(_f32p(b), _f32p(s), b.shape[0], C.byref(curve), _f32p(out))
    return out


def k_cnr(sdev, max_bin):
    s = _sq(sdev)
    out = np.empty_like(s)
    lib().musica_oracle_k_cnr(_f32p(s), s.shape[0], HistMaxPoint(0, max_bin), _f32p(out))
    return out


def k_noise_reduction(band, cnr, params):
    b = _sq(band)
    c = _sq(cnr)
    out = np.empty_like(b)
    lib().musica_oracle_k_noise_reduction(_f32p(b), b.shape[0], _f32p(c), c.shape[0], NrParams(*params), _f32p(out))
    return out


def k_relevant(normalized, cnr):
    a = _sq(normalized)
    c = _sq(cnr)
    out = np.empty_like(a)
    lib().musica_oracle_k_relevant(_f32p(a), a.shape[0], _f32p(c), c.shape[0], _f32p(out))
    return out


def k_gradation_histogram(img, relevant, groups):
    a = _sq(img)
    r = _sq(relevant)
    h = np.zeros(GRAD_BINS, dtype=np.uint32)
    lib().musica_oracle_k_gradation_histogram(_f32p(a), _f32p(r), a.shape[0], groups,
                                               h.ctypes.data_as(C.POINTER(C.c_uint32)))
    return h


def k_gradation_curve_generate(hist):
    h = np.ascontiguousarray(hist, dtype=np.uint32)
    c = GradCurve()
    lib().musica_oracle_k_gradation_curve_generate(h.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(c))
    return c


def k_apply_gradation_curve(img, curve):
    a = _sq(img)
    out = np.empty_like(a)
    lib().musica_oracle_k_apply_gradation_curve(_f32p(a), a.shape[0], C.byref(curve), _f32p(out))
    return out


def get_y(points_xy, x):
    pts = (Point * (len(points_xy) + 1))()
    for i, (px, py) in enumerate(points_xy):
        pts[i].x, pts[i].y = px, py
    return lib().musica_oracle_get_y(pts, len(points_xy), x)


class Tunables(C.Structure):
    """musica_tunables (include/musica.h)."""
    _fields_ = [("nr_high_cnr", C.c_float), ("nr_max_high_factor", C.c_float), ("nr_low_cnr", C.c_float), ("nr_min_low_factor", C.c_float),
                ("high_contrast_max_reduction", C.c_float), ("low_contrast_max_enhancement", C.c_float),
                ("linear_low_contrast", C.c_uint32), ("linear_high_contrast", C.c_uint32)]


def default_tunables(**overrides):
    t = Tunables()
    lib().musica_oracle_tunables_default(C.cast(C.byref(t), C.c_void_p))
    for k, v in overrides.items():
        if k not in dict(Tunables._fields_):
            raise KeyError(k)
        setattr(t, k, v)
    return t


def host_contrast_params(level, levels, tunables=None):
    if tunables is None:
        p = lib().musica_oracle_host_contrast_params(level, levels)
    else:
        p = lib().musica_oracle_host_contrast_params_ex(level, levels, C.cast(C.byref(tunables), C.c_void_p))
    return (p.lowContrastFactor, p.highContrastFactor)


def host_nr_params(i, tunables=None):
    if tunables is None:
        p = lib().musica_oracle_host_nr_params(i)
    else:
        p = lib().musica_oracle_host_nr_params_ex(i, C.cast(C.byref(tunables), C.c_void_p))
    return (p.lowCnr, p.lowFactor, p.highCnr, p.highFactor)


def write_bmp_gray(path, data):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    h, w = d.shape
    assert lib().musica_oracle_write_bmp_gray(os.fsencode(path), w, h, d.ctypes.data_as(C.POINTER(C.c_uint8))) == 1


def write_bmp_rgba(path, data):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    h, w, c = d.shape
    assert c == 4
    assert lib().musica_oracle_write_bmp_rgba(os.fsencode(path), w, h, d.ctypes.data_as(C.POINTER(C.c_uint8))) == 1


def read_raw(path, image_size):
    out = np.empty((image_size, image_size), dtype=np.uint16)
    ok = lib().musica_oracle_read_raw(os.fsencode(path), image_size, out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out if ok == 1 else None


def ref_bmp_available():
    return os.path.exists(REF_BMP_PATH)


def ref_write_bmp_gray(path, data):
    """The REFERENCE's stbi_write_bmp (compiled from its vendored stb header into oracle/_ref)."""
    L = C.CDLL(REF_BMP_PATH)
    L.ref_write_bmp_gray.restype = C.c_int
    L.ref_write_bmp_gray.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    d = np.ascontiguousarray(data, dtype=np.uint8)
    h, w = d.shape
    assert L.ref_write_bmp_gray(os.fsencode(path), w, h, d.ctypes.data_as(C.POINTER(C.c_uint8))) != 0


def ref_write_bmp_rgba(path, data):
    """The REFERENCE's stbi_write_bmp with four components, as debugProcess calls it for its two plots."""
    L = C.CDLL(REF_BMP_PATH)
    L.ref_write_bmp_rgba.restype = C.c_int
    L.ref_write_bmp_rgba.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    d = np.ascontiguousarray(data, dtype=np.uint8)
    h, w, c = d.shape
    assert c == 4
    assert L.ref_write_bmp_rgba(os.fsencode(path), w, h, d.ctypes.data_as(C.POINTER(C.c_uint8))) != 0
