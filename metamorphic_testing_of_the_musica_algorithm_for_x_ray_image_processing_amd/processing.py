"""ctypes mirror of the reference's pipeline object on top of libmusica_hip.so.

`MusicaProcessing` keeps the public surface of `class VulkanProcessing`
(reference include/vk_processing.h:281-355): init(imageSize), execute(imageData),
saveOutImage(filePath), debugProcess(), cleanup(), getImageSize() — same argument
meaning, `bool` results, and an error line on stderr when a call fails. Everything
else (`batch`, `levels`, getters for intermediates) is the extension surface the
parity tests and the batch driver use.

There is no CPU fallback: the shared library is HIP-only and every call fails
loudly when the extension or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmusica_hip.so")
CLI_PATH = os.path.join(_HERE, "musica-standalone")

MAX_POINTS = 256
NOISE_BINS = 2048
GRAD_BINS = 1024
OUT_MARGIN = 10

HIST_RENDER_WIDTH, HIST_RENDER_HEIGHT = 512, 128   # histRenderWidth / histRenderHeight, include/vk_processing.h:31-32

FLAG_CLAHE = 0x1
FLAG_NO_GRAPH = 0x2
FLAG_GENERIC_KERNELS = 0x4
FLAG_NO_AUTOTUNE = 0x8
FLAG_LINEAR = 0x10
FLAG_REFERENCE_ORDER = 0x20
FLAG_ONE_SHOT = 0x40

IMG_NORMALIZED, IMG_DOWNSAMPLED, IMG_BANDPASS, IMG_SDEV, IMG_CNR, IMG_EXPAND = 0, 1, 2, 3, 4, 5
IMG_GRADED, IMG_RELEVANT, IMG_LOWPASS, IMG_EXP_BANDPASS, IMG_SQRT, IMG_CLAHE_GRADED, IMG_CONTRAST_BAND = 6, 7, 8, 9, 10, 11, 12
STAGE_NORM, STAGE_REDUCE, STAGE_ANALYSIS, STAGE_EXPAND, STAGE_GRADATION = 0, 1, 2, 3, 4

KERNEL_NAMES = ["minmax", "normalize", "reduce_l0", "reduce_rest", "band_l0", "band_rest", "sdev_hist", "curves",
                "cnr", "expand_l0", "expand_rest", "grad_hist", "grad_curve", "grad_apply"]
KERNEL_ID = {n: i for i, n in enumerate(KERNEL_NAMES)}


class Params(C.Structure):
    _fields_ = [("image_size", C.c_uint32), ("levels", C.c_uint32), ("batch", C.c_uint32),
                ("device", C.c_int32), ("flags", C.c_uint32)]


class Tunables(C.Structure):
    """musica_tunables (include/musica.h): the constants of include/vk_processing.h:39-49 and the two LINEAR_* #defines of :16-17."""
    _fields_ = [("nr_high_cnr", C.c_float), ("nr_max_high_factor", C.c_float), ("nr_low_cnr", C.c_float), ("nr_min_low_factor", C.c_float),
                ("high_contrast_max_reduction", C.c_float), ("low_contrast_max_enhancement", C.c_float),
                ("linear_low_contrast", C.c_uint32), ("linear_high_contrast", C.c_uint32)]


def default_tunables(**overrides):
    """The reference's values, with keyword overrides (e.g. linear_low_contrast=1, nr_low_cnr=2.5)."""
    t = Tunables()
    load_library().musica_tunables_default(C.byref(t))
    for k, v in overrides.items():
        if k not in dict(Tunables._fields_):
            raise KeyError(k)
        setattr(t, k, v)
    return t


class HistMaxPoint(C.Structure):
    _fields_ = [("maxValue", C.c_uint32), ("maxBin", C.c_uint32)]


class ContrastParams(C.Structure):
    _fields_ = [("lowContrastFactor", C.c_float), ("highContrastFactor", C.c_float)]


class Point(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class ContrastCurve(C.Structure):
    _fields_ = [("points", Point * MAX_POINTS), ("pointsCount", C.c_uint32)]

    def as_array(self):
        return np.array([(self.points[i].x, self.points[i].y) for i in range(self.pointsCount)], dtype=np.float32)


class NrParams(C.Structure):
    _fields_ = [("lowCnr", C.c_float), ("lowFactor", C.c_float), ("highCnr", C.c_float), ("highFactor", C.c_float)]


class GradCurve(C.Structure):
    _fields_ = [("points", Point * MAX_POINTS), ("pointsCount", C.c_uint32),
                ("t0", C.c_float), ("ta", C.c_float), ("t1", C.c_float)]

    def as_array(self):
        return np.array([(self.points[i].x, self.points[i].y) for i in range(self.pointsCount)], dtype=np.float32)


class Stats(C.Structure):
    _fields_ = [("image_id", C.c_uint32), ("min_sqrt", C.c_float), ("max_sqrt", C.c_float),
                ("noise_max_bin", C.c_uint32 * 4), ("noise_max_value", C.c_uint32 * 4),
                ("grad_max_bin", C.c_uint32), ("grad_max_value", C.c_uint32),
                ("mean_cnr", C.c_float), ("t0", C.c_float), ("ta", C.c_float), ("t1", C.c_float)]

    FLOAT_FIELDS = ("min_sqrt", "max_sqrt", "mean_cnr", "t0", "ta", "t1")

    def as_row(self):
        """The struct as a flat list of 14 floats (comparisons in tests; the gather itself moves the raw bytes)."""
        return [float(self.image_id), self.min_sqrt, self.max_sqrt] + [float(v) for v in self.noise_max_bin] + \
               [float(self.grad_max_bin), self.mean_cnr, self.t0, self.ta, self.t1, float(self.grad_max_value)]


# Every symbol include/musica.h declares: (restype, argtypes). tests/test_abi.py checks the
# shared object exports exactly these.
_VP = C.c_void_p
_U8P, _U16P, _U32P, _F32P = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_float)
ABI = {
    "musica_create": (_VP, [C.POINTER(Params)]),
    "musica_destroy": (None, [_VP]),
    "musica_get_image_size": (C.c_uint32, [_VP]),
    "musica_get_levels": (C.c_uint32, [_VP]),
    "musica_get_batch": (C.c_uint32, [_VP]),
    "musica_get_level_size": (C.c_uint32, [_VP, C.c_uint32]),
    "musica_fuses_gradation_histogram": (C.c_int, [_VP]),
    "musica_fuses_reduce_band": (C.c_int, [_VP]),
    "musica_fuses_sdev": (C.c_int, [_VP]),
    "musica_get_dispatch": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "musica_execute": (C.c_int, [_VP, _U16P]),
    "musica_execute_device": (C.c_int, [_VP, _VP]),
    "musica_execute_stream": (C.c_int, [_VP, C.POINTER(_VP), C.c_uint32, C.POINTER(Stats)]),
    "musica_host_alloc": (_VP, [_VP, C.c_size_t]),
    "musica_host_free": (None, [_VP, _VP]),
    "musica_upload": (C.c_int, [_VP, _U16P]),
    "musica_input_device_ptr": (_VP, [_VP]),
    "musica_sync": (C.c_int, [_VP]),
    "musica_get_graded": (C.c_int, [_VP, _F32P]),
    "musica_save_out_image": (C.c_int, [_VP, C.c_uint32, C.c_char_p]),
    "musica_get_out_pixels": (C.c_int, [_VP, C.c_uint32, _U8P]),
    "musica_get_image": (C.c_int, [_VP, C.c_uint32, C.c_int, C.c_uint32, _F32P]),
    "musica_image_side": (C.c_uint32, [_VP, C.c_int, C.c_uint32]),
    "musica_get_noise_hist": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _U32P]),
    "musica_get_grad_hist": (C.c_int, [_VP, C.c_uint32, _U32P]),
    "musica_get_noise_hist_max": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.POINTER(HistMaxPoint)]),
    "musica_get_grad_hist_max": (C.c_int, [_VP, C.c_uint32, C.POINTER(HistMaxPoint)]),
    "musica_get_contrast_curve": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.POINTER(ContrastCurve)]),
    "musica_get_grad_curve": (C.c_int, [_VP, C.c_uint32, C.POINTER(GradCurve)]),
    "musica_get_contrast_params": (C.c_int, [_VP, C.c_uint32, C.POINTER(ContrastParams)]),
    "musica_get_nr_params": (C.c_int, [_VP, C.c_uint32, C.POINTER(NrParams)]),
    "musica_get_minmax": (C.c_int, [_VP, C.c_uint32, _F32P, _F32P]),
    "musica_get_stats": (C.c_int, [_VP, C.c_uint32, C.POINTER(Stats)]),
    "musica_stats_device": (C.c_int, [_VP, _VP, C.c_uint32]),
    "musica_stats_device_strided": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32]),
    "musica_get_clahe_hist": (C.c_int, [_VP, C.c_uint32, _U32P]),
    "musica_get_clahe_curves": (C.c_int, [_VP, C.c_uint32, C.POINTER(Point)]),
    "musica_debug_process": (C.c_int, [_VP, C.c_uint32, C.c_char_p]),
    "musica_render_noise_hist": (C.c_int, [_VP, C.c_uint32, _U8P]),
    "musica_render_grad_hist": (C.c_int, [_VP, C.c_uint32, _U8P]),
    "musica_debug_set_image": (C.c_int, [_VP, C.c_uint32, C.c_int, C.c_uint32, _F32P]),
    "musica_debug_run_stage": (C.c_int, [_VP, C.c_int]),
    "musica_profile_enable": (C.c_int, [_VP, C.c_int]),
    "musica_profile_reset": (C.c_int, [_VP]),
    "musica_profile_get": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "musica_k_reduce": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, _VP, C.c_uint32, C.c_uint32]),
    "musica_selftest_exact_math": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "musica_k_reduce_timed": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, _VP, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
    "musica_k_reduce_timed_rot": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, _VP, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
    "musica_k_copy41_timed_rot": (C.c_int, [_VP, _VP, C.c_uint32, _VP, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
    "musica_device_alloc": (_VP, [_VP, C.c_size_t]),
    "musica_device_free": (None, [_VP, _VP]),
    "musica_memcpy_h2d": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "musica_memcpy_d2h": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "musica_read_raw": (C.c_int, [C.c_char_p, C.c_uint32, _U16P]),
    "musica_write_bmp_gray": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, _U8P]),
    "musica_write_bmp_rgba": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, _U8P]),
    "musica_pipeline_create": (_VP, [C.POINTER(Params), C.c_uint32]),
    "musica_pipeline_destroy": (None, [_VP]),
    "musica_pipeline_depth": (C.c_uint32, [_VP]),
    "musica_pipeline_context": (_VP, [_VP, C.c_uint32]),
    "musica_pipeline_upload": (C.c_int, [_VP, _U16P]),
    "musica_pipeline_prime": (C.c_int, [_VP, C.c_uint32]),
    "musica_pipeline_calibration": (C.c_uint32, [_VP, _F32P]),
    "musica_pipeline_step": (C.c_int, [_VP, _VP]),
    "musica_pipeline_last": (_VP, [_VP]),
    "musica_pipeline_sync": (C.c_int, [_VP]),
    "musica_last_error": (C.c_char_p, []),
    "musica_abi_version": (C.c_uint32, []),
    "musica_create_ex": (_VP, [C.POINTER(Params), C.POINTER(Tunables)]),
    "musica_tunables_default": (None, [C.POINTER(Tunables)]),
    "musica_get_tunables": (C.c_int, [_VP, C.POINTER(Tunables)]),
    "musica_pipeline_create_ex": (_VP, [C.POINTER(Params), C.c_uint32, C.POINTER(Tunables)]),
    "musica_device_count": (C.c_int, []),
}

_lib = None


def load_library():
    """Load libmusica_hip.so (built in-tree by build.py). Raises if it is missing: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libmusica_hip.so is not built: run `python -m %s.build` (hipcc, gfx950)" % __package__)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return (load_library().musica_last_error() or b"").decode("utf-8", "replace")


def device_count():
    return load_library().musica_device_count()


def _f32p(a):
    return a.ctypes.data_as(_F32P)


class MusicaProcessing:
    """Drop-in for the reference's VulkanProcessing (one HIP stream on one device per object)."""

    def __init__(self, device=0):
        # reference: VulkanProcessing(VulkanState*) — the state object selected the physical device
        self._lib = load_library()
        self._device = int(device)
        self._h = None
        self._owned = True
        self.imageSize = 0
        self.pyramidLevels = 0
        self.batch = 1

    @classmethod
    def _borrow(cls, handle, device=0):
        """A view of a context something else owns (a MusicaPipeline): every getter works, cleanup() leaves it alone."""
        self = cls(device)
        self._h = handle
        self._owned = False
        self.imageSize = self._lib.musica_get_image_size(handle)
        self.pyramidLevels = self._lib.musica_get_levels(handle)
        self.batch = self._lib.musica_get_batch(handle)
        return self

    # ---- the reference's interface -------------------------------------------------
    def init(self, imageSize, outImageViews=None, levels=0, batch=1, flags=0, tunables=None):
        """bool init(uint32_t imageSize, std::vector<VkImageView>*) — src/vk_processing.cpp:1984-2020.
        tunables: a Tunables (default_tunables(...)) — the reference's compile-time constants as runtime values; None = the reference's."""
        if self._h:
            self.cleanup()
        p = Params(int(imageSize), int(levels), int(batch), self._device, int(flags))
        h = self._lib.musica_create_ex(C.byref(p), C.byref(tunables) if tunables is not None else None)
        if not h:
            return False
        self._h = h
        self.imageSize = self._lib.musica_get_image_size(h)
        self.pyramidLevels = self._lib.musica_get_levels(h)
        self.batch = self._lib.musica_get_batch(h)
        return True

    def execute(self, imageData):
        """bool execute(const uint16_t* imageData) — src/vk_processing.cpp:2104-2601 (synchronous)."""
        px = self._pixels(imageData)
        return self._lib.musica_execute(self._h, px.ctypes.data_as(_U16P)) == 1

    def saveOutImage(self, filePath, image_index=0):
        """bool saveOutImage(std::string filePath) — src/vk_processing.cpp:2603-2645."""
        return self._lib.musica_save_out_image(self._h, image_index, os.fsencode(filePath)) == 1

    def debugProcess(self, directory=".", image_index=0):
        """bool debugProcess() — src/vk_processing.cpp:2661-2809 (the reference writes into the cwd)."""
        return self._lib.musica_debug_process(self._h, image_index, os.fsencode(directory)) == 1

    def render_noise_hist(self, image_index=0):
        """noise_hist_render.comp on the cnr level's histogram (RENDER_HISTS, src/vk_processing.cpp:1260-1266, :2347): uint8 [128, 512, 4]."""
        out = np.empty((HIST_RENDER_HEIGHT, HIST_RENDER_WIDTH, 4), dtype=np.uint8)
        self._ok(self._lib.musica_render_noise_hist(self._h, image_index, out.ctypes.data_as(_U8P)), "musica_render_noise_hist")
        return out

    def render_grad_hist(self, image_index=0):
        """gradation_curve_debug_render.comp on the gradation histogram + tone curve (src/vk_processing.cpp:1668-1675, :2508)."""
        out = np.empty((HIST_RENDER_HEIGHT, HIST_RENDER_WIDTH, 4), dtype=np.uint8)
        self._ok(self._lib.musica_render_grad_hist(self._h, image_index, out.ctypes.data_as(_U8P)), "musica_render_grad_hist")
        return out

    def cleanup(self):
        """bool cleanup() — src/vk_processing.cpp:2647-2651."""
        if self._h:
            if self._owned:
                self._lib.musica_destroy(self._h)
            self._h = None
        return True

    def getImageSize(self):
        return self.imageSize

    # ---- extension surface ------------------------------------------------------------
    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass

    def _pixels(self, imageData):
        px = np.ascontiguousarray(imageData, dtype=np.uint16)
        need = self.batch * self.imageSize * self.imageSize
        if px.size != need:
            raise ValueError("expected %d pixels (batch %d x %d x %d), got %d" % (need, self.batch, self.imageSize, self.imageSize, px.size))
        return px

    def _ok(self, rc, what):
        if rc != 1:
            raise RuntimeError("%s failed: %s" % (what, last_error()))

    def level_size(self, level):
        return self._lib.musica_get_level_size(self._h, level)

    def upload(self, imageData):
        px = self._pixels(imageData)
        self._ok(self._lib.musica_upload(self._h, px.ctypes.data_as(_U16P)), "musica_upload")

    def input_device_ptr(self):
        return self._lib.musica_input_device_ptr(self._h)

    def execute_device(self, d_pixels=None):
        """Enqueue the pipeline on input already resident in HBM (default: the uploaded buffer)."""
        ptr = d_pixels if d_pixels is not None else self.input_device_ptr()
        return self._lib.musica_execute_device(self._h, ptr) == 1

    def sync(self):
        self._ok(self._lib.musica_sync(self._h), "musica_sync")

    def host_alloc(self, shape, dtype=np.uint16):
        """A numpy array in page-locked host memory (inputs of execute_stream move at the PCIe rate from it). Free with host_free()."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self._lib.musica_host_alloc(self._h, n)
        if not p:
            raise MemoryError(last_error())
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).view(dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            self._lib.musica_host_free(self._h, p)

    def execute_stream(self, batches, want_stats=False):
        """Pipelined execute over a list of (batch, N, N) uint16 arrays: H2D of batch j + 1 under the kernels of batch j.
        Returns True / False, or (ok, [Stats per image of the sequence]) with want_stats."""
        arrs = [self._pixels(b) for b in batches]
        ptrs = (_VP * len(arrs))(*[a.ctypes.data for a in arrs])
        st = (Stats * (len(arrs) * self.batch))() if want_stats else None
        ok = self._lib.musica_execute_stream(self._h, ptrs, len(arrs), st) == 1
        return (ok, list(st)) if want_stats else ok

    def graded(self):
        n = self.imageSize
        out = np.empty((self.batch, n, n), dtype=np.float32)
        self._ok(self._lib.musica_get_graded(self._h, _f32p(out)), "musica_get_graded")
        return out

    def out_pixels(self, image_index=0):
        n = self.imageSize - 2 * OUT_MARGIN
        out = np.empty((n, n), dtype=np.uint8)
        self._ok(self._lib.musica_get_out_pixels(self._h, image_index, out.ctypes.data_as(_U8P)), "musica_get_out_pixels")
        return out

    def image(self, kind, level=0, image_index=0):
        s = self._lib.musica_image_side(self._h, kind, level)
        if s == 0:
            raise KeyError("no image kind=%d level=%d" % (kind, level))
        out = np.empty((s, s), dtype=np.float32)
        self._ok(self._lib.musica_get_image(self._h, image_index, kind, level, _f32p(out)), "musica_get_image")
        return out

    def set_image(self, kind, level, arr, image_index=0):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        self._ok(self._lib.musica_debug_set_image(self._h, image_index, kind, level, _f32p(a)), "musica_debug_set_image")

    def run_stage(self, stage):
        self._ok(self._lib.musica_debug_run_stage(self._h, stage), "musica_debug_run_stage")

    def noise_hist(self, level, image_index=0):
        out = np.empty(NOISE_BINS, dtype=np.uint32)
        self._ok(self._lib.musica_get_noise_hist(self._h, image_index, level, out.ctypes.data_as(_U32P)), "musica_get_noise_hist")
        return out

    def grad_hist(self, image_index=0):
        out = np.empty(GRAD_BINS, dtype=np.uint32)
        self._ok(self._lib.musica_get_grad_hist(self._h, image_index, out.ctypes.data_as(_U32P)), "musica_get_grad_hist")
        return out

    def noise_hist_max(self, level, image_index=0):
        p = HistMaxPoint()
        self._ok(self._lib.musica_get_noise_hist_max(self._h, image_index, level, C.byref(p)), "musica_get_noise_hist_max")
        return (p.maxValue, p.maxBin)

    def grad_hist_max(self, image_index=0):
        p = HistMaxPoint()
        self._ok(self._lib.musica_get_grad_hist_max(self._h, image_index, C.byref(p)), "musica_get_grad_hist_max")
        return (p.maxValue, p.maxBin)

    def contrast_curve(self, level, image_index=0):
        c = ContrastCurve()
        self._ok(self._lib.musica_get_contrast_curve(self._h, image_index, level, C.byref(c)), "musica_get_contrast_curve")
        return c.as_array()

    def grad_curve(self, image_index=0):
        c = GradCurve()
        self._ok(self._lib.musica_get_grad_curve(self._h, image_index, C.byref(c)), "musica_get_grad_curve")
        return c.as_array(), (c.t0, c.ta, c.t1)

    def tunables(self):
        t = Tunables()
        self._ok(self._lib.musica_get_tunables(self._h, C.byref(t)), "musica_get_tunables")
        return t

    def contrast_params(self, level):
        p = ContrastParams()
        self._ok(self._lib.musica_get_contrast_params(self._h, level, C.byref(p)), "musica_get_contrast_params")
        return (p.lowContrastFactor, p.highContrastFactor)

    def nr_params(self, level):
        p = NrParams()
        self._ok(self._lib.musica_get_nr_params(self._h, level, C.byref(p)), "musica_get_nr_params")
        return (p.lowCnr, p.lowFactor, p.highCnr, p.highFactor)

    def minmax(self, image_index=0):
        a, b = C.c_float(), C.c_float()
        self._ok(self._lib.musica_get_minmax(self._h, image_index, C.byref(a), C.byref(b)), "musica_get_minmax")
        return a.value, b.value

    def stats(self, image_index=0):
        s = Stats()
        self._ok(self._lib.musica_get_stats(self._h, image_index, C.byref(s)), "musica_get_stats")
        return s

    def stats_device(self, d_dst, image_id_base=0, image_id_stride=1):
        """Write batch x musica_stats into caller-owned device memory (async on the ctx stream);
        image_id = image_id_base + index * image_id_stride."""
        self._ok(self._lib.musica_stats_device_strided(self._h, d_dst, image_id_base, image_id_stride), "musica_stats_device")

    def clahe_hist(self, image_index=0):
        out = np.empty((4, 4, 256), dtype=np.uint32)
        self._ok(self._lib.musica_get_clahe_hist(self._h, image_index, out.ctypes.data_as(_U32P)), "musica_get_clahe_hist")
        return out

    def clahe_curves(self, image_index=0):
        out = np.empty((4, 4, 256, 2), dtype=np.float32)
        self._ok(self._lib.musica_get_clahe_curves(self._h, image_index, C.cast(out.ctypes.data, C.POINTER(Point))), "musica_get_clahe_curves")
        return out

    def dispatch(self):
        """(streams, graph): 1 / 2 streams and whether steps replay a captured hipGraph (musica_get_dispatch)."""
        st, g = C.c_int(0), C.c_int(0)
        self._lib.musica_get_dispatch(self._h, C.byref(st), C.byref(g))
        return st.value, bool(g.value)

    def dispatch_text(self):
        st, g = self.dispatch()
        return "%s, %s" % ({1: "one stream", 2: "two streams (analysis beside the reduce tail)"}[st],
                           "hipGraph replay" if g else "eager launches")

    def fuses_gradhist(self):
        """True when the level-0 expand kernel also accumulates the gradation histogram (no separate k_grad_hist launch)."""
        return self._lib.musica_fuses_gradation_histogram(self._h) == 1

    def fuses_sdev(self):
        """True when the expand launches of levels 0 .. 2 compute sdev themselves and the sdev launches of those levels store nothing."""
        return self._lib.musica_fuses_sdev(self._h) == 1

    def fuses_reduce_band(self):
        """True when level 0's reduce and band kernels are one launch (profile family `reduce_l0` covers both)."""
        return self._lib.musica_fuses_reduce_band(self._h) == 1

    # ---- profiling ----------------------------------------------------------------------
    def profile_enable(self, which=True):
        """True: every kernel family; False: off; a list of kernel names: only those families."""
        if which is True:
            mask = -1
        elif not which:
            mask = 0
        else:
            mask = 0
            for name in which:
                mask |= 1 << KERNEL_ID[name]
        self._ok(self._lib.musica_profile_enable(self._h, mask), "musica_profile_enable")

    def profile_reset(self):
        self._ok(self._lib.musica_profile_reset(self._h), "musica_profile_reset")

    def profile(self):
        """{kernel name: (mean microseconds, launches)} since the last reset."""
        out = {}
        for name, kid in KERNEL_ID.items():
            us, n = C.c_double(), C.c_uint64()
            self._ok(self._lib.musica_profile_get(self._h, kid, C.byref(us), C.byref(n)), "musica_profile_get")
            out[name] = (us.value, n.value)
        return out

    # ---- the metric kernel on caller-owned device buffers ---------------------------------
    def device_alloc(self, nbytes):
        p = self._lib.musica_device_alloc(self._h, nbytes)
        if not p:
            raise MemoryError(last_error())
        return p

    def device_free(self, ptr):
        self._lib.musica_device_free(self._h, ptr)

    def h2d(self, d_dst, arr):
        a = np.ascontiguousarray(arr)
        self._ok(self._lib.musica_memcpy_h2d(self._h, d_dst, a.ctypes.data, a.nbytes), "musica_memcpy_h2d")

    def d2h(self, arr, d_src):
        self._ok(self._lib.musica_memcpy_d2h(self._h, arr.ctypes.data, d_src, arr.nbytes), "musica_memcpy_d2h")

    def k_reduce_host(self, images):
        """Fused smooth + downsample of a (batch, S, S) f32 array staged through device buffers."""
        a = np.ascontiguousarray(images, dtype=np.float32)
        if a.ndim == 2:
            a = a[None]
        b, s, _ = a.shape
        pitch, so = (s + 3) & ~3, (s + 1) // 2
        opitch = (so + 3) & ~3
        padded = np.zeros((b, s, pitch), dtype=np.float32)
        padded[:, :, :s] = a
        d_in, d_out = self.device_alloc(padded.nbytes), self.device_alloc(b * so * opitch * 4)
        try:
            self.h2d(d_in, padded)
            self._ok(self._lib.musica_k_reduce(self._h, d_in, s, pitch, d_out, opitch, b), "musica_k_reduce")
            out = np.empty((b, so, opitch), dtype=np.float32)
            self.d2h(out, d_out)
        finally:
            self.device_free(d_in)
            self.device_free(d_out)
        return out[:, :, :so].copy()

    def selftest_exact_math(self):
        """Mismatch counts of the device-side exact shortcuts over every float bit pattern (all must be 0)."""
        out = (C.c_uint64 * 4)()
        self._ok(self._lib.musica_selftest_exact_math(self._h, out), "musica_selftest_exact_math")
        return list(out)

    def k_reduce_timed(self, side, batch=1, iters=50, seed=0):
        """Mean microseconds per launch of the metric kernel on a random side x side image in HBM."""
        pitch, so = (side + 3) & ~3, (side + 1) // 2
        opitch = (so + 3) & ~3
        rng = np.random.default_rng(seed)
        src = rng.random((batch, side, pitch), dtype=np.float32)
        d_in, d_out = self.device_alloc(src.nbytes), self.device_alloc(batch * so * opitch * 4)
        try:
            self.h2d(d_in, src)
            us = C.c_double()
            self._ok(self._lib.musica_k_reduce_timed(self._h, d_in, side, pitch, d_out, opitch, batch, 3, C.byref(us)), "warmup")
            self._ok(self._lib.musica_k_reduce_timed(self._h, d_in, side, pitch, d_out, opitch, batch, iters, C.byref(us)), "musica_k_reduce_timed")
        finally:
            self.device_free(d_in)
            self.device_free(d_out)
        return us.value


    def k_reduce_cold(self, side, nbuf=8, iters=64, copy_ceiling=True, seed=0):
        """Mean microseconds per launch of the metric kernel on side x side f32 images rotating over `nbuf` distinct
        input / output planes (nbuf * 5 * side^2 bytes must exceed the 256 MiB Infinity Cache for an HBM number),
        and of the copy-shaped ceiling kernel timed the same way. Returns (kernel_us, copy_us or None)."""
        so = side // 2
        rng = np.random.default_rng(seed)
        d_in, d_out = self.device_alloc(nbuf * side * side * 4), self.device_alloc(nbuf * so * so * 4)
        try:
            for k in range(nbuf):
                self.h2d(d_in + k * side * side * 4, rng.random((side, side), dtype=np.float32))
            us, cus = C.c_double(), C.c_double()
            for it in (nbuf, iters):    # first pass: warm-up (code object, TLB)
                self._ok(self._lib.musica_k_reduce_timed_rot(self._h, d_in, side, side, d_out, so, nbuf, it, C.byref(us)), "musica_k_reduce_timed_rot")
            if copy_ceiling:
                for it in (nbuf, iters):
                    self._ok(self._lib.musica_k_copy41_timed_rot(self._h, d_in, side, d_out, nbuf, it, C.byref(cus)), "musica_k_copy41_timed_rot")
        finally:
            self.device_free(d_in)
            self.device_free(d_out)
        return us.value, (cus.value if copy_ceiling else None)


class MusicaPipeline:
    """The C ABI's steps-in-flight object (musica_pipeline_*, include/musica.h): `depth` one-stream contexts of one GPU whose
    steps alternate, with the choice of hardware queues made by musica_pipeline_prime() (what bench.py times)."""

    def __init__(self, imageSize, levels=0, batch=1, depth=3, flags=0, device=0):
        self._lib = load_library()
        p = Params(int(imageSize), int(levels), int(batch), int(device), int(flags))
        self._p = self._lib.musica_pipeline_create(C.byref(p), int(depth))
        if not self._p:
            raise RuntimeError("musica_pipeline_create failed: " + last_error())
        self._device = int(device)
        self.depth = int(depth)
        self._pixels = int(batch) * int(imageSize) * int(imageSize)
        self._borrowed = []

    def _ok(self, rc, what):
        if rc != 1:
            raise RuntimeError("%s failed: %s" % (what, last_error()))

    def upload(self, images):
        """The same batch (batch x N x N uint16) into every context's input buffer."""
        px = np.ascontiguousarray(images, dtype=np.uint16)
        if px.size != self._pixels:      # the C side copies batch * N * N pixels whatever it is handed
            raise ValueError("expected %d pixels (batch x N x N), got %d" % (self._pixels, px.size))
        self._ok(self._lib.musica_pipeline_upload(self._p, px.ctypes.data_as(_U16P)), "musica_pipeline_upload")

    def prime(self, calibration_steps=0):
        self._ok(self._lib.musica_pipeline_prime(self._p, int(calibration_steps)), "musica_pipeline_prime")
        for w in self._borrowed:         # prime() destroys the contexts outside the window it keeps: wrappers handed out before it are void
            w._h = None
        self._borrowed = []

    def calibration(self):
        """{first context of the window: ms per step} of the windows prime() timed ({} when there was nothing to choose)."""
        ms = np.zeros(4, dtype=np.float32)
        n = self._lib.musica_pipeline_calibration(self._p, _f32p(ms))
        return {k: float(ms[k]) for k in range(n)}

    def context(self, k):
        h = self._lib.musica_pipeline_context(self._p, int(k))
        if not h:
            raise IndexError(last_error())
        w = MusicaProcessing._borrow(h, self._device)
        self._borrowed.append(w)
        return w

    def step(self, d_pixels=None):
        self._ok(self._lib.musica_pipeline_step(self._p, d_pixels), "musica_pipeline_step")

    def last(self):
        h = self._lib.musica_pipeline_last(self._p)
        if not h:
            raise RuntimeError(last_error())
        return MusicaProcessing._borrow(h, self._device)

    def sync(self):
        self._ok(self._lib.musica_pipeline_sync(self._p), "musica_pipeline_sync")

    def cleanup(self):
        if self._p:
            self._lib.musica_pipeline_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


def read_raw(path, image_size):
    """The raw reader of test/standalone/main.cpp:54-75; None when the file size does not match."""
    out = np.empty((image_size, image_size), dtype=np.uint16)
    ok = load_library().musica_read_raw(os.fsencode(path), image_size, out.ctypes.data_as(_U16P))
    return out if ok == 1 else None


def write_bmp_gray(path, data):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    h, w = d.shape
    return load_library().musica_write_bmp_gray(os.fsencode(path), w, h, d.ctypes.data_as(_U8P)) == 1


def write_bmp_rgba(path, data):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    h, w, c = d.shape
    assert c == 4
    return load_library().musica_write_bmp_rgba(os.fsencode(path), w, h, d.ctypes.data_as(_U8P)) == 1
