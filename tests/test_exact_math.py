"""csrc/exact_math.h: the cheaper instruction sequences used by the kernels equal the plain IEEE
expressions of the shaders for EVERY non-negative float (exhaustive, 2^31 inputs each, ~20 s on 8 threads)."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "build", "libmusica_exhaustive.so")


@pytest.fixture(scope="module")
def ex(ob):
    ob.build(force=not os.path.exists(LIB))
    lib = C.CDLL(LIB)
    for name in ("musica_check_div25", "musica_check_noise_bin"):
        fn = getattr(lib, name)
        fn.restype = C.c_long
        fn.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.musica_check_norm_div.restype = C.c_long
    lib.musica_check_norm_div.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    return lib


def test_div25_is_exact_for_every_nonnegative_float(ex):
    first = C.c_uint32()
    bad = ex.musica_check_div25(0, 0x7F800001, C.byref(first))      # 0 .. +inf inclusive
    assert bad == 0, "first mismatch at bits 0x%08x" % first.value


def test_noise_bin_is_exact_for_every_nonnegative_float(ex):
    first = C.c_uint32()
    bad = ex.musica_check_noise_bin(0, 0x7F800001, C.byref(first))
    assert bad == 0, "first mismatch at bits 0x%08x" % first.value
    bad = ex.musica_check_noise_bin(0x7F800001, 0x7FFFFFFF, C.byref(first))   # NaNs: both say "break"
    assert bad == 0


def test_norm_div_is_exact_on_its_whole_domain(ex):
    """(sqrtf(v) - min) / den through one reciprocal multiply + FMA residual step == the IEEE division for every
    16-bit v, every integer min 0..255 and every integer den 1..255 (4.3e9 cases, a few seconds)."""
    first = C.c_uint32()
    bad = ex.musica_check_norm_div(0, 256, C.byref(first))
    assert bad == 0, "first mismatch (min << 24 | den << 16 | v) = 0x%08x" % first.value
