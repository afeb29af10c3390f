import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp
p = mp.MusicaProcessing(); assert p.init(64, levels=4)
lib = p._lib
for side, batch in [(4096, 1), (2048, 8)]:
    for pad in [0, 4, 16, 64, 128, 256, 1024]:
        pitch = side + pad; so = side // 2; opitch = so + (pad // 2 // 4) * 4
        d_in = p.device_alloc(batch * side * pitch * 4); d_out = p.device_alloc(batch * so * opitch * 4)
        src = np.random.default_rng(0).random((side * pitch,), dtype=np.float32)
        for b in range(batch):
            lib.musica_memcpy_h2d(p._h, d_in + b * side * pitch * 4, src.ctypes.data, src.nbytes)
        us = C.c_double()
        lib.musica_k_reduce_timed(p._h, d_in, side, pitch, d_out, opitch, batch, 10, C.byref(us))
        lib.musica_k_reduce_timed(p._h, d_in, side, pitch, d_out, opitch, batch, 100, C.byref(us))
        print("side %d x%d pad %4d: %.2f us  %.0f GB/s" % (side, batch, pad, us.value, 5 * side * side * batch / us.value / 1e3), flush=True)
        p.device_free(d_in); p.device_free(d_out)
