"""musica_execute_stream on small batches: fixed cost per call against cost per batch.   python devtools/stream_probe.py N L B"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom
n, levels, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = mp.MusicaProcessing()
assert p.init(n, levels=levels, batch=batch), mp.last_error()
px = np.stack([phantom(n, 3 + k) for k in range(batch)])
pinned = [p.host_alloc(px.shape) for _ in range(2)]
for b in pinned:
    b[...] = px
p.execute_stream([pinned[j & 1] for j in range(2)])
for reps in (1, 2, 8, 32, 128):
    t0 = time.perf_counter()
    assert p.execute_stream([pinned[j & 1] for j in range(reps)])
    dt = time.perf_counter() - t0
    print("%s reps %3d: %.3f ms per call, %.4f ms per batch, %.1f MP/s" % (p.dispatch_text(), reps, dt * 1e3, dt * 1e3 / reps, reps * batch * n * n / 1e6 / dt))
for b in pinned:
    p.host_free(b)
p.cleanup()
