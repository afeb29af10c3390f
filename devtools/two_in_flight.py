"""Experiment: K steps of the C4 shard with one context (steps back to back on one stream set) against two contexts whose
steps alternate (step n + 1 is enqueued on the other context's streams before step n has drained).
    python devtools/two_in_flight.py [n] [levels] [batch] [steps]
"""
import os
import sys
import time

import numpy as np

if os.environ.get("TIF_TORCH") == "1":
    import torch  # noqa: F401  (the library then binds to the HIP runtime bundled with torch)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp  # noqa: E402
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    levels = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    px = np.stack([phantom(n, 100 + k) for k in range(batch)])
    ctx = []
    flags = mp.FLAG_LINEAR if os.environ.get("TIF_LINEAR") == "1" else 0
    nmax = int(os.environ.get("TIF_MAX", "4"))
    for k in range(nmax):
        p = mp.MusicaProcessing(0)
        assert p.init(n, levels=levels, batch=batch, flags=flags), mp.last_error()
        p.upload(px)
        for _ in range(3):
            assert p.execute_device()
        p.sync()
        ctx.append(p)
    skip = int(os.environ.get("TIF_SKIP", "0"))       # use contexts skip .. skip + n - 1 (which hardware queues a set lands on)
    for nctx in list(range(1, nmax + 1 - skip)) * 2:
        use = ctx[skip:skip + nctx]
        for p in use:
            p.execute_device()
        for p in use:
            p.sync()
        t0 = time.perf_counter()
        for s in range(steps):
            use[s % nctx].execute_device()
        for p in use:
            p.sync()
        dt = (time.perf_counter() - t0) / steps
        print("contexts in flight %d: %.4f ms per step, %.1f GP/s" % (nctx, dt * 1e3, batch * n * n / dt / 1e9), flush=True)
    ref = ctx[0].graded().copy()
    for p in ctx[1:]:
        assert np.array_equal(p.graded(), ref)
    print("outputs identical")


if __name__ == "__main__":
    main()
