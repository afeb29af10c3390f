"""Target of the rocprofv3 --pmc passes: 20 stand-alone launches of the metric kernel at 4096^2 (f32), then
3 pipeline steps of workload c3 (8 x 2048^2, 6 levels) with eager launches (one dispatch = one counter row)."""
import os
import sys

import numpy as np

os.environ["MUSICA_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp  # noqa: E402
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom  # noqa: E402

p = mp.MusicaProcessing()
assert p.init(2048, levels=6, batch=8)
print("standalone us", p.k_reduce_timed(4096, batch=1, iters=20))
px = np.stack([phantom(2048, 100 + k) for k in range(8)])
p.upload(px)
for _ in range(3):
    assert p.execute_device()
p.sync()
print("done")
