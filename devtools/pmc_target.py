"""Target for the rocprofv3 --pmc passes: 20 stand-alone launches of the metric kernel at 4096^2, then 3 pipeline steps (c3)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom
p = mp.MusicaProcessing(); assert p.init(2048, levels=6, batch=8)
print("standalone us", p.k_reduce_timed(4096, batch=1, iters=20))
px = np.stack([phantom(2048, 100 + k) for k in range(8)])
p.upload(px)
for _ in range(3):
    assert p.execute_device()
p.sync()
print("done")
