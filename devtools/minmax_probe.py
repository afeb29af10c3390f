"""k_minmax_u16 back to back on one resident batch (stage "norm" alone, eager): under rocprofv3 --kernel-trace, devtools/trace_levels.py-free summary."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom
n, L, b = 2048, 6, 8
p = mp.MusicaProcessing()
assert p.init(n, levels=L, batch=b, flags=mp.FLAG_LINEAR | mp.FLAG_NO_AUTOTUNE | mp.FLAG_NO_GRAPH)
p.upload(np.stack([phantom(n, 100 + k) for k in range(b)]))
for _ in range(60):
    p.run_stage(mp.STAGE_NORM)
p.sync()
p.cleanup()
