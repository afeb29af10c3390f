"""One context, graph replays, for rocprofv3 --kernel-trace --stats: per-kernel averages of a C4 / C2 / C5 context (MUSICA_MM_* knobs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom
n, L, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = mp.MusicaProcessing()
assert p.init(n, levels=L, batch=b, flags=mp.FLAG_NO_AUTOTUNE | mp.FLAG_LINEAR)
p.upload(np.stack([phantom(n, 100 + k) for k in range(b)]))
for _ in range(40):
    p.execute_device()
p.sync()
p.cleanup()
