"""Target of the rocprofv3 --pmc passes (one dispatch = one counter row, so everything is launched eagerly):
  1. 16 stand-alone launches of the metric kernel at 4096^2 f32 rotating over 8 distinct buffer pairs (HBM: k_reduce_dma<4>)
     and 16 of the copy-shaped ceiling kernel k_copy41 on the same buffers;
  2. 16 stand-alone launches on one buffer pair (Infinity-Cache resident: k_reduce_dma<2>);
  3. 3 pipeline steps of workload C4 (8 x 2048^2, 6 levels).
Usage: rocprofv3 --pmc <counters> --output-format csv -d <dir> -o pmc -- python3 devtools/pmc_target.py"""
import os
import sys

import numpy as np

os.environ["MUSICA_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp  # noqa: E402
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom  # noqa: E402

p = mp.MusicaProcessing()
assert p.init(2048, levels=6, batch=8)
print("stand-alone rotating (kernel us, copy us)", p.k_reduce_cold(4096, nbuf=8, iters=16))
print("stand-alone one buffer us", p.k_reduce_timed(4096, batch=1, iters=16))
px = np.stack([phantom(2048, 100 + k) for k in range(8)])
p.upload(px)
for _ in range(3):
    assert p.execute_device()
p.sync()
print("done")
