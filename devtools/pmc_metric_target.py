"""Target of bench.py's own rocprofv3 --pmc passes: 8 warm-up + 16 counted stand-alone launches of the metric kernel at 4096^2 f32
rotating over 8 distinct plane pairs (from HBM; symbol k_reduce_dma<4>) and the same of the copy-shaped ceiling kernel. No torch:
the process is up in ~2 s.   rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -o pmc -- python3 devtools/pmc_metric_target.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp  # noqa: E402

p = mp.MusicaProcessing()
assert p.init(64, levels=4, batch=1, flags=mp.FLAG_NO_AUTOTUNE | mp.FLAG_NO_GRAPH), mp.last_error()
print("metric kernel / copy us:", p.k_reduce_cold(4096, nbuf=8, iters=16))
p.cleanup()
