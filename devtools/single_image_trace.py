"""One context, one image per execute, `steps` graph replays: the target of a rocprofv3 --kernel-trace run whose CSV
devtools/step_timeline.py turns into the timeline of one step.   python devtools/single_image_trace.py N L [flags] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom

n, levels = int(sys.argv[1]), int(sys.argv[2])
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
p = mp.MusicaProcessing()
assert p.init(n, levels=levels, batch=1, flags=flags), mp.last_error()
p.upload(phantom(n, 7)[None])
import time
for _ in range(3):
    p.execute_device()
p.sync()
t0 = time.perf_counter()
for _ in range(steps):
    p.execute_device()
p.sync()
print("N %d L %d flags %d: %.4f ms per image" % (n, p.pyramidLevels, flags, (time.perf_counter() - t0) / steps * 1e3))
p.cleanup()
