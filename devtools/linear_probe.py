"""One linear (one-stream) context without autotune, graph replays: every kernel alone on the chip. Target of
rocprofv3 --kernel-trace; devtools/trace_levels.py turns the trace into per-kernel, per-grid medians.
  python devtools/linear_probe.py N L batch [flags]      (MUSICA_* knobs choose the kernel forms)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom

if os.environ.get("PROBE_LIB"):   # A/B against a library built by devtools/build_variant.sh
    mp.LIB_PATH = os.path.join(ROOT, os.environ["PROBE_LIB"])
n, L, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
p = mp.MusicaProcessing()
assert p.init(n, levels=L, batch=b, flags=flags | mp.FLAG_LINEAR | (mp.FLAG_NO_AUTOTUNE if os.environ.get("PROBE_TUNE", "0") == "0" else 0))
p.upload(np.stack([phantom(n, 100 + k) for k in range(b)]))
idle = float(os.environ.get("PROBE_IDLE_MS", "0"))   # > 0: drain the device and sleep between steps (every step starts on an idle memory system)
import time
for _ in range(40):
    p.execute_device()
    if idle > 0:
        p.sync()
        time.sleep(idle / 1e3)
p.sync()
p.cleanup()
