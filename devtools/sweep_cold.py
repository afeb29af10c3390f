"""Stand-alone metric kernel from HBM (rotating buffers) for a few sizes and rows-per-wavefront choices, with the
copy-shaped ceiling beside it. Usage: python devtools/sweep_cold.py [side,nbuf ...] [--rows 0,4,8]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
rows_list = (0, 2, 4, 8, 16, 32)
for a in sys.argv[1:]:
    if a.startswith("--rows="):
        rows_list = tuple(int(v) for v in a[7:].split(","))
cases = [tuple(int(v) for v in a.split(",")) for a in args] or [(4096, 8), (2048, 32), (8192, 3)]
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("MUSICA_REDUCE_ROWS",) if k in os.environ)
p = mp.MusicaProcessing()
assert p.init(64, levels=4)
for side, nbuf in cases:
    b = 5 * side * side
    for rows in rows_list:
        us, cus = p.k_reduce_cold(side, nbuf=nbuf, iters=max(32, 2 * nbuf), rows_per_wave=rows, copy_ceiling=(rows == rows_list[0]))
        print("[%s] side %5d nbuf %2d rows %2d: %7.2f us = %6.0f GB/s (%.3f of 8 TB/s)%s" % (
            tag, side, nbuf, rows, us, b / us / 1e3, b / us / 8e6, "   copy41 %.2f us = %.0f GB/s" % (cus, b / cus / 1e3) if cus else ""))
    sys.stdout.flush()
p.cleanup()
