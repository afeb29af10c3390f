"""Dev tool: sweep the metric kernel's launch parameters on the GPU (not part of the product)."""
import os, sys, itertools, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp
    p = mp.MusicaProcessing(); assert p.init(64, levels=4)
    out = {}
    for side, batch in [(4096, 1), (2048, 8), (8192, 1)]:
        us = p.k_reduce_timed(side, batch=batch, iters=100)
        out["%dx%d" % (side, batch)] = (round(us, 2), round(5 * side * side * batch / us / 1e3, 0))
    print(json.dumps(out)); sys.exit(0)
combos = [(-1, 4), (-1, 6), (-1, 8), (-1, 10), (-1, 12), (-1, 16), (-1, 24), (-1, 32), (-2, 8), (-2, 16)]
for trip, rows in combos:
    env = dict(os.environ, MUSICA_REDUCE_TRIP=str(trip), MUSICA_REDUCE_ROWS=str(rows), MUSICA_MIN_WAVES="1")
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    print("trip", trip, "rows", rows, r.stdout.strip() or r.stderr[-300:], flush=True)
