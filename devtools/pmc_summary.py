"""Reduce the counter_collection.csv files of the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
devtools/pmc_target.py to HBM bytes per launch per kernel, corrected as MI355X_MICROARCH.md prescribes for
gfx950: both counters are in KiB; FETCH_SIZE reports half the bytes of wide streaming reads (x2); WRITE_SIZE
is exact. Usage: pmc_summary.py <fetch.csv> <write.csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    """{(kernel, grid): [values]} of the pipeline steps (the dispatches after the stand-alone `k_reduce_dma<2>` launches of
    pmc_target.py; everything before them is musica_create's autotune) and of the stand-alone launches themselves."""
    rows = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], int(row["Grid_Size"]), float(row["Counter_Value"])))
    rows.sort()
    standalone = ("k_reduce_dma<2>", "k_reduce_dma<4>", "k_copy41")
    last_standalone = max([d for d, n, g, v in rows if any(t in n for t in standalone)] or [0])
    acc = defaultdict(list)
    for d, n, g, v in rows:
        if d > last_standalone or any(t in n for t in standalone):
            acc[(n, g)].append(v)
    for key in acc:          # the first rotation of the stand-alone launches is the warm-up pass (code object, TLB)
        if any(t in key[0] for t in standalone) and len(acc[key]) > 16:
            acc[key] = acc[key][-16:]
    # musica_create's autotune may pick another rows-per-wavefront (hence grid) in another profiler pass: key the launches of
    # one kernel by the rank of their grid (largest = finest pyramid level) so that the two passes line up
    grids = defaultdict(set)
    for n, g in acc:
        grids[n].add(g)
    ranked = {}
    for (n, g), v in acc.items():
        ranked[(n, sorted(grids[n], reverse=True).index(g))] = (g, v)
    return ranked


def main():
    fetch, write, out = sys.argv[1:4]
    fe, wr = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    kernels = {}
    for key in sorted(set(fe) | set(wr)):
        name, rank = key
        if "musica" not in name:
            continue
        grid, f = fe.get(key, (0, []))
        wgrid, w = wr.get(key, (0, []))
        grid = grid or wgrid
        f2, w2 = f, w
        fb = 2.0 * 1024.0 * sum(f2) / max(len(f2), 1)
        wb = 1024.0 * sum(w2) / max(len(w2), 1)
        kernels["%s [grid %d]" % (name.split("(")[0], grid)] = {
            "launches": len(f), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
            "hbm_bytes_per_launch": round(fb + wb)}
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 devtools/pmc_target.py",
           "corrections": "KiB -> bytes (x1024); FETCH_SIZE x2 on gfx950 (wide streaming reads are tallied at half size); WRITE_SIZE exact",
           "kernels": kernels}
    # keys bench.py reads
    c4 = {}
    for k, v in kernels.items():
        if k.startswith("musica::k_reduce_u16_pf") or "k_reduce_band<true>" in k:   # the level-0 launch of the metric kernel (alone, or fused with the band)
            c4["reduce_l0_hbm_bytes_per_launch"] = v["hbm_bytes_per_launch"]
        if "k_reduce_dma<2>" in k:
            res["standalone_4096_warm_hbm_bytes_per_launch"] = v["hbm_bytes_per_launch"]
        if "k_reduce_dma<4>" in k:
            res["standalone_4096_cold_hbm_bytes_per_launch"] = v["hbm_bytes_per_launch"]
        if "k_copy41<0>" in k or k.startswith("musica::k_copy41 "):
            res["copy41_4096_cold_hbm_bytes_per_launch"] = v["hbm_bytes_per_launch"]
    res["C4"] = c4
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["C4"]), res.get("standalone_4096_cold_hbm_bytes_per_launch"), res.get("standalone_4096_warm_hbm_bytes_per_launch"))


if __name__ == "__main__":
    main()
