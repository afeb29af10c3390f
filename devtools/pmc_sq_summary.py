"""Reduce counter_collection.csv files of rocprofv3 --pmc SQ_* passes of devtools/pmc_target.py to per-kernel averages
and the ratios that say what a kernel is bound by. Usage: pmc_sq_summary.py <out.json> <pass1.csv> [<pass2.csv> ...]

Per dispatch (kernel name + grid size), averaged over the pipeline steps' launches:
  waves, valu / salu / lds / vmem instructions per wave, and — with P = the pixels the launch covers, taken from the grid —
  wave-instructions per 64 pixels; SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_ACTIVE_INST_VALU, SQ_WAIT_INST_ANY (quad-cycle units,
  MI355X_MICROARCH.md 'rocprofv3 PMC slots') as shares of SQ_WAVE_CYCLES."""
import csv
import json
import sys
from collections import defaultdict


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for path in paths:
        rows = []
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"]),
                             int(r["VGPR_Count"]), int(r["SGPR_Count"]), int(r["LDS_Block_Size"])))
        standalone = ("k_reduce_dma<2>", "k_reduce_dma<4>", "k_copy41")
        last = max([d for d, n, *_ in rows if any(t in n for t in standalone)] or [0])
        for d, n, g, cn, v, vg, sg, lds in rows:
            if "musica" not in n:
                continue
            if d > last or any(t in n for t in standalone):     # skip musica_create's autotune launches
                key = "%s [grid %d]" % (n.split("(")[0], g)
                acc[key][cn].append(v)
                acc[key]["_vgpr"] = [vg]
                acc[key]["_sgpr"] = [sg]
                acc[key]["_lds"] = [lds]
    res = {}
    for key, cs in sorted(acc.items()):
        m = {k: sum(v) / len(v) for k, v in cs.items()}
        e = {"launches": max(len(v) for k, v in cs.items() if not k.startswith("_")), "vgpr": int(m.pop("_vgpr")), "sgpr": int(m.pop("_sgpr")),
             "lds_bytes": int(m.pop("_lds"))}
        e.update({k: round(v, 1) for k, v in m.items()})
        w = m.get("SQ_WAVES")
        if w:
            for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
                if k in m:
                    e[k.lower()[3:] + "_per_wave"] = round(m[k] / w, 1)
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS"):
                if k in m:
                    e[k.lower()[3:] + "_share_of_wave_cycles"] = round(m[k] / wc, 4)
        res[key] = e
    json.dump({"source": "rocprofv3 --pmc SQ_* passes of devtools/pmc_target.py (eager launches; averages over the launches of 3 pipeline steps "
                         "of workload C4 and over the stand-alone launches)", "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves",
               "kernels": res}, open(out, "w"), indent=1)
    for k, e in res.items():
        print(k, {a: b for a, b in e.items() if a.endswith("per_wave") or a.endswith("cycles") or a in ("SQ_WAVES", "vgpr")})


if __name__ == "__main__":
    main()
