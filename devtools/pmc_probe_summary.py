"""Per-kernel means of a rocprofv3 `--pmc ... --kernel-trace` run of devtools/linear_probe.py: counters, duration and the shader clock
(GRBM_GUI_ACTIVE / 8 XCDs / duration). Usage: pmc_probe_summary.py <dir> <kernel name substring> [grid]"""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
acc = collections.defaultdict(list)
ids = set()
for r in csv.DictReader(open(cc)):
    if pat in r["Kernel_Name"] and (not grid or int(r["Grid_Size"]) == grid):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        ids.add(int(r["Dispatch_Id"]))
ids = sorted(ids)[len(ids) // 2:]   # steady-state half
m = {k: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for k, v in acc.items()}
ds = sorted(dur[i] for i in ids if i in dur)
med = ds[len(ds) // 2] if ds else float("nan")
print("%s: %d launches, median %.1f us" % (pat, len(ids), med))
for k, v in sorted(m.items()):
    print("  %-24s %14.0f" % (k, v))
if "GRBM_GUI_ACTIVE" in m and ds:
    print("  shader clock ~ %.2f GHz" % (m["GRBM_GUI_ACTIVE"] / 8 / med / 1e3))
if "SQ_WAVE_CYCLES" in m and "SQ_WAVES" in m and ds:
    print("  wave lifetime %.0f quad-cycles = %.2f GHz x median / 4" % (m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"], m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"] * 4 / med / 1e3))
for a in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
    if a in m and "SQ_WAVE_CYCLES" in m:
        print("  %-24s / SQ_WAVE_CYCLES = %.3f" % (a, m[a] / m["SQ_WAVE_CYCLES"]))
