"""Per-kernel, per-grid median durations of the steady-state pipeline steps in a rocprofv3 kernel_trace.csv (bench.py run with
--no-kernel-events --no-single-image --no-standalone): python devtools/trace_levels.py <trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void musica::', '').replace('musica::', ''),
             int(r.get('Grid_Size_X', 0) or 0) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0))) for r in rows)
clears = [i for i, k in enumerate(ks) if k[2].startswith('k_minmax_u16')]
acc = collections.defaultdict(list)
for a, b in zip(clears[5:-1], clears[6:]):          # skip autotune / warm-up steps
    for k in ks[a:b]:
        acc[(k[2], k[3])].append((k[1] - k[0]) / 1e3)
steps = [(ks[b][0] - ks[a][0]) / 1e3 for a, b in zip(clears[5:-1], clears[6:])]
steps.sort()
print("steps %d median %.1f us" % (len(steps), steps[len(steps) // 2]))
for (n, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-50s grid %9d n=%3d median %7.1f us  sum-share %.3f" % (n[:50], g, len(v), v[len(v) // 2], sum(v) / max(1e-9, sum(sum(x) for x in acc.values()))))
