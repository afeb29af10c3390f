"""Window of a rocprofv3 kernel_trace.csv of `bench.py` with contexts in flight: which queue ran what, and how much of the
window had at least one / no kernel resident.  python devtools/inflight_timeline.py <kernel_trace.csv> [window_us]
The window starts at the middle k_minmax_u16 of the longest run of steps that alternate over queues (the timed steps)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 800.0
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']),
             r['Kernel_Name'].split('(')[0].replace('void musica::', '').replace('musica::', ''), r.get('Queue_Id')) for r in rows)
clears = [k for k in ks if k[2].startswith('k_minmax_u16')]
# the timed steps of the pipeline: the longest run of k_minmax_u16 launches in which consecutive ones sit on different queues
runs, i = [], 0
while i < len(clears) - 1:
    j = i
    while j < len(clears) - 1 and clears[j + 1][3] != clears[j][3]:
        j += 1
    runs.append((j - i, i, j))
    i = j + 1
n, a, b = max(runs)
if n < 6:
    raise SystemExit("no stretch with contexts in flight found")
t0 = clears[(a + b) // 2][0]
t1 = t0 + int(win * 1000)
sel = [k for k in ks if k[1] > t0 and k[0] < t1]
for k in sel:
    print("%8.1f %8.1f %7.1f  q%s %s" % ((k[0] - t0) / 1000, (k[1] - t0) / 1000, (k[1] - k[0]) / 1000, k[3], k[2]))
# union of the resident intervals
ev = sorted((max(k[0], t0), min(k[1], t1)) for k in sel)
busy, cur_a, cur_b = 0, None, None
for a, b in ev:
    if cur_b is None or a > cur_b:
        if cur_b is not None:
            busy += cur_b - cur_a
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
if cur_b is not None:
    busy += cur_b - cur_a
steps = sum(1 for c in clears if t0 <= c[0] < t1)
print("window %.0f us: %d steps started, some kernel resident %.1f %% of the time, sum of kernel durations %.0f us (%.2f x the window)"
      % (win, steps, 100.0 * busy / (t1 - t0), sum(min(k[1], t1) - max(k[0], t0) for k in sel) / 1000, sum(min(k[1], t1) - max(k[0], t0) for k in sel) / (t1 - t0)))
