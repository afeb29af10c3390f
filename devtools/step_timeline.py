"""Print the kernel timeline of one steady-state pipeline step from a rocprofv3 kernel_trace.csv (bench.py run)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 14
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']),
             r['Kernel_Name'].split('(')[0].replace('void musica::', '').replace('musica::', ''), r.get('Queue_Id')) for r in rows)
clears = [i for i, k in enumerate(ks) if k[2].startswith('k_minmax_u16')]
i0, i1 = clears[which], clears[which + 1]
t0 = ks[i0][0]
for k in ks[i0:i1]:
    print("%8.1f %8.1f %7.1f  q%s %s" % ((k[0] - t0) / 1000, (k[1] - t0) / 1000, (k[1] - k[0]) / 1000, k[3], k[2]))
print('step', (ks[i1][0] - t0) / 1000, 'of', len(clears), 'steps')
