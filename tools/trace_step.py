"""Timeline of the last q3 step in a rocprofv3 kernel trace: EVERY dispatch with its gap to the previous one, then per-kernel totals.
usage: trace_step.py kernel_trace.csv [min_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'join_build' in r['Kernel_Name'] or 'probe_build' in r['Kernel_Name']]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2            # join builds per step
start = idx[-nb]
while start > 0 and (int(rows[start]['Start_Timestamp']) - int(rows[start - 1]['End_Timestamp'])) < 60000 and 'keyrange' not in rows[start]['Kernel_Name']:
    start -= 1
end = max(i for i, r in enumerate(rows) if 'gpuq' in r['Kernel_Name'] or 'k_' in r['Kernel_Name'])
run = rows[start:end + 1]
t0 = int(run[0]['Start_Timestamp']); prev = t0; ksum = 0; gsum = 0
tot = collections.OrderedDict()
for r in run:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d = (e - s) / 1e3; g = (s - prev) / 1e3
    ksum += d; gsum += max(g, 0)
    name = r['Kernel_Name'].split('(')[0][:60]
    a = tot.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += d
    if d >= thr or g >= thr:
        print("%9.1f us  gap %7.1f  dur %8.1f  grid=%-9s wg=%-4s %s" % ((s - t0) / 1e3, g, d, r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'), name))
    prev = max(prev, e)
print("kernels %.1f us, gaps %.1f us, total %.1f us, %d dispatches" % (ksum, gsum, (prev - t0) / 1e3, len(run)))
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %8.1f us  x%-3d %s" % (d, c, k))
