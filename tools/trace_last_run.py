"""Timeline of the last q3 run in a rocprofv3 kernel trace: kernels or gaps over 10 us, with totals."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'join_build' in r['Kernel_Name']]
start = max(0, idx[-2] - 14)
end = max(i for i, r in enumerate(rows) if 'gpuq' in r['Kernel_Name'])
run = rows[start:end + 1]
t0 = int(run[0]['Start_Timestamp']); prev = t0; ksum = 0; gsum = 0
for r in run:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d = (e - s) / 1e3; g = (s - prev) / 1e3
    ksum += d; gsum += max(g, 0)
    if d > 10 or g > 10:
        print("%9.1f us  gap %7.1f  dur %8.1f  grid=%-9s vgpr=%-3s %s" % ((s - t0) / 1e3, g, d, r.get('Grid_Size_X'), r.get('VGPR_Count'), r['Kernel_Name'][:70]))
    prev = e
print("kernels %.1f us, gaps %.1f us, total %.1f us, %d dispatches" % (ksum, gsum, (prev - t0) / 1e3, len(run)))
