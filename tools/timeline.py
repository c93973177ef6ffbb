"""Timeline of one pass from a rocprofv3 kernel_trace.csv: python tools/timeline.py <csv> [passes back from the end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
names = [r['Kernel_Name'].split('(')[0][:30] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith(('pd::screenKernel', 'pd::stage1Kernel'))]
a, b = idx[-back], idx[-back + 1]
t0 = int(rows[a]['Start_Timestamp'])
for i in range(a, b + 1):
    s, e = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
    print("%9.1f %8.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, names[i]))
