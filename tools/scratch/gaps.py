import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
prev_end = None; out = []
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append((n, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
# print a window in the middle of a Lloyd loop
idx = [i for i, o in enumerate(out) if 'filter' in o[0]]
mid = idx[len(idx) // 2]
for o in out[mid - 3: mid + 9]:
    print(f"{o[0]:50s} dur {o[1]:8.1f} us   gap before {o[2]:7.1f} us")
import collections
g = collections.defaultdict(list)
for o in out: g[o[0]].append(o[2])
for k, v in g.items():
    if 'kmeans' in k: print(f"{k:50s} n={len(v):4d} mean gap before {sum(v)/len(v):7.2f} us")
