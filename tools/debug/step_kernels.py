#!/usr/bin/env python3
"""Per-kernel time of ONE training step out of a rocprofv3 kernel trace (steps are delimited by the optimizer's
multi_tensor_apply launches).  usage: step_kernels.py <kernel_trace.csv> [top]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step ends with the optimizer: fused AdamW (kernel name mentions adam) or a burst of >= 30 multi_tensor_apply launches
idx = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name'].lower()]
burst = 1
if not idx:
    idx, burst = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name']], 30
cl, prev = [], -100
for i in idx:
    if i - prev > 50:
        cl.append([i, i])
    else:
        cl[-1][1] = i
    prev = i
cl = [c for c in cl if sum(1 for i in idx if c[0] <= i <= c[1]) >= burst]
a, b = cl[-3][1] + 1, cl[-2][1] + 1
step = rows[a:b]
print("step span %.2f ms, %d kernels" % ((int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6, len(step)))
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in step:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = r['Kernel_Name'][:100]
    agg[k][0] += d
    agg[k][1] += 1
    busy += d
print("sum of kernel time %.2f ms" % (busy / 1e6))
for k, (d, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:top]:
    print("%8.1f us %4d  %s" % (d / 1e3, n, k))
