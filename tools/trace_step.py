"""Summarise one decode step from a rocprofv3 --kernel-trace CSV: per-kernel count, mean duration, gaps."""
import collections
import csv
import glob
import sys

f = sys.argv[1] if len(sys.argv) > 1 else glob.glob('gpurun_out/prof*/**/*kernel_trace.csv', recursive=True)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'advance_decode' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
step = rows[a:b]
span = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev = None
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void nvllm::', '').replace('nvllm::', '')[:44]
    name += f" g{int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"
    agg[name][0] += 1
    agg[name][1] += (e - s) / 1e3
    agg[name][2] += (s - prev) / 1e3 if prev else 0
    prev = e
print(f"{f}: kernels {len(step)} span {span:.1f} us, sum of durations {sum(v[1] for v in agg.values()):.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:60s} n={v[0]:4d} total={v[1]:8.1f} us avg={v[1]/v[0]:7.2f} gaps={v[2]:7.1f}")
