"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv)."""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', r['Kernel_Name'])
    name = name.replace('void mrl::gemm_kernel', 'gemm')
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[name][r['Counter_Name']] += 1
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for k in sorted(acc):
    if flt and flt not in k:
        continue
    print(k[:90])
    for c in sorted(acc[k]):
        print(f'    {c:32s} {acc[k][c] / cnt[k][c]:16.1f}  (n={cnt[k][c]})')
