"""Summarise a rocprofv3 `--kernel-trace --stats --output-format csv` run: per-kernel calls, total, average, share."""
import csv
import re
import sys

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'# {path}: total kernel time {tot / 1e6:.2f} ms')
print(f"{'total_ms':>10} {'share':>6} {'calls':>6} {'avg_us':>9}  kernel")
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    name = re.sub(r'\(.*', '', r['Name'])[:100]
    t = float(r['TotalDurationNs'])
    print(f"{t / 1e6:10.2f} {100 * t / tot:5.1f}% {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f}  {name}")
