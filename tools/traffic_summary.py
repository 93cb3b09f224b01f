"""Per-kernel HBM bytes per launch from the two rocprofv3 counter passes of tools/pmc_traffic.sh.

    python tools/traffic_summary.py gpurun_out/<tag>_FETCH_SIZE/p_counter_collection.csv \
                                    gpurun_out/<tag>_WRITE_SIZE/p_counter_collection.csv profiles/<name>.json

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes for wide coalesced
reads (16 B per lane, the access pattern of every kernel here), so it is doubled; WRITE_SIZE is exact.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        tot[name] += float(r['Counter_Value'])
        n[name] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {'unit': 'bytes per launch', 'fetch_correction': 'FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read undercount)', 'kernels': {}}
for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
    if not (k.startswith('mrl::') or 'kernel' in k) or k.startswith('at::'):
        continue
    f = fetch[k][0] * 1024 * 2
    w = write.get(k, (0.0, 0))[0] * 1024
    out['kernels'][k] = {'launches': fetch[k][1], 'fetch_bytes': f, 'write_bytes': w, 'hbm_bytes': f + w}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
for k, v in list(out['kernels'].items())[:14]:
    print(f"{v['hbm_bytes'] / 1e6:9.1f} MB/launch (read {v['fetch_bytes'] / 1e6:8.1f}, write {v['write_bytes'] / 1e6:8.1f}) x{v['launches']:5d}  {k}")
