"""Print the per-kernel table of one or more bench.py result lines: python tools/bench_kernels.py a.json [b.json ...]"""
import json
import sys

for path in sys.argv[1:]:
    b = json.load(open(path))
    print(f"{path}: {b['ms_per_step']:.2f} ms per update (median {b.get('ms_per_step_median', 0):.2f}), "
          + (f"minibatch {b['secondary']['minibatch']}: {b['secondary']['ms_per_update']:.2f} ms, " if 'secondary' in b else '') +
          f"frac {b['roofline']['frac']:.3f}, update_frac {b['roofline'].get('update_frac', 0):.3f}")
    for k in b['roofline']['kernels']:
        print(f"  {k['name'][:62]:62s} {k['launches']:5d} x {k['avg_us']:7.1f} us = {k['total_ms']:6.2f} ms")
