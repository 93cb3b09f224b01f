"""Condense hipcc -Rpass-analysis=kernel-resource-usage remarks (stdin) into one line per kernel."""
import re
import subprocess
import sys

rows, cur = [], None
for line in sys.stdin:
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
try:
    names = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.split('\n')
except Exception:
    names = [r['name'] for r in rows]
print(f"{'VGPR':>5} {'AGPR':>5} {'spill':>5} {'occ':>3} {'LDS':>7}  kernel")
for r, n in zip(rows, names):
    n = re.sub(r'mrl::gemm_kernel<(.*?)>\(.*', r'gemm<\1>', n)
    n = re.sub(r'\(.*', '', n)
    print(f"{r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('VGPRs Spill','?'):>5} {r.get('Occupancy [waves/SIMD]','?'):>3} {r.get('LDS Size [bytes/block]','?'):>7}  {n}")
