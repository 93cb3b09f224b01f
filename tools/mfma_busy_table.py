"""Per-kernel matrix-pipe busy fraction from one rocprofv3 --pmc pass (tools/pmc_pass.sh with SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16).

    python tools/mfma_busy_table.py gpurun_out/<tag>/.../p_counter_collection.csv

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1,024 SIMDs); averages per launch, kernels by total active cycles."""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('mrl::gemm_kernel', 'gemm').replace('mrl::', '')
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[name][r['Counter_Name']] += 1
rows = []
for k in acc:
    a = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    gui = a.get('GRBM_GUI_ACTIVE', 0.0)
    if gui <= 0 or k.startswith('at::'):
        continue
    n = cnt[k]['GRBM_GUI_ACTIVE']
    rows.append((gui * n, k, a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui / 8 * 1024), a.get('SQ_INSTS_VALU', 0.0), a.get('SQ_LDS_BANK_CONFLICT', 0.0),
                 gui, a.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0), n))
print(f"{'kernel':62s} {'mfma_busy':>9s} {'insts_valu':>12s} {'lds_confl':>10s} {'gui_active':>12s} {'mfma_mops_bf16':>14s} {'launches':>8s}")
for _, k, busy, iv, lc, gui, mops, n in sorted(rows, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f'{k[:62]:62s} {busy:9.3f} {iv:12.0f} {lc:10.0f} {gui:12.0f} {mops:14.0f} {n:8d}')
