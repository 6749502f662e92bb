#!/usr/bin/env python
"""MFMA-pipe utilisation of the conv GEMM kernels from hardware counters, independent of any FLOP bookkeeping:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
              SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU --output-format csv -d out -- python3 tools/bench_layers.py --net D_V --batch 64 --autotune
    python tools/pmc_mfma_util.py out/*/*_counter_collection.csv
util  = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs)
TF    = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 FLOP / launch duration      (executed MFMA work; 157.3 = peak)
wait  = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked at s_waitcnt / barrier); pipe = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (waiting to issue,
        i.e. mostly for the matrix pipe); valu = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES."""
import collections
import csv
import re
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in rows:
        m = re.search(r'gemm_kernel<\(anonymous namespace\)::(\w+)<(\d+), (\d+), (\d+)>', r['Kernel_Name'])
        if not m:
            continue
        key = (m.group(1), '%sx%sx%s' % m.group(2, 3, 4), r['Grid_Size'])
        by[key][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    print('%-8s %-11s %10s %8s %9s %7s %8s %7s %7s %7s' % ('policy', 'tile', 'grid', 'launches', 'dur us', 'util', 'TFLOP/s', 'wait', 'pipe', 'valu'))
    for key, c in sorted(by.items()):
        n = len(c['SQ_VALU_MFMA_BUSY_CYCLES'])
        d = sum(dur[key]) / len(dur[key])
        wc = sum(c['SQ_WAVE_CYCLES'])
        print('%-8s %-11s %10s %8d %9.1f %7.3f %8.1f %7.3f %7.3f %7.3f' % (
            key[0], key[1], key[2], n, d / 1e3, sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / n / (d * 2.4 * 1024),
            sum(c['SQ_INSTS_VALU_MFMA_MOPS_F32']) / n * 512 / d / 1e3, sum(c['SQ_WAIT_ANY']) / wc, sum(c['SQ_WAIT_INST_ANY']) / wc,
            sum(c['SQ_ACTIVE_INST_VALU']) / wc))


if __name__ == '__main__':
    main(sys.argv[1])
