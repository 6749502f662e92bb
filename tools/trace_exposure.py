#!/usr/bin/env python
"""What runs alone in an iteration?  Reads a rocprofv3 --kernel-trace CSV of bench.py (side streams on) and, for the last iterations
(delimited by the third adam_wd launch of each), reports the time with only GEMM kernels running, only element-wise kernels, both, and
nothing; then the element-wise kernels ranked by the time they run with NO GEMM beside them (the exposed part a schedule could hide).
usage: python tools/trace_exposure.py kernel_trace.csv[.gz] [iterations = 4]"""
import collections
import csv
import gzip
import re
import sys


def is_gemm(n):
    return any(k in n for k in ('gemm_', 'dgrad_patch', 'fprop_c4', 'dgrad_c4', 'wgrad_c4', 'fc_'))


def main():
    path = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    f = gzip.open(path, 'rt') if path.endswith('.gz') else open(path)
    rows = list(csv.DictReader(f))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    adam = [i for i, r in enumerate(rows) if 'adam_wd' in r['Kernel_Name']]
    ends = [rows[adam[i]]['e'] for i in range(2, len(adam), 3)]
    spans = list(zip(ends[:-1], ends[1:]))[-iters:]
    exposed_by = collections.Counter()
    count = collections.Counter()
    for a, b in spans:
        ks = [r for r in rows if r['s'] >= a and r['e'] <= b]
        ev = []
        for r in ks:
            g = is_gemm(r['Kernel_Name'])
            ev += [(r['s'], 1, g), (r['e'], -1, g)]
        ev.sort()
        t_prev, ng, ne = a, 0, 0
        acc = collections.Counter()
        for t, d, g in ev:
            dt = t - t_prev
            acc['both' if ng and ne else 'gemm' if ng else 'elementwise' if ne else 'idle'] += dt
            if ng >= 2:
                acc['gemm2'] += dt
            if g:
                ng += d
            else:
                ne += d
            t_prev = t
        acc['idle'] += b - t_prev
        print('iteration %.3f ms: only GEMMs %.2f (two or more of them %.2f), only element-wise %.2f, both %.2f, idle %.2f | sum of GEMM '
              'durations %.2f, of element-wise %.2f' % ((b - a) / 1e6, acc['gemm'] / 1e6, acc['gemm2'] / 1e6, acc['elementwise'] / 1e6,
                                                        acc['both'] / 1e6, acc['idle'] / 1e6,
                                                        sum(r['e'] - r['s'] for r in ks if is_gemm(r['Kernel_Name'])) / 1e6,
                                                        sum(r['e'] - r['s'] for r in ks if not is_gemm(r['Kernel_Name'])) / 1e6))
        merged = []
        for s, e in sorted((r['s'], r['e']) for r in ks if is_gemm(r['Kernel_Name'])):
            if merged and s <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], e)
            else:
                merged.append([s, e])
        for r in ks:
            if is_gemm(r['Kernel_Name']):
                continue
            t = r['e'] - r['s']
            for ms, me in merged:
                lo, hi = max(r['s'], ms), min(r['e'], me)
                if hi > lo:
                    t -= hi - lo
            n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            n = re.sub(r'^void ', '', n).split('(')[0][:56]
            exposed_by[n] += t
            count[n] += 1
    print('element-wise kernels by exposed time (ms per iteration, launches per iteration):')
    for n, t in exposed_by.most_common(14):
        print('  %-58s %6.3f  %5.1f' % (n, t / 1e6 / len(spans), count[n] / len(spans)))


if __name__ == '__main__':
    main()
