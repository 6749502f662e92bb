#!/usr/bin/env python
"""Does the gradient exchange overlap the backward pass?  Reads a rocprofv3 --kernel-trace CSV of
    MCG_DP_REHEARSE_NCCL=1 python3 bench.py --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --secondary 0
(one rank over the real nccl = RCCL backend: the product's whole exchange path with a world of one) and, for every RCCL kernel of the last
iterations, prints when it ran, on which queue, and which of the iteration's kernels ran at the same time on OTHER queues -- the late
bucket's all-reduce is issued from the weight-gradient stream as soon as its last wgrad is queued and should run beside the remaining
weight / input gradient GEMMs (step.GradExchange, nets._Net.backward(on_late_bucket=...)).
usage: python tools/trace_dp_overlap.py kernel_trace.csv [iterations = 2]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    m = re.match(r'(\w+)<(\w+)<([\d, a-z]+)>', n)
    if m:
        return '%s<%s<%s>>' % (m.group(1), m.group(2), ','.join(m.group(3).split(',')[:3]).replace(' ', ''))
    return n.split('(')[0][:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    adam = [i for i, r in enumerate(rows) if 'adam_wd' in r['Kernel_Name']]
    ends = [rows[adam[i]]['e'] for i in range(2, len(adam), 3)]
    spans = list(zip(ends[:-1], ends[1:]))[-iters:]
    is_rccl = lambda n: any(k in n for k in ('nccl', 'Nccl', 'rccl', 'Rccl', 'AllReduce', 'ncclDevKernel'))
    names = sorted({short(r['Kernel_Name']) for r in rows if is_rccl(r['Kernel_Name'])})
    print('RCCL kernels in the trace: %s' % (names or 'NONE (a world of one: the collective is a no-op on the device, no ring kernel is launched)'))
    copies = [r for r in rows if 'copyBuffer' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name']]
    print('runtime copy / fill kernels: %d' % len(copies))
    for a, b in spans:
        ks = [r for r in rows if r['s'] >= a and r['e'] <= b]
        print('iteration of %.3f ms, %d kernels, %d of them RCCL' % ((b - a) / 1e6, len(ks), sum(is_rccl(r['Kernel_Name']) for r in ks)))
        tot = ov = 0
        for r in ks:
            if not is_rccl(r['Kernel_Name']):
                continue
            others = [(o, min(o['e'], r['e']) - max(o['s'], r['s'])) for o in ks
                      if o is not r and not is_rccl(o['Kernel_Name']) and o['s'] < r['e'] and o['e'] > r['s']]
            covered = 0
            ev = sorted([(max(o['s'], r['s']), 1) for o, _ in others] + [(min(o['e'], r['e']), -1) for o, _ in others])
            depth, prev = 0, r['s']
            for t, d in ev:
                if depth > 0:
                    covered += t - prev
                depth += d
                prev = t
            tot += r['e'] - r['s']
            ov += covered
            print('  RCCL kernel %-40s queue %-4s  +%8.3f ms .. +%8.3f ms (%.3f ms), %.0f %% of it beside other kernels:' % (
                short(r['Kernel_Name']), r.get('Queue_Id', '?'), (r['s'] - a) / 1e6, (r['e'] - a) / 1e6, (r['e'] - r['s']) / 1e6,
                100.0 * covered / max(r['e'] - r['s'], 1)))
            for o, t in sorted(others, key=lambda x: -x[1])[:4]:
                print('      %-56s queue %-4s overlap %.3f ms' % (short(o['Kernel_Name']), o.get('Queue_Id', '?'), t / 1e6))
        print('  RCCL kernel time %.3f ms, of which %.3f ms (%.0f %%) ran beside compute kernels' % (tot / 1e6, ov / 1e6, 100.0 * ov / max(tot, 1)))


if __name__ == '__main__':
    main()
