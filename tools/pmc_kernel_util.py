#!/usr/bin/env python
"""Per-kernel hardware-counter summary of the conv GEMM kernels (fp32, bf16 register-staged, bf16 LDS-DMA) from rocprofv3
--pmc CSVs (one or several passes over the SAME command; every pass carries SQ_WAVE_CYCLES or at least the timestamps).
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
              SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU --output-format csv -d out1 -- python3 tools/bench_layers.py ...
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE ...
    python tools/pmc_kernel_util.py out1/*/*counter_collection.csv out2/*/*counter_collection.csv
Columns: util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs) (share of the matrix pipes' cycles at the nominal clock);
TF = MFMA MOPS x 512 FLOP / duration (executed matrix work); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked at s_waitcnt / barrier);
pipe = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (waiting to issue); valu = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; ldsc = SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE (conflict share of the LDS array's cycles); ldsu = SQ_LDS_IDX_ACTIVE / (4 x duration cycles x 256 CUs) ...
clk = GRBM_GUI_ACTIVE / 8 / duration (effective shader clock, GHz; guide: reads high below ~0.3 ms)."""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r'(gemm_kernel|gemm_bf16_kernel|gemm_bf16_v2_kernel)<\(anonymous namespace\)::(\w+)<(\d+), (\d+), (\d+), (\d+)', name)
    if m:
        kind = {'gemm_kernel': 'f32', 'gemm_bf16_kernel': 'bf16', 'gemm_bf16_v2_kernel': 'v2'}[m.group(1)]
        if kind == 'v2':
            kind = 'v2.f32' if m.group(6) == '4' else 'v2.bf16'
        epi = re.search(r'>, \d+, \d+, (\d+), (\d+)>', name)
        m5 = re.search(r'>, \d+, \d+, (\d+), (\d+), (\d+)>', name)          # ..., STAGES, EPI, SPLIT>
        if m5:
            epi = m5
            if m5.group(3) == '1':
                kind = 'v2.f32x3'
        return '%s %s %sx%sx%s%s' % (kind, m.group(2), m.group(3), m.group(4), m.group(5), (' e' + epi.group(2)) if epi and epi.group(2) != '0' else '')
    m = re.search(r'dgrad_patch_kernel<(\d), (\d)>', name)
    if m:
        return 'patch dgrad%s%s' % (' f32x3' if m.group(2) == '1' else '', (' e' + m.group(1)) if m.group(1) != '0' else '')
    m = re.search(r'(\w+_c4\w*_kernel)', name)
    return m.group(1) if m else None


def main(paths):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for path in paths:
        seen = set()
        for r in csv.DictReader(open(path)):
            k = short(r['Kernel_Name'])
            if not k:
                continue
            key = (k, r['Grid_Size'])
            by[key][r['Counter_Name']].append(float(r['Counter_Value']))
            did = (path, r['Dispatch_Id'])
            if did not in seen:
                seen.add(did)
                dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    hdr = ('kernel', 'grid', 'n', 'dur us', 'util', 'TFLOP/s', 'wait', 'pipe', 'valu', 'ldsc', 'clk')
    print('%-34s %9s %4s %9s %6s %8s %6s %6s %6s %6s %5s' % hdr)
    for key, c in sorted(by.items()):
        d = sum(dur[key]) / len(dur[key])                              # ns
        avg = lambda n: (sum(c[n]) / len(c[n])) if c.get(n) else None
        wc = avg('SQ_WAVE_CYCLES')
        mops = avg('SQ_INSTS_VALU_MFMA_MOPS_BF16') or avg('SQ_INSTS_VALU_MFMA_MOPS_F32')
        f = lambda v, s='%6.3f': (s % v) if v is not None else '     -'
        print('%-34s %9s %4d %9.1f %s %s %s %s %s %s %s' % (
            key[0], key[1], len(dur[key]), d / 1e3,
            f(avg('SQ_VALU_MFMA_BUSY_CYCLES') / (d * 2.4 * 1024) if avg('SQ_VALU_MFMA_BUSY_CYCLES') else None),
            f(mops * 512 / d / 1e3 if mops else None, '%8.1f'),
            f(avg('SQ_WAIT_ANY') / wc if wc and avg('SQ_WAIT_ANY') else None),
            f(avg('SQ_WAIT_INST_ANY') / wc if wc and avg('SQ_WAIT_INST_ANY') else None),
            f(avg('SQ_ACTIVE_INST_VALU') / wc if wc and avg('SQ_ACTIVE_INST_VALU') else None),
            f(avg('SQ_LDS_BANK_CONFLICT') / avg('SQ_LDS_IDX_ACTIVE') if avg('SQ_LDS_IDX_ACTIVE') else None),
            f(avg('GRBM_GUI_ACTIVE') / 8 / d if avg('GRBM_GUI_ACTIVE') else None, '%5.2f')))


if __name__ == '__main__':
    main(sys.argv[1:])
