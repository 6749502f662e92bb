#!/usr/bin/env python
"""Offline refinement of the tile table INSIDE the iteration: the per-geometry choices of hiplib's tuner come from
isolated timings on scratch tensors, which sometimes mis-rank candidates (cache state and neighbours in the queue
differ).  This tool does coordinate descent on the whole-iteration time: for every (pass, geometry) of one training
step, most expensive first, it tries each candidate with all other choices fixed and keeps a candidate only if the
iteration gets measurably faster.  Writes the resulting table (the shipped tuned_tiles_mi355x.json was made this way).
    python tools/tune_table_in_step.py --out gpurun_out/tiles_refined.json [--batch 32] [--sweeps 1]
Caveat (round 3): the iteration time DRIFTS by 1-3 % over a run of this tool (clock / thermal state), which the confirm step does
not always catch -- switches it reports on keys the iteration does not even use (the fp32-form key of a launch that runs in its
split form) are that drift.  Confirm a refined table with alternating `bench.py --tiles` runs before adopting it."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mocogan_chainer_amd.hiplib as hl            # noqa: E402
import mocogan_chainer_amd.step as mstep           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--sweeps', type=int, default=1)
    ap.add_argument('--min-gain', type=float, default=0.002, help='relative iteration-time gain needed to switch')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'f32x3'])
    ap.add_argument('--cands', default='', help='comma-separated tile codes to try INSTEAD of the usual candidate lists (e.g. 10)')
    ap.add_argument('--overlap', type=int, default=1)
    ap.add_argument('--out', required=True)
    args = ap.parse_args()
    hl.load()
    hl.set_autotune(True)
    gen, di, dv = mstep.make_models('normal', num_labels=6, seed=0)
    ts = mstep.TrainStep('normal', gen, di, dv, seed=1234, overlap=bool(args.overlap), precision=args.dtype)
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    x = torch.rand((args.batch, 3, 16, 64, 64), device='cuda', generator=g) * 2 - 1
    t = torch.randint(0, 6, (args.batch,), device='cuda', dtype=torch.int32, generator=g)

    def measure():
        best = 1e9
        for _ in range(args.reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(args.steps):
                ts.run(x, t)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / args.steps)
        return best * 1e3

    for _ in range(6):
        ts.run(x, t)                                    # every geometry gets an entry (pre-tuned or tuned now)
    cache = hl._tile_cache
    # per-key cost of the current choice: one-stream pass with event timing
    ts.set_overlap(False)
    hl.timing_begin()
    for _ in range(4):
        ts.run(x, t)
    timing = hl.timing_end()
    ts.set_overlap(bool(args.overlap))
    for _ in range(3):
        ts.run(x, t)

    def key_cost(k):
        kind = k[0][6:] if str(k[0]).startswith('split-') else k[0]      # ('split-<pass>': which form of an f32x3 launch, 0 / 1)
        for name, (n_l, ms) in timing.items():
            if ' N=' in name and name.split('.')[1].split()[0] == kind and ('N=%d T=%d H=%d Ci=%d Co=%d' % (k[1], k[2], k[3], k[5], k[6])) in name:
                return ms
        return 0.0
    precs = (hl.PRECISIONS['bf16'], hl.PRECISIONS['bf16s'], hl.PRECISIONS['bf16y']) if args.dtype == 'bf16' else (hl.PRECISIONS['f32'], hl.PRECISIONS['f32x3']) if args.dtype == 'f32x3' else (hl.PRECISIONS[args.dtype],)   # bf16 networks: both operand forms
    keys = sorted([k for k in cache if k[9] in precs and k[1] in (args.batch, 2 * args.batch, 16 * args.batch)], key=key_cost, reverse=True)
    base = measure()
    print('baseline %.3f ms/iteration, %d geometries' % (base, len(keys)), flush=True)
    for sweep in range(args.sweeps):
        for k in keys:
            if key_cost(k) < 0.02 * 4:                  # < 0.02 ms per iteration: not worth a trial
                continue
            cands = list(hl.TILE_CANDIDATES)
            if k[0] in ('fprop', 'dgrad') and k[5] > 4:
                cands += list(hl.FPROP_SPLIT_CANDIDATES)
            if (k[0] == 'dgrad' and 4 < k[5] <= 64) or (k[0] == 'fprop' and k[6] <= 64 and k[5] > 4):
                cands += [4]                                 # 256 x 64: the widest tile a 64-column output admits
            if k[0] == 'wgrad':                             # more / fewer pixel splits around the current tile
                cands += [1000 + c for c in hl.TILE_CANDIDATES if c] + [2000 + c for c in hl.TILE_CANDIDATES if c]
            if args.cands:
                cands = [int(c) for c in args.cands.split(',')]
            if str(k[0]).startswith('split-'):
                cands = [0, 1]                                   # the other operand form of the launch
            cur = cache[k]
            best_c, best_t = cur, base
            for c in cands:
                if c == cur:
                    continue
                cache[k] = c
                try:
                    ts.run(x, t)
                except hl.McgError:
                    continue
                tm = measure()
                if tm < best_t * (1 - args.min_gain):
                    # confirm against a fresh measurement of the incumbent before switching
                    cache[k] = best_c
                    ts.run(x, t)
                    ref = measure()
                    cache[k] = c
                    ts.run(x, t)
                    tm2 = measure()
                    if tm2 < ref * (1 - args.min_gain):
                        best_c, best_t = c, tm2
            cache[k] = best_c
            base = best_t
            print('%-60s %5d -> %5d   %.3f ms' % (str(k[:8]), cur, best_c, base), flush=True)
        hl.save_tile_choices(args.out)
    print('final %.3f ms/iteration -> %s' % (measure(), args.out))


if __name__ == '__main__':
    main()
