#!/usr/bin/env python
"""Achieved bandwidth of the element-wise passes at the bf16 batch-256 step's largest tensor (D_V layer 2, one group of 256 clips:
655360 rows x 128 channels, bf16) next to a plain device copy of the same bytes (what the memory system gives a trivial kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mocogan_chainer_amd.hiplib as hl


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    hl.load()
    M, C = 256 * 10 * 16 * 16, 128
    for dt in (torch.bfloat16, torch.float32):
        es = 2 if dt == torch.bfloat16 else 4
        y = (torch.randn((M, C), device='cuda') * 1.3).to(dt)
        g = torch.randn((M, C), device='cuda').to(dt)
        gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
        ws = torch.empty(hl.bn_workspace_floats(1024), device='cuda')
        stats = torch.empty(4 * C, device='cuda')
        hl.bn_stats(M, C, y.float() if dt == torch.bfloat16 else y, gamma, beta, stats, None, None, ws)
        out = torch.empty_like(y)
        gx = torch.empty_like(y)
        dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
        nbytes = M * C * es
        rows = [
            ('copy (torch)', lambda: out.copy_(y), 2 * nbytes),
            ('bn_act_fwd + Philox noise', lambda: hl.bn_act_fwd(M, C, y, stats[2 * C:], hl.ACT_LRELU, out, sigma=0.2, seed=1, stream_id=5), 2 * nbytes),
            ('bn_act_fwd, no noise', lambda: hl.bn_act_fwd(M, C, y, stats[2 * C:], hl.ACT_LRELU, out), 2 * nbytes),
            ('bn_act_bwd (sums + apply)', lambda: hl.bn_act_bwd(M, C, g, y, stats, gamma, hl.ACT_LRELU, gx, dg, db, ws), 5 * nbytes),
        ]
        for name, fn, b in rows:
            ms = timeit(fn)
            print('%-8s %-28s %7.3f ms  %6.2f TB/s' % (str(dt).split('.')[1], name, ms, b / ms / 1e9))


if __name__ == '__main__':
    main()
