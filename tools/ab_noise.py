#!/usr/bin/env python
"""Round 5's review, item 1c, with numbers: D's first layer forward with the noise of layer 2's input (a) drawn in the GEMM epilogue
(Philox4x32-10 + Box-Muller per row quad and channel: what ships) against (b) LOADED by the epilogue from a tensor a separate kernel
produced (the `addend` form of mcg_conv_epilogue -- fp32 here; a bf16 tensor would halve its bytes), plus (c) that producer alone
(mcg_randn_rowquad: the same stream, what a side stream would run) and (d) the plain store without activation / noise.
    python tools/ab_noise.py [--clips 64] [--precision f32|bf16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mocogan_chainer_amd.hiplib as hl


def timeit(fn, iters=20):
    for _ in range(3):
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
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=64)
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16'])
    args = ap.parse_args()
    hl.load()
    N, T, H, Co = args.clips, 16, 64, 64
    g = hl.make_geom(N, T, H, H, 4, Co, 4, precision=args.precision, ci_valid=3)
    x = torch.rand((N, T, H, H, 4), device='cuda') * 2 - 1
    x[..., 3] = 0
    w = torch.randn((Co, 4, 4, 4, 4), device='cuda') * 0.05
    w[..., 3] = 0
    b = torch.zeros(Co, device='cuda')
    M = N * g.To * g.Ho * g.Wo
    odt = torch.bfloat16 if args.precision == 'bf16' else torch.float32
    y = torch.empty((N, g.To, g.Ho, g.Wo, Co), device='cuda', dtype=odt)
    mask = torch.zeros((M, 2), dtype=torch.int32, device='cuda')
    noise = torch.empty((M, Co), device='cuda')
    o16 = odt == torch.bfloat16
    res = {}
    res['plain store'] = timeit(lambda: hl.conv_fprop(g, x, w, b, y))
    res['act + Philox in the epilogue'] = timeit(lambda: hl.conv_fprop(
        g, x, w, b, y, ep=hl.epilogue(act=hl.ACT_LRELU, sigma=0.2, seed=1, stream_id=[3], mask_out=mask, out_bf16=o16), must_fuse=True))
    res['act + noise LOADED by the epilogue (fp32 tensor)'] = timeit(lambda: hl.conv_fprop(
        g, x, w, b, y, ep=hl.epilogue(act=hl.ACT_LRELU, addend=[noise.view(N, g.To, g.Ho, g.Wo, Co)], mask_out=mask, out_bf16=o16), must_fuse=True))
    res['the noise producer alone (mcg_randn_rowquad)'] = timeit(lambda: hl.randn_rowquad(noise, Co, 0.2, 1, 3))
    print('D first layer forward, %d clips, %s networks: %d M outputs' % (N, args.precision, M * Co // 1000000))
    for k, v in res.items():
        print('  %-52s %.3f ms' % (k, v))


if __name__ == '__main__':
    main()
