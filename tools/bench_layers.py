#!/usr/bin/env python
"""Per-layer timing of the implicit-GEMM conv kernels at the step's shapes (tuning aid).
usage: python tools/bench_layers.py [--batch 32] [--tile T] [--only fprop|dgrad|wgrad]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mocogan_chainer_amd.hiplib as hl


def layers(B):
    out = []
    ch = [4, 64, 128, 256, 512]
    real = [3, 64, 128, 256, 512]
    t, h = 16, 64
    for l in range(4):                                   # D_V dc1..dc4
        out.append(('D_V.dc%d' % (l + 1), B, t, h, ch[l], ch[l + 1], 4, real[l]))
        t, h = t - 3, h // 2
    h = 64
    for l in range(4):                                   # D_I dc1..dc4
        out.append(('D_I.dc%d' % (l + 1), B, 1, h, ch[l], ch[l + 1], 1, real[l]))
        h //= 2
    gch = [512, 256, 128, 64, 4]
    greal = [512, 256, 128, 64, 3]
    for l in range(4):                                   # G dc2..dc5 in conv form: x side = output
        out.append(('G.dc%d' % (l + 2), 16 * B, 1, 8 << l, gch[l + 1], gch[l], 1, greal[l + 1]))
    return out


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
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--tile', type=int, default=0)
    ap.add_argument('--only', default='')
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16', 'bf16s', 'f32x3'])
    ap.add_argument('--net', default='', help='restrict to layers whose name starts with this (e.g. D_V)')
    ap.add_argument('--layer', default='', help='restrict to layers whose name contains this (e.g. dc1)')
    ap.add_argument('--autotune', action='store_true', help='time the tile candidates per geometry first')
    ap.add_argument('--save-tiles', default='', help='write the tuned choices to this JSON file')
    ap.add_argument('--tiles', default='', help='use the tile choices of this JSON file (no tuning launches)')
    args = ap.parse_args()
    hl.load()
    if args.tiles:
        hl.load_tile_choices(args.tiles)
    hl.set_autotune(args.autotune or bool(args.tiles))
    hl.set_tile_override(args.tile)
    print('%-10s %-6s %10s %10s %8s' % ('layer', 'pass', 'ms', 'TFLOP/s', 'GFLOP'))
    tot = {}
    for name, N, T, H, Ci, Co, kt, ci_real in layers(args.batch):
        if args.net and not name.startswith(args.net):
            continue
        if args.layer and args.layer not in name:
            continue
        prec = args.precision
        if prec == 'f32x3' and (Ci % 16 or Co % 16):
            prec = 'f32'                                    # the 4-channel layers of an f32x3 network run the fp32 kernels
        if prec == 'bf16s' and (Ci % 8 or Co % 8):
            prec = 'bf16y' if Co % 8 == 0 else 'bf16'       # ... of a bf16 network: the clip fp32, its 64-channel neighbour bf16 (round 4)
        g = hl.make_geom(N, T, H, H, Ci, Co, kt, precision=prec, ci_valid=ci_real)
        x = torch.randn((N, T, H, H, Ci), device='cuda')
        y = torch.randn((N, g.To, g.Ho, g.Wo, Co), device='cuda')
        w = torch.randn((Co, kt, 4, 4, Ci), device='cuda') * 0.05
        dw = torch.zeros_like(w)
        flops = 2.0 * N * g.To * g.Ho * g.Wo * kt * 16 * ci_real * Co
        if prec == 'bf16s':                                 # operands bf16 in memory, outputs fp32
            xi, yi, wi = x.to(torch.bfloat16), y.to(torch.bfloat16), w.to(torch.bfloat16)
        elif prec == 'f32x3':                               # fp32 values as three bf16 terms (the split of the operands is not timed)
            xi, yi, wi = hl.split_planes(x), hl.split_planes(y), hl.split_planes(w)
            wd = hl.split_planes(w, run=16 * kt * 16 * Ci)      # the filter as dgrad reads it: planes of 16 filters
        elif prec == 'bf16y':                               # y bf16 in memory beside the fp32 clip and filter (wgrad, dgrad read it)
            xi, yi, wi = x, y.to(torch.bfloat16), w
        else:
            xi, yi, wi = x, y, w
        passes = [('fprop', lambda: hl.conv_fprop(g, xi, wi, None, y)),
                  ('dgrad', lambda: hl.conv_dgrad(g, yi, wi, None, x)),
                  ('wgrad', lambda: hl.conv_wgrad(g, xi, yi, dw))]
        if prec == 'f32x3':
            passes = [passes[0], ('dgrad', lambda: hl.conv_dgrad(g, yi, wd, None, x)), passes[2]]
        for p, fn in passes:
            if args.only and p != args.only:
                continue
            try:
                ms = timeit(fn)
            except hl.McgError:                             # a tile code the layer does not admit (e.g. 7 / 8 on the Ci = 4 layers)
                print('%-10s %-6s %10s' % (name, p, 'n/a'))
                continue
            tot[p] = tot.get(p, 0) + ms
            print('%-10s %-6s %10.3f %10.1f %8.1f' % (name, p, ms, flops / ms / 1e9, flops / 1e9))
    print('totals (ms):', {k: round(v, 3) for k, v in tot.items()})
    if args.save_tiles:
        hl.save_tile_choices(args.save_tiles)


if __name__ == '__main__':
    main()
