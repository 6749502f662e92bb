#!/usr/bin/env python
"""Diagnostic: where do the cycles of the LDS-DMA GEMM kernels (gemm_bf16_v2_kernel) go?  Uses a library built with -DMCG_STAMPS
(MCG_LIB_PATH=.../lib_stamps.so MCG_HIPCC_FLAGS=-DMCG_STAMPS python mocogan-chainer_amd/build.py, run with the same MCG_LIB_PATH) and
prints, per D_V layer and pass at the given batch / tile code: a wave's average cycles in the block prologue (row decode, first
loads), per K-step in [waiting for its own LDS-DMA pieces] / [at the barrier] / [the step's body: fragment reads, MFMAs, LDS-DMA
issue], and in the epilogue -- next to the cycles the step's MFMAs alone need on a SIMD shared by two waves.  Stamps serialise the
schedule around them (a few per K-step): shares are meaningful, the build's run time is not.
usage: MCG_LIB_PATH=... python tools/stamp_phases_v2.py [--batch 512] [--tile 7] [--precision bf16s]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import mocogan_chainer_amd.hiplib as hl
import bench_layers as BL


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--tile', type=int, default=7)
    ap.add_argument('--precision', default='bf16s')
    ap.add_argument('--net', default='D_V')
    args = ap.parse_args()
    lib = hl.load()
    lib.mcg_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    buf = (ctypes.c_ulonglong * 8)()
    hl.set_tile_override(args.tile)
    print('tile code %d, %s, batch %d: average cycles per WAVE' % (args.tile, args.precision, args.batch))
    print('%-10s %-6s %9s | per K-step: %8s %8s %8s %8s | %9s %9s | %s' % ('layer', 'pass', 'prologue', 'vmcnt', 'barrier', 'body', 'total', 'epilogue',
                                                                        'kernel', 'K-steps  K-loop share'))
    dt = torch.bfloat16 if args.precision == 'bf16s' else torch.float32
    for name, N, T, H, Ci, Co, kt, ci_real in BL.layers(args.batch):
        if not name.startswith(args.net) or Ci < 64:
            continue
        g = hl.make_geom(N, T, H, H, Ci, Co, kt, precision=args.precision)
        x = torch.randn((N, T, H, H, Ci), device='cuda').to(dt)
        y = torch.randn((N, g.To, g.Ho, g.Wo, Co), device='cuda').to(dt)
        w = (torch.randn((Co, kt, 4, 4, Ci), device='cuda') * 0.05).to(dt)
        yo = torch.empty((N, g.To, g.Ho, g.Wo, Co), device='cuda')
        xo = torch.empty((N, T, H, H, Ci), device='cuda')
        dw = torch.zeros((Co, kt, 4, 4, Ci), device='cuda')
        for p, fn in (('fprop', lambda: hl.conv_fprop(g, x, w, None, yo)), ('dgrad', lambda: hl.conv_dgrad(g, y, w, None, xo)),
                      ('wgrad', lambda: hl.conv_wgrad(g, x, y, dw))):
            try:
                fn()
            except hl.McgError as exc:
                print('%-10s %-6s refused: %s' % (name, p, exc))
                continue
            torch.cuda.synchronize()
            lib.mcg_debug_stamps(buf, 1)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            lib.mcg_debug_stamps(buf, 1)
            waves, steps = max(buf[4], 1), max(buf[6], 1)
            if buf[4] == 0:
                print('%-10s %-6s (not an LDS-DMA GEMM launch: no stamps)' % (name, p))
                continue
            vm, bar, body = buf[0] / steps, buf[1] / steps, buf[2] / steps
            kern = buf[7] / waves
            print('%-10s %-6s %9.0f |             %8.0f %8.0f %8.0f %8.0f | %9.0f %9.0f | %7.1f  %5.1f %%' % (
                name, p, buf[3] / waves, vm, bar, body, vm + bar + body, buf[5] / waves, kern, steps / waves,
                100.0 * (buf[0] + buf[1] + buf[2]) / max(buf[7], 1)))


if __name__ == '__main__':
    main()
