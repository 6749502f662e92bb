"""Diagnostic: are two launches of the same convolution on the same inputs bit-identical?  (tests/test_gpu_guardband.py found fprop /
dgrad launches at tile 0 that are not.)  Repeats each launch several times into plain torch outputs and into a poisoned arena and
reports how many elements differ from the first launch, by how much, and where (row / column pattern)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import guard                           # noqa: E402
import mocogan_chainer_amd.hiplib as hl     # noqa: E402
import mocogan_chainer_amd.layout as lay    # noqa: E402


def main():
    hl.load()
    cases = [(2, 4, 8, 64, 160, 4), (2, 7, 16, 8, 64, 4), (3, 1, 16, 128, 64, 1), (5, 1, 4, 32, 256, 1)]
    tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3]
    arena = guard.Arena()
    for case in cases:
        N, Ti, H, Ci, Co, kt = case
        rng = np.random.RandomState(1)
        x = torch.tensor(rng.uniform(-1, 1, (N, Ci, Ti, H, H)), dtype=torch.float32, device="cuda")
        W = torch.tensor(rng.randn(Co, Ci, kt, 4, 4) * 0.1, dtype=torch.float32, device="cuda")
        gy = torch.tensor(rng.randn(N, Co, Ti - kt + 1, H // 2, H // 2), dtype=torch.float32, device="cuda")
        xd, wd, gyd = lay.act_to_dev(x), lay.conv_w_to_dev(W), lay.act_to_dev(gy)
        for prec in ("f32", "bf16"):
            for tile in tiles:
                g = hl.make_geom(N, Ti, H, H, Ci, Co, kt, precision=prec)
                g.tile = tile
                for where in ("plain", "arena", "arena-zeroed"):
                    for kind in ("fprop", "dgrad"):
                        outs = []
                        if where != "plain":
                            arena.reset()
                            xa, wa, ga = arena.put(xd), arena.put(wd), arena.put(gyd)
                        else:
                            xa, wa, ga = xd, wd, gyd
                        shape = (N, g.To, g.Ho, g.Wo, Co) if kind == "fprop" else (N, Ti, H, H, Ci)
                        try:
                            for rep in range(4):
                                if where == "plain":
                                    out = torch.empty(shape, device="cuda")
                                elif where == "arena":
                                    out = arena.empty(shape)
                                else:
                                    out = arena.zeros(shape)
                                if kind == "fprop":
                                    hl.conv_fprop(g, xa, wa, None, out)
                                else:
                                    hl.conv_dgrad(g, ga, wa, None, out)
                                torch.cuda.synchronize()
                                outs.append(out)
                        except hl.McgError:
                            continue
                        msgs = []
                        for rep in range(1, 4):
                            d = (outs[rep] != outs[0])
                            if bool(d.any()):
                                idx = d.nonzero()
                                diff = (outs[rep] - outs[0]).abs().max().item()
                                rows = idx[:, :-1]
                                msgs.append("rep%d: %d elems differ (max %.3e), first idx %s, cols %s" % (
                                    rep, idx.shape[0], diff, idx[0].tolist(), sorted(set(idx[:, -1].tolist()))[:12]))
                        if msgs:
                            print(case, prec, "tile", tile, where, kind, "|", " ; ".join(msgs), flush=True)
    print("diag_repeat done")


if __name__ == "__main__":
    main()
