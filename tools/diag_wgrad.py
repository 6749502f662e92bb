"""Diagnostic for the round-3 driver failure (tests/test_gpu_ops.py::test_bf16_stored_operands_equal_rounding_in_the_kernel[0-case2]):
the weight gradient at (N2,T4,H8,Ci64,Co160,kt4) of the 'bf16' (fp32 tensors rounded in the kernel) and 'bf16s' (bf16 tensors) launches,
repeated with plain torch allocations and inside a poisoned arena (tests/guard.py), each result against the float64 oracle.
Prints which launch deviates, where (co, kt, kh, kw, ci) and by how much."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import functions as F      # noqa: E402  (diagnostic tool: the oracle is the checker)
import guard                           # noqa: E402
import mocogan_chainer_amd.hiplib as hl     # noqa: E402
import mocogan_chainer_amd.layout as lay    # noqa: E402


def bf16_round(a):
    return torch.tensor(np.asarray(a, np.float32)).to(torch.bfloat16).double().numpy()


def testbody(reps):
    """the body of the red test, repeated: fprop / dgrad / wgrad at tile 0 with 'bf16' then 'bf16s' operands in plain torch allocations;
    the first weight gradient is copied out BEFORE the second set of launches and compared again after them (a stray store from a
    later launch would show as a change), both against the float64 oracle."""
    hl.load()
    case = (2, 4, 8, 64, 160, 4)
    N, Ti, H, Ci, Co, kt = case
    rng = np.random.RandomState(hash(case) % 2**31 + 5)
    x, W = bf16_round(rng.uniform(-1, 1, (N, Ci, Ti, H, H))), bf16_round(rng.randn(Co, Ci, kt, 4, 4) * 0.1)
    b = rng.randn(Co)
    gy = bf16_round(rng.randn(N, Co, Ti - kt + 1, H // 2, H // 2))
    _, gW_ref, _ = F.conv3d_bwd(x, W, gy, (1, 2, 2), (0, 1, 1))
    dev = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    ref_dev = lay.conv_w_to_dev(dev(gW_ref)).cpu().double().numpy()
    bad = 0
    for rep in range(reps):
        xd, wd, bd, gyd = lay.act_to_dev(dev(x)), lay.conv_w_to_dev(dev(W)), dev(b), lay.act_to_dev(dev(gy))
        x16, w16, gy16 = xd.to(torch.bfloat16), wd.to(torch.bfloat16), gyd.to(torch.bfloat16)
        res, early = {}, {}
        for prec, (xa, wa, ga) in (('bf16', (xd, wd, gyd)), ('bf16s', (x16, w16, gy16))):
            g = hl.make_geom(N, Ti, H, H, Ci, Co, kt, precision=prec)
            yd = torch.empty((N, g.To, g.Ho, g.Wo, Co), device="cuda")
            hl.conv_fprop(g, xa, wa, bd, yd)
            gxd = torch.full((N, Ti, H, H, Ci), 7.0, device="cuda")
            hl.conv_dgrad(g, ga, wa, None, gxd)
            dwd = torch.zeros_like(wd)
            hl.conv_wgrad(g, xa, ga, dwd)
            res[prec] = dwd
            early[prec] = dwd.cpu().double().numpy()
        for prec in ('bf16', 'bf16s'):
            late = res[prec].cpu().double().numpy()
            for tag, got in (('right after its launch', early[prec]), ('at the end', late)):
                rel = np.linalg.norm(got - ref_dev) / np.linalg.norm(ref_dev)
                if not np.isfinite(rel) or rel > 1e-6:
                    bad += 1
                    idx = np.argwhere(~(np.abs(got - ref_dev) <= 1e-5 * np.abs(ref_dev).max()))
                    print("rep %d %s %s: rel-L2 %.3e, %d elements off; first: %s" % (
                        rep, prec, tag, rel, len(idx), [(tuple(int(v) for v in i), float(got[tuple(i)]), float(ref_dev[tuple(i)])) for i in idx[:6]]), flush=True)
    print("diag_wgrad testbody: %d deviating results of %d" % (bad, reps * 4))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'testbody':
        return testbody(int(sys.argv[2]) if len(sys.argv) > 2 else 200)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    hl.load()
    case = (2, 4, 8, 64, 160, 4)
    N, Ti, H, Ci, Co, kt = case
    rng = np.random.RandomState(hash(case) % 2**31 + 5)
    x, W = bf16_round(rng.uniform(-1, 1, (N, Ci, Ti, H, H))), bf16_round(rng.randn(Co, Ci, kt, 4, 4) * 0.1)
    gy = bf16_round(rng.randn(N, Co, Ti - kt + 1, H // 2, H // 2))
    _, gW_ref, _ = F.conv3d_bwd(x, W, gy, (1, 2, 2), (0, 1, 1))
    dev = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    xd, wd, gyd = lay.act_to_dev(dev(x)), lay.conv_w_to_dev(dev(W)), lay.act_to_dev(dev(gy))
    ref_dev = lay.conv_w_to_dev(dev(gW_ref)).cpu().double().numpy()
    scale = np.abs(ref_dev).max()
    arena = guard.Arena()
    bad = 0
    junk = []
    for rep in range(reps):
        for mode in ("plain", "arena"):
            for tile in tiles:
                for prec in ("bf16", "bf16s"):
                    if mode == "arena":
                        arena.reset()
                        dt = torch.float32 if prec == "bf16" else torch.bfloat16
                        xa, ga = arena.put(xd.to(dt)), arena.put(gyd.to(dt))
                        dwd = arena.zeros(tuple(wd.shape))
                    else:
                        junk.append(torch.full((int(rng.randint(1, 300000)),), 1e30, device="cuda"))   # perturb the allocator's neighbours
                        if len(junk) > 8:
                            junk.pop(0)
                        dt = torch.float32 if prec == "bf16" else torch.bfloat16
                        xa, ga = xd.to(dt).clone(), gyd.to(dt).clone()
                        dwd = torch.zeros_like(wd)
                    g = hl.make_geom(N, Ti, H, H, Ci, Co, kt, precision=prec)
                    g.tile = tile
                    hl.conv_wgrad(g, xa, ga, dwd)
                    torch.cuda.synchronize()
                    if mode == "arena":
                        arena.check()
                    got = dwd.cpu().double().numpy()
                    err = np.abs(got - ref_dev)
                    rel = np.linalg.norm(got - ref_dev) / np.linalg.norm(ref_dev)
                    if not np.isfinite(rel) or rel > 1e-6:
                        bad += 1
                        idx = np.argwhere(~(err <= 1e-5 * scale))
                        print("rep %d %s tile %d %s: rel-L2 %.3e, %d elements off; first: %s" % (
                            rep, mode, tile, prec, rel, len(idx),
                            [(tuple(int(v) for v in i), float(got[tuple(i)]), float(ref_dev[tuple(i)])) for i in idx[:6]]), flush=True)
    print("diag_wgrad: %d deviating launches of %d" % (bad, reps * 2 * 2 * len(tiles)))


if __name__ == "__main__":
    main()
