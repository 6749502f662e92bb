"""Data-parallel semantics on CPU: 2 ranks over gloo.  The reference is single-device, so
DP is defined here: each rank runs update_core on its shard of the batch (own BatchNorm statistics, own
noise, the SAME frame index t), the three flat gradients are averaged with GradExchange before each
Adam update, and all replicas stay identical.  The oracle plays the per-rank step (the HIP step needs a
GPU); the exchange code under test is the product's own (mocogan-chainer_amd/step.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dp_common
    return dp_common.setup()


def _flatten(d):
    keys = sorted(d)
    return keys, torch.cat([torch.as_tensor(d[k]).reshape(-1) for k in keys])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import updater as oupd
    import mocogan_chainer_amd.step as step
    ex = step.GradExchange()
    assert ex.world == 2
    (gen, di, dv), shards = _setup()
    og, oi, ov = (oupd.new_adam_state(p) for p in (gen, di, dv))

    def reduce(name, grads):
        keys, flat = _flatten(grads)
        half = flat.numel() // 2                            # two buckets in flight, like D_V's gradient
        handles = [ex.start(flat[:half]), ex.start(flat[half:])]   # async all-reduce (SUM) ...
        for h in handles:
            ex.finish(h)                                    # ... wait: the buffers hold the SUM over the ranks
        flat *= ex.grad_scale                               # what mcg_adam_wd's grad_scale argument does on the device
        o = 0
        for k in keys:
            n = grads[k].size
            grads[k][...] = flat[o:o + n].reshape(grads[k].shape).numpy()
            o += n

    x, rnd = shards[rank]
    out = oupd.update_core('normal', gen, di, dv, og, oi, ov, x, None, rnd, reduce=reduce)
    # broadcast_params is a no-op on identical replicas but must leave rank 0's values everywhere
    t0 = torch.as_tensor(gen['dc3/W']).clone()
    ex.broadcast_params([t0], src=0)
    q.put((rank, {k: v for k, v in gen.items() if k.endswith('/W')}, di['dc2/W'], dv['bn3/gamma'], out['loss_gen'], t0.numpy()))
    dist.destroy_process_group()


def test_two_rank_gradient_averaging_matches_the_sharded_oracle():
    from oracle import updater as oupd
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # replicas identical after the step
    for k in res[0][1]:
        assert np.array_equal(res[0][1][k], res[1][1][k]), k
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
    assert np.array_equal(res[0][5], res[1][5])
    assert res[0][4] != res[1][4]                                # different shards -> different losses

    # single-process emulation: per-shard gradients averaged by hand (tests/dp_common.py)
    import dp_common
    nets, shards = _setup()
    ref_gen, ref_di, ref_dv = dp_common.emulate(nets, shards)
    for k in res[0][1]:
        assert np.allclose(res[0][1][k], ref_gen[k], rtol=1e-12, atol=1e-15), k
    assert np.allclose(res[0][2], ref_di['dc2/W'], rtol=1e-12, atol=1e-15)
    assert np.allclose(res[0][3], ref_dv['bn3/gamma'], rtol=1e-12, atol=1e-15)


def test_rank_streams_are_disjoint_and_frame_index_is_shared():
    import mocogan_chainer_amd.step as step
    TS = step.TrainStep
    seen = set()
    for it in range(3):
        for rank in range(8):
            base = TS.stream_base(it, rank)
            ids = set(range(base, base + TS.STREAMS_PER_RANK))
            assert not (ids & seen)
            seen |= ids

    class Dummy:
        seed = 3
    ts = [TS.frame_index(Dummy(), it, 16) for it in range(50)]
    assert ts == [TS.frame_index(Dummy(), it, 16) for it in range(50)] and 0 <= min(ts) and max(ts) < 16 and len(set(ts)) > 5
