"""Data-parallel parity on the GPU: 2 ranks (both on cuda:0, gradients exchanged over gloo -- one GPU box) run the
HIP step on their shard of the batch with the product's GradExchange, side streams on; the updated parameters
must equal the float64 oracle's emulation of "per-shard iteration, gradients averaged before each Adam update"
and the two replicas must stay bit-identical."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NF, N, MODEL, DIM_ZL = 4, 2, 'infogan', 6


def _worker(rank, world, port, q, sync_bn=False, precision='f32', nf=NF, backend='gloo'):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import traceback
    import torch.distributed as dist
    torch.cuda.set_device(rank if backend == 'nccl' else 0)       # nccl (= RCCL): one GPU per rank; gloo: both ranks on cuda:0
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import dp_common
        from oracle import updater as oupd
        import mocogan_chainer_amd.hiplib as hl
        import mocogan_chainer_amd.layout as lay
        import mocogan_chainer_amd.nets as nets
        import mocogan_chainer_amd.step as step
        hl.load()
        (gen, di, dv), shards = dp_common.setup(nf=nf, n=N, seed=11, model=MODEL, dim_zl=DIM_ZL, world=world)
        G = nets.GenNet(dim_zl=DIM_ZL, n_filters=nf)
        DI = nets.DisNet(2, 3, 7, nf, use_noise=True)
        DV = nets.DisNet(3, 3, 7, nf, use_noise=True)
        ts = step.TrainStep(MODEL, G, DI, DV, exchange=step.GradExchange(force=world == 1), rank=rank, overlap=True, sync_bn=sync_bn,
                            precision=precision)
        assert dist.get_backend() == backend
        for net, p in ((G, gen), (DI, di), (DV, dv)):
            net.load_reference_params(p)
            net.load_adam_state(oupd.new_adam_state(p))
        x, rnd = shards[rank]
        dev = lambda a, dt=torch.float32: torch.tensor(np.asarray(a), dtype=dt, device='cuda')
        inject = {'t': rnd['t'], 'gen': {k: (dev(v, torch.int32) if k == 'labels' else dev(v)) for k, v in rnd['gen'].items()}}
        for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
            inject[k] = [lay.act_to_dev(dev(a)) for a in rnd[k]]
        t_real = dev(np.zeros(N), torch.int32)
        ts.run(dev(x), t_real, inject)
        torch.cuda.synchronize()
        out = {name: {k: np.asarray(v.cpu() if torch.is_tensor(v) else v) for k, v in net.export_reference_params().items()}
               for name, net in (('gen', G), ('di', DI), ('dv', DV))}
        q.put((rank, out, ts.losses()))
    except Exception:                                                # never leave the parent waiting on the queue
        q.put((rank, 'error', traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run_ranks(sync_bn, port_base, precision='f32', nf=NF, backend='gloo', world=2):
    import queue
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = port_base + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, sync_bn, precision, nf, backend)) for r in range(world)]
    [p.start() for p in procs]
    res, t0 = [], time.time()
    while len(res) < world:
        try:
            res.append(q.get(timeout=10))
        except queue.Empty:
            print('waiting for the ranks ... %.0f s' % (time.time() - t0), flush=True)
            assert time.time() - t0 < 400 and any(p.is_alive() for p in procs), "ranks died or hung"
    [p.join(60) for p in procs]
    for r in res:
        assert r[1] != 'error', r[2]
    res.sort(key=lambda r: r[0])
    assert all(p.exitcode == 0 for p in procs)
    return res


UPDATE_TOL = 1e-3       # measured 1.5e-4 .. 2e-4 (rounds 3-4); round 4's review: 1e-2 left 60x of slack


def _update_errors(res, nets, ref, grads=None):
    """Worst relative error of the parameter UPDATE (parameters move by ~alpha per step) of rank 0 against the oracle, every tensor
    held to UPDATE_TOL.  grads: the oracle's (averaged) gradients {'image_gen' | 'image_dis' | 'video_dis': {name: array}}.  Adam's
    first step with beta1 = 5e-5 is sign-like: an element whose gradient (incl. the WeightDecay term) is within the device / oracle
    difference of zero may step the other way -- an O(alpha) difference no tolerance on the gradient excludes -- so, as in
    test_gpu_step.check_params, elements with |g| below 1e-3 of the tensor's rms are left out of the norm."""
    worst = 0.0
    gname = {'gen': 'image_gen', 'di': 'image_dis', 'dv': 'video_dis'}
    for name, refp in zip(('gen', 'di', 'dv'), ref):
        for k, v in refp.items():
            if 'avg_' in k or k.endswith('/N') or v.dtype.kind != 'f':
                continue
            got = res[0][1][name][k].astype(np.float64)
            base = nets[('gen', 'di', 'dv').index(name)][k]
            du, dr = got - base, v - base
            if np.abs(dr).max() < 1e-12:
                continue
            if grads is not None and k in grads.get(gname[name], {}):
                gt = np.asarray(grads[gname[name]][k], np.float64) + 1e-5 * base          # WeightDecay(1e-5) hook (train.py:96)
                keep = np.abs(gt) > 1e-3 * np.sqrt(np.mean(gt * gt))
                if keep.sum() == 0:
                    continue
                du, dr = du[keep], dr[keep]
            err = np.linalg.norm(du - dr) / max(np.linalg.norm(dr), 1e-30)
            worst = max(worst, err)
            assert err < UPDATE_TOL, (name, k, err)
    return worst


def test_two_rank_synchronised_batchnorm_matches_the_global_batch_oracle():
    """sync_bn=True: BatchNorm statistics (and the sums of its backward pass) are all-reduced, so two ranks with n
    clips each must reproduce ONE oracle iteration on the concatenated batch of 2n clips -- with the per-shard
    "sample 0" terms of quirk Q1 (rows 0 and n), the only place where the ranks still differ from a single device."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dp_common
    from oracle import updater as oupd
    res = _run_ranks(True, 31600)
    for name in ('gen', 'di', 'dv'):
        for k, v in res[0][1][name].items():
            if k.endswith('/N'):
                continue
            assert np.array_equal(v, res[1][1][name][k]), (name, k)      # incl. the running statistics: global now
    nets, shards = dp_common.setup(nf=NF, n=N, seed=11, model=MODEL, dim_zl=DIM_ZL, world=2)
    gen, di, dv = nets
    import copy
    base = copy.deepcopy(nets)
    x = np.concatenate([s[0] for s in shards])
    r0, r1 = shards[0][1], shards[1][1]
    rnd = {'t': r0['t'],
           'gen': {'h0': np.concatenate((r0['gen']['h0'], r1['gen']['h0'])), 'zc': np.concatenate((r0['gen']['zc'], r1['gen']['zc'])),
                   'e': np.concatenate((r0['gen']['e'], r1['gen']['e']), axis=1),
                   'labels': None if r0['gen']['labels'] is None else np.concatenate((r0['gen']['labels'], r1['gen']['labels']))}}
    for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
        rnd[k] = [np.concatenate((a, b)) for a, b in zip(r0[k], r1[k])]
    og, oi, ov = (oupd.new_adam_state(p) for p in (gen, di, dv))
    o = oupd.update_core(MODEL, gen, di, dv, og, oi, ov, x, np.zeros(2 * N, dtype=np.int64), rnd, dim_zl=DIM_ZL, q1_rows=[0, N], keep=True)
    grads = {'image_gen': o['grads_gen'], 'image_dis': o['grads_dis_i'], 'video_dis': o['grads_dis_v']}
    print('sync-BN worst relative update error', _update_errors(res, base, (gen, di, dv), grads))
    for name, p in (('gen', gen), ('di', di), ('dv', dv)):
        for k, v in p.items():
            if 'avg_' in k:
                assert np.allclose(res[0][1][name][k], v, rtol=1e-4, atol=1e-6), (name, k)


def test_two_rank_hip_step_matches_the_sharded_oracle():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dp_common
    res = _run_ranks(False, 29600)
    # replicas bit-identical: same averaged gradients, same Adam
    for name in ('gen', 'di', 'dv'):
        for k, v in res[0][1][name].items():
            if 'avg_' in k or k.endswith('/N'):
                continue                                    # BatchNorm running statistics are per rank (own shard)
            assert np.array_equal(v, res[1][1][name][k]), (name, k)
    assert res[0][2]['image_gen/loss'] != res[1][2]['image_gen/loss']       # different shards

    # the oracle was drawn with labels in rnd; the HIP side used t_real = 0 for the real clips' labels
    nets, shards = dp_common.setup(nf=NF, n=N, seed=11, model=MODEL, dim_zl=DIM_ZL, world=2)
    import copy
    from oracle import updater as oupd
    # emulate() runs update_core with t_real=None; infogan's categorical term needs the real labels: patch them in
    orig = oupd.update_core

    def with_labels(model, gen, di, dv, og, oi, ov, x, t_real, rnd, **kw):
        return orig(model, gen, di, dv, og, oi, ov, x, np.zeros(N, dtype=np.int64), rnd, **kw)
    oupd.update_core = with_labels
    try:
        avg = {}
        ref = dp_common.emulate(nets, shards, model=MODEL, dim_zl=DIM_ZL, grads_out=avg)
    finally:
        oupd.update_core = orig
    worst = _update_errors(res, nets, ref, avg)
    print('worst relative update error', worst)



def test_two_rank_bf16_networks_with_synchronised_batchnorm():
    """bf16 networks keep the gradient BatchNorm's backward writes in bf16; with sync_bn the backward pass goes through
    mcg_bn_bwd_sums / mcg_bn_act_bwd_from_sums, which take the same MCG_IO_* flags as mcg_bn_act_bwd (round 2 raised
    'expected float32, got bfloat16' here).  n_filters = 8 puts layers 2..4 of D and 3..5 of G on bf16-stored operands.
    The replicas must stay bit-identical and the losses must agree with the fp32 run of the same shards to the bf16
    tolerance (loss abs <= 5e-2)."""
    res16 = _run_ranks(True, 33600, precision='bf16', nf=8)
    res32 = _run_ranks(True, 35600, precision='f32', nf=8)
    for name in ('gen', 'di', 'dv'):
        for k, v in res16[0][1][name].items():
            if not k.endswith('/N'):
                assert np.array_equal(v, res16[1][1][name][k]), (name, k)
    for r in range(2):
        for k in res16[r][2]:
            assert abs(res16[r][2][k] - res32[r][2][k]) < 5e-2, (r, k, res16[r][2][k], res32[r][2][k])


def test_two_rank_split_fp32_networks(monkeypatch):
    """precision 'f32x3' under data parallelism: the replicas stay bit-identical (every rank takes the same launch forms:
    MCG_SPLIT=always here, rank 0's table in train.py / bench.py) and the losses agree with the fp32-MFMA run of the same shards to
    fp32 rounding -- the split form is an fp32 computation."""
    monkeypatch.setenv('MCG_SPLIT', 'always')
    res3 = _run_ranks(False, 39600, precision='f32x3', nf=16)
    monkeypatch.setenv('MCG_SPLIT', 'never')
    res32 = _run_ranks(False, 41600, precision='f32', nf=16)
    for name in ('gen', 'di', 'dv'):
        for k, v in res3[0][1][name].items():
            if 'avg_' in k or k.endswith('/N'):
                continue
            assert np.array_equal(v, res3[1][1][name][k]), (name, k)
    for r in range(2):
        for k in res3[r][2]:
            assert abs(res3[r][2][k] - res32[r][2][k]) < 1e-4, (r, k, res3[r][2][k], res32[r][2][k])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs two GPUs (the driver's multi-GPU node)")
def test_two_rank_rccl_matches_the_sharded_oracle():
    """The same parity over the nccl backend (= RCCL over xGMI), one GPU per rank: covers what gloo cannot -- the
    late-bucket all_reduce issued from the weight-gradient stream and work.wait()'s stream semantics."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dp_common
    from oracle import updater as oupd
    res = _run_ranks(False, 37600, backend='nccl')
    for name in ('gen', 'di', 'dv'):
        for k, v in res[0][1][name].items():
            if 'avg_' in k or k.endswith('/N'):
                continue
            assert np.array_equal(v, res[1][1][name][k]), (name, k)
    nets, shards = dp_common.setup(nf=NF, n=N, seed=11, model=MODEL, dim_zl=DIM_ZL, world=2)
    orig = oupd.update_core
    oupd.update_core = lambda model, gen, di, dv, og, oi, ov, x, t_real, rnd, **kw: orig(
        model, gen, di, dv, og, oi, ov, x, np.zeros(N, dtype=np.int64), rnd, **kw)
    try:
        avg = {}
        ref = dp_common.emulate(nets, shards, model=MODEL, dim_zl=DIM_ZL, grads_out=avg)
    finally:
        oupd.update_core = orig
    print('RCCL: worst relative update error', _update_errors(res, nets, ref, avg))


def test_one_rank_rccl_matches_the_oracle():
    """One GPU cannot hold two RCCL ranks, but a world of ONE over the nccl backend (= RCCL) runs the product's whole exchange path --
    GradExchange(force=True): the late bucket's all_reduce issued from the weight-gradient stream, work.wait() on the main stream,
    the parameter broadcast -- and a SUM over one rank with grad_scale 1 must leave the teacher-forced iteration equal to the oracle's."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dp_common
    from oracle import updater as oupd
    res = _run_ranks(False, 38600, backend='nccl', world=1)
    nets, shards = dp_common.setup(nf=NF, n=N, seed=11, model=MODEL, dim_zl=DIM_ZL, world=1)
    orig = oupd.update_core
    oupd.update_core = lambda model, gen, di, dv, og, oi, ov, x, t_real, rnd, **kw: orig(
        model, gen, di, dv, og, oi, ov, x, np.zeros(N, dtype=np.int64), rnd, **kw)
    try:
        avg = {}
        ref = dp_common.emulate(nets, shards, model=MODEL, dim_zl=DIM_ZL, grads_out=avg)
    finally:
        oupd.update_core = orig
    print('RCCL, one rank: worst relative update error', _update_errors(res, nets, ref, avg))


def test_one_rank_rccl_rehearsal_of_the_bench_line():
    """bench.py's data-parallel code path with a world of ONE over nccl (MCG_DP_REHEARSE_NCCL): process group, tile-table broadcast,
    bucketed gradient all-reduce, the timing collectives, the compact line naming the backend."""
    import json
    import math
    import subprocess
    env = dict(os.environ, MCG_DP_REHEARSE_NCCL='1', MASTER_PORT='37711', MCG_BENCH_DETAIL=os.devnull)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2', '--batch', '4', '--dtype', 'f32', '--no-cpu-baseline', '--secondary', '0']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, lines                # (RCCL's version banner goes to descriptor 1: bench.py keeps it off its stdout)
    line = json.loads(lines[0])
    assert line['dist']['backend'] == 'nccl' and line['dist']['world_size'] == 1 and line['value'] > 0
    assert all(math.isfinite(v) for v in line['losses'].values()), line['losses']


def test_four_rank_single_device_rehearsal_of_the_eight_gpu_branch():
    """bench.py's `world == 8` branch -- every rank's preflight of a secondary workload, pretune_and_share_tiles, the configs[4]-shaped
    secondaries in both fp32 forms, Philox rank offsets, the MAX-over-ranks timing, ONE stdout line from rank 0 -- walked on a one-GPU
    box: the driver's own launch line (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) with FOUR ranks on
    cuda:0 over gloo (MCG_SINGLE_DEVICE / MCG_DIST_BACKEND), MCG_BENCH_SECONDARY_DP=1 taking the branch that `world == 8` takes, and
    MCG_BENCH_REHEARSAL_BATCH=2 standing in for 32 / 128 clips per rank.  Four, not eight: this pool's GPU boxes admit at most six
    processes on a card (the harness's process guard), so the eight-rank case itself cannot run before an 8-GPU node does; what
    differs at eight is the value of `world` in the same code.  (reference train.py:87-91: the single-device assumption replaced.)"""
    import json
    import math
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, 'detail.json')
        env = dict(os.environ, MCG_SINGLE_DEVICE='1', MCG_DIST_BACKEND='gloo', MCG_BENCH_SECONDARY_DP='1', MCG_BENCH_REHEARSAL_BATCH='2',
                   MCG_BENCH_DETAIL=detail, HSA_ENABLE_IPC_MODE_LEGACY='0')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '4', '--master-addr', '127.0.0.1',
               '--master-port', '37733', os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '2', '--warmup', '1', '--batch', '2',
               '--secondary-steps', '2']
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
        assert len(lines) == 1, out.stdout[-2000:]
        line = json.loads(lines[0])
        assert line['n_gpus'] == 4 and line['dist'] == {'backend': 'gloo', 'world_size': 4} and line['value'] > 0
        assert line['config']['per_gpu_batch'] == 2 and line['config']['global_batch'] == 8 and line['config']['parallelism'] == 'dp4'
        assert 'rehearsal' in line['data'] and 'cpu_baseline' not in line           # (a rehearsal is never a reported number)
        assert all(math.isfinite(v) for v in line['losses'].values()), line['losses']
        sec = line['secondary']
        assert [s['dtype'] for s in sec] == ['f32x3', 'f32'] and all('error' not in s and s['value'] > 0 for s in sec), sec
        full = json.load(open(detail))
        assert [s['config']['global_batch'] for s in full['secondary']] == [8, 8]
        assert all(s['dist']['world_size'] == 4 for s in full['secondary'])
        ranks_ms = full['dist']['per_rank_ms_per_step']
        assert ranks_ms['min'] > 0 and ranks_ms['max'] >= ranks_ms['median'] >= ranks_ms['min']
