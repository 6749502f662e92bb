"""Shared by the CPU (gloo, oracle per rank) and GPU (gloo, HIP step per rank) data-parallel tests: the two
shards of one global batch and the single-process emulation of 'per-shard step, gradients averaged before each
Adam update' (SURVEY 8e)."""
import copy

import numpy as np

from oracle import net as onet, updater as oupd


def setup(nf=4, n=2, seed=5, model='normal', dim_zl=0, world=2):
    rng = np.random.RandomState(seed)
    f64 = lambda p: {k: (v.astype(np.float64) if v.dtype.kind == 'f' else v) for k, v in p.items()}
    out_d = 7 if model == 'infogan' else 1
    nets = [f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf)), f64(onet.init_discriminator(rng, 2, 3, out_d, nf)),
            f64(onet.init_discriminator(rng, 3, 3, out_d, nf))]
    shards = []
    t = int(rng.randint(0, 16))
    for r in range(world):
        x = rng.uniform(-1, 1, (n, 3, 16, 64, 64))
        rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=np.float64)
        rnd['t'] = t                                        # one frame index for all ranks (Q7)
        shards.append((x, rnd))
    return nets, shards


class _Rendezvous(Exception):
    pass


def emulate(nets, shards, model='normal', dim_zl=0, grads_out=None):
    """-> (gen, di, dv) parameter dicts after ONE data-parallel iteration over the shards, computed in one
    process: update_core is re-run per phase with the averaged gradients of the earlier phases injected.
    grads_out (a dict): receives the averaged gradients, {'image_dis' | 'video_dis' | 'image_gen': {parameter name: array}}."""
    gen, di, dv = nets
    world = len(shards)
    store, avg = {}, {}

    def make_reduce(r):
        def reduce(name, grads):
            if name in avg:
                for k in grads:
                    grads[k][...] = avg[name][k]
            else:
                store.setdefault(name, {})[r] = {k: v.copy() for k, v in grads.items()}
                raise _Rendezvous()
        return reduce
    finals = []
    for phase in ('image_dis', 'video_dis', 'image_gen', None):
        finals = []
        for r in range(world):
            g_, i_, v_ = copy.deepcopy((gen, di, dv))
            st = [oupd.new_adam_state(p) for p in (g_, i_, v_)]
            try:
                oupd.update_core(model, g_, i_, v_, st[0], st[1], st[2], shards[r][0], None, shards[r][1], dim_zl=dim_zl,
                                 reduce=make_reduce(r))
                finals.append((g_, i_, v_))
            except _Rendezvous:
                pass
        if phase is not None:
            avg[phase] = {k: sum(store[phase][r][k] for r in range(world)) / world for k in store[phase][0]}
    if grads_out is not None:
        grads_out.update(avg)
    return finals[0]
