"""Run step 1 several times from an identical snapshot; report which saved tensors differ between repeats."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from oracle import net as onet, updater as oupd
import mocogan_chainer_amd.hiplib as hl, mocogan_chainer_amd.layout as lay, mocogan_chainer_amd.nets as nets, mocogan_chainer_amd.step as step
from test_gpu_step import dev, rel_l2, _f64, noise_to_dev, draw_to_dev
F64 = np.float64
model, dim_zl, nf, n, seed = 'normal', 0, 8, 3, 306
rng = np.random.RandomState(seed)
gen = _f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf)); di = _f64(onet.init_discriminator(rng, 2, 3, 1, nf)); dv = _f64(onet.init_discriminator(rng, 3, 3, 1, nf))
G = nets.GenNet(dim_zl=dim_zl, n_filters=nf); DI = nets.DisNet(2, 3, 1, nf, use_noise=True); DV = nets.DisNet(3, 3, 1, nf, use_noise=True)
G.load_reference_params(gen), DI.load_reference_params(di), DV.load_reference_params(dv)
ts = step.TrainStep(model, G, DI, DV)
def flat(d, pre=''):
    out = {}
    if isinstance(d, torch.Tensor): out[pre] = d
    elif isinstance(d, dict):
        for k, v in d.items(): out.update(flat(v, pre + '/' + str(k)))
    return out
# capture saved dicts of every forward
caps = {}
for name, net in (('G', G), ('DI', DI), ('DV', DV)):
    def mk(name, net, of):
        def fwd(*a, **k):
            r = of(*a, **k)
            caps.setdefault(name, []).append(r[1])
            return r
        return fwd
    net.forward = mk(name, net, net.forward)
def snapshot():
    return [(net.fp.p.clone(), net.fp.m.clone(), net.fp.v.clone(), net.t, {k: v.clone() for k, v in net.running.items()}) for net in (G, DI, DV)]
def restore(snap):
    for net, (p, m, v, t, run) in zip((G, DI, DV), snap):
        net.fp.p.copy_(p); net.fp.m.copy_(m); net.fp.v.copy_(v); net.t = t
        for k in run: net.running[k].copy_(run[k])
def mkinputs():
    x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64)); t_real = rng.randint(0, 6, n)
    rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=F64)
    inject = {'t': rnd['t'], 'gen': draw_to_dev(rnd['gen'])}
    for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
        inject[k] = noise_to_dev(lay, rnd[k])
    return dev(x_real), dev(t_real, torch.int32), inject
x0, t0, inj0 = mkinputs()
ts.run(x0, t0, inj0)
x1, t1, inj1 = mkinputs()
snap = snapshot()
results = []
# capture intermediate gradients of the D backward passes
import types
def wrap_backward(net, name):
    ob = net.backward
    def bwd(saved, g_logits, param_grads, gx=None, gx_geom=None, gx_accumulate=False):
        r = ob(saved, g_logits, param_grads, gx=gx, gx_geom=gx_geom, gx_accumulate=gx_accumulate)
        if gx is not None: caps.setdefault(name + '.gx', []).append({'gx': gx.clone(), 'glog': g_logits.clone()})
        return r
    net.backward = bwd
wrap_backward(DI, 'DI'); wrap_backward(DV, 'DV')
mode = sys.argv[1] if len(sys.argv) > 1 else 'none'
trace = []
def wrapcall(fname, out_idx):
    of = getattr(hl, fname)
    def f(*a, **k):
        r = of(*a, **k)
        o = a[out_idx]
        if isinstance(o, torch.Tensor): trace.append((fname + str(tuple(o.shape)), o.clone()))
        return r
    setattr(hl, fname, f)
wrapcall('fc_dgrad', 7); wrapcall('bn_act_bwd', 7); wrapcall('conv_dgrad', 4); wrapcall('loss_gen', 7); wrapcall('loss_gen', 8)
prng = np.random.RandomState(1)
for rep in range(6):
    restore(snap); ts.iteration = 1
    if mode == 'bias' and rep > 0:
        for net in (DI, DV):
            for l in (2, 3, 4):
                b = net.fp.param('dc%d/b' % l)
                b.add_(torch.tensor(prng.choice([-1e-4, 1e-4], size=tuple(b.shape)), dtype=torch.float32, device='cuda'))
    if mode == 'gbias' and rep > 0:
        for l in (1, 2, 3, 4):
            b = G.fp.param('dc%d/b' % l)
            b.add_(torch.tensor(prng.choice([-1e-4, 1e-4], size=tuple(b.shape)), dtype=torch.float32, device='cuda'))
    caps.clear(); trace.clear()
    out = ts.run(x1, t1, inj1)
    torch.cuda.synchronize()
    rec = {}
    for name in caps:
        for i, sv in enumerate(caps[name]):
            for k, v in flat(sv).items(): rec['%s#%d%s' % (name, i, k)] = v.clone()
    rec['gx'] = out['gx_fake'].clone()
    for name, net in (('G', G), ('DI', DI), ('DV', DV)):
        rec[name + '.grad'] = net.fp.g.clone()
    for i, (nm, tt) in enumerate(trace): rec['trace%03d_%s' % (i, nm)] = tt
    results.append(rec)
base = results[0]
for rep in range(1, 6):
    bad = []
    for k in base:
        a, b = base[k], results[rep][k]
        if a.dtype.is_floating_point:
            e = float((a - b).norm() / (a.norm() + 1e-30))
            if e > 1e-5: bad.append((k, '%.1e' % e))
    print('rep', rep, 'vs 0:', [b for b in bad if b[0].startswith('trace')][:12] if bad else 'identical within 1e-5')

# ---- element-level analysis of D_V layer-4 BN backward between rep 0 and the first differing rep
def find(rec, prefix):
    ks = sorted(k for k in rec if k.startswith('trace'))
    return ks
ks = find(base, 'trace')
k40 = [k for k in ks if 'bn_act_bwd(3, 4, 4, 4, 64)' in k][-1]
kin = ks[ks.index(k40) - 1]
for rep in range(1, 6):
    a, b = base[k40], results[rep][k40]
    if float((a - b).norm() / a.norm()) < 1e-5: continue
    print('analysing rep', rep, k40, 'input', kin)
    y4 = results[rep]['DV#1/y/4']; st = results[rep]['DV#1/stats/4']; y4b = base['DV#1/y/4']; stb = base['DV#1/stats/4']
    C = 64
    v_rep = torch.addcmul(st[3*C:4*C], y4.reshape(-1, C), st[2*C:3*C])
    v_base = torch.addcmul(stb[3*C:4*C], y4b.reshape(-1, C), stb[2*C:3*C])
    flips = ((v_rep < 0) != (v_base < 0)).nonzero()
    print('  mask flips (torch recompute):', flips.tolist(), [ (float(v_rep[i, j]), float(v_base[i, j])) for i, j in flips.tolist()])
    gin_a, gin_b = base[kin].reshape(-1, C), results[rep][kin].reshape(-1, C)
    print('  input rel diff', float((gin_a - gin_b).norm() / gin_a.norm()), 'max abs', float((gin_a - gin_b).abs().max()))
    d = (a - b).reshape(-1, C).abs()
    top = torch.topk(d.reshape(-1), 5).indices
    for t in top.tolist():
        i, j = t // C, t % C
        print('  out[%d,%d] base %.4e rep %.4e  v_base %.3e v_rep %.3e gin %.4e' % (i, j, float(a.reshape(-1, C)[i, j]), float(b.reshape(-1, C)[i, j]), float(v_base[i, j]), float(v_rep[i, j]), float(gin_a[i, j])))
    print('  |v| smallest:', torch.topk(-v_base.abs().reshape(-1), 5).values.tolist())
    break
