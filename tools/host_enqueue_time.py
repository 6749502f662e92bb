import sys, time, torch
sys.path.insert(0, '.')
import mocogan_chainer_amd.hiplib as hl, mocogan_chainer_amd.step as mstep
hl.load(); hl.set_autotune(True)
for prec in (sys.argv[1:] or ('f32', 'bf16', 'f32x3')):
    gen, di, dv = mstep.make_models('normal', num_labels=6, seed=0)
    ts = mstep.TrainStep('normal', gen, di, dv, seed=1, precision=prec, overlap=True)
    x = torch.rand((32, 3, 16, 64, 64), device='cuda') * 2 - 1
    t = torch.randint(0, 6, (32,), device='cuda', dtype=torch.int32)
    for _ in range(5): ts.run(x, t)
    torch.cuda.synchronize()
    hs = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ts.run(x, t); hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ts.run(x, t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(prec, 'host enqueue per step (ms):', [round(h * 1e3, 2) for h in hs], ' step (ms): %.2f' % (dt * 1e3))
