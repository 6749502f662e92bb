#!/usr/bin/env python
"""A/B: the iteration's main chain on a HIGH-priority stream (side streams stay at normal priority) against the default.
    python tools/ab_priority.py f32:32 bf16:256 f32x3:32
Alternates the two forms three times per workload and prints clips/s of each round."""
import sys
import time

import torch

sys.path.insert(0, '.')
import mocogan_chainer_amd.hiplib as hl            # noqa: E402
import mocogan_chainer_amd.step as mstep           # noqa: E402

hl.load(); hl.set_autotune(True)
hi = torch.cuda.Stream(priority=-1)
for spec in (sys.argv[1:] or ('f32:32', 'bf16:256', 'f32x3:32')):
    prec, n = spec.split(':'); n = int(n)
    gen, di, dv = mstep.make_models('normal', num_labels=6, seed=0)
    ts = mstep.TrainStep('normal', gen, di, dv, seed=1, precision=prec, overlap=True, input_ready_early=True)
    x = torch.rand((n, 3, 16, 64, 64), device='cuda') * 2 - 1
    t = torch.randint(0, 6, (n,), device='cuda', dtype=torch.int32)

    def clips_per_s(stream, steps=20):
        with torch.cuda.stream(stream):
            for _ in range(10):
                ts.run(x, t)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                ts.run(x, t)
            torch.cuda.synchronize()
        return n * steps / (time.perf_counter() - t0)
    main = torch.cuda.current_stream()
    for r in range(3):
        print('%-6s batch %4d  default %8.1f   high-priority main %8.1f' % (prec, n, clips_per_s(main), clips_per_s(hi)), flush=True)
