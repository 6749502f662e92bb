"""2 ranks, one GPU, gloo: checks that a CUDA-tensor all_reduce (async + wait) works at all in this
environment before trusting a single-device rehearsal of the DP step."""
import os, sys, time, torch, torch.distributed as dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get('MCG_DIST_BACKEND', 'gloo'), rank=rank, world_size=world)
t = torch.full((1 << 20,), float(rank + 1), device='cuda')
t0 = time.time()
w = dist.all_reduce(t, async_op=True)
w.wait()
torch.cuda.synchronize()
print('rank', rank, 'allreduce ok', float(t[0]), 'in %.2fs' % (time.time() - t0), flush=True)
dist.barrier()
print('rank', rank, 'barrier ok', flush=True)
dist.destroy_process_group()
