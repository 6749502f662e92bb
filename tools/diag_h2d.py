#!/usr/bin/env python
"""H2D copy rate of a 256-clip uint8 batch (50 MB) from (a) torch pinned memory, (b) a shared-memory segment page-locked with
hipHostRegister (what PrefetchIterator's batch slots are), (c) the same segment not registered (pageable)."""
import time
from multiprocessing import shared_memory

import numpy as np
import torch

n = 256 * 16 * 64 * 64 * 3
dev = torch.device('cuda')
out = torch.empty(n, dtype=torch.uint8, device=dev)


def rate(host, label, non_blocking=True):
    s = torch.cuda.Stream()
    for _ in range(2):
        with torch.cuda.stream(s):
            out.copy_(host, non_blocking=non_blocking)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        e0.record()
        for _ in range(5):
            out.copy_(host, non_blocking=non_blocking)
        e1.record()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print('%-44s %.2f ms per 50 MB copy = %.1f GB/s (host time in the calls %.2f ms each)' % (label, ms, n / ms / 1e6, th / 5 * 1e3))


rate(torch.empty(n, dtype=torch.uint8, pin_memory=True), 'torch pinned tensor')
shm = shared_memory.SharedMemory(create=True, size=n)
arr = np.ndarray((n,), dtype=np.uint8, buffer=shm.buf)
arr[:] = 1
rate(torch.from_numpy(arr), 'shared memory, pageable')
from torch.cuda._pin_memory_utils import pin_memory, unpin_memory
pin_memory(arr.ctypes.data, arr.nbytes)
t = torch.from_numpy(arr)
print('is_pinned:', t.is_pinned())
rate(t, 'shared memory, hipHostRegister')
unpin_memory(arr.ctypes.data)
del t, arr
shm.close(); shm.unlink()
