import torch, time
x = torch.empty(1<<30, dtype=torch.uint8).pin_memory()
d = torch.empty(1<<30, dtype=torch.uint8, device="cuda")
for _ in range(2): d.copy_(x, non_blocking=True); torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(5): d.copy_(x, non_blocking=True)
torch.cuda.synchronize(); dt=time.perf_counter()-t
print("H2D pinned GB/s:", 5*(1<<30)/dt/1e9)
for sz in (16<<20, 64<<20):
    t=time.perf_counter()
    for i in range(0, 1<<30, sz): d[i:i+sz].copy_(x[i:i+sz], non_blocking=True)
    torch.cuda.synchronize(); print(sz>>20, "MiB pieces GB/s:", (1<<30)/(time.perf_counter()-t)/1e9)
import numpy as np, os
open("/tmp/blob","wb").write(os.urandom(1<<20)*2048)
buf = x.numpy()
fd = os.open("/tmp/blob", os.O_RDONLY)
t=time.perf_counter(); n=os.preadv(fd,[memoryview(buf)],0); print("pread into pinned GB/s (1 thread):", n/(time.perf_counter()-t)/1e9, n)
t=time.perf_counter(); n=os.preadv(fd,[memoryview(buf)],1<<30); print("pread into pinned GB/s (1 thread, 2nd GiB):", n/(time.perf_counter()-t)/1e9, n)
