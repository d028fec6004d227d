"""Raw PCIe rates of the box in both directions (pinned host memory, one copy engine): what bench.py prices the host-pointer entry
points against.   python tools/probe_d2h.py"""
import json
import time

import torch

n = 800_000_000
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
out = {}
for name, dst, src in (("h2d", d, h), ("d2h", h, d)):
    for _ in range(2):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    out[name + "_pinned_GB_per_s"] = round(5 * n / (time.perf_counter() - t) / 1e9, 2)
# both directions at once (two streams): does the link carry them together?
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2):
        h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize()
out["both_directions_GB_per_s_each"] = round(3 * n / (time.perf_counter() - t) / 1e9, 2)
print(json.dumps(out))
