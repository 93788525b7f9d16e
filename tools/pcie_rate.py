#!/usr/bin/env python3
"""What the box allows: pinned H2D / D2H rate and host memcpy rate (context for tools/host_api_rate.py)."""
import time, torch, numpy as np
n = 1 << 30
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
for name, f in (("H2D pinned", lambda: dev.copy_(pin, non_blocking=True)), ("D2H pinned", lambda: pin.copy_(dev, non_blocking=True))):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("%s: %.1f GB/s" % (name, n / t / 1e9))
a = np.ones(n, dtype=np.uint8); b = np.empty(n, dtype=np.uint8); b[:] = a
t0 = time.perf_counter(); b[:] = a; t = time.perf_counter() - t0
print("host memcpy, one thread: %.1f GB/s" % (n / t / 1e9))
pn = pin.numpy()
t0 = time.perf_counter(); b[:] = pn; t = time.perf_counter() - t0
print("host memcpy from pinned, one thread: %.1f GB/s" % (n / t / 1e9))
