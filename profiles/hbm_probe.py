#!/usr/bin/env python3
"""What this pool's boxes deliver on plain streaming kernels (torch): read-only (sum), copy, write-only (fill) of a 4 GiB tensor."""
import time, torch
n = 1 << 30
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
gb = n * 4 / 1e9
print("read  (sum)   %.2f TB/s" % (gb / t(lambda: x.sum()) / 1e3))
print("read  (amax)  %.2f TB/s" % (gb / t(lambda: x.amax()) / 1e3))
print("copy          %.2f TB/s (read + write bytes)" % (2 * gb / t(lambda: y.copy_(x)) / 1e3))
print("write (fill)  %.2f TB/s" % (gb / t(lambda: y.fill_(1.0)) / 1e3))
