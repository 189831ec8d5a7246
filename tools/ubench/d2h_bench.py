"""How fast does a big device->host copy into a NumPy array go: pageable, registered (hipHostRegister), pinned?"""
import ctypes
import time

import numpy as np
import torch

hip = ctypes.CDLL("libamdhip64.so")
n = 2 << 30  # bytes
d = torch.empty(n, dtype=torch.uint8, device="cuda")
d.fill_(3)
torch.cuda.synchronize()
h = np.empty(n, dtype=np.uint8)
h[::4096] = 1  # touch the pages
for name in ("pageable", "pageable"):
    t0 = time.perf_counter()
    rc = hip.hipMemcpy(ctypes.c_void_p(h.ctypes.data), ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), 2)
    dt = time.perf_counter() - t0
    print(f"{name}: rc={rc} {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.0f} ms)")
t0 = time.perf_counter()
rc = hip.hipHostRegister(ctypes.c_void_p(h.ctypes.data), ctypes.c_size_t(n), 0)
dt = time.perf_counter() - t0
print(f"hipHostRegister: rc={rc} {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.0f} ms)")
for _ in range(2):
    t0 = time.perf_counter()
    rc = hip.hipMemcpy(ctypes.c_void_p(h.ctypes.data), ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), 2)
    dt = time.perf_counter() - t0
    print(f"registered: rc={rc} {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.0f} ms)")
t0 = time.perf_counter()
hip.hipHostUnregister(ctypes.c_void_p(h.ctypes.data))
print(f"unregister {(time.perf_counter() - t0) * 1e3:.0f} ms")
p = torch.empty(n, dtype=torch.uint8).pin_memory()
for _ in range(2):
    t0 = time.perf_counter()
    p.copy_(d)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"pinned: {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.0f} ms)")
t0 = time.perf_counter()
h[:] = p.numpy()
dt = time.perf_counter() - t0
print(f"host memcpy pinned->numpy: {n / dt / 1e9:.1f} GB/s")
