import ctypes, time, mmap
import numpy as np
hip = ctypes.CDLL("libamdhip64.so")
n = 6 << 30
for trial in range(2):
    p = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(n), 0); dt = time.perf_counter() - t0
    print(f"hipHostMalloc 6 GiB: rc={rc} {dt*1e3:.0f} ms ({n/dt/1e9:.1f} GB/s)")
    buf = (ctypes.c_char * n).from_address(p.value)
    a = np.frombuffer(buf, dtype=np.uint8)
    print("   already zero (sampled every 4 KB):", not a[::4096].any())
    t0 = time.perf_counter(); hip.hipHostFree(p); print(f"   hipHostFree {1e3*(time.perf_counter()-t0):.0f} ms")
# mmap + register in chunks
libc = ctypes.CDLL(None)
m = mmap.mmap(-1, n, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
MADV_HUGEPAGE = 14
print("madvise hugepage rc", libc.madvise(ctypes.c_void_p(addr), ctypes.c_size_t(n), MADV_HUGEPAGE))
chunk = 512 << 20
t_all = time.perf_counter()
for off in range(0, n, chunk):
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(ctypes.c_void_p(addr + off), ctypes.c_size_t(chunk), 0)
    dt = time.perf_counter() - t0
    if off < 3 * chunk or rc: print(f"hipHostRegister 512 MiB chunk: rc={rc} {dt*1e3:.0f} ms ({chunk/dt/1e9:.1f} GB/s)")
print(f"all chunks: {1e3*(time.perf_counter()-t_all):.0f} ms")
# H2D copy speed from registered memory
d = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(1 << 30))
for _ in range(2):
    t0 = time.perf_counter(); rc = hip.hipMemcpy(d, ctypes.c_void_p(addr), ctypes.c_size_t(1 << 30), 1); dt = time.perf_counter() - t0
    print(f"H2D 1 GiB from registered mmap memory: rc={rc} {(1<<30)/dt/1e9:.1f} GB/s")
