"""Probe (run by hand on a GPU box, never collected by pytest): the by-particle result array on
the host -- a pageable NumPy array costs its first device->host copy ~16 GB/s (the runtime pins it
on first use; fresh, pre-touched or huge-page backed alike), 48 GB/s from the second use on.

    python tools/ubench/bp_host_alloc_probe.py
"""
import sys, time, mmap, ctypes
import numpy as np
sys.path.insert(0, ".")
from transport_analysis_amd import _lib
from transport_analysis_amd._lib import lib, _ptr


def main():
    T, A = 10000, 50000
    ctx = _lib.Context(0)
    (slab,) = ctx.stage_alloc(T, A, 3, n_slabs=1, dtype=np.float32)
    rng = np.random.default_rng(5)
    blk = rng.standard_normal((200, A, 3), dtype=np.float32)
    for t in range(0, T, 200):
        slab[t:t + 200] = blk[: min(200, T - t)]
    ctx.stage_commit(0, T)
    def run(name, bp):
        ts = np.empty(T)
        t0 = time.perf_counter()
        ctx._check(lib().ta_vacf_fft(ctx._h, _ptr(ts), _ptr(bp)))
        dt = time.perf_counter() - t0
        print(f"{name}: {dt*1e3:.1f} ms  ({bp.nbytes/dt/1e9:.1f} GB/s of output)", flush=True)
        return ts
    print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
    for rep in range(2):
        run("np.empty (fresh)", np.empty((T, A)))
    warm = np.zeros((T, A))
    run("pre-touched", warm); run("pre-touched again", warm)
    def thp_array(shape):
        n = int(np.prod(shape)) * 8
        m = mmap.mmap(-1, (n + (2 << 20) - 1) // (2 << 20) * (2 << 20), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
        try:
            m.madvise(mmap.MADV_HUGEPAGE)
        except Exception as e:
            print("madvise failed", e)
        return np.frombuffer(m, dtype=np.float64, count=int(np.prod(shape))).reshape(shape)
    for rep in range(2):
        t0 = time.perf_counter(); a = thp_array((T, A)); t1 = time.perf_counter()
        run(f"mmap+MADV_HUGEPAGE (alloc {1e3*(t1-t0):.1f} ms)", a)



if __name__ == "__main__":
    main()
