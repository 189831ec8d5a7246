"""ctypes wrapper around oracle/c/libta_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_SO = os.path.join(_DIR, "libta_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _DIR])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        L = ctypes.c_long
        _lib.oracle_vacf_windowed.argtypes = [dp, L, L, L, dp, dp, ctypes.c_int]
        _lib.oracle_vacf_fft.argtypes = [dp, L, L, L, dp, dp, ctypes.c_int]
        _lib.oracle_vacf_fft_lagsum.argtypes = [dp, L, L, L, dp, ctypes.c_int]
        _lib.oracle_helfand.argtypes = [dp, dp, dp, dp, L, L, L, ctypes.c_double,
                                        ctypes.c_double, dp, dp, ctypes.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _vacf(fn, v, n_threads):
    v = np.ascontiguousarray(v, dtype=np.float64)
    T, A, D = v.shape
    bp = np.zeros((T, A))
    ts = np.zeros(T)
    rc = fn(_p(v), T, A, D, _p(bp), _p(ts), n_threads)
    if rc != 0:
        raise RuntimeError(f"oracle returned {rc}")
    return bp, ts


def vacf_windowed(v, n_threads=1):
    return _vacf(lib().oracle_vacf_windowed, v, n_threads)


def vacf_fft(v, n_threads=1):
    return _vacf(lib().oracle_vacf_fft, v, n_threads)


def vacf_fft_lagsum(v, n_threads=1):
    """Sum over atoms of the per-atom FFT VACF (throughput variant for bench.py's CPU line)."""
    v = np.ascontiguousarray(v, dtype=np.float64)
    T, A, D = v.shape
    out = np.zeros(T)
    rc = lib().oracle_vacf_fft_lagsum(_p(v), T, A, D, _p(out), n_threads)
    if rc != 0:
        raise RuntimeError(f"oracle returned {rc}")
    return out


def helfand(v, x, masses, volumes, temp_avg=300.0, boltzmann=8.314462159e-3, n_threads=1):
    v = np.ascontiguousarray(v, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    m = np.ascontiguousarray(masses, dtype=np.float64)
    vol = np.ascontiguousarray(volumes, dtype=np.float64)
    T, A, D = v.shape
    bp = np.zeros((T, A))
    ts = np.zeros(T)
    rc = lib().oracle_helfand(_p(v), _p(x), _p(m), _p(vol), T, A, D, temp_avg, boltzmann,
                              _p(bp), _p(ts), n_threads)
    if rc != 0:
        raise RuntimeError(f"oracle returned {rc}")
    return bp, ts
