"""NumPy twin of the library's benchmark generator (transport_analysis_amd/csrc/layout.hip,
include/ta_hip.h: ta_stage_synth).  TEST INFRASTRUCTURE: imported only by tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke().

Element i of the synthetic tensor is the sum of the eight 16-bit fields of
splitmix64(seed + 2 i) and splitmix64(seed + 2 i + 1), centred (minus 8 * 32767.5) and scaled
to unit variance by ONE float64 multiply: integer arithmetic plus a correctly rounded product,
hence bit-identical on the CPU and on the GPU (a stateless counter-based
generator, so the CPU baseline and every GPU shard see the same tensor without shipping it).
"""
import numpy as np

SCALE = float.fromhex("0x1.3988e1412ed76p-16")  # 1 / sqrt(8 (65536^2 - 1) / 12)


def _splitmix64(x):
    z = x + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synth(seed, idx):
    """Values of elements `idx` (any integer array) of the tensor with this seed."""
    with np.errstate(over="ignore"):
        i = np.asarray(idx).astype(np.uint64)
        s = np.uint64(seed)
        a = _splitmix64(s + np.uint64(2) * i)
        b = _splitmix64(s + np.uint64(2) * i + np.uint64(1))
        tot = np.zeros(i.shape, dtype=np.int64)
        for k in range(4):
            sh = np.uint64(16 * k)
            tot += ((a >> sh) & np.uint64(0xFFFF)).astype(np.int64)
            tot += ((b >> sh) & np.uint64(0xFFFF)).astype(np.int64)
    return (tot - 262140).astype(np.float64) * SCALE


def synthetic_block(seed, n_frames, n_cols_total, col_lo, col_hi):
    """(n_frames, col_hi - col_lo) float64: columns [col_lo, col_hi) of the (n_frames, n_cols_total) tensor."""
    t = np.arange(n_frames, dtype=np.int64)[:, None]
    c = np.arange(col_lo, col_hi, dtype=np.int64)[None, :]
    return synth(seed, t * n_cols_total + c)
