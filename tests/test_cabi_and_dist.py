"""CPU-side checks: the C-ABI library loads and exports every symbol the header
declares; host-side sharding logic; the N>1 reduce path over gloo (world_size 2)."""
import os
import re

import numpy as np
import pytest

from conftest import REPO, scale_rel_err


def test_library_exports_header_symbols():
    from transport_analysis_amd import _lib

    header = open(os.path.join(REPO, "include", "ta_hip.h")).read()
    declared = set(re.findall(r"\b(ta_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.ta_abi_version() == 6


def test_cpu_backend_is_opt_in():
    """Without a GPU the default product path fails loudly; the CPU backend behind the same symbols computes only for a
    caller who asks for it (ta_ctx_create(TA_DEVICE_CPU)) -- tests/test_cpu_backend.py has its parity set."""
    from transport_analysis_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.TAError, match="no usable HIP device"):
        _lib.Context(0)
    c = _lib.Context("cpu")
    assert c.is_cpu
    c.close()


def test_plan_info():
    from transport_analysis_amd import _lib

    # M = R0 * 512, R0 in {1..10, 12, 14, 16, 18, 20} (csrc/wfft.hpp)
    for T, M in ((1, 512), (16, 512), (17, 512), (500, 512), (512, 512), (513, 1024), (1000, 1024),
                 (1025, 1536), (1537, 2048), (2049, 2560), (2561, 3072), (3073, 3584), (3585, 4096),
                 (4097, 4608), (5001, 5120), (5121, 6144), (6145, 7168), (7169, 8192), (8193, 9216),
                 (10000, 10240), (10240, 10240)):
        assert _lib.fft_plan_info(T)["M"] == M
    # beyond one on-chip transform: outer radix R x on-chip M (csrc/wfft.hpp)
    # (outer radix 2, 3, 4, 5, 8, 16 in front of the plans R0 = 12, 14, 16, 18, 20; smallest M (1 + 0.15 R))
    for T, M in ((10241, 12288), (12289, 14336), (14337, 16384), (16385, 18432), (18433, 20480),
                 (20481, 21504), (24577, 27648), (27649, 30720), (30000, 30720), (30721, 32768),
                 (40961, 46080), (50000, 51200), (51201, 57344), (81921, 98304), (100000, 114688),
                 (163840, 163840)):
        info = _lib.fft_plan_info(T)
        assert info["M"] == M
    assert _lib.fft_plan_info(163841) is None  # handled by the direct correlator


def test_atom_shard_partition():
    from transport_analysis_amd.dist import atom_shard

    for n, w in ((100000, 8), (7, 3), (5, 8), (1, 2)):
        edges = [atom_shard(n, r, w) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    with pytest.raises(ValueError):
        atom_shard(10, 2, 2)


def _worker(rank, world, port, T, A, D, out_dir):
    import torch
    import torch.distributed as dist

    from oracle import numpy_oracle as orc
    from transport_analysis_amd.dist import sharded_timeseries

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    v = orc.synthetic_velocities(T, A, D, seed=77)  # every rank can rebuild its columns

    def lagsum(lo, hi):
        if hi == lo:
            return torch.zeros(T, dtype=torch.float64)
        bp, _ = orc.vacf_fft_batched(v[:, lo:hi])
        return torch.from_numpy(bp.sum(axis=1))

    ts = sharded_timeseries(lagsum, A, rank, world)
    np.save(os.path.join(out_dir, f"ts_{rank}.npy"), ts.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("A", [7, 1])
def test_sharded_reduce_gloo_world2(tmp_path, A):
    import torch.multiprocessing as mp

    from oracle import numpy_oracle as orc

    T, D, world = 50, 3, 2
    port = 29500 + (os.getpid() % 2000) + A
    mp.spawn(_worker, args=(world, port, T, A, D, str(tmp_path)), nprocs=world, join=True)
    _, want = orc.vacf_fft_batched(orc.synthetic_velocities(T, A, D, seed=77))
    for r in range(world):
        got = np.load(tmp_path / f"ts_{r}.npy")
        assert scale_rel_err(got, want) < 1e-13


def _class_worker(rank, world, port, T, A, out_dir):
    """The drop-in classes under torch.distributed (gloo here; nccl = RCCL on GPUs): every
    rank runs the same script, the library context is the oracle-backed stand-in."""
    import sys

    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_backend import OracleContext
    from oracle import numpy_oracle as orc
    from transport_analysis_amd import VelocityAutocorr, ViscosityHelfand, _lib
    from transport_analysis_amd._mini_mda import ArrayUniverse

    _lib.Context = OracleContext
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=123)
    u = ArrayUniverse(positions=x, velocities=v, masses=m, dimensions=[60, 60, 60, 90, 90, 90])
    out = {}
    for fft in (True, False):
        a = VelocityAutocorr(u.atoms, dim_type="xz", fft=fft, distributed=True).run()
        out[f"vacf_ts_{int(fft)}"] = a.results.timeseries
        out[f"vacf_bp_{int(fft)}"] = a.results.vacf_by_particle
        out["range"] = np.array(a.results.particle_range)
    h = ViscosityHelfand(u.atoms, distributed=True, linear_fit_window=(2, T - 2)).run()
    out["visc_ts"] = h.results.timeseries
    out["visc_bp"] = h.results.visc_by_particle
    out["visc"] = np.array(h.results.viscosity)
    np.savez(os.path.join(out_dir, f"cls_{rank}.npz"), **out)
    dist.destroy_process_group()


@pytest.mark.parametrize("A", [7, 1])
def test_classes_distributed_gloo_world2(tmp_path, A):
    import torch.multiprocessing as mp

    from oracle import numpy_oracle as orc

    T, world = 40, 2
    port = 31500 + (os.getpid() % 2000) + A
    mp.spawn(_class_worker, args=(world, port, T, A, str(tmp_path)), nprocs=world, join=True)
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=123)
    v32 = v.astype(np.float32).astype(np.float64)  # ArrayUniverse hands out float32 like MDAnalysis
    x32 = x.astype(np.float32).astype(np.float64)
    want_bp, want_ts = orc.vacf_fft_batched(v32[:, :, [0, 2]])
    hbp, hts = orc.helfand(v32, x32, m, np.full(T, 60.0**3), 300.0)
    for r in range(world):
        z = np.load(tmp_path / f"cls_{r}.npz", allow_pickle=True)
        lo, hi = z["range"]
        assert (lo, hi) == ((A * r) // world, (A * (r + 1)) // world)
        for fft in (0, 1):
            assert scale_rel_err(z[f"vacf_ts_{fft}"], want_ts) < 1e-12
            if hi > lo:
                assert scale_rel_err(z[f"vacf_bp_{fft}"], want_bp[:, lo:hi]) < 1e-12
            else:
                assert z[f"vacf_bp_{fft}"].shape == (T, 0)
        assert scale_rel_err(z["visc_ts"], hts) < 1e-12
        if hi > lo:
            assert scale_rel_err(z["visc_bp"], hbp[:, lo:hi]) < 1e-12
        np.testing.assert_allclose(z["visc"], orc.helfand_fit(hts, (2, T - 2)), rtol=1e-9)


def test_new_entry_points_reject_bad_arguments():
    """ta_host_alloc / ta_kernel_timeline argument checks (no GPU needed for the rejections); on a
    GPU-less host the explicit pinned allocation fails loudly, while the RESULT array of an analysis
    degrades to pageable memory with a warning (the run is not lost in _conclude because
    page-locking failed: ADVICE r03) and $TA_AMD_PINNED_RESULTS=0 skips pinning altogether."""
    import ctypes

    from transport_analysis_amd import _lib

    L = _lib.lib()
    p = ctypes.c_void_p()
    assert L.ta_host_alloc(-1, ctypes.byref(p)) != 0
    assert L.ta_host_alloc(16, None) != 0
    assert L.ta_host_alloc_on(-1, -1, ctypes.byref(p)) != 0 and L.ta_host_alloc_on(0, 16, None) != 0
    assert L.ta_host_free(None) == 0
    n = ctypes.c_int(7)
    assert L.ta_kernel_timeline(None, 4, None, None, ctypes.byref(n)) != 0
    if _lib.device_count() == 0:
        with pytest.raises(_lib.TAError, match="pinned host allocation failed|hipSetDevice"):
            _lib.pinned_empty((4, 4))
        with pytest.warns(RuntimeWarning, match="page-locked result array unavailable"):
            a = _lib.result_empty((4, 5))
        assert a.shape == (4, 5) and a.dtype == np.float64 and a.flags.c_contiguous
        with pytest.warns(RuntimeWarning):
            b = _lib.PinnedResult((3, 2), device=0).get()
        assert b.shape == (3, 2)


def test_pinned_results_opt_out(monkeypatch):
    import warnings

    from transport_analysis_amd import _lib

    monkeypatch.setenv("TA_AMD_PINNED_RESULTS", "0")
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # no allocation attempt, so nothing to warn about
        a = _lib.result_empty((2, 3))
    assert a.shape == (2, 3) and a.dtype == np.float64


def test_group_entry_points_reject_bad_arguments():
    """ta_group_* argument checks (no GPU needed); without a GPU a group cannot be created: the
    members are ordinary contexts and there is no CPU backend behind them."""
    import ctypes

    from transport_analysis_amd import _lib

    L = _lib.lib()
    h = ctypes.c_void_p()
    ids = (ctypes.c_int * 2)(0, 0)
    assert L.ta_group_create(None, 1, ctypes.byref(h)) != 0
    assert L.ta_group_create(ids, 0, ctypes.byref(h)) != 0
    assert L.ta_group_create(ids, 2, None) != 0
    assert L.ta_group_size(None) == 0 and L.ta_group_destroy(None) == 0
    assert L.ta_group_stage_commit(None, 0, 1) != 0
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    assert L.ta_group_shard(None, 10, 0, ctypes.byref(lo), ctypes.byref(hi)) != 0
    if _lib.device_count() == 0:
        assert L.ta_group_create(ids, 2, ctypes.byref(h)) != 0 and not h.value
        assert b"no usable HIP device" in L.ta_group_last_error(None)
        with pytest.raises(_lib.TAError, match="no usable HIP device"):
            _lib.Group([0])
    with pytest.raises(ValueError):
        _lib.Group([])



def test_bench_bare_gpus_n_is_its_own_launcher():
    """`python bench.py --gpus 2` with no launcher (the shape of the driver's command) starts its ranks
    itself as a child `torch.distributed.run`; here, without a GPU, both ranks get as far as the GPU
    check and the command fails with THEIR message and a non-zero status -- not with "launch me under
    torchrun".  (The run to rc 0 is the -m gpu test test_bench_bare_gpus_2_starts_its_own_ranks.)"""
    import subprocess
    import sys

    import torch

    if torch.cuda.is_available():
        pytest.skip("the GPU twin of this test covers a box with a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert res.returncode != 0
    assert "bench.py needs a GPU" in res.stderr and "torch.distributed.run" not in res.stdout
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("scaling,atoms,total", [("weak", 3, 24), ("strong", 29, 29), ("strong", 5, 5)])
def test_bench_eight_ranks_rehearsal_on_the_cpu_backend(scaling, atoms, total):
    """The driver's N = 8 command shape (`python bench.py --gpus 8`, which starts torch.distributed.run itself) with
    every rank on the library's opt-in CPU backend (TA_BENCH_CPU=1, gloo): rank-count plumbing, atom_shard
    remainders (29 atoms over 8 ranks; 5 atoms: three ranks hold none), config.collective.ranks == 8, ONE JSON line
    from rank 0, rc 0, and the reduced series equal to the whole tensor's.  A one-GPU box may not carry eight ranks
    on its card; the GPU-side rehearsals (tests/test_gpu_dist.py) stop at four."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TA_BENCH_CPU="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--frames", "300", "--atoms", str(atoms), "--scaling", scaling]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == scaling and d["config"]["n_atoms_total"] == total
    assert d["config"]["collective"]["ranks"] == 8 and d["config"]["sharding"] == "atoms x8"
    assert d["value"] == pytest.approx(300 * total / (d["ms_per_step"] * 1e-3), rel=1e-9)
    ranges = [tuple(r["atom_range"]) for r in sorted(d["config"]["per_rank"], key=lambda r: r["rank"])]
    assert ranges[0][0] == 0 and ranges[-1][1] == total and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    if scaling == "strong":
        from transport_analysis_amd.dist import atom_shard

        assert ranges == [atom_shard(total, r, 8) for r in range(8)]
    assert d["check"]["reduced_series_vs_one_rank_scale_rel"] < 1e-12
    assert "bench.py started" in d["config"]["collective"]["launcher"]
