"""Two ranks on ONE GPU (gloo process group): every rank runs the real drop-in classes with
distributed=True on the real HIP library -- stages and correlates only its atom block -- and
the reduced timeseries must equal the oracle's over ALL atoms (scale-relative 1e-10).  What a
1-GPU box can prove about the N > 1 path: HIP-computed partial lag sums reduced across ranks
(velocityautocorr.py:214,237, viscosity.py:233 are the reduce points in the reference)."""
import os

import numpy as np
import pytest

from conftest import REPO, scale_rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _worker(rank, world, port, T, A, out_dir):
    import torch.distributed as dist

    from oracle import numpy_oracle as orc
    from transport_analysis_amd import VelocityAutocorr, ViscosityHelfand
    from transport_analysis_amd._mini_mda import ArrayUniverse

    os.environ.pop("LOCAL_RANK", None)
    os.environ["TA_AMD_DEVICE"] = "0"  # both ranks share the box's one GPU
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=321)
    u = ArrayUniverse(positions=x, velocities=v, masses=m, dimensions=[60, 60, 60, 90, 90, 90])
    out = {}
    for fft in (True, False):
        for byp in (True, False):
            a = VelocityAutocorr(u.atoms, fft=fft, distributed=True, by_particle=byp).run()
            out[f"vacf_ts_{int(fft)}_{int(byp)}"] = a.results.timeseries
            if byp:
                out[f"vacf_bp_{int(fft)}"] = a.results.vacf_by_particle
            out["range"] = np.array(a.results.particle_range)
    h = ViscosityHelfand(u.atoms, distributed=True).run()
    out["visc_ts"] = h.results.timeseries
    out["visc_bp"] = h.results.visc_by_particle
    np.savez(os.path.join(out_dir, f"g_{rank}.npz"), **out)
    dist.destroy_process_group()


@pytest.mark.parametrize("T,A", [(700, 7), (64, 1), (1500, 33)])
def test_classes_distributed_two_ranks_one_gpu(tmp_path, T, A):
    import torch.multiprocessing as mp

    from oracle import numpy_oracle as orc

    world = 2
    port = 33500 + (os.getpid() % 2000) + A
    mp.spawn(_worker, args=(world, port, T, A, str(tmp_path)), nprocs=world, join=True)
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=321)
    v32 = v.astype(np.float32).astype(np.float64)  # ArrayUniverse hands out float32 like MDAnalysis
    x32 = x.astype(np.float32).astype(np.float64)
    want_bp, want_ts = orc.vacf_fft_batched(v32)
    hbp, hts = orc.helfand(v32, x32, m, np.full(T, 60.0**3), 300.0)
    for r in range(world):
        z = np.load(tmp_path / f"g_{r}.npz", allow_pickle=True)
        lo, hi = z["range"]
        assert (lo, hi) == ((A * r) // world, (A * (r + 1)) // world)
        for fft in (0, 1):
            for byp in (0, 1):
                assert scale_rel_err(z[f"vacf_ts_{fft}_{byp}"], want_ts) < TOL
            if hi > lo:
                assert scale_rel_err(z[f"vacf_bp_{fft}"], want_bp[:, lo:hi]) < TOL
            else:
                assert z[f"vacf_bp_{fft}"].shape == (T, 0)
        assert scale_rel_err(z["visc_ts"], hts) < TOL
        if hi > lo:
            assert scale_rel_err(z["visc_bp"], hbp[:, lo:hi]) < TOL


def _worker_nccl(rank, world, port, T, A, out_dir):
    """One rank, nccl (= RCCL) process group: the branch every real multi-GPU run of the classes
    takes -- lag sums produced into a device tensor on torch's stream
    (dist.staged_timeseries_on_device), all-reduced on the device, by-particle block copied back."""
    import torch
    import torch.distributed as dist

    from oracle import numpy_oracle as orc
    from transport_analysis_amd import VelocityAutocorr, ViscosityHelfand, dist as tad
    from transport_analysis_amd._mini_mda import ArrayUniverse

    os.environ.pop("TA_AMD_DEVICE", None)
    os.environ["LOCAL_RANK"] = "0"
    # a group of one rank reduces nothing unless asked: here RCCL is to run the collective
    os.environ["TA_AMD_FORCE_COLLECTIVE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            device_id=torch.device("cuda", 0))
    assert tad.uses_device_reduce() and tad.default_device() == 0
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=77)
    u = ArrayUniverse(positions=x, velocities=v, masses=m, dimensions=[60, 60, 60, 90, 90, 90])
    out = {}
    for fft in (True, False):
        for byp in (True, False):
            a = VelocityAutocorr(u.atoms, fft=fft, distributed=True, by_particle=byp).run()
            out[f"vacf_ts_{int(fft)}_{int(byp)}"] = a.results.timeseries
            if byp:
                out[f"vacf_bp_{int(fft)}"] = a.results.vacf_by_particle
            out["range"] = np.array(a.results.particle_range)
    for kw, tag in (({}, "f64"), ({"float32": True}, "f32"), ({"fft": True}, "fft")):
        h = ViscosityHelfand(u.atoms, distributed=True, **kw).run()
        out[f"visc_ts_{tag}"] = h.results.timeseries
        out[f"visc_bp_{tag}"] = h.results.visc_by_particle
    h = ViscosityHelfand(u.atoms, distributed=True, by_particle=False).run()
    out["visc_ts_nobp"] = h.results.timeseries
    # the collective itself, on a device tensor
    t = torch.arange(5, dtype=torch.float64, device="cuda")
    out["reduced"] = tad.reduce_lagsum(t, 2, force=True).cpu().numpy()
    # ... and without the request a one-rank group launches nothing: a CPU tensor stays legal under nccl
    os.environ.pop("TA_AMD_FORCE_COLLECTIVE")
    out["cpu_noop"] = tad.reduce_lagsum(torch.arange(5, dtype=torch.float64), 2).numpy()
    np.savez(os.path.join(out_dir, "nccl_0.npz"), **out)
    dist.destroy_process_group()


@pytest.mark.parametrize("T,A", [(700, 9), (1500, 4)])
def test_classes_distributed_rccl_world_size_one(tmp_path, T, A):
    """The nccl branch of both classes on one GPU (world size 1): RCCL is loaded and runs the
    all-reduce; results against the oracle to the north-star tolerance."""
    import torch.multiprocessing as mp

    from oracle import numpy_oracle as orc

    port = 36500 + (os.getpid() % 2000) + A
    mp.spawn(_worker_nccl, args=(1, port, T, A, str(tmp_path)), nprocs=1, join=True)
    z = np.load(tmp_path / "nccl_0.npz", allow_pickle=True)
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=77)
    v32 = v.astype(np.float32).astype(np.float64)
    x32 = x.astype(np.float32).astype(np.float64)
    want_bp, want_ts = orc.vacf_fft_batched(v32)
    hbp, hts = orc.helfand(v32, x32, m, np.full(T, 60.0**3), 300.0)
    assert tuple(z["range"]) == (0, A)
    for fft in (0, 1):
        for byp in (0, 1):
            assert scale_rel_err(z[f"vacf_ts_{fft}_{byp}"], want_ts) < TOL
        assert scale_rel_err(z[f"vacf_bp_{fft}"], want_bp) < TOL
    for tag, tol in (("f64", TOL), ("fft", TOL), ("f32", 2e-6)):
        assert scale_rel_err(z[f"visc_ts_{tag}"], hts) < tol
        assert scale_rel_err(z[f"visc_bp_{tag}"], hbp) < tol
    assert scale_rel_err(z["visc_ts_nobp"], hts) < TOL
    assert np.array_equal(z["reduced"], np.arange(5) / 2.0) and np.array_equal(z["cpu_noop"], np.arange(5) / 2.0)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_rehearsal_on_one_gpu(scaling):
    """bench.py's N > 1 path as the driver launches it (torch.distributed.run, one JSON line from
    rank 0, value = whole-job lag-points over the max-over-ranks time), rehearsed with both ranks
    on this box's one GPU (TA_BENCH_ONE_GPU=1: gloo instead of RCCL)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TA_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    port = 34500 + (os.getpid() % 1000) + (1 if scaling == "weak" else 2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "2000", "--atoms", "3000",
           "--scaling", scaling]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    total = 6000 if scaling == "weak" else 3000
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["config"]["n_atoms_total"] == total
    assert d["value"] == pytest.approx(2000 * total / (d["ms_per_step"] * 1e-3), rel=1e-9)
    assert d["roofline"]["frac"] > 0 and d["config"]["sharding"] == "atoms x2"
    assert len(d["config"]["rank_devices"]) == 2 and d["reduce_us"] > 0
    assert d["config"]["collective"]["ranks"] == 2 and d["config"]["collective"]["backend"] == "gloo"
    assert [r["rank"] for r in d["roofline"]["per_rank"]] == [0, 1] and d["cpu_baseline"]["value"] > 0


def test_bench_bare_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the shape of the driver's N = 1 command): the
    process starts `python -m torch.distributed.run` as a child before it touches a GPU, lets rank 0's
    ONE JSON line through and exits with the child's status.  Rehearsed with both ranks on this box's
    one GPU (TA_BENCH_ONE_GPU=1).  The line is complete: `roofline` (slowest rank's kernel, the reduce
    apart), `cpu_baseline`, and the collective with the rank count it saw."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["TA_BENCH_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--frames", "2000", "--atoms", "3000", "--cpu-sample-atoms", "200"]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["n_atoms_total"] == 6000 and d["scaling"] == "weak"
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and len(r["per_rank"]) == 2
    assert r["achieved"] == pytest.approx(min(x["GBps"] for x in r["per_rank"]))
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    c = d["config"]["collective"]
    assert c["ranks"] == 2 and "bench.py started" in c["launcher"]
    # a failing child is a failing command (here: more ranks asked for than --gpus says)
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "fft", "--float32"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


def test_bench_one_rank_under_rccl():
    """bench.py launched the way the driver launches N > 1, with ONE rank: the nccl (RCCL) branch
    of the step -- device lag sums all-reduced on the device -- runs and is timed (`reduce_us`)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TA_BENCH_FORCE_DIST="1")
    port = 35500 + (os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--frames", "2000", "--atoms", "3000",
           "--no-cpu-baseline", "--no-other-configs", "--no-host-path"]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["collective"]["backend"] == "nccl" and d["config"]["collective"]["ranks"] == 1
    assert len(d["config"]["rank_devices"]) == 1
    assert d["reduce_us"] > 0 and d["check"]["max_scale_rel_err_vs_torch_lags"] < 1e-10


GROUP_DRIVER = r"""
import sys, numpy as np
sys.path.insert(0, {repo!r})
from transport_analysis_amd import _lib          # torch is NOT imported in this process
from oracle import numpy_oracle as orc
rng = np.random.default_rng(3)
T, A, D = 300, 41, 3
v = rng.standard_normal((T, A, D))
bp_ref, ts_ref = orc.vacf_windowed(v)
def run(devices):
    g = _lib.Group(devices)
    (views,) = g.stage_alloc(T, A, D, n_slabs=1, dtype=np.float64)
    for view, (lo, hi) in zip(views, g.shards):
        if hi > lo:
            view[...] = v[:, lo:hi]
    g.stage_commit(0, T)
    ts, bp = g.vacf_fft(by_particle=True)
    kind = g.reduce_kind
    ts2, _ = g.vacf_direct(by_particle=False)
    g.close()
    return ts, bp, ts2, kind
ts1, bp1, d1, k1 = run([0])
maps = open("/proc/self/maps").read()
assert k1 == "none" and "librccl" not in maps, "one device: no reduce, RCCL must not be loaded"
c = _lib.Context(0)
(slab,) = c.stage_alloc(T, A, D, n_slabs=1, dtype=np.float64)
slab[...] = v
c.stage_commit(0, T)
ts0, bp0 = c.vacf_fft(by_particle=True)
assert np.array_equal(ts0, ts1) and np.array_equal(bp0, bp1), "devices=[0] must equal the plain context bit for bit"
ts2, bp2, d2, k2 = run([0, 0])
ts3, bp3, d3, k3 = run([0, 0, 0, 0, 0])
assert k2 == "peer-copy" and k3 == "peer-copy"
np.savez({out!r}, ts_ref=ts_ref, bp_ref=bp_ref, ts1=ts1, bp1=bp1, d1=d1, ts2=ts2, bp2=bp2, d2=d2, ts3=ts3, bp3=bp3, d3=d3)
"""


def test_group_c_abi_one_gpu(tmp_path):
    """The in-library fan-out on what a one-GPU box can run: a group of ONE device adds nothing to
    the plain context (bit-equal, no reduce, librccl never loaded); groups of 2 and 5 members that
    share the GPU stage disjoint column blocks, reduce by copy-and-add inside the call and fill
    the column ranges of one by-particle array.  (Distinct devices take the RCCL branch: not
    runnable here -- the driver's multi-GPU node is where it first executes.)"""
    import subprocess
    import sys

    out = tmp_path / "g.npz"
    (tmp_path / "drv.py").write_text(GROUP_DRIVER.format(repo=REPO, out=str(out)))
    r = subprocess.run([sys.executable, str(tmp_path / "drv.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(out)
    for k in ("1", "2", "3"):
        assert scale_rel_err(z["ts" + k], z["ts_ref"]) < TOL
        assert scale_rel_err(z["bp" + k], z["bp_ref"]) < TOL
        assert scale_rel_err(z["d" + k], z["ts_ref"]) < TOL


FORCED_RCCL_DRIVER = r"""
import sys, numpy as np
sys.path.insert(0, {repo!r})
from transport_analysis_amd import _lib          # torch is NOT imported in this process
rng = np.random.default_rng(5)
T, A, D = 900, 37, 3
v = rng.standard_normal((T, A, D))
c = _lib.Context(0)
(slab,) = c.stage_alloc(T, A, D, n_slabs=1, dtype=np.float64)
slab[...] = v
c.stage_commit(0, T)
ts0, bp0 = c.vacf_fft(by_particle=True)
d0, _ = c.vacf_direct(by_particle=False)
f0, _ = c.vacf_fft(by_particle=False)
g = _lib.Group([0])
g.set_option("force_rccl", 1)
(views,) = g.stage_alloc(T, A, D, n_slabs=1, dtype=np.float64)
views[0][...] = v
g.stage_commit(0, T)
ts1, bp1 = g.vacf_fft(by_particle=True)
kind, ranks = g.reduce_kind, g.rccl_ranks
d1, _ = g.vacf_direct(by_particle=False)
assert "librccl" in open("/proc/self/maps").read(), "the reduce must have loaded librccl"
assert kind == "rccl" and g.reduce_kind == "rccl" and ranks == 1, (kind, ranks, g.reduce_note)
assert np.array_equal(ts0, ts1) and np.array_equal(bp0, bp1) and np.array_equal(d0, d1), "sum over one member = that member"
# back to the automatic choice: one member reduces nothing
g.set_option("reduce_mode", 0)
ts2, _ = g.vacf_fft(by_particle=False)
assert g.reduce_kind == "none" and np.array_equal(ts2, f0)
g.close()
# members that share a device cannot form a communicator: RCCL-or-error must be the error
h = _lib.Group([0, 0])
h.set_option("reduce_mode", 2)
(views,) = h.stage_alloc(T, A, D, n_slabs=1, dtype=np.float64)
for view, (lo, hi) in zip(views, h.shards):
    view[...] = v[:, lo:hi]
h.stage_commit(0, T)
try:
    h.vacf_fft(by_particle=False)
except _lib.TAError as e:
    assert "rccl" in str(e)
else:
    raise AssertionError("reduce_mode 2 on a shared device must fail")
h.set_option("reduce_mode", 1)
ts3, _ = h.vacf_fft(by_particle=False)
assert h.reduce_kind == "peer-copy" and np.max(np.abs(ts3 - f0)) <= 1e-12 * np.max(np.abs(f0))
h.close()
print("ok")
"""


def test_group_forced_rccl_one_member(tmp_path):
    """The in-library RCCL branch EXECUTES: "force_rccl" on a one-member group runs ncclCommInitAll(1)
    and ncclReduce through the dlopen'ed, header-typed entry points (group.hip) and leaves the member's
    sums as they were -- bit-equal to the plain context -- with ta_group_reduce_kind == "rccl" and a
    communicator of one rank.  (Several distinct devices: only the driver's multi-GPU node can run it.)"""
    import subprocess
    import sys

    (tmp_path / "drv.py").write_text(FORCED_RCCL_DRIVER.format(repo=REPO))
    r = subprocess.run([sys.executable, str(tmp_path / "drv.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


@pytest.mark.parametrize("kind", ["fft", "direct", "helfand"])
def test_group_blocked_by_particle_copies_into_column_ranges(kind):
    """Members that process their atoms in blocks (bp_block: a block's device->host copy under the
    next block's compute) write every block into the right columns of the caller's ONE array: the
    host row stride is the full atom count, not the member's."""
    from oracle import numpy_oracle as orc
    from transport_analysis_amd import _lib

    T, A, D = 200, 700, 3
    v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=21)
    g = _lib.Group([0, 0, 0])
    g.set_option("bp_block", 64)  # members hold 233-234 atoms: four blocks each
    slabs = g.stage_alloc(T, A, D, n_slabs=2 if kind == "helfand" else 1, dtype=np.float64)
    for view, (lo, hi) in zip(slabs[0], g.shards):
        view[...] = v[:, lo:hi]
    if kind == "helfand":
        for view, (lo, hi) in zip(slabs[1], g.shards):
            view[...] = x[:, lo:hi]
    g.stage_commit(0, T)
    out = np.full((T, A), np.nan)
    if kind == "helfand":
        scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
        ts, bp = g.helfand_msd(m, scale, by_particle=True, out=out)
        want_bp, want_ts = orc.helfand(v, x, m, vol, 300.0)
    else:
        ts, bp = (g.vacf_fft if kind == "fft" else g.vacf_direct)(by_particle=True, out=out)
        want_bp, want_ts = orc.vacf_windowed(v)
    assert bp is out and not np.isnan(out).any() and g.reduce_kind == "peer-copy"
    assert scale_rel_err(ts, want_ts) < TOL
    assert scale_rel_err(out, want_bp) < TOL
    g.close()


@pytest.mark.parametrize("kind", ["direct", "helfand"])
def test_group_lag_sums_on_the_matrix_cores(kind):
    """Lag sums alone through a device group: every member runs the matrix-core kernel
    (band_kernels.hpp) on its own column block — shards of different widths, an odd column count in the
    last one — and the in-library reduce adds the members' lag sums."""
    from oracle import numpy_oracle as orc
    from transport_analysis_amd import _lib

    T, A, D = 700, 101, 3
    v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=23)
    g = _lib.Group([0, 0, 0])
    slabs = g.stage_alloc(T, A, D, n_slabs=2 if kind == "helfand" else 1, dtype=np.float64)
    for view, (lo, hi) in zip(slabs[0], g.shards):
        view[...] = v[:, lo:hi]
    if kind == "helfand":
        for view, (lo, hi) in zip(slabs[1], g.shards):
            view[...] = x[:, lo:hi]
    g.stage_commit(0, T)
    if kind == "helfand":
        scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
        ts, bp = g.helfand_msd(m, scale, by_particle=False)
        want_ts = orc.helfand(v, x, m, vol, 300.0)[1]
    else:
        ts, bp = g.vacf_direct(by_particle=False)
        want_ts = orc.vacf_windowed(v)[1]
    assert bp is None and g.reduce_kind == "peer-copy"
    assert scale_rel_err(ts, want_ts) < TOL
    g.close()


def test_bench_single_process_line_is_complete():
    """`bench.py --single-process` (the library's own fan-out and reduce, ta_group) prints the same shape
    of line as every other mode: `roofline` from the members' own kernel events, `cpu_baseline`, and the
    collective the library used.  Two members on this box's one GPU: the peer-copy reduce, by construction
    (distinct devices must reduce by RCCL or the run fails: bench.py single_process)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--single-process", "--gpus", "2", "--devices", "0,0",
           "--steps", "2", "--warmup", "1", "--frames", "2000", "--atoms", "3000", "--cpu-sample-atoms", "200"]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    r = d["roofline"]
    assert d["n_gpus"] == 2 and r["bound"] == "hbm" and 0 < r["frac"] < 1 and len(r["per_member"]) == 2
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9)
    assert d["config"]["collective"]["kind"] == "peer-copy" and d["config"]["collective"]["members"] == 2
    assert d["cpu_baseline"]["value"] > 0 and d["check"]["lag0_vs_numpy_block_mean_square"]


@pytest.mark.parametrize("A", [37, 5])
def test_group_of_eight_members_on_one_device(A):
    """The in-library fan-out at the member count of the driver's node, on what a one-GPU box can run: EIGHT members
    that share device 0 (peer-copy reduce in member order).  device_ranges = atom_shard over 8 (A = 5: three members
    hold no atom), the by-particle column ranges tile the array, and the lag sums equal the single context's to
    rounding (the sum is re-associated across members) -- and bit for bit from call to call."""
    from transport_analysis_amd import _lib
    from transport_analysis_amd.dist import atom_shard

    rng = np.random.default_rng(8)
    T, D = 1100, 3
    v = rng.standard_normal((T, A, D))
    c = _lib.Context(0)
    g = _lib.Group([0] * 8)
    try:
        (slab,) = c.stage_alloc(T, A, D, n_slabs=1, dtype=np.float64)
        slab[...] = v
        c.stage_commit(0, T)
        ts0, bp0 = c.vacf_fft(by_particle=True)
        (views,) = g.stage_alloc(T, A, D, n_slabs=1, dtype=np.float64)
        assert g.shards == [atom_shard(A, i, 8) for i in range(8)]
        for view, (lo, hi) in zip(views, g.shards):
            assert (view is None) == (hi == lo)
            if view is not None:
                view[...] = v[:, lo:hi]
        g.stage_commit(0, T)
        ts1, bp1 = g.vacf_fft(by_particle=True)
        assert g.reduce_kind == "peer-copy"
        # (a member's block starts on its own column-pair boundary: which columns share a complex transform differs from
        # the single context's, hence rounding-level differences per particle too)
        assert scale_rel_err(bp1, bp0) < 1e-13
        assert scale_rel_err(ts1, ts0) < 1e-14
        ts2, bp2 = g.vacf_fft(by_particle=True)
        assert np.array_equal(ts1, ts2) and np.array_equal(bp1, bp2)
        d1, _ = g.vacf_direct(by_particle=False)
        d0, _ = c.vacf_direct(by_particle=False)
        assert scale_rel_err(d1, d0) < 1e-13
    finally:
        g.close()
        c.close()


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_four_ranks_rehearsal_on_one_gpu(scaling):
    """bench.py's N > 1 path with FOUR ranks on this box's one GPU (TA_BENCH_ONE_GPU=1; a box allows six processes
    on its card, this test's own included): atom counts that do not divide (strong: 3001 over 4), one JSON line."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["TA_BENCH_ONE_GPU"] = "1"
    atoms = 1500 if scaling == "weak" else 3001
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
           "--frames", "2000", "--atoms", str(atoms), "--scaling", scaling, "--cpu-sample-atoms", "100"]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    total = 6000 if scaling == "weak" else 3001
    assert d["n_gpus"] == 4 and d["config"]["n_atoms_total"] == total and d["config"]["collective"]["ranks"] == 4
    assert sum(r["atoms"] for r in d["roofline"]["per_rank"]) == total and len(d["config"]["rank_devices"]) == 4
    assert d["value"] == pytest.approx(2000 * total / (d["ms_per_step"] * 1e-3), rel=1e-9)
