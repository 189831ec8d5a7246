"""Atom-sharded multi-GPU evaluation.

Every per-atom series depends on that atom's data only; the single cross-atom
operation of the path is the mean over atoms
(/root/reference/transport_analysis/velocityautocorr.py:214,237;
viscosity.py:233).  So each rank (one process per GPU) evaluates the
lag-indexed SUM over its contiguous block of atoms and ONE all-reduce of that
(n_frames,) float64 vector (RCCL over xGMI; 80 KB at 10 000 frames) followed by
a division by the total atom count gives ``results.timeseries`` on every rank.
``vacf_by_particle`` blocks stay on the rank that computed them.
"""
from __future__ import annotations

import numpy as np


def atom_shard(n_atoms, rank, world_size):
    """Contiguous atom range [lo, hi) of `rank`: floor(n*r/W) .. floor(n*(r+1)/W)."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    lo = (n_atoms * rank) // world_size
    hi = (n_atoms * (rank + 1)) // world_size
    return lo, hi


def _force_collective():
    import os

    return os.environ.get("TA_AMD_FORCE_COLLECTIVE", "") == "1" or os.environ.get("TA_BENCH_FORCE_DIST", "") == "1"


def reduce_lagsum(lagsum, n_atoms_total, group=None, force=False):
    """Sum the per-rank lag sums across ranks and divide by the total atom count.

    `lagsum` is a torch tensor (device tensor with the nccl/RCCL backend, CPU
    tensor with gloo); it is reduced in place and the mean is returned.  A group of one rank
    reduces nothing (no collective is launched, and a tensor the backend could not reduce --
    a CPU tensor under nccl -- stays legal) unless `force` or $TA_AMD_FORCE_COLLECTIVE=1 /
    $TA_BENCH_FORCE_DIST=1 asks for the collective anyway: the tests and the one-GPU rehearsal of
    bench.py use that to run RCCL on a box with a single GPU."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        if dist.get_world_size(group) > 1 or force or _force_collective():
            dist.all_reduce(lagsum, op=dist.ReduceOp.SUM, group=group)
    return lagsum / float(n_atoms_total)


def sharded_timeseries(compute_lagsum, n_atoms_total, rank, world_size, group=None):
    """Evaluate one rank's shard and reduce.

    compute_lagsum(lo, hi) -> torch tensor (n_frames,) holding
    sum_{n in [lo,hi)} by_particle[:, n] for this rank's atoms (the product path
    passes a closure over ``Context.vacf_fft_dev`` / ``vacf_direct_dev`` /
    ``helfand_msd_dev``; an empty shard must return zeros)."""
    lo, hi = atom_shard(n_atoms_total, rank, world_size)
    part = compute_lagsum(lo, hi)
    return reduce_lagsum(part, n_atoms_total, group)


def shard_of_this_rank(n_atoms):
    """(rank, world, lo, hi) of the calling process under torch.distributed (must be
    initialised): the contiguous block of atoms this process stages and correlates."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("distributed=True needs an initialised torch.distributed process group "
                           "(one process per GPU, e.g. under torchrun)")
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = atom_shard(n_atoms, rank, world)
    return rank, world, lo, hi


def default_device():
    """GPU index of the calling rank: $TA_AMD_DEVICE, else $LOCAL_RANK (torchrun: one process per
    GPU), else torch's current device when a process group is up, else 0."""
    import os

    for key in ("TA_AMD_DEVICE", "LOCAL_RANK"):
        if os.environ.get(key, "") != "":
            return int(os.environ[key])
    try:
        import torch
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and torch.cuda.is_available():
            return int(torch.cuda.current_device())
    except Exception:
        pass
    return 0


def uses_device_reduce():
    """True when the process group reduces device tensors (nccl = RCCL on ROCm)."""
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"


def staged_timeseries_on_device(ctx, which, n_frames, n_local, n_atoms_total, device, masses=None,
                                scale=1.0, by_particle=False):
    """This rank's staged block -> lag sums (device) -> ONE all-reduce over RCCL -> mean over
    ALL atoms.  The (n_frames,) float64 sums never leave the GPU before the reduce.
    which: "fft" | "direct" | "helfand".  Returns (timeseries ndarray, by_particle ndarray|None)."""
    import torch

    dev = torch.device("cuda", device)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        lag = torch.zeros(n_frames, dtype=torch.float64, device=dev)
        bp = torch.empty((n_frames, max(n_local, 1)), dtype=torch.float64, device=dev) if by_particle else None
        d_bp = bp.data_ptr() if bp is not None else 0
        if n_local:  # more ranks than atoms: this rank contributes zeros
            if which == "fft":
                ctx.vacf_fft_staged(lag.data_ptr(), d_bp, n_local, stream)
            elif which == "direct":
                ctx.vacf_direct_staged(lag.data_ptr(), d_bp, n_local, stream)
            else:
                m = torch.as_tensor(np.ascontiguousarray(masses, dtype=np.float64), device=dev)
                ctx.helfand_msd_staged(m.data_ptr(), float(scale), lag.data_ptr(), d_bp, n_local, stream)
        ts = reduce_lagsum(lag, n_atoms_total)
        out_bp = None
        if bp is not None:
            out_bp = bp.cpu().numpy() if n_local else np.zeros((n_frames, 0))
        return ts.cpu().numpy(), out_bp


def allreduce_mean_over_atoms(ts_local, n_local, n_atoms_total, device=None):
    """Turn this rank's mean over ITS atoms into the mean over ALL atoms: one all-reduce of the
    (n_frames,) float64 lag sums (RCCL when the group's backend is nccl, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    lagsum = torch.from_numpy(np.ascontiguousarray(ts_local, dtype=np.float64) * float(n_local))
    if dist.get_backend() == "nccl":
        lagsum = lagsum.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))
    out = reduce_lagsum(lagsum, n_atoms_total)
    return out.cpu().numpy()
