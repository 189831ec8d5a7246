"""Oracle-backed stand-in for transport_analysis_amd._lib.Context.

TEST INFRASTRUCTURE: lets the CPU-only suite exercise the host logic of the
analysis classes (frame slicing, staging order, results plumbing, fits,
post-processing) on machines without a GPU.  The product never uses it."""
import numpy as np

from oracle import numpy_oracle as orc


class OracleContext:
    def __init__(self, device=0):
        self.device = device
        self.commits = []

    def stage_alloc(self, n_frames, n_atoms, dim, n_slabs=1, dtype=np.float64):
        self.shape = (n_frames, n_atoms, dim)
        self.slabs = [np.zeros(self.shape, dtype=dtype) for _ in range(n_slabs)]
        self.commits = []
        return self.slabs

    def stage_commit(self, lo, hi):
        assert 0 <= lo <= hi <= self.shape[0]
        self.commits.append((lo, hi))

    def _covered(self):
        got = sorted(self.commits)
        pos = 0
        for lo, hi in got:
            assert lo == pos, f"staging gap or overlap at frame {pos}: {got}"
            pos = hi
        assert pos == self.shape[0], "not every frame was committed to the device"

    class _Home:
        def __init__(self, shape):
            self.arr = np.empty(shape)

        def get(self):
            return self.arr

    def result_home(self, shape):
        return self._Home(shape)

    device = 0

    @staticmethod
    def _out(bp, by_particle, out):
        if out is not None:
            out[...] = bp
            return out
        return bp if by_particle else None

    def vacf_fft(self, by_particle=False, out=None):
        self._covered()
        bp, ts = orc.vacf_fft_batched(self.slabs[0])
        return ts, self._out(bp, by_particle, out)

    def vacf_direct(self, by_particle=False, out=None):
        self._covered()
        bp, ts = orc.vacf_windowed(self.slabs[0])
        return ts, self._out(bp, by_particle, out)

    def helfand_msd(self, masses, scale, by_particle=False, out=None):
        self._covered()
        T = self.shape[0]
        # scale = 1/(2 kB <V> T): feed the oracle a unit denominator and rescale
        bp, ts = orc.helfand(self.slabs[0], self.slabs[1], masses, np.ones(T), temp_avg=1.0,
                             boltzmann=0.5)
        bp, ts = bp * scale, ts * scale
        return ts, self._out(bp, by_particle, out)

    def set_option(self, key, value):
        self.options = getattr(self, 'options', {})
        self.options[key] = int(value)

    def close(self):
        pass


class OracleGroup:
    """Stand-in for transport_analysis_amd._lib.Group: the same partition of the atoms over the
    `devices` (floor(A i / n) boundaries, as include/ta_hip.h:ta_group_shard), one oracle-backed
    member per device, the members' lag SUMS added and divided by the total atom count, and
    by-particle blocks written into the column ranges of one array."""

    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        self.device = self.devices[0]
        self.members = [OracleContext(d) for d in self.devices]
        self.shards = []
        self.reduce_kind = "none"

    def shard(self, n_atoms, i):
        n = len(self.devices)
        return n_atoms * i // n, n_atoms * (i + 1) // n

    def stage_alloc(self, n_frames, n_atoms, dim, n_slabs=1, dtype=np.float64):
        self.shape = (n_frames, n_atoms, dim)
        self.shards = [self.shard(n_atoms, i) for i in range(len(self.devices))]
        per_member = [m.stage_alloc(n_frames, hi - lo, dim, n_slabs, dtype) if hi > lo else [None] * n_slabs
                      for m, (lo, hi) in zip(self.members, self.shards)]
        return [[pm[s] for pm in per_member] for s in range(n_slabs)]

    def stage_commit(self, lo, hi):
        for m, (a, b) in zip(self.members, self.shards):
            if b > a:
                m.stage_commit(lo, hi)

    def result_home(self, shape):
        return OracleContext._Home(shape)

    def set_option(self, key, value):
        for m in self.members:
            m.set_option(key, value)

    def close(self):
        pass

    def _gather(self, call, by_particle, out):
        T, A, _ = self.shape
        total = np.zeros(T)
        bp = out if out is not None else (np.empty((T, A)) if by_particle else None)
        active = 0
        for m, (lo, hi) in zip(self.members, self.shards):
            if hi == lo:
                continue
            ts_i, bp_i = call(m, lo, hi)
            total += ts_i * (hi - lo)  # the member's mean -> its sum
            if bp is not None:
                bp[:, lo:hi] = bp_i
            active += 1
        self.reduce_kind = "none" if active <= 1 else "peer-copy"
        return total / A, bp

    def vacf_fft(self, by_particle=False, out=None):
        return self._gather(lambda m, lo, hi: m.vacf_fft(by_particle=True), by_particle, out)

    def vacf_direct(self, by_particle=False, out=None):
        return self._gather(lambda m, lo, hi: m.vacf_direct(by_particle=True), by_particle, out)

    def helfand_msd(self, masses, scale, by_particle=False, out=None):
        masses = np.asarray(masses)
        return self._gather(lambda m, lo, hi: m.helfand_msd(masses[lo:hi], scale, by_particle=True),
                            by_particle, out)
