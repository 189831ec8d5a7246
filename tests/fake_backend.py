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
