// cpu_backend.cpp — the CPU backend behind the C-ABI ("a CPU backend behind the same
// symbols for parity / CI"; BASELINE.md section 4: "the build's own C++/OpenMP CPU backend on all host cores").
//
// STRICTLY OPT-IN: a context created with ta_ctx_create(TA_DEVICE_CPU, ...) computes here; every other context is
// a GPU context and fails loudly without a GPU.  Nothing in this file is reached from a GPU context, and nothing
// here uses oracle/ (test infrastructure) -- this is independent code: its own Stockham radix-4 transform, the
// windowed form, the difference-first Einstein-Helfand form.
//
// What it computes, per atom n of the staged (n_frames, n_atoms, dim) slabs (float32 or float64 elements, widened
// exactly), in float64:
//   FFT VACF      velocityautocorr.py:208-215 (+ tidynamics.acf): B[k, n] = sum_d sum_i v[i,n,d] v[i+k,n,d] / (T - k) by
//                 zero-padded transforms of length L = the power of two >= 2T (any pad >= 2T - 1 gives the same
//                 correlation); two columns ride one complex transform (z = x + i y: |Z|^2 transforms back to
//                 acf_x + acf_y), two atoms' power spectra one inverse (both real: untangled by the mirror bins);
//   windowed VACF velocityautocorr.py:217-238: the same sums, lag by lag, products then sum (k = 0 .. T - 1);
//   Helfand       viscosity.py:201-233: P = (m v) x, H[k, n] = scale / D / (T - k) sum_i sum_d (P[i] - P[i+k])^2,
//                 difference first, like the reference; H[0, n] = 0.
// timeseries[k] = sum over atoms (the caller divides by n_atoms, as for the GPU path).  Atoms are processed in
// blocks of 8 by OpenMP threads; a block's lag sums are added in block order afterwards, so results do not depend on
// the number of threads.
#include <omp.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/ta_hip.h"
#include "cpu_backend.hpp"

namespace ta {
namespace cpu {
namespace {

constexpr int kBlock = 8;  // atoms per task: a block's by-particle values are written as rows of 8 doubles (one line)

// ---- power-of-two complex transform, split format, Stockham autosort, radix 4 (+ one radix-2 stage) -------------
// forward: X[k] = sum_n x[n] exp(-2 pi i n k / N)
struct Plan {
    int n = 0;
    // per radix-4 stage (sub-length m = n, n/4, ...): w1 = W_m^p, w2 = W_m^2p, w3 = W_m^3p, p < m / 4
    std::vector<double> tw;
    std::vector<size_t> off;  // offset of each stage's tables in tw
    explicit Plan(int n_) : n(n_) {
        const long double pi = 3.141592653589793238462643383279502884L;
        for (int m = n; m >= 4; m /= 4) {
            off.push_back(tw.size());
            const int q = m / 4;
            tw.resize(tw.size() + 6 * (size_t)q);
            double* t = tw.data() + off.back();
            for (int p = 0; p < q; ++p)
                for (int k = 1; k <= 3; ++k) {
                    const long double a = 2.0L * pi * (long double)(k * p) / (long double)m;
                    t[(2 * (k - 1)) * q + p] = (double)cosl(a);
                    t[(2 * (k - 1) + 1) * q + p] = (double)-sinl(a);
                }
        }
    }
};

// one radix-4 stage: x (sub-length m, stride s) -> y (sub-length m / 4, stride 4 s)
inline void stage4(int m, int s, const double* __restrict__ t, const double* __restrict__ xr, const double* __restrict__ xi,
                   double* __restrict__ yr, double* __restrict__ yi) {
    const int q4 = m / 4;
    const double *w1r = t, *w1i = t + q4, *w2r = t + 2 * q4, *w2i = t + 3 * q4, *w3r = t + 4 * q4, *w3i = t + 5 * q4;
    if (s == 1) {
        for (int p = 0; p < q4; ++p) {
            const double ar = xr[p], ai = xi[p], br = xr[p + q4], bi = xi[p + q4];
            const double cr = xr[p + 2 * q4], ci = xi[p + 2 * q4], dr = xr[p + 3 * q4], di = xi[p + 3 * q4];
            const double apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;
            const double bpdr = br + dr, bpdi = bi + di;
            const double jr = -(bi - di), ji = br - dr;  // i (b - d)
            const double t1r = amcr - jr, t1i = amci - ji, t2r = apcr - bpdr, t2i = apci - bpdi;
            const double t3r = amcr + jr, t3i = amci + ji;
            yr[4 * p] = apcr + bpdr;
            yi[4 * p] = apci + bpdi;
            yr[4 * p + 1] = w1r[p] * t1r - w1i[p] * t1i;
            yi[4 * p + 1] = w1r[p] * t1i + w1i[p] * t1r;
            yr[4 * p + 2] = w2r[p] * t2r - w2i[p] * t2i;
            yi[4 * p + 2] = w2r[p] * t2i + w2i[p] * t2r;
            yr[4 * p + 3] = w3r[p] * t3r - w3i[p] * t3i;
            yi[4 * p + 3] = w3r[p] * t3i + w3i[p] * t3r;
        }
        return;
    }
    for (int p = 0; p < q4; ++p) {
        const double a1r = w1r[p], a1i = w1i[p], a2r = w2r[p], a2i = w2i[p], a3r = w3r[p], a3i = w3i[p];
        const double *x0r = xr + (size_t)s * p, *x0i = xi + (size_t)s * p;
        const double *x1r = x0r + (size_t)s * q4, *x1i = x0i + (size_t)s * q4;
        const double *x2r = x1r + (size_t)s * q4, *x2i = x1i + (size_t)s * q4;
        const double *x3r = x2r + (size_t)s * q4, *x3i = x2i + (size_t)s * q4;
        double *y0r = yr + (size_t)s * 4 * p, *y0i = yi + (size_t)s * 4 * p;
        double *y1r = y0r + s, *y1i = y0i + s, *y2r = y1r + s, *y2i = y1i + s, *y3r = y2r + s, *y3i = y2i + s;
#pragma omp simd
        for (int q = 0; q < s; ++q) {
            const double ar = x0r[q], ai = x0i[q], br = x1r[q], bi = x1i[q];
            const double cr = x2r[q], ci = x2i[q], dr = x3r[q], di = x3i[q];
            const double apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;
            const double bpdr = br + dr, bpdi = bi + di;
            const double jr = -(bi - di), ji = br - dr;
            const double t1r = amcr - jr, t1i = amci - ji, t2r = apcr - bpdr, t2i = apci - bpdi;
            const double t3r = amcr + jr, t3i = amci + ji;
            y0r[q] = apcr + bpdr;
            y0i[q] = apci + bpdi;
            y1r[q] = a1r * t1r - a1i * t1i;
            y1i[q] = a1r * t1i + a1i * t1r;
            y2r[q] = a2r * t2r - a2i * t2i;
            y2i[q] = a2r * t2i + a2i * t2r;
            y3r[q] = a3r * t3r - a3i * t3i;
            y3i[q] = a3r * t3i + a3i * t3r;
        }
    }
}

// in: (ar, ai); scratch (br, bi); returns the pair of pointers that hold the result
struct Split {
    double *r, *i;
};
inline Split fft(const Plan& pl, double* ar, double* ai, double* br, double* bi) {
    int m = pl.n, s = 1;
    size_t st = 0;
    double *xr = ar, *xi = ai, *yr = br, *yi = bi;
    for (; m >= 4; m /= 4, s *= 4, ++st) {
        stage4(m, s, pl.tw.data() + pl.off[st], xr, xi, yr, yi);
        std::swap(xr, yr);
        std::swap(xi, yi);
    }
    if (m == 2) {  // the last radix-2 stage (log2 n odd): no twiddles
#pragma omp simd
        for (int q = 0; q < s; ++q) {
            const double a_r = xr[q], a_i = xi[q], b_r = xr[q + s], b_i = xi[q + s];
            yr[q] = a_r + b_r;
            yi[q] = a_i + b_i;
            yr[q + s] = a_r - b_r;
            yi[q + s] = a_i - b_i;
        }
        std::swap(xr, yr);
        std::swap(xi, yi);
    }
    return {xr, xi};
}

template <class E>
inline double elem(const void* slab, size_t idx) {
    return (double)static_cast<const E*>(slab)[idx];
}

struct Work {  // per-thread buffers
    std::vector<double> a, b, c, d, pr[2], cols, prod, out;
};

template <class E>
int vacf_fft_t(const State& s, double* ts, double* bp) {
    const int64_t T = s.T, A = s.A;
    const int D = s.D;
    int L = 2;
    while (L < 2 * T) L *= 2;
    const Plan plan(L);
    const int64_t n_blocks = (A + kBlock - 1) / kBlock;
    std::vector<double> part((size_t)n_blocks * T);
    const void* slab = s.slabs[0];
    bool oom = false;
#pragma omp parallel num_threads(s.threads)
    {
        Work w;
        try {
            for (auto* v : {&w.a, &w.b, &w.c, &w.d, &w.pr[0], &w.pr[1]}) v->assign((size_t)L, 0.0);
            w.out.assign((size_t)T * kBlock, 0.0);
        } catch (const std::bad_alloc&) {
#pragma omp atomic write
            oom = true;
        }
#pragma omp barrier
        if (!oom) {
#pragma omp for schedule(dynamic, 1)
            for (int64_t blk = 0; blk < n_blocks; ++blk) {
                const int64_t a0 = blk * kBlock, na = std::min<int64_t>(kBlock, A - a0);
                double* psum = part.data() + (size_t)blk * T;
                std::memset(psum, 0, sizeof(double) * T);
                for (int64_t j = 0; j < na; j += 2) {
                    const int n2 = j + 1 < na ? 2 : 1;
                    // the power spectra of one or two atoms
                    for (int h = 0; h < n2; ++h) {
                        const int64_t atom = a0 + j + h;
                        double* P = w.pr[h].data();
                        std::memset(P, 0, sizeof(double) * L);
                        for (int d0 = 0; d0 < D; d0 += 2) {
                            const bool two = d0 + 1 < D;
                            double *xr = w.a.data(), *xi = w.b.data();
                            for (int64_t t = 0; t < T; ++t) {
                                const size_t base = ((size_t)t * A + atom) * D + d0;
                                xr[t] = elem<E>(slab, base);
                                xi[t] = two ? elem<E>(slab, base + 1) : 0.0;
                            }
                            std::memset(xr + T, 0, sizeof(double) * (L - T));
                            std::memset(xi + T, 0, sizeof(double) * (L - T));
                            const Split z = fft(plan, xr, xi, w.c.data(), w.d.data());
#pragma omp simd
                            for (int k = 0; k < L; ++k) P[k] += z.r[k] * z.r[k] + z.i[k] * z.i[k];
                        }
                    }
                    // one transform of P_0 + i P_1 (both real: G[n] + conj G[L - n] = 2 F_0[n], the imaginary parts F_1)
                    double *gr = w.a.data(), *gi = w.b.data();
                    std::memcpy(gr, w.pr[0].data(), sizeof(double) * L);
                    if (n2 == 2) std::memcpy(gi, w.pr[1].data(), sizeof(double) * L);
                    else std::memset(gi, 0, sizeof(double) * L);
                    const Split g = fft(plan, gr, gi, w.c.data(), w.d.data());
                    for (int64_t n = 0; n < T; ++n) {
                        const int64_t mir = n == 0 ? 0 : L - n;
                        const double norm = (double)L * (double)(T - n);  // < 2^53: exact
                        const double f0 = 0.5 * (g.r[n] + g.r[mir]) / norm, f1 = 0.5 * (g.i[n] + g.i[mir]) / norm;
                        w.out[(size_t)n * kBlock + j] = f0;
                        if (n2 == 2) w.out[(size_t)n * kBlock + j + 1] = f1;
                    }
                }
                for (int64_t n = 0; n < T; ++n) {
                    double sum = 0.0;
                    const double* row = w.out.data() + (size_t)n * kBlock;
                    for (int64_t j = 0; j < na; ++j) sum += row[j];
                    psum[n] = sum;
                    if (bp) std::memcpy(bp + (size_t)n * A + a0, row, sizeof(double) * na);
                }
            }
        }
    }
    if (oom) return TA_E_NOMEM;
    for (int64_t n = 0; n < T; ++n) {
        double sum = 0.0;
        for (int64_t blk = 0; blk < n_blocks; ++blk) sum += part[(size_t)blk * T + n];
        ts[n] = sum;
    }
    return TA_OK;
}

// windowed VACF (helfand == false) and Einstein-Helfand (helfand == true) share the O(T^2) loop over lags
template <class E>
int direct_t(const State& s, bool helfand, const double* masses, double scale, double* ts, double* bp) {
    const int64_t T = s.T, A = s.A;
    const int D = s.D;
    const int64_t n_blocks = (A + kBlock - 1) / kBlock;
    std::vector<double> part((size_t)n_blocks * T);
    const void *vel = s.slabs[0], *pos = helfand ? s.slabs[1] : nullptr;
    bool oom = false;
#pragma omp parallel num_threads(s.threads)
    {
        Work w;
        try {
            w.cols.assign((size_t)T * D, 0.0);
            w.out.assign((size_t)T * kBlock, 0.0);
        } catch (const std::bad_alloc&) {
#pragma omp atomic write
            oom = true;
        }
#pragma omp barrier
        if (!oom) {
#pragma omp for schedule(dynamic, 1)
            for (int64_t blk = 0; blk < n_blocks; ++blk) {
                const int64_t a0 = blk * kBlock, na = std::min<int64_t>(kBlock, A - a0);
                double* psum = part.data() + (size_t)blk * T;
                for (int64_t j = 0; j < na; ++j) {
                    const int64_t atom = a0 + j;
                    // the atom's columns, contiguous in time: c[d][t] (Helfand: P = (m v) x, the reference's order)
                    double* c = w.cols.data();
                    for (int d = 0; d < D; ++d)
                        for (int64_t t = 0; t < T; ++t) {
                            const size_t idx = ((size_t)t * A + atom) * D + d;
                            c[(size_t)d * T + t] = helfand ? (masses[atom] * elem<E>(vel, idx)) * elem<E>(pos, idx) : elem<E>(vel, idx);
                        }
                    for (int64_t k = 0; k < T; ++k) {
                        double acc = 0.0;
                        if (!(helfand && k == 0)) {
                            for (int d = 0; d < D; ++d) {
                                const double *p0 = c + (size_t)d * T, *p1 = p0 + k;
                                double sd = 0.0;
                                if (helfand) {
#pragma omp simd reduction(+ : sd)
                                    for (int64_t i = 0; i < T - k; ++i) {
                                        const double df = p0[i] - p1[i];
                                        sd += df * df;
                                    }
                                } else {
#pragma omp simd reduction(+ : sd)
                                    for (int64_t i = 0; i < T - k; ++i) sd += p0[i] * p1[i];
                                }
                                acc += sd;
                            }
                            acc /= (double)(T - k);
                            if (helfand) acc = acc / (double)D * scale;  // mean over the columns (viscosity.py:222), then 1 / (2 kB V T)
                        }
                        w.out[(size_t)k * kBlock + j] = acc;
                    }
                }
                for (int64_t n = 0; n < T; ++n) {
                    double sum = 0.0;
                    const double* row = w.out.data() + (size_t)n * kBlock;
                    for (int64_t j = 0; j < na; ++j) sum += row[j];
                    psum[n] = sum;
                    if (bp) std::memcpy(bp + (size_t)n * A + a0, row, sizeof(double) * na);
                }
            }
        }
    }
    if (oom) return TA_E_NOMEM;
    for (int64_t n = 0; n < T; ++n) {
        double sum = 0.0;
        for (int64_t blk = 0; blk < n_blocks; ++blk) sum += part[(size_t)blk * T + n];
        ts[n] = sum;
    }
    return TA_OK;
}

}  // namespace

int hardware_threads() { return omp_get_max_threads(); }

// the benchmark generator of ta_stage_synth (layout.hip: synth_value; oracle/synth.py reproduces it): element (t, c) of the
// slab = value(seed, t n_cols_total + col_offset + c) -- integer arithmetic and ONE correctly rounded product: the same
// bits as on the GPU
static inline unsigned long long splitmix64(unsigned long long x) {
    unsigned long long z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double synth_value(unsigned long long seed, unsigned long long idx) {
    const unsigned long long a = splitmix64(seed + 2 * idx), b = splitmix64(seed + 2 * idx + 1);
    long s = 0;
    for (int k = 0; k < 4; ++k) s += (long)((a >> (16 * k)) & 0xFFFF) + (long)((b >> (16 * k)) & 0xFFFF);
    return (double)(s - 262140) * 0x1.3988e1412ed76p-16;
}
void synth(const State& s, int slab, unsigned long long seed, int64_t col_offset, int64_t n_cols_total) {
    const int64_t T = s.T, n_cols = s.A * s.D;
    void* p = s.slabs[slab];
#pragma omp parallel for num_threads(s.threads) schedule(static)
    for (int64_t t = 0; t < T; ++t)
        for (int64_t c = 0; c < n_cols; ++c) {
            const double v = synth_value(seed, (unsigned long long)(t * n_cols_total + col_offset + c));
            if (s.dtype == TA_F32) static_cast<float*>(p)[(size_t)t * n_cols + c] = (float)v;
            else static_cast<double*>(p)[(size_t)t * n_cols + c] = v;
        }
}

bool supported() { return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"); }

int vacf_fft(const State& s, double* ts, double* bp) {
    return s.dtype == TA_F32 ? vacf_fft_t<float>(s, ts, bp) : vacf_fft_t<double>(s, ts, bp);
}
int vacf_direct(const State& s, double* ts, double* bp) {
    return s.dtype == TA_F32 ? direct_t<float>(s, false, nullptr, 1.0, ts, bp) : direct_t<double>(s, false, nullptr, 1.0, ts, bp);
}
int helfand(const State& s, const double* masses, double scale, double* ts, double* bp) {
    return s.dtype == TA_F32 ? direct_t<float>(s, true, masses, scale, ts, bp) : direct_t<double>(s, true, masses, scale, ts, bp);
}

}  // namespace cpu
}  // namespace ta
