/* Plain-C restatement of the reference's time-correlation arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ as a second checker and by
 * bench.py's cpu_baseline leg (kind "port").  Never linked into the product.
 *
 * Follows, with the same loop structure and evaluation order:
 *   oracle_vacf_windowed : transport_analysis/velocityautocorr.py:217-238
 *   oracle_vacf_fft      : transport_analysis/velocityautocorr.py:208-215 +
 *                          tidynamics.acf (third party, restated; see
 *                          oracle/numpy_oracle.py for the pinning story)
 *   oracle_helfand       : transport_analysis/viscosity.py:201-233
 *
 * Layouts are the reference's: slabs (T, A, D) row-major float64, outputs
 * by_particle (T, A) row-major and timeseries (T,).  n_threads > 1 splits the
 * atom loop with OpenMP (the reference itself is single-threaded).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void atom_mean(const double *bp, long T, long A, double *ts) {
    for (long k = 0; k < T; ++k) {
        double s = 0.0;
        const double *row = bp + k * A;
        for (long n = 0; n < A; ++n) s += row[n];
        ts[k] = s / (double)A;
    }
}

/* velocityautocorr.py:223-235: for each lag, mean over the T-lag frame pairs of
 * the dot product over dims. */
int oracle_vacf_windowed(const double *v, long T, long A, long D, double *by_particle,
                         double *timeseries, int n_threads) {
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 8) num_threads(n_threads > 0 ? n_threads : 1)
    for (long n = 0; n < A; ++n) {
        for (long lag = 0; lag < T; ++lag) {
            double s = 0.0;
            for (long i = 0; i + lag < T; ++i) {
                const double *a = v + (i * A + n) * D;
                const double *b = v + ((i + lag) * A + n) * D;
                double dot = 0.0;
                for (long d = 0; d < D; ++d) dot += a[d] * b[d];
                s += dot;
            }
            by_particle[lag * A + n] = s / (double)(T - lag);
        }
    }
    atom_mean(by_particle, T, A, timeseries);
    return 0;
}

/* ---- radix-2 complex FFT (in place, iterative), sign = -1 forward, +1 inverse */
static void fft_radix2(double *re, double *im, long n, int sign) {
    for (long i = 1, j = 0; i < n; ++i) {
        long bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (long len = 2; len <= n; len <<= 1) {
        long half = len >> 1;
        for (long k = 0; k < half; ++k) {
            double ang = sign * 2.0 * M_PI * (double)k / (double)len;
            double wr = cos(ang), wi = sin(ang);
            for (long s = k; s < n; s += len) {
                long t = s + half;
                double xr = re[t] * wr - im[t] * wi;
                double xi = re[t] * wi + im[t] * wr;
                re[t] = re[s] - xr; im[t] = im[s] - xi;
                re[s] += xr;        im[s] += xi;
            }
        }
    }
}

static long tidynamics_n_fft(long n) {
    long p = 1;
    while (p < n + 1) p <<= 1; /* 2**ceil(log2(n+1)) */
    return p;                  /* n < p always holds here */
}

/* tidynamics autocorrelation_1d: zero-pad to 2*n_fft, full complex FFT,
 * |F|^2, inverse FFT, first N real parts divided by (N - lag); acf() sums the
 * per-column results. */
int oracle_vacf_fft(const double *v, long T, long A, long D, double *by_particle,
                    double *timeseries, int n_threads) {
    (void)n_threads;
    long L = 2 * tidynamics_n_fft(T);
    int fail = 0;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
        double *re = (double *)malloc(sizeof(double) * L);
        double *im = (double *)malloc(sizeof(double) * L);
        double *acc = (double *)malloc(sizeof(double) * T);
        if (!re || !im || !acc) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 8)
            for (long n = 0; n < A; ++n) {
                for (long k = 0; k < T; ++k) acc[k] = 0.0;
                for (long d = 0; d < D; ++d) {
                    memset(re, 0, sizeof(double) * L);
                    memset(im, 0, sizeof(double) * L);
                    for (long i = 0; i < T; ++i) re[i] = v[(i * A + n) * D + d];
                    fft_radix2(re, im, L, -1);
                    for (long k = 0; k < L; ++k) {
                        re[k] = re[k] * re[k] + im[k] * im[k];
                        im[k] = 0.0;
                    }
                    fft_radix2(re, im, L, +1);
                    for (long k = 0; k < T; ++k)
                        acc[k] += (re[k] / (double)L) / (double)(T - k);
                }
                for (long k = 0; k < T; ++k) by_particle[k * A + n] = acc[k];
            }
        }
        free(re); free(im); free(acc);
    }
    if (fail) return -1;
    atom_mean(by_particle, T, A, timeseries);
    return 0;
}

/* Throughput variant of oracle_vacf_fft for bench.py's all-cores CPU line: same arithmetic per
 * column (tidynamics padding, full complex FFT, |F|^2, inverse, / (N - lag)), but only the
 * lag-indexed SUM over atoms is produced (no (T, A) array), atoms are taken in blocks whose
 * columns are first copied out of the (T, A, D) slab row by row (contiguous reads instead of
 * one cache miss per sample), and the twiddles come from a per-thread table.  */
static void fft_radix2_tab(double *re, double *im, long n, int sign, const double *cs) {
    for (long i = 1, j = 0; i < n; ++i) {
        long bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (long len = 2; len <= n; len <<= 1) {
        long half = len >> 1, step = n / len;
        for (long s0 = 0; s0 < n; s0 += len)
            for (long k = 0; k < half; ++k) {
                double wr = cs[2 * k * step], wi = sign * cs[2 * k * step + 1];
                long s = s0 + k, t = s + half;
                double xr = re[t] * wr - im[t] * wi;
                double xi = re[t] * wi + im[t] * wr;
                re[t] = re[s] - xr; im[t] = im[s] - xi;
                re[s] += xr;        im[s] += xi;
            }
    }
}

int oracle_vacf_fft_lagsum(const double *v, long T, long A, long D, double *lagsum,
                           int n_threads) {
    enum { BLK = 8 };
    long L = 2 * tidynamics_n_fft(T);
    int fail = 0;
    for (long k = 0; k < T; ++k) lagsum[k] = 0.0;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
        double *re = (double *)malloc(sizeof(double) * L);
        double *im = (double *)malloc(sizeof(double) * L);
        double *cs = (double *)malloc(sizeof(double) * L);          /* (cos, sin) of 2 pi k / L */
        double *cols = (double *)malloc(sizeof(double) * BLK * D * T);
        double *mine = (double *)calloc((size_t)T, sizeof(double));
        if (!re || !im || !cs || !cols || !mine) {
#pragma omp atomic write
            fail = 1;
        } else {
            for (long k = 0; k < L / 2; ++k) {
                cs[2 * k] = cos(2.0 * M_PI * (double)k / (double)L);
                cs[2 * k + 1] = sin(2.0 * M_PI * (double)k / (double)L);
            }
#pragma omp for schedule(dynamic, 1)
            for (long n0 = 0; n0 < A; n0 += BLK) {
                long nb = A - n0 < BLK ? A - n0 : BLK, w = nb * D;
                for (long i = 0; i < T; ++i)
                    for (long c = 0; c < w; ++c) cols[c * T + i] = v[(i * A + n0) * D + c];
                for (long c = 0; c < w; ++c) {
                    memset(re, 0, sizeof(double) * L);
                    memset(im, 0, sizeof(double) * L);
                    memcpy(re, cols + c * T, sizeof(double) * T);
                    fft_radix2_tab(re, im, L, -1, cs);
                    for (long k = 0; k < L; ++k) {
                        re[k] = re[k] * re[k] + im[k] * im[k];
                        im[k] = 0.0;
                    }
                    fft_radix2_tab(re, im, L, +1, cs);
                    for (long k = 0; k < T; ++k) mine[k] += (re[k] / (double)L) / (double)(T - k);
                }
            }
#pragma omp critical
            for (long k = 0; k < T; ++k) lagsum[k] += mine[k];
        }
        free(re); free(im); free(cs); free(cols); free(mine);
    }
    return fail ? -1 : 0;
}

/* viscosity.py:205-233.  diff = (m*v)*x at frame i minus (m*v)*x at i+lag,
 * squared, MEAN over dims, mean over the T-lag frame pairs; lag 0 stays 0;
 * everything divided by 2*kB*mean(volumes)*temp_avg. */
int oracle_helfand(const double *v, const double *x, const double *masses,
                   const double *volumes, long T, long A, long D, double temp_avg,
                   double boltzmann, double *by_particle, double *timeseries,
                   int n_threads) {
    (void)n_threads;
    double vol = 0.0;
    for (long i = 0; i < T; ++i) vol += volumes[i];
    vol /= (double)T;
    double denom = 2 * boltzmann * vol * temp_avg;
#pragma omp parallel for schedule(dynamic, 8) num_threads(n_threads > 0 ? n_threads : 1)
    for (long n = 0; n < A; ++n) {
        double m = masses[n];
        by_particle[n] = 0.0 / denom;
        for (long lag = 1; lag < T; ++lag) {
            double s = 0.0;
            for (long i = 0; i + lag < T; ++i) {
                const double *va = v + (i * A + n) * D, *xa = x + (i * A + n) * D;
                const double *vb = v + ((i + lag) * A + n) * D, *xb = x + ((i + lag) * A + n) * D;
                double sq = 0.0;
                for (long d = 0; d < D; ++d) {
                    double diff = m * va[d] * xa[d] - m * vb[d] * xb[d];
                    sq += diff * diff;
                }
                s += sq / (double)D;
            }
            by_particle[lag * A + n] = (s / (double)(T - lag)) / denom;
        }
    }
    atom_mean(by_particle, T, A, timeseries);
    return 0;
}
