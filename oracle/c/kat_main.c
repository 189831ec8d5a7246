/* Known-answer run of the plain-C oracle (ta_oracle.c), built with AddressSanitizer and
 * UndefinedBehaviorSanitizer by `make -C oracle/c asan` (sanitizers on the
 * CPU restatement; the GPU build is never sanitized on this pool).  TEST INFRASTRUCTURE ONLY.
 *
 *  1. step trajectory v[i, n, d] = i (the reference's KAT family,
 *     transport_analysis/tests/test_velocityautocorr.py:79-93): windowed and FFT VACF against the
 *     closed form  C(k) = D ((n-1) n (2n-1) / 6 + k n (n-1) / 2) / n,  n = T - k, in exact integers;
 *  2. ragged shapes (T not a power of two, A = 1, D = 1..3) on pseudo-random data: FFT == windowed,
 *     the lag-sum variant == atom sum of the per-atom one, threads 1 and 3;
 *  3. Helfand (viscosity.py:201-233) against a long-double double loop, lag 0 exactly 0.
 * Exit code 0 = all good; any sanitizer report aborts with its own non-zero code.             */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int oracle_vacf_windowed(const double *, long, long, long, double *, double *, int);
int oracle_vacf_fft(const double *, long, long, long, double *, double *, int);
int oracle_vacf_fft_lagsum(const double *, long, long, long, double *, int);
int oracle_helfand(const double *, const double *, const double *, const double *, long, long, long, double,
                   double, double *, double *, int);

static unsigned long long s_rng = 88172645463325252ull;
static double rnd(void) {
    s_rng ^= s_rng << 13, s_rng ^= s_rng >> 7, s_rng ^= s_rng << 17;
    return (double)(s_rng >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}
static int fails = 0;
static void expect(int ok, const char *what, long a, long b, long c) {
    if (!ok) {
        ++fails;
        fprintf(stderr, "FAIL %s (T=%ld A=%ld D=%ld)\n", what, a, b, c);
    }
}
static double maxabs(const double *a, long n) {
    double m = 0;
    for (long i = 0; i < n; ++i) m = fabs(a[i]) > m ? fabs(a[i]) : m;
    return m;
}
static double maxdiff(const double *a, const double *b, long n) {
    double m = 0;
    for (long i = 0; i < n; ++i) m = fabs(a[i] - b[i]) > m ? fabs(a[i] - b[i]) : m;
    return m;
}

static void step_kat(long T, long A, long D) {
    double *v = malloc(sizeof(double) * T * A * D), *bp = malloc(sizeof(double) * T * A), *ts = malloc(sizeof(double) * T),
           *ref = malloc(sizeof(double) * T);
    for (long i = 0; i < T; ++i)
        for (long j = 0; j < A * D; ++j) v[i * A * D + j] = (double)i;
    for (long k = 0; k < T; ++k) {
        const long n = T - k;
        const long long s = (long long)(n - 1) * n * (2 * n - 1) / 6 + (long long)k * n * (n - 1) / 2;
        ref[k] = (double)D * (double)s / (double)n;
    }
    expect(oracle_vacf_windowed(v, T, A, D, bp, ts, 1) == 0 && maxdiff(ts, ref, T) <= 1e-12 * maxabs(ref, T),
           "step KAT, windowed", T, A, D);
    expect(oracle_vacf_fft(v, T, A, D, bp, ts, 2) == 0 && maxdiff(ts, ref, T) <= 1e-10 * maxabs(ref, T),
           "step KAT, fft", T, A, D);
    free(v), free(bp), free(ts), free(ref);
}

static void random_case(long T, long A, long D) {
    double *v = malloc(sizeof(double) * T * A * D), *x = malloc(sizeof(double) * T * A * D);
    double *bw = malloc(sizeof(double) * T * A), *bf = malloc(sizeof(double) * T * A), *tw = malloc(sizeof(double) * T),
           *tf = malloc(sizeof(double) * T), *ls = malloc(sizeof(double) * T), *m = malloc(sizeof(double) * A),
           *vol = malloc(sizeof(double) * T);
    for (long i = 0; i < T * A * D; ++i) v[i] = rnd(), x[i] = 10.0 * rnd();
    for (long n = 0; n < A; ++n) m[n] = 1.0 + (double)(n % 3);
    for (long i = 0; i < T; ++i) vol[i] = 8.0;
    expect(oracle_vacf_windowed(v, T, A, D, bw, tw, 3) == 0, "windowed rc", T, A, D);
    expect(oracle_vacf_fft(v, T, A, D, bf, tf, 1) == 0, "fft rc", T, A, D);
    const double sc = maxabs(bw, T * A);
    expect(maxdiff(bw, bf, T * A) <= 1e-12 * sc && maxdiff(tw, tf, T) <= 1e-12 * sc, "fft == windowed", T, A, D);
    expect(oracle_vacf_fft_lagsum(v, T, A, D, ls, 3) == 0, "lagsum rc", T, A, D);
    for (long k = 0; k < T; ++k) ls[k] /= (double)A;
    expect(maxdiff(ls, tw, T) <= 1e-12 * sc, "lag sums == atom sum", T, A, D);
    /* Helfand against a long-double loop */
    expect(oracle_helfand(v, x, m, vol, T, A, D, 300.0, 8.314462159e-3, bw, tw, 2) == 0, "helfand rc", T, A, D);
    const long double denom = 2.0L * 8.314462159e-3L * 8.0L * 300.0L;
    double worst = 0, hsc = maxabs(bw, T * A);
    for (long n = 0; n < A; ++n) {
        expect(bw[n] == 0.0, "helfand lag 0 is exactly 0", T, A, D);
        for (long lag = 1; lag < T; ++lag) {
            long double s = 0;
            for (long i = 0; i + lag < T; ++i)
                for (long d = 0; d < D; ++d) {
                    const long double p = (long double)m[n] * v[(i * A + n) * D + d] * x[(i * A + n) * D + d];
                    const long double q = (long double)m[n] * v[((i + lag) * A + n) * D + d] * x[((i + lag) * A + n) * D + d];
                    s += (p - q) * (p - q);
                }
            const double want = (double)(s / (long double)D / (long double)(T - lag) / denom);
            const double e = fabs(bw[lag * A + n] - want);
            worst = e > worst ? e : worst;
        }
    }
    expect(worst <= 1e-12 * hsc, "helfand == long-double loop", T, A, D);
    free(v), free(x), free(bw), free(bf), free(tw), free(tf), free(ls), free(m), free(vol);
}

int main(void) {
    for (long D = 1; D <= 3; ++D) step_kat(257, 2, D);
    step_kat(1, 1, 3);
    step_kat(10, 1, 1);
    const long shapes[][3] = {{1, 1, 1}, {2, 3, 2}, {7, 1, 3}, {64, 5, 3}, {100, 3, 1}, {129, 2, 2}, {255, 4, 3}};
    for (unsigned i = 0; i < sizeof(shapes) / sizeof(shapes[0]); ++i) random_case(shapes[i][0], shapes[i][1], shapes[i][2]);
    if (fails) {
        fprintf(stderr, "%d check(s) failed\n", fails);
        return 1;
    }
    puts("oracle KATs under ASan/UBSan: ok");
    return 0;
}
