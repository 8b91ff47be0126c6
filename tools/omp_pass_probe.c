// Why does the fused x/r update + reduction pass of the OpenMP baseline run at 70-100 GB/s on the 128-core host when
// the other passes run at 400-800?  Variants of that pass, timed apart.  gcc -O3 -march=native -fopenmp -ffp-contract=off
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double *skewed(size_t n, int k) { return (double *)((char *)malloc(n * 8 + 8192) + 576 * (k + 1)); }

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 10077696;
    const int reps = 8;
    double *x = skewed(n, 0), *r = skewed(n, 1), *p = skewed(n, 2), *q = skewed(n, 3), *inv = skewed(n, 4);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        x[i] = 0.0;
        r[i] = 1e-3 * (1 + i % 7);
        p[i] = 1e-3 * (1 + i % 5);
        q[i] = 1e-3 * (1 + i % 3);
        inv[i] = 1.0 / (6.0 + 1e-3 * (i % 7));
    }
    // a matrix-sized array streamed between the timed passes, as the SpMV does in the real loop (argv[3] = 1): the
    // vectors then come from DRAM, not from the L3 slices their threads left them in
    const int flush = argc > 3 ? atoi(argv[3]) : 0;
    const long nf = 110000000;
    double *big = flush ? skewed(nf, 6) : 0;
    if (big) {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < nf; ++i) big[i] = 1.0;
    }
    double sink = 0.0;
    const double t2 = 1e-7;
    int have_update = argc > 2 ? atoi(argv[2]) : 1;
    for (int variant = 0; variant < 6; ++variant) {
        double best = 1e30, rho = 0, norm = 0;
        for (int rep = 0; rep < reps; ++rep) {
            rho = norm = 0.0;
            if (big) {
                double sacc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : sacc)
                for (long i = 0; i < nf; ++i) sacc += big[i];
                sink += sacc;
            }
            const double t0 = omp_get_wtime();
            if (variant == 0) {  // the pass as the baseline has it
#pragma omp parallel for schedule(static) reduction(+ : rho, norm)
                for (long i = 0; i < n; ++i) {
                    double ri = r[i];
                    if (have_update) {
                        x[i] += t2 * p[i];
                        ri -= t2 * q[i];
                        r[i] = ri;
                    }
                    const double zi = inv ? ri * inv[i] : ri;
                    rho += ri * zi;
                    norm += fabs(ri);
                }
            } else if (variant == 1) {  // updates only
#pragma omp parallel for schedule(static)
                for (long i = 0; i < n; ++i) {
                    x[i] += t2 * p[i];
                    r[i] -= t2 * q[i];
                }
            } else if (variant == 2) {  // reductions only
#pragma omp parallel for schedule(static) reduction(+ : rho, norm)
                for (long i = 0; i < n; ++i) {
                    const double ri = r[i];
                    rho += ri * (ri * inv[i]);
                    norm += fabs(ri);
                }
            } else if (variant == 3) {  // both loops in one parallel region
#pragma omp parallel
                {
#pragma omp for schedule(static) nowait
                    for (long i = 0; i < n; ++i) {
                        x[i] += t2 * p[i];
                        r[i] -= t2 * q[i];
                    }
#pragma omp for schedule(static) reduction(+ : rho, norm)
                    for (long i = 0; i < n; ++i) {
                        const double ri = r[i];
                        rho += ri * (ri * inv[i]);
                        norm += fabs(ri);
                    }
                }
            } else if (variant == 4) {  // four independent partial sums per thread (still no reassociation by the compiler)
#pragma omp parallel reduction(+ : rho, norm)
                {
                    const int nt = omp_get_num_threads(), me = omp_get_thread_num();
                    const long b = n * me / nt, e = n * (me + 1) / nt;
                    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0;
                    long i = b;
                    for (; i + 4 <= e; i += 4) {
                        double r0 = r[i], r1 = r[i + 1], r2 = r[i + 2], r3 = r[i + 3];
                        x[i] += t2 * p[i]; x[i + 1] += t2 * p[i + 1]; x[i + 2] += t2 * p[i + 2]; x[i + 3] += t2 * p[i + 3];
                        r0 -= t2 * q[i]; r1 -= t2 * q[i + 1]; r2 -= t2 * q[i + 2]; r3 -= t2 * q[i + 3];
                        r[i] = r0; r[i + 1] = r1; r[i + 2] = r2; r[i + 3] = r3;
                        a0 += r0 * (r0 * inv[i]); a1 += r1 * (r1 * inv[i + 1]); a2 += r2 * (r2 * inv[i + 2]); a3 += r3 * (r3 * inv[i + 3]);
                        m0 += fabs(r0); m1 += fabs(r1); m2 += fabs(r2); m3 += fabs(r3);
                    }
                    for (; i < e; ++i) {
                        x[i] += t2 * p[i];
                        const double ri = r[i] - t2 * q[i];
                        r[i] = ri;
                        a0 += ri * (ri * inv[i]);
                        m0 += fabs(ri);
                    }
                    rho += (a0 + a1) + (a2 + a3);
                    norm += (m0 + m1) + (m2 + m3);
                }
            } else {  // p update (4 streams) for comparison
#pragma omp parallel for schedule(static)
                for (long i = 0; i < n; ++i) p[i] = r[i] * inv[i] + 0.5 * p[i];
            }
            const double t = omp_get_wtime() - t0;
            if (t < best) best = t;
        }
        const double bytes[6] = {56, 48, 16, 56, 56, 32};
        printf("variant %d: %.3f ms  %.1f GB/s  (rho %.6e norm %.6e)\n", variant, 1e3 * best, bytes[variant] * n / best / 1e9, rho, norm);
    }
    if (sink == 42.0) printf("%f\n", sink);
    return 0;
}
