/* TEST INFRASTRUCTURE -- plain C restatement of FIAT's tabulate() hot path for
 * simplices, used as the parity checker at full batch sizes and as the
 * `cpu_baseline` ("port") of bench.py.  Never linked into the product.
 *
 * Follows the reference step by step (paths relative to /root/reference):
 *   jacobi coefficients          FIAT/expansions.py:24-40   (jrc, integrated_jrc)
 *   collapsed-coordinate factors FIAT/expansions.py:43-63
 *   Leibniz rule                 FIAT/expansions.py:66-137  (orders <= 2)
 *   dubiner_recurrence           FIAT/expansions.py:140-267 (in-place per-codimension normalisation)
 *   C0_basis                     FIAT/expansions.py:270-322
 *   affine map of the cell       FIAT/expansions.py:416-418
 *   coeffs . expansion values    FIAT/polynomial_set.py:68-72
 * Pinned against the NumPy oracle and the golden vectors in tests/test_oracle_c.py.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC -o liboracle.so fiat_oracle.c -lm   (oracle/Makefile)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int member_index(int sd, int p, int q, int r) {
    if (sd == 1) return p;
    if (sd == 2) return (p + q) * (p + q + 1) / 2 + q;
    int t = p + q + r, u = q + r;
    return t * (t + 1) * (t + 2) / 6 + u * (u + 1) / 2 + r;
}

static int binom(int n, int k) {
    long long r = 1;
    if (k < 0 || k > n) return 0;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return (int)r;
}

static void jrc(double a, double b, int n, double* an, double* bn, double* cn) {
    double s = a + b;
    *an = (2 * n + 1 + s) * (2 * n + 2 + s) / (2 * (n + 1) * (n + 1 + s));
    *bn = s * (a - b) * (2 * n + 1 + s) / (2 * (n + 1) * (n + 1 + s) * (2 * n + s));
    *cn = (n + a) * (n + b) * (2 * n + 2 + s) / ((n + 1) * (n + 1 + s) * (2 * n + s));
}

static void integrated_jrc(double a, double b, int n, double* an, double* bn, double* cn) {
    if (n == 1) {
        *an = (a + b + 2) / 2;
        *bn = (a - 3 * b - 2) / 2;
        *cn = 0.0;
    } else {
        jrc(a - 1, b + 1, n - 1, an, bn, cn);
    }
}

#define NC_MAX 10 /* 1 + 3 + 6 components for sd = 3, order = 2 */

static int ncomp(int sd, int order) { return binom(sd + order, sd); }

/* D^alpha(F G) for all components up to `order`; F has gradient dF and (constant) Hessian ddF.
 * component layout: [value | gradient d | hessian (d1<=d2) in mis order] */
static void leibniz(int sd, int order, double F, const double* dF, const double* ddF, const double* G, double* out) {
    out[0] += F * G[0];
    if (order >= 1)
        for (int d = 0; d < sd; ++d) out[1 + d] += F * G[1 + d] + (dF ? dF[d] * G[0] : 0.0);
    if (order >= 2) {
        int h = 0;
        for (int d1 = 0; d1 < sd; ++d1)
            for (int d2 = d1; d2 < sd; ++d2, ++h) {
                double t = F * G[1 + sd + h];
                if (dF) t += dF[d1] * G[1 + d2] + dF[d2] * G[1 + d1];
                if (ddF) t += ddF[h] * G[0];
                out[1 + sd + h] += t;
            }
    }
}

/* members [nexp][nc] at one point. X: default-simplex coordinates, J: sd x sd rows = grad X_i */
static void dubiner_point(int sd, int n, int order, int variant, double scale, const double* X, const double* J,
                          double* phi /* nexp*nc */) {
    const int nc = ncomp(sd, order);
    const int nexp = binom(n + sd, sd);
    memset(phi, 0, sizeof(double) * nexp * nc);
    if (variant == 1) scale = -scale;
    phi[0] = scale;
    if (n == 0) return;
    const int beta = variant == 2 ? 1 : 0;
    double Xp[5], Jp[5][3];
    for (int i = 0; i < 5; ++i) {
        Xp[i] = i < sd ? X[i] : -1.0;
        for (int d = 0; d < 3; ++d) Jp[i][d] = (i < sd && d < sd) ? J[i * sd + d] : 0.0;
    }
    for (int codim = 0; codim < sd; ++codim) {
        double x = Xp[codim], y = Xp[codim + 1], z = Xp[codim + 2];
        double fb = 0.5 * (y + z), fa = x + (fb + 1.0), fc = fb * fb;
        double dfa[3], dfb[3], dfc[3], ddfc[6];
        for (int d = 0; d < sd; ++d) {
            dfb[d] = 0.5 * (Jp[codim + 1][d] + Jp[codim + 2][d]);
            dfa[d] = Jp[codim][d] + dfb[d];
            dfc[d] = 2 * fb * dfb[d];
        }
        {
            int h = 0;
            for (int d1 = 0; d1 < sd; ++d1)
                for (int d2 = d1; d2 < sd; ++d2) ddfc[h++] = 2 * dfb[d1] * dfb[d2];
        }
        /* prefixes (length codim, sum < n), last entry slowest (reference_element.py:64-76) */
        int nsub = codim == 0 ? 1 : (codim == 1 ? n : n * (n + 1) / 2);
        for (int si = 0; si < nsub; ++si) {
            int sub[2] = {0, 0};
            if (codim == 1) sub[0] = si;
            if (codim == 2) {
                int k = si, last = 0;
                while (k >= n - last) { k -= n - last; ++last; }
                sub[0] = k;
                sub[1] = last;
            }
            int s = sub[0] + sub[1];
            double alpha, a, b, c;
            if (variant == 1) {
                alpha = 2 * s;
                a = b = -0.5;
            } else {
                alpha = 2 * s + codim;
                if (variant == 2) alpha += 1 + codim;
                a = 0.5 * (alpha + beta) + 1.0;
                b = 0.5 * (alpha - beta);
            }
            int idx[3] = {0, 0, 0};
            for (int j = 0; j < codim; ++j) idx[j] = sub[j];
            idx[codim] = 0;
            int icur = member_index(sd, idx[0], idx[1], idx[2]);
            idx[codim] = 1;
            int inext = member_index(sd, idx[0], idx[1], idx[2]);
            double f = a * fa - b * fb, df[3], g, dg[3], ddg[6];
            for (int d = 0; d < sd; ++d) df[d] = a * dfa[d] - b * dfb[d];
            int ord = order < s + 1 ? order : s + 1;
            double tmp[NC_MAX];
            memset(tmp, 0, sizeof tmp);
            leibniz(sd, ord, f, df, NULL, phi + icur * nc, tmp);
            memcpy(phi + inext * nc, tmp, sizeof(double) * nc);
            int iprev;
            for (int i = 1; i < n - s; ++i) {
                iprev = icur;
                icur = inext;
                idx[codim] = i + 1;
                inext = member_index(sd, idx[0], idx[1], idx[2]);
                if (variant == 1)
                    integrated_jrc(alpha, beta, i, &a, &b, &c);
                else
                    jrc(alpha, beta, i, &a, &b, &c);
                f = a * fa - b * fb;
                g = -c * fc;
                for (int d = 0; d < sd; ++d) {
                    df[d] = a * dfa[d] - b * dfb[d];
                    dg[d] = -c * dfc[d];
                }
                for (int h = 0; h < sd * (sd + 1) / 2; ++h) ddg[h] = -c * ddfc[h];
                ord = order < s + 1 + i ? order : s + 1 + i;
                memset(tmp, 0, sizeof tmp);
                leibniz(sd, ord, f, df, NULL, phi + icur * nc, tmp);
                leibniz(sd, ord, g, dg, ddg, phi + iprev * nc, tmp);
                memcpy(phi + inext * nc, tmp, sizeof(double) * nc);
            }
        }
        /* normalise every member whose index has length d = codim + 1 (expansions.py:251-266) */
        int d = codim + 1, shift = variant == 2 ? 1 : 0;
        for (int i2 = 0; i2 <= (d >= 3 ? n : 0); ++i2)
            for (int i1 = 0; i1 <= (d >= 2 ? n - i2 : 0); ++i1)
                for (int i0 = 0; i0 <= n - i1 - i2; ++i0) {
                    int id[3] = {i0, i1, i2};
                    int sum = i0 + i1 + i2, lastv = id[d - 1];
                    double norm2;
                    if (variant != 0) {
                        int p = lastv + shift;
                        int al = 2 * ((sum - lastv) + d * shift) - 1;
                        norm2 = (0.5 + d) / d;
                        if (p > 0 && p + al > 0) norm2 *= (double)(p + al) * (2 * p + al) / p;
                    } else {
                        norm2 = (2.0 * sum + d) / d;
                    }
                    double w = sqrt(norm2);
                    int m = member_index(sd, id[0], id[1], id[2]);
                    for (int t = 0; t < nc; ++t) phi[m * nc + t] *= w;
                }
    }
}

/* member order (vertices, edges, faces, interior) of C0_basis (expansions.py:297-322) */
static int* c0_permutation(int sd, int n) {
    const int nexp = binom(n + sd, sd);
    int* dofs = (int*)malloc(sizeof(int) * nexp);
    int k = 0;
    for (int i = 0; i <= sd; ++i) dofs[k++] = i;
    if (sd == 1) {
        for (int i = 2; i <= n; ++i) dofs[k++] = i;
    } else if (sd == 2) {
        for (int i = 2; i <= n; ++i) dofs[k++] = member_index(2, 1, i - 1, 0);
        for (int i = 2; i <= n; ++i) dofs[k++] = member_index(2, 0, i, 0);
        for (int i = 2; i <= n; ++i) dofs[k++] = member_index(2, i, 0, 0);
        for (int j = 1; j <= n; ++j)
            for (int i = 2; i <= n - j; ++i) dofs[k++] = member_index(2, i, j, 0);
    } else {
        for (int i = 2; i <= n; ++i) dofs[k++] = member_index(3, 0, 1, i - 1);
        for (int i = 2; i <= n; ++i) dofs[k++] = member_index(3, 1, 0, i - 1);
        for (int i = 2; i <= n; ++i) dofs[k++] = member_index(3, 1, i - 1, 0);
        for (int i = 2; i <= n; ++i) dofs[k++] = member_index(3, 0, 0, i);
        for (int i = 2; i <= n; ++i) dofs[k++] = member_index(3, 0, i, 0);
        for (int i = 2; i <= n; ++i) dofs[k++] = member_index(3, i, 0, 0);
        for (int j = 1; j <= n; ++j)
            for (int i = 2; i <= n - j; ++i) dofs[k++] = member_index(3, 1, i - 1, j);
        for (int j = 1; j <= n; ++j)
            for (int i = 2; i <= n - j; ++i) dofs[k++] = member_index(3, 0, i, j);
        for (int j = 1; j <= n; ++j)
            for (int i = 2; i <= n - j; ++i) dofs[k++] = member_index(3, i, 0, j);
        for (int j = 1; j <= n; ++j)
            for (int i = 2; i <= n - j; ++i) dofs[k++] = member_index(3, i, j, 0);
        for (int kk = 1; kk <= n; ++kk)
            for (int j = 1; j <= n - kk; ++j)
                for (int i = 2; i <= n - j - kk; ++i) dofs[k++] = member_index(3, i, j, kk);
    }
    return dofs;
}

/* in-place C0_basis on members [nexp][nc]; `dofs` = c0_permutation, `work` nexp*nc doubles */
static void c0_basis(int sd, int n, int nc, double* phi, double* work, const int* dofs) {
    const int nexp = binom(n + sd, sd);
#define ROW(m) (phi + (m) * nc)
#define SUB(dst, src) for (int t = 0; t < nc; ++t) ROW(dst)[t] -= ROW(src)[t]
    for (int t = 0; t < nc; ++t) ROW(0)[t] *= -1.0;
    for (int j = 1; j <= sd; ++j) SUB(0, j);
    if (sd == 2) {
        for (int i = 2; i <= n; ++i) SUB(member_index(2, 0, i, 0), member_index(2, 1, i - 1, 0));
    } else if (sd == 3) {
        for (int i = 2; i <= n; ++i) {
            for (int j = 0; j <= n - i; ++j) SUB(member_index(3, 0, i, j), member_index(3, 1, i - 1, j));
            SUB(member_index(3, 0, 0, i), member_index(3, 0, 1, i - 1));
            SUB(member_index(3, 0, 0, i), member_index(3, 1, 0, i - 1));
        }
    }
    memcpy(work, phi, sizeof(double) * nexp * nc);
    for (int r = 0; r < nexp; ++r) memcpy(phi + r * nc, work + dofs[r] * nc, sizeof(double) * nc);
#undef ROW
#undef SUB
}

/* closed-form affine map of a simplex onto the default (-1,1)^sd simplex: X = A x + b */
static int cell_map(int sd, const double* v, double* A, double* b) {
    if (sd == 1) {
        A[0] = 2.0 / (v[1] - v[0]);
        b[0] = -1.0 - A[0] * v[0];
        return 0;
    }
    double E[9], inv[9];
    for (int c = 0; c < sd; ++c)
        for (int r = 0; r < sd; ++r) E[r * sd + c] = v[sd * (c + 1) + r] - v[r];
    if (sd == 2) {
        double det = E[0] * E[3] - E[1] * E[2];
        inv[0] = E[3] / det; inv[1] = -E[1] / det; inv[2] = -E[2] / det; inv[3] = E[0] / det;
    } else {
        double c00 = E[4] * E[8] - E[5] * E[7], c01 = E[5] * E[6] - E[3] * E[8], c02 = E[3] * E[7] - E[4] * E[6];
        double det = E[0] * c00 + E[1] * c01 + E[2] * c02;
        inv[0] = c00 / det; inv[1] = (E[2] * E[7] - E[1] * E[8]) / det; inv[2] = (E[1] * E[5] - E[2] * E[4]) / det;
        inv[3] = c01 / det; inv[4] = (E[0] * E[8] - E[2] * E[6]) / det; inv[5] = (E[2] * E[3] - E[0] * E[5]) / det;
        inv[6] = c02 / det; inv[7] = (E[1] * E[6] - E[0] * E[7]) / det; inv[8] = (E[0] * E[4] - E[1] * E[3]) / det;
    }
    for (int i = 0; i < sd; ++i) {
        double t = 0.0;
        for (int d = 0; d < sd; ++d) {
            A[i * sd + d] = 2.0 * inv[i * sd + d];
            t += A[i * sd + d] * v[d];
        }
        b[i] = -1.0 - t;
    }
    return 0;
}

/* Batched CiarletElement.tabulate:
 *   cell   [(sd+1)*sd] vertices of the element's own cell (used when verts == NULL)
 *   coeffs [rows][nexp] (rows = ndof * vdim), pts [nreq][npts][sd], verts [nreq][sd+1][sd] or NULL
 *   out    [nreq][ntab][rows][npts]
 * returns 0, or -1 on bad arguments.  OpenMP over requests. */
int fo_tabulate_batch(int sd, int n, int variant, double scale, const double* cell, const double* coeffs, int rows,
                      int order, long long nreq, int npts, const double* pts, const double* verts, double* out,
                      int nthreads) {
    if (sd < 1 || sd > 3 || order < 0 || order > 2 || n < 0) return -1;
    const int nc = ncomp(sd, order), nexp = binom(n + sd, sd);
    double A0[9], b0[3];
    cell_map(sd, cell, A0, b0);
    int* dofs = (variant == 1 && n >= 1) ? c0_permutation(sd, n) : NULL;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        double* phi = (double*)malloc(sizeof(double) * nexp * nc * 2);
        double* work = phi + nexp * nc;
#pragma omp for schedule(static)
        for (long long r = 0; r < nreq; ++r) {
            double A[9], b[3];
            if (verts) {
                cell_map(sd, verts + r * (sd + 1) * sd, A, b);
            } else {
                memcpy(A, A0, sizeof A);
                memcpy(b, b0, sizeof b);
            }
            double* o = out + (size_t)r * nc * rows * npts;
            for (int p = 0; p < npts; ++p) {
                const double* x = pts + ((size_t)r * npts + p) * sd;
                double X[3];
                for (int i = 0; i < sd; ++i) {
                    double t = b[i];
                    for (int d = 0; d < sd; ++d) t += A[i * sd + d] * x[d];
                    X[i] = t;
                }
                dubiner_point(sd, n, order, variant, scale, X, A, phi);
                if (variant == 1) c0_basis(sd, n, nc, phi, work, dofs);
                for (int t = 0; t < nc; ++t)
                    for (int i = 0; i < rows; ++i) {
                        double s = 0.0;
                        const double* ci = coeffs + (size_t)i * nexp;
                        for (int k = 0; k < nexp; ++k) s += ci[k] * phi[k * nc + t];
                        o[((size_t)t * rows + i) * npts + p] = s;
                    }
            }
        }
        free(phi);
    }
    free(dofs);
    return 0;
}

int fo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
