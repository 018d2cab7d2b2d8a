// Point production on the device: Gauss-Jacobi rules and the collapsed
// (Duffy) simplex rules built from them.
//
// Reference: FIAT/quadrature.py GaussJacobiQuadratureLineRule (:96-110) and
// CollapsedQuadratureSimplexRule (:171-181), which call recursivenodes' gaussjacobi /
// simplexgausslegendre; the Jacobi polynomials follow FIAT/jacobi.py:47-74 (eval_jacobi_batch)
// and :77-102 (derivative = (a+b+n+1)/2 P_{n-1}^{(a+1,b+1)}).  The roots are found as in
// Karniadakis & Sherwin (Newton with deflation from Chebyshev guesses), the weights by the
// classical formula  w_i = 2^{a+b+1} G(m+a+1) G(m+b+1) / (G(m+a+b+1) m!) / ((1-x_i^2) P'_m(x_i)^2).
// Construction-time work: one thread per rule, latency-bound and tiny.
#pragma once
#include <hip/hip_runtime.h>

namespace fxk {

constexpr int GJ_MAX = 64;  // points per direction

__device__ inline double jacobi_p(int n, double a, double b, double x) {
    if (n == 0) return 1.0;
    double p0 = 1.0, p1 = 0.5 * (a - b + (a + b + 2.0) * x);
    for (int k = 1; k < n; ++k) {
        const double a1 = 2.0 * (k + 1.0) * (k + a + b + 1.0) * (2.0 * k + a + b);
        const double a2 = (2.0 * k + a + b + 1.0) * (a * a - b * b);
        const double a3 = (2.0 * k + a + b) * (2.0 * k + a + b + 1.0) * (2.0 * k + a + b + 2.0);
        const double a4 = 2.0 * (k + a) * (k + b) * (2.0 * k + a + b + 2.0);
        const double p2 = ((a2 + a3 * x) * p1 - a4 * p0) / a1;
        p0 = p1;
        p1 = p2;
    }
    return p1;
}

__device__ inline double jacobi_dp(int n, double a, double b, double x) {
    return n == 0 ? 0.0 : 0.5 * (a + b + n + 1.0) * jacobi_p(n - 1, a + 1.0, b + 1.0, x);
}

// rules[j] = m-point Gauss-Jacobi(a = j, b = 0) rule, j = 0..nrules-1: x at rules[(2j)*m], w at rules[(2j+1)*m]
__global__ void gauss_jacobi_kernel(int m, int nrules, double* rules) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nrules) return;
    const double a = (double)j, b = 0.0;
    double* x = rules + (size_t)(2 * j) * m;
    double* w = x + m;
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < m; ++k) {
        double r = -cos((2.0 * k + 1.0) * pi / (2.0 * m));
        if (k > 0) r = 0.5 * (r + x[k - 1]);
        for (int it = 0; it < 100; ++it) {
            double s = 0.0;
            for (int i = 0; i < k; ++i) s += 1.0 / (r - x[i]);
            const double p = jacobi_p(m, a, b, r), dp = jacobi_dp(m, a, b, r);
            const double delta = -p / (dp - s * p);
            r += delta;
            if (fabs(delta) <= 4.0e-16 * fmax(1.0, fabs(r))) break;
        }
        x[k] = r;
    }
    const double c = exp2(a + b + 1.0) * exp(lgamma(m + a + 1.0) + lgamma(m + b + 1.0) - lgamma(m + a + b + 1.0) - lgamma(m + 1.0));
    for (int k = 0; k < m; ++k) {
        const double dp = jacobi_dp(m, a, b, x[k]);
        w[k] = c / ((1.0 - x[k] * x[k]) * dp * dp);
    }
}

struct RuleArgs {
    const double* rules;  // from gauss_jacobi_kernel, SD rules
    double* pts;          // [m^SD][SD] on the target cell
    double* wts;          // [m^SD]
    double Ainv[9];       // default simplex -> cell: x = Ainv (X - b)
    double b[3];
    double jac;           // |det Ainv|
    int m;
};

// point q = (i_0, ..., i_{SD-1}), the last index fastest (itertools.product order of the host facade):
// collapsed coordinates e_j = x^{(j)}_{i_j}, X_i = (1 + e_i) prod_{j>i} (1 - e_j)/2 - 1, w = prod_j w^{(j)}_{i_j} / 2^j
template <int SD> __global__ void collapsed_rule_kernel(const RuleArgs a) {
    int total = 1;
    for (int d = 0; d < SD; ++d) total *= a.m;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    int idx[SD], rem = q;
    for (int d = SD - 1; d >= 0; --d) {
        idx[d] = rem % a.m;
        rem /= a.m;
    }
    double e[SD], w = 1.0;
    for (int j = 0; j < SD; ++j) {
        e[j] = a.rules[(size_t)(2 * j) * a.m + idx[j]];
        w *= a.rules[(size_t)(2 * j + 1) * a.m + idx[j]] / (double)(1 << j);
    }
    double X[SD];
    for (int i = 0; i < SD; ++i) {
        double f = 1.0 + e[i];
        for (int j = i + 1; j < SD; ++j) f *= 0.5 * (1.0 - e[j]);
        X[i] = f - 1.0;
    }
    for (int i = 0; i < SD; ++i) {
        double t = 0.0;
        for (int d = 0; d < SD; ++d) t += a.Ainv[i * SD + d] * (X[d] - a.b[d]);
        a.pts[(size_t)q * SD + i] = t;
    }
    a.wts[q] = w * a.jac;
}

}  // namespace fxk
