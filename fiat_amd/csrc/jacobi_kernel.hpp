// 1-D Jacobi polynomials as a device function of their own.
//
// Reference: FIAT/jacobi.py eval_jacobi_batch (:47-74) -- all P_k^{(a,b)}(x), k = 0..n, by the three-term
// recurrence of Karniadakis & Sherwin, App. B:  P_k = (a2_k + a3_k x) P_{k-1} - a4_k P_{k-2} -- and
// eval_jacobi_deriv_batch (:85-102):  d^m/dx^m P_k^{(a,b)} = prod_{l<m} (a+b+k+1+l)/2 * P_{k-m}^{(a+m,b+m)}
// (rows k < m are zero).  Callers in the reference: LineExpansionSet (FIAT/expansions.py:659-678, (a,b) = (k,k)
// for the k-th derivative) and the edge-moment weights of dual sets (FIAT/functional.py:399).
//
// Layout: lane <-> point (x loaded once, coalesced), the two live members of the recurrence in registers, row k
// of the output written by consecutive lanes to consecutive addresses (out[k][p]).  The recurrence coefficients do
// not depend on the point: the host evaluates them (same expression order as the reference) and passes them in
// the kernel-argument segment, so the inner loop is 2 scalar-operand FMAs + 1 mul per member.  HBM-bound:
// 8 (n + 2) bytes per point.
#pragma once
#include <hip/hip_runtime.h>

namespace fxk {

constexpr int JACOBI_MAXN = 96;  // members per launch through the kernel-argument table

struct JacobiArgs {
    const double* xs;  // [npts]
    double* out;       // [n + 1][npts]
    long long npts;
    int n, order;
    double p1c, p1x;             // P_1 of the shifted weights = p1c + p1x x
    double a2[JACOBI_MAXN + 1];  // indexed by the member of the SHIFTED family, 2..n-order
    double a3[JACOBI_MAXN + 1];
    double a4[JACOBI_MAXN + 1];
    double z[JACOBI_MAXN + 1];   // derivative factor of row j (1 for order 0), indexed by row
};

__global__ __launch_bounds__(256) void jacobi_batch_kernel(const JacobiArgs A) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < A.npts; p += stride) {
        const double x = A.xs[p];
        double* col = A.out + p;
        const int lead = A.order < A.n + 1 ? A.order : A.n + 1;
        for (int k = 0; k < lead; ++k) col[(size_t)k * A.npts] = 0.0;
        if (A.n < A.order) continue;
        const int m = A.n - A.order;
        double prev = 1.0;
        col[(size_t)A.order * A.npts] = A.z[A.order];
        if (m == 0) continue;
        double cur = A.p1c + A.p1x * x;
        col[(size_t)(A.order + 1) * A.npts] = A.z[A.order + 1] * cur;
        for (int k = 2; k <= m; ++k) {
            const double next = (A.a2[k] + A.a3[k] * x) * cur - A.a4[k] * prev;
            prev = cur;
            cur = next;
            col[(size_t)(A.order + k) * A.npts] = A.z[A.order + k] * cur;
        }
    }
}

}  // namespace fxk
