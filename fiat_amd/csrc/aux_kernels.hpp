// Auxiliary gfx950 kernels of the tabulate path: Riesz/Vandermonde assembly and
// solve (construction side), 1-D barycentric Lagrange tabulation and
// tensor-product expansion (hex/quad side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "store.hpp"

namespace fxk {

// floor(c / d) for 0 <= c < 2^20, rinv = 1.0f / d
__device__ __forceinline__ int idiv_small(int c, float rinv);

// mat[i][k] = sum_q wts[i][q] * ev[k][q]      (dual_set.py:172)
__global__ void riesz_assemble_kernel(int nrows, int nq, int nexp, const double* __restrict__ wts,
                                      const double* __restrict__ ev, double* __restrict__ mat) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nrows * nexp) return;
    int i = (int)(idx / nexp), k = (int)(idx - (long long)i * nexp);
    const double* w = wts + (size_t)i * nq;
    const double* e = ev + (size_t)k * nq;
    double s = 0.0;
    for (int q = 0; q < nq; ++q) s += w[q] * e[q];
    mat[idx] = s;
}

// One workgroup per system:  V = A B^T ;  X = solve(V^T, B)  by LU with partial
// pivoting (finite_element.py:141-159; scipy.linalg.solve(V, B, transposed=True)
// is LAPACK gesv, the same elimination order).  LDS: M = V^T (n x n) followed
// by the right-hand sides / solution (n x m).
// GLOBAL: systems beyond the LDS (n^2 + n m doubles > 150 kB, e.g. 105 dofs of BDM4 or 120 of P7 on a
// tetrahedron) keep M and R in a global workspace `ws` (L2-resident; construction only).
template <bool GLOBAL>
__global__ __launch_bounds__(256) void vandermonde_solve_kernel(int n, int m, const double* __restrict__ Aall,
                                                               const double* __restrict__ Ball,
                                                               double* __restrict__ Xall,
                                                               double* __restrict__ Vall,
                                                               int* __restrict__ info_all, double* ws) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* M = GLOBAL ? ws + (size_t)blockIdx.x * ((size_t)n * n + (size_t)n * m) : sm;  // n x n, M[r][c] = V[c][r]
    double* R = M + (size_t)n * n;     // n x m
    __shared__ int s_piv;
    __shared__ int s_info;
    __shared__ double s_anorm;
    __shared__ double s_red[256];
    __shared__ int s_idx[256];
    const int tid = threadIdx.x, nt = blockDim.x;
    const size_t sys = blockIdx.x;
    const double* A = Aall + sys * (size_t)n * m;
    const double* B = Ball + sys * (size_t)n * m;
    if (tid == 0) s_info = 0;
    for (int e = tid; e < n * n; e += nt) {
        int i = e / n, j = e - i * n;  // V[i][j] = sum_k A[i][k] B[j][k]
        double s = 0.0;
        for (int k = 0; k < m; ++k) s += A[(size_t)i * m + k] * B[(size_t)j * m + k];
        M[(size_t)j * n + i] = s;
        if (Vall) Vall[sys * (size_t)n * n + e] = s;
    }
    for (int e = tid; e < n * m; e += nt) R[e] = B[e];
    __syncthreads();
    {   // max |V_ij|: scale for the singularity test
        double mx = 0.0;
        for (int e = tid; e < n * n; e += nt) mx = fmax(mx, fabs(M[e]));
        s_red[tid] = mx;
        __syncthreads();
        for (int s = nt >> 1; s > 0; s >>= 1) {
            if (tid < s) s_red[tid] = fmax(s_red[tid], s_red[tid + s]);
            __syncthreads();
        }
        if (tid == 0) s_anorm = s_red[0];
        __syncthreads();
    }
    // a pivot below n*eps*max|V| means V is singular to working precision: the
    // reference turns LAPACK's rcond < eps warning into LinAlgError
    // (finite_element.py:151-156)
    const double tiny = (double)n * 2.220446049250313e-16 * s_anorm;
    for (int k = 0; k < n; ++k) {
        // pivot search in column k, rows >= k
        double best = -1.0;
        int bi = k;
        for (int r = k + tid; r < n; r += nt) {
            double v = fabs(M[(size_t)r * n + k]);
            if (v > best) { best = v; bi = r; }
        }
        s_red[tid] = best;
        s_idx[tid] = bi;
        __syncthreads();
        for (int s = nt >> 1; s > 0; s >>= 1) {
            if (tid < s) {
                double o = s_red[tid + s];
                int oi = s_idx[tid + s];
                if (o > s_red[tid] || (o == s_red[tid] && oi < s_idx[tid])) { s_red[tid] = o; s_idx[tid] = oi; }
            }
            __syncthreads();
        }
        if (tid == 0) {
            s_piv = s_idx[0];
            if (!(s_red[0] > tiny) && s_info == 0) s_info = k + 1;
        }
        __syncthreads();
        const int piv = s_piv;
        if (piv != k) {
            for (int c = tid; c < n; c += nt) {
                double t = M[(size_t)k * n + c];
                M[(size_t)k * n + c] = M[(size_t)piv * n + c];
                M[(size_t)piv * n + c] = t;
            }
            for (int c = tid; c < m; c += nt) {
                double t = R[(size_t)k * m + c];
                R[(size_t)k * m + c] = R[(size_t)piv * m + c];
                R[(size_t)piv * m + c] = t;
            }
        }
        __syncthreads();
        const double pkk = M[(size_t)k * n + k];
        const double pinv = (pkk != 0.0) ? 1.0 / pkk : 0.0;
        // multipliers (stored in place), then rank-1 update of the trailing block and of R
        for (int r = k + 1 + tid; r < n; r += nt) M[(size_t)r * n + k] *= pinv;
        __syncthreads();
        const int nr = n - k - 1, ncM = n - k - 1;
        for (int e = tid; e < nr * (ncM + m); e += nt) {
            int r = k + 1 + e / (ncM + m);
            int c = e % (ncM + m);
            double l = M[(size_t)r * n + k];
            if (c < ncM)
                M[(size_t)r * n + k + 1 + c] -= l * M[(size_t)k * n + k + 1 + c];
            else
                R[(size_t)r * m + (c - ncM)] -= l * R[(size_t)k * m + (c - ncM)];
        }
        __syncthreads();
    }
    // back substitution, threads <-> right-hand-side columns
    for (int c = tid; c < m; c += nt) {
        for (int r = n - 1; r >= 0; --r) {
            double s = R[(size_t)r * m + c];
            for (int j = r + 1; j < n; ++j) s -= M[(size_t)r * n + j] * R[(size_t)j * m + c];
            R[(size_t)r * m + c] = s / M[(size_t)r * n + r];
        }
    }
    __syncthreads();
    for (int e = tid; e < n * m; e += nt) Xall[sys * (size_t)n * m + e] = R[e];
    if (tid == 0) info_all[sys] = s_info;
}

// ---------------------------------------------------------------------------
// 1-D Lagrange basis by the second barycentric formula
// (barycentric_interpolation.py:22-47).  tab layout [k][i] in registers of the
// calling lane for one point; NN_MAX bounds the node count.
constexpr int NN_MAX = 16;

struct LineDesc {
    const double* nodes;  // [nn]
    const double* wts;    // [nn] barycentric weights
    const double* dmat;   // [nn][nn] differentiation matrix
    int nn;
};

// values phi[i] at x; exact Kronecker delta when x hits a node
// (barycentric_interpolation.py:35-40: NaN -> 1 after the normalisation).
// (node data through the constant address space: never written while a kernel runs, so the loads are scalar)
typedef const __attribute__((address_space(4))) double LineConst;

__device__ __forceinline__ void lagrange_values(const LineDesc& L, double x, double* phi) {
    LineConst* nodes = (LineConst*)(unsigned long long)L.nodes;
    LineConst* wts = (LineConst*)(unsigned long long)L.wts;
    double sum = 0.0;
    int hit = -1;
#pragma unroll
    for (int i = 0; i < NN_MAX; ++i) {
        if (i < L.nn) {
            double d = x - nodes[i];
            if (d == 0.0) hit = i;
            double t = wts[i] / d;
            phi[i] = t;
            sum += t;
        }
    }
    double inv = 1.0 / sum;
#pragma unroll
    for (int i = 0; i < NN_MAX; ++i) {
        if (i < L.nn) phi[i] = (hit >= 0) ? ((i == hit) ? 1.0 : 0.0) : phi[i] * inv;
    }
}

// out = dmat . in
__device__ __forceinline__ void lagrange_diff(const LineDesc& L, const double* in, double* out) {
    LineConst* dmat = (LineConst*)(unsigned long long)L.dmat;
#pragma unroll
    for (int i = 0; i < NN_MAX; ++i) {
        if (i < L.nn) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < NN_MAX; ++j)
                if (j < L.nn) s += dmat[i * L.nn + j] * in[j];
            out[i] = s;
        }
    }
}

// the same with a compile-time node count: every index is a constant, the node data are uniform loads
template <int NN> __device__ __forceinline__ void lagrange_values_n(const LineDesc& L, double x, double (&phi)[NN]) {
    LineConst* nodes = (LineConst*)(unsigned long long)L.nodes;
    LineConst* wts = (LineConst*)(unsigned long long)L.wts;
    double sum = 0.0;
    int hit = -1;
#pragma unroll
    for (int i = 0; i < NN; ++i) {
        const double d = x - nodes[i];
        if (d == 0.0) hit = i;
        const double t = wts[i] / d;
        phi[i] = t;
        sum += t;
    }
    const double inv = 1.0 / sum;
#pragma unroll
    for (int i = 0; i < NN; ++i) phi[i] = hit >= 0 ? (i == hit ? 1.0 : 0.0) : phi[i] * inv;
}

template <int NN> __device__ __forceinline__ void lagrange_diff_n(const LineDesc& L, const double (&in)[NN], double (&out)[NN]) {
    LineConst* dmat = (LineConst*)(unsigned long long)L.dmat;
#pragma unroll
    for (int i = 0; i < NN; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NN; ++j) s += dmat[i * NN + j] * in[j];
        out[i] = s;
    }
}

// out[r][k][i][p], one thread per (r, p)
__global__ void line_tabulate_kernel(LineDesc L, int order, long long nreq, int npts,
                                     const double* __restrict__ pts, double* __restrict__ out) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nreq * npts) return;
    long long r = idx / npts;
    int p = (int)(idx - r * npts);
    double a[NN_MAX], b[NN_MAX];
    lagrange_values(L, pts[idx], a);
    double* o = out + (size_t)r * (order + 1) * L.nn * npts + p;
    for (int k = 0; k <= order; ++k) {
        if (k > 0) {
            lagrange_diff(L, a, b);
#pragma unroll
            for (int i = 0; i < NN_MAX; ++i) a[i] = b[i];
        }
#pragma unroll
        for (int i = 0; i < NN_MAX; ++i)
            if (i < L.nn) o[((size_t)k * L.nn + i) * npts] = a[i];
    }
}

// More than NN_MAX nodes (1-D spectral elements of degree 16 ... 255: construction and interpolation studies, not a
// throughput path): one thread per (r, p) again, but nothing is kept in registers -- the values go straight to `out`
// (two passes over the nodes: sum and node hit, then the normalised terms), derivative order k is dmat times the thread's
// own order k - 1 column of `out`, read back from memory.
constexpr int NN_BIG_MAX = 256;
__global__ void line_tabulate_big_kernel(LineDesc L, int order, long long nreq, int npts,
                                         const double* __restrict__ pts, double* out) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nreq * npts) return;
    long long r = idx / npts;
    int p = (int)(idx - r * npts);
    const double x = pts[idx];
    const int nn = L.nn;
    double sum = 0.0;
    int hit = -1;
    for (int i = 0; i < nn; ++i) {
        const double d = x - L.nodes[i];
        if (d == 0.0) hit = i;
        sum += L.wts[i] / d;
    }
    double* o = out + (size_t)r * (order + 1) * nn * npts + p;
    const double inv = 1.0 / sum;
    for (int i = 0; i < nn; ++i) o[(size_t)i * npts] = hit >= 0 ? (i == hit ? 1.0 : 0.0) : (L.wts[i] / (x - L.nodes[i])) * inv;
    for (int k = 1; k <= order; ++k) {
        const double* prev = o + (size_t)(k - 1) * nn * npts;
        double* cur = o + (size_t)k * nn * npts;
        for (int i = 0; i < nn; ++i) {
            double s = 0.0;
            for (int j = 0; j < nn; ++j) s += L.dmat[(size_t)i * nn + j] * prev[(size_t)j * npts];
            cur[(size_t)i * npts] = s;
        }
    }
}

// Tensor-product expansion (tensor_product.py:231-292, scalar factors, nested
// left to right).  One workgroup per request:
//   phase 1: factor tables T[f][k][i][p] -> LDS (threads <-> (f, p))
//   phase 2: out[t][(i0,i1,i2)][p] = prod_f T[f][alpha_t[f]][i_f][p_f], streamed
//            row by row with lanes <-> points.
// GRID == true: points are the tensor grid of per-request 1-D coordinates
// (grid[r][f][q], point index = (j0*q + j1)*q + j2), the factor tables are
// only q wide (sum-factorised form); otherwise pts[r][p][nf].
struct TensorArgs {
    LineDesc L[3];
    int nf, order, ntab;
    int alpha[10][3];  // derivative multi-indices in mis() order
    long long nreq;
    int npts;  // output points per request
    int q;     // grid mode: points per direction
    const double* pts;
    double* out;
};

constexpr int TP_NIT = 2;  // 16-byte chunks per lane and row group in the fast path (row groups <= 256 doubles)

// NNC > 0: every factor has NNC nodes (compile-time loops in the factor phase); 0: any node counts
template <bool GRID, int NNC = 0>
__global__ __launch_bounds__(256) void tensor_tabulate_kernel(TensorArgs a) {
    extern __shared__ __attribute__((aligned(16))) double T[];  // [f][k][i][w]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int w = GRID ? a.q : a.npts;  // width of the factor tables
    const int K = a.order + 1;
    int fofs[4];
    fofs[0] = 0;
    for (int f = 0; f < a.nf; ++f) fofs[f + 1] = fofs[f] + K * a.L[f].nn * w;
    const int nn1 = a.nf > 1 ? a.L[1].nn : 1, nn2 = a.nf > 2 ? a.L[2].nn : 1;
    const float rinv_nbf = 1.0f / (float)(a.L[0].nn * nn1 * nn2);
    const float rinv_n12 = 1.0f / (float)(nn1 * nn2);
    const float rinv_n2 = 1.0f / (float)nn2;
    const float rinv_q = 1.0f / (float)(a.q > 0 ? a.q : 1);
    const float rinv_qq = 1.0f / (float)(a.q > 0 ? a.q * a.q : 1);
    // LDS row table: for every output row (table t, basis function (i0,i1,i2)) the LDS offsets of
    // its three factor rows -- identical for every request, decoded once per kernel
    int* rowbase = reinterpret_cast<int*>(T + fofs[a.nf]);
    {
        const int n0_ = a.L[0].nn;
        const int nbf_ = n0_ * nn1 * nn2;
        for (int row = tid; row < a.ntab * nbf_; row += nthr) {
            const int t = row / nbf_;
            const int bf = row - t * nbf_;
            const int i0 = bf / (nn1 * nn2);
            const int rem = bf - i0 * (nn1 * nn2);
            const int i1 = rem / nn2;
            const int i2 = rem - i1 * nn2;
            rowbase[row * 4 + 0] = fofs[0] + (a.alpha[t][0] * n0_ + i0) * w;
            rowbase[row * 4 + 1] = a.nf > 1 ? fofs[1] + (a.alpha[t][1] * nn1 + i1) * w : 0;
            rowbase[row * 4 + 2] = a.nf > 2 ? fofs[2] + (a.alpha[t][2] * nn2 + i2) * w : 0;
            rowbase[row * 4 + 3] = 0;
        }
    }
    // per-lane decode of the two elements of chunk (it*64 + lane) inside a row group
    const int lane_id = tid & 63;
    int tp_sel[TP_NIT][2];
    int tp_j[TP_NIT][2][3];
    {
        const int np = a.npts;
#pragma unroll
        for (int it = 0; it < TP_NIT; ++it)
#pragma unroll
            for (int el = 0; el < 2; ++el) {
                const int e = 2 * (it * 64 + lane_id) + el;
                const int sel = e >= np ? 1 : 0;
                const int p = min(e - sel * np, np - 1);
                tp_sel[it][el] = sel;
                int j0 = p, j1 = p, j2 = p;
                if (GRID) {
                    if (a.nf == 2) {
                        j0 = p / a.q;
                        j1 = p - j0 * a.q;
                    } else if (a.nf == 3) {
                        j0 = p / (a.q * a.q);
                        const int rr = p - j0 * a.q * a.q;
                        j1 = rr / a.q;
                        j2 = rr - j1 * a.q;
                    }
                }
                tp_j[it][el][0] = j0;
                tp_j[it][el][1] = j1;
                tp_j[it][el][2] = j2;
            }
    }
    for (long long r = blockIdx.x; r < a.nreq; r += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < a.nf * w; e += nthr) {
            int f = e / w, p = e - f * w;
            const LineDesc& L = a.L[f];
            double x = GRID ? a.pts[((size_t)r * a.nf + f) * a.q + p] : a.pts[((size_t)r * a.npts + p) * a.nf + f];
            if constexpr (NNC > 0) {
                double ta[NNC], tb[NNC];
                lagrange_values_n<NNC>(L, x, ta);
                for (int k = 0; k < K; ++k) {
                    if (k > 0) {
                        lagrange_diff_n<NNC>(L, ta, tb);
#pragma unroll
                        for (int i = 0; i < NNC; ++i) ta[i] = tb[i];
                    }
#pragma unroll
                    for (int i = 0; i < NNC; ++i) T[fofs[f] + (k * NNC + i) * w + p] = ta[i];
                }
                continue;
            }
            double va[NN_MAX], vb[NN_MAX];
            lagrange_values(L, x, va);
            for (int k = 0; k < K; ++k) {
                if (k > 0) {
                    lagrange_diff(L, va, vb);
#pragma unroll
                    for (int i = 0; i < NN_MAX; ++i) va[i] = vb[i];
                }
#pragma unroll
                for (int i = 0; i < NN_MAX; ++i)
                    if (i < L.nn) T[fofs[f] + (k * L.nn + i) * w + p] = va[i];
            }
        }
        __syncthreads();
        const int n0 = a.L[0].nn;
        const int n1 = a.nf > 1 ? a.L[1].nn : 1;
        const int n2 = a.nf > 2 ? a.L[2].nn : 1;
        const int nbf = n0 * n1 * n2;
        const int nrows = a.ntab * nbf;
        const int npts = a.npts;
        const long long total = (long long)nrows * npts;  // doubles of this request, contiguous in HBM
        double* o = a.out + (size_t)r * total;
        // value of table row `row` at point p
        auto value = [&](int row, int p) -> double {
            const int t = idiv_small(row, rinv_nbf);
            const int bf = row - t * nbf;
            const int i0 = idiv_small(bf, rinv_n12);
            const int rem = bf - i0 * (n1 * n2);
            const int i1 = idiv_small(rem, rinv_n2);
            const int i2 = rem - i1 * n2;
            int j0 = p, j1 = p, j2 = p;
            if (GRID) {
                if (a.nf == 1) {
                    j0 = p;
                } else if (a.nf == 2) {
                    j0 = idiv_small(p, rinv_q);
                    j1 = p - j0 * a.q;
                } else {
                    j0 = idiv_small(p, rinv_qq);
                    const int rr = p - j0 * a.q * a.q;
                    j1 = idiv_small(rr, rinv_q);
                    j2 = rr - j1 * a.q;
                }
            }
            double v = T[fofs[0] + (a.alpha[t][0] * n0 + i0) * w + j0];
            if (a.nf > 1) v *= T[fofs[1] + (a.alpha[t][1] * n1 + i1) * w + j1];
            if (a.nf > 2) v *= T[fofs[2] + (a.alpha[t][2] * n2 + i2) * w + j2];
            return v;
        };
        // Fast path: the request is one contiguous block written 16 bytes per lane.  Rows are
        // taken in groups of RG (1 if npts is even, else 2) so that a group is a whole number of
        // 16-byte chunks; which row / point a lane's two elements belong to is the same for every
        // group and is decoded once per kernel (tp_*), the per-group row decode is scalar.
        const int RG = (npts & 1) ? 2 : 1;
        const int gch = (RG * npts) >> 1;  // chunks per group
        if ((total & 1) == 0 && (nrows % RG) == 0 && gch <= 64 * TP_NIT) {
            typedef double v2d_t __attribute__((ext_vector_type(2)));
            v2d_t* o2 = reinterpret_cast<v2d_t*>(o);
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
            const int ngroups = nrows / RG;
            for (int g = wave; g < ngroups; g += nw) {
                // LDS row bases of the (up to two) rows of this group, per factor (broadcast reads)
                int base[2][3];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int row = min(g * RG + s2, nrows - 1);
#pragma unroll
                    for (int f = 0; f < 3; ++f) base[s2][f] = rowbase[row * 4 + f];
                }
                const size_t gofs = (size_t)g * gch;
#pragma unroll
                for (int it = 0; it < TP_NIT; ++it) {
                    const int c = it * 64 + lane_id;
                    if (c < gch) {
                        v2d_t v;
#pragma unroll
                        for (int el = 0; el < 2; ++el) {
                            const int sel = tp_sel[it][el];
                            double x = T[(sel ? base[1][0] : base[0][0]) + tp_j[it][el][0]];
                            if (a.nf > 1) x *= T[(sel ? base[1][1] : base[0][1]) + tp_j[it][el][1]];
                            if (a.nf > 2) x *= T[(sel ? base[1][2] : base[0][2]) + tp_j[it][el][2]];
                            if (el == 0) v.x = x; else v.y = x;
                        }
                        // plain store: rows of q^nf doubles are not multiples of 128 B, neighbouring
                        // groups complete each other's lines in L2 (non-temporal stores measured 35 % slower here)
                        o2[gofs + c] = v;
                    }
                }
            }
        } else {
            const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
            for (int row = wave; row < nrows; row += nw)
                for (int p = lane; p < npts; p += 64) o[(size_t)row * npts + p] = value(row, p);
        }
    }
}

// ---------------------------------------------------------------------------------
// Push-forward of vector-valued tables to the requests' physical cells, in place
// (reference: FiniteElement.mapping(), FIAT/finite_element.py:84-88; the Piola formulas of
// finat/hdivcurl.py:95-191, checked in test/finat/test_point_evaluation.py:53-70).
//   covariant:      phi = J^{-T} Phi          contravariant:  phi = J Phi / det J
// J = dx/dX between the element's own cell and the request's cell = E_req * G with E_req the
// edge matrix (columns v_i - v_0) and G = A0 / 2 (host constant, A0 = map of the element's
// cell to the default simplex).  The derivative tables were taken w.r.t. physical x already
// (fx_tabulate_batch with verts), J is constant per cell: every table gets the same matrix.
// out[t][dof][c][p], vdim == SD.
struct PiolaArgs {
    const double* verts;  // [nreq][SD+1][SD]
    double* out;          // [nreq][ntab][ndof][SD][npts]
    double G[9];
    int ntab, ndof, npts, kind;  // kind 1 covariant, 2 contravariant; 3 / 4 / 5: the double maps of matrix-valued functions
    long long nreq;
    int rb;  // requests per workgroup pass (<= PIOLA_RB)
};

// The matrix of the Piola maps of one cell: J = E G (E: edge matrix of the physical cell, columns v_i - v_0; G = A0 / 2),
// kind 1 / 3 (covariant, double covariant): J^-T, kind 2 / 4 (contravariant): J / det J.
template <int SD> __device__ __forceinline__ void piola_matrix(const double* v, const double* G, int kind, double (&M)[SD][SD]) {
    double E[SD][SD], J[SD][SD];
#pragma unroll
    for (int c = 0; c < SD; ++c)
#pragma unroll
        for (int r = 0; r < SD; ++r) E[r][c] = v[(c + 1) * SD + r] - v[r];
#pragma unroll
    for (int r = 0; r < SD; ++r)
#pragma unroll
        for (int c = 0; c < SD; ++c) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < SD; ++k) t += E[r][k] * G[k * SD + c];
            J[r][c] = t;
        }
    double det, inv[SD][SD];
    if constexpr (SD == 2) {
        det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        inv[0][0] = J[1][1] / det;
        inv[0][1] = -J[0][1] / det;
        inv[1][0] = -J[1][0] / det;
        inv[1][1] = J[0][0] / det;
    } else {
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
        const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
        const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
        inv[0][0] = c00 / det;
        inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
        inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
        inv[1][0] = c01 / det;
        inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
        inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
        inv[2][0] = c02 / det;
        inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
        inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
    }
#pragma unroll
    for (int r = 0; r < SD; ++r)
#pragma unroll
        for (int c = 0; c < SD; ++c) M[r][c] = (kind == 1 || kind == 3) ? inv[c][r] : J[r][c] / det;
}

constexpr int PIOLA_RB = 64;  // requests a workgroup takes at a time (small requests: several per pass)

template <int SD> __global__ __launch_bounds__(256) void piola_apply_kernel(const PiolaArgs a) {
    // Persistent workgroups over blocks of `rb` requests (host: rb * elements per request ~ 2048, <= PIOLA_RB): the first rb
    // threads build the maps of the block's requests, then all threads walk the block's (request, table, dof, point)
    // elements.  (One workgroup per request left 250 of 256 threads idle on N1 / RT1 requests of 18 elements: 5-9 % of
    // the HBM peak for the whole mapped tabulation, tools/coverage_map.py --verts --pushforward.)
    __shared__ double sM[PIOLA_RB][SD * SD];
    __shared__ double sR[PIOLA_RB][SD * SD];  // kind 5: the right-hand matrix (J / det J), the left one is J^-T
    const int rb = a.rb;
    const int groups = a.ntab * a.ndof;  // (table, dof) pairs: SD (or SD * SD) rows of npts each
    const int per = groups * a.npts;     // elements per request
    const int rowsper = a.kind >= 3 ? SD * SD : SD;
    for (long long base = (long long)blockIdx.x * rb; base < a.nreq; base += (long long)gridDim.x * rb) {
        __syncthreads();  // the maps of the previous block are no longer read
        if ((int)threadIdx.x < rb && base + threadIdx.x < a.nreq) {
            double m0[SD][SD];
            piola_matrix<SD>(a.verts + (size_t)(base + threadIdx.x) * (SD + 1) * SD, a.G, a.kind == 5 ? 1 : a.kind, m0);
#pragma unroll
            for (int r = 0; r < SD; ++r)
#pragma unroll
                for (int c = 0; c < SD; ++c) sM[threadIdx.x][r * SD + c] = m0[r][c];
            if (a.kind == 5) {
                piola_matrix<SD>(a.verts + (size_t)(base + threadIdx.x) * (SD + 1) * SD, a.G, 2, m0);
#pragma unroll
                for (int r = 0; r < SD; ++r)
#pragma unroll
                    for (int c = 0; c < SD; ++c) sR[threadIdx.x][r * SD + c] = m0[r][c];
            }
        }
        __syncthreads();
        const int nb = (int)min((long long)rb, a.nreq - base);
        double* o = a.out + (size_t)base * per * rowsper;
        for (int e = threadIdx.x; e < nb * per; e += blockDim.x) {
            const int rl = e / per, rem = e - rl * per;
            const int g = rem / a.npts, p = rem - g * a.npts;
            double m[SD][SD];
#pragma unroll
            for (int r = 0; r < SD; ++r)
#pragma unroll
                for (int c = 0; c < SD; ++c) m[r][c] = sM[rl][r * SD + c];
            double* q = o + ((size_t)rl * groups + g) * rowsper * a.npts + p;
            if (a.kind >= 3) {
                // matrix-valued functions (value shape (SD, SD), row-major: SD*SD rows per dof): M Phi R^T -- double covariant
                // J^-T Phi J^-1 and double contravariant J Phi J^T / det^2 with R = M (Regge, Hellan-Herrmann-Johnson), the
                // mixed map J^-T Phi J^T / det of the GLS elements with M = J^-T, R = J / det
                double X[SD][SD], T[SD][SD], rm[SD][SD];
#pragma unroll
                for (int i = 0; i < SD; ++i)
#pragma unroll
                    for (int j = 0; j < SD; ++j) rm[i][j] = a.kind == 5 ? sR[rl][i * SD + j] : m[i][j];
#pragma unroll
                for (int i = 0; i < SD; ++i)
#pragma unroll
                    for (int j = 0; j < SD; ++j) X[i][j] = q[(i * SD + j) * a.npts];
#pragma unroll
                for (int i = 0; i < SD; ++i)
#pragma unroll
                    for (int j = 0; j < SD; ++j) {
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < SD; ++k) t += m[i][k] * X[k][j];
                        T[i][j] = t;
                    }
#pragma unroll
                for (int i = 0; i < SD; ++i)
#pragma unroll
                    for (int j = 0; j < SD; ++j) {
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < SD; ++k) t += T[i][k] * rm[j][k];
                        q[(i * SD + j) * a.npts] = t;
                    }
            } else {
                double x[SD], y[SD];
#pragma unroll
                for (int c = 0; c < SD; ++c) x[c] = q[c * a.npts];
#pragma unroll
                for (int r = 0; r < SD; ++r) {
                    double t = 0.0;
#pragma unroll
                    for (int c = 0; c < SD; ++c) t += m[r][c] * x[c];
                    y[r] = t;
                }
#pragma unroll
                for (int c = 0; c < SD; ++c) q[c * a.npts] = y[c];
            }
        }
    }
}

}  // namespace fxk
