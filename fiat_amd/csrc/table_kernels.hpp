// Table post-processing for the FInAT side of the boundary:
//  * classify_tables_kernel — the facts finat/fiat_elements.py:92-111 asserts about a tabulation
//    before it becomes a GEM literal (derivative == degree: constant over the points;
//    derivative > degree: zero), computed for a whole batch of GPU-resident tables;
//  * point_major_kernel — [rows][npts] tables to the [npts][rows] layout FInAT passes as kernel
//    arguments (finat/runtime_tabulated.py:79 `shape = point extents + index_shape + value_shape`).
// Both are single HBM passes over the tables.
#pragma once
#include <hip/hip_runtime.h>
#include "simplex_kernel.hpp"  // cell_map

namespace fxk {

struct ClassifyArgs {
    const double* tables;  // [ntables][rows][npts]
    double* stats;         // [ntables][2]: max |x|, max (|x - x[.., 0]| - rtol |x[.., 0]|)
    int rows, npts;
    double rtol;
};

// one workgroup per table
__global__ __launch_bounds__(256) void classify_tables_kernel(const ClassifyArgs a) {
    const double* t = a.tables + (size_t)blockIdx.x * a.rows * a.npts;
    const int n = a.rows * a.npts;
    double amax = 0.0, dev = -1.0e300;
    bool bad = false;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double x = t[i];
        const double x0 = t[i - i % a.npts];
        bad |= !(x == x);  // NaN
        amax = fmax(amax, fabs(x));
        dev = fmax(dev, fabs(x - x0) - a.rtol * fabs(x0));
    }
    __shared__ double s0[256], s1[256];
    __shared__ int sbad;
    if (threadIdx.x == 0) sbad = 0;
    __syncthreads();
    if (bad) atomicOr(&sbad, 1);
    s0[threadIdx.x] = amax;
    s1[threadIdx.x] = dev;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s0[threadIdx.x] = fmax(s0[threadIdx.x], s0[threadIdx.x + w]);
            s1[threadIdx.x] = fmax(s1[threadIdx.x], s1[threadIdx.x + w]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double nan = __builtin_nan("");
        a.stats[2 * (size_t)blockIdx.x + 0] = sbad ? nan : s0[0];
        a.stats[2 * (size_t)blockIdx.x + 1] = sbad ? nan : s1[0];
    }
}

struct SquaredNormArgs {
    const double* tables;   // [ntables][rows][vdim][npts]
    const double* weights;  // [npts]
    double* out;            // [ntables][rows]
    int rows, vdim, npts;
};

// FIAT/finite_element.py:250-260 (entity_support_dofs): out[t][r] = sum_p w_p sum_c tables[t][r][c][p]^2 -- the squared
// L2 norm of every basis function over the entity whose quadrature rule the tables were evaluated at.  One wave per
// (table, row): lanes stride over the vdim * npts entries of the row (contiguous), wave reduction by DPP shuffles.
__global__ __launch_bounds__(256) void squared_norm_kernel(const SquaredNormArgs a, long long nrows) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    const double* t = a.tables + (size_t)row * a.vdim * a.npts;
    const int n = a.vdim * a.npts;
    double acc = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double x = t[i];
        acc += a.weights[i % a.npts] * x * x;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) a.out[row] = acc;
}

struct PointMajorArgs {
    const double* in;  // [ntables][rows][npts]
    double* out;       // [ntables][npts][rows]
    int rows, npts;
};

// one workgroup per table and 32 x 32 tile; LDS tile padded against bank conflicts
__global__ __launch_bounds__(256) void point_major_kernel(const PointMajorArgs a) {
    __shared__ double tile[32][33];
    const int tr = (a.rows + 31) / 32, tp = (a.npts + 31) / 32;
    const size_t tab = blockIdx.x / (tr * tp);
    const int rem = blockIdx.x % (tr * tp);
    const int r0 = (rem / tp) * 32, p0 = (rem % tp) * 32;
    const double* in = a.in + tab * a.rows * a.npts;
    double* out = a.out + tab * a.rows * a.npts;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, p = p0 + tx;
        if (r < a.rows && p < a.npts) tile[j][tx] = in[(size_t)r * a.npts + p];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, r = r0 + tx;
        if (r < a.rows && p < a.npts) out[(size_t)p * a.rows + r] = tile[tx][j];
    }
}

// Chain rule across the derivative tables of a request, in place: the tables were computed with respect
// to the coordinates X of the element's own cell, the request lives on the affine image x of that cell;
//   d/dx_d = sum_c K[c][d] d/dX_c,   d2/dx_d dx_e = sum_{c,c'} K[c][d] K[c'][e] d2/dX_c dX_c',   K = A0^-1 A_req
// (A_req x + b_req and A0 X + b0 are the two maps to the default simplex).  One workgroup per request and
// slice of positions; every thread owns (row, point) positions and all ntab values at them.
constexpr int MIX_RB = 32;  // requests a workgroup takes at a time when a request has fewer positions than the workgroup threads
struct TableMixArgs {
    double* out;          // [nreq][ntab][n]   n = rows * npts
    const double* verts;  // [nreq][SD+1][SD]
    double A0inv[9];
    int n;
    int order;            // 1 or 2
    int slices;           // workgroups per request (large requests), or
    int rb;               // requests per workgroup pass (small requests: slices == 1, rb <= MIX_RB); one of the two is 1
    long long nreq;
};

// K = A0^-1 A_req of one request (thread-local)
template <int SD> __device__ __forceinline__ void mix_matrix(const double* verts, const double* A0inv, double* K) {
    double A[SD][SD], b[SD];
    cell_map<SD>(verts, A, b);
    for (int c = 0; c < SD; ++c)
        for (int d = 0; d < SD; ++d) {
            double t = 0.0;
            for (int k = 0; k < SD; ++k) t += A0inv[c * SD + k] * A[k][d];
            K[c * SD + d] = t;
        }
}

template <int SD> __device__ __forceinline__ void mix_position(double* base, int n, int i, int order, const double (&K)[SD][SD]) {
    constexpr int NH = SD * (SD + 1) / 2;
    double g[SD], og[SD];
#pragma unroll
    for (int c = 0; c < SD; ++c) g[c] = base[(size_t)(1 + c) * n + i];
#pragma unroll
    for (int d = 0; d < SD; ++d) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < SD; ++c) t += K[c][d] * g[c];
        og[d] = t;
    }
#pragma unroll
    for (int d = 0; d < SD; ++d) base[(size_t)(1 + d) * n + i] = og[d];
    if (order >= 2) {
        // Hessian tables in mis() order: (c, c'), c <= c' -> index c (2 SD - c - 1) / 2 + c'
        double H[SD][SD];
#pragma unroll
        for (int c = 0; c < SD; ++c)
#pragma unroll
            for (int e = c; e < SD; ++e) {
                const double v = base[(size_t)(1 + SD + c * (2 * SD - c - 1) / 2 + e) * n + i];
                H[c][e] = v;
                H[e][c] = v;
            }
        double T[SD][SD];  // T[c][e] = sum_c' H[c][c'] K[c'][e]
#pragma unroll
        for (int c = 0; c < SD; ++c)
#pragma unroll
            for (int e = 0; e < SD; ++e) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < SD; ++k) t += H[c][k] * K[k][e];
                T[c][e] = t;
            }
#pragma unroll
        for (int d = 0; d < SD; ++d)
#pragma unroll
            for (int e = d; e < SD; ++e) {
                double t = 0.0;
#pragma unroll
                for (int c = 0; c < SD; ++c) t += K[c][d] * T[c][e];
                base[(size_t)(1 + SD + d * (2 * SD - d - 1) / 2 + e) * n + i] = t;
            }
    }
    (void)NH;
}

// Large requests: one workgroup per (request, slice of positions).  Small requests (fewer positions than a few
// workgroups' worth of threads): persistent workgroups over blocks of `rb` requests -- the first rb threads build the
// blocks' matrices, then all threads walk the block's (request, position) pairs.  One workgroup per request left most
// of its 256 threads idle on requests of 60-500 positions (P2 / P3 triangles, N1 / RT1 ...), and paid a barrier per request.
template <int SD>
__global__ __launch_bounds__(256) void table_mix_kernel(const TableMixArgs a) {
    constexpr int NH = SD * (SD + 1) / 2;
    __shared__ double sK[MIX_RB][SD * SD];
    const int ntab = a.order >= 2 ? 1 + SD + NH : 1 + SD;
    if (a.rb <= 1) {
        const size_t req = blockIdx.x / a.slices;
        const int slice = blockIdx.x % a.slices;
        if (threadIdx.x == 0) mix_matrix<SD>(a.verts + req * (SD + 1) * SD, a.A0inv, sK[0]);
        __syncthreads();
        double K[SD][SD];
#pragma unroll
        for (int c = 0; c < SD; ++c)
#pragma unroll
            for (int d = 0; d < SD; ++d) K[c][d] = sK[0][c * SD + d];
        double* base = a.out + req * (size_t)ntab * a.n;
        for (int i = slice * 256 + threadIdx.x; i < a.n; i += a.slices * 256) mix_position<SD>(base, a.n, i, a.order, K);
        return;
    }
    const float rinv = 1.0f / (float)a.n;
    for (long long r0 = (long long)blockIdx.x * a.rb; r0 < a.nreq; r0 += (long long)gridDim.x * a.rb) {
        __syncthreads();  // the matrices of the previous block are no longer read
        const int nb = (int)min((long long)a.rb, a.nreq - r0);
        if ((int)threadIdx.x < nb) mix_matrix<SD>(a.verts + (size_t)(r0 + threadIdx.x) * (SD + 1) * SD, a.A0inv, sK[threadIdx.x]);
        __syncthreads();
        for (int e = threadIdx.x; e < nb * a.n; e += 256) {
            int rl = (int)((float)e * rinv);
            int i = e - rl * a.n;
            if (i < 0) {
                --rl;
                i += a.n;
            } else if (i >= a.n) {
                ++rl;
                i -= a.n;
            }
            double K[SD][SD];
#pragma unroll
            for (int c = 0; c < SD; ++c)
#pragma unroll
                for (int d = 0; d < SD; ++d) K[c][d] = sK[rl][c * SD + d];
            mix_position<SD>(a.out + (size_t)(r0 + rl) * ntab * a.n, a.n, i, a.order, K);
        }
    }
}

// ---- chain rule across the derivative tables for orders 3 and 4 (per-request cells on the differentiation-matrix route) ----
// The order-0 tabulation of the stacked element [C D^alpha] at points mapped into the element's cell yields derivatives with
// respect to the ELEMENT's coordinates X; with x the request's coordinates and K = dX/dx = A0^-1 A_req,
//   d^alpha_x = sum_beta M_k[alpha][beta] d^beta_X,   M_1[e_d][e_c] = K[c][d],
//   M_k[alpha][beta] = sum_{c: beta_c >= 1} K[c][d] M_{k-1}[alpha - e_d][beta - e_c]   (d = first non-zero entry of alpha)
// -- the symmetric k-th tensor power of K on multi-indices in mis() order.  One workgroup slice per request builds M_1..M_ORDER
// in LDS (host-made index tables say where alpha - e_c sits in the previous order) and mixes every (row, point) entry in place,
// all tables of one order in registers.
constexpr int MIXH_MAXT = 35;  // tables up to order 4 in 3-D
struct TableMixHighArgs {
    double* out;          // [nreq][ntab][n]   n = rows * npts
    const double* verts;  // [nreq][SD+1][SD]
    double A0inv[9];
    int n;
    int slices;          // workgroups per request (large requests; then rb == 1)
    int rb;              // requests per workgroup pass (small requests, <= MIXH_RB; then slices == 1)
    long long nreq;
    signed char down[MIXH_MAXT][3];  // index, within the previous order, of alpha_t - e_c (-1: alpha_t[c] == 0)
    unsigned char lead[MIXH_MAXT];   // first non-zero entry of alpha_t
};

template <int SD, int K>
struct MisCount {
    static constexpr int value = K == 0 ? 1 : (SD == 1 ? 1 : SD == 2 ? K + 1 : (K + 1) * (K + 2) / 2);
};

template <int SD, int ORDER, int K>
__device__ __forceinline__ void mix_one_order(double* base, int n, int i, const double* M, int first) {
    constexpr int CNT = MisCount<SD, K>::value;
    double in[CNT];
#pragma unroll
    for (int s = 0; s < CNT; ++s) in[s] = base[(size_t)(first + s) * n + i];
    // (target loop NOT unrolled: unrolled, hipcc hoists all CNT^2 matrix entries out of the element loop and spills 300 registers)
#pragma unroll 1
    for (int t = 0; t < CNT; ++t) {
        const double* Mt = M + t * CNT;
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < CNT; ++s) acc += Mt[s] * in[s];
        base[(size_t)(first + t) * n + i] = acc;
    }
}

constexpr int MIXH_RB = 8;  // requests per workgroup pass of the small-request path (their M_1..M_ORDER live in LDS: <= 370 doubles each)

template <int SD, int ORDER>
__global__ __launch_bounds__(256) void table_mix_high_kernel(const TableMixHighArgs a) {
    static_assert(ORDER == 3 || ORDER == 4, "orders 3 and 4");
    constexpr int C1 = MisCount<SD, 1>::value, C2 = MisCount<SD, 2>::value, C3 = MisCount<SD, 3>::value, C4 = MisCount<SD, 4>::value;
    constexpr int F1 = 1, F2 = F1 + C1, F3 = F2 + C2, F4 = F3 + C3;   // first table of each order
    constexpr int NTAB = ORDER == 3 ? F4 : F4 + C4;
    constexpr int O1 = 0, O2 = O1 + C1 * C1, O3 = O2 + C2 * C2, O4 = O3 + C3 * C3, MTOT = O4 + (ORDER >= 4 ? C4 * C4 : 0);
    __shared__ double sK[MIXH_RB][SD * SD];
    __shared__ double sM[MIXH_RB][MTOT];
    // Large requests (rb == 1): one workgroup per (request, slice of positions).  Small requests: persistent workgroups over
    // blocks of rb requests, as table_mix_kernel / piola_apply_kernel -- one workgroup per request spent four barriers and a
    // serial K on requests of a few hundred positions.
    const bool blocked = a.rb > 1;
    const float rinv = 1.0f / (float)a.n;
    const long long nblocks = blocked ? (a.nreq + a.rb - 1) / a.rb : 0;
    for (long long blk = blockIdx.x; blocked ? blk < nblocks : blk == (long long)blockIdx.x; blk += gridDim.x) {
        const long long r0 = blocked ? blk * a.rb : (long long)(blockIdx.x / a.slices);
        const int slice = blocked ? 0 : (int)(blockIdx.x % a.slices);
        const int nb = blocked ? (int)min((long long)a.rb, a.nreq - r0) : 1;
        __syncthreads();  // the matrices of the previous block are no longer read
        if ((int)threadIdx.x < nb) {
            double A[SD][SD], b[SD];
            cell_map<SD>(a.verts + (size_t)(r0 + threadIdx.x) * (SD + 1) * SD, A, b);
            for (int c = 0; c < SD; ++c)
                for (int d = 0; d < SD; ++d) {
                    double t = 0.0;
                    for (int k = 0; k < SD; ++k) t += a.A0inv[c * SD + k] * A[k][d];
                    sK[threadIdx.x][c * SD + d] = t;
                }
        }
        __syncthreads();
        // M_k[t][s] from M_{k-1}; tables of order k are first..first+cnt-1
        auto build = [&](int ok, int op, bool has_prev, int first, int cnt, int pcnt) {
            for (int idx = threadIdx.x; idx < nb * cnt * cnt; idx += 256) {
                const int rl = idx / (cnt * cnt), e = idx - rl * cnt * cnt;
                const int t = e / cnt, s2 = e - t * cnt;
                const int d = a.lead[first + t];
                const int tp = a.down[first + t][d];
                double acc = 0.0;
                for (int c = 0; c < SD; ++c) {
                    const int sp = a.down[first + s2][c];
                    if (sp >= 0) acc += sK[rl][c * SD + d] * (has_prev ? sM[rl][op + tp * pcnt + sp] : 1.0);
                }
                sM[rl][ok + e] = acc;
            }
            __syncthreads();
        };
        build(O1, 0, false, F1, C1, 1);
        build(O2, O1, true, F2, C2, C1);
        build(O3, O2, true, F3, C3, C2);
        if constexpr (ORDER >= 4) build(O4, O3, true, F4, C4, C3);
        const int total = nb * a.n;
        for (int e = slice * 256 + threadIdx.x; e < total; e += a.slices * 256) {
            int rl = 0, i = e;
            if (blocked) {
                rl = (int)((float)e * rinv);
                i = e - rl * a.n;
                if (i < 0) {
                    --rl;
                    i += a.n;
                } else if (i >= a.n) {
                    ++rl;
                    i -= a.n;
                }
            }
            double* base = a.out + (size_t)(r0 + rl) * NTAB * a.n;
            const double* M = sM[rl];
            mix_one_order<SD, ORDER, 1>(base, a.n, i, M + O1, F1);
            mix_one_order<SD, ORDER, 2>(base, a.n, i, M + O2, F2);
            mix_one_order<SD, ORDER, 3>(base, a.n, i, M + O3, F3);
            if constexpr (ORDER >= 4) mix_one_order<SD, ORDER, 4>(base, a.n, i, M + O4, F4);
        }
        if (!blocked) break;
    }
}

// ---- the same chain rule at ANY order up to FX_MAX_ORDER (orders 5..8 with per-request cells; round 4) ----
// Run-time order: M_1..M_order in dynamic LDS (3-D, order 8: 4916 doubles), one workgroup per (request, slice of positions);
// the tables of one order of a position in registers (<= 45 in 3-D).  FIAT/expansions.py:411-447 applies the chain rule through
// Jinv at any order; this is its per-request form.
constexpr int MIXA_MAXT = 165;   // tables up to order 8 in 3-D
struct TableMixAnyArgs {
    double* out;          // [nreq][ntab][n]   n = rows * npts
    const double* verts;  // [nreq][SD+1][SD]
    double A0inv[9];
    int n, slices, order;
    long long nreq;
    int first[10], cnt[10], moff[10];   // per order k: first table, number of tables, offset of M_k in LDS (doubles)
    signed char down[MIXA_MAXT][3];     // index, within the previous order, of alpha_t - e_c (-1: alpha_t[c] == 0)
    unsigned char lead[MIXA_MAXT];      // first non-zero entry of alpha_t
};

template <int SD>
__global__ __launch_bounds__(256) void table_mix_any_kernel(const TableMixAnyArgs a) {
    constexpr int CMAX = SD == 1 ? 1 : SD == 2 ? 9 : 45;   // tables of one order, order <= 8
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* sK = sm;            // [SD * SD]
    double* sM = sm + 16;       // M_1 .. M_order
    const long long r = (long long)(blockIdx.x / a.slices);
    const int slice = (int)(blockIdx.x % a.slices);
    if (threadIdx.x == 0) {
        double A[SD][SD], b[SD];
        cell_map<SD>(a.verts + (size_t)r * (SD + 1) * SD, A, b);
        for (int c = 0; c < SD; ++c)
            for (int d = 0; d < SD; ++d) {
                double t = 0.0;
                for (int k = 0; k < SD; ++k) t += a.A0inv[c * SD + k] * A[k][d];
                sK[c * SD + d] = t;
            }
    }
    __syncthreads();
    for (int k = 1; k <= a.order; ++k) {  // M_k[t][s] from M_{k-1}
        const int cnt = a.cnt[k], pcnt = a.cnt[k - 1], first = a.first[k];
        for (int e = threadIdx.x; e < cnt * cnt; e += 256) {
            const int t = e / cnt, s2 = e - t * cnt;
            const int d = a.lead[first + t];
            const int tp = a.down[first + t][d];
            double acc = 0.0;
            for (int c = 0; c < SD; ++c) {
                const int sp = a.down[first + s2][c];
                if (sp >= 0) acc += sK[c * SD + d] * (k > 1 ? sM[a.moff[k - 1] + tp * pcnt + sp] : 1.0);
            }
            sM[a.moff[k] + e] = acc;
        }
        __syncthreads();
    }
    const int ntab = a.first[a.order] + a.cnt[a.order];
    double* base = a.out + (size_t)r * ntab * a.n;
    for (int i = slice * 256 + threadIdx.x; i < a.n; i += a.slices * 256) {
        for (int k = 1; k <= a.order; ++k) {
            const int cnt = a.cnt[k], first = a.first[k];
            const double* M = sM + a.moff[k];
            double in[CMAX];
#pragma unroll
            for (int s = 0; s < CMAX; ++s) in[s] = s < cnt ? base[(size_t)(first + s) * a.n + i] : 0.0;
#pragma unroll 1
            for (int t = 0; t < cnt; ++t) {
                const double* Mt = M + t * cnt;
                double acc = 0.0;
#pragma unroll
                for (int s = 0; s < CMAX; ++s)
                    if (s < cnt) acc += Mt[s] * in[s];
                base[(size_t)(first + t) * a.n + i] = acc;
            }
        }
    }
}

// ---- tensor products of ANY two tabulated factors (TensorProductElement.tabulate, FIAT/tensor_product.py:231-336) ----
// out[r][t][a * rowsB + b][c][p] = A[r][tA(t)][a][cA][p] * B[r][tB(t)][b][cB][p]   for alpha_t = (alpha_A, alpha_B):
// scalar x scalar (:274-292), vector x scalar (:293-317) and scalar x vector (:318-335) are the same formula with the
// component index c riding on whichever factor is vector-valued.  The factor tables come from the factors' own kernels
// (simplex elements, 1-D Lagrange, nested products); this pass is write-bound: 8 * rowsA * rowsB * vdim * npts bytes per
// table against 8 * (rowsA * vdimA + rowsB * vdimB) * npts read (L2 hits: every factor entry is reused rowsB / rowsA times).
constexpr int OUTER_MAXTAB = 220;  // tables per request: C(sd + order, sd) for sd = 3, order = 9

struct OuterArgs {
    const double* A;  // [nreq][ntabA][rowsA][vdimA][npts]
    const double* B;  // [nreq][ntabB][rowsB][vdimB][npts]
    double* out;      // [nreq][ntab][rowsA * rowsB][vdim][npts]
    long long nreq;
    int ntab, ntabA, ntabB, rowsA, rowsB, vdimA, vdimB, npts;
    unsigned char tA[OUTER_MAXTAB], tB[OUTER_MAXTAB];
};

// x / d and x % d for x < 2^24 (single-precision reciprocal + correction)
__device__ __forceinline__ void divmod24(unsigned x, unsigned d, float inv, unsigned& q, unsigned& r) {
    q = (unsigned)((float)x * inv);
    int rem = (int)x - (int)(q * d);
    if (rem < 0) {
        --q;
        rem += (int)d;
    } else if (rem >= (int)d) {
        ++q;
        rem -= (int)d;
    }
    r = (unsigned)rem;
}

__global__ __launch_bounds__(256) void table_outer_kernel(const OuterArgs a) {
    const int vdim = a.vdimA > a.vdimB ? a.vdimA : a.vdimB;
    const unsigned npts = (unsigned)a.npts;
    const unsigned perB = (unsigned)(a.rowsB * vdim) * npts;   // elements of one `a` block
    const unsigned table = (unsigned)a.rowsA * perB;           // elements of one output table (< 2^24, checked on the host)
    const float inv_perB = 1.0f / (float)perB, inv_npts = 1.0f / (float)npts, inv_vdim = 1.0f / (float)vdim;
    const long long units = a.nreq * a.ntab;
    for (long long u = blockIdx.x; u < units; u += gridDim.x) {
        const long long r = u / a.ntab;
        const int t = (int)(u - r * a.ntab);
        const double* At = a.A + ((size_t)r * a.ntabA + a.tA[t]) * (size_t)(a.rowsA * a.vdimA) * npts;
        const double* Bt = a.B + ((size_t)r * a.ntabB + a.tB[t]) * (size_t)(a.rowsB * a.vdimB) * npts;
        double* o = a.out + (size_t)u * table;
        for (unsigned j = threadIdx.x; j < table; j += 256) {
            unsigned ia, rem, bc, p, ib, c;
            divmod24(j, perB, inv_perB, ia, rem);
            divmod24(rem, npts, inv_npts, bc, p);
            divmod24(bc, (unsigned)vdim, inv_vdim, ib, c);
            const double x = a.vdimA > 1 ? At[((size_t)ia * a.vdimA + c) * npts + p] : At[(size_t)ia * npts + p];
            const double y = a.vdimB > 1 ? Bt[((size_t)ib * a.vdimB + c) * npts + p] : Bt[(size_t)ib * npts + p];
            __builtin_nontemporal_store(x * y, o + j);
        }
    }
}

// The same product with the factor tables of a group of requests staged in LDS (the default; the kernel above is the
// fallback for factor tables that do not fit).  A persistent workgroup takes groups of G requests: the A and B tables of the
// NEXT group are fetched into registers while the current group is multiplied out of LDS, a row table (LDS offsets of the
// A row and the B row of every output row, built once per workgroup) replaces the per-element index decoding, and the
// output leaves as coalesced 8-byte stores of consecutive elements (a group's output is one contiguous block).
constexpr int OUTER_TILE = 4096;   // doubles of factor tables per group and buffer (32 KB; two buffers)
constexpr int OUTER_PF = OUTER_TILE / 256;

struct OuterLdsArgs {
    OuterArgs o;
    int G;            // requests per group
    int sizeA, sizeB; // doubles of one request's A / B tables
    int nrows;        // output rows of one request: ntab * rowsA * rowsB * vdim
};

__global__ __launch_bounds__(256) void table_outer_lds_kernel(const OuterLdsArgs a) {
    extern __shared__ __attribute__((aligned(16))) double olds[];
    const OuterArgs& o = a.o;
    const int vdim = o.vdimA > o.vdimB ? o.vdimA : o.vdimB;
    const int npts = o.npts;
    int* rowtab = reinterpret_cast<int*>(olds);                 // [nrows][2]
    double* buf0 = olds + ((2 * a.nrows + 1) / 2 + 1) / 2 * 2;   // 16-byte aligned
    double* buf1 = buf0 + OUTER_TILE;
    // row table: output row (t, ia, ib, c) -> start of its A row and its B row inside one request's staged tables
    for (int r = threadIdx.x; r < a.nrows; r += 256) {
        int rest = r;
        const int c = rest % vdim;
        rest /= vdim;
        const int ib = rest % o.rowsB;
        rest /= o.rowsB;
        const int ia = rest % o.rowsA;
        const int t = rest / o.rowsA;
        rowtab[2 * r] = ((o.tA[t] * o.rowsA + ia) * o.vdimA + (o.vdimA > 1 ? c : 0)) * npts;
        rowtab[2 * r + 1] = a.sizeA + ((o.tB[t] * o.rowsB + ib) * o.vdimB + (o.vdimB > 1 ? c : 0)) * npts;
    }
    const int per = a.sizeA + a.sizeB;                // staged doubles per request
    const long long ngroups = (o.nreq + a.G - 1) / a.G;
    const int S = a.nrows * npts;                     // output elements per request
    const float inv_S = 1.0f / (float)S, inv_npts = 1.0f / (float)npts, inv_per = 1.0f / (float)per;
    double pf[OUTER_PF];
#pragma unroll
    for (int k = 0; k < OUTER_PF; ++k) pf[k] = 0.0;
    auto fetch = [&](long long grp) {                 // factor tables of group `grp` -> registers
        const long long r0 = grp * a.G;
        const int g_here = (int)(o.nreq - r0 < a.G ? o.nreq - r0 : a.G);
        const unsigned total = (unsigned)(g_here * per);
#pragma unroll
        for (int k = 0; k < OUTER_PF; ++k) {
            const unsigned e = (unsigned)(k * 256 + threadIdx.x);
            double v = 0.0;
            if (e < total) {
                unsigned g, w;
                divmod24(e, (unsigned)per, inv_per, g, w);
                v = (int)w < a.sizeA ? o.A[(size_t)(r0 + g) * a.sizeA + w] : o.B[(size_t)(r0 + g) * a.sizeB + (w - a.sizeA)];
            }
            pf[k] = v;
        }
    };
    long long grp = blockIdx.x;
    if (grp < ngroups) fetch(grp);
    double* cur = buf0;
    double* nxt = buf1;
#pragma unroll
    for (int k = 0; k < OUTER_PF; ++k) cur[k * 256 + threadIdx.x] = pf[k];
    __syncthreads();
    for (; grp < ngroups; grp += gridDim.x) {
        const long long gnext = grp + gridDim.x;
        if (gnext < ngroups) fetch(gnext);
        const long long r0 = grp * a.G;
        const int g_here = (int)(o.nreq - r0 < a.G ? o.nreq - r0 : a.G);
        const unsigned total = (unsigned)g_here * (unsigned)S;
        double* out = o.out + (size_t)r0 * S;
        for (unsigned e = threadIdx.x; e < total; e += 256) {
            unsigned g, j, row, p;
            divmod24(e, (unsigned)S, inv_S, g, j);
            divmod24(j, (unsigned)npts, inv_npts, row, p);
            const double* base = cur + g * per;
            const double x = base[rowtab[2 * row] + p];
            const double y = base[rowtab[2 * row + 1] + p];
            __builtin_nontemporal_store(x * y, out + e);
        }
        if (gnext < ngroups) {
#pragma unroll
            for (int k = 0; k < OUTER_PF; ++k) nxt[k * 256 + threadIdx.x] = pf[k];
        }
        __syncthreads();
        double* tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
}

// ---- affine map of points: entity coordinates -> cell coordinates (reference_element.py:570-609) -----------------
// out[i] = M in[i] + b with M (dout x din, din may be 0: every point becomes the vertex b), for `n` points.
struct MapPointsArgs {
    const double* in;  // [n][din]
    double* out;       // [n][dout]
    long long n;
    int din, dout;
    double M[9], b[3];
};

__global__ __launch_bounds__(256) void map_points_kernel(const MapPointsArgs a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        double x[3] = {0.0, 0.0, 0.0};
        for (int d = 0; d < a.din; ++d) x[d] = a.in[(size_t)i * a.din + d];
        for (int e = 0; e < a.dout; ++e) {
            double y = a.b[e];
            for (int d = 0; d < a.din; ++d) y += a.M[e * a.din + d] * x[d];
            a.out[(size_t)i * a.dout + e] = y;
        }
    }
}

}  // namespace fxk
