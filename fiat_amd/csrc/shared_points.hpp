// Tabulation of one reference point set pushed forward to many cells (gfx950).
#pragma once
#include "simplex_kernel.hpp"
#include "store.hpp"

namespace fxk {

// ---------------------------------------------------------------------------------
// Same reference points in every cell (the quadrature-rule case): the tables of the element on
// its own cell are computed once (`ref`, [ntab][ndof][vdim][npts], derivatives w.r.t. the own
// cell's coordinates x^); request r is the affine image of that cell and gets
//   values        phi            (copied, or mixed by the Piola matrix)
//   gradients     d/dx_d       = sum_c K[c][d] d/dx^_c
//   Hessians      d2/dx_d dx_e = sum_{c,c'} K[c][d] K[c'][e] d2/dx^_c dx^_c'
// with K = dx^/dx = A0^{-1} A_r (A_r: cell map of request r to the default simplex, A0: the own
// cell's).  Piola maps reuse K: covariant J^{-T} = K^T, contravariant J / det J = adj(K).
// This is what FIAT's consumers do (tabulate on the reference cell once, FInAT/TSFC push forward
// per cell); here it is one streaming kernel: reads hit L2 (the reference tables are a few kB),
// the output is written once.  References: FIAT/finite_element.py:84-88 (mapping),
// finat/fiat_elements.py:69 (reference tabulation), finat/hdivcurl.py:95-191 (Piola).
struct SharedArgs {
    const double* ref;    // [ntab][rows][npts] tables on the element's own cell
    const double* verts;  // [nreq][SD+1][SD]
    double* out;          // [nreq][ntab][rows][npts]
    double A0inv[9];
    long long nreq;
    int rows, vdim, npts, kind;  // kind: 0 affine, 1 covariant Piola, 2 contravariant Piola
    int rb;                      // register-resident kernel: requests per block (<= SHARED_RB; fewer when the batch is small, so that every CU has work)
};

template <int SD, int ORDER> __global__ __launch_bounds__(256) void shared_points_kernel(const SharedArgs a) {
    constexpr int NTAB = NTab<SD, ORDER>::value;
    constexpr int NH = SD * (SD + 1) / 2;
    __shared__ double sK[SD * SD], sP[SD * SD], sH[NH * NH];
    const int table = a.rows * a.npts;
    const int total = NTAB * table;
    for (long long req = blockIdx.x; req < a.nreq; req += gridDim.x) {
        __syncthreads();  // the matrices of the previous request are no longer read
        if (threadIdx.x == 0) {
            double A[SD][SD], b[SD], K[SD][SD];
            cell_map<SD>(a.verts + (size_t)req * (SD + 1) * SD, A, b);
            for (int c = 0; c < SD; ++c)
                for (int d = 0; d < SD; ++d) {
                    double t = 0.0;
                    for (int k = 0; k < SD; ++k) t += a.A0inv[c * SD + k] * A[k][d];
                    K[c][d] = t;
                    sK[c * SD + d] = t;
                }
            // Piola matrix P[c][c'] (identity for affine elements)
            for (int c = 0; c < SD; ++c)
                for (int e = 0; e < SD; ++e) {
                    double v = c == e ? 1.0 : 0.0;
                    if (a.kind == 1) v = K[e][c];
                    if (a.kind == 2) {
                        if constexpr (SD == 1) v = 1.0;
                        else if constexpr (SD == 2) v = (c == e ? K[1 - c][1 - e] : -K[c][e]);
                        else {
                            const int c1 = (c + 1) % 3, c2 = (c + 2) % 3, e1 = (e + 1) % 3, e2 = (e + 2) % 3;
                            v = K[e1][c1] * K[e2][c2] - K[e1][c2] * K[e2][c1];  // adj(K)[c][e] = cofactor(K)[e][c]
                        }
                    }
                    sP[c * SD + e] = v;
                }
            if constexpr (ORDER >= 2) {
                // Hessian components in mis() order: (d,e), d <= e  <-  (c,c'), c <= c'
                int hd = 0;
                for (int d = 0; d < SD; ++d)
                    for (int e = d; e < SD; ++e, ++hd) {
                        int hc = 0;
                        for (int c = 0; c < SD; ++c)
                            for (int c2 = c; c2 < SD; ++c2, ++hc)
                                sH[hd * NH + hc] = c == c2 ? K[c][d] * K[c][e] : K[c][d] * K[c2][e] + K[c2][d] * K[c][e];
                    }
            }
        }
        __syncthreads();
        double* o = a.out + (size_t)req * total;
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
            const int t = idx / table, rem = idx - t * table;
            const int row = rem / a.npts, p = rem - row * a.npts;
            // source tables and weights of output table t
            int s0, ns;
            const double* w;
            if (t == 0) {
                s0 = 0; ns = 1; w = nullptr;
            } else if (t <= SD) {
                s0 = 1; ns = SD; w = nullptr;
            } else {
                s0 = 1 + SD; ns = NH; w = sH + (t - 1 - SD) * NH;
            }
            double acc = 0.0;
            if (a.kind == 0 || a.vdim != SD) {
                for (int s = 0; s < ns; ++s) {
                    const double ws = t == 0 ? 1.0 : (t <= SD ? sK[s * SD + (t - 1)] : w[s]);
                    acc += ws * a.ref[(size_t)(s0 + s) * table + rem];
                }
            } else {
                const int dof = row / SD, c = row - dof * SD;
                for (int s = 0; s < ns; ++s) {
                    const double ws = t == 0 ? 1.0 : (t <= SD ? sK[s * SD + (t - 1)] : w[s]);
                    double v = 0.0;
                    for (int e = 0; e < SD; ++e) v += sP[c * SD + e] * a.ref[(size_t)(s0 + s) * table + (dof * SD + e) * a.npts + p];
                    acc += ws * v;
                }
            }
            stream_store(&o[idx], acc);
        }
    }
}

// Tiny requests (a few doubles to ~2 KB of tables per request: P0 / P1 / DG1, N1 / RT1 at their small rules): the kernels below
// spend a workgroup pass (or a wave) per request and ran at 4-27 % of the HBM peak there (tools/coverage_map_cells.py).  Here a
// wave takes 64 requests at a time: lane r builds K (and the Piola matrix) of request r into LDS, then the lanes walk the
// CONTIGUOUS output of the 64 requests, lane <-> consecutive doubles -- request, table and position by two divisions, the
// (at most 10) reference values of a position through L1 -- so that every store instruction writes 512 consecutive bytes.
template <int SD, int ORDER> __global__ __launch_bounds__(256) void shared_points_flat_kernel(const SharedArgs a) {
    constexpr int NTAB = NTab<SD, ORDER>::value;
    constexpr int NH = SD * (SD + 1) / 2;
    __shared__ double sK[4][64][SD * SD], sP[4][64][SD * SD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int table = a.rows * a.npts;
    const int total = NTAB * table;
    const float rtotal = 1.0f / (float)total, rtable = 1.0f / (float)table, rnpts = 1.0f / (float)a.npts;
    const bool piola = a.kind != 0 && a.vdim == SD;
    for (long long base = ((long long)blockIdx.x * 4 + wave) * 64; base < a.nreq; base += (long long)gridDim.x * 4 * 64) {
        {
            const long long rq = min(base + lane, a.nreq - 1);
            double A[SD][SD], b[SD], K[SD][SD];
            cell_map<SD>(a.verts + (size_t)rq * (SD + 1) * SD, A, b);
#pragma unroll
            for (int c = 0; c < SD; ++c)
#pragma unroll
                for (int d = 0; d < SD; ++d) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < SD; ++k) t += a.A0inv[c * SD + k] * A[k][d];
                    K[c][d] = t;
                    sK[wave][lane][c * SD + d] = t;
                }
#pragma unroll
            for (int c = 0; c < SD; ++c)
#pragma unroll
                for (int e = 0; e < SD; ++e) {
                    double v = c == e ? 1.0 : 0.0;
                    if (a.kind == 1) v = K[e][c];
                    if (a.kind == 2) {
                        if constexpr (SD == 2) v = (c == e ? K[1 - c][1 - e] : -K[c][e]);
                        else if constexpr (SD == 3) {
                            constexpr int nx[3] = {1, 2, 0}, nn[3] = {2, 0, 1};
                            v = K[nx[e]][nx[c]] * K[nn[e]][nn[c]] - K[nx[e]][nn[c]] * K[nn[e]][nx[c]];
                        } else v = 1.0;
                    }
                    sP[wave][lane][c * SD + e] = v;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        const int nblk = (int)min(64LL, a.nreq - base);
        double* o = a.out + (size_t)base * total;
        for (int pos = lane; pos < nblk * total; pos += 64) {
            int rq = (int)((float)pos * rtotal);          // (exact after one correction step: pos < 64 * 4096)
            rq -= (rq * total > pos);
            rq += ((rq + 1) * total <= pos);
            const int e = pos - rq * total;
            int t = (int)((float)e * rtable);
            t -= (t * table > e);
            t += ((t + 1) * table <= e);
            const int rem = e - t * table;
            const double* K = sK[wave][rq];
            // source tables of output table t and their weights: values <- values; gradient d <- column d of K;
            // Hessian (d, e) <- K (x) K symmetrised (mis() order)
            int s0 = 0, ns = 1;
            if (t >= 1 && t <= SD) s0 = 1, ns = SD;
            if (t > SD) s0 = 1 + SD, ns = NH;
            int hd = 0, he = 0;   // (d, e) of Hessian table t
            if constexpr (ORDER >= 2) {
                if (t > SD) {
                    int q = t - 1 - SD;
                    while (q >= SD - hd) q -= SD - hd, ++hd;
                    he = hd + q;
                }
            }
            int row = 0, p = rem, dof = 0, cmp = 0;
            if (piola) {
                row = (int)((float)rem * rnpts);
                row -= (row * a.npts > rem);
                row += ((row + 1) * a.npts <= rem);
                p = rem - row * a.npts;
                dof = row / SD;
                cmp = row - dof * SD;
            }
            double acc = 0.0;
            int hc = 0, c1 = 0, c2 = 0;   // running (c, c') of the Hessian source tables
            for (int sidx = 0; sidx < ns; ++sidx) {
                double ws = 1.0;
                if (t >= 1 && t <= SD) ws = K[sidx * SD + (t - 1)];
                if constexpr (ORDER >= 2) {
                    if (t > SD) {
                        ws = c1 == c2 ? K[c1 * SD + hd] * K[c1 * SD + he] : K[c1 * SD + hd] * K[c2 * SD + he] + K[c2 * SD + hd] * K[c1 * SD + he];
                        if (++c2 == SD) ++c1, c2 = c1;
                    }
                }
                double v;
                if (piola) {
                    v = 0.0;
                    for (int q = 0; q < SD; ++q) v += sP[wave][rq][cmp * SD + q] * a.ref[(size_t)(s0 + sidx) * table + (dof * SD + q) * a.npts + p];
                } else {
                    v = a.ref[(size_t)(s0 + sidx) * table + rem];
                }
                acc += ws * v;
            }
            (void)hc;
            o[pos] = acc;   // (plain: consecutive lanes, consecutive doubles; the blocks of neighbouring waves share lines)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
    }
}

// Register-resident version: every thread of the (persistent) workgroup owns NP fixed pairs of
// consecutive output positions of a table and keeps the reference values of all source tables for
// them in registers (for Piola maps: of the SD components of the dof each position belongs to).
// Per request it takes K from LDS (computed for 256 requests at a time, one per thread), forms
// adj(K) / sym^2 K in registers and writes its positions with 16-byte non-temporal stores: no
// reads besides 8*(SD+1)*SD bytes per request.  Needs table = rows*npts <= 256*EL*NP.
constexpr int SHARED_RB = 64;  // requests per block of the register-resident kernel

// EL = 2: a thread owns pairs of consecutive doubles (16-byte stores, table even); EL = 1: single
// doubles (odd tables, e.g. RT2 with 23 points: 45 x 23)
template <int SD, int ORDER, int NP, bool PIOLA, int EL = 2>
__global__ __launch_bounds__(256) void shared_points_reg_kernel(const SharedArgs a) {
    constexpr int NTAB = NTab<SD, ORDER>::value;
    constexpr int NH = SD * (SD + 1) / 2;
    constexpr int NE = PIOLA ? SD : 1;
    const int table = a.rows * a.npts;
    const int npairs = EL == 2 ? table >> 1 : table;  // units (pairs or single doubles) per table
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    // reference values: [slot][element of the pair][source table][component]
    double rv[NP][EL][NTAB][NE];
    int comp[NP][EL];  // component (row % SD) of each element, for the Piola matrix row
    // blockIdx.y = slice of the table: this workgroup owns the units [pair0, pair0 + 256 NP) of every table, so that the
    // register-resident reference values stay within budget for any table size (NP * EL * NTAB * NE doubles per thread)
    const int pair0 = (int)blockIdx.y * NP * 256;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int pr = min(pair0 + (int)threadIdx.x + 256 * i, npairs - 1);
#pragma unroll
        for (int el = 0; el < EL; ++el) {
            const int q = EL * pr + el;
            const int row = q / a.npts, p = q - row * a.npts;
            comp[i][el] = PIOLA ? row % SD : 0;
            const int dof = PIOLA ? row / SD : 0;
#pragma unroll
            for (int s = 0; s < NTAB; ++s)
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    rv[i][el][s][e] = PIOLA ? a.ref[(size_t)s * table + (dof * SD + e) * a.npts + p] : a.ref[(size_t)s * table + q];
        }
    }
    // K of SHARED_RB requests at a time: thread i < SHARED_RB inverts the cell of request base + i
    // (the divisions and the 3x3 inverse, ~150 fp64 instructions, would otherwise be repeated by
    // every thread for every request), the workgroup then walks through them reading K from LDS.
    __shared__ double sK[SHARED_RB][SD * SD];
    const int RB = a.rb;
    for (long long base = (long long)blockIdx.x * RB; base < a.nreq; base += (long long)gridDim.x * RB) {
        __syncthreads();  // sK of the previous block is no longer read
        if ((int)threadIdx.x < RB) {
            const long long rq = min(base + (long long)threadIdx.x, a.nreq - 1);
            double A[SD][SD], b[SD];
            cell_map<SD>(a.verts + (size_t)rq * (SD + 1) * SD, A, b);
#pragma unroll
            for (int c = 0; c < SD; ++c)
#pragma unroll
                for (int d = 0; d < SD; ++d) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < SD; ++k) t += a.A0inv[c * SD + k] * A[k][d];
                    sK[threadIdx.x][c * SD + d] = t;
                }
        }
        __syncthreads();
        const int nblk = (int)min((long long)RB, a.nreq - base);
      for (int rb = 0; rb < nblk; ++rb) {
        const long long req = base + rb;
        double K[SD][SD];
#pragma unroll
        for (int c = 0; c < SD; ++c)
#pragma unroll
            for (int d = 0; d < SD; ++d) K[c][d] = sK[rb][c * SD + d];
        double P[SD][SD];
        if constexpr (PIOLA) {
#pragma unroll
            for (int c = 0; c < SD; ++c)
#pragma unroll
                for (int e = 0; e < SD; ++e) {
                    double v;
                    if (a.kind == 1) {
                        v = K[e][c];
                    } else if constexpr (SD == 2) {
                        v = (c == e ? K[1 - c][1 - e] : -K[c][e]);
                    } else if constexpr (SD == 3) {
                        constexpr int nx[3] = {1, 2, 0}, nn[3] = {2, 0, 1};
                        v = K[nx[e]][nx[c]] * K[nn[e]][nn[c]] - K[nx[e]][nn[c]] * K[nn[e]][nx[c]];
                    } else {
                        v = 1.0;
                    }
                    P[c][e] = v;
                }
        }
        double H[NH][NH];
        if constexpr (ORDER >= 2) {
            int hd = 0;
#pragma unroll
            for (int d = 0; d < SD; ++d)
#pragma unroll
                for (int e = d; e < SD; ++e, ++hd) {
                    int hc = 0;
#pragma unroll
                    for (int c = 0; c < SD; ++c)
#pragma unroll
                        for (int c2 = c; c2 < SD; ++c2, ++hc)
                            H[hd][hc] = c == c2 ? K[c][d] * K[c][e] : K[c][d] * K[c2][e] + K[c2][d] * K[c][e];
                }
        }
        double* o1 = a.out + (size_t)req * NTAB * table;
        v2d_t* o2 = reinterpret_cast<v2d_t*>(o1);
        (void)o2;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int pr = pair0 + (int)threadIdx.x + 256 * i;
            if (pr < npairs) {
                // source values after the Piola mix: m[el][s]
                double m[EL][NTAB];
#pragma unroll
                for (int el = 0; el < EL; ++el)
#pragma unroll
                    for (int s = 0; s < NTAB; ++s) {
                        if constexpr (PIOLA) {
                            double v = 0.0;
#pragma unroll
                            for (int e = 0; e < SD; ++e) {
                                // row comp[i][el] of P, selected without dynamic register indexing
                                double pc = P[0][e];
#pragma unroll
                                for (int c = 1; c < SD; ++c) pc = comp[i][el] == c ? P[c][e] : pc;
                                v += pc * rv[i][el][s][e];
                            }
                            m[el][s] = v;
                        } else {
                            m[el][s] = rv[i][el][s][0];
                        }
                    }
#pragma unroll
                for (int t = 0; t < NTAB; ++t) {
                    double vv[2] = {0.0, 0.0};
#pragma unroll
                    for (int el = 0; el < EL; ++el) {
                        double acc;
                        if (t == 0) {
                            acc = m[el][0];
                        } else if (t <= SD) {
                            acc = 0.0;
#pragma unroll
                            for (int c = 0; c < SD; ++c) acc += K[c][t - 1] * m[el][1 + c];
                        } else {
                            acc = 0.0;
                            if constexpr (ORDER >= 2) {
#pragma unroll
                                for (int hc = 0; hc < NH; ++hc) acc += H[t - 1 - SD][hc] * m[el][1 + SD + hc];
                            }
                        }
                        vv[el] = acc;
                    }
                    // plain stores: chunks are not line-aligned, neighbours complete the lines in L2
                    if constexpr (EL == 2) o2[(size_t)t * npairs + pr] = v2d_t{vv[0], vv[1]};
                    else o1[(size_t)t * npairs + pr] = vv[0];
                }
            }
        }
      }
    }
}

// Small affine requests (order <= 1, at most 64*NS pairs of doubles per request): one WAVE per
// request.  Lane l of slot i owns the flat pair 64*i + l of the request, so every store
// instruction writes 1 KB of consecutive, line-aligned output (requests are multiples of 128 B for
// the registered shapes) -- the pattern that reaches the write ceiling in tools/ubench3.hip -- and
// keeps the (at most SD) reference values it needs in registers.  A slot lies inside one table
// except where a table boundary cuts it; the weights are then selected per lane.
template <int SD, int NS> __global__ __launch_bounds__(256) void shared_points_wave_kernel(const SharedArgs a, const int nwr) {
    // nwr (1, 2 or 4) waves share a request, NS slots each
    constexpr int NTAB = 1 + SD;
    const int table = a.rows * a.npts;
    const int npairs = (NTAB * table) >> 1;  // per request; table is even (host-checked)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int part = wave % nwr;                         // which NS slots of the request
    const int team = (blockIdx.x * 4 + wave) / nwr;      // global index of the wave team
    const long long nteams = (long long)gridDim.x * 4 / nwr;
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    double rv[NS][2][SD];  // value table: rv[..][0]; gradient tables: the SD reference gradients
    int tsel[NS];          // table of this lane's pair in slot i
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int f0 = 64 * (NS * part + i);
        const int f = min(f0 + lane, npairs - 1);
        const int t = (2 * f) / table, q = 2 * f - t * table;
        tsel[i] = t;
#pragma unroll
        for (int el = 0; el < 2; ++el)
#pragma unroll
            for (int c = 0; c < SD; ++c) rv[i][el][c] = t == 0 ? (c == 0 ? a.ref[q + el] : 0.0) : a.ref[(size_t)(1 + c) * table + q + el];
    }
    __shared__ double sK[4][64][SD * SD];  // per wave: K of its team's next 64 requests
    for (long long base = (long long)team * 64; base < a.nreq; base += nteams * 64) {
        {
            const long long rq = min(base + lane, a.nreq - 1);
            double A[SD][SD], b[SD];
            cell_map<SD>(a.verts + (size_t)rq * (SD + 1) * SD, A, b);
#pragma unroll
            for (int c = 0; c < SD; ++c)
#pragma unroll
                for (int d = 0; d < SD; ++d) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < SD; ++k) t += a.A0inv[c * SD + k] * A[k][d];
                    sK[wave][lane][c * SD + d] = t;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        const int nblk = (int)min(64LL, a.nreq - base);
        for (int rb = 0; rb < nblk; ++rb) {
            double K[SD][SD];
#pragma unroll
            for (int c = 0; c < SD; ++c)
#pragma unroll
                for (int d = 0; d < SD; ++d) K[c][d] = sK[wave][rb][c * SD + d];
            v2d_t* o2 = reinterpret_cast<v2d_t*>(a.out + (size_t)(base + rb) * NTAB * table);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int f = 64 * (NS * part + i) + lane;
                // weights of this lane's table (value -> (1, 0, 0); gradient d -> column d of K),
                // selected per lane: a table boundary may cut the slot.  (A wave-uniform fast path
                // for the slots inside one table, with scalar branches per table, measured slower.)
                double w[SD];
#pragma unroll
                for (int c = 0; c < SD; ++c) {
                    double x = c == 0 ? 1.0 : 0.0;
#pragma unroll
                    for (int d = 0; d < SD; ++d) x = tsel[i] == 1 + d ? K[c][d] : x;
                    w[c] = x;
                }
                v2d_t v;
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int c = 0; c < SD; ++c) {
                    acc0 += w[c] * rv[i][0][c];
                    acc1 += w[c] * rv[i][1][c];
                }
                v.x = acc0;
                v.y = acc1;
                if (f < npairs) o2[f] = v;  // plain store: measured 3 % faster than non-temporal for this store-only kernel
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();  // sK is rewritten for the next 64 requests
    }
}

}  // namespace fxk
