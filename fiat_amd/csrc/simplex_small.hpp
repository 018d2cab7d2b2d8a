// Low-order simplex elements (degree <= 4 on triangles, <= 2 on tetrahedra; gfx950).
//
// For these shapes the MFMA path of the generic kernel spends most of its time around the matrix
// instruction (a 16x16x4 tile of which 3-10 rows are used, column decoding, the LDS round trip of
// the expansion values).  Here the contraction is lane-local: lane <-> (request, point), the
// recurrence is unrolled from the constexpr step table with every member in registers, and
//   table[t][row] = sum_k C[row][k] * Phi_k^(t)
// is rows*nexp*ntab FMAs per lane with the coefficients as scalar operands (P2 tet: 400).  The
// results go through a per-wave LDS image of the item's P whole requests, which is contiguous in
// HBM, and leave as 16-byte-per-lane full-line stores.
#pragma once
#include "aux_kernels.hpp"
#include "simplex_fixed.hpp"
#include "store.hpp"

namespace fxk {

constexpr int SMALL_MAXSTEPS = 20;  // (sd, n) = (2, 5): 21 members and (3, 3): 20 (values only; with derivatives up to (2, 4): 15 members)

struct SmallArgs {
    const double* pts;    // [nreq][npts][SD]
    const double* verts;  // [nreq][SD+1][SD] or nullptr
    double* out;          // [nreq][ntab][rows][npts]
    const double* cmat;   // [rows][nexp] coefficient matrix (C0 transform folded in), device
    double coef[3 * SMALL_MAXSTEPS];  // A, B, C of every step: by value = scalar loads
    double phi0;
    double A0[9];
    double b0[3];
    long long nreq, nitems;
    int npts, rows;
    int P;              // whole requests per wave item (P * npts <= 64)
    int stage_doubles;  // per-wave LDS doubles (>= P * ntab * rows * npts, even)
    int debug;
    int piola;          // PIOLA instances: 1 covariant, 2 contravariant map of the request's cell (rows = dofs x SD components)
    double G[9];        // A0 / 2 (piola_matrix, aux_kernels.hpp)
    int shared_pts;     // 1: `pts` is ONE point set [npts][SD] on the element's own cell, pushed forward to every request's cell
                        // (fx_tabulate_batch_shared's route for tiny requests): X from (A0, b0), derivatives through the request's cell
};

// PIOLA: vector-valued functions on per-request cells leave already pushed forward (phi = M Phi per dof, M from
// piola_matrix) -- the separate pass over the tables (read + write) costs twice the tabulation itself for these shapes.
template <int SD, int N, int ORDER, int NW, bool PIOLA = false>
__global__ __launch_bounds__(64 * NW) void tabulate_simplex_small(const SmallArgs a) {
    constexpr int NTAB = NTab<SD, ORDER>::value;
    constexpr StepTable<SD, N> TBL{};
    constexpr int NEXP = StepTable<SD, N>::NEXP;
    static_assert(NEXP - 1 <= SMALL_MAXSTEPS, "step table too long for SmallArgs");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double* stage = lds + (size_t)wave * a.stage_doubles;
    const int npts = a.npts, rows = a.rows;
    const int table = rows * npts;
    const long long reqsize = (long long)NTAB * table;
    const float rinv = 1.0f / (float)npts;
    const int rl = idiv_small(lane, rinv);   // request of this lane inside the item
    const int pl = lane - rl * npts;
    typedef const __attribute__((address_space(4))) double CDouble;
    const __attribute__((address_space(4))) char* kargs =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    CDouble* kcoef = (CDouble*)(kargs + __builtin_offsetof(SmallArgs, coef));
    // the coefficient matrix is never written while the kernel runs: through the constant address space its rows are SCALAR
    // loads (as generic global loads the compiler, unable to rule out aliasing with the output stores, issued five vector
    // loads per row and waited for them row by row -- half of the launch for values-only P3 triangles, tools/small_ablation.py)
    CDouble* kcmat = (CDouble*)(unsigned long long)a.cmat;

    for (long long item = (long long)blockIdx.x * NW + wave; item < a.nitems; item += (long long)gridDim.x * NW) {
        const long long r0 = item * a.P;
        const long long left = a.nreq - r0;
        const int Pcur = left < a.P ? (int)left : a.P;
        const bool active = rl < Pcur;
        const long long req = r0 + (active ? rl : 0);

        // ---------------- points -> reference coordinates ----------------
        double X[SD];
        double J[SD][SD];
        {
            double x[SD];
            const double* pp = a.pts + ((size_t)(a.shared_pts ? 0 : req) * npts + (active ? pl : 0)) * SD;
#pragma unroll
            for (int d = 0; d < SD; ++d) x[d] = pp[d];
            double bb[SD];
            if (a.verts != nullptr) {
                cell_map<SD>(a.verts + (size_t)req * (SD + 1) * SD, J, bb);
            } else {
#pragma unroll
                for (int i = 0; i < SD; ++i) {
                    bb[i] = a.b0[i];
#pragma unroll
                    for (int d = 0; d < SD; ++d) J[i][d] = a.A0[i * SD + d];
                }
            }
            if (a.shared_pts) {  // (uniform branch) the point is on the element's cell: the same X for every request
#pragma unroll
                for (int i = 0; i < SD; ++i) {
                    double t = a.b0[i];
#pragma unroll
                    for (int d = 0; d < SD; ++d) t += a.A0[i * SD + d] * x[d];
                    X[i] = t;
                }
            } else {
#pragma unroll
                for (int i = 0; i < SD; ++i) {
                    double t = bb[i];
#pragma unroll
                    for (int d = 0; d < SD; ++d) t += J[i][d] * x[d];
                    X[i] = t;
                }
            }
        }

        // ---------------- recurrence, every member in registers ----------------
        Jet<SD, ORDER> mem[NEXP];
        Jet<SD, ORDER> zero;
        jet_zero(zero);
        jet_zero(mem[0]);
        mem[0].v = a.phi0;
        if (!FX_ABL(a, 1)) {
            Factors<SD, ORDER> F;
            int fcodim = -1;
#pragma unroll
            for (int s = 0; s < NEXP - 1; ++s) {
                if (TBL.codim[s] != fcodim) {
                    fcodim = TBL.codim[s];
                    make_factors<SD, ORDER>(F, fcodim, X, J);
                }
                apply_step<SD, ORDER>(mem[TBL.dst[s]], mem[TBL.cur[s]], TBL.prv[s] < 0 ? zero : mem[TBL.prv[s]], F,
                                      kcoef[3 * s], kcoef[3 * s + 1], kcoef[3 * s + 2]);
            }
        } else {
#pragma unroll
            for (int k = 1; k < NEXP; ++k) mem[k] = mem[0];
        }

        // ---------------- lane-local contraction -> LDS image of the item ----------------
        if constexpr (PIOLA) {
            double M[SD][SD];
            piola_matrix<SD>(a.verts + (size_t)req * (SD + 1) * SD, a.G, a.piola, M);
            double* sp = stage + (size_t)(active ? rl : 0) * reqsize + (active ? pl : 0);
            for (int dof = 0; dof < rows / SD; ++dof) {
                double acc[SD][NTAB];
#pragma unroll
                for (int c = 0; c < SD; ++c) {
                    CDouble* crow = kcmat + (size_t)(dof * SD + c) * NEXP;
#pragma unroll
                    for (int t = 0; t < NTAB; ++t) acc[c][t] = 0.0;
#pragma unroll
                    for (int k = 0; k < NEXP; ++k) {
                        const double cf = crow[k];
                        acc[c][0] += cf * mem[k].v;
                        if constexpr (ORDER >= 1) {
#pragma unroll
                            for (int d = 0; d < SD; ++d) acc[c][1 + d] += cf * mem[k].g[d];
                        }
                        if constexpr (ORDER >= 2) {
#pragma unroll
                            for (int h = 0; h < SD * (SD + 1) / 2; ++h) acc[c][1 + SD + h] += cf * mem[k].h[h];
                        }
                    }
                }
                if (active) {
#pragma unroll
                    for (int t = 0; t < NTAB; ++t)
#pragma unroll
                        for (int r = 0; r < SD; ++r) {
                            double y = 0.0;
#pragma unroll
                            for (int c = 0; c < SD; ++c) y += M[r][c] * acc[c][t];
                            sp[(size_t)t * table + (dof * SD + r) * npts] = y;
                        }
                }
            }
        } else if (!FX_ABL(a, 2)) {
            double* sp = stage + (size_t)(active ? rl : 0) * reqsize + (active ? pl : 0);
            for (int row = 0; row < rows; ++row) {
                CDouble* crow = kcmat + (size_t)row * NEXP;
                double acc[NTAB];
#pragma unroll
                for (int t = 0; t < NTAB; ++t) acc[t] = 0.0;
#pragma unroll
                for (int k = 0; k < NEXP; ++k) {
                    const double c = crow[k];
                    acc[0] += c * mem[k].v;
                    if constexpr (ORDER >= 1) {
#pragma unroll
                        for (int d = 0; d < SD; ++d) acc[1 + d] += c * mem[k].g[d];
                    }
                    if constexpr (ORDER >= 2) {
#pragma unroll
                        for (int h = 0; h < SD * (SD + 1) / 2; ++h) acc[1 + SD + h] += c * mem[k].h[h];
                    }
                }
                if (active) {
#pragma unroll
                    for (int t = 0; t < NTAB; ++t) sp[(size_t)t * table + row * npts] = acc[t];
                }
            }
        }
        wave_lds_fence();

        // ---------------- image -> HBM: P whole requests, contiguous ----------------
        if (!FX_ABL(a, 4)) {
            const long long total = (long long)Pcur * reqsize;
            double* gout = a.out + (size_t)r0 * reqsize;
            if ((reqsize & 1) == 0) {
                const v2d* s2 = reinterpret_cast<const v2d*>(stage);
                v2d* g2 = reinterpret_cast<v2d*>(gout);
                flush_block(g2, s2, (int)(total >> 1), lane);  // whole-line non-temporal body, plain partial edges (store.hpp)
            } else {
                for (long long i = lane; i < total; i += 64) gout[i] = stage[i];
            }
        }
        wave_lds_fence();  // the next item overwrites the image
    }
}

}  // namespace fxk
