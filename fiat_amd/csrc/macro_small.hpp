// Macro elements of low order (gfx950): lane-local tabulation on a split cell.
//
// On a simplicial complex every point has its own coefficient matrix -- the columns of the members of
// the sub-cell it lies in (FIAT/expansions.py:449-490) -- so the columns of an MFMA tile do not share
// an A operand; stacking the sub-cells along K (the MACRO instance of the generic kernel) multiplies the
// matrix work by the number of sub-cells.  Here lane <-> (request, point) as in simplex_small.hpp:
// the lane bins its point (expansions.py:771-811), runs the unrolled recurrence on the sub-cell's
// collapsed coordinates with every member in registers, and contracts with ITS sub-cell's coefficient
// block, read from an LDS copy of all blocks (per-lane addresses; lanes in the same sub-cell read the
// same words).  Points on interfaces take one more pass per further sub-cell and accumulate with weight
// 1/multiplicity (:469-477): the first pass covers every column and leaves through a per-wave LDS image,
// RC rows of all tables of the item's requests at a time (runs of RC*npts contiguous doubles per request
// and table), the rare further passes add their share to the stored tables directly.
#pragma once
#include "simplex_fixed.hpp"
#include "store.hpp"

namespace fxk {

constexpr int MACRO_SMALL_MAXSTEPS = 19;  // (sd, n) = (3, 3): 20 members

struct MacroSmallArgs {
    const double* pts;    // [nreq][npts][SD]
    const double* verts;  // [nreq][SD+1][SD] or nullptr
    double* out;          // [nreq][ntab][rows][npts]
    const double* cmat;   // [ncell][rows][nexp]: C[:, map[c]] T s_c (cell-node map, C0 transform, scale folded in)
    const double* cells;  // [parent: L(4x3) l(4)] then per sub-cell [M(3x3) m(3) L(4x3) l(4)] (simplex_kernel.hpp)
    double coef[3 * MACRO_SMALL_MAXSTEPS];
    double phi0;
    double A0[9];
    double b0[3];
    long long nreq, nitems;
    int npts, rows, ncell, unique;
    int P;              // whole requests per wave item (P * npts <= 64)
    int RC;             // rows per image round
    int Ls;             // image stride of one (request, table) run: >= RC * npts, even
    int vec2;           // every run starts and ends on a 16-byte boundary: 16-byte stores
    int stage_doubles;  // per-wave LDS doubles (>= P * ntab * Ls, even)
    int cmat_doubles;   // ncell * rows * nexp, rounded up to even
    int debug;          // measurement only (FIAT_AMD_DEBUG): 1 bin every point to sub-cell 0, 2 skip the contraction, 4 skip HBM stores
};

template <int SD, int N, int ORDER, int NW>
__global__ __launch_bounds__(64 * NW) void tabulate_macro_small(const MacroSmallArgs a) {
    constexpr int NTAB = NTab<SD, ORDER>::value;
    constexpr StepTable<SD, N> TBL{};
    constexpr int NEXP = StepTable<SD, N>::NEXP;
    static_assert(NEXP - 1 <= MACRO_SMALL_MAXSTEPS, "step table too long for MacroSmallArgs");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double* cm = lds;
    double* stage = lds + a.cmat_doubles + (size_t)wave * a.stage_doubles;
    for (int i = threadIdx.x; i < a.ncell * a.rows * NEXP; i += 64 * NW) cm[i] = a.cmat[i];
    __syncthreads();
    const int npts = a.npts, rows = a.rows;
    const int table = rows * npts;
    const long long reqsize = (long long)NTAB * table;
    const float rinv = 1.0f / (float)npts;
    const int rl = idiv_small(lane, rinv);
    const int pl = lane - rl * npts;
    typedef const __attribute__((address_space(4))) double CDouble;
    const __attribute__((address_space(4))) char* kargs =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    CDouble* kcoef = (CDouble*)(kargs + __builtin_offsetof(MacroSmallArgs, coef));
    CDouble* kcells = (CDouble*)(unsigned long long)a.cells;  // never written while the kernel runs

    for (long long item = (long long)blockIdx.x * NW + wave; item < a.nitems; item += (long long)gridDim.x * NW) {
        const long long r0 = item * a.P;
        const long long left = a.nreq - r0;
        const int Pcur = left < a.P ? (int)left : a.P;
        const bool active = rl < Pcur;
        const long long req = r0 + (active ? rl : 0);

        // ---------------- points -> (-1,1)^SD coordinates of the parent simplex ----------------
        double X[SD];
        double J[SD][SD];
        {
            double x[SD];
            const double* pp = a.pts + ((size_t)req * npts + (active ? pl : 0)) * SD;
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
#pragma unroll
            for (int i = 0; i < SD; ++i) {
                double t = bb[i];
#pragma unroll
                for (int d = 0; d < SD; ++d) t += J[i][d] * x[d];
                X[i] = t;
            }
        }

        // ---------------- bin the point: l1 distance in rescaled barycentric coordinates ----------------
        unsigned cellmask = 0;
        {
            // sum of the negative parts = 0.5 * |sum(|lambda| - lambda)| bit for bit; the tables are read through
            // the constant address space: uniform addresses become scalar loads
            auto dist = [&](CDouble* Lp) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i <= SD; ++i) {
                    double lam = Lp[12 + i];
#pragma unroll
                    for (int d = 0; d < SD; ++d) lam += Lp[i * 3 + d] * X[d];
                    s += fmax(-lam, 0.0);
                }
                return s;
            };
            const double tol = dist(kcells) + 1e-12;
            const int ncell_scan = FX_ABL(a, 1) ? 0 : a.ncell;
            for (int c = 0; c < ncell_scan; ++c)
                if (dist(kcells + 16 + c * 28 + 12) < tol) cellmask |= 1u << c;
            if (a.unique) cellmask &= ~cellmask + 1u;
            if (FX_ABL(a, 1)) cellmask = 1u;
            if (!active) cellmask = 0;
        }
        const int mult = __popc(cellmask);
        const double seed = mult > 1 ? a.phi0 / (double)mult : a.phi0;
        double* gout = a.out + (size_t)r0 * reqsize;
        const int RC = a.RC, Ls = a.Ls;
        const int runs = Pcur * NTAB;
        const int rla = active ? rl : 0, pla = active ? pl : 0;

        bool first = true;  // wave-uniform: the pass that covers every column
        while (__any(cellmask != 0)) {
            const bool wr = cellmask != 0;
            const int c = wr ? __ffs((int)cellmask) - 1 : 0;
            cellmask &= cellmask - 1u;
            CDouble* cd = kcells + 16 + c * 28;
            double Xc[SD];
            double Jc[SD][SD];
#pragma unroll
            for (int i = 0; i < SD; ++i) {
                double t = cd[9 + i];
#pragma unroll
                for (int d = 0; d < SD; ++d) t += cd[i * 3 + d] * X[d];
                Xc[i] = t;
#pragma unroll
                for (int d = 0; d < SD; ++d) {
                    double u = 0.0;
#pragma unroll
                    for (int e = 0; e < SD; ++e) u += cd[i * 3 + e] * J[e][d];
                    Jc[i][d] = u;
                }
            }

            // ---- recurrence on the sub-cell, every member in registers ----
            Jet<SD, ORDER> mem[NEXP];
            Jet<SD, ORDER> zero;
            jet_zero(zero);
            jet_zero(mem[0]);
            mem[0].v = seed;
            {
                Factors<SD, ORDER> F;
                int fcodim = -1;
#pragma unroll
                for (int s = 0; s < NEXP - 1; ++s) {
                    if (TBL.codim[s] != fcodim) {
                        fcodim = TBL.codim[s];
                        make_factors<SD, ORDER>(F, fcodim, Xc, Jc);
                    }
                    apply_step<SD, ORDER>(mem[TBL.dst[s]], mem[TBL.cur[s]], TBL.prv[s] < 0 ? zero : mem[TBL.prv[s]], F,
                                          kcoef[3 * s], kcoef[3 * s + 1], kcoef[3 * s + 2]);
                }
            }

            // ---- lane-local contraction with the sub-cell's coefficient block ----
            const double* cblock = cm + (size_t)c * rows * NEXP;
            auto contract = [&](int row, double* acc) {
                const double* crow = cblock + row * NEXP;
#pragma unroll
                for (int t = 0; t < NTAB; ++t) acc[t] = 0.0;
#pragma unroll
                for (int k = 0; k < NEXP; ++k) {
                    const double cf = crow[k];
                    acc[0] += cf * mem[k].v;
                    if constexpr (ORDER >= 1) {
#pragma unroll
                        for (int d = 0; d < SD; ++d) acc[1 + d] += cf * mem[k].g[d];
                    }
                    if constexpr (ORDER >= 2) {
#pragma unroll
                        for (int h = 0; h < SD * (SD + 1) / 2; ++h) acc[1 + SD + h] += cf * mem[k].h[h];
                    }
                }
            };
            if (first) {
                double* sp = stage + (size_t)rla * NTAB * Ls + pla;
                for (int row0 = 0; row0 < rows; row0 += RC) {
                    const int rc = min(RC, rows - row0);
                    const int rc_run = FX_ABL(a, 2) ? 0 : rc;
                    for (int r = 0; r < rc_run; ++r) {
                        double acc[NTAB];
                        contract(row0 + r, acc);
                        if (active) {  // a point in no sub-cell keeps a zero column (as in the reference)
#pragma unroll
                            for (int t = 0; t < NTAB; ++t) sp[t * Ls + r * npts] = wr ? acc[t] : 0.0;
                        }
                    }
                    wave_lds_fence();
                    // image -> HBM: one run of rc*npts doubles per (request, table)
                    const int L = rc * npts;
                    double* gchunk = gout + (size_t)row0 * npts;
                    if (FX_ABL(a, 4)) {
                    } else if (a.vec2 && rc == rows && Ls == table) {
                        // one round holds the item's whole requests: the image is contiguous in HBM
                        const int total = (Pcur * NTAB * table) >> 1;
                        const v2d* s2 = reinterpret_cast<const v2d*>(stage);
                        v2d* g2 = reinterpret_cast<v2d*>(gout);
                        flush_block(g2, s2, total, lane);  // whole-line non-temporal body, plain partial edges (store.hpp)
                    } else if (a.vec2) {
                        const int hp = L >> 1;
                        const int total = runs * hp;
                        const float rinv_hp = 1.0f / (float)hp;
#pragma unroll 4
                        for (int j = lane; j < total; j += 64) {
                            const int run = idiv_small(j, rinv_hp);
                            const int q = j - run * hp;
                            const int p = run / NTAB, t = run - p * NTAB;
                            const v2d v = *reinterpret_cast<const v2d*>(stage + (size_t)run * Ls + 2 * q);
                            // (plain stores, hoping for L2 to merge the edge lines of neighbouring rounds, measured
                            // equal or slower: 352 us stores-only either way for P3 on Alfeld tetrahedra)
                            stream_store(reinterpret_cast<v2d*>(gchunk + (size_t)p * reqsize + (size_t)t * table + 2 * q), v);
                        }
                    } else {
                        const int total = runs * L;
                        const float rinv_L = 1.0f / (float)L;
                        for (int j = lane; j < total; j += 64) {
                            const int run = idiv_small(j, rinv_L);
                            const int q = j - run * L;
                            const int p = run / NTAB, t = run - p * NTAB;
                            gchunk[(size_t)p * reqsize + (size_t)t * table + q] = stage[(size_t)run * Ls + q];
                        }
                    }
                    wave_lds_fence();  // the next round overwrites the image
                }
                first = false;
                if (__any(cellmask != 0)) __threadfence();  // the further passes read what this one stored
            } else {
                // points on interfaces: add this sub-cell's share (weight 1/multiplicity is in the seed)
                double* gp = gout + (size_t)rla * reqsize + pla;
                for (int row = 0; row < rows; ++row) {
                    double acc[NTAB];
                    contract(row, acc);
                    if (wr) {
#pragma unroll
                        for (int t = 0; t < NTAB; ++t) gp[(size_t)t * table + (size_t)row * npts] += acc[t];
                    }
                }
            }
        }
        if (first && active) {  // no point of this item lies in any sub-cell: zero tables
            double* gp = gout + (size_t)rl * reqsize + pl;
            for (int i = 0; i < NTAB * rows; ++i) gp[(size_t)i * npts] = 0.0;
        }
    }
}

}  // namespace fxk
