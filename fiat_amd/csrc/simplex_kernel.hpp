// Batched simplex tabulation kernel for gfx950 (MI355X).
//
// One wavefront owns one work item at a time (one packed group of requests, or
// one point-chunk of a request) and runs three phases on it with no
// workgroup-level synchronisation:
//
//  1. recurrence  lanes <-> points.  Every lane walks the step table of the
//     expansion set (plan.hpp) for its own point and stores each finished
//     member (value, gradient, Hessian components) into the wave's LDS tile,
//     already laid out as v_mfma_f64_16x16x4_f64 B-operand fragments.
//     Reference: FIAT/expansions.py:140-267 (+ :54-63, :66-137).
//  2. contraction tables = coeffs x expansion-values (polynomial_set.py:71)
//     as 16x16x4 f64 MFMAs; the K order is the member index, A fragments are
//     packed on the host (C0_basis of the "bubble" variant folded in).
//  3. store       D tiles are scattered into an LDS image of the item's output
//     block and streamed to HBM with 16-byte-per-lane coalesced stores (or,
//     for point-chunked items whose output is not contiguous, written directly).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "store.hpp"

namespace fxk {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

struct Step {  // must match fx::Step
    int dst, cur, prv, codim;
    double A, B, C;
};

struct TabArgs {
    const double* pts;    // [nreq][npts][SD]
    const double* verts;  // [nreq][SD+1][SD] or nullptr
    double* out;          // [nreq][ntab][rows][npts]
    const double* afrag;  // [MT][KS][64]
    const Step* steps;
    double phi0;
    double A0[9];
    double b0[3];
    long long nreq;
    long long nitems;
    int npts, rows, nexp, nsteps, KS, MT, ntab;
    int P;       // requests per item (whole requests, pc == npts)
    int pc;      // points per item and request
    int nchunk;  // point chunks per request (1 unless chunked)
    int phi_doubles;    // per-wave LDS doubles for the Phi fragments
    int stage_doubles;  // per-wave LDS doubles for the output image (0: direct stores)
    int debug;          // measurement builds only: 1 skip recurrence, 2 skip MFMA, 4 skip HBM stores
    // MACRO instances (cells of a simplicial complex, FIAT/expansions.py:449-490): the members of
    // sub-cell c occupy the K rows [c*nexp, (c+1)*nexp) of the Phi tile (cell-node map, C0 transform
    // and per-cell scale are folded into the A fragments on the host).
    // cells: [parent: L(4x3) l(4)] then per sub-cell [M(3x3) m(3) L(4x3) l(4)], all in the (-1,1)^SD
    // coordinates of the parent: X_c = M xi + m, rescaled barycentric coordinates lambda = L xi + l
    const double* cells;
    int ncell;
    int unique;         // bin every point to its first cell only (C0 sets at order 0, expansions.py:452)
};

template <int SD> struct Dims {
    static constexpr int NG = SD;
    static constexpr int NH = SD * (SD + 1) / 2;
};

template <int SD, int ORDER> struct Jet {
    double v;
    double g[ORDER >= 1 ? SD : 1];
    double h[ORDER >= 2 ? SD * (SD + 1) / 2 : 1];
};

template <int SD, int ORDER> __device__ __forceinline__ void jet_zero(Jet<SD, ORDER>& j) {
    j.v = 0.0;
    if constexpr (ORDER >= 1) {
#pragma unroll
        for (int d = 0; d < SD; ++d) j.g[d] = 0.0;
    }
    if constexpr (ORDER >= 2) {
#pragma unroll
        for (int h = 0; h < SD * (SD + 1) / 2; ++h) j.h[h] = 0.0;
    }
}

// collapsed-coordinate factors of one codimension and their derivatives
// (expansions.py:54-63 jacobi_factors, :205 ddfc)
template <int SD, int ORDER> struct Factors {
    double fa, fb, fc;
    double dfa[ORDER >= 1 ? SD : 1], dfb[ORDER >= 1 ? SD : 1], dfc[ORDER >= 1 ? SD : 1];
    double ddfc[ORDER >= 2 ? SD * (SD + 1) / 2 : 1];
};

template <int SD, int ORDER, int CODIM>
__device__ __forceinline__ void make_factors_c(Factors<SD, ORDER>& F, const double* X,
                                               const double (*J)[SD]) {
    // coordinates padded with -1, Jacobian rows padded with 0 (expansions.py:43-51)
    const double x = X[CODIM];
    double y = -1.0, z = -1.0;
    if constexpr (CODIM + 1 < SD) y = X[CODIM + 1];
    if constexpr (CODIM + 2 < SD) z = X[CODIM + 2];
    F.fb = 0.5 * (y + z);
    F.fa = x + (F.fb + 1.0);
    F.fc = F.fb * F.fb;
    if constexpr (ORDER >= 1) {
#pragma unroll
        for (int d = 0; d < SD; ++d) {
            const double dx = J[CODIM][d];
            double dy = 0.0, dz = 0.0;
            if constexpr (CODIM + 1 < SD) dy = J[CODIM + 1][d];
            if constexpr (CODIM + 2 < SD) dz = J[CODIM + 2][d];
            F.dfb[d] = 0.5 * (dy + dz);
            F.dfa[d] = dx + F.dfb[d];
            F.dfc[d] = 2.0 * F.fb * F.dfb[d];
        }
    }
    if constexpr (ORDER >= 2) {
        int h = 0;
#pragma unroll
        for (int d1 = 0; d1 < SD; ++d1)
#pragma unroll
            for (int d2 = d1; d2 < SD; ++d2) F.ddfc[h++] = 2.0 * F.dfb[d1] * F.dfb[d2];
    }
}

template <int SD, int ORDER>
__device__ __forceinline__ void make_factors(Factors<SD, ORDER>& F, int codim, const double* X,
                                             const double (*J)[SD]) {
    if (codim == 0) {
        make_factors_c<SD, ORDER, 0>(F, X, J);
    } else if (codim == 1) {
        if constexpr (SD > 1) make_factors_c<SD, ORDER, 1>(F, X, J);
    } else {
        if constexpr (SD > 2) make_factors_c<SD, ORDER, 2>(F, X, J);
    }
}

// floor(c / d) for 0 <= c < 2^20, rinv = 1.0f / d
__device__ __forceinline__ int idiv_small(int c, float rinv) {
    return (int)(((float)c + 0.5f) * rinv);
}

// one three-term step with derivatives by the Leibniz rule (expansions.py:66-137):
//   nw = (A fa - B fb) cur - C fc prv
template <int SD, int ORDER>
__device__ __forceinline__ void apply_step(Jet<SD, ORDER>& nw, const Jet<SD, ORDER>& cur,
                                           const Jet<SD, ORDER>& prv, const Factors<SD, ORDER>& F,
                                           double A, double B, double C) {
    const double f = A * F.fa - B * F.fb;
    const double g = -C * F.fc;
    nw.v = cur.v * f + prv.v * g;
    if constexpr (ORDER >= 1) {
        double df[SD], dg[SD];
#pragma unroll
        for (int d = 0; d < SD; ++d) {
            df[d] = A * F.dfa[d] - B * F.dfb[d];
            dg[d] = -C * F.dfc[d];
            nw.g[d] = cur.g[d] * f + cur.v * df[d] + prv.g[d] * g + prv.v * dg[d];
        }
        if constexpr (ORDER >= 2) {
            int h = 0;
#pragma unroll
            for (int d1 = 0; d1 < SD; ++d1)
#pragma unroll
                for (int d2 = d1; d2 < SD; ++d2) {
                    double t = cur.h[h] * f + df[d1] * cur.g[d2] + df[d2] * cur.g[d1];
                    t += prv.h[h] * g + dg[d1] * prv.g[d2] + dg[d2] * prv.g[d1];
                    t += (-C * F.ddfc[h]) * prv.v;
                    nw.h[h] = t;
                    ++h;
                }
        }
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave complete in order; this only stops the compiler from
    // moving LDS accesses of different lanes across a phase boundary.  The fences are
    // restricted to the LDS address space ("local"): an unrestricted release fence makes
    // hipcc wait vmcnt(0), i.e. for every outstanding HBM store of the wave -- measured
    // to serialise the store phase with the next request's compute.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// affine map of the request's cell onto the default (-1,1)^SD simplex:
// X = A x + b, A = 2 E^{-1}, b = -1 - A v0 with E = [v1-v0, ...] as columns
// (reference_element.py:1621-1654 make_affine_mapping, solved in closed form).
template <int SD>
__device__ __forceinline__ void cell_map(const double* __restrict__ v, double (*A)[SD], double* b) {
    if constexpr (SD == 1) {
        double inv = 2.0 / (v[1] - v[0]);
        A[0][0] = inv;
        b[0] = -1.0 - inv * v[0];
    } else if constexpr (SD == 2) {
        double e00 = v[2] - v[0], e10 = v[3] - v[1];  // column 0 = v1 - v0
        double e01 = v[4] - v[0], e11 = v[5] - v[1];  // column 1 = v2 - v0
        double inv = 2.0 / (e00 * e11 - e01 * e10);
        A[0][0] = e11 * inv;
        A[0][1] = -e01 * inv;
        A[1][0] = -e10 * inv;
        A[1][1] = e00 * inv;
#pragma unroll
        for (int i = 0; i < 2; ++i) b[i] = -1.0 - (A[i][0] * v[0] + A[i][1] * v[1]);
    } else {
        double e[3][3];  // e[r][c] = (v_{c+1} - v_0)[r]
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) e[r][c] = v[3 * (c + 1) + r] - v[r];
        double c00 = e[1][1] * e[2][2] - e[1][2] * e[2][1];
        double c01 = e[1][2] * e[2][0] - e[1][0] * e[2][2];
        double c02 = e[1][0] * e[2][1] - e[1][1] * e[2][0];
        double inv = 2.0 / (e[0][0] * c00 + e[0][1] * c01 + e[0][2] * c02);
        A[0][0] = c00 * inv;
        A[0][1] = (e[0][2] * e[2][1] - e[0][1] * e[2][2]) * inv;
        A[0][2] = (e[0][1] * e[1][2] - e[0][2] * e[1][1]) * inv;
        A[1][0] = c01 * inv;
        A[1][1] = (e[0][0] * e[2][2] - e[0][2] * e[2][0]) * inv;
        A[1][2] = (e[0][2] * e[1][0] - e[0][0] * e[1][2]) * inv;
        A[2][0] = c02 * inv;
        A[2][1] = (e[0][1] * e[2][0] - e[0][0] * e[2][1]) * inv;
        A[2][2] = (e[0][0] * e[1][1] - e[0][1] * e[1][0]) * inv;
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] = -1.0 - (A[i][0] * v[0] + A[i][1] * v[1] + A[i][2] * v[2]);
    }
}

template <int SD, int ORDER> struct NTab {
    static constexpr int value = (ORDER == 0) ? 1 : (ORDER == 1) ? 1 + SD : 1 + SD + SD * (SD + 1) / 2;
};

// KS_T/MT_T > 0: compile-time fragment counts, A fragments live in registers.
// MACRO: the element's cell is a complex; every point is binned to its sub-cell(s) and the recurrence
// runs on the sub-cell's own collapsed coordinates.
template <int SD, int ORDER, int NW, int KS_T, int MT_T, bool MACRO = false>
__global__ __launch_bounds__(64 * NW) void tabulate_simplex_kernel(const TabArgs a) {
    constexpr int NTAB = NTab<SD, ORDER>::value;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int KS = KS_T > 0 ? KS_T : a.KS;
    const int MT = MT_T > 0 ? MT_T : a.MT;
    typedef const __attribute__((address_space(4))) Step CStep;
    typedef const __attribute__((address_space(4))) double CDbl;
    CStep* ksteps = (CStep*)(unsigned long long)a.steps;
    CDbl* kcells = (CDbl*)(unsigned long long)a.cells;
    (void)kcells;

    // ---- LDS carve-up: per wave [Phi fragments | output image]
    double* phi = lds + (size_t)wave * (a.phi_doubles + a.stage_doubles);
    double* stage = phi + a.phi_doubles;
    for (int i = lane; i < a.phi_doubles; i += 64) phi[i] = 0.0;  // K padding must read as zero
    wave_lds_fence();

    // A fragments in registers for the compile-time shapes
    double areg[(KS_T > 0 ? KS_T : 1) * (MT_T > 0 ? MT_T : 1)];
    if constexpr (KS_T > 0) {
#pragma unroll
        for (int i = 0; i < KS_T * MT_T; ++i) areg[i] = a.afrag[i * 64 + lane];
    }

    const int pc = a.pc;
    const float rinv_pc = 1.0f / (float)pc;
    const float rinv_req = 1.0f / (float)(NTAB * pc);
    const int npts = a.npts;
    const int rows = a.rows;
    const long long table = (long long)rows * npts;       // doubles per table
    const long long reqsize = (long long)NTAB * table;    // doubles per request

    for (long long item = (long long)blockIdx.x * NW + wave; item < a.nitems;
         item += (long long)gridDim.x * NW) {
        long long r0;
        int p0, pcur, Pcur;
        if (a.nchunk > 1) {
            r0 = item / a.nchunk;
            int ch = (int)(item - r0 * a.nchunk);
            p0 = ch * pc;
            pcur = min(pc, npts - p0);
            Pcur = 1;
        } else {
            r0 = item * a.P;
            p0 = 0;
            pcur = pc;
            long long left = a.nreq - r0;
            Pcur = left < a.P ? (int)left : a.P;
        }
        const int Q = Pcur * pcur;
        const bool active = lane < Q;
        const int lane_c = active ? lane : 0;
        const int rl = idiv_small(lane_c, 1.0f / (float)pcur);
        const int pl = lane_c - rl * pcur;
        const long long req = r0 + rl;
        if constexpr (MACRO) {  // rows of the cells a point is not in must read as zero
            for (int i = lane; i < a.phi_doubles; i += 64) phi[i] = 0.0;
            wave_lds_fence();
        }

        // ---------------- phase 1: points -> reference coordinates ----------------
        double X[SD];
        double J[SD][SD];
        {
            double x[SD];
            const double* pp = a.pts + ((size_t)req * npts + p0 + pl) * SD;
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

        // ---------------- MACRO: bin the point (expansions.py:771-811) ----------------
        // l1 distance = sum of the negative parts of the rescaled barycentric coordinates
        // (reference_element.py:778-780); a point belongs to every sub-cell within 1e-12 of its
        // distance to the parent simplex
        unsigned cellmask = 0;
        double seed = a.phi0;
        if constexpr (MACRO) {
            auto dist = [&](CDbl* Lp) {   // = 0.5 |sum(|lambda| - lambda)| bit for bit
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
            for (int c = 0; c < a.ncell; ++c)
                if (dist(kcells + 16 + c * 28 + 12) < tol) cellmask |= 1u << c;
            if (a.unique) cellmask &= ~cellmask + 1u;
            if (!active) cellmask = 0;
            const int mult = __popc(cellmask);
            if (mult > 1) seed = a.phi0 / (double)mult;  // non-unique binning: average (expansions.py:469-477)
        }
        int kb = 0;          // first K row of the current sub-cell
        bool wr = active;    // this lane writes expansion values in the current pass

        // LDS slot of (member k, column c): ((c>>4)*KS + (k>>2))*64 + (k&3)*16 + (c&15)
        int colbase[NTAB];
#pragma unroll
        for (int t = 0; t < NTAB; ++t) {
            int c = (rl * NTAB + t) * pc + pl;
            colbase[t] = (c >> 4) * KS * 64 + (c & 15);
        }
        auto put = [&](int k, const Jet<SD, ORDER>& j) {
            if (!wr) return;
            if constexpr (MACRO) k += kb;
            const int kofs = (k >> 2) * 64 + (k & 3) * 16;
            phi[colbase[0] + kofs] = j.v;
            if constexpr (ORDER >= 1) {
#pragma unroll
                for (int d = 0; d < SD; ++d) phi[colbase[1 + d] + kofs] = j.g[d];
            }
            if constexpr (ORDER >= 2) {
#pragma unroll
                for (int h = 0; h < SD * (SD + 1) / 2; ++h) phi[colbase[1 + SD + h] + kofs] = j.h[h];
            }
        };
        auto get = [&](int k, Jet<SD, ORDER>& j) {
            if constexpr (MACRO) k += kb;
            const int kofs = (k >> 2) * 64 + (k & 3) * 16;
            j.v = phi[colbase[0] + kofs];
            if constexpr (ORDER >= 1) {
#pragma unroll
                for (int d = 0; d < SD; ++d) j.g[d] = phi[colbase[1 + d] + kofs];
            }
            if constexpr (ORDER >= 2) {
#pragma unroll
                for (int h = 0; h < SD * (SD + 1) / 2; ++h) j.h[h] = phi[colbase[1 + SD + h] + kofs];
            }
        };

        // ---------------- phase 1: recurrence ----------------
        auto recurrence = [&](const double* X, const double (*J)[SD]) {
            Jet<SD, ORDER> cur, prv, nw;
            jet_zero(cur);
            jet_zero(prv);
            cur.v = seed;
            put(0, cur);
            Factors<SD, ORDER> F;
            int fcodim = -1;
            int last_dst = 0;
            const int nsteps = FX_ABL(a, 1) ? 0 : a.nsteps;
            for (int s = 0; s < nsteps; ++s) {
                // the step table is never written while the kernel runs: read through the constant address space,
                // so that the (uniform) loads are scalar -- next to the output stores the compiler cannot prove a
                // plain global load invariant and falls back to vector loads
                Step st;
                st.dst = ksteps[s].dst;
                st.cur = ksteps[s].cur;
                st.prv = ksteps[s].prv;
                st.codim = ksteps[s].codim;
                st.A = ksteps[s].A;
                st.B = ksteps[s].B;
                st.C = ksteps[s].C;
                if (st.codim != fcodim) {
                    fcodim = st.codim;
                    make_factors<SD, ORDER>(F, fcodim, X, J);
                }
                if (st.prv < 0) {
                    // chain start: the seed was produced earlier
                    if (st.cur != last_dst) {
                        wave_lds_fence();
                        get(st.cur, cur);
                    }
                    jet_zero(prv);
                }
                apply_step<SD, ORDER>(nw, cur, prv, F, st.A, st.B, st.C);
                put(st.dst, nw);
                prv = cur;
                cur = nw;
                last_dst = st.dst;
            }
        };
        if constexpr (!MACRO) {
            recurrence(X, J);
        } else {
            // one pass per sub-cell a lane's point lies in (one pass unless points sit on interfaces)
            while (__any(cellmask != 0)) {
                wr = cellmask != 0;
                const int c = wr ? __ffs((int)cellmask) - 1 : 0;
                cellmask &= cellmask - 1u;
                kb = c * a.nexp;
                CDbl* cd = kcells + 16 + c * 28;
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
                recurrence(Xc, Jc);
                wave_lds_fence();
            }
        }
        wave_lds_fence();

        // ---------------- phase 2 + 3: contraction and store ----------------
        const int ncols = Pcur * NTAB * pc;
        const int NT = (ncols + 15) >> 4;
        const bool staged = a.stage_doubles > 0;
        double* gout = a.out + (size_t)r0 * reqsize;
        const int NTrun = FX_ABL(a, 2) ? 0 : NT;
        for (int nt = 0; nt < NTrun; ++nt) {
            // decode this lane's output column
            const int c = (nt << 4) + (lane & 15);
            const int cr = idiv_small(c, rinv_req);
            const int crem = c - cr * (NTAB * pc);
            const int ct = idiv_small(crem, rinv_pc);
            const int cp = crem - ct * pc;
            const bool cvalid = (c < ncols) && (cp < pcur);
            // offset of (request cr, table ct, row 0, point p0+cp) relative to request r0
            const long long cofs = (long long)cr * reqsize + (long long)ct * table + p0 + cp;
            const double* bptr = phi + (size_t)nt * KS * 64 + lane;

            auto emit = [&](int mt, const v4d& acc) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = (mt << 4) + (lane >> 4) + 4 * j;
                    if (cvalid && m < rows) {
                        const long long o = cofs + (long long)m * npts;
                        if (staged)
                            stage[o] = acc[j];
                        else
                            gout[o] = acc[j];
                    }
                }
            };

            if constexpr (KS_T > 0) {
                double breg[KS_T];
#pragma unroll
                for (int ks = 0; ks < KS_T; ++ks) breg[ks] = bptr[ks * 64];
#pragma unroll
                for (int mt = 0; mt < MT_T; ++mt) {
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < KS_T; ++ks)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[mt * KS_T + ks], breg[ks], acc, 0, 0, 0);
                    emit(mt, acc);
                }
            } else {
                const double* ap = a.afrag + lane;  // 512 B per fragment, L1/L2 resident
                int mt = 0;
                for (; mt + 1 < MT; mt += 2) {  // two independent accumulator chains
                    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                    const double* a0 = ap + (size_t)mt * KS * 64;
                    const double* a1 = a0 + (size_t)KS * 64;
                    for (int ks = 0; ks < KS; ++ks) {
                        const double b = bptr[ks * 64];
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ks * 64], b, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ks * 64], b, acc1, 0, 0, 0);
                    }
                    emit(mt, acc0);
                    emit(mt + 1, acc1);
                }
                if (mt < MT) {
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
                    const double* a0 = ap + (size_t)mt * KS * 64;
                    for (int ks = 0; ks < KS; ++ks)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ks * 64], bptr[ks * 64], acc, 0, 0, 0);
                    emit(mt, acc);
                }
            }
        }

        if (staged && !FX_ABL(a, 4)) {
            wave_lds_fence();
            const long long total = (long long)Pcur * reqsize;  // doubles, contiguous in HBM
            if ((reqsize & 1) == 0) {
                const v2d* s2 = reinterpret_cast<const v2d*>(stage);
                v2d* g2 = reinterpret_cast<v2d*>(gout);
                flush_block(g2, s2, (int)(total >> 1), lane);  // whole-line non-temporal body, plain partial edges (store.hpp)
            } else {
                for (long long i = lane; i < total; i += 64) gout[i] = stage[i];
            }
        }
        wave_lds_fence();  // the next item's recurrence overwrites Phi / stage
    }
}

}  // namespace fxk
