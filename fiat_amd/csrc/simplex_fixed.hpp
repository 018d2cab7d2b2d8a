// Shape-specialised simplex tabulation kernel (gfx950).
//
// Same three phases as simplex_kernel.hpp, with the element shape known at
// compile time <SD, N, ORDER, ROWS> and one whole request per wave iteration:
//  * the recurrence is fully unrolled from a constexpr step table (member
//    indices, codimensions); only the three coefficients per step are loaded,
//    from constant offsets of a device table (scalar loads the compiler batches);
//    every LDS store uses an immediate offset;
//  * all B fragments of the request are read into registers before the first
//    MFMA, so the output image is built in the SAME LDS bytes the Phi tile
//    occupied: LDS per wave = max(Phi, image) instead of the sum;
//  * rows beyond the last full 16-row tile go through v_mfma_f64_4x4x4_4b
//    (four 4x4x4 blocks sharing the 16x16x4 B-operand layout: lane = 16*k + col,
//    D lane = 16*row + col -- probed on MI355X, tools/probe_mfma.hip), so a
//    20-row coefficient matrix costs 64 + 16 MFMA cycles per K-step, not 128;
//  * per-lane column offsets of the image are computed once per kernel.
#pragma once
#include "simplex_kernel.hpp"

// build-time switches of the LDS-image kernel (A/B measured)
#ifndef FX_PACK
#define FX_PACK 0          // 1: pack two tables per LDS store with v_permlane32_swap
#endif
#ifndef FX_UNROLL_COPY
#define FX_UNROLL_COPY 0   // 1: issue all image reads before the first HBM store
#endif

namespace fxk {

constexpr int cx_binom(int n, int k) {
    if (k < 0 || k > n) return 0;
    long long r = 1;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return (int)r;
}

constexpr int cx_member_index(int sd, int p, int q, int r) {
    if (sd == 1) return p;
    if (sd == 2) return (p + q) * (p + q + 1) / 2 + q;
    int t = p + q + r, u = q + r;
    return t * (t + 1) * (t + 2) / 6 + u * (u + 1) / 2 + r;
}

// Index part of the recurrence program; must enumerate exactly as
// fx::build_program (plan.hpp) so that coefficient i belongs to step i.
template <int SD, int N> struct StepTable {
    static constexpr int NEXP = cx_binom(N + SD, SD);
    static constexpr int NSTEPS = NEXP > 1 ? NEXP - 1 : 1;
    int dst[NSTEPS] = {}, cur[NSTEPS] = {}, prv[NSTEPS] = {}, codim[NSTEPS] = {};
    int count = 0;
    constexpr void chain(int cd, int p, int q) {
        int s = (cd >= 1 ? p : 0) + (cd >= 2 ? q : 0);
        int len = N - s;
        int id_prev = 0, id_cur = 0;
        for (int i = 0; i <= len; ++i) {
            int a = cd == 0 ? i : p, b = cd == 0 ? 0 : (cd == 1 ? i : q), c = cd == 2 ? i : 0;
            int id = cx_member_index(SD, a, b, c);
            if (i >= 1) {
                dst[count] = id;
                cur[count] = id_cur;
                prv[count] = i >= 2 ? id_prev : -1;
                codim[count] = cd;
                ++count;
            }
            id_prev = id_cur;
            id_cur = id;
        }
    }
    constexpr StepTable() {
        for (int cd = 0; cd < SD; ++cd) {
            if (cd == 0) {
                chain(0, 0, 0);
            } else if (cd == 1) {
                for (int p = 0; p < N; ++p) chain(1, p, 0);
            } else {
                for (int last = 0; last < N; ++last)
                    for (int first = 0; first < N - last; ++first) chain(2, first, last);
            }
        }
    }
};

// NC = 3 * number of steps.  The coefficients travel BY VALUE in the kernel
// argument segment: that is constant address space, so every use is a scalar load
// with a compile-time offset (a pointer into global memory makes hipcc fall back to
// vector loads + vmcnt(0) stalls inside the recurrence, measured 2x slower).
// Two doubles that are meaningful in the lower 32 lanes -> one double whose lower
// half carries `a` and whose upper half carries b's lower half (v_permlane32_swap).
// Lets one LDS store instruction carry two table components of <= 32 points.
__device__ __forceinline__ double pack_halves(double a, double b) {
    unsigned a0 = (unsigned)__double2loint(a), a1 = (unsigned)__double2hiint(a);
    unsigned b0 = (unsigned)__double2loint(b), b1 = (unsigned)__double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    return __hiloint2double((int)r1[0], (int)r0[0]);
}

template <int NC> struct FixedArgs {
    const double* pts;    // [nreq][npts][SD]
    const double* verts;  // [nreq][SD+1][SD] or nullptr
    double* out;          // [nreq][NTAB][ROWS][npts]
    const double* afrag;  // 16x16x4 fragments [MT16][KS][64] then 4x4x4 fragments [M4][KS][64]
    double coef[NC > 0 ? NC : 1];  // [nsteps][3] = A, B, C
    // uniform-cell case (verts == nullptr): factor derivatives do not depend on the
    // point, the host folds them with the step coefficients:
    //   ucoef[s][0..2]  = d/dx_d (A fa - B fb)          = A dfa[d] - B dfb[d]
    //   ucoef[s][3..5]  = K_d with d/dx_d (-C fc) = K_d fb,  K_d = -2 C dfb[d]
    //   ucoef[s][6..11] = -2 C dfb[d1] dfb[d2]  (d1 <= d2), the Hessian of -C fc
    double ucoef[NC > 0 ? 4 * NC : 1];
    double phi0;
    double A0[9];
    double b0[3];
    long long nreq;
    int npts;
    int lds_doubles;  // per-wave LDS doubles (>= Phi fragments and >= output image)
    int debug;
};

// number of full 16-row tiles / trailing 4-row blocks of a ROWS-row matrix
constexpr int rows_full16(int rows) { return (rows % 16 != 0 && rows % 16 <= 12) ? rows / 16 : (rows + 15) / 16; }
constexpr int rows_blk4(int rows) { return (rows % 16 != 0 && rows % 16 <= 12) ? (rows % 16 + 3) / 4 : 0; }

// uniform-cell step: same algebra as apply_step with the point-independent factor
// derivatives taken from scalar registers (u = ucoef row of the step)
template <int SD, int ORDER>
__device__ __forceinline__ void apply_step_uniform(Jet<SD, ORDER>& nw, const Jet<SD, ORDER>& cur,
                                                   const Jet<SD, ORDER>& prv, double fa, double fb, double fc,
                                                   double A, double B, double C,
                                                   const __attribute__((address_space(4))) double* u) {
    const double f = A * fa - B * fb;
    const double g = -C * fc;
    nw.v = cur.v * f + prv.v * g;
    if constexpr (ORDER >= 1) {
        double dg[SD];
#pragma unroll
        for (int d = 0; d < SD; ++d) {
            dg[d] = u[3 + d] * fb;
            nw.g[d] = cur.g[d] * f + cur.v * u[d] + prv.g[d] * g + prv.v * dg[d];
        }
        if constexpr (ORDER >= 2) {
            int h = 0;
#pragma unroll
            for (int d1 = 0; d1 < SD; ++d1)
#pragma unroll
                for (int d2 = d1; d2 < SD; ++d2) {
                    double t = cur.h[h] * f + u[d1] * cur.g[d2] + u[d2] * cur.g[d1];
                    t += prv.h[h] * g + dg[d1] * prv.g[d2] + dg[d2] * prv.g[d1];
                    t += u[6 + h] * prv.v;
                    nw.h[h] = t;
                    ++h;
                }
        }
    }
}

// In-place form: the new member overwrites `p` (the older of the two inputs).  Components
// are produced highest derivative first, so every input is read before it is replaced.
template <int SD, int ORDER, typename UPtr>
__device__ __forceinline__ void apply_step_uniform_inplace(const Jet<SD, ORDER>& cur, Jet<SD, ORDER>& p, double fa,
                                                           double fb, double fc, double A, double B, double C,
                                                           UPtr u) {
    const double f = A * fa - B * fb;
    const double g = -C * fc;
    if constexpr (ORDER >= 1) {
        double dg[SD];
#pragma unroll
        for (int d = 0; d < SD; ++d) dg[d] = u[3 + d] * fb;
        if constexpr (ORDER >= 2) {
            int h = 0;
#pragma unroll
            for (int d1 = 0; d1 < SD; ++d1)
#pragma unroll
                for (int d2 = d1; d2 < SD; ++d2) {
                    double t = cur.h[h] * f + u[d1] * cur.g[d2] + u[d2] * cur.g[d1];
                    t += p.h[h] * g + dg[d1] * p.g[d2] + dg[d2] * p.g[d1];
                    t += u[6 + h] * p.v;
                    p.h[h] = t;
                    ++h;
                }
        }
#pragma unroll
        for (int d = 0; d < SD; ++d) p.g[d] = cur.g[d] * f + cur.v * u[d] + p.g[d] * g + p.v * dg[d];
    }
    p.v = cur.v * f + p.v * g;
}

template <int SD, int ORDER>
__device__ __forceinline__ void apply_step_inplace(const Jet<SD, ORDER>& cur, Jet<SD, ORDER>& p,
                                                   const Factors<SD, ORDER>& F, double A, double B, double C) {
    const double f = A * F.fa - B * F.fb;
    const double g = -C * F.fc;
    if constexpr (ORDER >= 1) {
        double df[SD], dg[SD];
#pragma unroll
        for (int d = 0; d < SD; ++d) {
            df[d] = A * F.dfa[d] - B * F.dfb[d];
            dg[d] = -C * F.dfc[d];
        }
        if constexpr (ORDER >= 2) {
            int h = 0;
#pragma unroll
            for (int d1 = 0; d1 < SD; ++d1)
#pragma unroll
                for (int d2 = d1; d2 < SD; ++d2) {
                    double t = cur.h[h] * f + df[d1] * cur.g[d2] + df[d2] * cur.g[d1];
                    t += p.h[h] * g + dg[d1] * p.g[d2] + dg[d2] * p.g[d1];
                    t += (-C * F.ddfc[h]) * p.v;
                    p.h[h] = t;
                    ++h;
                }
        }
#pragma unroll
        for (int d = 0; d < SD; ++d) p.g[d] = cur.g[d] * f + cur.v * df[d] + p.g[d] * g + p.v * dg[d];
    }
    p.v = cur.v * f + p.v * g;
}

// fa, fb, fc of one codimension from the reference coordinates (no derivatives)
template <int SD>
__device__ __forceinline__ void point_factors(int codim, const double* X, double& fa, double& fb, double& fc) {
    double x = X[0], y = -1.0, z = -1.0;
    if (codim == 0) {
        x = X[0];
        if constexpr (SD > 1) y = X[1];
        if constexpr (SD > 2) z = X[2];
    } else if (codim == 1) {
        if constexpr (SD > 1) x = X[1];
        if constexpr (SD > 2) y = X[2];
    } else {
        if constexpr (SD > 2) x = X[2];
    }
    fb = 0.5 * (y + z);
    fa = x + (fb + 1.0);
    fc = fb * fb;
}

template <int SD, int N> struct FixedNC {
    static constexpr int value = 3 * (cx_binom(N + SD, SD) - 1);
};

template <int SD, int N, int ORDER, int ROWS, int NT, int NW>
__global__ __launch_bounds__(64 * NW, 2) void tabulate_simplex_fixed(const FixedArgs<FixedNC<SD, N>::value> a) {
    constexpr int NTAB = NTab<SD, ORDER>::value;
    constexpr StepTable<SD, N> TBL{};
    constexpr int NEXP = StepTable<SD, N>::NEXP;
    constexpr int KS = (NEXP + 3) / 4;
    constexpr int MT16 = rows_full16(ROWS);
    constexpr int M4 = rows_blk4(ROWS);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double* phi = lds + (size_t)wave * a.lds_doubles;

    const int npts = a.npts;
    const int table = ROWS * npts;
    const int reqsize = NTAB * table;
    const int ncols = NTAB * npts;

    for (int i = lane; i < a.lds_doubles; i += 64) phi[i] = 0.0;
    wave_lds_fence();

    // A fragments stay in registers for the whole kernel
    double areg[MT16 * KS > 0 ? MT16 * KS : 1];
    double areg4[M4 * KS > 0 ? M4 * KS : 1];
#pragma unroll
    for (int i = 0; i < MT16 * KS; ++i) areg[i] = a.afrag[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < M4 * KS; ++i) areg4[i] = a.afrag[(MT16 * KS + i) * 64 + lane];

    // image offset (doubles) of this lane's output column in tile nt, row (lane>>4); -1: no column
    int soff[NT];
    {
        const float rinv = 1.0f / (float)npts;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = (nt << 4) + (lane & 15);
            const int ct = idiv_small(c, rinv);
            const int cp = c - ct * npts;
            soff[nt] = (c < ncols) ? ct * table + cp + (lane >> 4) * npts : -1;
        }
    }
    // recurrence lanes: lane <-> point (lanes 0..npts-1; npts <= 32 for this kernel).
    // LDS stores are packed: lanes 32.. carry the odd component of each pair.
    const bool active = lane < npts;
    const int pl = active ? lane : 0;
    const int pu = ((lane & 31) < npts) ? (lane & 31) : 0;  // point of this lane for packed stores
    const bool active_pair = (lane & 31) < npts;
    int colbase[NTAB];
#pragma unroll
    for (int t = 0; t < NTAB; ++t) {
        const int c = t * npts + pl;
        colbase[t] = (c >> 4) * KS * 64 + (c & 15);
    }
    int pairbase[NTAB / 2 > 0 ? NTAB / 2 : 1];
#pragma unroll
    for (int u = 0; u < NTAB / 2; ++u) {
        const int c = (2 * u + (lane >> 5)) * npts + pu;
        pairbase[u] = (c >> 4) * KS * 64 + (c & 15);
    }

    const long long stride = (long long)gridDim.x * NW;
    long long req = (long long)blockIdx.x * NW + wave;
    double xnext[SD];  // points of the NEXT request are fetched while this one is computed
    if (req < a.nreq) {
        const double* pp = a.pts + ((size_t)req * npts + pl) * SD;
#pragma unroll
        for (int d = 0; d < SD; ++d) xnext[d] = pp[d];
    }
    for (; req < a.nreq; req += stride) {
        // ---------------- phase 1: recurrence ----------------
        double X[SD];
        double J[SD][SD];
        {
            double x[SD];
#pragma unroll
            for (int d = 0; d < SD; ++d) x[d] = xnext[d];
            if (req + stride < a.nreq) {
                const double* pp = a.pts + ((size_t)(req + stride) * npts + pl) * SD;
#pragma unroll
                for (int d = 0; d < SD; ++d) xnext[d] = pp[d];
            }
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
        auto put = [&](int k, const Jet<SD, ORDER>& j) {
            const int kofs = (k >> 2) * 64 + (k & 3) * 16;
            double comp[NTAB];
            comp[0] = j.v;
            if constexpr (ORDER >= 1) {
#pragma unroll
                for (int d = 0; d < SD; ++d) comp[1 + d] = j.g[d];
            }
            if constexpr (ORDER >= 2) {
#pragma unroll
                for (int h = 0; h < SD * (SD + 1) / 2; ++h) comp[1 + SD + h] = j.h[h];
            }
            if constexpr (FX_PACK) {
#pragma unroll
                for (int u = 0; u < NTAB / 2; ++u) {
                    const double packed = pack_halves(comp[2 * u], comp[2 * u + 1]);
                    if (active_pair) phi[pairbase[u] + kofs] = packed;
                }
                if constexpr (NTAB % 2 == 1) {
                    if (active) phi[colbase[NTAB - 1] + kofs] = comp[NTAB - 1];
                }
            } else {
                if (active) {
#pragma unroll
                    for (int t = 0; t < NTAB; ++t) phi[colbase[t] + kofs] = comp[t];
                }
            }
        };
        if (!FX_ABL(a, 1)) {
            // every member lives in a register array with compile-time indices: the
            // compiler keeps only the seeds still needed by later chains alive, and
            // the recurrence never reads LDS back
            Jet<SD, ORDER> mem[NEXP];
            jet_zero(mem[0]);
            mem[0].v = a.phi0;
            put(0, mem[0]);
            if constexpr (4 * KS > NEXP) {  // K padding rows must read as zero
                Jet<SD, ORDER> z;
                jet_zero(z);
#pragma unroll
                for (int k = NEXP; k < 4 * KS; ++k) put(k, z);
            }
            Factors<SD, ORDER> F;
            Jet<SD, ORDER> zero;
            jet_zero(zero);
            int fcodim = -1;
#pragma unroll
            for (int s = 0; s < (NEXP > 1 ? NEXP - 1 : 0); ++s) {
                if (TBL.codim[s] != fcodim) {
                    fcodim = TBL.codim[s];
                    make_factors<SD, ORDER>(F, fcodim, X, J);
                }
                const double cA = a.coef[3 * s + 0], cB = a.coef[3 * s + 1], cC = a.coef[3 * s + 2];
                apply_step<SD, ORDER>(mem[TBL.dst[s]], mem[TBL.cur[s]], TBL.prv[s] < 0 ? zero : mem[TBL.prv[s]], F, cA,
                                      cB, cC);
                put(TBL.dst[s], mem[TBL.dst[s]]);
            }
        }
        wave_lds_fence();

        // ---------------- phase 2: contraction ----------------
        double* gout = a.out + (size_t)req * reqsize;
        if (!FX_ABL(a, 2)) {
            double breg[NT][KS];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) breg[nt][ks] = phi[(nt * KS + ks) * 64 + lane];
            wave_lds_fence();  // every Phi read is done: the image may overwrite it
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int so = soff[nt];
#pragma unroll
                for (int mt = 0; mt < MT16; ++mt) {
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[mt * KS + ks], breg[nt][ks], acc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        constexpr int dummy = 0;
                        (void)dummy;
                        const int mbase = 16 * mt + 4 * j;  // + (lane >> 4)
                        if (mbase + 3 < ROWS) {
                            if (so >= 0) phi[so + mbase * npts] = acc[j];
                        } else if (mbase < ROWS) {
                            if (so >= 0 && mbase + (lane >> 4) < ROWS) phi[so + mbase * npts] = acc[j];
                        }
                    }
                }
#pragma unroll
                for (int m4 = 0; m4 < M4; ++m4) {
                    double acc = 0.0;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(areg4[m4 * KS + ks], breg[nt][ks], acc, 0, 0, 0);
                    const int mbase = 16 * MT16 + 4 * m4;
                    if (mbase + 3 < ROWS) {
                        if (so >= 0) phi[so + mbase * npts] = acc;
                    } else {
                        if (so >= 0 && mbase + (lane >> 4) < ROWS) phi[so + mbase * npts] = acc;
                    }
                }
            }
        }
        wave_lds_fence();

        // ---------------- phase 3: image -> HBM, 16 B per lane ----------------
        if (!FX_ABL(a, 4)) {
            if ((reqsize & 1) == 0 && !FX_UNROLL_COPY) {
                const v2d* s2 = reinterpret_cast<const v2d*>(phi);
                v2d* g2 = reinterpret_cast<v2d*>(gout);
#pragma unroll 4
                for (int i = lane; i < (reqsize >> 1); i += 64) g2[i] = s2[i];
            } else if ((reqsize & 1) == 0) {
                // all LDS reads are issued before the first store so that their
                // latencies overlap (NT column tiles bound the image size)
                constexpr int NIT = (NT * 16 * ROWS / 2 + 63) / 64;
                const v2d* s2 = reinterpret_cast<const v2d*>(phi);
                v2d* g2 = reinterpret_cast<v2d*>(gout);
                const int nchunk = reqsize >> 1;
                v2d buf[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int i = it * 64 + lane;
                    if (i < nchunk) buf[it] = s2[i];
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int i = it * 64 + lane;
                    if (i < nchunk) g2[i] = buf[it];
                }
            } else {
                for (int i = lane; i < reqsize; i += 64) gout[i] = phi[i];
            }
        }
        wave_lds_fence();
    }
}

}  // namespace fxk
