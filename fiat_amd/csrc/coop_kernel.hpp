// Workgroup-cooperative, warp-specialised simplex tabulation kernel for LARGE shapes
// (gfx950), e.g. DG P6 tetrahedron with Hessians: 84 members x 10 tables x 23 points,
// 84 x 84 coefficients, 155 kB of tables per request.
//
// One request per 512-thread workgroup iteration, eight waves, two per SIMD:
//   * waves 0-3, PRODUCERS: each walks its share of the Dubiner recurrence (host plan
//     build_coop_plan: depth-first order, chains of the last codimension partitioned
//     over the four waves, three chain levels in registers) for all points and, once
//     per K-step, publishes ONE finished member (all tables) into row `wave` of a
//     double-buffered LDS slab laid out as MFMA B fragments;
//   * waves 4-7, CONSUMERS: own every fourth 16-column tile, keep its accumulators in
//     registers for the whole request (the per-wave kernels cannot: 15 tiles x 21
//     registers), read the slab of K-step j and the A fragments (LDS resident, 64 kB)
//     and issue v_mfma_f64_16x16x4 / 4x4x4_4b.
// A producer and a consumer share every SIMD, so the fp64 VALU work of K-step j+1 runs
// under the MFMAs of K-step j.  One workgroup barrier per K-step; the double buffer
// makes the second one unnecessary (see the hazard note at the barrier).
// Finished tiles leave through an LDS image of TR tables, flushed by all 512 threads
// with 16-byte stores of whole lines (direct 8-byte stores double the HBM traffic).
#pragma once
#include "simplex_fixed.hpp"
#include "store.hpp"

namespace fxk {

struct CoopArgs {
    const double* pts;    // [nreq][npts][SD]
    const double* verts;  // [nreq][SD+1][SD] or nullptr
    double* out;          // [nreq][ntab][rows][npts]
    const double* afrag;  // [(MT16+M4)][KS][64], K in slot order (4*j + producer)
    const int* eint;      // [4][emax][4] = level, seed, publish, member
    const double* edbl;   // [4][emax][16] = A, B, C, u[12], pad
    const int* kstart;    // [4][KS+1]
    double phi0;
    double A0[9];
    double b0[3];
    long long nreq;
    int npts, rows, KS, NT, emax;
    int TR;           // tables per image round
    int slab_doubles; // NT*64 B-fragment doubles + a 128-double dump row for inactive lanes (lane + 16 * K slot)
    int img_doubles;  // >= max(2*slab_doubles, TR*rows*npts)
    int debug;        // measurement only: 1 skip recurrence math, 2 skip MFMAs, 4 skip output rounds, 8 no LDS chain state
    // Piola push-forward fused into the output rounds (vector-valued elements with vdim == SD,
    // per-request cells): 0 none, 1 covariant (M = K^T), 2 contravariant (M = adj K), K = A0inv * A_req
    int piola;
    double A0inv[9];
};

// One output round: the image of nd doubles (whole tables of rows x npts) -> HBM.  With a Piola map
// every output element (dof, c, p) is the M-combination of the SD components (dof, ., p) of the
// image; which image offsets and which row of M a thread's elements need does not change from
// round to round or request to request and is precomputed (PiolaSlots).
constexpr int PIOLA_SLOTS = 6;  // pairs of output doubles per thread and round (512 threads)
struct PiolaSlots {
    int base[PIOLA_SLOTS][2];  // image offset of component 0 of the element's dof at its point
    int comp[PIOLA_SLOTS][2];  // its component
};

template <int SD, bool PIOLA>
__device__ __forceinline__ void coop_flush(const double* img, double* g, long long nd, bool pairs, int tid, int npts,
                                           const PiolaSlots& ps, const double* sM) {
    if constexpr (!PIOLA) {
        if (pairs) {
            const v2d* s2 = reinterpret_cast<const v2d*>(img);
            v2d* g2 = reinterpret_cast<v2d*>(g);
            for (int i = tid; i < (int)(nd >> 1); i += 512) stream_store(&g2[i], s2[i]);
        } else {
            for (int i = tid; i < (int)nd; i += 512) g[i] = img[i];
        }
        return;
    } else {
    double M[SD][SD];
#pragma unroll
    for (int r = 0; r < SD; ++r)
#pragma unroll
        for (int c = 0; c < SD; ++c) M[r][c] = sM[r * SD + c];
    v2d* g2 = reinterpret_cast<v2d*>(g);  // (host: Piola rounds are whole pairs)
#pragma unroll
    for (int k = 0; k < PIOLA_SLOTS; ++k) {
        const int i = tid + 512 * k;
        if (i < (int)(nd >> 1)) {
            v2d v;
#pragma unroll
            for (int el = 0; el < 2; ++el) {
                double acc = 0.0;
#pragma unroll
                for (int c = 0; c < SD; ++c) {
                    double m = M[0][c];
#pragma unroll
                    for (int r = 1; r < SD; ++r) m = ps.comp[k][el] == r ? M[r][c] : m;
                    acc += m * img[ps.base[k][el] + c * npts];
                }
                if (el == 0) v.x = acc; else v.y = acc;
            }
            stream_store(&g2[i], v);
        }
    }
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a full
// workgroup-scope fence, for which hipcc emits s_waitcnt vmcnt(0): every barrier after a
// flush would wait for the HBM stores to be acknowledged (microseconds, 11 times per request).
__device__ __forceinline__ void wg_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int SD, int ORDER, int MT16, int M4, int TPW, bool UNIFORM, bool PIOLA = false>
__global__ __launch_bounds__(512, 2) void tabulate_simplex_coop(const CoopArgs a) {
    static_assert(!PIOLA || !UNIFORM, "a Piola push-forward needs per-request cells");
    constexpr int NTAB = NTab<SD, ORDER>::value;
    constexpr int NAFT = MT16 + M4;
    typedef const __attribute__((address_space(4))) int CInt;
    typedef const __attribute__((address_space(4))) double CDouble;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KS = a.KS, NT = a.NT, npts = a.npts, rows = a.rows;
    const int table = rows * npts;
    const long long reqsize = (long long)NTAB * table;
    double* afr = lds;                               // NAFT*KS*64
    double* img = lds + (size_t)NAFT * KS * 64;      // image; the two slabs alias its start
    // producer chain state of levels 0/1: [producer][4 slots][NTAB][32 lanes]
    double* lstate = img + a.img_doubles + (size_t)(wave & 3) * 4 * NTAB * 32;

    for (int i = tid; i < NAFT * KS * 64; i += 512) afr[i] = a.afrag[i];
    for (int i = tid; i < a.img_doubles; i += 512) img[i] = 0.0;
    __syncthreads();

    const int nrounds = FX_ABL(a, 4) ? 0 : (NTAB + a.TR - 1) / a.TR;
    __shared__ double sM[9];  // Piola matrix of the current request (written by producer 0 before the K loop)
    PiolaSlots pslots;
    if constexpr (PIOLA) {
#pragma unroll
        for (int k = 0; k < PIOLA_SLOTS; ++k)
#pragma unroll
            for (int el = 0; el < 2; ++el) {
                const int e = 2 * (tid + 512 * k) + el;
                const int tt = e / table, rem = e - tt * table;
                const int row = rem / npts, p = rem - row * npts;
                const int dof = row / SD;
                pslots.comp[k][el] = row - dof * SD;
                // (positions past the image belong to no round: clamped so that the reads stay inside it)
                pslots.base[k][el] = min(tt * table + dof * SD * npts + p, a.img_doubles - (SD - 1) * npts - 1);
            }
    }

    if (wave < 4) {
        // =========================== producer ===========================
        const int w = wave;
        CInt* eint = (CInt*)a.eint + (size_t)w * a.emax * 4;
        CDouble* edbl = (CDouble*)a.edbl + (size_t)w * a.emax * 16;
        CInt* kst = (CInt*)a.kstart + (size_t)w * (KS + 1);
        const bool active = lane < npts;
        const int pl = active ? lane : 0;
        int colbase[NTAB];
#pragma unroll
        for (int t = 0; t < NTAB; ++t) {
            const int c = t * npts + pl;
            // (+ 16 * K slot, taken from the entry; inactive lanes store into the 128-double dump
            // row that closes every slab)
            colbase[t] = active ? (c >> 4) * 64 + (c & 15) : NT * 64 + lane;
        }
        for (long long req = blockIdx.x; req < a.nreq; req += gridDim.x) {
            double X[SD];
            double J[SD][SD];
            {
                double x[SD];
                const double* pp = a.pts + ((size_t)req * npts + pl) * SD;
#pragma unroll
                for (int d = 0; d < SD; ++d) x[d] = pp[d];
                double bb[SD];
                if constexpr (!UNIFORM) {
                    cell_map<SD>(a.verts + (size_t)req * (SD + 1) * SD, J, bb);
                    if (PIOLA && w == 0 && lane == 0) {
                        // K = A0inv * A_req = d(own cell coordinates)/dx; read after the K loop's barriers.
                        // (The previous request's last output round ended with a barrier.)
                        double K[SD][SD];
                        for (int c = 0; c < SD; ++c)
                            for (int d = 0; d < SD; ++d) {
                                double t = 0.0;
                                for (int k = 0; k < SD; ++k) t += a.A0inv[c * SD + k] * J[k][d];
                                K[c][d] = t;
                            }
                        for (int c = 0; c < SD; ++c)
                            for (int e = 0; e < SD; ++e) {
                                double v = K[e][c];  // covariant: J^{-T} = K^T
                                if (a.piola == 2) {  // contravariant: J / det J = adj(K)
                                    if constexpr (SD == 2) v = (c == e ? K[1 - c][1 - e] : -K[c][e]);
                                    else if constexpr (SD == 3) {
                                        const int c1 = (c + 1) % 3, c2 = (c + 2) % 3, e1 = (e + 1) % 3, e2 = (e + 2) % 3;
                                        v = K[e1][c1] * K[e2][c2] - K[e1][c2] * K[e2][c1];
                                    } else v = 1.0;
                                }
                                sM[c * SD + e] = v;
                            }
                    }
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
            // point factors of the three codimensions
            double pfa[SD], pfb[SD], pfc[SD];
#pragma unroll
            for (int c = 0; c < SD; ++c) point_factors<SD>(c, X, pfa[c], pfb[c], pfc[c]);

            // Chain state.  Level 2 (the long chains, ~2/3 of all steps) lives in registers as
            // two members (a2, b2) of which `par2` names the newer; a step overwrites the older
            // in place, no copies.  Levels 0 and 1 are touched rarely and live in LDS (two slots
            // each, 32 lanes): keeping all three levels in registers spilled into scratch, and
            // scratch reloads in this loop cost microseconds each.
            Jet<SD, ORDER> a2, b2;
            const double phi0 = a.phi0;
            auto set_const = [&](Jet<SD, ORDER>& j) {
                jet_zero(j);
                j.v = phi0;
            };
            jet_zero(a2);
            jet_zero(b2);
            int par0 = 0, par1 = 0, par2 = 0;  // which slot / register holds the newer member
            const int l31 = lane & 31;
            auto lst = [&](int slot, const Jet<SD, ORDER>& j) {  // registers -> LDS state slot
                if (lane < 32) {
                    double* q = lstate + (size_t)slot * NTAB * 32 + l31;
                    q[0] = j.v;
                    if constexpr (ORDER >= 1) {
#pragma unroll
                        for (int d = 0; d < SD; ++d) q[(1 + d) * 32] = j.g[d];
                    }
                    if constexpr (ORDER >= 2) {
#pragma unroll
                        for (int h = 0; h < SD * (SD + 1) / 2; ++h) q[(1 + SD + h) * 32] = j.h[h];
                    }
                }
            };
            auto lld = [&](int slot, Jet<SD, ORDER>& j) {  // LDS state slot -> registers
                const double* q = lstate + (size_t)slot * NTAB * 32 + l31;
                j.v = q[0];
                if constexpr (ORDER >= 1) {
#pragma unroll
                    for (int d = 0; d < SD; ++d) j.g[d] = q[(1 + d) * 32];
                }
                if constexpr (ORDER >= 2) {
#pragma unroll
                    for (int h = 0; h < SD * (SD + 1) / 2; ++h) j.h[h] = q[(1 + SD + h) * 32];
                }
            };

            auto put = [&](double* slab, const Jet<SD, ORDER>& j) {
                slab[colbase[0]] = j.v;
                if constexpr (ORDER >= 1) {
#pragma unroll
                    for (int d = 0; d < SD; ++d) slab[colbase[1 + d]] = j.g[d];
                }
                if constexpr (ORDER >= 2) {
#pragma unroll
                    for (int h = 0; h < SD * (SD + 1) / 2; ++h) slab[colbase[1 + SD + h]] = j.h[h];
                }
            };
            auto put_zero = [&](double* slab, double v0) {
                slab[colbase[0]] = v0;
#pragma unroll
                for (int t = 1; t < NTAB; ++t) slab[colbase[t]] = 0.0;
            };
            // older <- step(newer, older): the result replaces the older member
            auto step = [&](const Jet<SD, ORDER>& newer, Jet<SD, ORDER>& older, int codim, const double* d) {
                const double cA = d[0], cB = d[1], cC = d[2];
                if constexpr (UNIFORM) {
                    const double fa = codim == 0 ? pfa[0] : (codim == 1 ? pfa[SD > 1 ? 1 : 0] : pfa[SD > 2 ? 2 : 0]);
                    const double fb = codim == 0 ? pfb[0] : (codim == 1 ? pfb[SD > 1 ? 1 : 0] : pfb[SD > 2 ? 2 : 0]);
                    const double fc = codim == 0 ? pfc[0] : (codim == 1 ? pfc[SD > 1 ? 1 : 0] : pfc[SD > 2 ? 2 : 0]);
                    apply_step_uniform_inplace<SD, ORDER, const double*>(newer, older, fa, fb, fc, cA, cB, cC, d + 3);
                } else {
                    Factors<SD, ORDER> F;
                    make_factors<SD, ORDER>(F, codim, X, J);
                    apply_step_inplace<SD, ORDER>(newer, older, F, cA, cB, cC);
                }
            };
            // one step of an LDS-resident level (0 or 1)
            auto lds_level_step = [&](int level, int& par, int seed, int publish, double* slab, const double* d) {
                Jet<SD, ORDER> nw, od;
                if (FX_ABL(a, 8)) {  // ablation: no LDS chain state (wrong results)
                    nw = a2;
                    od = b2;
                    if (level == 0) step(nw, od, 0, d); else step(nw, od, 1, d);
                    if (publish) put(slab + (publish - 1) * 16, od);
                    return;
                }
                if (seed != -2) {  // chain start: newer = seed, older = 0
                    if (seed == -1) set_const(nw); else lld(0 + par0, nw);  // only level 1 has a level-0 seed
                    jet_zero(od);
                    par = 0;
                } else {
                    lld(2 * level + par, nw);
                    lld(2 * level + (par ^ 1), od);
                }
                if (seed != -2) lst(2 * level + 0, nw);
                if (level == 0) step(nw, od, 0, d); else step(nw, od, 1, d);
                par ^= 1;
                lst(2 * level + par, od);
                if (publish) put(slab + (publish - 1) * 16, od);
            };

            // The entry records are wave uniform and read with scalar loads.  Each record is
            // fetched as ONE block (ints + 15 doubles), one entry ahead of its use: per-field
            // loads cost a scalar-cache round trip and an lgkmcnt(0) drain each (~1000 s_load per
            // request measured before).
            int nlevel = eint[0], nseed = eint[1], npublish = eint[2];
            double nd_[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) nd_[i] = edbl[i];
            int e = 0;
            for (int ks = 0; ks < KS; ++ks) {
                double* slab = img + (size_t)(ks & 1) * a.slab_doubles;
                // readfirstlane: make the control values provably wave uniform (scalar branches,
                // no exec-mask juggling around every level / parity test)
                const int e1 = __builtin_amdgcn_readfirstlane(kst[ks + 1]);
                if (FX_ABL(a, 1)) e = e1;
                for (; e < e1; ++e) {
                    const int level = __builtin_amdgcn_readfirstlane(nlevel);
                    const int seed = __builtin_amdgcn_readfirstlane(nseed);
                    const int publish = __builtin_amdgcn_readfirstlane(npublish);
                    double dreg[15];
#pragma unroll
                    for (int i = 0; i < 15; ++i) dreg[i] = nd_[i];
                    {   // prefetch the next record (clamped: the table has emax rows)
                        const int en = min(e + 1, a.emax - 1);
                        nlevel = eint[en * 4 + 0];
                        nseed = eint[en * 4 + 1];
                        npublish = eint[en * 4 + 2];
#pragma unroll
                        for (int i = 0; i < 15; ++i) nd_[i] = edbl[(size_t)en * 16 + i];
                    }
                    const double* d = dreg;
                    if (level < 0) {
                        // zero pad row (seed == -2) or the constant member
                        if (publish) put_zero(slab + (publish - 1) * 16, seed == -2 ? 0.0 : phi0);
                    } else if (level == 0) {
                        lds_level_step(0, par0, seed, publish, slab, d);
                    } else if (level == 1) {
                        lds_level_step(1, par1, seed, publish, slab, d);
                    } else {
                        if (seed != -2) {
                            if (seed == -1) set_const(a2);
                            else if (seed == 0) lld(0 + par0, a2);
                            else lld(2 + par1, a2);
                            jet_zero(b2);
                            par2 = 0;
                        }
                        if (par2 == 0) {
                            step(a2, b2, 2, d);
                            if (publish) put(slab + (publish - 1) * 16, b2);
                        } else {
                            step(b2, a2, 2, d);
                            if (publish) put(slab + (publish - 1) * 16, a2);
                        }
                        par2 ^= 1;
                    }
                }
                // Barrier #ks: slab ks is complete.  No second barrier is needed before this
                // buffer is overwritten at K-step ks+2: a producer gets past barrier #ks+1 only
                // when every consumer has arrived there, i.e. after its reads of slab ks.
                wg_lds_barrier();
            }
            wg_lds_barrier();  // every consumer has read the last slab: the image may overwrite it
            // output rounds: the consumers fill the image, everybody flushes it
            for (int r = 0; r < nrounds; ++r) {
                wg_lds_barrier();  // image round r written
                const int t0 = r * a.TR;
                const int nt_r = min(a.TR, NTAB - t0);
                const long long nd = (long long)nt_r * table;
                double* g = a.out + (size_t)req * reqsize + (size_t)t0 * table;
                coop_flush<SD, PIOLA>(img, g, nd, ((a.TR * table) & 1) == 0 && (reqsize & 1) == 0, tid, npts, pslots, sM);
                wg_lds_barrier();  // image may be overwritten
            }
        }
    } else {
        // =========================== consumer ===========================
        const int c4 = wave - 4;
        // image offsets of this lane's column in each own tile: tile nt = c4 + 4*t
        int ioff[TPW];    // offset inside the round's image at row (lane>>4); -1: no column
        int iround[TPW];  // round the column's table belongs to
        {
            const float rinv = 1.0f / (float)npts;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int nt = c4 + 4 * t;
                const int c = (nt << 4) + (lane & 15);
                const int ct = idiv_small(c, rinv);
                const int cp = c - ct * npts;
                const bool ok = nt < NT && c < NTAB * npts;
                const int rr = ct / a.TR;
                iround[t] = ok ? rr : -1;
                ioff[t] = ok ? (ct - rr * a.TR) * table + cp + (lane >> 4) * npts : -1;
            }
        }
        for (long long req = blockIdx.x; req < a.nreq; req += gridDim.x) {
            v4d acc16[TPW][MT16 > 0 ? MT16 : 1];
            double acc4[TPW][M4 > 0 ? M4 : 1];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
#pragma unroll
                for (int mt = 0; mt < MT16; ++mt) acc16[t][mt] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int m4 = 0; m4 < M4; ++m4) acc4[t][m4] = 0.0;
            }
            // A fragments of K-step ks+1 are fetched while the MFMAs of K-step ks run
            double a16n[MT16 > 0 ? MT16 : 1], a4n[M4 > 0 ? M4 : 1];
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) a16n[mt] = afr[((size_t)mt * KS) * 64 + lane];
#pragma unroll
            for (int m4 = 0; m4 < M4; ++m4) a4n[m4] = afr[((size_t)(MT16 + m4) * KS) * 64 + lane];
            for (int ks = 0; ks < KS; ++ks) {
                wg_lds_barrier();  // barrier #ks: slab ks published
                const double* slab = img + (size_t)(ks & 1) * a.slab_doubles;
                double b[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int nt = min(c4 + 4 * t, NT - 1);
                    b[t] = slab[nt * 64 + lane];
                }
                double a16[MT16 > 0 ? MT16 : 1], a4[M4 > 0 ? M4 : 1];
#pragma unroll
                for (int mt = 0; mt < MT16; ++mt) a16[mt] = a16n[mt];
#pragma unroll
                for (int m4 = 0; m4 < M4; ++m4) a4[m4] = a4n[m4];
                {
                    const int kn = min(ks + 1, KS - 1);
#pragma unroll
                    for (int mt = 0; mt < MT16; ++mt) a16n[mt] = afr[((size_t)mt * KS + kn) * 64 + lane];
#pragma unroll
                    for (int m4 = 0; m4 < M4; ++m4) a4n[m4] = afr[((size_t)(MT16 + m4) * KS + kn) * 64 + lane];
                }
                if (!FX_ABL(a, 2))
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
#pragma unroll
                    for (int mt = 0; mt < MT16; ++mt)
                        acc16[t][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a16[mt], b[t], acc16[t][mt], 0, 0, 0);
#pragma unroll
                    for (int m4 = 0; m4 < M4; ++m4)
                        acc4[t][m4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[m4], b[t], acc4[t][m4], 0, 0, 0);
                }
            }
            wg_lds_barrier();  // every consumer has read the last slab: the image may overwrite it
            for (int r = 0; r < nrounds; ++r) {
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const bool mine = iround[t] == r;
                    if (__ballot(mine) == 0ull) continue;  // wave uniform: no lane of this tile in round r
                    const int so = ioff[t];
#pragma unroll
                    for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int mbase = 16 * mt + 4 * jj;
                            if (mine && mbase + (lane >> 4) < rows) img[so + mbase * npts] = acc16[t][mt][jj];
                        }
#pragma unroll
                    for (int m4 = 0; m4 < M4; ++m4) {
                        const int mbase = 16 * MT16 + 4 * m4;
                        if (mine && mbase + (lane >> 4) < rows) img[so + mbase * npts] = acc4[t][m4];
                    }
                }
                wg_lds_barrier();  // image round r written
                const int t0 = r * a.TR;
                const int nt_r = min(a.TR, NTAB - t0);
                const long long nd = (long long)nt_r * table;
                double* g = a.out + (size_t)req * reqsize + (size_t)t0 * table;
                coop_flush<SD, PIOLA>(img, g, nd, ((a.TR * table) & 1) == 0 && (reqsize & 1) == 0, tid, npts, pslots, sM);
                wg_lds_barrier();  // image may be overwritten
            }
        }
    }
}

}  // namespace fxk
