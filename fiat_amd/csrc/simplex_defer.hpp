// K-streamed simplex tabulation kernel with DEFERRED output (gfx950).
//
// Same contraction as simplex_stream.hpp (one request per wave, K-streamed slab, accumulators
// in registers), different epilogue: the finished D tiles of request i are written to a
// whole-request LDS image, and that image is copied to HBM in KS parts DURING the K loop of
// request i+1 (a few 16-byte-per-lane full-line stores after the MFMAs of every K-step).
// Why: measured on MI355X, the burst of a whole request's stores at the end of each request
// blocks the wave at store issue while the memory system drains (2 waves per SIMD cannot cover
// it): 205 us of compute + 220-240 us of HBM writes took ~300 us.  Spreading the stores over
// the compute phase keeps the write queues fed without stalling the issuing wave, removes the
// masked half-image LDS stores (every tile is written once, all lanes) and hides the LDS
// read-back behind the MFMAs.
// LDS per wave: image (NTAB*ROWS*npts doubles) + slab (NT*64) + dump row (64).
#pragma once
#include "simplex_stream.hpp"

#ifndef FX_DBG
#define FX_DBG 0
#endif
#ifndef FX_DEFER_WAVES
#define FX_DEFER_WAVES 2
#endif

namespace fxk {

template <int SD, int N, int ORDER, int ROWS, int NT, int NW, bool UNIFORM>
__global__ __launch_bounds__(64 * NW, FX_DEFER_WAVES) void tabulate_simplex_defer(const FixedArgs<FixedNC<SD, N>::value> a,
                                                                                  double* __restrict__ trash) {
    constexpr int NTAB = NTab<SD, ORDER>::value;
    constexpr StepTable<SD, N> TBL{};
    constexpr int NEXP = StepTable<SD, N>::NEXP;
    constexpr int KS = (NEXP + 3) / 4;
    constexpr int MT16 = rows_full16(ROWS);
    constexpr int M4 = rows_blk4(ROWS);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NAF = (MT16 + M4) * KS;
    unsigned int* wq = reinterpret_cast<unsigned int*>(lds);  // work counter of the workgroup (first 16 bytes)
    double* afr = lds + 2;
    double* img = afr + NAF * 64 + (size_t)wave * a.lds_doubles;
    const int dump = a.lds_doubles - 64;       // doubles from img
    double* slab = img + (dump - NT * 64);     // [NT][4][16], separate from the image
    const int sdump = NT * 64;                 // the dump row, seen from the slab

    typedef const __attribute__((address_space(4))) double CDouble;
    typedef FixedArgs<FixedNC<SD, N>::value> ArgsT;
    const __attribute__((address_space(4))) char* kargs =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    CDouble* kcoef = (CDouble*)(kargs + __builtin_offsetof(ArgsT, coef));
    CDouble* kucoef = (CDouble*)(kargs + __builtin_offsetof(ArgsT, ucoef));

    const int npts = a.npts;
    const int table = ROWS * npts;
    const int reqsize = NTAB * table;
    const int ncols = NTAB * npts;

    for (int i = lane; i < a.lds_doubles; i += 64) img[i] = 0.0;
    for (int i = threadIdx.x; i < NAF * 64; i += 64 * NW) afr[i] = a.afrag[i];
    if (threadIdx.x == 0) wq[0] = 0;
    __syncthreads();

    const bool active = lane < npts;
    const int pl = active ? lane : 0;
    int colbase[NTAB];
#pragma unroll
    for (int t = 0; t < NTAB; ++t) {
        const int c = t * npts + pl;
        colbase[t] = active ? (c >> 4) * 64 + (c & 15) : sdump + (lane & 15);
    }
    // store instructions per request (16 B per lane); NT tiles bound the points: 16*NT >= NTAB*npts
    constexpr int NFLT = (NTAB * ROWS * ((16 * NT) / NTAB) / 2 + 63) / 64;
    int ioff[NT];  // image offset of this lane's column of tile nt, row (lane >> 4); -1: padding column
    {
        const float rinv = 1.0f / (float)npts;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = (nt << 4) + (lane & 15);
            const int ct = idiv_small(c, rinv);
            const int cp = c - ct * npts;
            ioff[nt] = (c < ncols) ? ct * table + cp + (lane >> 4) * npts : -1;
        }
    }
    // The waves of a workgroup (NW = 8: two per SIMD, the whole CU) claim the workgroup's
    // requests one at a time from a counter in LDS.  A static split per
    // wave leaves a long tail: the SIMD arbiter favours the older of its two resident waves,
    // measured lifetimes 267 us vs 336 us for equal shares, i.e. the CU runs at one wave per
    // SIMD for the last fifth of the launch.  (A counter in global memory costs a device-scope
    // atomic per request, ~0.5 us latency and 12 ns per claim when shared by every wave.)
    // claim k of workgroup b is request b + k*gridDim.x: the whole grid works inside one moving
    // window of the batch, the concurrent HBM writes cover a compact address range
    const long long hi = a.nreq;
    const int cshift = (a.debug >> 8) & 31;
    auto claim = [&]() -> long long {
        unsigned int r = 0;
        if (lane == 0) r = __hip_atomic_fetch_add(wq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const long long k = (long long)__builtin_amdgcn_readfirstlane(r);
        if (cshift == 31) {  // one contiguous range per workgroup
            const long long lo = (long long)blockIdx.x * a.nreq / gridDim.x, up = (long long)(blockIdx.x + 1) * a.nreq / gridDim.x;
            return lo + k < up ? lo + k : a.nreq;
        }
        const long long blk = k >> cshift;  // chunks of 2^cshift consecutive units per workgroup
        return ((blk * gridDim.x + blockIdx.x) << cshift) + (k & ((1LL << cshift) - 1));
    };
    long long req = claim();
    if (req >= hi) return;
    long long rnext = claim();
#if FX_DBG & 512
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // destination of the image currently held in LDS; the first one (zeros) goes to a scratch
    // area so that the loop body is the same for every request (no branch: hipcc's vmcnt
    // bookkeeping for the prefetched points stays exact, see simplex_stream.hpp)
    double* pout = trash;

    double xnext[SD];
    {
        const double* pp = a.pts + ((size_t)req * npts + pl) * SD;
#pragma unroll
        for (int d = 0; d < SD; ++d) xnext[d] = pp[d];
    }
#pragma unroll
    for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xnext[d]));

    // part `part` of KS of the image -> HBM (whole 128-B lines, 16 B per lane)
    auto flush_part = [&](double* dst, int part, int flane) {
        const int nch = reqsize >> 1;  // 16-byte chunks (even sizes only, checked by the host)
        const v2d* s2 = reinterpret_cast<const v2d*>(img);
        v2d* g2 = reinterpret_cast<v2d*>(dst);
        const int it0 = part * NFLT / KS, it1 = (part + 1) * NFLT / KS;
        v2d buf[(NFLT + KS - 1) / KS + 1];
#pragma unroll
        for (int it = it0; it < it1; ++it) buf[it - it0] = s2[min(it * 64 + flane, nch - 1)];
#pragma unroll
        for (int it = it0; it < it1; ++it) stream_store(&g2[min(it * 64 + flane, nch - 1)], buf[it - it0]);
    };

    while (true) {
        double X[SD];
        double J[SD][SD];
        {
            double x[SD];
#pragma unroll
            for (int d = 0; d < SD; ++d) x[d] = xnext[d];
            // fetch the points of the next request (of this one again when the range is
            // exhausted: the load is always issued so that the count stays exact)
            {
                const long long rn = rnext < hi ? rnext : req;
                const double* pp = a.pts + ((size_t)rn * npts + pl) * SD;
#pragma unroll
                for (int d = 0; d < SD; ++d) xnext[d] = pp[d];
            }

            double bb[SD];
            if constexpr (!UNIFORM) {
                cell_map<SD>(a.verts + (size_t)req * (SD + 1) * SD, J, bb);
#pragma unroll
                for (int i = 0; i < SD; ++i) {
                    double t = bb[i];
#pragma unroll
                    for (int d = 0; d < SD; ++d) t += J[i][d] * x[d];
                    X[i] = t;
                }
            } else {
#pragma unroll
                for (int i = 0; i < SD; ++i) {
                    double t = a.b0[i];
#pragma unroll
                    for (int d = 0; d < SD; ++d) t += a.A0[i * SD + d] * x[d];
                    X[i] = t;
                }
            }
        }

        v4d acc16[NT][MT16 > 0 ? MT16 : 1];
        double acc4[NT][M4 > 0 ? M4 : 1];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) acc16[nt][mt] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int m4 = 0; m4 < M4; ++m4) acc4[nt][m4] = 0.0;
        }

        Jet<SD, ORDER> mem[NEXP];
        Jet<SD, ORDER> zero;
        jet_zero(zero);
        Factors<SD, ORDER> F;
        double ufa = 0.0, ufb = 0.0, ufc = 0.0;
        int fcodim = -1;

        auto produce = [&](int slot) {
            if (slot == 0) {
                jet_zero(mem[0]);
                mem[0].v = a.phi0;
            } else if (slot < NEXP) {
                const int s = slot - 1;
                const CDouble* cb = kcoef;
                const CDouble* ub = kucoef;
                asm volatile("" : "+s"(cb), "+s"(ub));  // see simplex_stream.hpp
                const CDouble* cp = cb + 3 * s;
                const CDouble* up = ub + 12 * s;
                const double cA = cp[0], cB = cp[1], cC = cp[2];
                if constexpr (UNIFORM) {
                    if (TBL.codim[s] != fcodim) {
                        fcodim = TBL.codim[s];
                        point_factors<SD>(fcodim, X, ufa, ufb, ufc);
                    }
                    apply_step_uniform<SD, ORDER>(mem[TBL.dst[s]], mem[TBL.cur[s]],
                                                  TBL.prv[s] < 0 ? zero : mem[TBL.prv[s]], ufa, ufb, ufc, cA, cB, cC,
                                                  up);
                } else {
                    if (TBL.codim[s] != fcodim) {
                        fcodim = TBL.codim[s];
                        make_factors<SD, ORDER>(F, fcodim, X, J);
                    }
                    apply_step<SD, ORDER>(mem[TBL.dst[s]], mem[TBL.cur[s]], TBL.prv[s] < 0 ? zero : mem[TBL.prv[s]],
                                          F, cA, cB, cC);
                }
            }
        };
        auto slot_jet = [&](int slot) -> const Jet<SD, ORDER>& {
            if (slot == 0) return mem[0];
            if (slot < NEXP) return mem[TBL.dst[slot - 1]];
            return zero;
        };
        auto put = [&](int kk, const Jet<SD, ORDER>& j) {
            slab[colbase[0] + kk * 16] = j.v;
            if constexpr (ORDER >= 1) {
#pragma unroll
                for (int d = 0; d < SD; ++d) slab[colbase[1 + d] + kk * 16] = j.g[d];
            }
            if constexpr (ORDER >= 2) {
#pragma unroll
                for (int h = 0; h < SD * (SD + 1) / 2; ++h) slab[colbase[1 + SD + h] + kk * 16] = j.h[h];
            }
        };

        // K-steps strictly one after the other (fp64 MFMA and VALU share the SIMD's pipe:
        // nothing to gain from interleaving them, and the live registers stay at
        // accumulators + max(fragments, new members)); after the MFMAs of K-step j have been
        // issued, part j of the PREVIOUS request's image goes to HBM.
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) produce(4 * j + kk);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) put(kk, slot_jet(4 * j + kk));
            wave_lds_fence();
            {
                double b[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = slab[nt * 64 + lane];
                double a16[MT16 > 0 ? MT16 : 1], a4[M4 > 0 ? M4 : 1];
#pragma unroll
                for (int mt = 0; mt < MT16; ++mt) a16[mt] = afr[(mt * KS + j) * 64 + lane];
#pragma unroll
                for (int m4 = 0; m4 < M4; ++m4) a4[m4] = afr[((MT16 + m4) * KS + j) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int mt = 0; mt < MT16; ++mt)
                        acc16[nt][mt] =
                            __builtin_amdgcn_mfma_f64_16x16x4f64(a16[mt], b[nt], acc16[nt][mt], 0, 0, 0);
#pragma unroll
                    for (int m4 = 0; m4 < M4; ++m4)
                        acc4[nt][m4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[m4], b[nt], acc4[nt][m4], 0, 0, 0);
                }
            }
            {
                int flane = lane;
                asm volatile("" : "+v"(flane));  // chunk offsets recomputed here, not held across the loop
                flush_part(pout, j, flane);
            }
            wave_lds_fence();  // fragments and image part read before the slab / image are rewritten
        }

        // ---------------- D tiles -> whole-request image (flushed during the next request) ----------------
        {
            int elane = lane;
            asm volatile("" : "+v"(elane));
            const int dsink = dump + elane;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                int so = ioff[nt];
                asm volatile("" : "+v"(so));
                const bool mine = so >= 0;
#pragma unroll
                for (int mt = 0; mt < MT16; ++mt) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int mbase = 16 * mt + 4 * jj;  // + (lane >> 4)
                        if (mbase + 3 < ROWS) {
                            img[mine ? so + mbase * npts : dsink] = acc16[nt][mt][jj];
                        } else if (mbase < ROWS) {
                            img[(mine && mbase + (elane >> 4) < ROWS) ? so + mbase * npts : dsink] = acc16[nt][mt][jj];
                        }
                    }
                }
#pragma unroll
                for (int m4 = 0; m4 < M4; ++m4) {
                    const int mbase = 16 * MT16 + 4 * m4;
                    if (mbase + 3 < ROWS) {
                        img[mine ? so + mbase * npts : dsink] = acc4[nt][m4];
                    } else {
                        img[(mine && mbase + (elane >> 4) < ROWS) ? so + mbase * npts : dsink] = acc4[nt][m4];
                    }
                }
            }
            wave_lds_fence();
        }
        pout = a.out + (size_t)req * reqsize;
        if (a.debug & 4) pout = trash;
        // first use of the prefetched points: hipcc places s_waitcnt vmcnt(NFLT) here
#pragma unroll
        for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xnext[d]));
        if (rnext >= hi) break;
        req = rnext;
        rnext = claim();
    }
    // the last image
#pragma unroll
    for (int j = 0; j < KS; ++j) flush_part(pout, j, lane);
#if FX_DBG & 512
    // ablation build: shader cycles and 100 MHz real-time ticks this wave was alive -> average shader clock
    if (lane == 0) {
        const long long gw = (long long)blockIdx.x * NW + wave;
        if (gw < 3000) {
            trash[2048 + 2 * gw] = (double)(__builtin_readcyclecounter() - clk0);
            trash[2049 + 2 * gw] = (double)(__builtin_amdgcn_s_memrealtime() - rt0);
        }
    }
#endif
}

}  // namespace fxk
