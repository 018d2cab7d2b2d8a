// K-streamed, shape-specialised simplex tabulation kernel (gfx950).
//
// The coefficient x expansion contraction is accumulated WHILE the recurrence
// runs: every four finished members (one MFMA K-step, in production order --
// the K order of a dot product is free, the host packs the A fragments in the
// same order) are written to a single 16*NT-column LDS slab, read back as
// B-operand fragments and multiplied into accumulators that stay in registers
// for the whole request.  Consequences, all measured on MI355X:
//   * LDS per wave drops from 15 KB (whole Phi tile) to NT*512 B, so occupancy is
//     bounded by registers only (fp64 VALU needs >= 3 waves/SIMD to approach its
//     4-cycle issue rate; at 2 waves it costs ~7 cycles per instruction);
//   * the recurrence of K-step j+1 is issued between the LDS stores of K-step j
//     and their read-back, hiding the LDS latency, and overlaps with the MFMAs
//     of K-step j (separate pipes);
//   * finished D tiles go to HBM through an LDS image of HALF a request (the
//     first / last ceil(NTAB/2) tables), flushed twice per request with 16-byte-
//     per-lane full-line stores.  Storing the accumulators directly (8 B per lane,
//     partial 128-B lines) was measured to double the HBM traffic: the memory
//     side reads every partially written line back (FETCH_SIZE ~ output size).
#pragma once
#include "simplex_fixed.hpp"
#include "store.hpp"

#ifndef FX_STREAM_WAVES
#define FX_STREAM_WAVES 2  // minimum waves per SIMD requested from the register allocator
#endif

namespace fxk {

template <int SD, int N, int ORDER, int ROWS, int NT, int NW, bool UNIFORM>
__global__ __launch_bounds__(64 * NW, FX_STREAM_WAVES) void tabulate_simplex_stream(const FixedArgs<FixedNC<SD, N>::value> a) {
    constexpr int NTAB = NTab<SD, ORDER>::value;
    constexpr StepTable<SD, N> TBL{};
    constexpr int NEXP = StepTable<SD, N>::NEXP;
    constexpr int KS = (NEXP + 3) / 4;
    constexpr int MT16 = rows_full16(ROWS);
    constexpr int M4 = rows_blk4(ROWS);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    // wave index as a scalar: everything derived from it (request number, output base) stays in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // LDS: [A fragments, shared by the workgroup] [per wave: half image (the slab aliases its
    // start) | 64-double dump row for inactive lanes]
    constexpr int NAF = (MT16 + M4) * KS;
    double* afr = lds;
    double* img = lds + NAF * 64 + (size_t)wave * a.lds_doubles;
    double* slab = img;
    const int dump = a.lds_doubles - 64;  // doubles from img

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
    // A fragments (production-order K): LDS resident, two 512-B reads per K-step
    for (int i = threadIdx.x; i < NAF * 64; i += 64 * NW) afr[i] = a.afrag[i];
    __syncthreads();

    // recurrence: lanes 0..npts-1 <-> points; the other lanes store into the dump row,
    // so no store needs an exec mask
    const bool active = lane < npts;
    const int pl = active ? lane : 0;
    int colbase[NTAB];
#pragma unroll
    for (int t = 0; t < NTAB; ++t) {
        const int c = t * npts + pl;
        colbase[t] = active ? (c >> 4) * 64 + (c & 15) : dump + (lane & 15);  // + kk*16 stays inside the dump row
    }
    // output image: tables [0, TH) form half 0, [TH, NTAB) half 1
    constexpr int TH = (NTAB + 1) / 2;
    // store instructions per half-image flush: NT tiles bound the points (16*NT >= NTAB*npts)
    constexpr int NFL = (TH * ROWS * ((16 * NT) / NTAB) / 2 + 63) / 64;
    // (NFL * (NTAB > TH ? 2 : 1) vector-memory stores per request)
    int ioff[NT];   // offset (doubles) inside its half image of this lane's column, row (lane>>4); -1: none
    int ihalf = 0;  // bit nt set: the column of tile nt belongs to half 1
    {
        const float rinv = 1.0f / (float)npts;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = (nt << 4) + (lane & 15);
            const int ct = idiv_small(c, rinv);
            const int cp = c - ct * npts;
            const int h = ct >= TH ? 1 : 0;
            ioff[nt] = (c < ncols) ? (ct - h * TH) * table + cp + (lane >> 4) * npts : -1;
            ihalf |= h << nt;
        }
    }
    const long long stride = (long long)gridDim.x * NW;
    long long req = (long long)blockIdx.x * NW + wave;
    // Points of the NEXT request are fetched while this one is computed.  vmcnt counts loads
    // and stores together, in issue order: if the prefetched registers were first used at the
    // top of the next iteration, hipcc (which cannot count stores across the loop back-edge)
    // would wait with vmcnt(0) there, i.e. for every HBM store of this request -- measured: the
    // wave then idles for the write latency once per request.  Instead the load is issued
    // before the request's stores, whose number is a compile-time constant, and "used" by an
    // empty asm at the END of the same iteration: hipcc then emits the exact vmcnt(NSTORE),
    // which leaves the stores in flight.
    double xnext[SD];
    if (req < a.nreq) {
        const double* pp = a.pts + ((size_t)req * npts + pl) * SD;
#pragma unroll
        for (int d = 0; d < SD; ++d) xnext[d] = pp[d];
    }
    // complete the first fetch before the loop: with a load pending on loop entry hipcc
    // puts a conservative vmcnt wait at the top of EVERY iteration (= wait for the stores)
#pragma unroll
    for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xnext[d]));
    for (; req < a.nreq; req += stride) {
        double X[SD];
        double J[SD][SD];
        {
            double x[SD];
#pragma unroll
            for (int d = 0; d < SD; ++d) x[d] = xnext[d];
            {
                // always issued (clamped to the last request) so that the count below is exact
                const long long rn = (req + stride < a.nreq) ? req + stride : req;
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

        // member of production slot `slot` (0: the constant, s: destination of step s-1)
        auto produce = [&](int slot) {
            if (slot == 0) {
                jet_zero(mem[0]);
                mem[0].v = a.phi0;
            } else if (slot < NEXP) {
                const int s = slot - 1;
                // this step's coefficients, read from the kernel-argument segment (constant
                // address space -> scalar loads).  The pointers are made opaque so that the
                // loads are issued here, one step at a time, instead of all being hoisted to
                // the top of the request (hundreds of SGPR spills otherwise).
                // (the BASE pointers are made opaque, the step's offset stays an immediate of the
                // scalar load: opaque per-step pointers get precomputed outside the request loop
                // and spilled, 2 x 19 64-bit SGPR pairs read back with v_readlane per request)
                const CDouble* cb = kcoef;
                const CDouble* ub = kucoef;
                asm volatile("" : "+s"(cb), "+s"(ub));
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

        // accumulators of tile nt -> half image `half` (lanes whose column belongs to it)
        auto image_tile = [&](int nt, int half) {
            const int so = ioff[nt];
            const bool mine = so >= 0 && ((ihalf >> nt) & 1) == half;
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int mbase = 16 * mt + 4 * jj;  // + (lane >> 4)
                    if (mbase + 3 < ROWS) {
                        if (mine) img[so + mbase * npts] = acc16[nt][mt][jj];
                    } else if (mbase < ROWS) {
                        if (mine && mbase + (lane >> 4) < ROWS) img[so + mbase * npts] = acc16[nt][mt][jj];
                    }
                }
            }
#pragma unroll
            for (int m4 = 0; m4 < M4; ++m4) {
                const int mbase = 16 * MT16 + 4 * m4;
                if (mbase + 3 < ROWS) {
                    if (mine) img[so + mbase * npts] = acc4[nt][m4];
                } else {
                    if (mine && mbase + (lane >> 4) < ROWS) img[so + mbase * npts] = acc4[nt][m4];
                }
            }
        };
        // half image -> HBM, 16 B per lane, whole 128-B lines except at the seam of the halves.
        // The number of store instructions is a compile-time constant (lanes past the end
        // rewrite the last chunk): the exact vmcnt wait on the prefetched points relies on it.
        auto flush_half = [&](int half) {
            const int ntab_h = half == 0 ? TH : NTAB - TH;
            const int nch = (ntab_h * table) >> 1;  // 16-byte chunks (even sizes only, checked by the host)
            const v2d* s2 = reinterpret_cast<const v2d*>(img);
            v2d* g2 = reinterpret_cast<v2d*>(a.out + (size_t)req * reqsize + (size_t)half * TH * table);
            v2d buf[NFL];
#pragma unroll
            for (int it = 0; it < NFL; ++it) {
                const int i = min(it * 64 + lane, nch - 1);
                buf[it] = s2[i];
            }
#pragma unroll
            for (int it = 0; it < NFL; ++it) {
                const int i = min(it * 64 + lane, nch - 1);
                stream_store(&g2[i], buf[it]);
            }
        };

#pragma unroll
        for (int kk = 0; kk < 4; ++kk) produce(kk);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            // K-step j -> LDS; the recurrence of K-step j+1 hides the store latency
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) put(kk, slot_jet(4 * j + kk));
            if (j + 1 < KS) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) produce(4 * (j + 1) + kk);
            }
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
                wave_lds_fence();  // reads issued before the next K-step's stores
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
        }

        // ---------------- D tiles -> half images -> HBM ----------------
        if (!FX_ABL(a, 4)) {
            wave_lds_fence();  // the slab (aliasing the image) has been read for the last K-step
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) image_tile(nt, 0);
            wave_lds_fence();
            flush_half(0);
            wave_lds_fence();
            if constexpr (NTAB > TH) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) image_tile(nt, 1);
                wave_lds_fence();
                flush_half(1);
                wave_lds_fence();
            }
            // first use of the prefetched points: hipcc places s_waitcnt vmcnt(NSTORE) here
#pragma unroll
            for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xnext[d]));
        } else {
#pragma unroll
            for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xnext[d]));
        }
    }
}

}  // namespace fxk
