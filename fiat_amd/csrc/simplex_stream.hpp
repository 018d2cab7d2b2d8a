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
//   * finished D tiles are stored straight from the accumulators (8 B per lane,
//     16 lanes = one 128-B run of a table row); staging them through an LDS image
//     for 16-B-per-lane stores measured no faster and cost the LDS.
#pragma once
#include "simplex_fixed.hpp"

namespace fxk {

template <int SD, int N, int ORDER, int ROWS, int NT, int NW, bool UNIFORM>
__global__ __launch_bounds__(64 * NW, 2) void tabulate_simplex_stream(const FixedArgs<FixedNC<SD, N>::value> a) {
    constexpr int NTAB = NTab<SD, ORDER>::value;
    constexpr StepTable<SD, N> TBL{};
    constexpr int NEXP = StepTable<SD, N>::NEXP;
    constexpr int KS = (NEXP + 3) / 4;
    constexpr int MT16 = rows_full16(ROWS);
    constexpr int M4 = rows_blk4(ROWS);
    constexpr int SLAB = NT * 64;  // doubles
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    // wave index as a scalar: everything derived from it (request number, output base) stays in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // LDS: [A fragments, shared by the workgroup] [one slab per wave]
    constexpr int NAF = (MT16 + M4) * KS;
    double* afr = lds;
    double* slab = lds + NAF * 64 + (size_t)wave * SLAB;

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

    for (int i = lane; i < SLAB; i += 64) slab[i] = 0.0;
    // A fragments (production-order K): LDS resident, two 512-B reads per K-step
    for (int i = threadIdx.x; i < NAF * 64; i += 64 * NW) afr[i] = a.afrag[i];
    __syncthreads();

    // offset (doubles, within the request) of this lane's output column in tile nt, row (lane>>4)
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
    // recurrence: lanes 0..npts-1 <-> points (npts <= 32); the LDS stores are packed,
    // lanes 32.. carry the odd table of each pair (v_permlane32_swap)
    const bool active = lane < npts;
    const int pl = active ? lane : 0;
    const int pu = ((lane & 31) < npts) ? (lane & 31) : 0;
    const bool active_pair = (lane & 31) < npts;
    int pairbase[NTAB / 2 > 0 ? NTAB / 2 : 1];
#pragma unroll
    for (int u = 0; u < NTAB / 2; ++u) {
        const int c = (2 * u + (lane >> 5)) * npts + pu;
        pairbase[u] = (c >> 4) * 64 + (c & 15);
    }
    int lastbase = 0;
    if constexpr (NTAB % 2 == 1) {
        const int c = (NTAB - 1) * npts + pl;
        lastbase = (c >> 4) * 64 + (c & 15);
    }

    const long long stride = (long long)gridDim.x * NW;
    long long req = (long long)blockIdx.x * NW + wave;
    double xnext[SD];
    if (req < a.nreq) {
        const double* pp = a.pts + ((size_t)req * npts + pl) * SD;
#pragma unroll
        for (int d = 0; d < SD; ++d) xnext[d] = pp[d];
    }
    for (; req < a.nreq; req += stride) {
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
                const CDouble* cp = kcoef + 3 * s;
                const CDouble* up = kucoef + 12 * s;
                asm volatile("" : "+s"(cp), "+s"(up));
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
#pragma unroll
            for (int u = 0; u < NTAB / 2; ++u) {
                const double packed = pack_halves(comp[2 * u], comp[2 * u + 1]);
                if (active_pair) slab[pairbase[u] + kk * 16] = packed;
            }
            if constexpr (NTAB % 2 == 1) {
                if (active) slab[lastbase + kk * 16] = comp[NTAB - 1];
            }
        };

        // D tile nt -> HBM: address = scalar base (request, row block) + 32-bit per-lane byte offset
        char* gbase = reinterpret_cast<char*>(a.out + (size_t)req * reqsize);
        auto store_tile = [&](int nt) {
            const int so = soff[nt];
            unsigned lane_off = (unsigned)(so < 0 ? 0 : so) * 8u;
            // opaque to the optimiser: keeps hipcc from materialising all the 64-bit
            // store addresses at the top of the request (60 VGPRs, spilled)
            asm volatile("" : "+v"(lane_off));
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int mbase = 16 * mt + 4 * jj;  // + (lane >> 4)
                    double* dst = reinterpret_cast<double*>(gbase + (size_t)(mbase * npts) * 8 + lane_off);
                    if (mbase + 3 < ROWS) {
                        if (so >= 0) *dst = acc16[nt][mt][jj];
                    } else if (mbase < ROWS) {
                        if (so >= 0 && mbase + (lane >> 4) < ROWS) *dst = acc16[nt][mt][jj];
                    }
                }
            }
#pragma unroll
            for (int m4 = 0; m4 < M4; ++m4) {
                const int mbase = 16 * MT16 + 4 * m4;
                double* dst = reinterpret_cast<double*>(gbase + (size_t)(mbase * npts) * 8 + lane_off);
                if (mbase + 3 < ROWS) {
                    if (so >= 0) *dst = acc4[nt][m4];
                } else {
                    if (so >= 0 && mbase + (lane >> 4) < ROWS) *dst = acc4[nt][m4];
                }
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
                    // the tile is final after the last K-step: store it while the next tile's MFMAs run
                    if (j == KS - 1 && !(a.debug & 4)) store_tile(nt);
                }
            }
        }

    }
}

}  // namespace fxk
