// Paired K-streamed simplex tabulation kernel (gfx950): TWO requests per wave.
//
// Measured on MI355X (tools/ubench2.hip): fp64 MFMA and fp64/fp32 VALU instructions of
// different waves do not overlap on a SIMD -- a wave issuing v_mfma_f64 stalls the other
// waves' VALU work for the 64 (16x16x4) / 16 (4x4x4) cycles it takes, and the fp64 MFMA
// rate equals the fp64 FMA rate (32 flop/cycle/SIMD).  The specialised kernels are therefore
// bound by MFMA cycles + VALU cycles per request, not by HBM alone, and the recurrence
// (lanes <-> points, 23 of 64 lanes busy for a degree-6 rule) is the part that can shrink:
// here lanes 0..31 carry the points of request 2i and lanes 32..63 those of request 2i+1,
// so every recurrence instruction, coordinate map and LDS store of expansion values serves
// two requests.  The contraction runs over 2*NT column tiles (same MFMA count per request);
// the K-streamed slab aliases a whole-request output image in LDS, the output leaves as full-line
// 16-byte stores with an exact vmcnt on the prefetched points (as in simplex_stream.hpp), and the
// pairs are handed to the waves dynamically (work_queue.hpp).
#pragma once
#include "simplex_stream.hpp"
#include "work_queue.hpp"

#ifndef FX_DBG
// ablation builds only (make dbg-libs): 1 skip recurrence, 2 skip MFMA, 8 skip LDS stores of Phi,
// 64 L2-resident output window, 512 record wave lifetimes
#define FX_DBG 0
#endif
#ifndef FX_PAIR_FULLIMG
#define FX_PAIR_FULLIMG 1  // 1: the LDS image holds a whole request (one write/read-back round per request), 0: half
#endif
#ifndef FX_PAIR_PLAIN_EDGES
#define FX_PAIR_PLAIN_EDGES 1  // plain (write-back) stores for the two store instructions of a request that hold lines shared with its neighbours
#endif
#ifndef FX_PAIR_WAVES
#define FX_PAIR_WAVES 2  // waves per SIMD requested from the register allocator (2*NT accumulator tiles)
#endif

namespace fxk {

// RPW: requests per wave.  2 (the default): two requests of <= 32 points share a wave.  1: one request
// of <= 64 points per wave -- shapes whose accumulators leave no room for a second request (many rows)
// or with more than 32 points; everything else is the same kernel.  FULLIMG: the LDS image holds a
// whole request (one write / read-back round per request) or half of its tables (large requests).
// PIOLA: vector-valued elements (ROWS = ndof * SD, per-request cells): the covariant / contravariant
// Piola map of the request (kind in bits 16-17 of a.debug) is applied to the LDS image in place before it
// is copied out -- lane <-> (dof, point), the SD components of a dof are SD rows of the image.
template <int SD, int N, int ORDER, int ROWS, int NT, int NW, bool UNIFORM, int RPW = 2, bool FULLIMG = (FX_PAIR_FULLIMG != 0),
          bool PIOLA = false>
__global__ __launch_bounds__(64 * NW, NW <= 4 ? 1 : FX_PAIR_WAVES) void tabulate_simplex_pair(const FixedArgs<FixedNC<SD, N>::value> a,
                                                                                double* __restrict__ trash,
                                                                                unsigned int* __restrict__ gqueue) {
    constexpr int NTAB = NTab<SD, ORDER>::value;
    constexpr StepTable<SD, N> TBL{};
    constexpr int NEXP = StepTable<SD, N>::NEXP;
    constexpr int KS = (NEXP + 3) / 4;
    constexpr int MT16 = rows_full16(ROWS);
    constexpr int M4 = rows_blk4(ROWS);
    static_assert(RPW == 1 || RPW == 2, "one or two requests per wave");
    static_assert(!PIOLA || (!UNIFORM && ROWS % SD == 0), "the Piola map needs per-request cells and SD components per dof");
    constexpr int NT2 = RPW * NT;  // column tiles of the unit: [0, NT) request RPW*i, [NT, 2NT) request 2i+1
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#if FX_DBG & 512
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
    // LDS: [A fragments, shared by the workgroup] [per wave: half image of ONE request (the
    // K-step slab of the pair aliases its start) | 64-double dump row for inactive lanes]
    constexpr int NAF = (MT16 + M4) * KS;
    double* afr = lds + WQ_CTL_DOUBLES;  // the work queue's control block comes first
    double* img = afr + NAF * 64 + (size_t)wave * a.lds_doubles;
    double* slab = img;
    const int dump = a.lds_doubles - 64;

    typedef const __attribute__((address_space(4))) double CDouble;
    typedef FixedArgs<FixedNC<SD, N>::value> ArgsT;
    const __attribute__((address_space(4))) char* kargs =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    CDouble* kcoef = (CDouble*)(kargs + __builtin_offsetof(ArgsT, coef));
    CDouble* kucoef = (CDouble*)(kargs + __builtin_offsetof(ArgsT, ucoef));

    const int npts = a.npts;  // <= 32 * (3 - RPW) (host-checked)
    double a0inv[SD][SD];  // inverse of the element's own cell map (Piola: K = A0inv * A_req)
    if constexpr (PIOLA) {
        if constexpr (SD == 2) {
            const double det = a.A0[0] * a.A0[3] - a.A0[1] * a.A0[2];
            a0inv[0][0] = a.A0[3] / det;
            a0inv[0][1] = -a.A0[1] / det;
            a0inv[1][0] = -a.A0[2] / det;
            a0inv[1][1] = a.A0[0] / det;
        } else {
            const double* A = a.A0;
            const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
            const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
            a0inv[0][0] = c00 / det;
            a0inv[0][1] = (A[2] * A[7] - A[1] * A[8]) / det;
            a0inv[0][2] = (A[1] * A[5] - A[2] * A[4]) / det;
            a0inv[1][0] = c01 / det;
            a0inv[1][1] = (A[0] * A[8] - A[2] * A[6]) / det;
            a0inv[1][2] = (A[2] * A[3] - A[0] * A[5]) / det;
            a0inv[2][0] = c02 / det;
            a0inv[2][1] = (A[1] * A[6] - A[0] * A[7]) / det;
            a0inv[2][2] = (A[0] * A[4] - A[1] * A[3]) / det;
        }
    }
    const int table = ROWS * npts;

    for (int i = lane; i < a.lds_doubles; i += 64) img[i] = 0.0;
    for (int i = threadIdx.x; i < NAF * 64; i += 64 * NW) afr[i] = a.afrag[i];
    const long long npairs = (a.nreq + RPW - 1) / RPW;  // units (pairs or single requests)
#ifndef FX_WQ_TAIL
#define FX_WQ_TAIL 0  // 1: no look-ahead in the last two rounds (work_queue.hpp TAIL; measured: mean idle at the end 12.0 -> 10.5-11.5 us, launch times within noise)
#endif
    WorkQueueT<(FX_WQ_TAIL != 0)> wqueue;
    wqueue.init(lds, gqueue, npairs);  // pairs are handed out dynamically, see work_queue.hpp
    __syncthreads();

    const int sub = RPW == 2 ? lane >> 5 : 0;  // which request of the pair this lane's point belongs to
    const int lp = RPW == 2 ? lane & 31 : lane;
    const bool active = lp < npts;
    const int pl = active ? lp : 0;
    // Column order of the contraction: the tables of output half h (h = 0: tables [0, TH),
    // h = 1: tables [TH, NTAB)) occupy the column tiles [h*NT/2, (h+1)*NT/2) of their request, so
    // that no tile straddles the two half images (host-checked: (NT/2)*16 >= TH*npts).  The
    // epilogue is then free of branches and masks, which keeps hipcc's vmcnt bookkeeping exact.
    static_assert(FULLIMG || NT % 2 == 0 || NTAB == 1, "half-aligned column tiles need an even tile count");
    constexpr int TH = FULLIMG ? NTAB : (NTAB + 1) / 2;  // tables per image round
    constexpr int NTH = NTAB > TH ? NT / 2 : NT;  // tiles per half
    int colbase[NTAB];
#pragma unroll
    for (int t = 0; t < NTAB; ++t) {
        const int h = t >= TH ? 1 : 0;
        const int c = (t - h * TH) * npts + pl;  // column inside the half
        colbase[t] = active ? (sub * NT + h * NTH + (c >> 4)) * 64 + (c & 15) : dump + (lane & 15);
    }
    constexpr int NFL = (TH * ROWS * ((16 * NTH) / TH) / 2 + 63) / 64;
    constexpr int NSTORE = RPW * NFL * (NTAB > TH ? 2 : 1);  // vector-memory stores per unit
    (void)NSTORE;
    // image offset (inside its half image) of this lane's column of tile nt, row (lane >> 4);
    // padding columns go to the dump row
    int ioff[NT];
    {
        const float rinv = 1.0f / (float)npts;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int h = nt / NTH;
            const int c = ((nt - h * NTH) << 4) + (lane & 15);
            const int ct = idiv_small(c, rinv);
            const int cp = c - ct * npts;
            const int nth = h == 0 ? TH : NTAB - TH;
            ioff[nt] = (ct < nth) ? ct * table + cp + (lane >> 4) * npts : -1;
        }
    }

    const long long phi = npairs;
    auto claim = [&]() -> long long { return wqueue.claim(); };
    long long pr = claim();
    wqueue.service();
    if (pr >= phi) {
        wqueue.finish();
        return;
    }
    long long pnext = claim();
#if FX_DBG & 512
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = rt_entry;
    const unsigned long long rt_first = __builtin_amdgcn_s_memrealtime();   // init + first claim + first point loads issued
#endif
    // request of this lane's half of the pair (the last pair of an odd batch has no second
    // request: its lanes recompute the first one and the stores are skipped)
    auto lane_req = [&](long long p) -> long long {
        long long r = RPW * p + sub;
        return r < a.nreq ? r : a.nreq - 1;
    };
    double xnext[SD];
    {
        const double* pp = a.pts + ((size_t)lane_req(pr) * npts + pl) * SD;
#pragma unroll
        for (int d = 0; d < SD; ++d) xnext[d] = pp[d];
    }
#pragma unroll
    for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xnext[d]));
    while (true) {
        double X[SD];
        double J[SD][SD];
        double Mreq[RPW][SD][SD];  // (PIOLA only)
        {
            double x[SD];
#pragma unroll
            for (int d = 0; d < SD; ++d) x[d] = xnext[d];
            {
                const long long pn = pnext < phi ? pnext : pr;
                const double* pp = a.pts + ((size_t)lane_req(pn) * npts + pl) * SD;
#pragma unroll
                for (int d = 0; d < SD; ++d) xnext[d] = pp[d];
            }
            double bb[SD];
            if constexpr (!UNIFORM) {
                cell_map<SD>(a.verts + (size_t)lane_req(pr) * (SD + 1) * SD, J, bb);  // per lane: two cells per wave
                if constexpr (PIOLA) {
                    // Piola matrix of each request of the unit, wave uniform (scalar registers): K = A0inv * A_req
                    // read from a lane of that request; covariant J^{-T} = K^T, contravariant J / det J = adj(K)
                    const int kind = (a.debug >> 16) & 3;
#pragma unroll
                    for (int rq = 0; rq < RPW; ++rq) {
                        double Km[SD][SD];
#pragma unroll
                        for (int c = 0; c < SD; ++c)
#pragma unroll
                            for (int d = 0; d < SD; ++d) {
                                double t = 0.0;
#pragma unroll
                                for (int k = 0; k < SD; ++k) {
                                    const int lo = __builtin_amdgcn_readlane(__double2loint(J[k][d]), 32 * rq);
                                    const int hi = __builtin_amdgcn_readlane(__double2hiint(J[k][d]), 32 * rq);
                                    t += a0inv[c][k] * __hiloint2double(hi, lo);
                                }
                                Km[c][d] = t;
                            }
#pragma unroll
                        for (int c = 0; c < SD; ++c)
#pragma unroll
                            for (int e = 0; e < SD; ++e) {
                                double v = Km[e][c];
                                if (kind == 2) {
                                    if constexpr (SD == 2) v = (c == e ? Km[1 - c][1 - e] : -Km[c][e]);
                                    else if constexpr (SD == 3) {
                                        constexpr int nx[3] = {1, 2, 0}, nn[3] = {2, 0, 1};
                                        v = Km[nx[e]][nx[c]] * Km[nn[e]][nn[c]] - Km[nx[e]][nn[c]] * Km[nn[e]][nx[c]];
                                    }
                                }
                                Mreq[rq][c][e] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                                                                  __builtin_amdgcn_readfirstlane(__double2loint(v)));
                            }
                    }
                }
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

        v4d acc16[NT2][MT16 > 0 ? MT16 : 1];
        double acc4[NT2][M4 > 0 ? M4 : 1];

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
            } else if (FX_DBG & 1) {
                if (slot < NEXP) {
                    jet_zero(mem[TBL.dst[slot - 1]]);
                    mem[TBL.dst[slot - 1]].v = X[0];
                }
            } else if (slot < NEXP) {
                const int s = slot - 1;
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
            if (FX_DBG & 8) return;
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

        int elane = lane;  // epilogue copies of the lane- and npts-derived offsets, see below
        int eoff[NT];
        int enpts = npts;
        // accumulators of tile nt (of half nt / NTH) of request `rq` of the pair -> half image;
        // every lane stores: padding columns and rows past the end land in the dump row
        auto image_tile = [&](int rq, int nt) {
            const int so = eoff[nt];
            const bool mine = so >= 0;
            const int an = rq * NT + nt;
            const int dsink = dump + elane;
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int mbase = 16 * mt + 4 * jj;  // + (lane >> 4)
                    if (mbase + 3 < ROWS) {
                        img[mine ? so + mbase * enpts : dsink] = acc16[an][mt][jj];
                    } else if (mbase < ROWS) {
                        img[(mine && mbase + (elane >> 4) < ROWS) ? so + mbase * enpts : dsink] = acc16[an][mt][jj];
                    }
                }
            }
#pragma unroll
            for (int m4 = 0; m4 < M4; ++m4) {
                const int mbase = 16 * MT16 + 4 * m4;
                if (mbase + 3 < ROWS) {
                    img[mine ? so + mbase * enpts : dsink] = acc4[an][m4];
                } else {
                    img[(mine && mbase + (elane >> 4) < ROWS) ? so + mbase * enpts : dsink] = acc4[an][m4];
                }
            }
        };
        v2d carry = v2d{0.0, 0.0};  // (CARRY instances) trailing partial line of half 0, one 16-byte chunk per lane
        auto flush_half = [&](long long req, int half) {
            const int ntab_h = half == 0 ? TH : NTAB - TH;
            const int nch = (ntab_h * ROWS * enpts) >> 1;
            const v2d* s2 = reinterpret_cast<const v2d*>(img);
            v2d* g2 = reinterpret_cast<v2d*>(a.out + (size_t)req * (NTAB * ROWS * enpts) + (size_t)half * TH * ROWS * enpts);
            // in batches of 8 x 16 B per lane (32 VGPRs in flight)
            constexpr int FB = 8;
            // Requests whose size is not a multiple of the 128-byte line (RT2, P4 at 21-23 points) start anywhere in a
            // line.  Every store instruction then covers WHOLE lines: lane i of instruction `it` takes the 16-byte chunk
            // 64 it + i - shift, shift = chunks between the last line boundary and the start of the image (wave uniform,
            // from the address); chunks before the start / past the end are clamped (the same bytes written twice).
            // Non-temporal stores of partial lines cost ~10 % of the launch (tools/ubench6.hip: 173 -> 156 us per 0.84 GB);
            // only the first and the last line of an image stay partial.
            constexpr bool SHIFT = (ROWS * 8 * (FULLIMG ? NTAB : 1)) % 128 != 0;
            constexpr int NFLS = SHIFT ? NFL + 1 : NFL;
            int shift = 0;
            if constexpr (SHIFT) shift = (int)((reinterpret_cast<unsigned long long>(g2) >> 4) & 7ull);
            // Two half images per request (RT2, P4): the boundary between them falls inside a line as well, and that line
            // used to be written twice, partially, a few microseconds apart.  Half 0 now leaves its trailing partial
            // line (tl chunks) out and keeps it in a register per lane (`carry`); half 1's first store instruction
            // starts tl chunks earlier -- the two halves are adjacent in memory -- and writes carry + its own first
            // chunks as ONE whole line: two partial lines per request instead of four (tools/ubench6.hip: two
            // shifted half blocks cost 2.6-4.5 % more than one shifted block of the same bytes).
            constexpr bool CARRY = SHIFT && !FULLIMG && NTAB > TH;
            int tl = 0, lo = 0;
            if constexpr (CARRY) {
                if (half == 0) tl = (shift + nch) & 7;
                else lo = -shift;
            }
            const int last = nch - 1 - tl;
#pragma unroll
            for (int b0 = 0; b0 < NFLS; b0 += FB) {
                v2d buf[FB];
#pragma unroll
                for (int it = b0; it < NFLS && it < b0 + FB; ++it) {
                    const int i = SHIFT ? max(0, min(it * 64 + elane - shift, last)) : min(it * 64 + elane, nch - 1);
                    buf[it - b0] = s2[i];
                    if constexpr (CARRY) {
                        if (it == 0 && half == 1 && elane < shift) buf[0] = carry;
                    }
                }
                if constexpr (CARRY) {
                    // (read before the image is overwritten by the next half's tiles: lane e < tl holds chunk nch - tl + e)
                    if (b0 == 0 && half == 0) carry = s2[min(nch - tl + elane, nch - 1)];
                }
#pragma unroll
                for (int it = b0; it < NFLS && it < b0 + FB; ++it) {
                    const int i = SHIFT ? max(lo, min(it * 64 + elane - shift, last)) : min(it * 64 + elane, nch - 1);
                    // The store instruction that holds the request's first (last) line is a PLAIN store when requests are not line
                    // multiples: that line is shared with the neighbouring request, which another wave writes at another time.  A
                    // non-temporal partial write leaves the L2 at once and the two parts of the line reach memory as separate
                    // masked writes; a plain one stays in the L2 until the other part has arrived (RT2, 25 000 requests:
                    // 167 -> 152.5 us, 100 000: 615 -> 581 us; P4: neutral).  N2 / P3 requests are line multiples: all non-temporal.
                    constexpr bool EDGES = FX_PAIR_PLAIN_EDGES && SHIFT && (NTAB * ROWS * 8) % 128 != 0;
                    if (EDGES && ((it == 0 && (!CARRY || half == 0)) || (it == NFLS - 1 && (!CARRY || half == 1)))) g2[i] = buf[it - b0];
                    else stream_store(&g2[i], buf[it - b0]);
                }
            }
        };

        // fp64 MFMA and VALU share the SIMD's pipe (no overlap to win), so the K-steps run
        // strictly one after the other: produce -> LDS -> fragments -> MFMA.  That keeps the
        // live registers at  accumulators + max(fragments, new members);  the LDS round trip
        // is covered by the other wave of the SIMD.
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) produce(4 * j + kk);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) put(kk, slot_jet(4 * j + kk));
            wave_lds_fence();
            double a16[MT16 > 0 ? MT16 : 1], a4[M4 > 0 ? M4 : 1];
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) a16[mt] = afr[(mt * KS + j) * 64 + lane];
#pragma unroll
            for (int m4 = 0; m4 < M4; ++m4) a4[m4] = afr[((MT16 + m4) * KS + j) * 64 + lane];
#pragma unroll
            for (int rq = 0; rq < RPW; ++rq) {
                double b[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = slab[(rq * NT + nt) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int an = rq * NT + nt;
                    if (FX_DBG & 2) {
                        acc4[an][0] += b[nt] * a4[0];
                        continue;
                    }
                    // (the first K-step starts from the constant 0: no accumulator clearing per pair)
#pragma unroll
                    for (int mt = 0; mt < MT16; ++mt)
                        acc16[an][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a16[mt], b[nt], j == 0 ? v4d{0.0, 0.0, 0.0, 0.0} : acc16[an][mt], 0, 0, 0);
#pragma unroll
                    for (int m4 = 0; m4 < M4; ++m4)
                        acc4[an][m4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[m4], b[nt], j == 0 ? 0.0 : acc4[an][m4], 0, 0, 0);
                }
            }
            wave_lds_fence();  // fragments read before the next K-step overwrites the slab
        }

        // ---------------- D tiles -> half images -> HBM, one request of the pair after the other ----------------
        wqueue.service();  // the previous pair's stores have drained by now, the points were fetched long ago
        const bool second = RPW * pr + 1 < a.nreq;
        // Everything the epilogue derives from the lane number (image offsets per row block,
        // chunk indices and 64-bit addresses of the stores: ~50 VGPRs) is recomputed here per
        // pair: left to itself hipcc hoists it out of the request loop and the K loop then
        // spills.  The opaque copies pin the computations below this point.
        elane = lane;
        asm volatile("" : "+v"(elane));
        enpts = npts;
        asm volatile("" : "+s"(enpts));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            eoff[nt] = ioff[nt];
            asm volatile("" : "+v"(eoff[nt]));
        }
        if (!FX_ABL(a, 4)) {
            wave_lds_fence();
#pragma unroll
            for (int rq = 0; rq < RPW; ++rq) {
#pragma unroll
                for (int half = 0; half < (NTAB > TH ? 2 : 1); ++half) {
#pragma unroll
                    for (int nt = 0; nt < NTH; ++nt) image_tile(rq, half * NTH + nt);
                    wave_lds_fence();
                    if constexpr (PIOLA) {
                        double Mp[SD][SD];
#pragma unroll
                        for (int c = 0; c < SD; ++c)
#pragma unroll
                            for (int e = 0; e < SD; ++e) Mp[c][e] = Mreq[rq][c][e];
                        // blocks of SD rows = one dof of one table of this image
                        const float prinv = 1.0f / (float)enpts;
                        const int lb = idiv_small(elane, prinv), lq = elane - lb * enpts;
                        const int per = idiv_small(64, prinv);
                        const int nblk = (half == 0 ? TH : NTAB - TH) * (ROWS / SD);
                        for (int b0 = 0; b0 < nblk; b0 += per) {
                            const int b = b0 + lb;
                            const bool on = lb < per && b < nblk;
                            double* q = img + (on ? b * SD * enpts + lq : dump + elane);
                            double xin[SD];
#pragma unroll
                            for (int c = 0; c < SD; ++c) xin[c] = on ? q[c * enpts] : 0.0;
#pragma unroll
                            for (int c = 0; c < SD; ++c) {
                                double y = 0.0;
#pragma unroll
                                for (int e = 0; e < SD; ++e) y += Mp[c][e] * xin[e];
                                if (on) q[c * enpts] = y;
                            }
                        }
                        wave_lds_fence();
                    }
                    // the missing second request of an odd batch is written onto the first
                    // (same values), which keeps the store count per pair constant
                    long long oreq = rq == 0 || second ? RPW * pr + rq : RPW * pr;
                    if (FX_DBG & 64) oreq &= 1023;  // ablation: L2-resident output window
                    flush_half(oreq, half);
                    wave_lds_fence();
                }
            }
            // first use of the prefetched points, in the SAME block as the stores: hipcc places
            // the exact s_waitcnt vmcnt(NSTORE) here (after a control-flow join it gives up: vmcnt(0))
#pragma unroll
            for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xnext[d]));
        } else {
#pragma unroll
            for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xnext[d]));
        }
        if (pnext >= phi) break;
        pr = pnext;
        pnext = claim();
    }
    wqueue.service();
    wqueue.finish();
#if FX_DBG & 512
    if (lane == 0) {  // ablation build: lifetime of every wave (shader cycles, 100 MHz ticks)
        const long long gw = (long long)blockIdx.x * NW + wave;
        if (gw < 3000) {
            const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
            trash[2048 + 2 * gw] = (double)(__builtin_readcyclecounter() - clk0);
            trash[2049 + 2 * gw] = (double)(rt1 - rt0);
            trash[2048 + 6000 + 2 * gw] = (double)(rt0 & 0xffffffffffffull);   // absolute start / end (100 MHz ticks)
            trash[2049 + 6000 + 2 * gw] = (double)(rt1 & 0xffffffffffffull);
            trash[2048 + 12000 + gw] = (double)(rt_first - rt0);
        }
    }
#endif
}

}  // namespace fxk
