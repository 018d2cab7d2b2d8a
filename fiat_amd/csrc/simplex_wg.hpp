// Request-per-workgroup simplex tabulation kernel (gfx950): rules of more points than one wave's column budget.
//
// The same matrix product as simplex_stacked.hpp,
//     out[req][(table, row)][point] = A_stack[(table, row)][k] * Phi_k(point)     (FIAT/polynomial_set.py:68-72 with the
//     derivative tables as rows of one matrix, FIAT/expansions.py:438-446),
// laid out for requests of 49..128 points (the 74- and 122-point rules of degree-5 / 6 tetrahedra, FIAT/xg_quad_data.py).
// The stacked kernel serves those as point chunks: a unit is 32 or 48 points of one request, its row tiles leave as 8-byte
// stores of 256- / 384-byte row segments whose first and last lines are shared with units that run elsewhere at another time,
// and every chunk re-reads all of A_stack from L2 (ablation without MFMAs: 395 of 478 us -- the store pattern bounds it).
// Here ONE WORKGROUP owns a request:
//   * production: thread <-> point, the order-0 recurrence once per point, every member to a [4 KS][16 CT] slab in LDS
//     (degree 6, 122 points: 84 x 128 doubles = 86 KB);
//   * sweep: the four waves (one per SIMD) take the row tiles of A_stack round robin.  A wave holds the A fragments of its
//     tile in registers (streamed from L2 one tile ahead: A_stack is read ONCE per request instead of once per chunk), reads
//     the B fragments of K-step ks + 1 from the slab while the CT MFMAs of K-step ks run, and keeps the 16 x 16 CT result in
//     accumulators;
//   * flush: a finished row tile is 16 x npts CONTIGUOUS doubles of the output (stacked rows are contiguous,
//     [ntab][rows][npts] is [R][npts]): accumulators -> per-wave LDS image -> 16-byte non-temporal stores of whole 128-byte
//     lines, software-pipelined under the MFMAs of the wave's next tile (two accumulator sets, unrolled by two), LDS and
//     memory instructions spread over the K-steps exactly as in simplex_stacked.hpp.
// Requests go to the persistent workgroups round robin (request b, b + G, ... to workgroup b of G): the kernel is bound by
// the fp64 pipe, which every CU runs at the same rate -- unlike the store-bound kernels, whose XCDs drift apart and need the
// dynamic queue of work_queue.hpp -- and a device-scope ticket per request cost its full latency (hipcc waits for the
// returned value at once).  The next request's points are loaded during the sweep.  Every wave of a workgroup leaves the
// request loop at the same id >= nreq: the grid always drains.
#pragma once
#include "coop_kernel.hpp"  // wg_lds_barrier
#include "simplex_stacked.hpp"

#ifndef FX_WG_ABL
#define FX_WG_ABL 0  // measurement builds: 1 no recurrence, 2 no MFMAs, 4 no flush (image reads + output stores), 8 no image writes
#endif

#ifndef FX_WG_DBG
#define FX_WG_DBG 0  // 1: every global access range-checked against the lim_* fields, offenders redirected to / reported in `trash`
#endif

namespace fxk {

#if FX_WG_DBG
// site: 1 pts, 3 afrag, 4 out; trash[4096 + 4 site ..] = {count, first offending index, limit, request}
__device__ __forceinline__ long long wg_dbg_check(long long idx, long long lim, int site, long long req, double* trash) {
    if (idx >= 0 && idx < lim) return idx;
    double* t = trash + 4096 + 4 * site;
    if (atomicAdd(reinterpret_cast<unsigned long long*>(t), 1ULL) == 0ULL) {
        t[1] = (double)idx;
        t[2] = (double)lim;
        t[3] = (double)req;
    }
    return -1;
}
#endif

#ifndef FX_WG_TIME
#define FX_WG_TIME 0  // measurement build: workgroup 0 sums the clock ticks of its phases (production, barrier, first operands, sweep) per wave
#endif
#if FX_WG_TIME
#define WG_TICK(i)                                          \
    do {                                                    \
        const long long t_ = __builtin_readcyclecounter(); \
        tacc[i] += t_ - tprev;                              \
        tprev = t_;                                         \
    } while (0)
#else
#define WG_TICK(i)
#endif

template <bool FIRST, class A, class B> __device__ __forceinline__ auto& choose_ref(A& x, B& y) {
    if constexpr (FIRST) return x;
    else return y;
}

constexpr int WG_NW = 4;  // one wave per SIMD
// LDS doubles: control block, recurrence coefficients [3 (4 KS - 1)] (rounded up), expansion values [4 KS][16 CT], per-wave
// row-tile images [16][<= 16 CT] + dump row
constexpr int wg_image_doubles(int CT) { return 16 * 16 * CT + 64; }
constexpr int wg_coef_doubles(int KS) { return (3 * 4 * KS + 7) & ~7; }
constexpr int WG_KBUF = 128;  // (MIX instances) K = A0^-1 A_req of the group's <= 12 requests
constexpr int wg_lds_doubles(int CT, int KS) { return WQ_CTL_DOUBLES + WG_KBUF + wg_coef_doubles(KS) + 4 * KS * 16 * CT + WG_NW * wg_image_doubles(CT); }

// Production is split over the waves by MEMBERS: the chains of the last recurrence level (two thirds of the members of a
// tetrahedron's expansion set) are independent of each other, so the NSUB wave sets each run the lower levels (needed as
// chain heads) and their share of the last level's chains -- degree 6: 27 + 28 steps a wave instead of 83.
template <int SD, int N, int NSUB> struct StepSubsets {
    static constexpr int NS = StepTable<SD, N>::NSTEPS;
    int owner[NS] = {};  // -1: every subset computes it (subset 0 stores it), else the one subset that computes and stores it
    int load[NSUB] = {};
    constexpr StepSubsets() {
        StepTable<SD, N> T{};
        int cur_owner = 0;
        // (subset 0 also stores the lower levels, and computes all of them -- the others only the chain heads they need: it
        // starts with a handicap of half the lower levels' steps.  FX_WG_TIME, degree 5: 144 against 85 fp64 instructions, 2560
        // against 1620 clocks before)
        if (SD >= 2 && NSUB > 1) {
            int shared = 0;
            for (int s = 0; s < T.count; ++s) shared += T.codim[s] < SD - 1 ? 1 : 0;
            load[0] = shared / 2;
        }
        for (int s = 0; s < T.count; ++s) {
            if (SD < 2 || T.codim[s] < SD - 1) {
                owner[s] = -1;
                continue;
            }
            if (T.prv[s] < 0) {  // first step of a chain: to the least loaded subset
                int best = 0;
                for (int q = 1; q < NSUB; ++q)
                    if (load[q] < load[best]) best = q;
                cur_owner = best;
            }
            owner[s] = cur_owner;
            load[cur_owner]++;
        }
    }
};

// the steps subset SUB runs, in table order (its recurrence coefficients are fetched ahead in batches)
template <int SD, int N, int NSUB, int SUB> struct SubsetSteps {
    static constexpr int NS = StepTable<SD, N>::NSTEPS;
    int list[NS] = {};
    int count = 0;
    constexpr SubsetSteps() {
        StepSubsets<SD, N, NSUB> S{};
        StepTable<SD, N> T{};
        int raw[NS] = {};
        int nraw = 0;
        for (int s = 0; s < StepTable<SD, N>::NEXP - 1; ++s)
            if (S.owner[s] < 0 || S.owner[s] == SUB) raw[nraw++] = s;
        // Within a recurrence level the chains are independent of each other and a step needs the two steps before it in its
        // chain: step by step down ONE chain, every fp64 instruction waits for the one before (a wave has its SIMD to itself --
        // ~16 clocks an instruction instead of 4, FX_WG_TIME).  So the chains of a level advance together: first steps of all
        // chains, second steps, ...
        int pos = 0;
        while (pos < nraw) {
            const int cd = T.codim[raw[pos]];
            int end = pos;
            while (end < nraw && T.codim[raw[end]] == cd) ++end;
            int start[NS] = {}, len[NS] = {};
            int nch = 0;
            for (int i = pos; i < end; ++i) {
                if (T.prv[raw[i]] < 0 || nch == 0) start[nch++] = i;
                len[nch - 1]++;
            }
            for (int d = 0; d < end - pos; ++d)
                for (int c = 0; c < nch; ++c)
                    if (d < len[c]) list[count++] = raw[start[c] + d];
            pos = end;
        }
    }
};

// PC: waves that share a row tile (each takes CT / PC of its column tiles); PR = 4 / PC row tiles are in work at a time.
//   PC 1: a wave owns whole row tiles, waves never wait for each other inside a request -- but a request's RT row tiles go
//         round robin over four waves (values only, degree 6: 6 tiles -> 2, 2, 1, 1: a third of the MFMA slots idle);
//   PC 4: all four waves on one row tile, two column tiles each (the MIX instances of degree >= 5 tetrahedra at eight column
//         tiles: the accumulators of 1 + SD tables x 4 column tiles x two sets do not fit the registers);
//   PC 2: two waves per row tile, a step of the workgroup finishes two row tiles (6 tiles -> 3 steps, no idle slots; 21 -> 11
//         steps, 53 -> 27).  The two halves meet in a row-tile image shared by the pair (double-buffered: one workgroup
//         barrier per step), and each wave then writes every other 1-KB piece of the finished tile.
// MIX 1 (per-request cells, values + gradients): the row tiles come DOF-MAJOR -- the 1 + SD tables of 16 dofs one after the
//         other (fragment buffer of the stacked kernel's chain-rule instances) -- a wave keeps the accumulators of all tables of
//         its dof tile, applies the chain rule d/dx_d = sum_c K[c][d] d/dX_c to them (lane-local in the MFMA result layout; K of
//         the column's request from LDS) and flushes table by table under the next dof tile's MFMAs: one pass instead of
//         kernel + table_mix_kernel (FIAT/expansions.py:411-447 through Jinv).
// FAST (PC 1, at most four row tiles -- values-only requests of up to 64 rows): every wave has ONE row tile a group; its own
//         instances, so that the accumulators that cross the production phase are there only.  With 16-byte flush pieces they are
//         compiled for 256 registers and two workgroups share a CU (one's recurrence phase under the other's MFMAs: degree-5
//         tetrahedra at 74 points 320 -> 298 us; the 8-byte twins spill in the sweep at 256 registers, 394 -> 850 us, and stay
//         at one workgroup a CU)
template <int SD, int N, int CT, bool ODD, int PC, int MIX = 0, bool FAST = false>
__global__ __launch_bounds__(64 * WG_NW, FAST && !ODD ? 2 : 1) void tabulate_simplex_wg(const StackedArgs<FixedNC<SD, N>::value> a, double* __restrict__ trash,
                                                                   unsigned int* __restrict__ gctr) {
    constexpr StepTable<SD, N> TBL{};
    constexpr int NEXP = StepTable<SD, N>::NEXP;
    constexpr int KS = (NEXP + 3) / 4;
    constexpr int LDC = 16 * CT;  // columns of the slab
    constexpr int DUMP = 16 * LDC;
    using FlushT = typename std::conditional<ODD, double, v2d>::type;
    constexpr int EPP = ODD ? 1 : 2;                        // doubles per flush piece
    constexpr int NRD = (16 * LDC / EPP + 63) / 64;         // image reads = output stores per row tile
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* kbuf = lds + WQ_CTL_DOUBLES;          // (MIX) [G][SD][SD]
    double* cof = kbuf + WG_KBUF;                 // [nsteps][3] = A, B, C of the recurrence steps
    static_assert(MIX == 0 || (MIX == 1 && PC >= 2 && SD >= 2), "chain rule on the accumulators: order 1, two or four waves per row tile");
    static_assert(!FAST || (PC == 1 && MIX == 0 && !(FX_WG_ABL & 12)), "one row tile per wave: one wave per row tile, no chain rule");
    double* phi = cof + wg_coef_doubles(KS);
    static_assert((PC == 1 || PC == 2 || PC == 4) && CT % PC == 0, "column tiles split evenly over the waves of a row tile");
    constexpr int PR = WG_NW / PC, CTW = CT / PC;
    const int pr = wave / PC, pc = wave % PC;     // (wave-uniform) row group, column group
    // four row-tile images: PC 1 one per wave; PC 2 two per pair of waves (step parity)
    double* const imgs = phi + 4 * KS * LDC;
    auto image_of = [&](int parity) { return imgs + (size_t)(PC == 1 ? wave : pr * 2 + parity) * wg_image_doubles(CT); };

    typedef const __attribute__((address_space(4))) double CDouble;
    typedef StackedArgs<FixedNC<SD, N>::value> ArgsT;
    const __attribute__((address_space(4))) char* kargs =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    CDouble* kcoef = (CDouble*)(kargs + __builtin_offsetof(ArgsT, coef));

    const int npts = a.npts;
    const int kk = lane >> 4, col = lane & 15;
    const long long nreq = a.nreq;

    // A slab holds G consecutive requests of npts points each (G npts <= 16 CT; G = 1 for rules of more than 64 points):
    // column j <-> (request j / npts of the group, point j % npts).  The row-tile image is [request][16 rows][npts], so that a
    // request's part of a row tile -- 16 x npts contiguous doubles of the output -- is contiguous in the image too.
    const int G = a.gslab;
    const int cols = G * npts;
    const int BLKD = 16 * npts;                 // doubles of one request in a row-tile image
    const float rnpts = 1.0f / (float)npts;
    // image offsets of this lane's accumulator elements: element jj of column tile c0 + c is row 4 jj + kk, column 16 (c0 + c) +
    // col; padding columns go to the dump row.  (This wave's column tiles are c0 .. c0 + CTW - 1.)
    const int c0 = pc * CTW;
    int ioff[CTW][4];
    int kofs[MIX ? CTW : 1];  // (MIX) where K of the column's request sits in kbuf
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int j = 16 * (c0 + c) + col;
        const int g = idiv_small(j, rnpts);
        if constexpr (MIX != 0) kofs[c] = j < cols ? g * SD * SD : 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ioff[c][jj] = j < cols ? g * BLKD + (4 * jj + kk) * npts + (j - g * npts) : DUMP + lane;
    }
    // flush pieces of this wave (EPP doubles each, 64 an instruction; instruction q of the wave is r = PC q + pc of the tile):
    // piece u = 64 r + lane lies in request u / BLK of the group at position u % BLK (BLK = 16 npts / EPP pieces a request);
    // pieces past the group's last request repeat its last piece.  pu_l: index into the image, pu_o: byte offset from the
    // group's first request in the output (requests are R npts doubles apart) -- loop invariants of the launch, except in a
    // last group of fewer than G requests (rebuilt there: the missing requests repeat the group's last one).
    constexpr int NRW = (NRD + PC - 1) / PC;
    const int BLK = BLKD / EPP;
    const float rblk = 1.0f / (float)BLK;
    const unsigned RN8 = (unsigned)a.R * (unsigned)npts * 8u;
    int pu_l[NRW];
    unsigned pu_o[NRW];
    auto build_pieces = [&](int gmax, int nrows) {  // requests 0..gmax of the group exist; nrows rows in the tile
        const int blkn = nrows * npts / EPP;
#pragma unroll
        for (int q = 0; q < NRW; ++q) {
            const int u = min((PC * q + pc) * 64 + lane, G * BLK - 1);
            const int g = idiv_small(u, rblk);
            const int w = min(u - g * BLK, blkn - 1);
            pu_l[q] = g * BLK + w;
            pu_o[q] = (unsigned)min(g, gmax) * RN8 + (unsigned)w * (unsigned)(8 * EPP);
        }
    };
    build_pieces(G - 1, 16);

    // production: WPP waves cover the LDC columns, NSUB such wave sets split the members (StepSubsets)
    constexpr int WPP = LDC > 64 ? 2 : 1, NSUB = WG_NW / WPP;
    constexpr StepSubsets<SD, N, NSUB> SUBS{};
    const int pcol = (wave % WPP) * 64 + lane;   // column this thread produces
    const int psub = wave / WPP;                 // (wave-uniform) its member subset
    // the column this thread produces: (request pg of the group, point pp); threads past the group recompute its last column
    const int pj = min(pcol, cols - 1);
    const int pg = idiv_small(pj, rnpts);
    const int pp_ = pj - pg * npts;
    const long long ngroups = (nreq + G - 1) / G;
    auto request_of = [&](long long grp) {  // this thread's request in group grp (groups / requests past the batch: the last one)
        const long long g0 = (grp < ngroups ? grp : ngroups - 1) * G;
        return g0 + pg < nreq ? g0 + pg : nreq - 1;
    };
    auto load_points = [&](long long grp, double (&x)[SD]) {
        const long long rr = request_of(grp);
        const double* ppt = a.pts + ((size_t)rr * npts + pp_) * SD;
#if FX_WG_DBG
        if (wg_dbg_check(((long long)rr * npts + pp_) * SD, a.lim_pts - SD + 1, 1, grp, trash) < 0) ppt = trash;
#endif
#pragma unroll
        for (int d = 0; d < SD; ++d) x[d] = ppt[d];   // (threads past the slab read its last column's point: no branch, exact waits)
    };
    // once per launch: the recurrence coefficients from the kernel arguments to LDS (broadcast reads in the recurrence: in
    // order with the slab writes, so the waits are exact counts -- scalar loads return out of order and made every step wait
    // for lgkmcnt(0), i.e. for its own load AND the previous member's LDS write: 83 serialised round trips per request), and
    // the zero rows that pad the slab to whole K-steps
    for (int i = tid; i < 3 * (NEXP - 1); i += 64 * WG_NW) cof[i] = kcoef[i];
    for (int i = tid; i < (4 * KS - NEXP) * LDC; i += 64 * WG_NW) phi[NEXP * LDC + i] = 0.0;
    wg_lds_barrier();

    long long cur = blockIdx.x, nxt = (long long)blockIdx.x + gridDim.x;
    double xcur[SD], xnext[SD];
    load_points(cur, xcur);
    // (complete before the loop: pending at the loop header, they made the loop's first use of xcur a wait for everything but the
    // loads just issued -- on the back edge that is a wait for the previous group's output stores)
#pragma unroll
    for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xcur[d]));

    // (PC 1) the wave's last full row tile of the previous group, not yet written: accumulators, output base and first row
    v4d accP[PC == 1 && !FAST ? CTW : 1];
    bool pend = false;
    double* pend_base = FAST ? trash : a.out;
    int pend_frow = 0, pend_lim = 0;
    // (PC 1) ONE row tile per wave and group -- values-only requests of up to 64 rows: the two accumulator sets alternate between
    // groups and a group's tile leaves from where it was computed, under the next group's MFMAs in the other set (the copy to
    // accP was 40 v_accvgpr_read + 40 v_cndmask behind the last MFMA, ~600 of the 11 700 clocks of a degree-5 request); the wave's
    // A fragments are the same for every group and stay in registers
    v4d accA0[FAST ? CTW : 1], accB0[FAST ? CTW : 1];
    bool flip = false;   // the pending tile is in set A
    if constexpr (FAST) {
#pragma unroll
        for (int c = 0; c < CTW; ++c) accB0[c] = v4d{0.0, 0.0, 0.0, 0.0};
    }
    double faF[FAST ? KS : 1];
    constexpr bool fast = FAST;
    if constexpr (FAST) {
        const double* ap0 = a.afrag + (size_t)min(pr, a.RT) * KS * 64 + lane;   // (waves past the last row tile: the zero tile)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) faF[ks] = ap0[ks * 64];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(faF[ks]));   // (complete before the loop, like the first points)
    }
#if FX_WG_TIME
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#endif
    while (cur < ngroups) {
        WG_TICK(4);
        // next group's points, then the A fragments of this wave's first row tile: in flight during the production phase (the
        // points first: the wait for the fragments after the slab barrier is then a wait for everything, an exact count)
        load_points(nxt, xnext);
        double fa0[KS], fa1[KS];
        if constexpr (FAST) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fa0[ks] = faF[ks];
        } else {
            const double* ap0 = a.afrag + (size_t)(MIX ? min(pr, (a.R / (1 + SD) + 15) / 16) * (1 + SD) : min(pr, a.RT)) * KS * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fa0[ks] = ap0[ks * 64];
        }
        // ---------------- expansion values of the group's requests -> LDS slab ----------------
        if (pcol < LDC) {
            double X[SD];
            if (a.verts) {  // (uniform branch) physical point -> default simplex through the cell of the column's request
                double J[SD][SD], bb[SD];
                cell_map<SD>(a.verts + (size_t)request_of(cur) * (SD + 1) * SD, J, bb);
#pragma unroll
                for (int i = 0; i < SD; ++i) {
                    double t = bb[i];
#pragma unroll
                    for (int d = 0; d < SD; ++d) t += J[i][d] * xcur[d];
                    X[i] = t;
                }
            } else {
#pragma unroll
                for (int i = 0; i < SD; ++i) {
                    double t = a.b0[i];
#pragma unroll
                    for (int d = 0; d < SD; ++d) t += a.A0[i * SD + d] * xcur[d];
                    X[i] = t;
                }
            }
            // The coefficients of a step are broadcast LDS reads, and the compiler may not move them across the slab writes of
            // the steps before (same LDS array): read next to their use, every step waited a whole LDS round trip -- ~108 clocks a
            // step, 3800 of the 11 700 clocks of a values-only degree-5 request (FX_WG_TIME).  So they are fetched a batch of CB
            // steps ahead: the reads of batch b + 1 stand in front of the arithmetic and the writes of batch b.
            auto produce_subset = [&](auto sub_c) {
                constexpr int SUB = decltype(sub_c)::value;
                constexpr SubsetSteps<SD, N, NSUB, SUB> SL{};
                constexpr int CB = (MIX || FAST) ? 4 : 8, NBATCH = (SL.count + CB - 1) / CB;
                double mem[NEXP];
                double ufa = 0.0, ufb = 0.0, ufc = 0.0;
                int fcodim = -1;
                mem[0] = a.phi0;
                if constexpr (SUB == 0) phi[pcol] = mem[0];
                double cfe[3 * CB], cfo[3 * CB];   // batches of even / odd index
                auto fetch = [&](int b, double (&cf)[3 * CB]) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < CB; ++i) {
                        if (b * CB + i >= SL.count) continue;
                        const int s = SL.list[b * CB + i];
#pragma unroll
                        for (int e = 0; e < 3; ++e) cf[3 * i + e] = cof[3 * s + e];
                    }
                };
                auto run = [&](int b, const double (&cf)[3 * CB]) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < CB; ++i) {
                        if (b * CB + i >= SL.count) continue;
                        if ((FX_WG_ABL & 1) && nreq > 8) continue;
                        const int s = SL.list[b * CB + i];
                        const double cA = cf[3 * i], cB = cf[3 * i + 1], cC = cf[3 * i + 2];
                        if (TBL.codim[s] != fcodim) {
                            fcodim = TBL.codim[s];
                            point_factors<SD>(fcodim, X, ufa, ufb, ufc);
                        }
                        const double f = cA * ufa - cB * ufb;
                        double v = mem[TBL.cur[s]] * f;
                        if (TBL.prv[s] >= 0) v -= cC * ufc * mem[TBL.prv[s]];
                        mem[TBL.dst[s]] = v;
                        if (SUBS.owner[s] == SUB || (SUBS.owner[s] < 0 && SUB == 0)) phi[(s + 1) * LDC + pcol] = v;
                    }
                };
                fetch(0, cfe);
#pragma unroll
                for (int b = 0; b < NBATCH; ++b) {
                    if (b & 1) {
                        if (b + 1 < NBATCH) fetch(b + 1, cfe);
                        run(b, cfo);
                    } else {
                        if (b + 1 < NBATCH) fetch(b + 1, cfo);
                        run(b, cfe);
                    }
                }
            };
            WG_TICK(5);
            if constexpr (NSUB == 2) {
                if (psub == 0) produce_subset(std::integral_constant<int, 0>{});
                else produce_subset(std::integral_constant<int, 1>{});
            } else {
                if (psub == 0) produce_subset(std::integral_constant<int, 0>{});
                else if (psub == 1) produce_subset(std::integral_constant<int, 1>{});
                else if (psub == 2) produce_subset(std::integral_constant<int, 2>{});
                else produce_subset(std::integral_constant<int, 3>{});
            }
        }
        if constexpr (MIX != 0) {  // K = A0^-1 A_req of the group's requests (requests past the batch: its last one)
            if (tid < G) {
                const long long g0 = cur * G;
                const long long rq = g0 + tid < nreq ? g0 + tid : nreq - 1;
                double J[SD][SD], bb[SD];
                cell_map<SD>(a.verts + (size_t)rq * (SD + 1) * SD, J, bb);
#pragma unroll
                for (int i = 0; i < SD; ++i)
#pragma unroll
                    for (int d = 0; d < SD; ++d) {
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < SD; ++k) t += a.A0inv[i * SD + k] * J[k][d];
                        kbuf[tid * SD * SD + i * SD + d] = t;
                    }
            }
        }
        WG_TICK(0);
        wg_lds_barrier();  // slab complete
        WG_TICK(1);
        // (the next group's points have had the production phase to arrive; waiting here keeps the wait at the end of the
        // iteration from counting the sweep's stores)
#pragma unroll
        for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xnext[d]));

        // ---------------- sweep: row tiles pr, pr + PR, ... of A_stack, column tiles c0 .. c0 + CTW - 1 ----------------
        const int RT = a.RT;
        // steps of this wave: PC 1 its own row tiles; PC 2 the same count for every wave (the barriers must match; a pair
        // whose last tile does not exist multiplies the zero tile and skips the image)
        const int nsteps = PC == 1 ? (RT > pr ? (RT - pr + PR - 1) / PR : 0) : (RT + PR - 1) / PR;
        const int last_rows = a.R - 16 * (RT - 1);
        double* const obase = a.out + (size_t)cur * G * a.R * npts;   // the group's first request
        const int gmax = (int)min((long long)G - 1, nreq - 1 - cur * G);
        if (gmax < G - 1) build_pieces(gmax, 16);              // (the batch's last group, when short)

        auto image_put = [&](double* img, const v4d (&acc)[CTW], int nrows) {  // the wave's last tile: may have fewer than 16 rows
#pragma unroll
            for (int c = 0; c < CTW; ++c)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) img[4 * jj + kk < nrows ? ioff[c][jj] : DUMP + lane] = acc[c][jj];
        };
        auto image_get = [&](const double* img, FlushT& f, int q) { f = reinterpret_cast<const FlushT*>(img)[pu_l[q]]; };
        auto image_out = [&](const FlushT& f, int q, int frow, double* fbase) {  // frow: first row (of a request's stacked rows) of the tile
            // (wave-uniform tile base + 32-bit lane offset: scalar-base addressing, no 64-bit lane addresses to hoist and spill)
            char* tb = reinterpret_cast<char*>(fbase + (size_t)frow * npts);
            FlushT* g2 = reinterpret_cast<FlushT*>(tb + pu_o[q]);
#if FX_WG_DBG
            {
                const long long idx = (reinterpret_cast<double*>(g2) - a.out);
                if (wg_dbg_check(idx, a.lim_out - (EPP - 1), 4, cur, trash) < 0) g2 = reinterpret_cast<FlushT*>(trash);
            }
#endif
            if constexpr (ODD) *g2 = f;  // 8-byte pieces, lines shared with the neighbours: plain stores
            else stream_store(g2, f);
        };
        auto image_out1 = [&](const FlushT& f, int q, int frow, double* fbase, int plim) {  // (PC 1: piece min(u, plim) of the tile)
            char* tb = reinterpret_cast<char*>(fbase + (size_t)frow * npts);
            FlushT* g2 = reinterpret_cast<FlushT*>(tb + (unsigned)min(pu_l[q], plim) * (unsigned)(8 * EPP));
#if FX_WG_DBG
            {
                const long long idx = (reinterpret_cast<double*>(g2) - a.out);
                if (wg_dbg_check(idx, a.lim_out - (EPP - 1), 4, cur, trash) < 0) g2 = reinterpret_cast<FlushT*>(trash);
            }
#endif
            if constexpr (ODD) *g2 = f;
            else stream_store(g2, f);
        };
        // B fragments of K-step ks: members 4 ks + kk at this lane's column of the wave's column tiles
        auto load_b = [&](double (&b)[CTW], int ks) {
#pragma unroll
            for (int c = 0; c < CTW; ++c) b[c] = phi[(4 * ks + kk) * LDC + 16 * (c0 + c) + col];
        };
        auto load_a = [&](double (&af)[KS], int tile, int k0, int k1) {
            const int t = min(tile, RT);  // (the fragment buffer ends with a zero tile)
            const double* ap = a.afrag + (size_t)t * KS * 64 + lane;
#if FX_WG_DBG
            if (wg_dbg_check((long long)t * KS * 64 + lane + (KS - 1) * 64, a.lim_afrag, 3, cur, trash) < 0) ap = trash;
#endif
#pragma unroll
            for (int ks = k0; ks < k1 && ks < KS; ++ks) af[ks] = ap[ks * 64];
        };
        // One pipeline stage = one step: MFMAs of tile `tile` (fragments af) into acc; meanwhile the previous step's tile (a
        // full one, in the LDS image `imgr`) goes out and the fragments of the next step's tile come in.  The memory
        // instructions are spread over the K-steps (a wave issues in order: a burst between two MFMAs idles the matrix pipe):
        // fragment loads in the first third -- older than every output store of the stage, so the wait for them is an exact
        // vmcnt(#stores) -- then the image leaves in batches of PB instructions, read in one K-step and stored in the next.  At
        // the end the accumulators go to the image `imgw` (the only LDS traffic not under MFMAs: 4 CTW ds_write_b64).
        constexpr int T3 = KS / 3;
        constexpr int LPK = (KS + T3 - 1) / T3;
        constexpr int NB = KS - T3 - 1;                 // batches: read at K-step T3 + j, stored at T3 + j + 1
        constexpr int PB = (NRW + NB - 1) / NB;
        static_assert(T3 >= 1 && NB >= 1, "at least one K-step for the fragment loads and two for the flush");
        // Inside a K-step the memory instructions are dealt out over its CTW MFMAs (slice c in front of MFMA c): a wave issues
        // in order, and what stands between two MFMAs issues while the first one occupies the pipe for 64 cycles -- a block of
        // memory instructions behind the last MFMA of a K-step has only that one MFMA to hide under.
        // Two accumulator sets: the finished tile of step s - 1 goes to its LDS image during the first K-steps of step s
        // (PC 2: then the pair's barrier), and leaves for HBM during the rest of step s.
        constexpr int NPUT = 4 * CTW;                       // image writes of a tile
        constexpr int PPS = (NPUT + T3 * CTW - 1) / (T3 * CTW) > 2 ? (NPUT + T3 * CTW - 1) / (T3 * CTW) : 2;  // ... a slice (two; more when KS is small)
        constexpr int KPUT = (NPUT + PPS * CTW - 1) / (PPS * CTW);  // K-steps they take
        static_assert(KPUT <= T3, "the image is complete before its first read");
        double b0[CTW], b1[CTW];
        // (ntile: fragment tile to prefetch, clamped to the zero tile `nzero` that ends the buffer; frow: first output row of the
        // tile in `prev`)
        // (plim, PC 1 only -- one request per slab, so piece u of a tile is at u in the image and 8 EPP u bytes into the output:
        // the last piece of the tile in `prev`, which may have fewer than 16 rows)
        auto stage = [&](int ntile, int nzero, int frow, v4d (&acc)[CTW], const v4d (&prev)[CTW], double* img, const double (&af)[KS], double (&an)[KS],
                         bool flush, double* fbase, int plim = 0) __attribute__((always_inline)) {
            FlushT fb[2][PB];
            const double* ap = a.afrag + (size_t)min(ntile, nzero) * KS * 64 + lane;
#pragma unroll
            for (int c = 0; c < CTW; ++c) acc[c] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int kn = ks + 1 < KS ? ks + 1 : 0;  // (K-step 0 of the next stage is loaded by the last)
#pragma unroll
                for (int c = 0; c < CTW; ++c) {
                    // ---- memory slice c ----
                    // B fragment c of the next K-step (b0 holds even K-steps, b1 odd ones)
                    if (ks & 1) b0[c] = phi[(4 * kn + kk) * LDC + 16 * (c0 + c) + col];
                    else b1[c] = phi[(4 * kn + kk) * LDC + 16 * (c0 + c) + col];
                    // A fragments of the next step's tile: first third of the stage (older than every output store of the stage)
                    if (!FAST && ks < T3) {   // (FAST: the wave's one tile of fragments stays in registers)
#pragma unroll
                        for (int l = ks * LPK; l < (ks + 1) * LPK && l < KS; ++l)
                            if (l % CTW == c) an[l] = ap[l * 64];
                    }
                    if (flush && !((FX_WG_ABL & 4) && nreq > 8)) {
                        // the previous step's accumulators -> image
                        if (ks < KPUT && !((FX_WG_ABL & 8) && nreq > 8)) {
#pragma unroll
                            for (int w = 0; w < PPS; ++w) {
                                const int p = (ks * CTW + c) * PPS + w;  // image write p: column tile p / 4, element p % 4
                                if (p < NPUT) {
                                    img[ioff[p >> 2][p & 3]] = prev[p >> 2][p & 3];
                                }
                            }
                        }
                        // ... and out: batch j read at K-step T3 + j, stored at T3 + j + 1
                        if (ks >= T3) {
                            const int j = ks - T3;
                            if (j >= 1) {
#pragma unroll
                                for (int q = 0; q < PB; ++q)
                                    if (q % CTW == c && (j - 1) * PB + q < NRW) {
                                        if constexpr (PC == 1) image_out1(fb[(j - 1) & 1][q], (j - 1) * PB + q, frow, fbase, plim);
                                        else image_out(fb[(j - 1) & 1][q], (j - 1) * PB + q, frow, fbase);
                                    }
                            }
                            if (j < NB) {
#pragma unroll
                                for (int q = 0; q < PB; ++q)
                                    if ((q + CTW / 2) % CTW == c && j * PB + q < NRW) {
                                        if constexpr (PC == 1) fb[j & 1][q] = reinterpret_cast<const FlushT*>(img)[min(pu_l[j * PB + q], plim)];
                                        else image_get(img, fb[j & 1][q], j * PB + q);
                                    }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- MFMA c ----
                    if ((FX_WG_ABL & 2) && nreq > 8) acc[c][0] += af[ks] + ((ks & 1) ? b1[c] : b0[c]);  // (one add keeps the operands alive)
                    else if (ks & 1) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks], b1[c], acc[c], 0, 0, 0);
                    else acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks], b0[c], acc[c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (flush && ks == KPUT - 1) {  // the image is written: (PC 2) both halves, behind the pair's barrier
                    if constexpr (PC > 1) wg_lds_barrier();
                    else wave_lds_fence();
                }
            }
            if constexpr ((KS & 1) != 0) {  // (odd KS: the last K-step used b0 and loaded K-step 0 into b1 -- keep "b0 = K-step 0")
#pragma unroll
                for (int c = 0; c < CTW; ++c) b0[c] = b1[c];
            }
            wave_lds_fence();  // the image has been read
            // first use of the prefetched fragments in the same block as the stores: exact vmcnt
            if constexpr (!FAST) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(an[ks]));
            }
        };

        if constexpr (MIX == 0) {
        if (nsteps > 0) {
            v4d accA1[FAST ? 1 : CTW], accB1[FAST ? 1 : CTW];
            auto& accA = choose_ref<FAST>(accA0, accA1);
            auto& accB = choose_ref<FAST>(accB0, accB1);
            load_b(b0, 0);
            if constexpr (!FAST) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(fa0[ks]));
            }
            WG_TICK(2);
            if constexpr (fast) {
                // (the first group "flushes" the other set's garbage to the scratch buffer: piece 0 only -- no flag in the slices)
                if (flip) stage(RT, RT, pend_frow, accB, accA, image_of(1), fa0, fa1, true, pend_base, pend_lim);
                else stage(RT, RT, pend_frow, accA, accB, image_of(1), fa0, fa1, true, pend_base, pend_lim);
                flip = !flip;
                pend = true;
                pend_base = obase;
                pend_frow = 16 * pr;
                pend_lim = (pr == RT - 1 ? last_rows : 16) * npts / EPP - 1;
            } else {
            // even steps: fragments fa0, accumulators A; odd steps: fa1, B.  The tile of step s lives in image s & 1 (PC 1: the
            // wave's one image) from the first K-steps of step s + 1 until that step has sent it out.
            if constexpr (PC == 1) {
                // (one wave per row tile: the previous group's last tile, kept in accP, leaves under this group's first MFMAs)
                if (pend) stage(pr + PR, RT, pend_frow, accA, accP, image_of(1), fa0, fa1, true, pend_base, pend_lim);
                else stage(pr + PR, RT, 0, accA, accB, image_of(1), fa0, fa1, false, obase);
            } else {
                stage(pr + PR, RT, 0, accA, accB, image_of(1), fa0, fa1, false, obase);  // first step: nothing to flush yet
            }
            int i = 1;
            for (; i + 1 < nsteps; i += 2) {
                stage(pr + PR * (i + 1), RT, 16 * (pr + PR * (i - 1)), accB, accA, image_of(0), fa1, fa0, true, obase, BLK - 1);
                stage(pr + PR * (i + 2), RT, 16 * (pr + PR * i), accA, accB, image_of(1), fa0, fa1, true, obase, BLK - 1);
            }
            if (i < nsteps) {
                stage(pr + PR * (i + 1), RT, 16 * (pr + PR * (i - 1)), accB, accA, image_of(0), fa1, fa0, true, obase, BLK - 1);
                ++i;
            }
            // the last step's tile (the only one that may have fewer than 16 rows, or -- PC 2 -- not exist).  One wave per row tile
            // (PC 1, always one request per slab): the tile stays in registers (accP) and leaves under the first MFMAs of the
            // wave's next group -- with one or two row tiles per wave and group (values-only requests) the flush was a fifth of
            // the launch; a tile of fewer than 16 rows too (its rows that do not exist land in unused rows of the image, its
            // pieces stop at pend_lim): the wave that flushed it on the spot kept the other three waiting at the slab barrier,
            // 1000 of 11 700 clocks a request for 56 rows (FX_WG_TIME).  Otherwise: image, then out.
            const int ltile = pr + PR * (i - 1);
            const int lrows = ltile == RT - 1 ? last_rows : 16;
            bool defer = false;
            if constexpr (PC == 1) defer = !(FX_WG_ABL & 12);
            if (defer) {
#pragma unroll
                for (int c = 0; c < CTW; ++c) accP[c] = ((i - 1) & 1) ? accB[c] : accA[c];
                pend = true;
                pend_base = obase;
                pend_frow = 16 * ltile;
                pend_lim = lrows * npts / EPP - 1;
            } else {
                if constexpr (PC == 1) pend = false;
                double* imgl = image_of((i - 1) & 1);
                if (ltile < RT && !((FX_WG_ABL & 8) && nreq > 8)) {
                    if ((i - 1) & 1) image_put(imgl, accB, lrows);
                    else image_put(imgl, accA, lrows);
                }
                if constexpr (PC > 1) wg_lds_barrier();
                else wave_lds_fence();
                if (ltile < RT && !((FX_WG_ABL & 4) && nreq > 8)) {
                    if (lrows < 16) build_pieces(gmax, lrows);   // (the pieces of a request's rows that exist; rebuilt below)
                    constexpr int HB = NRW < 8 ? NRW : 8;
#pragma unroll
                    for (int r0 = 0; r0 < NRW; r0 += HB) {
                        FlushT fl[HB];
#pragma unroll
                        for (int q = 0; q < HB; ++q)
                            if (r0 + q < NRW) image_get(imgl, fl[q], r0 + q);
#pragma unroll
                        for (int q = 0; q < HB; ++q)
                            if (r0 + q < NRW) image_out(fl[q], r0 + q, 16 * ltile, obase);
                    }
                    if (lrows < 16) build_pieces(G - 1, 16);
                }
                wave_lds_fence();
            }
            }
            WG_TICK(3);
        }
        } else {
            // ---------------- dof-major sweep with the chain rule on the accumulators (MIX 1) ----------------
            constexpr int NTAB = 1 + SD;
            const int rows = a.R / NTAB;                       // rows per table
            const int RTd = (rows + 15) / 16;                  // dof tiles; fragment tile (i, t) = i NTAB + t, zero tile NZ
            const int rows_last = rows - 16 * (RTd - 1);
            const int NZ = RTd * NTAB;
            const int ndsteps = (RTd + PR - 1) / PR;           // dof steps of every wave pair (the barriers must match)
            v4d accA[NTAB][CTW], accB[NTAB][CTW];
            // d/dx_d = sum_c K[c][d] d/dX_c on the accumulators of one dof tile: element (c, jj) of every table is the same
            // (row, column) entry -- lane-local; K of the column's request from LDS
            auto mix = [&](v4d (&acc)[NTAB][CTW]) __attribute__((always_inline)) {
#pragma unroll
                for (int c = 0; c < CTW; ++c) {
                    double Kl[SD][SD];
#pragma unroll
                    for (int i = 0; i < SD; ++i)
#pragma unroll
                        for (int d = 0; d < SD; ++d) Kl[i][d] = kbuf[kofs[c] + i * SD + d];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        double gr[SD];
#pragma unroll
                        for (int i = 0; i < SD; ++i) gr[i] = acc[1 + i][c][jj];
#pragma unroll
                        for (int d = 0; d < SD; ++d) {
                            double t = 0.0;
#pragma unroll
                            for (int i = 0; i < SD; ++i) t += Kl[i][d] * gr[i];
                            acc[1 + d][c][jj] = t;
                        }
                    }
                }
            };
            // one dof step: the NTAB tiles of dof tile i = pr + PR ds into `cur`; meanwhile the (mixed) tables of the previous dof
            // tile leave, table t under the MFMAs of tile t.  DSODD: parity of ds -- which accumulator set is `cur`, and (odd
            // table counts) which fragment buffer / image the step starts on
            auto dstep = [&](int ds, auto dsodd_c, bool flush) __attribute__((always_inline)) {
                constexpr bool DSODD = decltype(dsodd_c)::value;
                constexpr int PAR = (NTAB & 1) ? (DSODD ? 1 : 0) : 0;
                const int i = pr + PR * ds;
                static_for<NTAB>([&](auto t_c) __attribute__((always_inline)) {
                    constexpr int t = decltype(t_c)::value;
                    const int ntile = t + 1 < NTAB ? i * NTAB + t + 1 : (i + PR) * NTAB;
                    const int frow = t * rows + 16 * (i - PR);
                    if constexpr (((t + PAR) & 1) == 0) {
                        if constexpr (DSODD) stage(ntile, NZ, frow, accB[t], accA[t], image_of(0), fa0, fa1, flush, obase);
                        else stage(ntile, NZ, frow, accA[t], accB[t], image_of(0), fa0, fa1, flush, obase);
                    } else {
                        if constexpr (DSODD) stage(ntile, NZ, frow, accB[t], accA[t], image_of(1), fa1, fa0, flush, obase);
                        else stage(ntile, NZ, frow, accA[t], accB[t], image_of(1), fa1, fa0, flush, obase);
                    }
                });
                if constexpr (DSODD) mix(accB);
                else mix(accA);
            };
            load_b(b0, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(fa0[ks]));
            dstep(0, std::false_type{}, false);
            int ds = 1;
            for (; ds + 1 < ndsteps; ds += 2) {
                dstep(ds, std::true_type{}, true);
                dstep(ds + 1, std::false_type{}, true);
            }
            if (ds < ndsteps) {
                dstep(ds, std::true_type{}, true);
                ++ds;
            }
            // the tables of the last dof step's tile (the only one that may have fewer than 16 rows, or not exist): image, out
            {
                const int i = pr + PR * (ds - 1);
                const int nrows = i == RTd - 1 ? rows_last : 16;
                if (i < RTd && nrows < 16) build_pieces(gmax, nrows);
                static_for<NTAB>([&](auto t_c) __attribute__((always_inline)) {
                    constexpr int t = decltype(t_c)::value;
                    double* imgl = image_of(t & 1);
                    if (i < RTd) {
                        if ((ds - 1) & 1) image_put(imgl, accB[t], nrows);
                        else image_put(imgl, accA[t], nrows);
                    }
                    wg_lds_barrier();
                    if (i < RTd) {
                        constexpr int HB = NRW < 8 ? NRW : 8;
#pragma unroll
                        for (int r0 = 0; r0 < NRW; r0 += HB) {
                            FlushT fl[HB];
#pragma unroll
                            for (int q = 0; q < HB; ++q)
                                if (r0 + q < NRW) image_get(imgl, fl[q], r0 + q);
#pragma unroll
                            for (int q = 0; q < HB; ++q)
                                if (r0 + q < NRW) image_out(fl[q], r0 + q, t * rows + 16 * i, obase);
                        }
                    }
                });
                if (i < RTd && nrows < 16) build_pieces(G - 1, 16);
                wave_lds_fence();
            }
        }
        wg_lds_barrier();  // every wave is done with the slab
        cur = nxt;
        nxt += gridDim.x;
#pragma unroll
        for (int d = 0; d < SD; ++d) xcur[d] = xnext[d];
    }
#if FX_WG_TIME
    if (blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 6; ++i) trash[8192 + wave * 8 + i] = (double)tacc[i];
        trash[8192 + wave * 8 + 6] = (double)((ngroups + gridDim.x - 1) / gridDim.x);
    }
#endif
    if constexpr (PC == 1 && MIX == 0) {  // the tile still in registers
        if (pend) {
            double* imgl = imgs + (size_t)wave * wg_image_doubles(CT);
#pragma unroll
            for (int c = 0; c < CTW; ++c)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if constexpr (FAST) imgl[ioff[c][jj]] = flip ? accA0[c][jj] : accB0[c][jj];
                    else imgl[ioff[c][jj]] = accP[c][jj];
                }
            wave_lds_fence();
            char* tb = reinterpret_cast<char*>(pend_base + (size_t)pend_frow * npts);
#pragma unroll
            for (int q = 0; q < NRW; ++q) {
                const int u = min(pu_l[q], pend_lim);
                const FlushT f = reinterpret_cast<const FlushT*>(imgl)[u];
                FlushT* g2 = reinterpret_cast<FlushT*>(tb + (unsigned)u * (unsigned)(8 * EPP));
                if constexpr (ODD) *g2 = f;
                else stream_store(g2, f);
            }
        }
    }
}

}  // namespace fxk
