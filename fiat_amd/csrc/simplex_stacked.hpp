// Stacked-matrix simplex tabulation kernel (gfx950): large shapes, requests on the element's own cell.
//
// All tables of a request at once as ONE matrix product:
//     out[req][(table, row)][point] = A_stack[(table, row)][k] * Phi_k(point),
//     A_stack = [C; C D^(1,0,0); C D^(0,1,0); ...]   (R = ntab * rows stacked rows, host: api.hip)
// where D^alpha are the derivative matrices of the expansion set on the element's cell
// (d^alpha phi_j = sum_k D^alpha[j][k] phi_k: the route FIAT itself takes above its recurrence order,
// FIAT/expansions.py:438-446), so the recurrence runs for the VALUES only (order 0: ~6 flop per member and
// point instead of ~86 with Hessians) and everything else is fp64 MFMA work.  The cooperative kernel
// (coop_kernel.hpp) spends 2/3 of its time in the order-2 recurrence of DG P6; here it is ~5 %.
//
// One wave per SIMD, every wave on its own: a wave takes a group of G requests (G * npts <= 16 CT
// columns = CT column tiles), runs the order-0 recurrence with lane <-> column, turns the members into MFMA
// B fragments through LDS ONCE per group and keeps them in registers, then sweeps the R / 16 row tiles of
// A_stack: A fragments stream from L2 one row tile ahead (A_stack is 0.5 MB for DG P6 with Hessians,
// far beyond LDS), CT MFMAs per K-step, the 16 x (16 CT) result goes through a per-wave LDS image and
// leaves as 16-byte stores of 16 * npts contiguous doubles per request.  No workgroup barriers;
// groups are handed out dynamically (work_queue.hpp).
#pragma once
#include "simplex_stream.hpp"
#include "store.hpp"
#include "work_queue.hpp"
#include <type_traits>

namespace fxk {

template <int NC> struct StackedArgs {
    const double* pts;    // [nreq][npts][SD]
    const double* verts;  // [nreq][SD+1][SD] per-request cells, or nullptr: the element's own cell (A0, b0).  With
                          // cells the tables are still derivatives w.r.t. the ELEMENT's cell coordinates; the chain
                          // rule across tables is applied by table_mix_kernel afterwards (api.hip)
    double* out;          // [nreq][R][npts]
    const double* afrag;  // [RT + 1][KS][64]: 16x16x4 A fragments of A_stack, K in production order; last tile zero
    double coef[NC > 0 ? NC : 1];  // [nsteps][3] = A, B, C
    double phi0;
    double A0[9];
    double b0[3];
    double A0inv[9];  // (MIXT instances: K = A0^-1 A_req of the chain rule)
    double G[9];      // (PIO instances: A0 / 2, the reference part of the Piola Jacobian J = E_req G)
    int piola;        // (PIO instances: 1 covariant, 2 contravariant)
    long long nreq;
    int npts;
    int R;   // stacked rows = ntab * rows
    int RT;  // row tiles = ceil(R / 16)
    int debug;
    // FX_DBG & 1024 builds: sizes (in doubles) of the buffers behind pts / verts / out / afrag; every global access
    // of the kernel is range-checked against them, redirected to the scratch area when outside and reported there
    long long lim_pts, lim_verts, lim_out, lim_afrag;
    int gslab;  // (simplex_wg.hpp) requests that share a slab of 16 CT columns
};

#if FX_DBG & 1024
// site: 1 pts, 2 verts, 3 afrag, 4 out; trash[4096 + 4 site ..] = {count, first offending index, limit, group}
__device__ __forceinline__ long long dbg_check(long long idx, long long lim, int site, long long grp, double* trash) {
    if (idx >= 0 && idx < lim) return idx;
    double* t = trash + 4096 + 4 * site;
    if (atomicAdd(reinterpret_cast<unsigned long long*>(t), 1ULL) == 0ULL) {
        t[1] = (double)idx;
        t[2] = (double)lim;
        t[3] = (double)grp;
    }
    return 0;
}
#define FX_CHK(idx, lim, site) dbg_check((long long)(idx), (lim), (site), (long long)grp_dbg, trash)
#else
#define FX_CHK(idx, lim, site) (idx)
#endif

// per-wave LDS: the output image of a row tile (16 x 16 CT doubles + dump row + read slack); the expansion
// values of a group ([4 KS slots][16 CT columns]) alias it while they are produced
// (four column tiles: the values are produced in two passes of two tiles, which halves the slab)
constexpr int stacked_passes(int CT) { return CT > 3 ? 2 : 1; }
constexpr int stacked_image_doubles(int CT, int KS, int slots = 1) {
    return (4 * KS * 16 * CT / stacked_passes(CT) > slots * 16 * 16 * CT + 128) ? 4 * KS * 16 * CT / stacked_passes(CT)
                                                                                : slots * 16 * 16 * CT + 128;
}

// G requests of <= (16 CT / G) points per group.  RTC > 0: the stacked matrix has exactly RTC row tiles and
// all of its A fragments stay in registers for the whole launch (small shapes: nothing is loaded in the
// sweep, which is fully unrolled); RTC == 0: any number of row tiles, fragments streamed from L2.
// WPS: waves per SIMD the register allocation aims for (short sweeps need other waves to cover the
// production phase of a group).
// CHUNK: a unit is one request's next 16 CT points (any number of points per request, odd table sizes too):
// the image of a row tile is [row][points of the chunk] and leaves row by row as 8-byte stores.  (With MIXR: the same
// units on dof-major tiles, chain rule on the accumulators.)
// MIXT > 0 (= 1 + SD, per-request cells, order 1): the row tiles come dof-major -- the MIXT tables of 16 dofs one
// after the other (each table padded to whole tiles) -- so that a wave holds values and all first derivatives of
// those dofs at once; their images go to LDS together and the flush applies the chain rule
// d/dx_d = sum_c K[c][d] d/dX_c while it copies them out.
// MIXR (with MIXT = 1 + SD: order 1, or MIXT = 1 + SD + SD (SD + 1) / 2: order 2): the chain rule is applied to the ACCUMULATORS
// instead -- in the MFMA result layout a lane holds the same (row, column) entry of every table, so the mix across tables is
// lane-local: per column tile the lane reads K of its column's request from LDS (9 doubles, transient) and rewrites its
// accumulators; the images then hold final values and the flush is a plain copy (the flush-side mix reads every image SD
// times, and its wave-uniform K of all G requests lives in up to 72 scalar registers).  A dof tile is worked off in "halves":
// values + gradient (1 + SD tables), then the Hessian tables (SD (SD + 1) / 2; they mix among themselves only -- the map is
// affine), each half flushing the images of the half before it under its MFMAs.
// ODD: requests of an odd number of doubles (odd row count x odd point count: P4 / RT2 / N3 tetrahedra, P5 triangles at rules
// with odd point counts) start on 8-byte boundaries only -- the whole-request flush then moves 8 bytes per lane instead of 16.
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int CNT, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, CNT>{}, f); }

constexpr int stacked_mix_slots(int sd, int mixt, bool mixr) {  // row-tile images per wave
    return mixt <= 1 ? 1 : (mixr && mixt > 1 + sd && sd * (sd + 1) / 2 > 1 + sd) ? sd * (sd + 1) / 2 : 1 + sd;
}
constexpr int STACKED_KBUF = 80;  // doubles per wave behind the images (MIXR): K, and the Piola matrices, of the group's (<= 4) requests
// PIO (vector-valued elements, value shape (SD,), rows = (dof, component)): rows per dof tile and where the tile position
// 4 jj + kk (element jj of lane kk of an MFMA result column) sits among them -- the SD components of a dof in ONE lane:
//   SD 3: lane kk holds dof kk of the tile, jj = component (jj 3 unused: 12 rows per tile);  SD 2: two dofs per lane, 16 rows
constexpr int stacked_tile_rows(int sd, int pio) { return pio && sd == 3 ? 12 : 16; }
__host__ __device__ constexpr int stacked_pio_row(int sd, int jj, int kk) {
    return sd == 3 ? (jj < 3 ? 3 * kk + jj : -1) : 8 * (jj >> 1) + 2 * kk + (jj & 1);
}

template <int SD, int N, int CT, int G, int RTC = 0, int WPS = 1, bool CHUNK = false, int MIXT = 0, bool ODD = false, bool MIXR = false, int PIO = 0>
__global__ __launch_bounds__(256, WPS) void tabulate_simplex_stacked(const StackedArgs<FixedNC<SD, N>::value> a,
                                                                   double* __restrict__ trash,
                                                                   unsigned int* __restrict__ gqueue) {
    constexpr StepTable<SD, N> TBL{};
    constexpr int NEXP = StepTable<SD, N>::NEXP;
    constexpr int KS = (NEXP + 3) / 4;
    constexpr int CPR = 16 * CT / G;                // column budget of one request
    constexpr int TR = stacked_tile_rows(SD, PIO);  // rows of a row tile (16; 12 with three components per lane)
    constexpr int NST = ODD ? (TR * CPR + 63) / 64 : (TR * CPR / 2 + 63) / 64;   // 16-byte (ODD: 8-byte) stores per lane and request chunk
    static_assert(!ODD || (!CHUNK && (MIXT == 0 || MIXR) && RTC == 0), "8-byte flush: whole-request groups, streamed fragments");
    using FlushT = typename std::conditional<ODD, double, v2d>::type;
    constexpr int PCH = 16 * CT;                    // points per chunk (CHUNK)
    static_assert(!CHUNK || (G == 1 && RTC == 0), "point-chunked units: one request per group, streamed fragments");
    constexpr int NTA = MIXT == 1 ? 1 : 1 + SD, NTB = SD * (SD + 1) / 2;   // tables of orders <= 1, of order 2
    constexpr int MORD = MIXT <= 1 ? 0 : MIXT == NTA ? 1 : 2;              // derivative order of a table-mixing instance
    static_assert(PIO == 0 || (MIXR && !CHUNK && SD >= 2), "fused Piola map: accumulator-side mixing, whole requests");
    static_assert(MIXT != 1 || PIO != 0, "one table: only the Piola map is left to mix");
    constexpr int SLOTS = stacked_mix_slots(SD, MIXT, MIXR);
    constexpr int IMG = stacked_image_doubles(CT, KS, SLOTS) + (MIXR ? STACKED_KBUF : 0);
    static_assert(MIXT == 0 || ((MIXT == NTA || (MIXR && MIXT == NTA + NTB)) && (!CHUNK || MIXR) && RTC == 0 && (!ODD || MIXR)),
                  "table mixing: orders 1 and 2, whole-request groups");
    static_assert(!MIXR || (MIXT > 0 && G * SD * SD <= STACKED_KBUF && KS >= 3), "accumulator-side mixing");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double* img = lds + WQ_CTL_DOUBLES + (size_t)wave * IMG;
    constexpr int DUMP = SLOTS * 16 * 16 * CT;

    typedef const __attribute__((address_space(4))) double CDouble;
    typedef StackedArgs<FixedNC<SD, N>::value> ArgsT;
    const __attribute__((address_space(4))) char* kargs =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    CDouble* kcoef = (CDouble*)(kargs + __builtin_offsetof(ArgsT, coef));

    const int npts = a.npts;
    const int chunk = TR * npts;  // doubles one request contributes to a row tile
    const int nchunk = CHUNK ? (npts + PCH - 1) / PCH : 1;
    const long long ngroups = CHUNK ? a.nreq * nchunk : (a.nreq + G - 1) / G;
    WorkQueue wqueue;
    wqueue.init(lds, gqueue, ngroups);
    __syncthreads();

    // column of this lane in column tile c: (request of the group, point); padding columns recompute
    // a valid point and drop their results in the dump row
    const int kk = lane >> 4;
    int ioff[CT];
    int kofs[MIXR ? CT : 1];  // MIXR: where K of the column's request sits in the wave's K buffer
    {
        const float rinv = 1.0f / (float)npts;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int j = 16 * c + (lane & 15);
            const int g = idiv_small(j, rinv);
            const bool valid = g < G;
            ioff[c] = valid ? g * chunk + (PIO ? (SD == 3 ? 3 : 2) : 1) * kk * npts + (j - g * npts) : -1;
            if constexpr (MIXR) kofs[c] = valid ? g * SD * SD : 0;
        }
    }

    // production: lane <-> column `h PWP + lane` of the group in pass h (lanes < PWP)
    constexpr int PH = stacked_passes(CT), PWP = 16 * CT / PH;
    int pg[PH], ppt[PH];
#pragma unroll
    for (int h = 0; h < PH; ++h) {
        const int j = h * PWP + lane;
        const int g = idiv_small(j, 1.0f / (float)npts);
        const bool valid = g < G && lane < PWP;
        pg[h] = valid ? g : 0;
        ppt[h] = valid ? j - g * npts : 0;
    }

#if FX_DBG & 512
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    long long grp = wqueue.claim();
    long long grp_dbg = grp;
    (void)grp_dbg;
    wqueue.service();
    // points of the wave's next group, one group ahead (their latency would otherwise be paid per group)
    auto load_points = [&](long long g_, double (&x)[PH][SD]) {
#pragma unroll
        for (int h = 0; h < PH; ++h) {
            const long long gg = g_ < ngroups ? g_ : ngroups - 1;
            long long req;
            int pt;
            if constexpr (CHUNK) {
                req = gg / nchunk;
                const int p0 = (int)(gg - req * nchunk) * PCH;
                pt = min(p0 + h * PWP + lane, npts - 1);  // (lanes past the chunk recompute a valid point)
            } else {
                req = gg * G + pg[h];
                req = req < a.nreq ? req : a.nreq - 1;
                pt = ppt[h];
            }
            const double* pp = a.pts + FX_CHK(((size_t)req * npts + pt) * SD, a.lim_pts - SD + 1, 1);
#pragma unroll
            for (int d = 0; d < SD; ++d) x[h][d] = pp[d];
        }
    };
    double xnext[PH][SD];
    load_points(grp, xnext);
    double areg[RTC > 0 ? RTC : 1][KS];
    if constexpr (RTC > 0) {
#pragma unroll
        for (int t = 0; t < RTC; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) areg[t][ks] = a.afrag[((size_t)t * KS + ks) * 64 + lane];
    }
#pragma unroll
    for (int h = 0; h < PH; ++h)
#pragma unroll
        for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xnext[h][d]));
    while (grp < ngroups) {
        const long long gnext = wqueue.claim();
        wqueue.service();
        grp_dbg = grp;
        // ---------------- expansion values -> B fragments ----------------
        // lane <-> column (request of the group, point): the order-0 recurrence once per column, every member to
        // a [slot][column] slab in LDS (it aliases the output image, which is idle until the sweep starts), then
        // each lane (kk, col) picks the members 4 ks + kk of its columns: the MFMA B fragments, kept in registers.
        double bf[KS][CT];
        {
            double* phi = img;
            double xcur[PH][SD];
#pragma unroll
            for (int h = 0; h < PH; ++h)
#pragma unroll
                for (int d = 0; d < SD; ++d) xcur[h][d] = xnext[h][d];
            load_points(gnext, xnext);
#pragma unroll
            for (int h = 0; h < PH; ++h) {
                if (lane < PWP) {
                    double X[SD];
                    if (a.verts) {  // (wave-uniform branch) physical point -> default simplex through the request's cell
                        long long req = CHUNK ? grp / nchunk : grp * G + pg[h];
                        req = req < a.nreq ? req : a.nreq - 1;
                        double J[SD][SD], bb[SD];
                        cell_map<SD>(a.verts + FX_CHK((size_t)req * (SD + 1) * SD, a.lim_verts - (SD + 1) * SD + 1, 2), J, bb);
#pragma unroll
                        for (int i = 0; i < SD; ++i) {
                            double t = bb[i];
#pragma unroll
                            for (int d = 0; d < SD; ++d) t += J[i][d] * xcur[h][d];
                            X[i] = t;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < SD; ++i) {
                            double t = a.b0[i];
#pragma unroll
                            for (int d = 0; d < SD; ++d) t += a.A0[i * SD + d] * xcur[h][d];
                            X[i] = t;
                        }
                    }
                    double mem[NEXP];
                    double ufa = 0.0, ufb = 0.0, ufc = 0.0;
                    int fcodim = -1;
                    auto produce = [&](int slot) -> double {
                        if (slot == 0) {
                            mem[0] = a.phi0;
                            return mem[0];
                        }
                        if (slot >= NEXP) return 0.0;
                        const int s = slot - 1;
                        const CDouble* cb = kcoef;
                        // opaque base, immediate offsets (see simplex_pair.hpp); tied to the step's input so that
                        // the pointer copies are not all made (and spilled) ahead of the recurrence
                        asm volatile("" : "+s"(cb) : "v"(mem[TBL.cur[s]]));
                        const CDouble* cq = cb + 3 * s;
                        const double cA = cq[0], cB = cq[1], cC = cq[2];
                        if (TBL.codim[s] != fcodim) {
                            fcodim = TBL.codim[s];
                            point_factors<SD>(fcodim, X, ufa, ufb, ufc);
                        }
                        const double f = cA * ufa - cB * ufb;
                        double v = mem[TBL.cur[s]] * f;
                        if (TBL.prv[s] >= 0) v -= cC * ufc * mem[TBL.prv[s]];
                        mem[TBL.dst[s]] = v;
                        return v;
                    };
#if FX_DBG & 1  // ablation: no recurrence
#pragma unroll
                    for (int slot = 0; slot < 4 * KS; ++slot) phi[slot * PWP + lane] = X[0] + slot;
                    (void)produce;
#else
#pragma unroll
                    for (int slot = 0; slot < 4 * KS; ++slot) phi[slot * PWP + lane] = produce(slot);
#endif
                }
                wave_lds_fence();
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int cc = 0; cc < CT / PH; ++cc)
                        bf[ks][h * (CT / PH) + cc] = phi[(4 * ks + kk) * PWP + 16 * cc + (lane & 15)];
                wave_lds_fence();
            }
        }

        // ---------------- sweep the row tiles of A_stack ----------------
        // Software pipeline over the row tiles: while the MFMAs of tile rt run, the finished tile rt-1 goes
        // through the LDS image to HBM (LDS and vector-memory instructions issue under the 64-cycle MFMAs;
        // two accumulator sets, the loop is unrolled by two so that they swap roles without copies).
        const double* ap = a.afrag + lane;
        double fa0[KS], fa1[KS];  // A fragments of the even / odd row tiles (loaded one tile ahead)
        if constexpr (RTC == 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fa0[ks] = a.afrag[FX_CHK(ks * 64 + lane, a.lim_afrag, 3)];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(fa0[ks]));
        }
        long long oreq[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            oreq[g] = CHUNK ? grp / nchunk : grp * G + g;
            // (MIXR: the padding requests of the last group ARE its last request -- same points, same cell, same values to the
            // same place -- so the stores need no second destination)
            if constexpr (MIXR) oreq[g] = oreq[g] < a.nreq ? oreq[g] : a.nreq - 1;
        }
        const int p0 = CHUNK ? (int)(grp - oreq[0] * nchunk) * PCH : 0;  // first point of the chunk
        const int pc = CHUNK ? min(PCH, npts - p0) : npts;             // points of the chunk = row stride of the image
        const int RTn = RTC > 0 ? RTC : a.RT;
        // opaque per-group copies of the lane-derived offsets: otherwise every (tile, store) address of the
        // unrolled sweep is precomputed outside the group loop and spilled (see simplex_pair.hpp)
        int elane = lane, ekk = kk, enpts = npts;
        int eoff[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if constexpr (CHUNK) {
                const int j = 16 * c + (lane & 15);
                eoff[c] = j < pc ? kk * pc + j : -1;
            } else {
                eoff[c] = ioff[c];
            }
        }
        if constexpr (RTC > 0) {
            asm volatile("" : "+v"(elane), "+v"(ekk), "+s"(enpts));
#pragma unroll
            for (int c = 0; c < CT; ++c) asm volatile("" : "+v"(eoff[c]));
        }
        const int echunk = TR * enpts;
        const int last_rows = a.R - 16 * (RTn - 1);

        // D tile: element jj of lane (kk, col) is row 4 jj + kk -> image [request][row][point]
        // (`nrows`: 16 for every tile but possibly the last one -- the pipelined stages only ever flush full
        // tiles, which makes all their LDS and output offsets loop invariants)
        // D tile: element jj of lane (kk, col) is row 4 jj + kk -> image [request][row][point]
        const int estr = CHUNK ? pc : enpts;  // row stride of the image
        auto image_put = [&](const v4d (&acc)[CT], int w, int nrows, int soff = 0) {  // w-th of the 4 CT image stores
            const int c = w >> 2, jj = w & 3;
            if constexpr (PIO != 0) {  // (rows of the tile: stacked_pio_row)
                if (SD == 3 && jj == 3) return;
                const int jrow = SD == 3 ? jj : 8 * (jj >> 1) + (jj & 1);
                const bool ok = eoff[c] >= 0 && jrow + (SD == 3 ? 3 : 2) * ekk < nrows;
                img[ok ? soff + eoff[c] + jrow * estr : DUMP + elane] = acc[c][jj];
            } else {
                const bool ok = eoff[c] >= 0 && 4 * jj + ekk < nrows;
                img[ok ? soff + eoff[c] + 4 * jj * estr : DUMP + elane] = acc[c][jj];
            }
        };
        constexpr int NRD = CHUNK ? (16 * PCH + 63) / 64 : G * NST;  // image reads = output stores per row tile
        FlushT fbuf[CHUNK ? 1 : NRD];
        double fbuf1[CHUNK ? NRD : 1];
        const float rpc = 1.0f / (float)pc;
        auto image_get = [&](int r, int nrows, int soff = 0) {  // r-th image read
            if constexpr (CHUNK) {
                fbuf1[r] = img[soff + min(r * 64 + elane, nrows * pc - 1)];
            } else {
                const int g = r / NST, it = r % NST;
                const int nch = ODD ? nrows * enpts : (nrows * enpts) >> 1;  // pieces of a request's chunk (16-byte: even, host-checked)
                fbuf[r] = reinterpret_cast<const FlushT*>(img + soff + g * echunk)[min(it * 64 + elane, nch - 1)];
            }
        };
        auto image_out = [&](int r, int rowbase, int nrows) {  // r-th output store of the row tile starting at row `rowbase`
            if constexpr (CHUNK) {
                const int i = min(r * 64 + elane, nrows * pc - 1);
                const int row = idiv_small(i, rpc);
                double* dst = a.out + FX_CHK(((size_t)oreq[0] * a.R + (size_t)rowbase + row) * enpts + p0 + (i - row * pc), a.lim_out, 4);
#if FX_DBG & 1024
                if (dst == a.out && !(oreq[0] == 0 && rowbase + row == 0 && p0 + (i - row * pc) == 0)) dst = trash;
#endif
                // plain stores: the chunks of a request are pieces of its rows (16 CT points out of npts), which start anywhere
                // in a 128-byte line; the neighbouring chunk completes the line in L2 (non-temporal stores of partial lines are
                // expensive)
                *dst = fbuf1[r];
            } else {
                const int g = r / NST, it = r % NST;
                const int nch = ODD ? nrows * enpts : (nrows * enpts) >> 1;
#if FX_DBG & 1024
                constexpr int EPP = ODD ? 1 : 2;  // doubles per piece
                const long long o0 = ((long long)oreq[g] * a.R + (long long)rowbase) * enpts + EPP * min(it * 64 + elane, nch - 1);
                FlushT* g2 = oreq[g] < a.nreq ? reinterpret_cast<FlushT*>(a.out + FX_CHK(o0, a.lim_out - (EPP - 1), 4)) : reinterpret_cast<FlushT*>(trash);
                if (oreq[g] < a.nreq && (o0 < 0 || o0 >= a.lim_out - (EPP - 1))) g2 = reinterpret_cast<FlushT*>(trash);
                stream_store(g2, fbuf[r]);
#else
                FlushT* g2 = (MIXR || oreq[g] < a.nreq) ? reinterpret_cast<FlushT*>(a.out + ((size_t)oreq[g] * a.R + (size_t)rowbase) * enpts)
                                                        : reinterpret_cast<FlushT*>(trash);
                if constexpr (ODD) g2[min(it * 64 + elane, nch - 1)] = fbuf[r];  // (8-byte pieces, lines shared with the neighbours: plain stores)
                else stream_store(&g2[min(it * 64 + elane, nch - 1)], fbuf[r]);  // (all-plain here: 30 shapes, geometric mean 1.05 x the time, 0.85 ... 1.64)
#endif
            }
        };
        auto mfma_steps = [&](v4d (&acc)[CT], const double (&af)[KS], int k0, int k1) {
#if FX_DBG & 2  // ablation: no MFMAs (one add keeps the operands alive)
#pragma unroll
            for (int ks = k0; ks < k1; ++ks)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c][0] += af[ks] + bf[ks][c];
            return;
#endif
#pragma unroll
            for (int ks = k0; ks < k1; ++ks)
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks], bf[ks][c], acc[c], 0, 0, 0);
        };
        // One pipeline stage: MFMAs of tile rt (fragments `af`) into `cur`, tile rt-1 (in `prev`) out, fragments
        // of tile rt+1 into `an`.  The memory instructions are spread over the K-steps (a wave issues in order:
        // a burst of 17 loads or 18 LDS operations between two MFMAs leaves the matrix pipe idle): every K-step
        // CT MFMAs + its share of the fragment loads (first two thirds), image stores (first third), image
        // reads (second third) or output stores (last third); the scheduler may not move anything across a K-step.
        constexpr int NWR = 4 * CT, T3 = KS / 3;
        constexpr int LPK = (KS + 2 * T3 - 1) / (2 * T3);
        // (image stores and image reads are packed into the first TP K-steps of their thirds: the LDS fence between them and
        // the first output store then find their operands complete instead of stalling the only wave of the SIMD)
        constexpr int TP = T3 > 3 ? T3 - 2 : T3;
        constexpr int WPK = (NWR + TP - 1) / TP, RPK = (NRD + TP - 1) / TP, SPK = (NRD + (KS - 2 * T3) - 1) / (KS - 2 * T3);
        auto stage = [&](v4d (&cur)[CT], const v4d (&prev)[CT], int rt, const double (&af)[KS], double (&an)[KS]) {
            const double* anp = ap + (size_t)(rt + 1) * KS * 64;  // (the buffer ends with a zero tile)
#pragma unroll
            for (int c = 0; c < CT; ++c) cur[c] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                mfma_steps(cur, af, ks, ks + 1);
                // fragment loads in the first two thirds only: at the end of the stage they are older than every
                // output store, so the wait for them is an exact vmcnt(#stores), not a wait for the stores
                if (RTC == 0 && ks < 2 * T3) {
#pragma unroll
                    for (int q = ks * LPK; q < (ks + 1) * LPK && q < KS; ++q)
                        an[q] = anp[FX_CHK(q * 64 + (anp - a.afrag), a.lim_afrag, 3) - (anp - a.afrag)];
                }
                if (ks < TP) {
#pragma unroll
                    for (int w = ks * WPK; w < (ks + 1) * WPK && w < NWR; ++w) image_put(prev, w, 16);
                } else if (ks >= T3 && ks < T3 + TP) {
                    if (ks == T3) wave_lds_fence();
#pragma unroll
                    for (int r = (ks - T3) * RPK; r < (ks - T3 + 1) * RPK && r < NRD; ++r) image_get(r, 16);
                } else if (ks < 2 * T3) {
                } else {
#pragma unroll
                    for (int r = (ks - 2 * T3) * SPK; r < (ks - 2 * T3 + 1) * SPK && r < NRD; ++r) image_out(r, 16 * (rt - 1), 16);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            wave_lds_fence();
            // first use of the prefetched fragments in the same block as the stores: exact vmcnt
            if constexpr (RTC == 0) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(an[ks]));
            }
        };
        auto flush_last = [&](const v4d (&acc)[CT]) {
#pragma unroll
            for (int w = 0; w < NWR; ++w) image_put(acc, w, last_rows);
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < NRD; ++r) image_get(r, last_rows);
#pragma unroll
            for (int r = 0; r < NRD; ++r) image_out(r, 16 * (RTn - 1), last_rows);
            wave_lds_fence();
        };
        if constexpr (MIXT > 0 && MIXR) {
            // ---- dof-major tiles, chain rule applied to the accumulators (lane-local), plain-copy flush ----
            const int rows = a.R / MIXT;                      // rows per table
            const int RTd = (rows + TR - 1) / TR;             // dof tiles; tile (i, t) = table t, rows [TR i, TR i + TR)
            const int rows_last = rows - TR * (RTd - 1);
            constexpr int SLOT = 16 * 16 * CT;
            double* kbuf = img + (IMG - STACKED_KBUF);        // [G][SD][SD]: K = A0^-1 A_req of the group's requests
            double* pbuf = kbuf + STACKED_KBUF / 2;           // [G][SD][SD]: the Piola matrices (PIO)
            if constexpr (PIO != 0) {
                if (lane < G) {
                    long long req = grp * G + lane;
                    req = req < a.nreq ? req : a.nreq - 1;
                    double M[SD][SD];
                    piola_matrix<SD>(a.verts + FX_CHK((size_t)req * (SD + 1) * SD, a.lim_verts - (SD + 1) * SD + 1, 2), a.G, a.piola, M);
#pragma unroll
                    for (int i = 0; i < SD; ++i)
#pragma unroll
                        for (int d = 0; d < SD; ++d) pbuf[lane * SD * SD + i * SD + d] = M[i][d];
                }
            }
            if (MORD > 0 && lane < G) {
                long long req = CHUNK ? grp / nchunk : grp * G + lane;
                req = req < a.nreq ? req : a.nreq - 1;
                double J[SD][SD], bb[SD];
                cell_map<SD>(a.verts + FX_CHK((size_t)req * (SD + 1) * SD, a.lim_verts - (SD + 1) * SD + 1, 2), J, bb);
#pragma unroll
                for (int i = 0; i < SD; ++i)
#pragma unroll
                    for (int d = 0; d < SD; ++d) {
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < SD; ++k) t += a.A0inv[i * SD + k] * J[k][d];
                        kbuf[lane * SD * SD + i * SD + d] = t;
                    }
            }
            wave_lds_fence();
            v4d acc[SLOTS][CT];
            // Hessian tables in mis() order: (c, c'), c <= c' -> c (2 SD - c - 1) / 2 + c'
            auto hidx = [](int c, int e) { return c <= e ? c * (2 * SD - c - 1) / 2 + e : e * (2 * SD - e - 1) / 2 + c; };
            // chain rule on the accumulators of one half: d/dx_d = sum_c K[c][d] d/dX_c; d2/dx_d dx_e = sum K[c][d] K[c'][e] d2/dX_c dX_c'
            auto mix_acc = [&](auto phase_c) {
                constexpr int PHS = decltype(phase_c)::value;
                if constexpr (PIO != 0) {  // phi = M Phi on every table of the half: the components of a dof sit in one lane
                    constexpr int NCUR = PHS == 0 ? NTA : NTB;
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        double Ml[SD][SD];
#pragma unroll
                        for (int i = 0; i < SD; ++i)
#pragma unroll
                            for (int d = 0; d < SD; ++d) Ml[i][d] = pbuf[kofs[c] + i * SD + d];
#pragma unroll
                        for (int t = 0; t < NCUR; ++t)
#pragma unroll
                            for (int h = 0; h < (SD == 2 ? 2 : 1); ++h) {  // (two dofs per lane in 2-D)
                                double v[SD];
#pragma unroll
                                for (int i = 0; i < SD; ++i) v[i] = acc[t][c][SD * h + i];
#pragma unroll
                                for (int i = 0; i < SD; ++i) {
                                    double w = 0.0;
#pragma unroll
                                    for (int d = 0; d < SD; ++d) w += Ml[i][d] * v[d];
                                    acc[t][c][SD * h + i] = w;
                                }
                            }
                    }
                }
                if constexpr (MORD > 0) {
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    double Kl[SD][SD];
#pragma unroll
                    for (int i = 0; i < SD; ++i)
#pragma unroll
                        for (int d = 0; d < SD; ++d) Kl[i][d] = kbuf[kofs[c] + i * SD + d];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        if constexpr (PHS == 0) {
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
                        } else {
                            double T[SD][SD];  // T[i][e] = sum_k H[i][k] K[k][e]
#pragma unroll
                            for (int i = 0; i < SD; ++i)
#pragma unroll
                                for (int e = 0; e < SD; ++e) {
                                    double t = 0.0;
#pragma unroll
                                    for (int k = 0; k < SD; ++k) t += acc[hidx(i, k)][c][jj] * Kl[k][e];
                                    T[i][e] = t;
                                }
#pragma unroll
                            for (int d = 0; d < SD; ++d)
#pragma unroll
                                for (int e = d; e < SD; ++e) {
                                    double t = 0.0;
#pragma unroll
                                    for (int i = 0; i < SD; ++i) t += Kl[i][d] * T[i][e];
                                    acc[hidx(d, e)][c][jj] = t;
                                }
                        }
                    }
                }
                }
            };
            // One tile: MFMAs of tile q into `cur`, fragments of tile q + 1 into `an` (all in the first third of the K-steps:
            // older than every output store of the stage), and NF images of the previous half -- slots slot0 .., output rows
            // rowbase0 + f rows .. -- copied out in the remaining K-steps, image after image (reads, then stores)
            constexpr int LPK3 = (KS + T3 - 1) / T3;
            auto rstage = [&](v4d (&cur)[CT], int q, const double (&af)[KS], double (&an)[KS], auto nf_c, int slot0, int rowbase0, int pn) {
                constexpr int NF = decltype(nf_c)::value;
#pragma unroll
                for (int c = 0; c < CT; ++c) cur[c] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    mfma_steps(cur, af, ks, ks + 1);
                    if (ks < T3) {
#pragma unroll
                        for (int l = ks * LPK3; l < (ks + 1) * LPK3 && l < KS; ++l)
                            an[l] = a.afrag[FX_CHK((size_t)(q + 1) * KS * 64 + l * 64 + elane, a.lim_afrag, 3)];
                    }
#pragma unroll
                    for (int f = 0; f < (NF > 0 ? NF : 0); ++f) {
                        constexpr int NFD = NF > 0 ? NF : 1;
                        const int w0 = T3 + (KS - T3) * f / NFD, w1 = T3 + (KS - T3) * (f + 1) / NFD;  // K-steps of image f
                        const int wm = w1 - w0 >= 2 ? w0 + (w1 - w0) / 2 : w0;                        // reads [w0, wm), stores [wm, w1)
                        if (w1 - w0 < 2) {
                            if (ks == w0) {
#pragma unroll
                                for (int r = 0; r < NRD; ++r) image_get(r, pn, (slot0 + f) * SLOT);
#pragma unroll
                                for (int r = 0; r < NRD; ++r) image_out(r, rowbase0 + f * rows, pn);
                            }
                        } else if (ks >= w0 && ks < wm) {
                            const int per = (NRD + (wm - w0) - 1) / (wm - w0);
#pragma unroll
                            for (int r = (ks - w0) * per; r < (ks - w0 + 1) * per && r < NRD; ++r) image_get(r, pn, (slot0 + f) * SLOT);
                        } else if (ks >= wm && ks < w1) {
                            const int per = (NRD + (w1 - wm) - 1) / (w1 - wm);
#pragma unroll
                            for (int r = (ks - wm) * per; r < (ks - wm + 1) * per && r < NRD; ++r) image_out(r, rowbase0 + f * rows, pn);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(an[ks]));
            };
            // One half of dof tile i: PHS 0 = tables [0, NTA), 1 = tables [NTA, MIXT).  FL: the images hold the half before it
            // (dof tile i - 1 for PHS 0 of an order-1 instance and for PHS 0 after PHS 1, dof tile i for PHS 1), which leaves
            // under this half's MFMAs.  IODD: tile i MIXT is odd (odd table counts: the fragment buffers alternate per tile).
            auto half_tile = [&](int i, auto phase_c, auto flush_c, auto iodd_c) {
                constexpr int PHS = decltype(phase_c)::value;
                constexpr bool FL = decltype(flush_c)::value;
                constexpr int NCUR = PHS == 0 ? NTA : NTB, T0 = PHS == 0 ? 0 : NTA;
                constexpr int NPREV = !FL ? 0 : MORD <= 1 ? NTA : PHS == 0 ? NTB : NTA;
                constexpr int TP0 = (MORD == 2 && PHS == 0) ? NTA : 0;            // first table of the half before
                i = __builtin_amdgcn_readfirstlane(i);
                asm volatile("" : "+s"(i));  // (opaque per half: no induction variables for the ~40 output bases of a dof tile)
                const int iprev = (MORD == 2 && PHS == 1) ? i : i - 1;
                const int pn = iprev == RTd - 1 ? rows_last : TR;
                static_for<NCUR>([&](auto s_c) {
                    constexpr int s = decltype(s_c)::value;
                    constexpr int F0 = s * NPREV / NCUR, F1 = (s + 1) * NPREV / NCUR;
                    const int q = i * MIXT + T0 + s;
                    if constexpr (((T0 + s + (decltype(iodd_c)::value ? 1 : 0)) & 1) == 0)
                        rstage(acc[s], q, fa0, fa1, std::integral_constant<int, F1 - F0>{}, F0, (TP0 + F0) * rows + TR * iprev, pn);
                    else
                        rstage(acc[s], q, fa1, fa0, std::integral_constant<int, F1 - F0>{}, F0, (TP0 + F0) * rows + TR * iprev, pn);
                });
                mix_acc(phase_c);
                wave_lds_fence();  // (the images of the half before have been read)
                const int nrows = i == RTd - 1 ? rows_last : TR;
                static_for<NCUR>([&](auto s_c) {
                    constexpr int s = decltype(s_c)::value;
#pragma unroll
                    for (int w = 0; w < NWR; ++w) image_put(acc[s], w, nrows, s * SLOT);
                });
                wave_lds_fence();
            };
            using IC0 = std::integral_constant<int, 0>;
            using IC1 = std::integral_constant<int, 1>;
            half_tile(0, IC0{}, std::false_type{}, std::false_type{});
            if constexpr (MORD == 2) {
                static_assert(MIXT % 2 == 0, "even table count: the fragment buffer of a tile depends on its table only");
                half_tile(0, IC1{}, std::true_type{}, std::false_type{});
                for (int i = 1; i < RTd; ++i) {
                    half_tile(i, IC0{}, std::true_type{}, std::false_type{});
                    half_tile(i, IC1{}, std::true_type{}, std::false_type{});
                }
            } else if constexpr (MIXT % 2 == 0) {
                for (int i = 1; i < RTd; ++i) half_tile(i, IC0{}, std::true_type{}, std::false_type{});
            } else {
                int i = 1;
                for (; i + 1 < RTd; i += 2) {
                    half_tile(i, IC0{}, std::true_type{}, std::true_type{});
                    half_tile(i + 1, IC0{}, std::true_type{}, std::false_type{});
                }
                if (i < RTd) half_tile(i, IC0{}, std::true_type{}, std::true_type{});
            }
            // the images of the last half
            constexpr int NLAST = MORD == 2 ? NTB : NTA, TL0 = MORD == 2 ? NTA : 0;
#pragma unroll
            for (int t = 0; t < NLAST; ++t) {
#pragma unroll
                for (int r = 0; r < NRD; ++r) image_get(r, rows_last, t * SLOT);
#pragma unroll
                for (int r = 0; r < NRD; ++r) image_out(r, (TL0 + t) * rows + TR * (RTd - 1), rows_last);
            }
            wave_lds_fence();
        } else if constexpr (MIXT > 0) {
            // ---- dof-major tiles with the chain rule across the tables applied in the flush ----
            const int rows = a.R / MIXT;                      // rows per table
            const int RTd = (rows + 15) / 16;                 // dof tiles; tile (i, t) = table t, rows [16 i, 16 i + 16)
            const int rows_last = rows - 16 * (RTd - 1);
            constexpr int SLOT = 16 * 16 * CT;
            // K = A0^-1 A_req of every request of the group, wave uniform (scalar registers).  The tables go to the
            // LDS images UNMIXED; the flush of derivative table d reads the SD reference-derivative images at its
            // position and combines them with K[.][d] of the request it is copying -- a scalar operand, because the
            // copy runs request by request (mixing the accumulators in registers instead costs ~90 more VGPRs)
            double Ks[G][SD][SD];
            {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    long long req = grp * G + g;
                    req = req < a.nreq ? req : a.nreq - 1;
                    double J[SD][SD], bb[SD];
                    cell_map<SD>(a.verts + FX_CHK((size_t)req * (SD + 1) * SD, a.lim_verts - (SD + 1) * SD + 1, 2), J, bb);
#pragma unroll
                    for (int i = 0; i < SD; ++i)
#pragma unroll
                        for (int d = 0; d < SD; ++d) {
                            double t = 0.0;
#pragma unroll
                            for (int k = 0; k < SD; ++k) t += a.A0inv[i * SD + k] * J[k][d];
                            Ks[g][i][d] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(t)),
                                                           __builtin_amdgcn_readfirstlane(__double2loint(t)));
                        }
                }
            }
            // r-th 16-byte piece of table `slot` of the image set: the values as they are, derivative d = slot - 1 as
            // sum_c K[c][d] (reference derivative c)
            auto mixed_get = [&](int r, int nrows, int slot) {
                if (slot == 0) {
                    image_get(r, nrows, 0);
                } else {
                    const int g = r / NST, it = r % NST, d = slot - 1;
                    const int nch = (nrows * enpts) >> 1;
                    const int idx = min(it * 64 + elane, nch - 1);
                    v2d m = v2d{0.0, 0.0};
#pragma unroll
                    for (int c = 0; c < SD; ++c) {
                        const v2d x = reinterpret_cast<const v2d*>(img + (1 + c) * SLOT + g * echunk)[idx];
                        m += Ks[g][c][d] * x;
                    }
                    fbuf[r] = m;
                }
            };
            v4d acc[MIXT][CT];
            // one tile: MFMAs into `cur`, fragments of the next tile into `an`, and (FLUSH) slot `slot` of the previous
            // dof tile out -- the same K-step schedule as `stage`
            auto mix_stage = [&](v4d (&cur)[CT], int q, int slot, int prev_rowbase, const double (&af)[KS], double (&an)[KS], auto flush) {
                // (uniform tile base + 32-bit lane offset: scalar-base addressing; per-lane 64-bit pointers for the four
                // stages of a dof tile get strength-reduced into ~50 registers of induction variables and spill)
#pragma unroll
                for (int c = 0; c < CT; ++c) cur[c] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    mfma_steps(cur, af, ks, ks + 1);
                    if (ks < 2 * T3) {
#pragma unroll
                        for (int l = ks * LPK; l < (ks + 1) * LPK && l < KS; ++l) an[l] = a.afrag[FX_CHK((size_t)(q + 1) * KS * 64 + l * 64 + elane, a.lim_afrag, 3)];
                    }
                    if constexpr (decltype(flush)::value) {
                        if (ks >= T3 && ks < 2 * T3) {
#pragma unroll
                            for (int r = (ks - T3) * RPK; r < (ks - T3 + 1) * RPK && r < NRD; ++r) mixed_get(r, 16, slot);
                        } else if (ks >= 2 * T3) {
#pragma unroll
                            for (int r = (ks - 2 * T3) * SPK; r < (ks - 2 * T3 + 1) * SPK && r < NRD; ++r) image_out(r, prev_rowbase, 16);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(an[ks]));
            };
            // (`odd`: parity of the dof tile's first row tile -- the fragment buffers alternate per row tile, and with an
            // odd number of tables per dof tile (triangles) every other dof tile starts on the second buffer)
            auto dof_tile = [&](int i, auto flush, auto odd) {
#pragma unroll
                for (int t = 0; t < MIXT; ++t) {
                    if ((t + (decltype(odd)::value ? 1 : 0)) & 1) mix_stage(acc[t], i * MIXT + t, t, t * rows + 16 * (i - 1), fa1, fa0, flush);
                    else mix_stage(acc[t], i * MIXT + t, t, t * rows + 16 * (i - 1), fa0, fa1, flush);
                }
                wave_lds_fence();  // (the previous dof tile's images have been read)
                const int nrows = i == RTd - 1 ? rows_last : 16;
#pragma unroll
                for (int t = 0; t < MIXT; ++t)
#pragma unroll
                    for (int w = 0; w < NWR; ++w) image_put(acc[t], w, nrows, t * SLOT);
                wave_lds_fence();
            };
            dof_tile(0, std::false_type{}, std::false_type{});
            if constexpr (MIXT % 2 == 0) {
                for (int i = 1; i < RTd; ++i) dof_tile(i, std::true_type{}, std::false_type{});
            } else {
                int i = 1;
                for (; i + 1 < RTd; i += 2) {
                    dof_tile(i, std::true_type{}, std::true_type{});
                    dof_tile(i + 1, std::true_type{}, std::false_type{});
                }
                if (i < RTd) dof_tile(i, std::true_type{}, std::true_type{});
            }
            // the last dof tile's images
#pragma unroll
            for (int t = 0; t < MIXT; ++t) {
#pragma unroll
                for (int r = 0; r < NRD; ++r) mixed_get(r, rows_last, t);
#pragma unroll
                for (int r = 0; r < NRD; ++r) image_out(r, t * rows + 16 * (RTd - 1), rows_last);
            }
            wave_lds_fence();
        } else if constexpr (RTC > 0) {
            // register-resident fragments, fully unrolled; one accumulator set and no pipelining inside the wave:
            // these shapes run several waves per SIMD, which overlap each other's flushes
            v4d acc[CT];
#pragma unroll
            for (int t = 0; t < RTC; ++t) {
                const int nrows = t == RTC - 1 ? last_rows : 16;
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = v4d{0.0, 0.0, 0.0, 0.0};
                mfma_steps(acc, areg[t], 0, KS);
#pragma unroll
                for (int w = 0; w < NWR; ++w) image_put(acc, w, nrows);
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < NRD; ++r) image_get(r, nrows);
#pragma unroll
                for (int r = 0; r < NRD; ++r) image_out(r, 16 * t, nrows);
                wave_lds_fence();
            }
        } else {
            v4d accA[CT], accB[CT];
            {   // tile 0: nothing to flush yet
                const double* anp = ap + (size_t)KS * 64;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fa1[ks] = anp[FX_CHK(ks * 64 + (anp - a.afrag), a.lim_afrag, 3) - (anp - a.afrag)];
#pragma unroll
                for (int c = 0; c < CT; ++c) accA[c] = v4d{0.0, 0.0, 0.0, 0.0};
                mfma_steps(accA, fa0, 0, KS);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(fa1[ks]));
            }
            int rt = 1;
            for (; rt + 1 < RTn; rt += 2) {  // odd tiles: fragments fa1 -> accB, even tiles: fa0 -> accA
                stage(accB, accA, rt, fa1, fa0);
                stage(accA, accB, rt + 1, fa0, fa1);
            }
            if (rt < RTn) {  // odd number of remaining tiles: one more stage, the last tile ends up in accB
                stage(accB, accA, rt, fa1, fa0);
                flush_last(accB);
            } else {
                flush_last(accA);
            }
        }
        // first use of the prefetched points in the same block as the last stores: exact vmcnt
#pragma unroll
        for (int h = 0; h < PH; ++h)
#pragma unroll
            for (int d = 0; d < SD; ++d) asm volatile("" : "+v"(xnext[h][d]));
        grp = gnext;
    }
    wqueue.finish();
#if FX_DBG & 512
    if (lane == 0) {  // ablation build: lifetime of every wave (shader cycles, 100 MHz ticks)
        const long long gw = (long long)blockIdx.x * 4 + wave;
        if (gw < 3000) {
            trash[2048 + 2 * gw] = (double)(__builtin_readcyclecounter() - clk0);
            trash[2049 + 2 * gw] = (double)(__builtin_amdgcn_s_memrealtime() - rt0);
        }
    }
#endif
}

}  // namespace fxk
