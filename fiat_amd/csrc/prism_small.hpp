// Prisms: (element on a triangle) x (1-D Lagrange element), lane-local and fused (gfx950).
//
// Reference behaviour: TensorProductElement.tabulate, FIAT/tensor_product.py:231-336 -- the table of the derivative
// alpha = (alpha_A, alpha_B) is the per-point product of A's table alpha_A and B's table alpha_B, basis function (a, b) has
// index a * dim(B) + b, the value component rides on the vector-valued factor (:293-317: vector A x scalar B).
//
// The general route (api.hip, fx_table_outer_batch) tabulates both factors with their own kernels and multiplies the
// tables in a third pass: the factor tables are written and read back, three launches and a column split of the points --
// 21-33 % of the HBM peak on P2 x P1 / P3 x P2 / RT1 x DG0 prisms (tools/bench_prism.py).  Here one wave takes
// P = floor(64 / npts) requests, lane <-> (request, point): the lane runs the triangle factor's Dubiner recurrence and
// contraction as tabulate_simplex_small does (every member in registers, coefficient rows as scalar loads), evaluates the
// interval factor by barycentric interpolation (compile-time node count), multiplies them out row by row into the wave's
// LDS image of the P requests and streams the image out as 16-byte pieces.
#pragma once
#include "aux_kernels.hpp"
#include "simplex_small.hpp"
#include "store.hpp"

namespace fxk {

struct PrismArgs {
    const double* pts;   // [nreq][npts][3]: (x, y) in the triangle factor's cell, z in the interval
    double* out;         // [nreq][ntab][rowsA / vdimA * NN][vdimA][npts]
    const double* cmat;  // [rowsA][nexp] coefficient matrix of the triangle factor (C0 transform folded in), device
    double coef[3 * SMALL_MAXSTEPS];
    double phi0;
    double A0[4];  // triangle factor: its cell -> default simplex
    double b0[2];
    LineDesc L;    // interval factor
    long long nreq, nitems;
    int npts, rowsA, vdimA;
    int P;              // whole requests per wave item
    int stage_doubles;  // per-wave LDS doubles
};

// tables of the product in mis(3, k) order -> (table of the triangle factor in mis(2, .) order, derivative of the interval factor)
template <int ORDER> struct PrismTables {
    static constexpr int NTAB = (ORDER + 1) * (ORDER + 2) * (ORDER + 3) / 6;
    int ta[NTAB], tz[NTAB];
    constexpr PrismTables() : ta{}, tz{} {
        int t = 0;
        for (int k = 0; k <= ORDER; ++k)
            for (int i = 0; i <= k; ++i)
                for (int j = 0; j <= i; ++j) {  // alpha = (k - i, i - j, j)
                    const int m = k - j;        // order of the triangle factor's table (k - i, i - j)
                    ta[t] = m * (m + 1) / 2 + (i - j);
                    tz[t] = j;
                    ++t;
                }
    }
};

template <int N, int NN, int ORDER, int NW>
__global__ __launch_bounds__(64 * NW) void prism_small_kernel(const PrismArgs a) {
    constexpr int SD = 2;
    constexpr int NTA = NTab<SD, ORDER>::value;
    constexpr PrismTables<ORDER> PT{};
    constexpr int NTAB = PrismTables<ORDER>::NTAB;
    constexpr int K = ORDER + 1;
    constexpr StepTable<SD, N> TBL{};
    constexpr int NEXP = StepTable<SD, N>::NEXP;
    static_assert(NEXP - 1 <= SMALL_MAXSTEPS, "step table too long for PrismArgs");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double* stage = lds + (size_t)wave * a.stage_doubles;
    const int npts = a.npts, rowsA = a.rowsA, vdim = a.vdimA;
    const int table = rowsA * NN * npts;  // doubles per output table
    const long long reqsize = (long long)NTAB * table;
    const int rl = idiv_small(lane, 1.0f / (float)npts);
    const int pl = lane - rl * npts;
    typedef const __attribute__((address_space(4))) double CDouble;
    const __attribute__((address_space(4))) char* kargs =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    CDouble* kcoef = (CDouble*)(kargs + __builtin_offsetof(PrismArgs, coef));
    CDouble* kcmat = (CDouble*)(unsigned long long)a.cmat;  // never written while the kernel runs: scalar loads

    for (long long item = (long long)blockIdx.x * NW + wave; item < a.nitems; item += (long long)gridDim.x * NW) {
        const long long r0 = item * a.P;
        const long long left = a.nreq - r0;
        const int Pcur = left < a.P ? (int)left : a.P;
        const bool active = rl < Pcur;
        const long long req = r0 + (active ? rl : 0);
        const double* pp = a.pts + ((size_t)req * npts + (active ? pl : 0)) * 3;
        const double x0 = pp[0], x1 = pp[1], z = pp[2];

        // ---- triangle factor: reference coordinates, recurrence with every member in registers ----
        double X[SD], J[SD][SD];
#pragma unroll
        for (int i = 0; i < SD; ++i) {
            J[i][0] = a.A0[i * SD];
            J[i][1] = a.A0[i * SD + 1];
            X[i] = a.b0[i] + J[i][0] * x0 + J[i][1] * x1;
        }
        Jet<SD, ORDER> mem[NEXP];
        Jet<SD, ORDER> zero;
        jet_zero(zero);
        jet_zero(mem[0]);
        mem[0].v = a.phi0;
        {
            Factors<SD, ORDER> F;
            int fcodim = -1;
#pragma unroll
            for (int s = 0; s < NEXP - 1; ++s) {
                if (TBL.codim[s] != fcodim) {
                    fcodim = TBL.codim[s];
                    make_factors<SD, ORDER>(F, fcodim, X, J);
                }
                apply_step<SD, ORDER>(mem[TBL.dst[s]], mem[TBL.cur[s]], TBL.prv[s] < 0 ? zero : mem[TBL.prv[s]], F,
                                      kcoef[3 * s], kcoef[3 * s + 1], kcoef[3 * s + 2]);
            }
        }
        // ---- interval factor: TB[k][b] = k-th derivative of basis function b at z ----
        double TB[K][NN];
        lagrange_values_n<NN>(a.L, z, TB[0]);
#pragma unroll
        for (int k = 1; k < K; ++k) lagrange_diff_n<NN>(a.L, TB[k - 1], TB[k]);

        // ---- rows of the triangle factor x basis functions of the interval factor -> LDS image ----
        double* sp = stage + (size_t)(active ? rl : 0) * reqsize + (active ? pl : 0);
        int dofA = 0, comp = 0;
        for (int row = 0; row < rowsA; ++row) {
            CDouble* crow = kcmat + (size_t)row * NEXP;
            double acc[NTA];
#pragma unroll
            for (int t = 0; t < NTA; ++t) acc[t] = 0.0;
#pragma unroll
            for (int k = 0; k < NEXP; ++k) {
                const double c = crow[k];
                acc[0] += c * mem[k].v;
                if constexpr (ORDER >= 1) {
#pragma unroll
                    for (int d = 0; d < SD; ++d) acc[1 + d] += c * mem[k].g[d];
                }
                if constexpr (ORDER >= 2) {
#pragma unroll
                    for (int h = 0; h < SD * (SD + 1) / 2; ++h) acc[1 + SD + h] += c * mem[k].h[h];
                }
            }
            if (active) {
#pragma unroll
                for (int b = 0; b < NN; ++b) {
                    const int orow = (dofA * NN + b) * vdim + comp;  // (a, b) -> a * dim(B) + b, component innermost
#pragma unroll
                    for (int t = 0; t < NTAB; ++t) sp[(size_t)t * table + orow * npts] = acc[PT.ta[t]] * TB[PT.tz[t]][b];
                }
            }
            if (++comp == vdim) {
                comp = 0;
                ++dofA;
            }
        }
        wave_lds_fence();

        // ---- image -> HBM: P whole requests, contiguous ----
        {
            const long long total = (long long)Pcur * reqsize;
            double* gout = a.out + (size_t)r0 * reqsize;
            if ((reqsize & 1) == 0) {
                const v2d* s2 = reinterpret_cast<const v2d*>(stage);
                v2d* g2 = reinterpret_cast<v2d*>(gout);
                flush_block(g2, s2, (int)(total >> 1), lane);  // whole-line non-temporal body, plain partial edges (store.hpp)
            } else {
                for (long long i = lane; i < total; i += 64) gout[i] = stage[i];
            }
        }
        wave_lds_fence();  // the next item overwrites the image
    }
}

}  // namespace fxk
