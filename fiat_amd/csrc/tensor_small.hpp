// Lane-local tensor-product tabulation for small requests (gfx950): Q1 / Q2 quadrilaterals and hexahedra, Q3 / Q4
// quadrilaterals -- requests of a few hundred bytes to 16 KB, for which one workgroup per request (tensor_tabulate_kernel,
// aux_kernels.hpp) spends its time in barriers and in the factor tables of 4..27 points: 0.4-6 % of the HBM peak
// (tools/coverage_map_tensor.py).
//
// Reference behaviour: TensorProductElement.tabulate, FIAT/tensor_product.py:231-292 (scalar factors nested left to
// right), factors = 1-D Lagrange elements tabulated by barycentric interpolation (FIAT/barycentric_interpolation.py:22-93).
//
// A wave takes P = floor(64 / npts) requests at a time, lane <-> (request of the group, point).  The lane evaluates the NF
// factor bases (and their derivatives, dmat^k) at its own point into registers -- NF, the node count NN and the
// derivative order are template parameters, so every index below is a compile-time constant -- multiplies them out row
// by row into the wave's LDS image [request][table][basis function][point], and the group's contiguous output leaves as
// 16-byte pieces (8-byte pieces when a request is an odd number of doubles).  No workgroup barriers.
#pragma once
#include "aux_kernels.hpp"
#include "simplex_kernel.hpp"
#include "store.hpp"

namespace fxk {

// derivative multi-indices in mis() order (polynomial_set.py:23-32), at compile time
template <int NF, int ORDER> struct TensorAlpha {
    static constexpr int NTAB = NF == 2 ? (ORDER + 1) * (ORDER + 2) / 2 : (ORDER + 1) * (ORDER + 2) * (ORDER + 3) / 6;
    int a[NTAB][3];
    constexpr TensorAlpha() : a{} {
        int t = 0;
        for (int k = 0; k <= ORDER; ++k) {
            if (NF == 2) {
                for (int i = 0; i <= k; ++i) {
                    a[t][0] = k - i;
                    a[t][1] = i;
                    a[t][2] = 0;
                    ++t;
                }
            } else {
                for (int i = 0; i <= k; ++i)
                    for (int j = 0; j <= i; ++j) {
                        a[t][0] = k - i;
                        a[t][1] = i - j;
                        a[t][2] = j;
                        ++t;
                    }
            }
        }
    }
};

template <int NF, int NN, int ORDER, bool GRID>
__global__ __launch_bounds__(256) void tensor_small_kernel(const TensorArgs a, const int P, const int img_doubles) {
    static_assert(NF == 2 || NF == 3, "two or three interval factors");
    constexpr TensorAlpha<NF, ORDER> AL{};
    constexpr int NTAB = TensorAlpha<NF, ORDER>::NTAB;
    constexpr int K = ORDER + 1;
    constexpr int NDOF = NF == 2 ? NN * NN : NN * NN * NN;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    double* img = lds + (size_t)wave * img_doubles;
    const int npts = a.npts;
    const int total = NTAB * NDOF * npts;  // doubles per request
    const int g = idiv_small(lane, 1.0f / (float)npts);
    const int p = lane - g * npts;
    const bool lane_on = g < P;
    int j[3] = {p, p, p};
    if constexpr (GRID) {
        const int q = a.q;
        if constexpr (NF == 2) {
            j[0] = idiv_small(p, 1.0f / (float)q);
            j[1] = p - j[0] * q;
        } else {
            j[0] = idiv_small(p, 1.0f / (float)(q * q));
            const int rr = p - j[0] * q * q;
            j[1] = idiv_small(rr, 1.0f / (float)q);
            j[2] = rr - j[1] * q;
        }
    }
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    const long long nitems = (a.nreq + P - 1) / P;
    for (long long item = (long long)blockIdx.x * nw + wave; item < nitems; item += (long long)gridDim.x * nw) {
        const long long r0 = item * P;
        const bool on = lane_on && r0 + g < a.nreq;
        const long long r = on ? r0 + g : r0;  // (idle lanes recompute a valid point)
        // factor bases at this lane's point: T[f][k][i] = k-th derivative of basis function i of factor f
        double T[NF][K][NN];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const double x = GRID ? a.pts[((size_t)r * NF + f) * a.q + j[f]] : a.pts[((size_t)r * npts + p) * NF + f];
            lagrange_values_n<NN>(a.L[f], x, T[f][0]);
#pragma unroll
            for (int k = 1; k < K; ++k) lagrange_diff_n<NN>(a.L[f], T[f][k - 1], T[f][k]);
        }
        if (on) {
            double* o = img + g * total + p;
#pragma unroll
            for (int t = 0; t < NTAB; ++t) {
#pragma unroll
                for (int i0 = 0; i0 < NN; ++i0) {
#pragma unroll
                    for (int i1 = 0; i1 < NN; ++i1) {
                        const double v01 = T[0][AL.a[t][0]][i0] * T[1][AL.a[t][1]][i1];
                        if constexpr (NF == 2) {
                            o[(t * NDOF + i0 * NN + i1) * npts] = v01;
                        } else {
#pragma unroll
                            for (int i2 = 0; i2 < NN; ++i2)
                                o[(t * NDOF + (i0 * NN + i1) * NN + i2) * npts] = v01 * T[2][AL.a[t][2]][i2];
                        }
                    }
                }
            }
        }
        wave_lds_fence();
        // the group's requests are contiguous in HBM
        const int nr = (int)min((long long)P, a.nreq - r0);
        const int n = nr * total;
        double* dst = a.out + (size_t)r0 * total;
        if ((total & 1) == 0) {
            const v2d_t* s2 = reinterpret_cast<const v2d_t*>(img);
            v2d_t* d2 = reinterpret_cast<v2d_t*>(dst);
            flush_block(d2, s2, n >> 1, lane);  // whole-line non-temporal body, plain partial edges (store.hpp)
        } else {
            for (int c = lane; c < n; c += 64) dst[c] = img[c];
        }
        wave_lds_fence();  // the next group overwrites the image
    }
}

}  // namespace fxk
