// Table post-processing for the FInAT side of the boundary:
//  * classify_tables_kernel — the facts finat/fiat_elements.py:92-111 asserts about a tabulation
//    before it becomes a GEM literal (derivative == degree: constant over the points;
//    derivative > degree: zero), computed for a whole batch of GPU-resident tables;
//  * point_major_kernel — [rows][npts] tables to the [npts][rows] layout FInAT passes as kernel
//    arguments (finat/runtime_tabulated.py:79 `shape = point extents + index_shape + value_shape`).
// Both are single HBM passes over the tables.
#pragma once
#include <hip/hip_runtime.h>

namespace fxk {

struct ClassifyArgs {
    const double* tables;  // [ntables][rows][npts]
    double* stats;         // [ntables][2]: max |x|, max (|x - x[.., 0]| - rtol |x[.., 0]|)
    int rows, npts;
    double rtol;
};

// one workgroup per table
__global__ __launch_bounds__(256) void classify_tables_kernel(const ClassifyArgs a) {
    const double* t = a.tables + (size_t)blockIdx.x * a.rows * a.npts;
    const int n = a.rows * a.npts;
    double amax = 0.0, dev = -1.0e300;
    bool bad = false;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double x = t[i];
        const double x0 = t[i - i % a.npts];
        bad |= !(x == x);  // NaN
        amax = fmax(amax, fabs(x));
        dev = fmax(dev, fabs(x - x0) - a.rtol * fabs(x0));
    }
    __shared__ double s0[256], s1[256];
    __shared__ int sbad;
    if (threadIdx.x == 0) sbad = 0;
    __syncthreads();
    if (bad) atomicOr(&sbad, 1);
    s0[threadIdx.x] = amax;
    s1[threadIdx.x] = dev;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s0[threadIdx.x] = fmax(s0[threadIdx.x], s0[threadIdx.x + w]);
            s1[threadIdx.x] = fmax(s1[threadIdx.x], s1[threadIdx.x + w]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double nan = __builtin_nan("");
        a.stats[2 * (size_t)blockIdx.x + 0] = sbad ? nan : s0[0];
        a.stats[2 * (size_t)blockIdx.x + 1] = sbad ? nan : s1[0];
    }
}

struct PointMajorArgs {
    const double* in;  // [ntables][rows][npts]
    double* out;       // [ntables][npts][rows]
    int rows, npts;
};

// one workgroup per table and 32 x 32 tile; LDS tile padded against bank conflicts
__global__ __launch_bounds__(256) void point_major_kernel(const PointMajorArgs a) {
    __shared__ double tile[32][33];
    const int tr = (a.rows + 31) / 32, tp = (a.npts + 31) / 32;
    const size_t tab = blockIdx.x / (tr * tp);
    const int rem = blockIdx.x % (tr * tp);
    const int r0 = (rem / tp) * 32, p0 = (rem % tp) * 32;
    const double* in = a.in + tab * a.rows * a.npts;
    double* out = a.out + tab * a.rows * a.npts;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, p = p0 + tx;
        if (r < a.rows && p < a.npts) tile[j][tx] = in[(size_t)r * a.npts + p];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, r = r0 + tx;
        if (r < a.rows && p < a.npts) out[(size_t)p * a.rows + r] = tile[tx][j];
    }
}

}  // namespace fxk
