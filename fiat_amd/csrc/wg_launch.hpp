// Interface between api.hip and wg.hip (the request-per-workgroup kernel's translation unit).
#pragma once
#include <hip/hip_runtime.h>

namespace fxk {
template <int NC> struct StackedArgs;
}

namespace fxwg {
// dynamic LDS of the <sd, n, ct> instance; whether it exists (4 <= ct <= 8 column tiles, slab + images within 160 KB)
int lds_bytes(int sd, int n, int ct);
bool has_instance(int sd, int n, int ct, bool odd);
// one launch: `grid` persistent workgroups of 256 threads, requests handed out through queue[0] (zero before the launch,
// zero again after it), queue[1] = finished workgroups
hipError_t launch_simplex_wg(int sd, int n, int ct, bool odd, const fxk::StackedArgs<0>& head, const double* coef, int ncoef, int lds_bytes,
                             int grid, double* trash, unsigned int* queue, hipStream_t s);
}  // namespace fxwg
