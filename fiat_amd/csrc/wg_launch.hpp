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
// ... and with the order-1 chain rule of per-request cells on the accumulators (mix = 1: dof-major fragment buffer, A0inv set)
bool has_mix_instance(int sd, int n, int ct, bool odd);
// column tiles of the chain-rule instance that holds ctn column tiles of points (0: none)
int mix_ct(int sd, int n, int ctn);
// persistent workgroups a CU holds of the instance that takes requests of rt row tiles (2: the one-row-tile instances)
int workgroups_per_cu(int sd, int n, int ct, bool odd, int mix, int rt);
// one launch: `grid` persistent workgroups of 256 threads, requests handed out through queue[0] (zero before the launch,
// zero again after it), queue[1] = finished workgroups
hipError_t launch_simplex_wg(int sd, int n, int ct, bool odd, int mix, const fxk::StackedArgs<0>& head, const double* coef, int ncoef, int lds_bytes,
                             int grid, double* trash, unsigned int* queue, hipStream_t s);
}  // namespace fxwg
