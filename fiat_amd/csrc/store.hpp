// Output stores of the tabulation kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include "ablation.hpp"

#ifndef FX_NT_STORES
#define FX_NT_STORES 1
#endif

namespace fxk {

// Output tables are written once and read by later kernels: non-temporal stores let the L2
// stream them out instead of holding them as dirty lines until an eviction is forced
// (measured on the P3 tet benchmark, K-streamed kernel, same box and run, interleaved:
// 330 -> 301 us per 100 000 requests).
template <class T> __device__ __forceinline__ void stream_store(T* p, const T& v) {
#if FX_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

}  // namespace fxk
