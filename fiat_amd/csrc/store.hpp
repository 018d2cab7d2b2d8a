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

// One wave copies a contiguous block of `nch` 16-byte chunks from its LDS image to global memory.  The block (an item of
// whole requests) starts anywhere in a 128-byte line.  The body goes out as non-temporal stores that cover WHOLE lines
// (lane i of an instruction takes chunk head + 64 it + i: every instruction starts on a line boundary); the partial first
// and last line of the block are shared with the neighbouring blocks, which other waves write at other times, and go out as
// PLAIN masked stores: a non-temporal partial write leaves the L2 at once and the two parts of the line reach memory as
// separate masked writes, a plain one waits in the L2 for its other part.  Round 3, lane-local kernel, 50 shapes, interleaved
// A/B (tools/map_ab.sh): geometric mean of the launch times 0.956 of the plain loop over all chunks, best 0.81, worst 1.04;
// all-plain stores: 1.18.  (RT2 requests, which are not line multiples either: simplex_pair.hpp)
template <class V2> __device__ __forceinline__ void flush_block(V2* g2, const V2* s2, int nch, int lane) {
    const int shift = (int)((reinterpret_cast<unsigned long long>(g2) >> 4) & 7ull);
    const int head = min(nch, (8 - shift) & 7);
    const int body_end = head + ((nch - head) & ~7);
    if (lane < head) g2[lane] = s2[lane];
    for (int i = head + lane; i < body_end; i += 64) stream_store(&g2[i], s2[i]);
    if (body_end + lane < nch) g2[body_end + lane] = s2[body_end + lane];
}

}  // namespace fxk
