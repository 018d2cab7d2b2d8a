// Store-pattern microbenchmark (measurement tooling): what limits the epilogue of the one-request-per-wave kernels?
// Writes nreq blocks of `bytes` bytes (16-byte multiples) with 16 B per lane, persistent workgroups of NW waves:
//   pattern 0: one wave per block (the kernels' pattern: 8 waves of a CU write 8 different blocks at a time)
//   pattern 1: the NW waves of a workgroup write ONE block together (wave k takes the 1 KB pieces k, k + NW, ...)
//   pattern 2: like 0, but waves of a workgroup take ADJACENT blocks in lock step (pieces interleaved in time)
// nt = 1: non-temporal stores.  Prints us and GB/s; compare with hipMemsetAsync of the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

// NT: 0 plain, 1 non-temporal, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt (cache-policy bits of global_store_dwordx4)
template <int NT>
__device__ __forceinline__ void st(v2d* p, v2d v) {
    if (NT == 0) *p = v;
    else if (NT == 1) __builtin_nontemporal_store(v, p);
    else if (NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    else if (NT == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
    else if (NT == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
}

template <int NT>
__global__ __launch_bounds__(512) void store_blocks(double* out, long long nblk, int blk16, int stride16, int pattern) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v2d val = {1.0 + lane, 2.0};
    v2d* base = reinterpret_cast<v2d*>(out);
    if (pattern == 2) {  // memset-like: every wave of the grid takes 1 KB pieces with a grid stride, block boundaries ignored
        const long long gw = (long long)blockIdx.x * nw + wv, tw = (long long)gridDim.x * nw;
        const long long total16 = nblk * (long long)stride16;
        for (long long i = gw * 64 + lane; i < total16; i += tw * 64) st<NT>(base + i, val);
    } else if (pattern == 3) {  // like 0, but 4 KB per wave and visit: the wave's block advances in 4 KB steps, blocks of the
                                // grid interleaved (piece k of every block before piece k + 1 of any) -- same window, other order
        const long long gw = (long long)blockIdx.x * nw + wv, tw = (long long)gridDim.x * nw;
        for (long long b0 = 0; b0 < nblk; b0 += tw) {
            const long long b = b0 + gw;
            if (b >= nblk) break;
            v2d* g = base + b * stride16;
            for (int i = lane; i < blk16; i += 64) st<NT>(g + i, val);
            __builtin_amdgcn_s_sleep(0);
        }
    } else if (pattern == 7 || pattern == 8) {  // as 0, but every wave starts at a different 1 KB piece of its block and wraps around
        const long long gw = (long long)blockIdx.x * nw + wv, tw = (long long)gridDim.x * nw;
        const int npieces = (blk16 + 63) / 64;
        const int rot = (int)((pattern == 7 ? gw : gw * 13) % npieces);
        for (long long b = gw; b < nblk; b += tw) {
            v2d* g = base + b * stride16;
            for (int j = 0; j < npieces; ++j) {
                int pc = j + rot;
                if (pc >= npieces) pc -= npieces;
                const int i = pc * 64 + lane;
                if (i < blk16) st<NT>(g + i, val);
            }
        }
    } else if (pattern == 9) {  // as 0, but every store instruction covers WHOLE 128-byte lines: lane i of instruction `it`
                                // writes chunk 64 it + i - s, s = 16-byte chunks between the last line boundary and the block
        const long long gw = (long long)blockIdx.x * nw + wv, tw = (long long)gridDim.x * nw;
        const int nit = (blk16 + 7 + 63) / 64;
        for (long long b = gw; b < nblk; b += tw) {
            v2d* g = base + b * stride16;
            const int sft = (int)((reinterpret_cast<unsigned long long>(g) >> 4) & 7);
            for (int it = 0; it < nit; ++it) {
                int i = it * 64 + lane - sft;
                i = i < 0 ? 0 : (i >= blk16 ? blk16 - 1 : i);
                st<NT>(g + i, val);
            }
        }
    } else if (pattern >= 20 && pattern < 30) {  // as 0, but at most (pattern - 20) KB... of stores in flight per wave: wait after every k stores
        const long long gw = (long long)blockIdx.x * nw + wv, tw = (long long)gridDim.x * nw;
        const int k = 1 << (pattern - 20);
        for (long long b = gw; b < nblk; b += tw) {
            v2d* g = base + b * stride16;
            int cnt = 0;
            for (int i = lane; i < blk16; i += 64) {
                st<NT>(g + i, val);
                if (++cnt == k) {
                    cnt = 0;
                    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0) only
                }
            }
        }
    } else if (pattern >= 40 && pattern < 60) {  // as 0, but the 1 KB pieces of a block are visited with a stride of (pattern - 40):
                                                 // consecutive store instructions go to different 4 KB / 8 KB ... stripes
        const long long gw = (long long)blockIdx.x * nw + wv, tw = (long long)gridDim.x * nw;
        const int npieces = (blk16 + 63) / 64, strd = pattern - 40;
        for (long long b = gw; b < nblk; b += tw) {
            v2d* g = base + b * stride16;
            for (int r = 0; r < strd; ++r)
                for (int pc = r; pc < npieces; pc += strd) {
                    const int i = pc * 64 + lane;
                    if (i < blk16) st<NT>(g + i, val);
                }
        }
    } else if (pattern >= 200 && pattern < 300) {  // as 0, but every wave starts its block at the piece that contains the first
                                                    // (pattern - 200) KB-aligned address inside the block and wraps around: waves in
                                                    // lock step then write the same phase of that period at the same time
        const long long gw = (long long)blockIdx.x * nw + wv, tw = (long long)gridDim.x * nw;
        const int npieces = (blk16 + 63) / 64;
        const unsigned long long P = (unsigned long long)(pattern - 200) * 1024ull;
        for (long long b = gw; b < nblk; b += tw) {
            v2d* g = base + b * stride16;
            const unsigned long long a0 = reinterpret_cast<unsigned long long>(g);
            const unsigned long long to_boundary = (P - a0 % P) % P;          // bytes from the block start to the next multiple of P
            int rot = (int)(to_boundary >> 10);
            if (rot >= npieces) rot = 0;
            for (int j = 0; j < npieces; ++j) {
                int pc = j + rot;
                if (pc >= npieces) pc -= npieces;
                const int i = pc * 64 + lane;
                if (i < blk16) st<NT>(g + i, val);
            }
        }
    } else if (pattern == 5) {  // as 4, but the waves of a workgroup store ONE AFTER THE OTHER: a workgroup writes its run of NW
                                // adjacent blocks as one sequential stream (256 streams on the chip instead of 2048)
        const long long ngroups = (nblk + nw - 1) / nw;
        for (long long gidx = blockIdx.x; gidx < ngroups; gidx += gridDim.x) {
            const long long b = gidx * nw + wv;
            for (int turn = 0; turn < nw; ++turn) {
                if (turn == wv && b < nblk) {
                    v2d* g = base + b * stride16;
                    for (int i = lane; i < blk16; i += 64) st<NT>(g + i, val);
                }
                __syncthreads();
            }
        }
    } else if (pattern >= 100 && pattern < 200) {  // segments of (pattern - 100) KB: wave w writes segments w, w + W, ... of the whole output
        const long long gw = (long long)blockIdx.x * nw + wv, tw = (long long)gridDim.x * nw;
        const long long seg16 = (long long)(pattern - 100) * 64, total16 = nblk * (long long)stride16;
        for (long long s0 = gw * seg16; s0 < total16; s0 += tw * seg16)
            for (long long i = s0 + lane; i < s0 + seg16 && i < total16; i += 64) st<NT>(base + i, val);
    } else if (pattern == 4) {  // one wave per block, blocks assigned in contiguous runs per WORKGROUP of 8 (dynamic-queue order)
        const long long ngroups = (nblk + nw - 1) / nw;
        for (long long gidx = blockIdx.x; gidx < ngroups; gidx += gridDim.x) {
            const long long b = gidx * nw + wv;
            if (b >= nblk) continue;
            v2d* g = base + b * stride16;
            for (int i = lane; i < blk16; i += 64) st<NT>(g + i, val);
        }
    } else if (pattern == 1) {
        for (long long b = blockIdx.x; b < nblk; b += gridDim.x) {
            v2d* g = base + b * stride16;
            for (int i = wv * 64 + lane; i < blk16; i += nw * 64) st<NT>(g + i, val);
        }
    } else {
        const long long gw = (long long)blockIdx.x * nw + wv, tw = (long long)gridDim.x * nw;
        for (long long b = gw; b < nblk; b += tw) {
            v2d* g = base + b * stride16;
            for (int i = lane; i < blk16; i += 64) st<NT>(g + i, val);
        }
    }
}

int main(int argc, char** argv) {
    const long long nblk = argc > 1 ? atoll(argv[1]) : 25000;
    const int bytes = argc > 2 ? atoi(argv[2]) : 33120;
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    double* out;
    const long long cap = nblk * (long long)(bytes + 256) + 65536;
    hipMalloc(&out, cap);
    auto timeit = [&](auto fn) {
        fn(); hipDeviceSynchronize();
        hipEventRecord(t0);
        for (int r = 0; r < 20; ++r) fn();
        hipEventRecord(t1); hipEventSynchronize(t1);
        float ms; hipEventElapsedTime(&ms, t0, t1);
        return ms / 20 * 1e3;
    };
    const double total = (double)nblk * bytes;
    double us = timeit([&] { hipMemsetAsync(out, 0, (size_t)total, 0); });
    printf("memset of the same bytes: %7.1f us %6.0f GB/s\n", us, total / us / 1e3);
    const bool quick = argc > 3;
    std::vector<int> pats = {0, 1, 2};
    if (argc > 4) {
        pats.clear();
        for (char* tok = strtok(argv[4], ","); tok; tok = strtok(nullptr, ",")) pats.push_back(atoi(tok));
    }
    for (int pad : {0}) {
        const int stride = pad ? (bytes + 127) / 128 * 128 : bytes;
        for (int nt : {0, 1, 2, 3, 4, 5})
            for (int nw : {4, 8})
                for (int wgs : {1, 2, 4})
                    for (int pattern : pats) {
                        if (nw * wgs > 16) continue;
                        if (quick && (wgs != 1)) continue;
                        const int grid = 256 * wgs;
                        auto fn = [&] {
                            switch (nt) {
                                case 0: store_blocks<0><<<grid, 64 * nw>>>(out, nblk, bytes / 16, stride / 16, pattern); break;
                                case 1: store_blocks<1><<<grid, 64 * nw>>>(out, nblk, bytes / 16, stride / 16, pattern); break;
                                case 2: store_blocks<2><<<grid, 64 * nw>>>(out, nblk, bytes / 16, stride / 16, pattern); break;
                                case 3: store_blocks<3><<<grid, 64 * nw>>>(out, nblk, bytes / 16, stride / 16, pattern); break;
                                case 4: store_blocks<4><<<grid, 64 * nw>>>(out, nblk, bytes / 16, stride / 16, pattern); break;
                                default: store_blocks<5><<<grid, 64 * nw>>>(out, nblk, bytes / 16, stride / 16, pattern); break;
                            }
                        };
                        us = timeit(fn);
                        printf("block %6d B stride %6d nt %d waves/WG %d WG/CU %d pattern %d : %7.1f us %6.0f GB/s\n", bytes, stride, nt, nw, wgs, pattern, us, total / us / 1e3);
                    }
    }
    return 0;
}
