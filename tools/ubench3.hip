// Store-pattern microbenchmark (measurement tooling): how fast can persistent waves write
// 1.47 GB in per-wave contiguous chunks (the epilogue pattern of the tabulation kernels)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));

// each wave writes chunk `c` (chunk_b bytes, 16-byte multiples) with 16 B per lane
__global__ __launch_bounds__(256) void store_chunks(double* out, long long nchunks, int chunk16, int mode) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    v2d val = {1.0 + lane, 2.0};
    if (mode == 3 || mode == 4) {  // like 2, but every chunk is first written to LDS and read back (the kernels' epilogue)
        extern __shared__ double lds[];
        double* img = lds + (threadIdx.x >> 6) * 1024;
        const int nit = (chunk16 + 63) / 64;
        for (long long c = wave; c < nchunks; c += nw) {
            v2d* g = reinterpret_cast<v2d*>(out) + c * chunk16;
            const int nw8 = mode == 3 ? 15 : 30;
            for (int k = 0; k < nw8; ++k) img[(k * 64 + lane) & 1023] = val[0] + k;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            v2d buf[8];
            for (int it = 0; it < 8; ++it) buf[it] = reinterpret_cast<v2d*>(img)[(it * 64 + lane) & 511];
            for (int it = 0; it < nit && it < 8; ++it) {
                int i = it * 64 + lane;
                g[i < chunk16 ? i : chunk16 - 1] = buf[it];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
        }
    } else if (mode == 2) {  // wave-strided, fixed trip count, lanes past the end rewrite the last 16 bytes
        const int nit = (chunk16 + 63) / 64;
        for (long long c = wave; c < nchunks; c += nw) {
            v2d* g = reinterpret_cast<v2d*>(out) + c * chunk16;
            for (int it = 0; it < nit; ++it) {
                int i = it * 64 + lane;
                g[i < chunk16 ? i : chunk16 - 1] = val;
            }
        }
    } else if (mode == 0) {  // wave-strided: wave w handles chunks w, w+nw, ...
        for (long long c = wave; c < nchunks; c += nw) {
            v2d* g = reinterpret_cast<v2d*>(out) + c * chunk16;
            for (int i = lane; i < chunk16; i += 64) g[i] = val;
        }
    } else {  // blocked: wave w handles a contiguous range of chunks
        long long per = (nchunks + nw - 1) / nw;
        long long c0 = wave * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
        for (long long c = c0; c < c1; ++c) {
            v2d* g = reinterpret_cast<v2d*>(out) + c * chunk16;
            for (int i = lane; i < chunk16; i += 64) g[i] = val;
        }
    }
}

int main() {
    const long long total = 100000LL * 14720;  // bytes
    double* out;
    hipMalloc(&out, total + 65536);
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    int chunks[] = {7360};
    int wgs[] = {2, 4};
    for (int rep = 0; rep < 2; ++rep) for (int mode = 0; mode < 5; mode += (mode == 0 ? 2 : 1))
        for (int cb : chunks)
            for (int wg : wgs) {
                long long nchunks = total / cb;
                int grid = 256 * wg;
                store_chunks<<<grid, 256, (mode >= 3 ? 4 * 1024 * 8 : 0)>>>(out, nchunks, cb / 16, mode);
                hipDeviceSynchronize();
                hipEventRecord(t0);
                for (int r = 0; r < 10; ++r) store_chunks<<<grid, 256, (mode >= 3 ? 4 * 1024 * 8 : 0)>>>(out, nchunks, cb / 16, mode);
                hipEventRecord(t1);
                hipEventSynchronize(t1);
                float ms; hipEventElapsedTime(&ms, t0, t1);
                ms /= 10;
                printf("mode %d (%s) chunk %6d B  waves/CU %2d : %7.1f us  %6.0f GB/s\n", mode, mode == 2 ? "clamped" : mode == 3 ? "lds15" : "lds30", cb, wg * 4,
                       ms * 1e3, nchunks * (double)cb / ms / 1e6);
            }
    return 0;
}
