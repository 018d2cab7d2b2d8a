// Store cache-policy microbenchmark (measurement tooling): per-wave 7360-byte chunks written with
// plain / nt / sc1 / sc0 sc1 / sc0 sc1 nt global_store_dwordx4, 8 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int POLICY> __device__ __forceinline__ void st(v2d* p, v2d v) {
    if constexpr (POLICY == 0) *p = v;
    else if constexpr (POLICY == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}

template <int POLICY> __global__ __launch_bounds__(256) void store_chunks(double* out, long long nchunks, int chunk16) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    v2d val = {1.0 + lane, 2.0};
    for (long long c = wave; c < nchunks; c += nw) {
        v2d* g = reinterpret_cast<v2d*>(out) + c * chunk16;
        for (int i = lane; i < chunk16; i += 64) st<POLICY>(&g[i], val);
    }
}

template <int POLICY> void run(const char* name, double* out, long long total) {
    const int cb = 7360;
    long long nchunks = total / cb;
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    store_chunks<POLICY><<<512, 256>>>(out, nchunks, cb / 16);
    hipDeviceSynchronize();
    hipEventRecord(t0);
    for (int r = 0; r < 10; ++r) store_chunks<POLICY><<<512, 256>>>(out, nchunks, cb / 16);
    hipEventRecord(t1);
    hipEventSynchronize(t1);
    float ms; hipEventElapsedTime(&ms, t0, t1);
    ms /= 10;
    printf("%-14s %7.1f us  %6.0f GB/s\n", name, ms * 1e3, nchunks * (double)cb / ms / 1e6);
}

int main() {
    const long long total = 100000LL * 14720;
    double* out;
    hipMalloc(&out, total + 65536);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("plain", out, total);
        run<1>("nt", out, total);
        run<2>("sc1", out, total);
        run<5>("sc0", out, total);
        run<3>("sc0 sc1", out, total);
        run<4>("sc0 sc1 nt", out, total);
    }
    return 0;
}
