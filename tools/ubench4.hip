// Work-queue atomics microbenchmark (measurement tooling): rate of wave-level
// atomicAdd claims on 1 / 8 / 256 counters with 2048 persistent waves.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void claim(unsigned long long* ctr, int ngroups, long long total_per_group, long long* sink) {
    const int wave = threadIdx.x >> 6;
    unsigned long long* c = ctr + (blockIdx.x % ngroups) * 16;  // 128 B apart
    long long n = 0;
    while (true) {
        unsigned long long r = 0;
        if ((threadIdx.x & 63) == 0) r = atomicAdd(c, 1ULL);
        r = __shfl(r, 0);
        if ((long long)r >= total_per_group) break;
        ++n;
    }
    if ((threadIdx.x & 63) == 0) sink[blockIdx.x * 4 + wave] = n;
}
int main() {
    unsigned long long* ctr; long long* sink;
    hipMalloc(&ctr, 256 * 128); hipMalloc(&sink, 4096 * 8);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    const long long total = 1000000;
    for (int ng : {1, 8, 256}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(ctr, 0, 256 * 128);
            hipEventRecord(t0);
            claim<<<512, 256>>>(ctr, ng, total / ng, sink);
            hipEventRecord(t1); hipEventSynchronize(t1);
            float ms; hipEventElapsedTime(&ms, t0, t1);
            printf("%3d counters: %lld claims by 2048 waves in %.1f us -> %.1f ns per claim\n", ng, total, ms * 1e3, ms * 1e6 / total);
        }
    }
    return 0;
}
