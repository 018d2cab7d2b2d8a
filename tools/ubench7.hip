// Store-pattern microbenchmark, round 3 (measurement tooling): can an epilogue that drains ALIGNED windows of the
// output cross the ~5.5 TB/s plateau of the one-request-block-per-wave epilogues (tools/ubench6.hip)?
//
// Unlike ubench6 the waves are de-synchronised the way the real kernels are: between two bursts of stores a wave
// "computes" for `think` ns (s_memrealtime, +-25 % per wave and burst), so that the store phases of the 2048 waves are
// random instead of in lock step.  The output (total bytes fixed) is cut into units of `unit` bytes; a unit is written
// by ONE wave as a run of 1 KB store instructions (16 B per lane), or by the 8 waves of a workgroup together (coop).
//   order 0  R: unit u -> wave (u mod TW), round (u div TW)            [global round robin: dense at coarse scale]
//   order 1  C: runs of NW consecutive units per workgroup, runs round robin over the workgroups [dynamic-queue order]
//   order 2  S: workgroup w owns the contiguous range of units [w U/G, (w+1) U/G), its waves take them in turn
//   order 3  S-coop: as 2, but the NW waves of the workgroup write every unit together (1 KB pieces interleaved)
//   order 4  R-coop: units round robin over the workgroups, written by the workgroup together
// usage: ubench7 total_MB think_ns "unit,unit,..." "order,order,..." [nw list] [nt list]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int NT>
__device__ __forceinline__ void st(v2d* p, v2d v) {
    if (NT == 0) *p = v;
    else __builtin_nontemporal_store(v, p);
}

__device__ __forceinline__ void think(unsigned long long ticks) {
    if (ticks == 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}

template <int NT>
__global__ __launch_bounds__(512) void store_units(double* out, long long nunits, int unit16, int order, int think_ticks) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long G = gridDim.x, w = blockIdx.x, gw = w * nw + wv, tw = G * nw;
    v2d val = {1.0 + lane, 2.0};
    v2d* base = reinterpret_cast<v2d*>(out);
    unsigned int rng = (unsigned int)gw * 2654435761u + 12345u;
    auto pause = [&]() {
        rng = rng * 1664525u + 1013904223u;
        // think_ticks * (0.75 .. 1.25), wave uniform
        const unsigned int r = __builtin_amdgcn_readfirstlane(rng >> 16) & 255u;
        think((unsigned long long)think_ticks * (192u + r / 2u) / 256u);
    };
    auto wave_unit = [&](long long u) {
        v2d* g = base + u * unit16;
        for (int i = lane; i < unit16; i += 64) st<NT>(g + i, val);
    };
    auto coop_unit = [&](long long u) {
        v2d* g = base + u * unit16;
        for (int i = wv * 64 + lane; i < unit16; i += nw * 64) st<NT>(g + i, val);
    };
    // start de-synchronised
    {
        rng = rng * 1664525u + 1013904223u;
        const unsigned int r = __builtin_amdgcn_readfirstlane(rng >> 16) & 255u;
        think((unsigned long long)think_ticks * r / 256u);
    }
    if (order == 0) {
        for (long long u = gw; u < nunits; u += tw) {
            wave_unit(u);
            pause();
        }
    } else if (order == 1) {
        const long long nruns = (nunits + nw - 1) / nw;
        for (long long r = w; r < nruns; r += G) {
            const long long u = r * nw + wv;
            if (u < nunits) wave_unit(u);
            pause();
        }
    } else if (order == 2) {
        const long long u0 = w * nunits / G, u1 = (w + 1) * nunits / G;
        for (long long u = u0 + wv; u < u1; u += nw) {
            wave_unit(u);
            pause();
        }
    } else if (order == 3) {
        const long long u0 = w * nunits / G, u1 = (w + 1) * nunits / G;
        for (long long u = u0; u < u1; ++u) {
            coop_unit(u);
            // a cooperative unit carries 1/NW of a wave's think time per unit
            if (((u - u0) % nw) == nw - 1) pause();
        }
    } else {
        long long k = 0;
        for (long long u = w; u < nunits; u += G, ++k) {
            coop_unit(u);
            if ((k % nw) == nw - 1) pause();
        }
    }
}

static std::vector<int> ints(const char* s) {
    std::vector<int> v;
    char* dup = strdup(s);
    for (char* tok = strtok(dup, ","); tok; tok = strtok(nullptr, ",")) v.push_back(atoi(tok));
    free(dup);
    return v;
}

int main(int argc, char** argv) {
    const long long total = (argc > 1 ? atoll(argv[1]) : 1472) * 1000000ll;
    const int think_ns = argc > 2 ? atoi(argv[2]) : 0;
    const std::vector<int> units = ints(argc > 3 ? argv[3] : "14720,32768");
    const std::vector<int> orders = ints(argc > 4 ? argv[4] : "0,1,2,3,4");
    const std::vector<int> nws = ints(argc > 5 ? argv[5] : "8");
    const std::vector<int> nts = ints(argc > 6 ? argv[6] : "0,1");
    hipEvent_t t0, t1;
    hipEventCreate(&t0);
    hipEventCreate(&t1);
    double* out;
    hipMalloc(&out, total + (1 << 20));
    // 1 MB-aligned base so that "aligned unit" means aligned in memory
    double* outa = reinterpret_cast<double*>((reinterpret_cast<unsigned long long>(out) + 0xfffffull) & ~0xfffffull);
    auto timeit = [&](auto fn) {
        for (int r = 0; r < 3; ++r) fn();
        hipDeviceSynchronize();
        hipEventRecord(t0);
        for (int r = 0; r < 20; ++r) fn();
        hipEventRecord(t1);
        hipEventSynchronize(t1);
        float ms;
        hipEventElapsedTime(&ms, t0, t1);
        return ms / 20 * 1e3;
    };
    double us = timeit([&] { hipMemsetAsync(outa, 0, (size_t)total, 0); });
    printf("memset %lld B: %7.1f us %6.0f GB/s   think %d ns\n", total, us, total / us / 1e3, think_ns);
    for (int unit : units)
        for (int order : orders)
            for (int nw : nws)
                for (int nt : nts) {
                    const long long nunits = total / unit;
                    const double bytes = (double)nunits * unit;
                    const int grid = 256 * (8 / nw > 0 ? 1 : 1);
                    // think time per unit scales with the unit size (same compute per output byte): think_ns is quoted per 14720 B
                    const int ticks = (int)((double)think_ns * unit / 14720.0 / 10.0);
                    auto fn = [&] {
                        if (nt == 0) store_units<0><<<grid, 64 * nw>>>(outa, nunits, unit / 16, order, ticks);
                        else store_units<1><<<grid, 64 * nw>>>(outa, nunits, unit / 16, order, ticks);
                    };
                    us = timeit(fn);
                    printf("unit %7d B order %d waves/WG %d nt %d : %7.1f us %6.0f GB/s\n", unit, order, nw, nt, us, bytes / us / 1e3);
                    fflush(stdout);
                }
    return 0;
}
