// Throughput microbenchmarks (measurement tooling): fp64 VALU / MFMA / LDS-store rates on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE> __global__ __launch_bounds__(256) void bench(double* out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double x0 = a, x1 = b, x2 = a + b, x3 = a - b, x4 = 1.5, x5 = 2.5, x6 = 3.5, x7 = 4.5;
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    extern __shared__ double lds[];
    double* my = lds + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {  // 8 independent DFMA
            x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
            x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
        } else if constexpr (MODE == 1) {  // 1 dependent DFMA chain
            x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b);
            x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b);
        } else if constexpr (MODE == 2) {  // 8 independent DMUL
            x0 *= a; x1 *= a; x2 *= a; x3 *= a; x4 *= a; x5 *= a; x6 *= a; x7 *= a;
        } else if constexpr (MODE == 3) {  // 2 independent MFMA 16x16x4 f64 (x4 per iter)
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        } else if constexpr (MODE == 4) {  // 4 independent MFMA 4x4x4 (x2)
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0); s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0); s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0); s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0); s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
        } else if constexpr (MODE == 5) {  // 8 ds_write_b64 full wave
            my[0] = x0; my[256] = x1; my[512] = x2; my[768] = x3; my[1024] = x0; my[1280] = x1; my[1536] = x2; my[1792] = x3;
            asm volatile("" ::: "memory");
        } else if constexpr (MODE == 6) {  // 8 ds_write_b64 with 23 of 64 lanes active
            if ((threadIdx.x & 63) < 23) { my[0] = x0; my[256] = x1; my[512] = x2; my[768] = x3; my[1024] = x0; my[1280] = x1; my[1536] = x2; my[1792] = x3; }
            asm volatile("" ::: "memory");
        } else if constexpr (MODE == 7) {  // 8 ds_read_b64
            x0 += my[0]; x1 += my[256]; x2 += my[512]; x3 += my[768]; x4 += my[1024]; x5 += my[1280]; x6 += my[1536]; x7 += my[1792];
            asm volatile("" ::: "memory");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + acc0[0] + acc1[1] + s0 + s1 + s2 + s3;
}

template <int MODE> void run(const char* name, double ops_per_iter_per_wave, const char* unit, int wg_per_cu) {
    double* out;
    const int ncu = 256, iters = 20000;
    int grid = ncu * wg_per_cu;
    hipMalloc(&out, (size_t)grid * 256 * 8);
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    bench<MODE><<<grid, 256, 2048 * 8>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(t0);
    bench<MODE><<<grid, 256, 2048 * 8>>>(out, iters);
    hipEventRecord(t1);
    hipEventSynchronize(t1);
    float ms; hipEventElapsedTime(&ms, t0, t1);
    double waves = (double)grid * 4;
    double total = ops_per_iter_per_wave * iters * waves;
    double per_simd_cycle = total / (ms * 1e-3 * 2.4e9 * 1024);
    printf("%-44s wg/cu=%d  %8.3f ms  %10.3f G%s/s   %.3f %s per SIMD-cycle(@2.4GHz)  -> %.2f cycles per wave-instr\n", name, wg_per_cu, ms,
           total / (ms * 1e-3) / 1e9, unit, per_simd_cycle, unit, 1.0 / per_simd_cycle);
    hipFree(out);
}

int main() {
    for (int w : {1, 2}) {
        run<0>("8x indep v_fma_f64 (wave-instr)", 8, "inst", w);
        run<1>("8x dependent v_fma_f64", 8, "inst", w);
        run<2>("8x indep v_mul_f64", 8, "inst", w);
        run<3>("8x mfma_f64_16x16x4 (2 chains)", 8, "inst", w);
        run<4>("8x mfma_f64_4x4x4_4b (4 chains)", 8, "inst", w);
        run<5>("8x ds_write_b64 (64 lanes)", 8, "inst", w);
        run<6>("8x ds_write_b64 (23 lanes)", 8, "inst", w);
        run<7>("8x ds_read_b64", 8, "inst", w);
    }
    return 0;
}
