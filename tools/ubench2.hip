// Microbenchmarks, part 2 (measurement tooling): do fp64 MFMA and fp64 VALU share an
// execution pipe on gfx950?  What do 16-byte LDS stores cost?
// One 512-thread workgroup per CU: waves 0-3 and 4-7 land pairwise on the four SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// ROLE_A for waves 0-3, ROLE_B for waves 4-7.  roles: 0 idle, 1 fp64 VALU, 2 MFMA 16x16x4, 3 ds_write_b64,
// 4 ds_write_b128 (23 lanes), 5 ds_write_b128 (64 lanes), 6 ds_write_b64 (23 lanes), 7 f32 VALU
template <int ROLE> __device__ __forceinline__ double work(int iters, double* lds) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double x0 = a, x1 = b, x2 = a + b, x3 = a - b, x4 = 1.5, x5 = 2.5, x6 = 3.5, x7 = 4.5;
    float f0 = (float)a, f1 = (float)b, f2 = 1.f, f3 = 2.f, f4 = 3.f, f5 = 4.f, f6 = 5.f, f7 = 6.f, fa = (float)a, fb = (float)b;
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double* my = lds + threadIdx.x * 2;
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < iters; ++i) {
        if constexpr (ROLE == 1) {
            x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
            x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
        } else if constexpr (ROLE == 2) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        } else if constexpr (ROLE == 3) {
            my[0] = x0; my[1024] = x1; my[2048] = x2; my[3072] = x3; my[4096] = x0; my[5120] = x1; my[6144] = x2; my[7168] = x3;
            asm volatile("" ::: "memory");
        } else if constexpr (ROLE == 4) {
            if (lane < 23) {
                v2d* m2 = reinterpret_cast<v2d*>(my);
                m2[0] = v2d{x0, x1}; m2[512] = v2d{x2, x3}; m2[1024] = v2d{x0, x1}; m2[1536] = v2d{x2, x3};
                m2[2048] = v2d{x0, x1}; m2[2560] = v2d{x2, x3}; m2[3072] = v2d{x0, x1}; m2[3584] = v2d{x2, x3};
            }
            asm volatile("" ::: "memory");
        } else if constexpr (ROLE == 5) {
            v2d* m2 = reinterpret_cast<v2d*>(my);
            m2[0] = v2d{x0, x1}; m2[512] = v2d{x2, x3}; m2[1024] = v2d{x0, x1}; m2[1536] = v2d{x2, x3};
            m2[2048] = v2d{x0, x1}; m2[2560] = v2d{x2, x3}; m2[3072] = v2d{x0, x1}; m2[3584] = v2d{x2, x3};
            asm volatile("" ::: "memory");
        } else if constexpr (ROLE == 6) {
            if (lane < 23) { my[0] = x0; my[1024] = x1; my[2048] = x2; my[3072] = x3; my[4096] = x0; my[5120] = x1; my[6144] = x2; my[7168] = x3; }
            asm volatile("" ::: "memory");
        } else if constexpr (ROLE == 7) {
            f0 = __builtin_fmaf(f0, fa, fb); f1 = __builtin_fmaf(f1, fa, fb); f2 = __builtin_fmaf(f2, fa, fb); f3 = __builtin_fmaf(f3, fa, fb);
            f4 = __builtin_fmaf(f4, fa, fb); f5 = __builtin_fmaf(f5, fa, fb); f6 = __builtin_fmaf(f6, fa, fb); f7 = __builtin_fmaf(f7, fa, fb);
        }
    }
    return x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + acc0[0] + acc1[1] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int RA, int RB> __global__ __launch_bounds__(512) void bench(double* out, int ia, int ib, long long* clk) {
    extern __shared__ double lds[];
    const int wave = threadIdx.x >> 6;
    long long t0 = __builtin_readcyclecounter();
    double r;
    if (wave < 4) r = work<RA>(ia, lds); else r = work<RB>(ib, lds);
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) clk[wave] = t1 - t0;
}

template <int RA, int RB> void run(const char* name, int ia, int ib) {
    double* out; long long* clk;
    const int grid = 256;
    hipMalloc(&out, (size_t)grid * 512 * 8);
    hipMalloc(&clk, 8 * 8);
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    const size_t lds = 8192 * 2 * 8;
    bench<RA, RB><<<grid, 512, lds>>>(out, 10, 10, clk);
    hipDeviceSynchronize();
    hipEventRecord(t0);
    bench<RA, RB><<<grid, 512, lds>>>(out, ia, ib, clk);
    hipEventRecord(t1);
    hipEventSynchronize(t1);
    float ms; hipEventElapsedTime(&ms, t0, t1);
    long long h[8];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    printf("%-52s %8.3f ms   wave0 %9lld clk  wave4 %9lld clk (s_memtime ticks)\n", name, ms, h[0], h[4]);
    hipFree(out); hipFree(clk);
}

int main() {
    const int NV = 20000, NM = 10000;   // NV x 8 DFMA  ~  NM x 2 MFMA(16x16x4)  in pipe cycles (8*4 vs 2*64 -> x4): tune below
    run<1, 0>("VALU f64 x4 waves, other 4 idle", NV, 0);
    run<1, 1>("VALU f64 x8 waves", NV, NV);
    run<2, 0>("MFMA f64 16x16x4 x4 waves, other 4 idle", NM, 0);
    run<2, 2>("MFMA f64 x8 waves", NM, NM);
    run<1, 2>("VALU f64 (w0-3) + MFMA f64 (w4-7)", NV, NM);
    run<7, 0>("VALU f32 x4 waves, other idle", NV, 0);
    run<7, 2>("VALU f32 (w0-3) + MFMA f64 (w4-7)", NV, NM);
    run<7, 1>("VALU f32 (w0-3) + VALU f64 (w4-7)", NV, NV);
    run<3, 0>("ds_write_b64 64 lanes x4 waves", NV, 0);
    run<6, 0>("ds_write_b64 23 lanes x4 waves", NV, 0);
    run<5, 0>("ds_write_b128 64 lanes x4 waves", NV, 0);
    run<4, 0>("ds_write_b128 23 lanes x4 waves", NV, 0);
    run<3, 3>("ds_write_b64 64 lanes x8 waves", NV, NV);
    run<4, 4>("ds_write_b128 23 lanes x8 waves", NV, NV);
    run<1, 3>("VALU f64 (w0-3) + ds_write_b64 (w4-7)", NV, NV);
    run<2, 3>("MFMA f64 (w0-3) + ds_write_b64 (w4-7)", NM, NV);
    return 0;
}
