// Hardware probe (measurement tooling): lane layouts of the f64 MFMAs on gfx950.
// One-hot A and B operands; prints which (lane, reg) of D becomes non-zero.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void probe4x4(const double* a, const double* b, double* d) {
    double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, 0, 0, 0);
    d[threadIdx.x] = r;
}
__global__ void probe16(const double* a, const double* b, double* d) {
    v4d acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) d[threadIdx.x * 4 + j] = acc[j];
}
__global__ void time_mfma(double* out, int iters, int mode) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        } else if (mode == 1) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        } else if (mode == 2) {
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
        } else if (mode == 3) {
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
        } else {
            s0 = __builtin_fma(a, b, s0); s1 = __builtin_fma(a, b, s1);
            s2 = __builtin_fma(a, b, s2); s3 = __builtin_fma(a, b, s3);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc0[0] + acc1[1] + s0 + s1 + s2 + s3;
    if (threadIdx.x == 0) out[64 + mode] = (double)(t1 - t0) / iters;
}

int main() {
    double *da, *db, *dd;
    hipMalloc(&da, 64 * 8); hipMalloc(&db, 64 * 8); hipMalloc(&dd, 512 * 8);
    std::vector<double> a(64), b(64), d(256);
    printf("== v_mfma_f64_4x4x4_4b: A lane la, B lane lb -> D lane\n");
    std::vector<int> Aout(64 * 64, -1);
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            std::fill(a.begin(), a.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
            a[la] = 1.0; b[lb] = 1.0;
            hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
            probe4x4<<<1, 64>>>(da, db, dd);
            hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l) if (d[l] != 0.0) Aout[la * 64 + lb] = l;
        }
    for (int la = 0; la < 64; ++la) {
        printf("A%02d:", la);
        for (int lb = 0; lb < 64; ++lb) if (Aout[la * 64 + lb] >= 0) printf(" B%02d->D%02d", lb, Aout[la * 64 + lb]);
        printf("\n");
    }
    printf("== v_mfma_f64_16x16x4: A lane la, B lane lb -> D (lane,reg)\n");
    for (int la : {0, 1, 15, 16, 17, 33, 63})
        for (int lb : {0, 1, 15, 16, 17, 33, 63}) {
            std::fill(a.begin(), a.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
            a[la] = 1.0; b[lb] = 1.0;
            hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
            probe16<<<1, 64>>>(da, db, dd);
            hipMemcpy(d.data(), dd, 2048, hipMemcpyDeviceToHost);
            for (int l = 0; l < 256; ++l) if (d[l] != 0.0) printf("A%02d B%02d -> lane %d reg %d\n", la, lb, l / 4, l % 4);
        }
    for (int mode = 0; mode < 5; ++mode) {
        time_mfma<<<1, 64>>>(dd, 4096, mode);
        hipDeviceSynchronize();
        hipMemcpy(d.data(), dd, 70 * 8, hipMemcpyDeviceToHost);
        const char* nm[] = {"2x mfma16x16x4 (indep)", "1x mfma16x16x4 (dep chain)", "4x mfma4x4x4 (indep)",
                            "1x mfma4x4x4 (dep chain)", "4x v_fma_f64 (indep)"};
        printf("ticks/iter %-28s %.2f\n", nm[mode], d[64 + mode]);
    }
    return 0;
}
