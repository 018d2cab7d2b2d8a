// C ABI of libfiat_amd (include/fiat_amd.h): contexts, element plans, launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <string>
#include <numeric>
#include <vector>

#include "../../include/fiat_amd.h"
#include "aux_kernels.hpp"
#include "plan.hpp"
#include "simplex_kernel.hpp"
#include "simplex_fixed.hpp"
#include "simplex_stream.hpp"
#include "simplex_pair.hpp"
#include "coop_kernel.hpp"
#include "shared_points.hpp"
#include "quadrature.hpp"
#include "simplex_small.hpp"
#include "macro_small.hpp"
#include "table_kernels.hpp"
#include "simplex_stacked.hpp"
#include "jacobi_kernel.hpp"
#include "tensor_small.hpp"
#include "prism_small.hpp"
#include "wg_launch.hpp"

namespace {

// Measurement scaffolding (ablation bits, verbose launch reports, occupancy caps) is compiled into the A/B build only
// (make ab-lib: -DFX_AB); the product library never reads the environment on a launch path.  Kernel-selection
// switches that tests and tools need are an explicit per-context policy (fx_ctx_set_policy).
#ifdef FX_AB
inline const char* ab_env(const char* name) { return getenv(name); }
#else
inline const char* ab_env(const char*) { return nullptr; }
#endif

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (void)hipGetLastError(); /* reported here: not again by the next launch check */ \
            return fail(FX_EHIP, "%s: %s", #expr, hipGetErrorString(e_));                  \
        }                                                                                  \
    } while (0)

const double UFC[3][12] = {
    {0.0, 1.0},
    {0.0, 0.0, 1.0, 0.0, 0.0, 1.0},
    {0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0},
};

// host version of the device cell_map (same closed form)
bool invert_small(int sd, const double* A, double* inv) {
    if (sd == 1) {
        if (A[0] == 0.0) return false;
        inv[0] = 1.0 / A[0];
        return true;
    }
    if (sd == 2) {
        const double det = A[0] * A[3] - A[1] * A[2];
        if (det == 0.0) return false;
        inv[0] = A[3] / det;
        inv[1] = -A[1] / det;
        inv[2] = -A[2] / det;
        inv[3] = A[0] / det;
        return true;
    }
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    if (det == 0.0) return false;
    inv[0] = c00 / det;
    inv[1] = (A[2] * A[7] - A[1] * A[8]) / det;
    inv[2] = (A[1] * A[5] - A[2] * A[4]) / det;
    inv[3] = c01 / det;
    inv[4] = (A[0] * A[8] - A[2] * A[6]) / det;
    inv[5] = (A[2] * A[3] - A[0] * A[5]) / det;
    inv[6] = c02 / det;
    inv[7] = (A[1] * A[6] - A[0] * A[7]) / det;
    inv[8] = (A[0] * A[4] - A[1] * A[3]) / det;
    return true;
}

bool host_cell_map(int sd, const double* v, double* A, double* b) {
    if (sd == 1) {
        double den = v[1] - v[0];
        if (den == 0.0) return false;
        A[0] = 2.0 / den;
        b[0] = -1.0 - A[0] * v[0];
        return true;
    }
    if (sd == 2) {
        double e00 = v[2] - v[0], e10 = v[3] - v[1], e01 = v[4] - v[0], e11 = v[5] - v[1];
        double det = e00 * e11 - e01 * e10;
        if (det == 0.0) return false;
        double inv = 2.0 / det;
        A[0] = e11 * inv;
        A[1] = -e01 * inv;
        A[2] = -e10 * inv;
        A[3] = e00 * inv;
        for (int i = 0; i < 2; ++i) b[i] = -1.0 - (A[i * 2] * v[0] + A[i * 2 + 1] * v[1]);
        return true;
    }
    double e[3][3];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) e[r][c] = v[3 * (c + 1) + r] - v[r];
    double c00 = e[1][1] * e[2][2] - e[1][2] * e[2][1];
    double c01 = e[1][2] * e[2][0] - e[1][0] * e[2][2];
    double c02 = e[1][0] * e[2][1] - e[1][1] * e[2][0];
    double det = e[0][0] * c00 + e[0][1] * c01 + e[0][2] * c02;
    if (det == 0.0) return false;
    double inv = 2.0 / det;
    A[0] = c00 * inv;
    A[1] = (e[0][2] * e[2][1] - e[0][1] * e[2][2]) * inv;
    A[2] = (e[0][1] * e[1][2] - e[0][2] * e[1][1]) * inv;
    A[3] = c01 * inv;
    A[4] = (e[0][0] * e[2][2] - e[0][2] * e[2][0]) * inv;
    A[5] = (e[0][2] * e[1][0] - e[0][0] * e[1][2]) * inv;
    A[6] = c02 * inv;
    A[7] = (e[0][1] * e[2][0] - e[0][0] * e[2][1]) * inv;
    A[8] = (e[0][0] * e[1][1] - e[0][1] * e[1][0]) * inv;
    for (int i = 0; i < 3; ++i) b[i] = -1.0 - (A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2]);
    return true;
}

double default_simplex_volume(int sd) {
    // (-1,1)^sd simplex: 2^sd / sd!
    double v = 1.0;
    for (int i = 1; i <= sd; ++i) v *= 2.0 / i;
    return v;
}

}  // namespace

struct fx_ctx {
    int device = 0;
    int num_cu = 0;
    int lds_per_cu = 0;
    std::string name;
    double* d_trash = nullptr;  // 64 KB scratch (ablation builds: wave lifetimes, FX_DBG & 512)
    unsigned long long* d_queue = nullptr;  // chunk counters of the dynamically scheduled kernels (work_queue.hpp)
    unsigned int launch_seq = 0;
    unsigned policy = 0;      // FX_POLICY_* bits (fx_ctx_set_policy)
};
constexpr int FX_QUEUE_SLOTS = 64;  // counters handed to consecutive launches round-robin (128 B apart)

#include "comm.hpp"

constexpr int FX_MAX_ORDER = 8;  // highest derivative order served (orders > 2 through differentiation matrices)

struct fx_element {
    fx_ctx* ctx = nullptr;
    int sd = 0, n = 0, variant = 0, nexp = 0, ndof = 0, vdim = 1;
    double scale = 0.0;
    double A0[9] = {0}, b0[3] = {0};
    fx::Program prog;
    std::vector<double> T;  // C0 transform (bubble) or empty
    int KS = 0, MT = 0;
    fxk::Step* d_steps = nullptr;
    double* d_afrag = nullptr;
    double* d_cmat = nullptr;          // plain coefficient matrix [rows][nexp] (C0 transform folded in)
    double* d_afrag_split = nullptr;  // layout of the shape-specialised kernels
    double* d_afrag_stream = nullptr; // same, K in production order (K-streamed kernel)
    // cooperative (large-shape) plan
    int coop_KS = 0, coop_emax = 0;
    int* d_coop_eint = nullptr;
    double* d_coop_edbl = nullptr;
    int* d_coop_kstart = nullptr;
    double* d_afrag_coop = nullptr;
    // stacked-matrix kernel (simplex_stacked.hpp): effective coefficients on the host, A fragments of
    // [C; C D^alpha ...] per derivative order (built at first use, ensure_stacked)
    std::vector<double> hC;
    double* d_astack[3] = {nullptr, nullptr, nullptr};
    double* d_astack_dm[3] = {nullptr, nullptr, nullptr};   // orders 1 and 2, dof-major tiles (the tables of 16 dofs one after the other, each padded to a tile): MIXT instances
    double* d_astack_dmp[3] = {nullptr, nullptr, nullptr};  // orders 0-2, vector-valued elements: dof-major tiles with the components of a dof in one MFMA lane (PIO instances)
    int stack_state[3] = {0, 0, 0};  // 0 not built, 1 built, -1 failed
    bool raw_expansion = false;      // internal helper element (identity coefficients): never takes the stacked path
    // derivative orders 3..FX_MAX_ORDER (ensure_high_order): an internal element whose rows are the stacked matrix
    // [C D^alpha], |alpha| <= order, tabulated at order 0
    fx_element* high[FX_MAX_ORDER + 1] = {nullptr};
    int high_state[FX_MAX_ORDER + 1] = {0};
};

struct fx_line_element {
    fx_ctx* ctx = nullptr;
    int nn = 0;
    double* d_buf = nullptr;  // nodes | wts | dmat
    std::vector<double> nodes, wts, dmat;
};

extern "C" {

const char* fx_last_error(void) { return g_err.c_str(); }
int fx_abi_version(void) { return 2; }  // 2: fx_allgather_tables takes recv_count, fx_tables_squared_norm

int fx_num_tables(int sd, int order) {
    if (sd < 1 || sd > 3 || order < 0) return fail(FX_EINVAL, "fx_num_tables: bad sd/order");
    return fx::binom(sd + order, sd);
}

int fx_ctx_create(int device_id, fx_ctx** out) {
    if (!out) return fail(FX_EINVAL, "fx_ctx_create: null output");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(FX_EHIP, "fx_ctx_create: no HIP device available (%s); fiat_amd has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= count) return fail(FX_EINVAL, "fx_ctx_create: device %d out of range", device_id);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    fx_ctx* c = new fx_ctx;
    c->device = device_id;
    c->num_cu = prop.multiProcessorCount;
    c->lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (c->lds_per_cu <= 0) c->lds_per_cu = 160 * 1024;
    c->name = prop.gcnArchName;
#if defined(FX_DBG) && (FX_DBG & 512)
    const size_t trash_bytes = 256 * 1024;  // wave timelines of the ablation build
#else
    const size_t trash_bytes = 64 * 1024;
#endif
    if (hipMalloc(&c->d_trash, trash_bytes) != hipSuccess || hipMalloc(&c->d_queue, FX_QUEUE_SLOTS * 128) != hipSuccess ||
        hipMemset(c->d_queue, 0, FX_QUEUE_SLOTS * 128) != hipSuccess) {
        if (c->d_trash) (void)hipFree(c->d_trash);
        delete c;
        return fail(FX_ENOMEM, "fx_ctx_create: out of device memory");
    }
    *out = c;
    return FX_OK;
}

int fx_ctx_destroy(fx_ctx* ctx) {
    if (ctx && ctx->d_trash) (void)hipFree(ctx->d_trash);
    if (ctx && ctx->d_queue) (void)hipFree(ctx->d_queue);
    delete ctx;
    return FX_OK;
}

int fx_ctx_info(fx_ctx* ctx, int* num_cu, int* lds, char* name, int name_len) {
    if (!ctx) return fail(FX_EINVAL, "fx_ctx_info: null context");
    if (num_cu) *num_cu = ctx->num_cu;
    if (lds) *lds = ctx->lds_per_cu;
    if (name && name_len > 0) {
        strncpy(name, ctx->name.c_str(), name_len - 1);
        name[name_len - 1] = 0;
    }
    return FX_OK;
}

int fx_ctx_set_policy(fx_ctx* ctx, unsigned flags) {
    if (!ctx) return fail(FX_EINVAL, "fx_ctx_set_policy: null context");
    if (flags & ~(unsigned)FX_POLICY_ALL) return fail(FX_EINVAL, "fx_ctx_set_policy: unknown policy bits 0x%x", flags & ~(unsigned)FX_POLICY_ALL);
    if ((flags & FX_POLICY_KERNEL_IMAGE) && (flags & FX_POLICY_KERNEL_STREAM))
        return fail(FX_EINVAL, "fx_ctx_set_policy: KERNEL_IMAGE and KERNEL_STREAM exclude each other");
    ctx->policy = flags;
    return FX_OK;
}

int fx_ctx_get_policy(const fx_ctx* ctx, unsigned* flags) {
    if (!ctx || !flags) return fail(FX_EINVAL, "fx_ctx_get_policy: null argument");
    *flags = ctx->policy;
    return FX_OK;
}

// ---------------------------------------------------------------------------------
// plan introspection (host only; used by the CPU test-suite to check the folded
// recurrence tables against the oracle without a GPU)
int fx_plan_steps(int sd, int n, int variant, double scale, int cap, int* nsteps, double* phi0,
                  int* ints /* [cap][4] */, double* coefs /* [cap][3] */) {
    if (sd < 1 || sd > 3 || n < 0 || variant < 0 || variant > 2) return fail(FX_EINVAL, "fx_plan_steps: bad arguments");
    if (variant == FX_VARIANT_BUBBLE && n < 1) return fail(FX_EINVAL, "bubble variant needs degree >= 1");
    if (scale <= 0.0) scale = std::sqrt(1.0 / default_simplex_volume(sd));
    fx::Program P = fx::build_program(sd, n, variant, scale);
    if (nsteps) *nsteps = (int)P.steps.size();
    if (phi0) *phi0 = P.phi0;
    int m = std::min<int>(cap, (int)P.steps.size());
    for (int i = 0; i < m; ++i) {
        if (ints) {
            ints[4 * i + 0] = P.steps[i].dst;
            ints[4 * i + 1] = P.steps[i].cur;
            ints[4 * i + 2] = P.steps[i].prv;
            ints[4 * i + 3] = P.steps[i].codim;
        }
        if (coefs) {
            coefs[3 * i + 0] = P.steps[i].A;
            coefs[3 * i + 1] = P.steps[i].B;
            coefs[3 * i + 2] = P.steps[i].C;
        }
    }
    return FX_OK;
}

int fx_plan_coop(int sd, int n, int variant, double scale, int cap, int* KS, int* nentries /* [4] */,
                 int* ints /* [4][cap][5] */, int* kstart /* [4][cap_k] */, int cap_k, int* kperm /* [4*cap_k] */) {
    if (sd < 1 || sd > 3 || n < 0 || variant < 0 || variant > 2) return fail(FX_EINVAL, "fx_plan_coop: bad arguments");
    if (variant == FX_VARIANT_BUBBLE && n < 1) return fail(FX_EINVAL, "bubble variant needs degree >= 1");
    if (scale <= 0.0) scale = std::sqrt(1.0 / default_simplex_volume(sd));
    fx::Program P = fx::build_program(sd, n, variant, scale);
    fx::CoopPlan C = fx::build_coop_plan(P);
    if (KS) *KS = C.KS;
    for (int w = 0; w < 4; ++w) {
        if (nentries) nentries[w] = (int)C.entries[w].size();
        const int m = std::min<int>(cap, (int)C.entries[w].size());
        for (int i = 0; i < m && ints; ++i) {
            const fx::CoopEntry& e = C.entries[w][i];
            int* q = ints + ((size_t)w * cap + i) * 5;
            q[0] = e.level;
            q[1] = e.seed;
            q[2] = e.publish;
            q[3] = e.member;
            q[4] = 0;
        }
        for (int j = 0; j <= C.KS && j < cap_k && kstart; ++j) kstart[(size_t)w * cap_k + j] = C.kstart[w][j];
    }
    for (int k = 0; k < 4 * C.KS && k < 4 * cap_k && kperm; ++k) kperm[k] = C.kperm[k];
    return FX_OK;
}

int fx_plan_c0_transform(int sd, int n, double* T /* [nexp][nexp] */) {
    if (sd < 1 || sd > 3 || n < 1 || !T) return fail(FX_EINVAL, "fx_plan_c0_transform: bad arguments");
    std::vector<double> M = fx::c0_transform(sd, n);
    memcpy(T, M.data(), M.size() * sizeof(double));
    return FX_OK;
}

// ---------------------------------------------------------------------------------
static int upload_coeffs(fx_element* e, int ndof, int vdim, const double* coeffs) {
    const int rows = ndof * vdim, nexp = e->nexp;
    std::vector<double> C((size_t)rows * nexp, 0.0);
    if (coeffs) {
        C.assign(coeffs, coeffs + (size_t)rows * nexp);
    } else {
        if (rows != nexp) return fail(FX_EINVAL, "identity coefficients need ndof*vdim == nexp");
        for (int i = 0; i < nexp; ++i) C[(size_t)i * nexp + i] = 1.0;
    }
    if (!e->T.empty()) {  // tables = coeffs . C0(phi) = (coeffs . T) . phi
        std::vector<double> CT((size_t)rows * nexp, 0.0);
        for (int i = 0; i < rows; ++i)
            for (int j = 0; j < nexp; ++j) {
                double cij = C[(size_t)i * nexp + j];
                if (cij == 0.0) continue;
                const double* trow = &e->T[(size_t)j * nexp];
                for (int k = 0; k < nexp; ++k) CT[(size_t)i * nexp + k] += cij * trow[k];
            }
        C.swap(CT);
    }
    e->hC = C;
    for (int o = 0; o < 3; ++o) {
        if (e->d_astack[o]) (void)hipFree(e->d_astack[o]);
        e->d_astack[o] = nullptr;
        e->stack_state[o] = 0;
    }
    for (int o = 0; o < 3; ++o) {
        if (e->d_astack_dm[o]) (void)hipFree(e->d_astack_dm[o]);
        e->d_astack_dm[o] = nullptr;
        if (e->d_astack_dmp[o]) (void)hipFree(e->d_astack_dmp[o]);
        e->d_astack_dmp[o] = nullptr;
    }
    for (int o = 0; o <= FX_MAX_ORDER; ++o) {
        if (e->high[o]) fx_element_destroy(e->high[o]);
        e->high[o] = nullptr;
        e->high_state[o] = 0;
    }
    if (e->d_cmat) {
        HIP_TRY(hipFree(e->d_cmat));
        e->d_cmat = nullptr;
    }
    HIP_TRY(hipMalloc(&e->d_cmat, C.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(e->d_cmat, C.data(), C.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> F = fx::pack_a_fragments(C, rows, nexp);
    if (e->d_afrag) {
        HIP_TRY(hipFree(e->d_afrag));
        e->d_afrag = nullptr;
    }
    HIP_TRY(hipMalloc(&e->d_afrag, F.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(e->d_afrag, F.data(), F.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> F2 = fx::pack_a_fragments_split(C, rows, nexp);
    if (e->d_afrag_split) {
        HIP_TRY(hipFree(e->d_afrag_split));
        e->d_afrag_split = nullptr;
    }
    HIP_TRY(hipMalloc(&e->d_afrag_split, F2.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(e->d_afrag_split, F2.data(), F2.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<int> kperm(nexp, 0);
    for (size_t i = 0; i < e->prog.steps.size() && (int)i + 1 < nexp; ++i) kperm[i + 1] = e->prog.steps[i].dst;
    std::vector<double> F3 = fx::pack_a_fragments_split(C, rows, nexp, &kperm);
    if (e->d_afrag_stream) {
        HIP_TRY(hipFree(e->d_afrag_stream));
        e->d_afrag_stream = nullptr;
    }
    HIP_TRY(hipMalloc(&e->d_afrag_stream, F3.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(e->d_afrag_stream, F3.data(), F3.size() * sizeof(double), hipMemcpyHostToDevice));
    if (e->n >= 1) {
        // cooperative plan: entry tables of the four producers + A fragments in slot order
        fx::CoopPlan cp = fx::build_coop_plan(e->prog);
        int emax = 1;
        for (int w = 0; w < 4; ++w) emax = std::max<int>(emax, (int)cp.entries[w].size());
        std::vector<int> eint((size_t)4 * emax * 4, 0);
        std::vector<double> edbl((size_t)4 * emax * 16, 0.0);
        std::vector<int> kst((size_t)4 * (cp.KS + 1), 0);
        const int sd = e->sd;
        for (int w = 0; w < 4; ++w) {
            for (size_t i = 0; i < cp.entries[w].size(); ++i) {
                const fx::CoopEntry& ce = cp.entries[w][i];
                int* ip = &eint[((size_t)w * emax + i) * 4];
                ip[0] = ce.level;
                ip[1] = ce.seed;
                ip[2] = ce.publish;
                ip[3] = ce.member;
                double* dp = &edbl[((size_t)w * emax + i) * 16];
                dp[0] = ce.A;
                dp[1] = ce.B;
                dp[2] = ce.C;
                if (ce.level >= 0) {  // uniform-cell factor derivatives, as for the stream kernel
                    double dfa[3] = {0, 0, 0}, dfb[3] = {0, 0, 0};
                    for (int d = 0; d < sd; ++d) {
                        double dx = e->A0[ce.level * sd + d];
                        double dy = ce.level + 1 < sd ? e->A0[(ce.level + 1) * sd + d] : 0.0;
                        double dz = ce.level + 2 < sd ? e->A0[(ce.level + 2) * sd + d] : 0.0;
                        dfb[d] = 0.5 * (dy + dz);
                        dfa[d] = dx + dfb[d];
                    }
                    double* u = dp + 3;
                    for (int d = 0; d < sd; ++d) {
                        u[d] = ce.A * dfa[d] - ce.B * dfb[d];
                        u[3 + d] = -2.0 * ce.C * dfb[d];
                    }
                    int h = 0;
                    for (int d1 = 0; d1 < sd; ++d1)
                        for (int d2 = d1; d2 < sd; ++d2) u[6 + h++] = -2.0 * ce.C * dfb[d1] * dfb[d2];
                }
            }
            for (int j = 0; j <= cp.KS; ++j) kst[(size_t)w * (cp.KS + 1) + j] = cp.kstart[w][j];
        }
        std::vector<double> F4 = fx::pack_a_fragments_split(C, rows, nexp, &cp.kperm);
        auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
            if (*dst) {
                (void)hipFree(*dst);
                *dst = nullptr;
            }
            hipError_t he = hipMalloc(dst, bytes);
            if (he == hipSuccess) he = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
            return he;
        };
        HIP_TRY(upload((void**)&e->d_coop_eint, eint.data(), eint.size() * sizeof(int)));
        HIP_TRY(upload((void**)&e->d_coop_edbl, edbl.data(), edbl.size() * sizeof(double)));
        HIP_TRY(upload((void**)&e->d_coop_kstart, kst.data(), kst.size() * sizeof(int)));
        HIP_TRY(upload((void**)&e->d_afrag_coop, F4.data(), F4.size() * sizeof(double)));
        e->coop_KS = cp.KS;
        e->coop_emax = emax;
    }
    e->ndof = ndof;
    e->vdim = vdim;
    e->MT = (rows + 15) / 16;
    e->KS = (nexp + 3) / 4;
    return FX_OK;
}

int fx_element_create(fx_ctx* ctx, int sd, int n, int variant, double scale, const double* verts,
                      int ndof, int vdim, const double* coeffs, fx_element** out) {
    if (!ctx || !out) return fail(FX_EINVAL, "fx_element_create: null argument");
    if (sd < 1 || sd > 3) return fail(FX_EINVAL, "Invalid number of spatial dimensions");
    if (n < 0) return fail(FX_EINVAL, "negative degree");
    if (variant < 0 || variant > 2) return fail(FX_EINVAL, "Invalid variant %d", variant);
    if (variant == FX_VARIANT_BUBBLE && n < 1) return fail(FX_EINVAL, "bubble variant needs degree >= 1");
    if (ndof < 1 || vdim < 1) return fail(FX_EINVAL, "bad ndof/vdim");
    HIP_TRY(hipSetDevice(ctx->device));
    fx_element* e = new fx_element;
    e->ctx = ctx;
    e->sd = sd;
    e->n = n;
    e->variant = variant;
    e->nexp = fx::binom(n + sd, sd);
    if (scale <= 0.0) {
        scale = std::sqrt(1.0 / default_simplex_volume(sd));
        if (n == 0 && sd > 1) scale = 1.0;  // expansions.py:396-398
    }
    e->scale = scale;
    if (!host_cell_map(sd, verts ? verts : UFC[sd - 1], e->A0, e->b0)) {
        delete e;
        return fail(FX_EINVAL, "degenerate cell");
    }
    e->prog = fx::build_program(sd, n, variant, scale);
    if (variant == FX_VARIANT_BUBBLE) e->T = fx::c0_transform(sd, n);
    static_assert(sizeof(fx::Step) == sizeof(fxk::Step), "step layout");
    size_t sbytes = std::max<size_t>(1, e->prog.steps.size()) * sizeof(fxk::Step);
    hipError_t he = hipMalloc(&e->d_steps, sbytes);
    if (he == hipSuccess && !e->prog.steps.empty())
        he = hipMemcpy(e->d_steps, e->prog.steps.data(), e->prog.steps.size() * sizeof(fxk::Step), hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        fx_element_destroy(e);
        return fail(FX_EHIP, "fx_element_create: %s", hipGetErrorString(he));
    }
    int rc = upload_coeffs(e, ndof, vdim, coeffs);
    if (rc != FX_OK) {
        fx_element_destroy(e);
        return rc;
    }
    *out = e;
    return FX_OK;
}

int fx_element_destroy(fx_element* e) {
    if (!e) return FX_OK;
    if (e->d_steps) (void)hipFree(e->d_steps);
    if (e->d_cmat) (void)hipFree(e->d_cmat);
    if (e->d_afrag) (void)hipFree(e->d_afrag);
    if (e->d_afrag_split) (void)hipFree(e->d_afrag_split);
    if (e->d_afrag_stream) (void)hipFree(e->d_afrag_stream);
    if (e->d_coop_eint) (void)hipFree(e->d_coop_eint);
    if (e->d_coop_edbl) (void)hipFree(e->d_coop_edbl);
    if (e->d_coop_kstart) (void)hipFree(e->d_coop_kstart);
    if (e->d_afrag_coop) (void)hipFree(e->d_afrag_coop);
    for (int o = 0; o < 3; ++o)
        if (e->d_astack[o]) (void)hipFree(e->d_astack[o]);
    for (int o = 0; o < 3; ++o) {
        if (e->d_astack_dm[o]) (void)hipFree(e->d_astack_dm[o]);
        if (e->d_astack_dmp[o]) (void)hipFree(e->d_astack_dmp[o]);
    }
    for (int o = 0; o <= FX_MAX_ORDER; ++o)
        if (e->high[o]) fx_element_destroy(e->high[o]);
    delete e;
    return FX_OK;
}

int fx_element_set_coeffs(fx_element* e, int ndof, int vdim, const double* coeffs) {
    if (!e || ndof < 1 || vdim < 1) return fail(FX_EINVAL, "fx_element_set_coeffs: bad argument");
    HIP_TRY(hipSetDevice(e->ctx->device));
    return upload_coeffs(e, ndof, vdim, coeffs);
}

int fx_element_dims(const fx_element* e, int* sd, int* n, int* nexp, int* ndof, int* vdim) {
    if (!e) return fail(FX_EINVAL, "fx_element_dims: null element");
    if (sd) *sd = e->sd;
    if (n) *n = e->n;
    if (nexp) *nexp = e->nexp;
    if (ndof) *ndof = e->ndof;
    if (vdim) *vdim = e->vdim;
    return FX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------
namespace {

struct Launch {
    fxk::TabArgs args;
    int grid = 0, lds_bytes = 0;
    // shape-specialised path
    int fixed_id = -1;
    fxk::FixedArgs<0> fhead;           // everything but the coefficients
    std::vector<double> fcoef;         // [nsteps][3]
    std::vector<double> fucoef;        // [nsteps][12], uniform-cell factor derivatives
    int fgrid = 0, flds_bytes = 0, ncu = 0;
    bool fused_mapping = false;
    bool kodd = false;  // stacked kernel: requests of an odd number of doubles (8-byte flush instance)
    int kpiola = 0;     // stacked kernel: Piola map applied to the accumulators (PIO instance), FX_MAP_*
    double* trash = nullptr;
    unsigned long long* queue = nullptr;
    // 0: LDS-image kernel (simplex_fixed.hpp), 1: K-streamed kernel (simplex_stream.hpp),
    // 2: K-streamed, two requests per wave (simplex_pair.hpp)
    int fkind = 0;
    // low-order lane-local kernel (simplex_small.hpp)
    int small_id = -1;
    fxk::SmallArgs sargs;
    int sgrid = 0, slds_bytes = 0;
    // stacked-matrix kernel (simplex_stacked.hpp)
    int stacked_id = -1;
    fxk::StackedArgs<0> khead;
    int kgrid = 0, klds_bytes = 0, kmix_order = 0;
    int wg_ct = 0;      // request-per-workgroup kernel: column tiles of the instance (its slab holds khead.gslab requests)
    int wg_mix = 0;     // ... 1: the order-1 chain rule of per-request cells on its accumulators (dof-major fragments, no mixing pass)
    // cooperative large-shape kernel
    int coop_id = -1;
    fxk::CoopArgs cargs;
    int cgrid = 0, clds_bytes = 0;
};

// ---- registry of shape-specialised kernels ----------------------------------------
// One entry per <SD, N, ORDER, ROWS, NT>; everything else runs the generic kernel.
constexpr int FIXED_NW = 4;  // 256-thread workgroups: one wave per SIMD, 2 workgroups per CU
struct FixedShape {
    int sd, n, order, rows, nt;
    bool pair_only;  // only the paired kernel is instantiated (no A/B partners)
    int rpw;         // requests per wave of the paired kernel: 2, or 1 for shapes with many rows / > 32 points
    bool fullimg;    // LDS image of a whole request (false: half of its tables)
    int nw;          // waves per workgroup of the paired kernel (8 = a whole CU; fewer when the images are large)
    bool can_piola;  // an instance with the Piola map applied to the LDS image exists (vector-valued, per-request cells)
};
const FixedShape kFixedShapes[] = {
    {3, 3, 1, 20, 6, false, 2, true, 8, false},  // Lagrange P3 tetrahedron, values + gradient, 21..24 points (the benchmark shape)
    {3, 3, 1, 20, 5, true, 2, true, 8, false},   // ... 17..20 points
    {3, 3, 1, 20, 4, true, 2, true, 8, false},   // ... 13..16 points
    {3, 3, 1, 20, 3, true, 2, true, 8, false},   // ... 9..12 points
    {3, 3, 1, 20, 8, true, 1, false, 8, false},  // P3 tetrahedron, 25..32 points: one request per wave, half image
    {3, 3, 1, 20, 10, true, 1, false, 8, false}, // ... 33..40 points
    {3, 3, 1, 20, 12, true, 1, false, 8, false}, // ... 41..48 points
    {3, 2, 1, 45, 6, true, 1, false, 8, true},   // RT2 tetrahedron (15 x 3 rows), 21..24 points: one request per wave, half image
    {3, 4, 1, 35, 6, true, 1, false, 8, false},  // Lagrange P4 tetrahedron, 21..24 points
    {3, 2, 1, 60, 6, true, 1, false, 4, true},   // N2 tetrahedron (20 x 3 rows), 21..24 points: 22 KB half images, four waves (one per SIMD: 180 accumulator registers)
    {3, 3, 0, 20, 2, true, 2, true, 8, false},   // Lagrange P3 tetrahedron, values only, 17..32 points (45 % on the stacked kernel -> 56 %)
    {3, 3, 0, 20, 3, true, 1, true, 8, false},   // ... 33..48 points (40 -> 54-58 %; <= 16 points: the stacked kernel is faster)
    {3, 4, 0, 35, 2, true, 2, true, 8, false},   // Lagrange P4 tetrahedron, values only, 17..32 points (even point counts: 35 rows)
    {3, 4, 0, 35, 3, true, 1, true, 8, false},   // ... 33..48 points (the 44-point degree-8 rule: 31 -> 46 %)
};

template <int SD, int N>
bool table_matches(const fx::Program& P) {
    constexpr fxk::StepTable<SD, N> T{};
    if ((int)P.steps.size() != T.count) return false;
    for (int i = 0; i < T.count; ++i)
        if (P.steps[i].dst != T.dst[i] || P.steps[i].cur != T.cur[i] || P.steps[i].prv != T.prv[i] ||
            P.steps[i].codim != T.codim[i])
            return false;
    return true;
}

#if defined(FX_DBG) && (FX_DBG & 512)
// ablation build: lifetimes of the waves of the last launch (written to the scratch area by the kernel)
hipError_t report_wave_lifetimes(const double* trash, int grid, int wg_waves) {
    const int nw = std::min(grid * wg_waves, 3000);
    std::vector<double> t(2 * nw);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    e = hipMemcpy(t.data(), trash + 2048, t.size() * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    std::vector<double> us(nw), mhz(nw);
    for (int i = 0; i < nw; ++i) {
        us[i] = t[2 * i + 1] / 100.0;
        mhz[i] = t[2 * i] / us[i];
    }
    std::vector<double> su = us, sm = mhz;
    std::sort(su.begin(), su.end());
    std::sort(sm.begin(), sm.end());
    fprintf(stderr, "[fiat_amd] %d waves: lifetime us min %.1f p10 %.1f median %.1f p90 %.1f max %.1f | clock MHz min %.0f median %.0f max %.0f\n",
            nw, su[0], su[nw / 10], su[nw / 2], su[nw * 9 / 10], su[nw - 1], sm[0], sm[nw / 2], sm[nw - 1]);
    for (int x = 0; x < 8; ++x) {  // workgroup b runs on XCD b % 8
        std::vector<double> v;
        for (int i = 0; i < nw; ++i)
            if (((i / wg_waves) % 8) == x) v.push_back(us[i]);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        fprintf(stderr, " xcd%d %.0f..%.0f..%.0f", x, v[0], v[v.size() / 2], v.back());
    }
    {   // absolute timeline: when did the waves enter the kernel, when did they leave (relative to the first entry)
        std::vector<double> ab(2 * nw), ini(nw);
        if (hipMemcpy(ab.data(), trash + 2048 + 6000, ab.size() * 8, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(ini.data(), trash + 2048 + 12000, ini.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            double t0 = 1e300, t1 = 0;
            for (int i = 0; i < nw; ++i) {
                t0 = std::min(t0, ab[2 * i]);
                t1 = std::max(t1, ab[2 * i + 1]);
            }
            std::vector<double> st(nw), en(nw);
            for (int i = 0; i < nw; ++i) {
                st[i] = (ab[2 * i] - t0) / 100.0;
                en[i] = (t1 - ab[2 * i + 1]) / 100.0;
            }
            std::vector<double> ss = st, se = en, si = ini;
            std::sort(ss.begin(), ss.end());
            std::sort(se.begin(), se.end());
            std::sort(si.begin(), si.end());
            fprintf(stderr, "\n span %.1f us | entry after first entry: p50 %.1f p90 %.1f max %.1f us | init->first unit: p50 %.1f max %.1f us | idle before the last exit: p10 %.1f p50 %.1f p90 %.1f max %.1f us (mean %.1f)",
                    (t1 - t0) / 100.0, ss[nw / 2], ss[nw * 9 / 10], ss[nw - 1], si[nw / 2] / 100.0, si[nw - 1] / 100.0, se[nw / 10], se[nw / 2],
                    se[nw * 9 / 10], se[nw - 1], std::accumulate(se.begin(), se.end(), 0.0) / nw);
            for (int x = 0; x < 8; ++x) {
                double ms = 0, me = 0;
                int n = 0;
                for (int i = 0; i < nw; ++i)
                    if (((i / wg_waves) % 8) == x) ms += st[i], me += en[i], ++n;
                if (n) fprintf(stderr, "%s xcd%d entry +%.1f idle-at-end %.1f", x == 0 ? "\n" : " |", x, ms / n, me / n);
            }
        }
    }
    fprintf(stderr, "\n workgroup lifetimes (max over its waves), XCD0:");
    for (int b = 0; b < grid; b += 8) {
        double m = 0;
        for (int w = 0; w < wg_waves && b * wg_waves + w < nw; ++w) m = std::max(m, us[b * wg_waves + w]);
        fprintf(stderr, " %.0f", m);
    }
    fprintf(stderr, "\n");
    return hipSuccess;
}
#endif

#ifndef FX_PAIR_BALANCE
#define FX_PAIR_BALANCE 0  // 1: grid of the paired kernel chosen for whole rounds (measured within noise: RT2 152.0 vs 152.1 us, N2 205.0 vs 205.9, C2 265.1 vs 261.9)
#endif
// PAIR_ONLY: only the paired kernel is instantiated for this shape (the A/B partners
// `stream` and `image` exist for the benchmark shape)
template <int SD, int N, int ORDER, int ROWS, int NT, bool PAIR_ONLY = false, int RPW = 2, bool FULLIMG = true, int PAIR_NW = 8,
          bool CAN_PIOLA = false>
int launch_fixed(const Launch& L, hipStream_t s) {
    using KernT = void (*)(const fxk::FixedArgs<fxk::FixedNC<SD, N>::value>);
    constexpr int NC = fxk::FixedNC<SD, N>::value;
    fxk::FixedArgs<NC> fa;
    fa.pts = L.fhead.pts;
    fa.verts = L.fhead.verts;
    fa.out = L.fhead.out;
    fa.afrag = L.fhead.afrag;
    fa.phi0 = L.fhead.phi0;
    memcpy(fa.A0, L.fhead.A0, sizeof fa.A0);
    memcpy(fa.b0, L.fhead.b0, sizeof fa.b0);
    fa.nreq = L.fhead.nreq;
    fa.npts = L.fhead.npts;
    fa.lds_doubles = L.fhead.lds_doubles;
    fa.debug = L.fhead.debug;
    if ((int)L.fcoef.size() != NC || (int)L.fucoef.size() != 4 * NC)
        return fail(FX_EINVAL, "internal: coefficient table size mismatch");
    memcpy(fa.coef, L.fcoef.data(), NC * sizeof(double));
    memcpy(fa.ucoef, L.fucoef.data(), 4 * NC * sizeof(double));
    if (L.fkind == 2) {
        // PAIR_NW = 8: one 512-thread workgroup per CU, two waves per SIMD; LDS work counter
        using KernP = void (*)(const fxk::FixedArgs<NC>, double*, unsigned int*);
        KernP kp = L.fhead.verts ? (KernP)fxk::tabulate_simplex_pair<SD, N, ORDER, ROWS, NT, PAIR_NW, false, RPW, FULLIMG>
                                 : (KernP)fxk::tabulate_simplex_pair<SD, N, ORDER, ROWS, NT, PAIR_NW, true, RPW, FULLIMG>;
        if ((fa.debug >> 16) & 3) {  // fused Piola push-forward
            if constexpr (CAN_PIOLA) kp = (KernP)fxk::tabulate_simplex_pair<SD, N, ORDER, ROWS, NT, PAIR_NW, false, RPW, FULLIMG, true>;
            else return fail(FX_EINVAL, "internal: no fused push-forward for this shape");
        }
        const int lds_bytes = fxk::WQ_CTL_DOUBLES * 8 + (L.flds_bytes - (int)(fa.lds_doubles * 8) * FIXED_NW) + (int)(fa.lds_doubles * 8) * PAIR_NW;
        // attribute and occupancy are properties of (kernel, LDS size): asked once, not per launch
        static thread_local const void* cached_kp = nullptr;
        static thread_local int cached_lds = -1, cached_occ = 0;
        if (cached_kp != reinterpret_cast<const void*>(kp) || cached_lds != lds_bytes) {
            if (lds_bytes > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            int q = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, reinterpret_cast<const void*>(kp), 64 * PAIR_NW, (size_t)lds_bytes));
            cached_kp = reinterpret_cast<const void*>(kp);
            cached_lds = lds_bytes;
            cached_occ = q;
        }
        const int occ = cached_occ;
        const long long nwg = ((L.fhead.nreq + RPW - 1) / RPW + PAIR_NW - 1) / PAIR_NW;   // = chunks of the work queue
        int grid = (int)std::max<long long>(1, std::min<long long>(nwg, (long long)L.ncu * std::max(1, occ)));
#if FX_PAIR_BALANCE
        // Whole rounds: with nwg chunks over `grid` workgroups the last round keeps only nwg mod grid of them busy (RT2 / N2 at
        // 25 000 requests: 1563 chunks over 256 workgroups = 6.1 rounds -- nine tenths of the chip idle through the last one).
        // The kernel is bound by the write path, not by the number of CUs in use: take the grid within 1/8 of the maximum
        // that wastes the fewest workgroup-rounds (224 x 7 = 1568 for RT2, 250 x 25 = 6250 for the 100 000 requests of C2).
        if (nwg > grid) {
            long long best = (nwg + grid - 1) / grid * grid - nwg;
            for (int g = grid - 1; g >= grid - grid / 8 && best > 0; --g) {
                const long long waste = (nwg + g - 1) / g * g - nwg;
                if (waste < best) {
                    best = waste;
                    grid = g;
                }
            }
        }
#endif
        static const bool verbose = ab_env("FIAT_AMD_VERBOSE") != nullptr;
        if (verbose) fprintf(stderr, "[fiat_amd] pair kernel: occupancy %d WG/CU, grid %d, lds %d B\n", occ, grid, lds_bytes);
        hipLaunchKernelGGL(kp, dim3(grid), dim3(64 * PAIR_NW), lds_bytes, s, fa, L.trash, reinterpret_cast<unsigned int*>(L.queue));
        HIP_TRY(hipGetLastError());
#if defined(FX_DBG) && (FX_DBG & 512)
        if (verbose) HIP_TRY(report_wave_lifetimes(L.trash, grid, PAIR_NW));
#endif
        return FX_OK;
    }
    if constexpr (PAIR_ONLY) {
        return fail(FX_EINVAL, "internal: only the paired kernel exists for this shape");
    } else {
        KernT kern;
        if (L.fkind == 1)
            kern = L.fhead.verts ? (KernT)fxk::tabulate_simplex_stream<SD, N, ORDER, ROWS, NT, FIXED_NW, false>
                                 : (KernT)fxk::tabulate_simplex_stream<SD, N, ORDER, ROWS, NT, FIXED_NW, true>;
        else
            kern = (KernT)fxk::tabulate_simplex_fixed<SD, N, ORDER, ROWS, NT, FIXED_NW>;
        if (L.flds_bytes > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        L.flds_bytes));
        const int grid = L.fgrid;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * FIXED_NW), L.flds_bytes, s, fa);
        HIP_TRY(hipGetLastError());
        return FX_OK;
    }
}

// ---- registry of the stacked-matrix kernel: <SD, N, CT, G> ----------------------------------
// (sd, n) of the expansion set, CT column tiles per group of G requests: npts <= 16 CT / G
struct StackedShape {
    int sd, n, ct, g;
    int rtc;  // > 0: instance for exactly rtc row tiles with register-resident A fragments (small shapes); 0: any
              // -2: per-request cells, order 1 on tetrahedra: dof-major tiles, chain rule across the tables in registers
              // -3: per-request cells, order 2: the same with the Hessian tables (accumulator-side mixing, simplex_stacked.hpp MIXR)
              // -4 / -5: point-chunked units with the order-1 / order-2 chain rule inside the kernel
              // -6: per-request cells, order 0, Piola map of a vector-valued element applied in the kernel
              // -1: point-chunked instance (a unit = 16 ct points of one request): any number of points >= 13, odd
              //      table sizes too -- taken when no whole-request instance applies
};
const StackedShape kStackedShapes[] = {
    {3, 3, 3, 2, 5},  // Lagrange P3 tetrahedron, values + gradient (80 stacked rows): the benchmark shape C2, 17..24 points
    {3, 3, 2, 1, 5},  // ... 25..32 points
    {3, 3, 3, 1, 5},  // ... 33..48 points
    // rtc -2: per-request cells, order 1 (tetrahedra and triangles), chain rule applied inside the kernel (simplex_stacked.hpp MIXT;
    // FIAT_AMD_STACKED_MIX=0 switches them off: whole-request instances + table_mix_kernel).  Degree 6 with three
    // column tiles is not registered: 150+ spilled registers make it slower than the two-pass route.
    {3, 6, 2, 1, -2}, {3, 5, 3, 2, -2}, {3, 5, 2, 1, -2}, {3, 5, 3, 1, -2},
    {2, 6, 3, 2, -2}, {2, 6, 2, 1, -2}, {2, 6, 3, 1, -2}, {2, 5, 3, 2, -2}, {2, 5, 2, 1, -2}, {2, 5, 3, 1, -2},
    {3, 4, 3, 2, -2}, {3, 4, 2, 1, -2}, {3, 4, 3, 1, -2}, {3, 3, 3, 2, -2}, {3, 3, 2, 1, -2}, {3, 3, 3, 1, -2},
    {3, 6, 3, 2, 0},  // degree-6 tetrahedron (DG P6 with Hessians: C4), 17..24 points
    {3, 6, 2, 1, 0},  // ... 25..32 points
    {3, 6, 3, 1, 0},  // ... 33..48 points
    {3, 5, 3, 2, 0},  // degree-5 tetrahedron
    {3, 5, 2, 1, 0},
    {3, 5, 3, 1, 0},
    {3, 4, 3, 2, 0},  // degree-4 tetrahedron (with Hessians: 350 stacked rows)
    {3, 4, 2, 1, 0},
    {3, 4, 3, 1, 0},
    {3, 3, 3, 2, 0},  // degree-3 tetrahedron (vector-valued elements, Hessians of P3)
    {3, 3, 2, 1, 0},
    {3, 3, 3, 1, 0},
    {2, 6, 3, 2, 0},  // degree-5 / 6 triangles
    {2, 6, 2, 1, 0},
    {2, 6, 3, 1, 0},
    {2, 5, 3, 2, 0},
    {2, 5, 2, 1, 0},
    {2, 5, 3, 1, 0},
    {3, 6, 3, 3, 0},                   // 13..16 points ((3, 6, 4, 1) would spill 62 registers: 49..64 points are point-chunked)
    // 13..16 and 49..64 points
    {3, 5, 3, 3, 0}, {3, 5, 4, 1, 0},
    {3, 4, 3, 3, 0}, {3, 4, 4, 1, 0},
    {3, 3, 3, 3, 0}, {3, 3, 4, 1, 0},
    {2, 6, 3, 3, 0}, {2, 6, 4, 1, 0},
    {2, 5, 3, 3, 0}, {2, 5, 4, 1, 0},
    {3, 2, 3, 2, 0}, {3, 2, 2, 1, 0}, {3, 2, 3, 1, 0}, {3, 2, 3, 3, 0}, {3, 2, 4, 1, 0},  // degree-2 tetrahedron (N2, RT2, BDM2 ...)
    {3, 6, 3, 1, -1}, {3, 5, 3, 1, -1}, {3, 4, 3, 1, -1}, {3, 3, 3, 1, -1}, {3, 2, 3, 1, -1}, {2, 6, 3, 1, -1}, {2, 5, 3, 1, -1},  // point-chunked
    // round 2 (tools/coverage_map.py): the quadrature sizes of the mid-degree vector-valued elements that were left to the
    // lane-local / generic kernels at 8-30 % of the HBM peak
    {3, 2, 3, 4, 0},                                   // degree-2 tetrahedron, 9..12 points (the 11-point degree-4 rule)
    {2, 4, 3, 3, 0}, {2, 4, 3, 2, 0}, {2, 4, 2, 1, 0}, {2, 4, 3, 1, 0},  // degree-4 triangle (N4 / RT4 / BDM3 ...): 13..16, 17..24, 25..32, 33..48 points
    {2, 3, 3, 4, 0}, {2, 3, 3, 3, 0}, {2, 3, 3, 2, 0}, {2, 3, 2, 1, 0},  // degree-3 triangle: 9..12, 13..16, 17..24, 25..32 points
    {3, 2, 3, 6, 0}, {3, 2, 3, 8, 0}, {3, 2, 3, 12, 0},                  // few points: 7..8, 5..6, <= 4 (six / eight / twelve requests per group)
    {2, 3, 3, 6, 0}, {2, 3, 3, 8, 0}, {2, 4, 3, 4, 0}, {2, 4, 3, 6, 0},
    {2, 4, 3, 1, -1}, {2, 3, 3, 1, -1},                                  // point-chunked: odd table sizes, more points
    // per-request cells, values + gradient, chain rule inside the kernel (as the rtc -2 block above)
    {3, 2, 3, 4, -2}, {3, 2, 3, 3, -2}, {3, 2, 3, 2, -2}, {3, 2, 2, 1, -2},
    {2, 3, 3, 4, -2}, {2, 3, 3, 3, -2}, {2, 3, 3, 2, -2}, {2, 4, 3, 3, -2}, {2, 4, 3, 2, -2}, {2, 4, 2, 1, -2},
    // round 3: expansion degrees 7 and 8 (were on the generic kernel at 11-30 % of the HBM peak).  Tetrahedra of degree 7 hold
    // 30 K-steps x 2 column tiles of B fragments (60 doubles per lane, as degree 6 with three tiles); their quadrature rules
    // have hundreds of points, so the point-chunked instance (32 points a unit) is the one that matters
    {3, 7, 2, 1, 0}, {3, 7, 2, 1, -1},
    {2, 7, 3, 1, 0}, {2, 7, 2, 1, 0}, {2, 7, 3, 1, -1}, {2, 8, 4, 1, 0}, {2, 8, 3, 1, -1},
    // round 3: per-request cells with Hessians, chain rule on the accumulators (rtc -3; were two passes: kernel + table_mix_kernel)
    {3, 2, 3, 4, -3}, {3, 2, 3, 3, -3}, {3, 2, 3, 2, -3}, {3, 2, 2, 1, -3},
    {3, 3, 3, 2, -3}, {3, 3, 2, 1, -3}, {3, 3, 3, 1, -3},
    {3, 4, 3, 2, -3}, {3, 4, 2, 1, -3}, {3, 4, 3, 1, -3},
    {3, 5, 2, 1, -3}, {3, 6, 2, 1, -3},
    {2, 3, 3, 4, -3}, {2, 3, 3, 3, -3}, {2, 3, 3, 2, -3},
    {2, 4, 3, 3, -3}, {2, 4, 3, 2, -3}, {2, 4, 2, 1, -3},
    {2, 5, 3, 2, -3}, {2, 5, 2, 1, -3}, {2, 5, 3, 1, -3},
    {2, 6, 3, 2, -3}, {2, 6, 2, 1, -3}, {2, 6, 3, 1, -3},
    // degree-6 tetrahedra with three column tiles, order 1 with cells: accumulator-side mixing needs ~150 registers fewer than
    // the flush-side version that could not be registered (see the rtc -2 block above): 0 scratch
    {3, 6, 3, 2, -2}, {3, 6, 3, 1, -2},
    // point-chunked units with the chain rule inside the kernel: order 1 (rtc -4), order 2 (rtc -5) -- rules of more points than a
    // whole-request instance holds (the 74- and 122-point rules of P5 / P6 tetrahedra ...), any table size
    {3, 6, 3, 1, -4}, {3, 5, 3, 1, -4}, {3, 4, 3, 1, -4}, {3, 3, 3, 1, -4}, {3, 2, 3, 1, -4}, {2, 6, 3, 1, -4}, {2, 5, 3, 1, -4},
    {3, 6, 2, 1, -5}, {3, 5, 2, 1, -5}, {3, 4, 2, 1, -5}, {3, 3, 3, 1, -5}, {3, 2, 3, 1, -5}, {2, 6, 3, 1, -5}, {2, 5, 3, 1, -5},
    // vector-valued elements on per-request cells with their Piola map, values only (rtc -6): the map is applied to the
    // accumulators (simplex_stacked.hpp PIO; with derivatives: the PIO twins of the rtc -2 / -3 instances of these shapes)
    {3, 2, 3, 4, -6}, {3, 2, 3, 3, -6}, {3, 2, 3, 2, -6}, {3, 2, 2, 1, -6}, {3, 3, 3, 2, -6}, {3, 3, 2, 1, -6}, {3, 3, 3, 1, -6},
    {2, 3, 3, 4, -6}, {2, 3, 3, 3, -6}, {2, 3, 3, 2, -6}, {2, 4, 3, 3, -6}, {2, 4, 3, 2, -6}, {2, 4, 2, 1, -6},
    // point chunks of two column tiles: rules whose 32-point chunks need fewer column tiles than their 48-point chunks
    // (49..64 points: 4 against 6; 97..128 points -- the 122-point rule of degree 6 -- 8 against 9)
    {3, 6, 2, 1, -1}, {3, 5, 2, 1, -1},
    {3, 6, 2, 1, -4}, {3, 5, 2, 1, -4},  // ... with the order-1 chain rule of per-request cells inside
    // round 4, rtc -7: the request-per-workgroup kernel (simplex_wg.hpp, its own translation unit wg.hip) -- rules of 49..128
    // points, expansion values of the whole request in LDS, a row tile leaves as 16 x npts contiguous doubles; ct = the most
    // column tiles an instance has (the launch takes ceil(npts / 16)).  Visited before the point-chunked instances.
    {3, 6, 8, 1, -7}, {3, 5, 8, 1, -7}, {3, 4, 8, 1, -7}, {3, 3, 8, 1, -7}, {2, 6, 8, 1, -7}, {2, 5, 8, 1, -7},
};
// shapes whose in-kernel chain-rule instances (rtc -2 / -3 / -6) have a twin that applies the Piola map too
constexpr bool stacked_has_pio(int sd, int n) { return (sd == 3 && (n == 2 || n == 3)) || (sd == 2 && (n == 3 || n == 4)); }
constexpr int STACKED_NW = 4;  // one wave per SIMD
// Order-1 in-kernel chain rule (rtc -2): on the accumulators (MIXR) or in the flush.  tools/mixr_ab.sh, all 26 instances, ABAB:
// accumulator-side 0.84-1.00 of the flush-side launch time except on four single-request triangle instances (1.01-1.05); of
// those, the degree-5 and -6 ones gain from a second wave per SIMD (below), which only the accumulator-side version has the
// registers for -- the flush-side version stays on <2, 4, 2, 1>.
#ifndef FX_MIXR1
#define FX_MIXR1 1  // 0: flush-side everywhere, 2: accumulator-side everywhere (A/B switch)
#endif
constexpr bool mixr1(int sd, int n, int ct, int g) {
    return FX_MIXR1 == 2 || (FX_MIXR1 != 0 && !(sd == 2 && g == 1 && n == 4 && ct == 2));
}
// Waves per SIMD of the instances with accumulator-side mixing: triangles of degree 5 and 6 need < 256 registers and run two
// (ABAB, orders 1 and 2: 0.83-0.94 of the one-wave launch time on the three-tile instances, 0.98-1.00 on the two-tile ones;
// degrees 3 and 4: 0.97-1.06, left at one)
constexpr int mixr_wps(int sd, int n) { return sd == 2 && n >= 5 ? 2 : 1; }
// chain rule across the derivative tables of requests with their own cells, in place (after a kernel that wrote the
// derivatives with respect to the ELEMENT's cell)
template <int SD>
int stacked_mix_pass(const Launch& L, hipStream_t s) {
    if (L.khead.verts && L.kmix_order >= 1) {
        fxk::TableMixArgs ma;
        ma.out = L.khead.out;
        ma.verts = L.khead.verts;
        if (!invert_small(SD, L.khead.A0, ma.A0inv)) return fail(FX_EINVAL, "degenerate cell");
        const int ntab = fx::binom(SD + L.kmix_order, SD);
        ma.n = L.khead.R / ntab * L.khead.npts;
        ma.order = L.kmix_order;
        ma.slices = std::max(1, std::min(8, (ma.n + 2047) / 2048));
        ma.nreq = L.khead.nreq;
        // small requests: blocks of requests per (persistent) workgroup, ~2048 positions a pass
        ma.rb = ma.n >= 1024 ? 1 : std::max(1, std::min(fxk::MIX_RB, 2048 / std::max(1, ma.n)));
        if (L.khead.nreq * ma.slices > 0x7fffffffLL) return fail(FX_EINVAL, "batch too large for the table-mixing pass");
        if ((long long)fxk::MIX_RB * ma.n > 0x3fffffffLL) return fail(FX_EINVAL, "request too large for the table-mixing pass");
        const long long blocks = ma.rb > 1 ? std::min<long long>((L.khead.nreq + ma.rb - 1) / ma.rb, (long long)L.ncu * 16) : L.khead.nreq * ma.slices;
        hipLaunchKernelGGL((fxk::table_mix_kernel<SD>), dim3((unsigned)std::max<long long>(1, blocks)), dim3(256), 0, s, ma);
        HIP_TRY(hipGetLastError());
    }
    return FX_OK;
}

template <int SD, int N, int CT, int G, int RTC = 0, int WPS = 1, bool CHUNK = false, int MIXT = 0, bool ODD = false, bool MIXR = false, int PIO = 0>
int launch_stacked(const Launch& L, hipStream_t s) {
    constexpr int NC = fxk::FixedNC<SD, N>::value;
    fxk::StackedArgs<NC> ka;
    ka.pts = L.khead.pts;
    ka.verts = L.khead.verts;
    ka.out = L.khead.out;
    ka.afrag = L.khead.afrag;
    ka.phi0 = L.khead.phi0;
    memcpy(ka.A0, L.khead.A0, sizeof ka.A0);
    memcpy(ka.b0, L.khead.b0, sizeof ka.b0);
    memcpy(ka.A0inv, L.khead.A0inv, sizeof ka.A0inv);
    memcpy(ka.G, L.khead.G, sizeof ka.G);
    ka.piola = L.khead.piola;
    ka.nreq = L.khead.nreq;
    ka.npts = L.khead.npts;
    ka.R = L.khead.R;
    ka.RT = L.khead.RT;
    ka.debug = L.khead.debug;
    ka.lim_pts = L.khead.lim_pts;
    ka.lim_verts = L.khead.lim_verts;
    ka.lim_out = L.khead.lim_out;
    ka.lim_afrag = L.khead.lim_afrag;
    if ((int)L.fcoef.size() != NC) return fail(FX_EINVAL, "internal: coefficient table size mismatch");
    memcpy(ka.coef, L.fcoef.data(), NC * sizeof(double));
    auto kern = fxk::tabulate_simplex_stacked<SD, N, CT, G, RTC, WPS, CHUNK, MIXT, ODD, MIXR, PIO>;
    // as many workgroups per CU as registers and LDS allow (low degrees need few registers: their short
    // row sweeps rely on other waves to cover the production phase); asked once per kernel
    static thread_local int occ = 0;
    if (occ == 0) {
        if (L.klds_bytes > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L.klds_bytes));
        int q = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, reinterpret_cast<const void*>(kern), 64 * STACKED_NW, (size_t)L.klds_bytes));
        occ = std::max(1, q);
        static const int cap = ab_env("FIAT_AMD_STACKED_WPS") ? atoi(ab_env("FIAT_AMD_STACKED_WPS")) : 8;
        occ = std::min(occ, std::max(1, cap));
        if (ab_env("FIAT_AMD_VERBOSE")) fprintf(stderr, "[fiat_amd] stacked kernel <%d,%d,%d,%d>: %d workgroups per CU, lds %d B\n", SD, N, CT, G, occ, L.klds_bytes);
    }
    const long long groups = CHUNK ? L.khead.nreq * ((L.khead.npts + 16 * CT - 1) / (16 * CT)) : (L.khead.nreq + G - 1) / G;
    const int grid = (int)std::max<long long>(1, std::min<long long>((groups + STACKED_NW - 1) / STACKED_NW, (long long)L.ncu * occ));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * STACKED_NW), L.klds_bytes, s, ka, L.trash,
                       reinterpret_cast<unsigned int*>(L.queue));
    HIP_TRY(hipGetLastError());
#if defined(FX_DBG) && (FX_DBG & 512)
    if (ab_env("FIAT_AMD_VERBOSE")) HIP_TRY(report_wave_lifetimes(L.trash, grid, STACKED_NW));
#endif
#if defined(FX_DBG) && (FX_DBG & 1024)
    {   // range-check build: report accesses that left their buffers (redirected to the scratch area by the kernel)
        double rep[20];
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(rep, L.trash + 4096, sizeof rep, hipMemcpyDeviceToHost));
        static const char* site[] = {"", "pts", "verts", "afrag", "out"};
        for (int k = 1; k <= 4; ++k) {
            unsigned long long cnt;
            memcpy(&cnt, &rep[4 * k], sizeof cnt);
            if (cnt)
                fprintf(stderr, "[fiat_amd] RANGE CHECK <%d,%d,%d,%d,mixt %d>: %llu accesses outside `%s`: first index %.0f, limit %.0f, group %.0f\n",
                        SD, N, CT, G, MIXT, cnt, site[k], rep[4 * k + 1], rep[4 * k + 2], rep[4 * k + 3]);
        }
        HIP_TRY(hipMemset(L.trash + 4096, 0, sizeof rep));
        if (ab_env("FIAT_AMD_VERBOSE")) fprintf(stderr, "[fiat_amd] range check done <%d,%d,%d,%d,mixt %d>\n", SD, N, CT, G, MIXT);
    }
#endif
    if (MIXT == 0 && PIO == 0) return stacked_mix_pass<SD>(L, s);
    return FX_OK;
}

// the request-per-workgroup kernel (rtc -7; simplex_wg.hpp, compiled in wg.hip)
int launch_wg(const Launch& L, hipStream_t s) {
    const StackedShape& k = kStackedShapes[L.stacked_id];
    const int ct = L.wg_ct;
    hipError_t e = fxwg::launch_simplex_wg(k.sd, k.n, ct, L.kodd, L.wg_mix, L.khead, L.fcoef.data(), (int)L.fcoef.size(), L.klds_bytes, L.kgrid, L.trash,
                                           reinterpret_cast<unsigned int*>(L.queue), s);
    if (e != hipSuccess) return fail(FX_EHIP, "tabulate_simplex_wg<%d,%d,%d>: %s", k.sd, k.n, ct, hipGetErrorString(e));
    if (L.wg_mix) return FX_OK;   // (the chain rule has been applied)
    switch (k.sd) {
        case 2: return stacked_mix_pass<2>(L, s);
        case 3: return stacked_mix_pass<3>(L, s);
    }
    return FX_OK;
}

int run_stacked(const Launch& L, hipStream_t s) {
    // (WPS = 2 variants: instances a few registers above 256 recompiled for two waves per SIMD -- values and gradients have short
    // row sweeps, the second wave covers a group's production phase: P6 triangles 27-35 -> 32-43 %, P4 tetrahedra at 13-24
    // points 28-45 -> 32-50 %; with Hessians the sweep is MFMA-bound and one 512-register wave is faster, tools/wps_probe.py)
    if (L.stacked_id >= 0 && L.stacked_id < (int)(sizeof(kStackedShapes) / sizeof(kStackedShapes[0])) && kStackedShapes[L.stacked_id].rtc == -7)
        return launch_wg(L, s);
    switch (L.stacked_id) {  // (same order as kStackedShapes)
        case 0: return launch_stacked<3, 3, 3, 2, 5, 3>(L, s);
        case 1: return launch_stacked<3, 3, 2, 1, 5, 3>(L, s);
        case 2: return launch_stacked<3, 3, 3, 1, 5, 3>(L, s);
        case 3: return launch_stacked<3, 6, 2, 1, 0, 1, false, 4, false, mixr1(3, 6, 2, 1)>(L, s);
        case 4: return launch_stacked<3, 5, 3, 2, 0, 1, false, 4, false, mixr1(3, 5, 3, 2)>(L, s);
        case 5: return launch_stacked<3, 5, 2, 1, 0, 1, false, 4, false, mixr1(3, 5, 2, 1)>(L, s);
        case 6: return launch_stacked<3, 5, 3, 1, 0, 1, false, 4, false, mixr1(3, 5, 3, 1)>(L, s);
        case 7: return launch_stacked<2, 6, 3, 2, 0, mixr_wps(2, 6), false, 3, false, mixr1(2, 6, 3, 2)>(L, s);
        case 8: return launch_stacked<2, 6, 2, 1, 0, mixr_wps(2, 6), false, 3, false, mixr1(2, 6, 2, 1)>(L, s);
        case 9: return launch_stacked<2, 6, 3, 1, 0, mixr_wps(2, 6), false, 3, false, mixr1(2, 6, 3, 1)>(L, s);
        case 10: return launch_stacked<2, 5, 3, 2, 0, mixr_wps(2, 5), false, 3, false, mixr1(2, 5, 3, 2)>(L, s);
        case 11: return L.kodd ? launch_stacked<2, 5, 2, 1, 0, 1, false, 3, true, true>(L, s) : launch_stacked<2, 5, 2, 1, 0, mixr_wps(2, 5), false, 3, false, mixr1(2, 5, 2, 1)>(L, s);
        case 12: return L.kodd ? launch_stacked<2, 5, 3, 1, 0, 1, false, 3, true, true>(L, s) : launch_stacked<2, 5, 3, 1, 0, mixr_wps(2, 5), false, 3, false, mixr1(2, 5, 3, 1)>(L, s);
        case 13: return L.kodd ? launch_stacked<3, 4, 3, 2, 0, 1, false, 4, true, true>(L, s) : launch_stacked<3, 4, 3, 2, 0, 1, false, 4, false, mixr1(3, 4, 3, 2)>(L, s);
        case 14: return launch_stacked<3, 4, 2, 1, 0, 1, false, 4, false, mixr1(3, 4, 2, 1)>(L, s);
        case 15: return launch_stacked<3, 4, 3, 1, 0, 1, false, 4, false, mixr1(3, 4, 3, 1)>(L, s);
        case 16: return L.kpiola ? (L.kodd ? launch_stacked<3, 3, 3, 2, 0, 1, false, 4, true, true, 1>(L, s) : launch_stacked<3, 3, 3, 2, 0, 1, false, 4, false, true, 1>(L, s)) : L.kodd ? launch_stacked<3, 3, 3, 2, 0, 1, false, 4, true, true>(L, s) : launch_stacked<3, 3, 3, 2, 0, 1, false, 4, false, mixr1(3, 3, 3, 2)>(L, s);
        case 17: return L.kpiola ? launch_stacked<3, 3, 2, 1, 0, 1, false, 4, false, true, 1>(L, s) : launch_stacked<3, 3, 2, 1, 0, 1, false, 4, false, mixr1(3, 3, 2, 1)>(L, s);
        case 18: return L.kpiola ? launch_stacked<3, 3, 3, 1, 0, 1, false, 4, false, true, 1>(L, s) : launch_stacked<3, 3, 3, 1, 0, 1, false, 4, false, mixr1(3, 3, 3, 1)>(L, s);
        case 19: return launch_stacked<3, 6, 3, 2>(L, s);
        case 20: return launch_stacked<3, 6, 2, 1>(L, s);
        case 21: return launch_stacked<3, 6, 3, 1>(L, s);
        case 22: return launch_stacked<3, 5, 3, 2>(L, s);
        case 23: return launch_stacked<3, 5, 2, 1>(L, s);
        case 24: return launch_stacked<3, 5, 3, 1>(L, s);
        case 25: return L.kodd ? launch_stacked<3, 4, 3, 2, 0, 1, false, 0, true>(L, s) : L.kmix_order <= 1 ? launch_stacked<3, 4, 3, 2, 0, 2>(L, s) : launch_stacked<3, 4, 3, 2>(L, s);
        case 26: return launch_stacked<3, 4, 2, 1>(L, s);
        case 27: return launch_stacked<3, 4, 3, 1>(L, s);
        case 28: return L.kodd ? launch_stacked<3, 3, 3, 2, 0, 1, false, 0, true>(L, s) : launch_stacked<3, 3, 3, 2>(L, s);
        case 29: return launch_stacked<3, 3, 2, 1>(L, s);
        case 30: return launch_stacked<3, 3, 3, 1>(L, s);
        case 31: return L.kmix_order <= 1 ? launch_stacked<2, 6, 3, 2, 0, 2>(L, s) : launch_stacked<2, 6, 3, 2>(L, s);
        case 32: return L.kmix_order <= 1 ? launch_stacked<2, 6, 2, 1, 0, 2>(L, s) : launch_stacked<2, 6, 2, 1>(L, s);
        case 33: return L.kmix_order <= 1 ? launch_stacked<2, 6, 3, 1, 0, 2>(L, s) : launch_stacked<2, 6, 3, 1>(L, s);
        case 34: return L.kmix_order <= 1 ? launch_stacked<2, 5, 3, 2, 0, 2>(L, s) : launch_stacked<2, 5, 3, 2>(L, s);
        case 35: return L.kodd ? launch_stacked<2, 5, 2, 1, 0, 1, false, 0, true>(L, s) : L.kmix_order <= 1 ? launch_stacked<2, 5, 2, 1, 0, 2>(L, s) : launch_stacked<2, 5, 2, 1>(L, s);
        case 36: return L.kmix_order <= 1 ? launch_stacked<2, 5, 3, 1, 0, 2>(L, s) : launch_stacked<2, 5, 3, 1>(L, s);
        case 37: return launch_stacked<3, 6, 3, 3>(L, s);
        case 38: return launch_stacked<3, 5, 3, 3>(L, s);
        case 39: return launch_stacked<3, 5, 4, 1>(L, s);
        case 40: return L.kmix_order <= 1 ? launch_stacked<3, 4, 3, 3, 0, 2>(L, s) : launch_stacked<3, 4, 3, 3>(L, s);
        case 41: return launch_stacked<3, 4, 4, 1>(L, s);
        case 42: return launch_stacked<3, 3, 3, 3>(L, s);
        case 43: return launch_stacked<3, 3, 4, 1>(L, s);
        case 44: return launch_stacked<2, 6, 3, 3>(L, s);
        case 45: return launch_stacked<2, 6, 4, 1>(L, s);
        case 46: return launch_stacked<2, 5, 3, 3>(L, s);
        case 47: return launch_stacked<2, 5, 4, 1>(L, s);
        case 48: return L.kodd ? launch_stacked<3, 2, 3, 2, 0, 1, false, 0, true>(L, s) : launch_stacked<3, 2, 3, 2>(L, s);
        case 49: return launch_stacked<3, 2, 2, 1>(L, s);
        case 50: return launch_stacked<3, 2, 3, 1>(L, s);
        case 51: return launch_stacked<3, 2, 3, 3>(L, s);
        case 52: return launch_stacked<3, 2, 4, 1>(L, s);
        case 53: return launch_stacked<3, 6, 3, 1, 0, 1, true>(L, s);
        case 54: return launch_stacked<3, 5, 3, 1, 0, 1, true>(L, s);
        case 55: return launch_stacked<3, 4, 3, 1, 0, 1, true>(L, s);
        case 56: return launch_stacked<3, 3, 3, 1, 0, 1, true>(L, s);
        case 57: return launch_stacked<3, 2, 3, 1, 0, 1, true>(L, s);
        case 58: return L.kmix_order <= 1 ? launch_stacked<2, 6, 3, 1, 0, 2, true>(L, s) : launch_stacked<2, 6, 3, 1, 0, 1, true>(L, s);
        case 59: return L.kmix_order <= 1 ? launch_stacked<2, 5, 3, 1, 0, 2, true>(L, s) : launch_stacked<2, 5, 3, 1, 0, 1, true>(L, s);
        case 60: return L.kodd ? launch_stacked<3, 2, 3, 4, 0, 1, false, 0, true>(L, s) : launch_stacked<3, 2, 3, 4>(L, s);
        case 61: return launch_stacked<2, 4, 3, 3>(L, s);
        case 62: return launch_stacked<2, 4, 3, 2>(L, s);
        case 63: return launch_stacked<2, 4, 2, 1>(L, s);
        case 64: return launch_stacked<2, 4, 3, 1>(L, s);
        case 65: return launch_stacked<2, 3, 3, 4>(L, s);
        case 66: return launch_stacked<2, 3, 3, 3>(L, s);
        case 67: return launch_stacked<2, 3, 3, 2>(L, s);
        case 68: return launch_stacked<2, 3, 2, 1>(L, s);
        case 69: return launch_stacked<3, 2, 3, 6>(L, s);
        case 70: return launch_stacked<3, 2, 3, 8>(L, s);
        case 71: return launch_stacked<3, 2, 3, 12>(L, s);
        case 72: return launch_stacked<2, 3, 3, 6>(L, s);
        case 73: return launch_stacked<2, 3, 3, 8>(L, s);
        case 74: return launch_stacked<2, 4, 3, 4>(L, s);
        case 75: return launch_stacked<2, 4, 3, 6>(L, s);
        case 76: return launch_stacked<2, 4, 3, 1, 0, 1, true>(L, s);
        case 77: return launch_stacked<2, 3, 3, 1, 0, 1, true>(L, s);
        case 78: return L.kpiola ? (L.kodd ? launch_stacked<3, 2, 3, 4, 0, 1, false, 4, true, true, 1>(L, s) : launch_stacked<3, 2, 3, 4, 0, 1, false, 4, false, true, 1>(L, s)) : launch_stacked<3, 2, 3, 4, 0, 1, false, 4, false, mixr1(3, 2, 3, 4)>(L, s);
        case 79: return L.kpiola ? launch_stacked<3, 2, 3, 3, 0, 1, false, 4, false, true, 1>(L, s) : launch_stacked<3, 2, 3, 3, 0, 1, false, 4, false, mixr1(3, 2, 3, 3)>(L, s);
        case 80: return L.kpiola ? launch_stacked<3, 2, 3, 2, 0, 1, false, 4, false, true, 1>(L, s) : launch_stacked<3, 2, 3, 2, 0, 1, false, 4, false, mixr1(3, 2, 3, 2)>(L, s);
        case 81: return L.kpiola ? launch_stacked<3, 2, 2, 1, 0, 1, false, 4, false, true, 1>(L, s) : launch_stacked<3, 2, 2, 1, 0, 1, false, 4, false, mixr1(3, 2, 2, 1)>(L, s);
        case 82: return L.kpiola ? launch_stacked<2, 3, 3, 4, 0, 1, false, 3, false, true, 1>(L, s) : launch_stacked<2, 3, 3, 4, 0, mixr_wps(2, 3), false, 3, false, mixr1(2, 3, 3, 4)>(L, s);
        case 83: return L.kpiola ? launch_stacked<2, 3, 3, 3, 0, 1, false, 3, false, true, 1>(L, s) : launch_stacked<2, 3, 3, 3, 0, mixr_wps(2, 3), false, 3, false, mixr1(2, 3, 3, 3)>(L, s);
        case 84: return L.kpiola ? launch_stacked<2, 3, 3, 2, 0, 1, false, 3, false, true, 1>(L, s) : launch_stacked<2, 3, 3, 2, 0, mixr_wps(2, 3), false, 3, false, mixr1(2, 3, 3, 2)>(L, s);
        case 85: return L.kpiola ? launch_stacked<2, 4, 3, 3, 0, 1, false, 3, false, true, 1>(L, s) : launch_stacked<2, 4, 3, 3, 0, mixr_wps(2, 4), false, 3, false, mixr1(2, 4, 3, 3)>(L, s);
        case 86: return L.kpiola ? launch_stacked<2, 4, 3, 2, 0, 1, false, 3, false, true, 1>(L, s) : launch_stacked<2, 4, 3, 2, 0, mixr_wps(2, 4), false, 3, false, mixr1(2, 4, 3, 2)>(L, s);
        case 87: return L.kpiola ? launch_stacked<2, 4, 2, 1, 0, 1, false, 3, false, true, 1>(L, s) : L.kodd ? launch_stacked<2, 4, 2, 1, 0, 1, false, 3, true, true>(L, s) : launch_stacked<2, 4, 2, 1, 0, mixr_wps(2, 4), false, 3, false, mixr1(2, 4, 2, 1)>(L, s);
        case 88: return launch_stacked<3, 7, 2, 1>(L, s);
        case 89: return launch_stacked<3, 7, 2, 1, 0, 1, true>(L, s);
        case 90: return launch_stacked<2, 7, 3, 1>(L, s);
        case 91: return launch_stacked<2, 7, 2, 1>(L, s);
        case 92: return launch_stacked<2, 7, 3, 1, 0, 1, true>(L, s);
        case 93: return launch_stacked<2, 8, 4, 1>(L, s);
        case 94: return launch_stacked<2, 8, 3, 1, 0, 1, true>(L, s);
        case 95: return L.kpiola ? (L.kodd ? launch_stacked<3, 2, 3, 4, 0, 1, false, 10, true, true, 1>(L, s) : launch_stacked<3, 2, 3, 4, 0, 1, false, 10, false, true, 1>(L, s)) : L.kodd ? launch_stacked<3, 2, 3, 4, 0, 1, false, 10, true, true>(L, s) : launch_stacked<3, 2, 3, 4, 0, 1, false, 10, false, true>(L, s);
        case 96: return L.kpiola ? launch_stacked<3, 2, 3, 3, 0, 1, false, 10, false, true, 1>(L, s) : launch_stacked<3, 2, 3, 3, 0, 1, false, 10, false, true>(L, s);
        case 97: return L.kpiola ? launch_stacked<3, 2, 3, 2, 0, 1, false, 10, false, true, 1>(L, s) : launch_stacked<3, 2, 3, 2, 0, 1, false, 10, false, true>(L, s);
        case 98: return L.kpiola ? launch_stacked<3, 2, 2, 1, 0, 1, false, 10, false, true, 1>(L, s) : launch_stacked<3, 2, 2, 1, 0, 1, false, 10, false, true>(L, s);
        case 99: return L.kpiola ? (L.kodd ? launch_stacked<3, 3, 3, 2, 0, 1, false, 10, true, true, 1>(L, s) : launch_stacked<3, 3, 3, 2, 0, 1, false, 10, false, true, 1>(L, s)) : L.kodd ? launch_stacked<3, 3, 3, 2, 0, 1, false, 10, true, true>(L, s) : launch_stacked<3, 3, 3, 2, 0, 1, false, 10, false, true>(L, s);
        case 100: return L.kpiola ? launch_stacked<3, 3, 2, 1, 0, 1, false, 10, false, true, 1>(L, s) : launch_stacked<3, 3, 2, 1, 0, 1, false, 10, false, true>(L, s);
        case 101: return L.kpiola ? launch_stacked<3, 3, 3, 1, 0, 1, false, 10, false, true, 1>(L, s) : launch_stacked<3, 3, 3, 1, 0, 1, false, 10, false, true>(L, s);
        case 102: return L.kodd ? launch_stacked<3, 4, 3, 2, 0, 1, false, 10, true, true>(L, s) : launch_stacked<3, 4, 3, 2, 0, 1, false, 10, false, true>(L, s);
        case 103: return launch_stacked<3, 4, 2, 1, 0, 1, false, 10, false, true>(L, s);
        case 104: return launch_stacked<3, 4, 3, 1, 0, 1, false, 10, false, true>(L, s);
        case 105: return launch_stacked<3, 5, 2, 1, 0, 1, false, 10, false, true>(L, s);
        case 106: return launch_stacked<3, 6, 2, 1, 0, 1, false, 10, false, true>(L, s);
        case 107: return L.kpiola ? launch_stacked<2, 3, 3, 4, 0, 1, false, 6, false, true, 1>(L, s) : launch_stacked<2, 3, 3, 4, 0, mixr_wps(2, 3), false, 6, false, true>(L, s);
        case 108: return L.kpiola ? launch_stacked<2, 3, 3, 3, 0, 1, false, 6, false, true, 1>(L, s) : launch_stacked<2, 3, 3, 3, 0, mixr_wps(2, 3), false, 6, false, true>(L, s);
        case 109: return L.kpiola ? launch_stacked<2, 3, 3, 2, 0, 1, false, 6, false, true, 1>(L, s) : launch_stacked<2, 3, 3, 2, 0, mixr_wps(2, 3), false, 6, false, true>(L, s);
        case 110: return L.kpiola ? launch_stacked<2, 4, 3, 3, 0, 1, false, 6, false, true, 1>(L, s) : launch_stacked<2, 4, 3, 3, 0, mixr_wps(2, 4), false, 6, false, true>(L, s);
        case 111: return L.kpiola ? launch_stacked<2, 4, 3, 2, 0, 1, false, 6, false, true, 1>(L, s) : launch_stacked<2, 4, 3, 2, 0, mixr_wps(2, 4), false, 6, false, true>(L, s);
        case 112: return L.kpiola ? launch_stacked<2, 4, 2, 1, 0, 1, false, 6, false, true, 1>(L, s) : L.kodd ? launch_stacked<2, 4, 2, 1, 0, 1, false, 6, true, true>(L, s) : launch_stacked<2, 4, 2, 1, 0, mixr_wps(2, 4), false, 6, false, true>(L, s);
        case 113: return launch_stacked<2, 5, 3, 2, 0, mixr_wps(2, 5), false, 6, false, true>(L, s);
        case 114: return L.kodd ? launch_stacked<2, 5, 2, 1, 0, 1, false, 6, true, true>(L, s) : launch_stacked<2, 5, 2, 1, 0, mixr_wps(2, 5), false, 6, false, true>(L, s);
        case 115: return L.kodd ? launch_stacked<2, 5, 3, 1, 0, 1, false, 6, true, true>(L, s) : launch_stacked<2, 5, 3, 1, 0, mixr_wps(2, 5), false, 6, false, true>(L, s);
        case 116: return launch_stacked<2, 6, 3, 2, 0, mixr_wps(2, 6), false, 6, false, true>(L, s);
        case 117: return launch_stacked<2, 6, 2, 1, 0, mixr_wps(2, 6), false, 6, false, true>(L, s);
        case 118: return launch_stacked<2, 6, 3, 1, 0, mixr_wps(2, 6), false, 6, false, true>(L, s);
        case 119: return launch_stacked<3, 6, 3, 2, 0, 1, false, 4, false, true>(L, s);
        case 120: return launch_stacked<3, 6, 3, 1, 0, 1, false, 4, false, true>(L, s);
        case 121: return launch_stacked<3, 6, 3, 1, 0, 1, true, 4, false, true>(L, s);
        case 122: return launch_stacked<3, 5, 3, 1, 0, 1, true, 4, false, true>(L, s);
        case 123: return launch_stacked<3, 4, 3, 1, 0, 1, true, 4, false, true>(L, s);
        case 124: return launch_stacked<3, 3, 3, 1, 0, 1, true, 4, false, true>(L, s);
        case 125: return launch_stacked<3, 2, 3, 1, 0, 1, true, 4, false, true>(L, s);
        case 126: return launch_stacked<2, 6, 3, 1, 0, 1, true, 3, false, true>(L, s);  // (two waves per SIMD: 116-364 B of scratch)
        case 127: return launch_stacked<2, 5, 3, 1, 0, 1, true, 3, false, true>(L, s);  // (two waves per SIMD: 116-364 B of scratch)
        case 128: return launch_stacked<3, 6, 2, 1, 0, 1, true, 10, false, true>(L, s);
        case 129: return launch_stacked<3, 5, 2, 1, 0, 1, true, 10, false, true>(L, s);
        case 130: return launch_stacked<3, 4, 2, 1, 0, 1, true, 10, false, true>(L, s);
        case 131: return launch_stacked<3, 3, 3, 1, 0, 1, true, 10, false, true>(L, s);
        case 132: return launch_stacked<3, 2, 3, 1, 0, 1, true, 10, false, true>(L, s);
        case 133: return launch_stacked<2, 6, 3, 1, 0, 1, true, 6, false, true>(L, s);  // (two waves per SIMD: 116-364 B of scratch)
        case 134: return launch_stacked<2, 5, 3, 1, 0, 1, true, 6, false, true>(L, s);  // (two waves per SIMD: 116-364 B of scratch)
        case 135: return L.kodd ? launch_stacked<3, 2, 3, 4, 0, 1, false, 1, true, true, 1>(L, s) : launch_stacked<3, 2, 3, 4, 0, 1, false, 1, false, true, 1>(L, s);
        case 136: return launch_stacked<3, 2, 3, 3, 0, 1, false, 1, false, true, 1>(L, s);
        case 137: return launch_stacked<3, 2, 3, 2, 0, 1, false, 1, false, true, 1>(L, s);
        case 138: return launch_stacked<3, 2, 2, 1, 0, 1, false, 1, false, true, 1>(L, s);
        case 139: return L.kodd ? launch_stacked<3, 3, 3, 2, 0, 1, false, 1, true, true, 1>(L, s) : launch_stacked<3, 3, 3, 2, 0, 1, false, 1, false, true, 1>(L, s);
        case 140: return launch_stacked<3, 3, 2, 1, 0, 1, false, 1, false, true, 1>(L, s);
        case 141: return launch_stacked<3, 3, 3, 1, 0, 1, false, 1, false, true, 1>(L, s);
        case 142: return launch_stacked<2, 3, 3, 4, 0, 1, false, 1, false, true, 1>(L, s);
        case 143: return launch_stacked<2, 3, 3, 3, 0, 1, false, 1, false, true, 1>(L, s);
        case 144: return launch_stacked<2, 3, 3, 2, 0, 1, false, 1, false, true, 1>(L, s);
        case 145: return launch_stacked<2, 4, 3, 3, 0, 1, false, 1, false, true, 1>(L, s);
        case 146: return launch_stacked<2, 4, 3, 2, 0, 1, false, 1, false, true, 1>(L, s);
        case 147: return launch_stacked<2, 4, 2, 1, 0, 1, false, 1, false, true, 1>(L, s);
        case 148: return launch_stacked<3, 6, 2, 1, 0, 1, true>(L, s);
        case 149: return launch_stacked<3, 5, 2, 1, 0, 1, true>(L, s);
        case 150: return launch_stacked<3, 6, 2, 1, 0, 1, true, 4, false, true>(L, s);
        case 151: return launch_stacked<3, 5, 2, 1, 0, 1, true, 4, false, true>(L, s);
    }
    return fail(FX_EINVAL, "internal: unknown stacked kernel %d", L.stacked_id);
}

int ensure_stacked(fx_ctx* ctx, fx_element* e, int order);  // defined after the C entry points it uses
int ensure_high_order(fx_ctx* ctx, fx_element* e, int order);
int mix_high_order(fx_ctx* ctx, const fx_element* e, int order, int64_t nreq, int npts, const double* verts, double* out, hipStream_t s);

// ---- registry of cooperative (large-shape) kernels: <SD, ORDER, MT16, M4, TPW> ------
struct CoopShape {
    int sd, order, mt16, m4, tpw;
    bool can_piola;   // a kernel with the Piola map fused into the output rounds is instantiated
    bool piola_only;  // registered only for launches that fuse a Piola map (slower than the generic kernel otherwise)
};
const CoopShape kCoopShapes[] = {
    {3, 2, 5, 1, 4, false, false},  // DG P6 tetrahedron (84 rows) with Hessians, <= 25 points
    {3, 1, 5, 1, 2, false, false},  // DG P6 tetrahedron, values + gradient, <= 32 points
    {3, 1, 3, 3, 2, true, false},   // N2 tetrahedron (20 x 3 = 60 rows), values + gradient: 319 vs 431 us per 25 000 requests
    {3, 1, 3, 0, 2, true, true},    // RT2 tetrahedron (15 x 3 = 45 rows): 349 vs 252 us on the generic kernel, but the
                             // Piola map fuses into this kernel's output rounds (generic + second pass: 616 us)
};

template <int SD, int ORDER, int MT16, int M4, int TPW, bool CAN_PIOLA = false>
int launch_coop(const Launch& L, hipStream_t s) {
    using KernT = void (*)(const fxk::CoopArgs);
    KernT kern = L.cargs.verts ? (KernT)fxk::tabulate_simplex_coop<SD, ORDER, MT16, M4, TPW, false>
                               : (KernT)fxk::tabulate_simplex_coop<SD, ORDER, MT16, M4, TPW, true>;
    if (L.cargs.piola) {
        if constexpr (CAN_PIOLA) kern = (KernT)fxk::tabulate_simplex_coop<SD, ORDER, MT16, M4, TPW, false, true>;
        else return fail(FX_EINVAL, "internal: no fused push-forward for this cooperative shape");
    }
    if (L.clds_bytes > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    L.clds_bytes));
    hipLaunchKernelGGL(kern, dim3(L.cgrid), dim3(512), L.clds_bytes, s, L.cargs);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

int run_coop(const Launch& L, hipStream_t s) {
    switch (L.coop_id) {
        case 0: return launch_coop<3, 2, 5, 1, 4>(L, s);
        case 1: return launch_coop<3, 1, 5, 1, 2>(L, s);
        case 2: return launch_coop<3, 1, 3, 3, 2, true>(L, s);
        case 3: return launch_coop<3, 1, 3, 0, 2, true>(L, s);
    }
    return fail(FX_EINVAL, "internal: unknown cooperative kernel %d", L.coop_id);
}

// ---- registry of the low-order lane-local kernel: (sd, n), orders 0 and 1 -----------------
struct SmallShape {
    int sd, n;
};
// (measured against the generic kernel: (2,5) and (3,3) are slower lane-local -- 1600+ FMAs per lane -- and stay generic)
const SmallShape kSmallShapes[] = {{2, 1}, {2, 2}, {2, 3}, {3, 1}, {3, 2}, {2, 4}, {2, 0}, {3, 0},  // degree 0 (P0 / DG0: one member, no steps): 6-19 % on the generic kernel
                                   // round 4, VALUES ONLY: rows x members FMAs per lane stay below the HBM time of the lane's output up to ~56
                                   // members (members / 8 FMAs per output byte), while the MFMA tiles of the other kernels are a third
                                   // full at these row / point counts (P5 triangles, 21 rows x 25 points: 38 %)
                                   // Measured, sustained runs, 0.8 GB (tools/instance_ab.py --own-cell [--policy no_small_values]): P5 triangles
                                   // 345 / 264 / 247 / 551 us -> 222 / 179 / 180 / 218 at 25 / 12 / 30 / 7 points; P3 tetrahedra at 14 points
                                   // 208 -> 165 (at 23 / 44 points the paired kernel keeps them: 187 / 161 against 176 / 178).  Tried and NOT
                                   // registered: P6 triangles (0.82 ... 1.12 x by point count), P4 tetrahedra (1.1-1.4 x), P5 tetrahedra
                                   // (3.7-5.4 x: 358 spilled scalar registers around the 56 x 56 coefficient stream)
                                   {2, 5}, {3, 3}};
constexpr int SMALL_NW = 4;

template <int SD, int N>
int launch_small(int order, const Launch& L, hipStream_t s) {
    if (L.sargs.piola) {  // Piola map of the request's cell fused into the contraction
        if constexpr (SD >= 2) {
            if (order == 0)
                hipLaunchKernelGGL((fxk::tabulate_simplex_small<SD, N, 0, SMALL_NW, true>), dim3(L.sgrid), dim3(64 * SMALL_NW), L.slds_bytes, s, L.sargs);
            else if (order == 1)
                hipLaunchKernelGGL((fxk::tabulate_simplex_small<SD, N, 1, SMALL_NW, true>), dim3(L.sgrid), dim3(64 * SMALL_NW), L.slds_bytes, s, L.sargs);
            else if constexpr (!(SD == 3 && N == 2))
                hipLaunchKernelGGL((fxk::tabulate_simplex_small<SD, N, 2, SMALL_NW, true>), dim3(L.sgrid), dim3(64 * SMALL_NW), L.slds_bytes, s, L.sargs);
            else
                return fail(FX_EINVAL, "internal: no fused push-forward for this lane-local shape");
            HIP_TRY(hipGetLastError());
            return FX_OK;
        }
    }
    if (order == 0)
        hipLaunchKernelGGL((fxk::tabulate_simplex_small<SD, N, 0, SMALL_NW>), dim3(L.sgrid), dim3(64 * SMALL_NW), L.slds_bytes, s, L.sargs);
    else if (order == 1)
        hipLaunchKernelGGL((fxk::tabulate_simplex_small<SD, N, 1, SMALL_NW>), dim3(L.sgrid), dim3(64 * SMALL_NW), L.slds_bytes, s, L.sargs);
    else
        hipLaunchKernelGGL((fxk::tabulate_simplex_small<SD, N, 2, SMALL_NW>), dim3(L.sgrid), dim3(64 * SMALL_NW), L.slds_bytes, s, L.sargs);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

int run_small(int order, const Launch& L, hipStream_t s) {
    switch (L.small_id) {
        case 0: return launch_small<2, 1>(order, L, s);
        case 1: return launch_small<2, 2>(order, L, s);
        case 2: return launch_small<2, 3>(order, L, s);
        case 3: return launch_small<3, 1>(order, L, s);
        case 4: return launch_small<3, 2>(order, L, s);
        case 5: return launch_small<2, 4>(order, L, s);
        case 6: return launch_small<2, 0>(order, L, s);
        case 7: return launch_small<3, 0>(order, L, s);
        case 8: case 9:
            if (order != 0 || L.sargs.piola) break;
            if (L.small_id == 8) hipLaunchKernelGGL((fxk::tabulate_simplex_small<2, 5, 0, SMALL_NW>), dim3(L.sgrid), dim3(64 * SMALL_NW), L.slds_bytes, s, L.sargs);
            if (L.small_id == 9) hipLaunchKernelGGL((fxk::tabulate_simplex_small<3, 3, 0, SMALL_NW>), dim3(L.sgrid), dim3(64 * SMALL_NW), L.slds_bytes, s, L.sargs);
            HIP_TRY(hipGetLastError());
            return FX_OK;
    }
    return fail(FX_EINVAL, "internal: unknown small kernel %d", L.small_id);
}

bool small_table_matches(int id, const fx::Program& P) {
    switch (id) {
        case 0: return table_matches<2, 1>(P);
        case 1: return table_matches<2, 2>(P);
        case 2: return table_matches<2, 3>(P);
        case 3: return table_matches<3, 1>(P);
        case 4: return table_matches<3, 2>(P);
        case 5: return table_matches<2, 4>(P);
        case 6: return table_matches<2, 0>(P);
        case 7: return table_matches<3, 0>(P);
        case 8: return table_matches<2, 5>(P);
        case 9: return table_matches<3, 3>(P);
    }
    return false;
}

int run_fixed(const Launch& L, hipStream_t s) {
    switch (L.fixed_id) {
        case 0: return launch_fixed<3, 3, 1, 20, 6>(L, s);
        case 1: return launch_fixed<3, 3, 1, 20, 5, true>(L, s);
        case 2: return launch_fixed<3, 3, 1, 20, 4, true>(L, s);
        case 3: return launch_fixed<3, 3, 1, 20, 3, true>(L, s);
        case 4: return launch_fixed<3, 3, 1, 20, 8, true, 1, false>(L, s);
        case 5: return launch_fixed<3, 3, 1, 20, 10, true, 1, false>(L, s);
        case 6: return launch_fixed<3, 3, 1, 20, 12, true, 1, false>(L, s);
        case 7: return launch_fixed<3, 2, 1, 45, 6, true, 1, false, 8, true>(L, s);
        case 8: return launch_fixed<3, 4, 1, 35, 6, true, 1, false>(L, s);
        case 9: return launch_fixed<3, 2, 1, 60, 6, true, 1, false, 4, true>(L, s);
        case 10: return launch_fixed<3, 3, 0, 20, 2, true>(L, s);
        case 11: return launch_fixed<3, 3, 0, 20, 3, true, 1>(L, s);
        case 12: return launch_fixed<3, 4, 0, 35, 2, true>(L, s);
        case 13: return launch_fixed<3, 4, 0, 35, 3, true, 1>(L, s);
    }
    return fail(FX_EINVAL, "internal: unknown fixed kernel %d", L.fixed_id);
}

template <int SD, int ORDER, int KS_T, int MT_T>
int launch_one(const Launch& L, hipStream_t s) {
    auto kern = fxk::tabulate_simplex_kernel<SD, ORDER, 1, KS_T, MT_T>;
    if (L.lds_bytes > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    L.lds_bytes));
    }
    hipLaunchKernelGGL(kern, dim3(L.grid), dim3(64), L.lds_bytes, s, L.args);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

template <int SD>
int launch_sd(int order, const Launch& L, hipStream_t s) {
    const int KS = L.args.KS, MT = L.args.MT;
    switch (order) {
        case 0:
            return launch_one<SD, 0, 0, 0>(L, s);
        case 1:
            if (SD == 3 && KS == 5 && MT == 2) return launch_one<SD, 1, 5, 2>(L, s);  // P3 tet
            if (SD == 3 && KS == 3 && MT == 4) return launch_one<SD, 1, 3, 4>(L, s);  // N2 tet
            if (SD == 3 && KS == 3 && MT == 3) return launch_one<SD, 1, 3, 3>(L, s);  // RT2 tet
            return launch_one<SD, 1, 0, 0>(L, s);
        case 2:
            return launch_one<SD, 2, 0, 0>(L, s);
    }
    return fail(FX_ENOTIMPL, "derivative order %d > 2 is not implemented on the device", order);
}

// `mapping` != FX_MAP_AFFINE asks for a kernel that fuses the Piola push-forward (L.fused_mapping tells
// whether one was found; otherwise the caller runs fx_pushforward_batch after the launch)
int plan_launch(fx_ctx* ctx, const fx_element* e, int order, int64_t nreq, int npts, const double* pts,
                const double* verts, double* out, Launch& L, int mapping = 0) {
    if (!ctx || !e) return fail(FX_EINVAL, "null context/element");
    if (order < 0) return fail(FX_EINVAL, "negative derivative order");
    if (order > 2) return fail(FX_ENOTIMPL, "derivative order %d > 2 is not implemented on the device", order);
    if (nreq < 0 || npts < 0) return fail(FX_EINVAL, "negative batch size");
    if ((nreq > 0 && npts > 0) && (!pts || !out)) return fail(FX_EINVAL, "null device pointer");
    const int ntab = fx::binom(e->sd + order, e->sd);
    const int rows = e->ndof * e->vdim;
    // the element's Piola map asked for with the tabulation, and can the stacked kernel apply it to its accumulators (PIO
    // instances: whole requests of <= 48 points, even table sizes)?  Then the cooperative and the lane-local kernel leave
    // their fused variants of these shapes to it (coverage map with the map: N2 tetrahedra at 11 points, gradients,
    // 14 % of the HBM peak on the cooperative kernel against 37 %; N3 triangles 29 % lane-local against 50 %)
    const bool want_piola = (mapping == FX_MAP_COVARIANT_PIOLA || mapping == FX_MAP_CONTRAVARIANT_PIOLA) && verts && e->vdim == e->sd && e->sd >= 2;
    const int TRp = fxk::stacked_tile_rows(e->sd, 1);
    const bool pio_even = !(((long long)rows * npts) % 2) && !(((long long)(rows - TRp * ((rows + TRp - 1) / TRp - 1)) * npts) % 2);
    // (odd table sizes: 8-byte twins of the two instances RT2 / N3 tetrahedra meet at their 11- / 23-point rules)
    const bool pio_odd_twin = e->sd == 3 && ((e->n == 2 && npts > 8 && npts <= 12) || (e->n == 3 && npts > 16 && npts <= 24));
    // ... and is there an instance that holds THIS request: the same predicates as the selection loop below (column budget of a
    // request, more than two thirds of the group's column tiles filled) -- without the probe the cooperative and the lane-local
    // kernel gave up their fused variants for shapes no stacked instance then took (N2 / RT2 tetrahedra at <= 8 points: generic
    // kernel + a separate push-forward pass)
    bool pio_instance = false;
    for (const StackedShape& k : kStackedShapes)
        pio_instance = pio_instance || (k.sd == e->sd && k.n == e->n && k.rtc == (order == 0 ? -6 : order == 1 ? -2 : -3) &&
                                        npts <= 16 * k.ct / k.g && 3LL * k.g * npts > 2LL * 16 * k.ct);
    const bool stacked_pio_ok = want_piola && !(ctx->policy & (FX_POLICY_NO_STACKED | FX_POLICY_NO_STACKED_MIX)) && !e->raw_expansion &&
                                stacked_has_pio(e->sd, e->n) && npts <= 48 && (pio_even || pio_odd_twin) && (long long)ntab * rows >= 15 &&
                                pio_instance && order <= 2;
    fxk::TabArgs& a = L.args;
    memset(&a, 0, sizeof a);
    a.pts = pts;
    a.verts = verts;
    a.out = out;
    a.afrag = e->d_afrag;
    a.steps = e->d_steps;
    a.phi0 = e->prog.phi0;
    memcpy(a.A0, e->A0, sizeof a.A0);
    memcpy(a.b0, e->b0, sizeof a.b0);
    a.nreq = nreq;
    a.npts = npts;
    a.rows = rows;
    a.nexp = e->nexp;
    a.nsteps = (int)e->prog.steps.size();
    a.KS = e->KS;
    a.MT = e->MT;
    a.ntab = ntab;
    if (const char* dbg = ab_env("FIAT_AMD_DEBUG")) a.debug = atoi(dbg);  // ablation switches (measurement only)
    if (nreq == 0 || npts == 0) {  // empty batch / empty point set: nothing to launch
        a.nitems = 0;
        L.grid = 0;
        return FX_OK;
    }

    // ---- choose the work-item shape under a per-wave LDS budget ----
    const long long budget = 32 * 1024;  // bytes per wave: >= 5 resident waves per CU
    // (hard limit: one wave's tile may take what a CU has -- the kernel runs one wave per workgroup; reached by expansion
    // degrees beyond 12 on tetrahedra, one column tile of 16 x nexp doubles: degree 16, 969 members, 124 KB)
    const long long hard = std::max<long long>(64 * 1024, (long long)ctx->lds_per_cu - 4096);
    auto phi_bytes = [&](long long cols) { return ((cols + 15) / 16) * (long long)e->KS * 64 * 8; };
    const long long reqbytes = (long long)ntab * rows * npts * 8;
    int P = 1, pc = npts, nchunk = 1;
    long long stage = 0;
    if (npts <= 64 && phi_bytes((long long)ntab * npts) + reqbytes <= budget) {
        // whole requests, output staged through LDS; pack as many as fit in a wave
        int pmax = 64 / npts;
        P = 1;
        for (int p = 2; p <= pmax; ++p)
            if (phi_bytes((long long)p * ntab * npts) + p * reqbytes <= budget) P = p;
        stage = P * reqbytes;
    } else {
        // point-chunked, direct stores
        pc = std::min(npts, 64);
        while (pc > 1 && phi_bytes((long long)ntab * pc) > budget) --pc;
        if (phi_bytes((long long)ntab * pc) > hard)
            return fail(FX_ENOTIMPL, "expansion degree too large for the LDS tile (nexp=%d, ntab=%d)", e->nexp, ntab);
        nchunk = (npts + pc - 1) / pc;
    }
    a.P = P;
    a.pc = pc;
    a.nchunk = nchunk;
    a.nitems = nchunk > 1 ? nreq * nchunk : (nreq + P - 1) / P;
    a.phi_doubles = (int)(phi_bytes((long long)P * ntab * pc) / 8);
    a.stage_doubles = (int)((stage / 8 + 1) & ~1LL);
    if (stage == 0) a.stage_doubles = 0;
    L.lds_bytes = (a.phi_doubles + a.stage_doubles) * 8;
    // ---- cooperative kernel for large shapes? ----
    L.coop_id = -1;
    {
        const bool nocoop = (ctx->policy & FX_POLICY_NO_COOP) != 0;
        const int rem = rows % 16;
        const bool split = rem != 0 && rem <= 12;
        const int mt16 = split ? rows / 16 : (rows + 15) / 16, m4 = split ? (rem + 3) / 4 : 0;
        const int nt_need = (ntab * npts + 15) / 16;
        if (!nocoop && npts <= 64 && e->d_coop_eint) {
            for (size_t i = 0; i < sizeof(kCoopShapes) / sizeof(kCoopShapes[0]); ++i) {
                const CoopShape& c = kCoopShapes[i];
                if (c.sd != e->sd || c.order != order || c.mt16 != mt16 || c.m4 != m4 || nt_need > 4 * c.tpw) continue;
                const bool piola = c.can_piola && (mapping == FX_MAP_COVARIANT_PIOLA || mapping == FX_MAP_CONTRAVARIANT_PIOLA) && verts && e->vdim == e->sd;
                if (c.piola_only && !piola) continue;
                if (piola && stacked_pio_ok) continue;
                const int table = rows * npts;
                // tables per output round: as many as LDS holds next to the A fragments and the chain
                // state while two workgroups still fit a CU (fewer rounds = fewer workgroup barriers per
                // request); when one workgroup fills the CU anyway, whatever is left of its LDS
                const long long other = ((long long)(mt16 + m4) * e->coop_KS * 64 + 4LL * 4 * ntab * 32) * 8;
                long long img_budget = ctx->lds_per_cu / 2 - other - 1024;
                if (img_budget < 16 * 1024) img_budget = std::min<long long>(ctx->lds_per_cu - other - 1024, 56 * 1024);
                int TR = std::max(1, (int)(img_budget / ((long long)table * 8)));
                TR = std::min(TR, ntab);
                if (piola) {  // the fused flush writes pairs: every round a whole number of them, <= 4 per thread
                    if ((table & 1) && (TR & 1)) TR = TR > 1 ? TR - 1 : 0;
                    while (TR > 0 && (long long)TR * table > 2 * 512 * fxk::PIOLA_SLOTS) TR -= (table & 1) ? 2 : 1;
                    if (TR <= 0 || (((long long)ntab * table) & 1)) continue;
                    if ((table & 1) && (ntab % TR) % 2) continue;  // the last round must be even too
                }
                long long img = std::max<long long>(2LL * (nt_need * 64 + 128), (long long)TR * table);
                img = (img + 1) & ~1LL;
                if (npts > 32) continue;  // LDS-resident chain state holds 32 points
                long long ldsb = ((long long)(mt16 + m4) * e->coop_KS * 64 + img + 4LL * 4 * ntab * 32) * 8;
                if (ldsb > ctx->lds_per_cu) continue;
                fxk::CoopArgs& ca = L.cargs;
                memset(&ca, 0, sizeof ca);
                ca.pts = pts;
                ca.verts = verts;
                ca.out = out;
                ca.afrag = e->d_afrag_coop;
                ca.eint = e->d_coop_eint;
                ca.edbl = e->d_coop_edbl;
                ca.kstart = e->d_coop_kstart;
                ca.phi0 = e->prog.phi0;
                memcpy(ca.A0, e->A0, sizeof ca.A0);
                memcpy(ca.b0, e->b0, sizeof ca.b0);
                ca.nreq = nreq;
                ca.npts = npts;
                ca.rows = rows;
                ca.KS = e->coop_KS;
                ca.NT = nt_need;
                ca.emax = e->coop_emax;
                ca.TR = TR;
                ca.slab_doubles = nt_need * 64 + 128;
                ca.img_doubles = (int)img;
                ca.debug = a.debug;
                ca.piola = piola ? mapping : 0;
                if (piola && !invert_small(e->sd, e->A0, ca.A0inv)) return fail(FX_EINVAL, "degenerate cell");
                L.fused_mapping = piola;
                L.clds_bytes = (int)ldsb;
                int per_cu = std::max(1, std::min(2, ctx->lds_per_cu / L.clds_bytes));
                L.cgrid = (int)std::max<long long>(1, std::min<long long>(nreq, (long long)ctx->num_cu * per_cu));
                L.coop_id = (int)i;
                break;
            }
        }
    }
    // ---- shape-specialised kernel available? ----
    L.fixed_id = -1;
    const bool nofixed = (ctx->policy & FX_POLICY_NO_FIXED) != 0;
    const bool stacked_small = (ctx->policy & FX_POLICY_STACKED_SMALL) != 0;  // A/B: the stacked kernel's small-shape instances instead
    const bool ab_small = stacked_small && !verts && mapping == FX_MAP_AFFINE;
    if (!nofixed && !ab_small && npts <= 64) {
        const int nt_need = (ntab * npts + 15) / 16;
        for (size_t i = 0; i < sizeof(kFixedShapes) / sizeof(kFixedShapes[0]); ++i) {
            const FixedShape& f = kFixedShapes[i];
            // (half-image shapes keep the tables of each half in their own tiles: one spare tile is fine)
            const bool nt_ok = f.nt == nt_need || (!f.fullimg && f.nt == nt_need + 1);
            if (f.sd != e->sd || f.n != e->n || f.order != order || f.rows != rows || !nt_ok) continue;
            {   // the image of a flush round leaves as 16-byte pieces: an even number of doubles per round (and so per request)
                const int th = f.fullimg ? ntab : (ntab + 1) / 2;
                if (((long long)th * rows * npts) % 2 || ((long long)(ntab - th) * rows * npts) % 2) continue;
            }
            const bool want_piola = mapping == FX_MAP_COVARIANT_PIOLA || mapping == FX_MAP_CONTRAVARIANT_PIOLA;
            const bool fuse_here = want_piola && f.can_piola && verts && e->vdim == e->sd;
            if (L.fused_mapping && !fuse_here) continue;  // the cooperative kernel fuses the map, this one cannot
            bool ok = false;
            if (e->sd == 3 && e->n == 2) ok = table_matches<3, 2>(e->prog);
            if (e->sd == 3 && e->n == 3) ok = table_matches<3, 3>(e->prog);
            if (e->sd == 3 && e->n == 4) ok = table_matches<3, 4>(e->prog);
            if (!ok) continue;
            fxk::FixedArgs<0>& fa = L.fhead;
            memset(&fa, 0, sizeof fa);
            L.fcoef.resize(e->prog.steps.size() * 3);
            L.fucoef.assign(e->prog.steps.size() * 12, 0.0);
            for (size_t k = 0; k < e->prog.steps.size(); ++k) {
                const fx::Step& st = e->prog.steps[k];
                L.fcoef[3 * k + 0] = st.A;
                L.fcoef[3 * k + 1] = st.B;
                L.fcoef[3 * k + 2] = st.C;
                // point-independent derivatives of the collapsed-coordinate factors on the
                // element's own cell (rows of A0 padded with zeros, expansions.py:43-63)
                const int sd = e->sd;
                double dfa[3] = {0, 0, 0}, dfb[3] = {0, 0, 0};
                for (int d = 0; d < sd; ++d) {
                    double dx = e->A0[st.codim * sd + d];
                    double dy = st.codim + 1 < sd ? e->A0[(st.codim + 1) * sd + d] : 0.0;
                    double dz = st.codim + 2 < sd ? e->A0[(st.codim + 2) * sd + d] : 0.0;
                    dfb[d] = 0.5 * (dy + dz);
                    dfa[d] = dx + dfb[d];
                }
                double* u = &L.fucoef[12 * k];
                for (int d = 0; d < sd; ++d) {
                    u[d] = st.A * dfa[d] - st.B * dfb[d];
                    u[3 + d] = -2.0 * st.C * dfb[d];
                }
                int h = 0;
                for (int d1 = 0; d1 < sd; ++d1)
                    for (int d2 = d1; d2 < sd; ++d2) u[6 + h++] = -2.0 * st.C * dfb[d1] * dfb[d2];
            }
            fa.pts = pts;
            fa.verts = verts;
            fa.out = out;
            const char* kk = (ctx->policy & FX_POLICY_KERNEL_IMAGE) ? "image" : (ctx->policy & FX_POLICY_KERNEL_STREAM) ? "stream" : nullptr;
            // default: K-streamed kernel, two requests per wave when the points of two requests fit
            // one wave (A/B: FX_POLICY_KERNEL_IMAGE | _STREAM)
            // (pair kernel: the tables of each output half must fit half of the column tiles)
            const bool pair_ok = npts <= 32 * (3 - f.rpw) && (f.fullimg || ntab == 1 ||
                                                              (f.nt % 2 == 0 && (f.nt / 2) * 16 >= ((ntab + 1) / 2) * npts));
            L.fkind = pair_ok ? 2 : 1;
            if (kk && !strcmp(kk, "image") && !f.pair_only) L.fkind = 0;
            if (kk && !strcmp(kk, "stream") && !f.pair_only) L.fkind = 1;
            if (f.pair_only && L.fkind != 2) continue;
            if (kk && !strcmp(kk, "pair") && pair_ok) L.fkind = 2;
            L.ncu = ctx->num_cu;
            L.trash = ctx->d_trash;
            // the kernel leaves its counter zeroed; launches in flight at the same time (other
            // streams) get different counters
            L.queue = ctx->d_queue + (size_t)(ctx->launch_seq++ % FX_QUEUE_SLOTS) * 16;
            fa.afrag = L.fkind >= 1 ? e->d_afrag_stream : e->d_afrag_split;
            fa.phi0 = e->prog.phi0;
            memcpy(fa.A0, e->A0, sizeof fa.A0);
            memcpy(fa.b0, e->b0, sizeof fa.b0);
            fa.nreq = nreq;
            fa.npts = npts;
            fa.debug = (a.debug & 0xffff) | (fuse_here ? (mapping << 16) : 0);
            long long need = std::max<long long>((long long)f.nt * e->KS * 64, (long long)ntab * rows * npts);
            need = (need + 1) & ~1LL;
            if (L.fkind >= 1) {
                {
                    // per wave: half image (>= the K-step slab that aliases it) + 64-double dump row
                    const int th = (L.fkind == 2 && f.fullimg) ? ntab : (ntab + 1) / 2;
                    long long per_wave =
                        std::max<long long>((long long)th * rows * npts, (long long)f.nt * 64 * (L.fkind == 2 ? f.rpw : 1)) + 64;
                    per_wave = (per_wave + 1) & ~1LL;
                    fa.lds_doubles = (int)per_wave;
                    int rem = rows % 16;
                    bool split = rem != 0 && rem <= 12;
                    int nfrag = ((split ? rows / 16 : (rows + 15) / 16) + (split ? (rem + 3) / 4 : 0)) * e->KS;
                    L.flds_bytes = (int)((nfrag * 64 + per_wave * FIXED_NW) * 8);
                }
                // registers bound the occupancy: ask for every workgroup the CU can hold
                long long want = (long long)ctx->num_cu * 4;
                const long long units = L.fkind == 2 ? (nreq + f.rpw - 1) / f.rpw : nreq;  // requests or pairs, one per wave
                long long nwg = (units + FIXED_NW - 1) / FIXED_NW;
                L.fgrid = (int)std::max<long long>(1, std::min<long long>(nwg, L.fkind >= 2 ? nwg : want));
                if (L.flds_bytes > ctx->lds_per_cu) continue;
                if (fuse_here && L.fkind != 2) continue;
                if (fuse_here) L.fused_mapping = true;
                L.fixed_id = (int)i;
                break;
            }
            if (fuse_here) continue;
            // 8 waves per CU (2 workgroups of 4 waves): pad the request to 1/2 of the CU's LDS
            long long per_wave = std::max<long long>(need * 8, (long long)(ctx->lds_per_cu / 8));
            per_wave &= ~15LL;
            if (per_wave < need * 8) per_wave = need * 8;
            fa.lds_doubles = (int)(per_wave / 8);
            L.flds_bytes = (int)(per_wave * FIXED_NW);
            if (L.flds_bytes > ctx->lds_per_cu) continue;
            int wg_per_cu = std::max(1, ctx->lds_per_cu / L.flds_bytes);
            long long want = (long long)ctx->num_cu * wg_per_cu;
            long long nwg = (nreq + FIXED_NW - 1) / FIXED_NW;
            L.fgrid = (int)std::max<long long>(1, std::min<long long>(nwg, want));
            L.fixed_id = (int)i;
            break;
        }
    }
    // ---- low-order lane-local kernel? ----
    L.small_id = -1;
    {
        const bool nosmall = (ctx->policy & FX_POLICY_NO_SMALL) != 0;
        const long long reqbytes8 = (long long)ntab * rows * npts * 8;
        if (!nosmall && order <= 2 && npts <= 64 && rows <= 96 && reqbytes8 <= 16 * 1024 && e->d_cmat) {
            for (size_t i = 0; i < sizeof(kSmallShapes) / sizeof(kSmallShapes[0]); ++i) {
                if (kSmallShapes[i].sd != e->sd || kSmallShapes[i].n != e->n) continue;
                const bool values_only_shape = i >= 8;   // (the round-4 entries)
                if (values_only_shape && (order != 0 || mapping != FX_MAP_AFFINE || e->vdim != 1 || (ctx->policy & (FX_POLICY_WG_SMALL | FX_POLICY_NO_SMALL_VALUES)))) continue;
                if (values_only_shape && e->sd == 3 && npts > 16) continue;   // (P3 tetrahedra: up to the 14-point rule)
                if (!small_table_matches((int)i, e->prog) || (int)e->prog.steps.size() > fxk::SMALL_MAXSTEPS) continue;
                int P = std::max(1, 64 / npts);
                // (17-24 rows -- vector-valued degree-2 triangles -- with Hessians: 6.9 KB a request left ONE request a wave, six of 64 lanes
                // at the 6-point rule, and the shape went to the generic kernel at 31 % of the HBM peak; up to 28 KB a wave there)
                const long long image_cap = ((rows > 16 && e->sd == 2) || rows > 24) ? 28 * 1024 : 12 * 1024;   // (N1 tetrahedra, 18 rows, with Hessians: 199 us at two requests a wave, 213 at four)
                while (P > 1 && P * reqbytes8 > image_cap) --P;  // per-wave image: several workgroups per CU
                // many rows (vector-valued elements): the MFMA contraction of the generic / stacked kernels wins over
                // rows x members FMAs per lane (tools/coverage_map.py: 30+ rows 15-30 % here against 26-48 % there; 17-24 rows
                // only while several requests share a wave)
                // ... unless the Piola map of a vector-valued element on per-request cells fuses here (one pass instead of
                // tabulation + read-modify-write of every table: tools/coverage_map.py --verts --pushforward)
                const bool fuse_small = (mapping == FX_MAP_COVARIANT_PIOLA || mapping == FX_MAP_CONTRAVARIANT_PIOLA) && verts &&
                                        e->vdim == e->sd && rows % e->sd == 0 && L.fixed_id < 0 && L.coop_id < 0 && !L.fused_mapping &&
                                        !(e->sd == 3 && e->n == 2 && order == 2);  // (that instance spills 39 registers)
                if (fuse_small && rows > 24 && order >= 1 && stacked_pio_ok) break;
                // (36 rows of degree-1 tetrahedra -- BDM1, N2 of degree 1 -- stayed generic at 42-46 %: four members, 144 FMAs a table.
                // Second half of round 4, tools/instance_ab.py, 0.8 GB, generic -> lane-local: BDM2 / N2 triangles with Hessians at 6 / 12
                // points 306 / 251 -> 188 / 188 us (with cells 319 / 260 -> 188 / 188), BDM1 tetrahedra values / gradients at 4 points
                // 159 / 216 -> 141 / 169, at 11 points 231 -> 177; with Hessians the ten tables of 36 rows are slower here, 209 -> 350: not taken)
                const int row_cap = (e->sd == 3 && e->n == 1 && order <= 1) ? 36 : 24;
                if (fuse_small ? rows > 36 : !values_only_shape && (rows > row_cap || (rows > 16 && P == 1))) break;
                fxk::SmallArgs& sa = L.sargs;
                memset(&sa, 0, sizeof sa);
                if (fuse_small) {
                    sa.piola = mapping;
                    for (int q = 0; q < 9; ++q) sa.G[q] = 0.5 * e->A0[q];
                    L.fused_mapping = true;
                }
                sa.pts = pts;
                sa.verts = verts;
                sa.out = out;
                sa.cmat = e->d_cmat;
                for (size_t k = 0; k < e->prog.steps.size(); ++k) {
                    sa.coef[3 * k + 0] = e->prog.steps[k].A;
                    sa.coef[3 * k + 1] = e->prog.steps[k].B;
                    sa.coef[3 * k + 2] = e->prog.steps[k].C;
                }
                sa.phi0 = e->prog.phi0;
                memcpy(sa.A0, e->A0, sizeof sa.A0);
                memcpy(sa.b0, e->b0, sizeof sa.b0);
                sa.nreq = nreq;
                sa.nitems = (nreq + P - 1) / P;
                sa.npts = npts;
                sa.rows = rows;
                sa.P = P;
                sa.stage_doubles = (int)(((long long)P * ntab * rows * npts + 1) & ~1LL);
                sa.debug = a.debug;
                L.slds_bytes = sa.stage_doubles * 8 * SMALL_NW;
                const int wg_per_cu = std::max(1, std::min(8, ctx->lds_per_cu / std::max(1, L.slds_bytes)));
                const long long nwg = (sa.nitems + SMALL_NW - 1) / SMALL_NW;
                L.sgrid = (int)std::max<long long>(1, std::min<long long>(nwg, (long long)ctx->num_cu * wg_per_cu * 4));
                L.small_id = (int)i;
                break;
            }
        }
    }
    // ---- stacked-matrix kernel (requests on the element's own cell, large shapes)? ----
    L.stacked_id = -1;
    {
        const bool nostacked = (ctx->policy & FX_POLICY_NO_STACKED) != 0;
        // (fewer stacked rows: the production of the B fragments is no longer amortised over enough row tiles)
        // (measured, tools/coverage_map.py: values-only requests of P3 / P4 tetrahedra, 20 / 35 rows, run at 45 / 30 % of the HBM
        // peak here against 23 / 13 % on the generic kernel; below one row tile nothing is left to amortise)
        // (15 rows: values-only P4 triangles 27-30 % lane-local -> 38-49 % here; 10 rows -- P3 triangles, P2 tetrahedra -- stay
        // lane-local, 39-43 % against 34-39 %)
        // (round 3, tools/coverage_map.py --audit: since the lane-local kernel packs several requests per lane group it serves
        // the 15-row tables faster wherever it holds them -- 6 / 16 / 25 / 32 points: 165 / 157 / 211 / 157 us against
        // 249 / 191 / 412 / 282 us here -- so below one full row tile this kernel only takes what the lane-local one refuses)
        static const long long stacked_min_rows = ab_env("FIAT_AMD_STACKED_MIN_ROWS") ? atoll(ab_env("FIAT_AMD_STACKED_MIN_ROWS")) : 15;
        const long long R = (long long)ntab * rows;
        const int RT = (int)((R + 15) / 16);
        const bool even = ((R * npts) % 2 == 0) && (((R - 16LL * (RT - 1)) * npts) % 2 == 0);
        // (after the shape-specialised registries: the tuned paired instances and the lane-local kernel keep their shapes)
        // (per-request cells: the kernel maps the points through the request's cell and a second pass applies the
        // chain rule across the derivative tables, table_mix_kernel; a Piola map is left to fx_pushforward_batch)
        // (the lane-local kernel keeps the shapes no stacked instance takes -- degrees 1 and 2 on triangles, degree 1 on
        // tetrahedra, fewer than 16 stacked rows -- and per-request cells, where it applies the chain rule itself; elsewhere
        // the MFMA contraction wins: tools/coverage_map.py, P3 / P4 triangles and P2 tetrahedra with derivatives 26-52 % -> 50-69 %)
        // (order 2 with per-request cells: the rtc -3 instances beat the lane-local kernel on P4 triangles and P2 tetrahedra,
        // 21 -> 58 % and 19 -> 38 % of the HBM peak, tools/coverage_map.py --verts --order 2 [--policy no_small])
        const bool small_keeps = L.small_id >= 0 && verts;
        // (round 3 audit, ABAB with tools/instance_ab.py [--policy no_small]: gradients of P4 triangles 264 / 213 / 260 / 293 / 218 us
        // lane-local at 12 / 16 / 24 / 25 / 32 points against 193 / 164 / 162 / 259 / 164 us on the rtc -2 instances; Hessians of P3
        // triangles at 15 / 16 points 280 / 259 against 216 / 202 us -- level or behind at the other sizes; P3 triangles and P2
        // tetrahedra with gradients stay lane-local, 151-220 against 177-262 us)
        const bool over_small = (order == 2 && ((e->sd == 2 && e->n >= 4) || (e->sd == 3 && e->n >= 2) ||
                                                (e->sd == 2 && e->n == 3 && npts >= 13 && npts <= 16))) ||
                                (order == 1 && e->sd == 2 && e->n == 4);
        // (round 3, tools/coverage_map.py --audit + tools/instance_ab.py --own-cell, ABAB at 1.5 GB of tables: on the element's own
        // cell the stacked kernel now beats the paired instances of P3 tetrahedra with gradients outside the 21..24-point
        // benchmark shape -- 12 / 14 / 28 / 32 / 44 points: 294 / 296 / 304 / 282 / 284 us paired, 270 / 256 / 259 / 249 / 257 us
        // here -- and of P4 tetrahedra, 400 / 373 us -> 316 / 303 us at 22 / 24 points; with per-request cells the paired kernel
        // applies the chain rule on its accumulators and keeps every shape, 285-345 us against 365-750 us.  Those entries yield
        // when an instance here holds the request.)
        const bool fixed_yields = L.fixed_id >= 0 && !verts && !L.fused_mapping && order == 1 && e->sd == 3 &&
                                  ((e->n == 3 && kFixedShapes[L.fixed_id].nt != 6) || e->n == 4);
        if (!nostacked && (L.fixed_id < 0 || fixed_yields) && (!small_keeps || over_small) && !L.fused_mapping && !e->raw_expansion && order <= 2) {
            // among the instances of one kind that hold the request, the one with the fewest padding columns
            auto tighter_instance = [&](const StackedShape& k) {
                for (const StackedShape& o : kStackedShapes)
                    if (o.sd == k.sd && o.n == k.n && o.rtc == k.rtc && 16 * o.ct / o.g >= npts &&
                        (16 * o.ct / o.g < 16 * k.ct / k.g || (16 * o.ct / o.g == 16 * k.ct / k.g && o.ct < k.ct)))
                        return true;
                return false;
            };
            // ... and only when the group's points fill more than two thirds of its column tiles (measured: at 16 of 48
            // columns the in-kernel chain-rule instance runs at 18 % where the generic kernel reaches 43 %)
            auto fills_tiles = [&](const StackedShape& k) { return 3LL * k.g * npts > 2LL * 16 * k.ct; };
            // (first the instances that apply the chain rule themselves, then the rest, each in table order)
            // the request-per-workgroup kernel applies the order-1 chain rule of per-request cells itself (round 4, MIX instances):
            // rules of 49..128 points, the window of P5 triangles, or policy wg_small -- the per-wave chain-rule instances yield
            auto wg_mix_takes = [&](const StackedShape& k) {
                if ((ctx->policy & (FX_POLICY_NO_WG | FX_POLICY_NO_STACKED_MIX)) || !verts || order != 1 || npts > 128) return false;
                // (49..64 points, two requests per slab: tools/instance_ab.py [--policy wg_small], sustained, 0.8 GB -- N3 / RT3 tetrahedra at
                // 57 / 50 points 315 / 316 -> 199 / 190 us, P3 / P4 / P5 / P6 at 57 points 384 / 412 / 464 / 544 -> 279 / 328 / 333 / 434, P6
                // triangles at 49 / 60 points 427 / 394 -> 302 / 253, P5 at 55 points 515 -> 313: the former routes were a second
                // 48-point chunk that is mostly padding, or a four-tile whole-request instance plus the table-mixing pass)
                // (... and two windows the off-default audits found, tools/coverage_map.py --audit --verts --qdeg-offset -1, confirmed in
                // sustained runs: vector-valued degree-3 tetrahedra at 13..15 points -- N3 / RT3 / BDM3 at the 14-point rule 407 / 352 / 374 ->
                // 229 / 174 / 181 us, nine requests per slab against the three-request instance + table-mixing pass -- and tables of an odd
                // number of doubles at 17..48 points, which only the point chunks or a whole-request instance + mixing pass held: P5
                // triangles at 19 points 589 -> 307 us, P4 tetrahedra at 31 points 401 -> 314)
                // (where the per-wave kernel has an 8-byte twin of its chain-rule instance it keeps the odd tables: N3 tetrahedra at the
                // 23-point rule 198 us there against 213 here)
                const bool odd_pt = ((long long)rows * npts) % 2 != 0;
                const bool twin2 = (k.sd == 3 && (k.n == 3 || k.n == 4) && npts <= 24) || (k.sd == 2 && k.n == 5 && npts >= 25);
                const bool window = npts > 48 || (k.sd == 2 && k.n == 5 && npts >= 25 && npts <= 33) ||
                                    (k.sd == 3 && k.n == 3 && e->vdim > 1 && npts >= 13 && npts <= 15) ||
                                    // (degree >= 4 tetrahedra at 13..15 points, nine requests per slab: P4 / P5 / P6 at 14 points 403 / 485 / 580 ->
                                    // 272 / 292 / 393 us)
                                    (k.sd == 3 && k.n >= 4 && npts >= 13 && npts <= 15) ||
                                    (odd_pt && npts >= 17 && npts <= 48 && !twin2);
                if (!(window || (ctx->policy & FX_POLICY_WG_SMALL))) return false;
                const int g = npts > 64 ? 1 : std::min(12, 128 / npts);
                const int ctn = std::max(4, (g * npts + 15) / 16);
                const int ctm = fxwg::mix_ct(k.sd, k.n, ctn);
                if (ctm == 0) return false;
                // (two waves per row tile work on two dof tiles at a time: with few rows the second pair multiplies padding -- P3 / P4
                // tetrahedra at 70 / 74 points 424 / 406 us against 388 / 403 us on the point chunks -- so at least 70 % of those
                // MFMA rows must be dofs; the eight-tile instances of tetrahedra, and of triangles with odd tables, run all four waves on one dof tile)
                const bool oddm0 = ((long long)rows * npts) % 2 || ((long long)(rows - 16 * ((rows + 15) / 16 - 1)) * npts) % 2;
                if (!(ctm == 8 && (k.sd == 3 || oddm0))) {   // (wg.hip wg_mix_pc: those instances run four waves per dof tile)
                    const int rtd = (rows + 15) / 16;
                    if (10LL * rows < 7LL * 16 * 2 * ((rtd + 1) / 2)) return false;
                }
                const bool oddm = ((long long)rows * npts) % 2 || ((long long)(rows - 16 * ((rows + 15) / 16 - 1)) * npts) % 2;
                return fxwg::has_mix_instance(k.sd, k.n, ctm, oddm) && fxwg::lds_bytes(k.sd, k.n, ctm) <= ctx->lds_per_cu;
            };
            // (then the request-per-workgroup instances, then the rest)
            constexpr size_t NSH = sizeof(kStackedShapes) / sizeof(kStackedShapes[0]);
            for (size_t i2 = 0; i2 < 3 * NSH && L.stacked_id < 0; ++i2) {
                const size_t i = i2 % NSH, pass = i2 / NSH;
                const StackedShape& k = kStackedShapes[i];
                bool mix_odd = false, pio = false;
                int wg_g = 1, wg_ctw = 0;
                bool wg_mix1 = false, wg_odd = false;
                const bool inmix = k.rtc == -2 || k.rtc == -3 || k.rtc == -4 || k.rtc == -5 || k.rtc == -6;  // chain rule / Piola map inside the kernel
                const bool chunked = k.rtc == -1 || k.rtc == -4 || k.rtc == -5;
                const bool wgk = k.rtc == -7;
                if (pass != (inmix ? 0u : wgk ? 1u : 2u)) continue;
                if (small_keeps && k.rtc != -3 && k.rtc != -2) continue;
                if (k.sd != e->sd || k.n != e->n) continue;
                // small shapes with register-resident fragments: A/B partner of the paired kernel only (measured
                // equal on C2, 302 vs 303 us -- both sit on the store-path plateau -- and the paired kernel's
                // recurrence derivatives are two digits more accurate), opt-in with FIAT_AMD_STACKED_SMALL=1
                if (k.rtc > 0 ? (RT != k.rtc || verts || !stacked_small) : (R < stacked_min_rows || (R < 16 && L.small_id >= 0))) continue;
                // (per-request cells with derivatives on the low-degree shapes: the second pass over the tables costs more
                // than the generic kernel's in-kernel chain rule -- tools/small_vs_stacked.py --verts, 18-29 % against 25-48 %)
                if (!inmix && verts && order >= 1 && (k.n <= 2 || (k.sd == 2 && k.n <= 4))) continue;
                // (the same for vector-valued degree-3 tetrahedra at up to 16 points, round 3 audit: N3 / RT3 / BDM3 with Hessians at
                // 11 / 16 points 525 / 450, 447 / 425, 492 / 461 us on the three-request instance plus mixing pass against 404 / 312,
                // 375 / 288, 426 / 312 us generic; N3 with gradients 392 / 324 against 338 / 221, but 379 against 398-444 at 14 points: kept there)
                if (!inmix && verts && k.sd == 3 && k.n == 3 && e->vdim > 1 && (order == 2 ? npts <= 16 : order == 1 && (npts <= 12 || npts == 16))) continue;
                // (own cell, gradients of P3 / P4 triangles where the lane-local kernel holds the request -- round 3, ABAB with
                // tools/instance_ab.py --own-cell [--policy no_stacked]: P3 at 6 / 7 / 10 / 15 / 25 points 195 / 197 / 172 / 170 / 199 us
                // here against 149 / 149 / 144 / 157 / 149 us lane-local, while 8 / 12 / 16 / 24 points are level or better here;
                // P4 at odd sizes up to 16 points falls to the point-chunked instance, 407 us against 264 us at 15 points)
                if (!verts && order == 1 && L.small_id >= 0 && k.sd == 2 &&
                    ((k.n == 3 && npts % 8 != 0 && npts != 12) || (k.n == 4 && k.rtc == -1 && npts <= 16)))
                    continue;
                // (values-only P5 triangles where the lane-local kernel holds several requests per wave: round 4)
                if (order == 0 && L.small_id >= 8) continue;
                if (k.rtc == -2 || k.rtc == -3 || k.rtc == -6) {  // per-request cells, order 1 / 2: chain rule across the tables inside the kernel (dof-major tiles)
                    const bool nomix = (ctx->policy & FX_POLICY_NO_STACKED_MIX) != 0;
                    if (nomix || !verts || order != (k.rtc == -2 ? 1 : k.rtc == -3 ? 2 : 0)) continue;
                    if (k.rtc == -2 && wg_mix_takes(k)) continue;
                    // the element's Piola map too (vector-valued elements: the twin with the components of a dof in one lane,
                    // 12 rows per tile in 3-D); values only: nothing else to mix, so only with the map
                    const bool odd_pio_instance = k.sd == 3 && k.ct == 3 && ((k.n == 2 && k.g == 4) || (k.n == 3 && k.g == 2));
                    pio = want_piola && stacked_has_pio(k.sd, k.n) && (pio_even || (pio_odd_twin && odd_pio_instance));
                    if (k.rtc == -6 && !pio) continue;
                    // (16-byte pieces of whole tables; three shapes have an 8-byte twin for odd table sizes: P5 triangles at the
                    // 25-point rule, N3 and RT2 tetrahedra at the 23- and 11-point rules)
                    mix_odd = ((long long)rows * npts) % 2 || ((long long)(rows - 16 * ((rows + 15) / 16 - 1)) * npts) % 2;
                    // (RT2 at order 1 stays on the generic kernel: 44.7 % of the HBM peak there, 39.2 % on the twin)
                    if (pio) mix_odd = !pio_even;
                    if (mix_odd && !pio && !((k.sd == 2 && k.n == 5 && k.ct == 2 && k.g == 1) || (k.sd == 3 && k.n == 3 && k.ct == 3 && k.g == 2) ||
                                             (k.rtc == -3 && k.sd == 3 && k.n == 2 && k.ct == 3 && k.g == 4) ||
                                             // off-default rules of odd size: P4 tetrahedra at 17-24 points (the 23-point rule), P4 triangles at
                                             // 25-32, P5 triangles at 33-48 (the 33-point rule)
                                             (k.sd == 3 && k.n == 4 && k.ct == 3 && k.g == 2) || (k.sd == 2 && k.n == 4 && k.ct == 2 && k.g == 1) ||
                                             (k.sd == 2 && k.n == 5 && k.ct == 3 && k.g == 1)))
                        continue;
                    if (npts > 16 * k.ct / k.g || tighter_instance(k) || !fills_tiles(k)) continue;
                } else if (wgk) {  // a request (or a group of small requests) per workgroup
                    if ((ctx->policy & FX_POLICY_NO_WG) || npts > 16 * k.ct) continue;
                    // (fewer than 49 points: the whole-request instances of the per-wave kernel -- except where several requests per
                    // workgroup are robustly ahead in sustained runs, tools/instance_ab.py --own-cell [--policy wg_small], 0.8 GB:
                    // P5 triangles with derivatives at 25..33 points 232 / 183 / 252 -> 189 / 169 / 225 us, degree-6 tetrahedra with
                    // Hessians at 33..48 points 432 -> 352 us.  Elsewhere the two are within +-10 % of each other with either sign,
                    // and the map's short interleaved runs disagree with the sustained ones: opt-in, policy wg_small)
                    // (vector-valued degree-3 tetrahedra of 180 rows a table -- BDM3, N2 of degree 3 -- at 17..24 points with cells, values:
                    // 892 -> 744 us at the 23-point rule in 4 GB batches.  What the 0.8 GB runs of the audits showed beside it -- their
                    // other orders, 13..15 points on the own cell, 49..64 points with derivatives -- is within +-4 % at 4 GB: not taken)
                    // (... and on the own cell: 904 -> 742 us; gradients / Hessians there 1.07 / 1.01, RT3 / N3 values 0.91 / 1.10: not taken)
                    const bool vec3_rows = k.sd == 3 && k.n == 3 && e->vdim > 1 && rows >= 160;
                    // (after the kernel's second half of round 4, sustained default / wg_small sweep over 135 shapes, own cell and cells:
                    // degree >= 5 tetrahedra at 13..15 points, nine requests per slab -- P5 / P6 at 14 points, values / gradients / Hessians
                    // 352 / 272 / 302 -> 316 / 237 / 236 us and 465 / 371 / 439 -> 401 / 328 / 314, with cells 368 / - / 571 -> 351 / - / 472 and
                    // 477 / - / 697 -> 422 / - / 549; P4 with Hessians 208 -> 187, with cells 477 -> 438; values of degree >= 5 at 25..32
                    // points, four per slab, 363 / 466 -> 319 / 410.  Confirmed in 4 GB batches (CAP_GB=4): 0.91-0.97 of the per-wave
                    // instances there, 0.89 / 0.91 at 25..32 points -- while P6 with Hessians at the 23-point rule, 438 -> 366 us at 0.8 GB,
                    // is 1.04 at 4 GB and 7.51 against 6.93 ms in bench.py --workload dg6tet: heavy requests are few at 0.8 GB and the
                    // two kernels quantise differently over the chip; windows are taken from the larger batches)
                    const bool hi14 = k.sd == 3 && npts >= 13 && npts <= 15 && (k.n >= 5 || (k.n == 4 && order == 2));
                    const bool hi_more = k.sd == 3 && k.n >= 5 && order == 0 && npts >= 25 && npts <= 32;
                    const bool small_window = (k.sd == 2 && k.n == 5 && order >= 1 && npts >= 25 && npts <= 33) ||
                                              (k.sd == 3 && k.n == 6 && order == 2 && npts >= 33 && npts <= 48) || hi14 || (hi_more && !verts);
                    // per-request cells with gradients: the chain rule on the accumulators (MIX instances); other orders with
                    // cells: values straight from the kernel, derivatives + the table-mixing pass
                    const bool nomix = (ctx->policy & FX_POLICY_NO_STACKED_MIX) != 0;
                    const bool want_mix = verts && order == 1 && !nomix;
                    // (tables of an odd number of doubles at 17..48 points had only the point chunks: P5 triangles with gradients at 19
                    // points 361 -> 201 us, values of P4 tetrahedra at 31 points 515 -> 394; with cells 586 -> 459)
                    // (not where a whole-request instance has an 8-byte twin: values of P4 tetrahedra at the 23-point rule 315 us there,
                    // 422 here)
                    const bool twin1 = (k.sd == 3 && (k.n == 3 || k.n == 4) && npts <= 24) || (k.sd == 2 && k.n == 5 && npts >= 25 && npts <= 32);
                    const bool odd_window = !even && npts >= 17 && npts <= 48 && (!verts || order == 0) && !twin1;
                    if (npts <= 48 && !(ctx->policy & FX_POLICY_WG_SMALL) && !(small_window && (!verts || want_mix)) && !odd_window &&
                        !(order == 0 && vec3_rows && npts >= 17 && npts <= 24) &&
                        !(verts && order != 1 && (hi14 || hi_more)) &&
                        !(want_mix && wg_mix_takes(k)))
                        continue;
                    // requests per slab of <= 128 columns and the instance's column tiles
                    wg_g = npts > 64 ? 1 : std::min(12, 128 / npts);
                    wg_ctw = std::max(4, (wg_g * npts + 15) / 16);
                    if (want_mix) {
                        // (tetrahedra: 4 column tiles with two waves per row tile or 8 with four; triangles 4 / 6 / 8 with two)
                        wg_ctw = fxwg::mix_ct(k.sd, k.n, wg_ctw);
                        wg_odd = ((long long)rows * npts) % 2 || ((long long)(rows - 16 * ((rows + 15) / 16 - 1)) * npts) % 2;
                        wg_mix1 = wg_ctw > 0 && fxwg::has_mix_instance(k.sd, k.n, wg_ctw, wg_odd) && wg_mix_takes(k);
                    }
                    if (!wg_mix1) {
                        if (verts && order >= 1 && npts <= 48 && !(ctx->policy & FX_POLICY_WG_SMALL) && !(hi14 && order == 2)) continue;
                        // (65..96 points -- five or six column tiles -- with a short K loop: a stage is a few dozen MFMAs and the
                        // per-request recurrence + barriers dominate.  tools/instance_ab.py --own-cell [--policy no_wg], 0.8 GB, this
                        // kernel against the point chunks: P4 tetrahedra at 70 points, values / gradients 779 / 435 against 455 / 270 us,
                        // P3 528 / 232 against 345 / 217, P5 triangles at 79 points 556 / 358 against 334 / 234, P6 triangles values
                        // 455 against 320 -- while degree >= 5 tetrahedra (K steps >= 14) and P6 triangles with Hessians win, as does
                        // everything at 97..128 points.  Policy wg_small opts in regardless.)
                        wg_odd = !even;
                        wg_ctw = std::max(4, (wg_g * npts + 15) / 16);
                        if (wg_ctw == 7) wg_ctw = 8;        // (no seven-tile instance)
                        if (wg_ctw == 5) {                   // one wave per row tile on 5 column tiles, or two on 3 + 3: the fewer MFMA slots per wave
                            const long long one = (long long)((RT + 3) / 4) * 5, two = (long long)((RT + 1) / 2) * 3;
                            if (two < one) wg_ctw = 6;
                        }
                        // (... except the instances of one wave per row tile on five column tiles, since the recurrence coefficients are
                        // fetched ahead and the last tile of a group leaves under the next group's MFMAs (second half of round 4;
                        // sustained, same tool): values of P4 tetrahedra at 74 / 75 points 358 / 394 us against 432 / 429 on the point
                        // chunks, RT3 values / P3 Hessians at 74 points 175 / 164 against 202 / 197, P5 triangles with gradients at 65 / 72 /
                        // 79 points 240 / 187 / 224 against 261 / 240 / 235 -- while two waves per row tile on six column tiles still lose
                        // with few rows: P6 triangles, values at 79 points 359 against 296, P3 tetrahedra at 70 points 460 against 348)
                        if (npts > 64 && (wg_g * npts + 15) / 16 <= 6 && !(ctx->policy & FX_POLICY_WG_SMALL) &&
                            !((e->nexp + 3) / 4 >= 14 || (k.sd == 2 && k.n == 6 && order == 2) || wg_ctw == 5))
                            continue;
                        if (!fxwg::has_instance(k.sd, k.n, wg_ctw, wg_odd)) continue;
                    }
                    // (49..64 points: where a whole-request instance of four column tiles exists it keeps the rule)
                    // (... except degree >= 5 tetrahedra: two requests per slab 0.87-0.93 of the four-tile instance at 57 points)
                    if (!wg_mix1 && npts > 48 && npts <= 64 && even && !(ctx->policy & FX_POLICY_WG_SMALL) && !(k.sd == 3 && k.n >= 5)) {
                        bool whole = false;
                        for (const StackedShape& o : kStackedShapes)
                            whole = whole || (o.sd == k.sd && o.n == k.n && o.rtc == 0 && o.g == 1 && o.ct == 4);
                        if (whole) continue;
                    }
                    if (nreq + 3LL * ctx->num_cu > 0x7fffffffLL) continue;
                } else if (chunked) {  // point-chunked: whatever the whole-request instances above did not take
                    if (npts < 13 || nreq * (long long)((npts + 16 * k.ct - 1) / (16 * k.ct)) > 0x7fffffffLL) continue;
                    if ((k.rtc == -1 || k.rtc == -4) && k.sd == 3 && (k.n == 6 || k.n == 5)) {
                        // two-tile or three-tile chunks: the fewer column tiles.  Measured (tools/instance_ab.py --own-cell, 0.8 GB): degree 6
                        // at the 122-point rule, values / gradients / Hessians 581 / 469 / 540 us on 48-point chunks (9 tiles) -> 535 / 364 /
                        // 353 us on 32-point chunks (8 tiles); at 57 points (6 -> 4 tiles) gradients 426 us.  On a tie (74 points: 6 tiles
                        // either way) degree 6 is faster on 48-point chunks (402 against 448 us) and degree 5 within the spread of the two
                        // measurement protocols (361-370 us in sustained runs, 410-428 in the map's short ones, against 385-400): ties stay
                        // on 48-point chunks
                        const long long t2 = 2LL * ((npts + 31) / 32), t3 = 3LL * ((npts + 47) / 48);
                        if ((k.ct == 2) != (t2 < t3)) continue;
                    }
                    if (inmix) {  // ... with the chain rule inside: rules of more than one chunk
                        const bool nomix = (ctx->policy & FX_POLICY_NO_STACKED_MIX) != 0;
                        if (nomix || !verts || order != (k.rtc == -4 ? 1 : 2) || npts <= 16 * k.ct) continue;
                        if (k.rtc == -4 && wg_mix_takes(k)) continue;
                        // (49..64 points: the second chunk of 48 is mostly padding.  Round 3 audit, ABAB tools/instance_ab.py
                        // [--policy no_stacked_mix]: where a whole-request instance of four column tiles exists and the stacked
                        // matrix is small, that instance plus the mixing pass is faster -- gradients of P3 / P4 / P5 tetrahedra at
                        // 50 points 532 / 557 / 614 -> 392 / 428 / 516 us, P5 / P6 triangles 567 / 479 -> 421 / 423 us, Hessians of P3
                        // tetrahedra and P5 triangles 512 / 513 -> 411 / 424 us; from ~250 stacked rows on the pass over the tables
                        // costs more than the padding: N3 tetrahedra 318 against 395 us, Hessians of P4 / P5 tetrahedra 397 / 446
                        // against 467 / 539 us, and of P6 triangles at 168 rows 397 against 426 us)
                        if (npts <= 64 && even && R <= 224 && !(order == 2 && k.sd == 2 && k.n == 6)) {
                            bool whole = false;
                            for (const StackedShape& o : kStackedShapes)
                                whole = whole || (o.sd == k.sd && o.n == k.n && o.rtc == 0 && o.g == 1 && o.ct == 4);
                            if (whole) continue;
                        }
                    }
                } else {
                    // (16-byte stores of whole request chunks; a few instances have an 8-byte twin for odd request sizes)
                    const bool odd_twin = (k.sd == 3 && k.n == 4 && k.ct == 3 && k.g == 2) || (k.sd == 3 && k.n == 2 && k.ct == 3 && (k.g == 2 || k.g == 4)) ||
                                          (k.sd == 3 && k.n == 3 && k.ct == 3 && k.g == 2) || (k.sd == 2 && k.n == 5 && k.ct == 2 && k.g == 1);
                    // (with per-request cells the 8-byte twin works as it is: the cells only enter the production phase, and the
                    // chain rule across the tables is the separate mixing pass -- P5 triangles at the 25-point rule with cells ran on
                    // the point-chunked instance at 17.6 % against 27-37 % on their own cell)
                    if (!even && !odd_twin) continue;
                    if (npts > 16 * k.ct / k.g || tighter_instance(k) || !fills_tiles(k)) continue;  // 16 ct / g: points one request may have
                }
                bool ok = false;
                if (e->sd == 3 && e->n == 6) ok = table_matches<3, 6>(e->prog);
                if (e->sd == 3 && e->n == 5) ok = table_matches<3, 5>(e->prog);
                if (e->sd == 3 && e->n == 4) ok = table_matches<3, 4>(e->prog);
                if (e->sd == 3 && e->n == 3) ok = table_matches<3, 3>(e->prog);
                if (e->sd == 2 && e->n == 6) ok = table_matches<2, 6>(e->prog);
                if (e->sd == 2 && e->n == 5) ok = table_matches<2, 5>(e->prog);
                if (e->sd == 3 && e->n == 2) ok = table_matches<3, 2>(e->prog);
                if (e->sd == 2 && e->n == 4) ok = table_matches<2, 4>(e->prog);
                if (e->sd == 2 && e->n == 3) ok = table_matches<2, 3>(e->prog);
                if (e->sd == 3 && e->n == 7) ok = table_matches<3, 7>(e->prog);
                if (e->sd == 2 && e->n == 7) ok = table_matches<2, 7>(e->prog);
                if (e->sd == 2 && e->n == 8) ok = table_matches<2, 8>(e->prog);
                if (!ok) continue;
                int rc = ensure_stacked(ctx, const_cast<fx_element*>(e), order);
                if (rc != FX_OK) return rc;
                if (e->stack_state[order] != 1) continue;
                fxk::StackedArgs<0>& ka = L.khead;
                memset(&ka, 0, sizeof ka);
                L.fcoef.resize(e->prog.steps.size() * 3);
                for (size_t q = 0; q < e->prog.steps.size(); ++q) {
                    L.fcoef[3 * q + 0] = e->prog.steps[q].A;
                    L.fcoef[3 * q + 1] = e->prog.steps[q].B;
                    L.fcoef[3 * q + 2] = e->prog.steps[q].C;
                }
                ka.pts = pts;
                ka.verts = verts;
                ka.out = out;
                const bool dofmajor = inmix || wg_mix1;
                ka.afrag = pio ? e->d_astack_dmp[order] : dofmajor ? e->d_astack_dm[order] : e->d_astack[order];
                if (!ka.afrag) continue;
                if (pio) {
                    for (int q = 0; q < 9; ++q) ka.G[q] = 0.5 * e->A0[q];
                    ka.piola = mapping;
                }
                if (dofmajor && !invert_small(e->sd, e->A0, ka.A0inv)) continue;
                ka.phi0 = e->prog.phi0;
                memcpy(ka.A0, e->A0, sizeof ka.A0);
                memcpy(ka.b0, e->b0, sizeof ka.b0);
                ka.nreq = nreq;
                ka.npts = npts;
                ka.R = (int)R;
                ka.RT = RT;
                ka.debug = a.debug;
                ka.lim_pts = (long long)nreq * npts * e->sd;
                ka.lim_verts = verts ? (long long)nreq * (e->sd + 1) * e->sd : 0;
                ka.lim_out = (long long)nreq * R * npts;
                const int TRk = fxk::stacked_tile_rows(e->sd, pio ? 1 : 0);
                ka.lim_afrag = (dofmajor ? ((long long)((rows + TRk - 1) / TRk) * ntab + 1) : (long long)(RT + 1)) * ((e->nexp + 3) / 4) * 64;
                const bool mixr = pio || k.rtc == -3 || k.rtc == -4 || k.rtc == -5 || (k.rtc == -2 && (mix_odd || mixr1(k.sd, k.n, k.ct, k.g) || (k.sd == 3 && k.n == 6 && k.ct == 3)));
                const int slots = fxk::stacked_mix_slots(e->sd, dofmajor ? ntab : 0, mixr);
                L.klds_bytes = (fxk::WQ_CTL_DOUBLES + STACKED_NW * (fxk::stacked_image_doubles(k.ct, (e->nexp + 3) / 4, slots) + (mixr ? fxk::STACKED_KBUF : 0))) * 8;
                if (wgk) {
                    L.klds_bytes = fxwg::lds_bytes(k.sd, k.n, wg_ctw);
                    L.wg_ct = wg_ctw;
                    L.wg_mix = wg_mix1 ? 1 : 0;
                    ka.gslab = wg_g;
                }
                if (L.klds_bytes > ctx->lds_per_cu) continue;
                const long long groups = wgk ? ((nreq + wg_g - 1) / wg_g) * STACKED_NW : chunked ? nreq * ((npts + 16 * k.ct - 1) / (16 * k.ct)) : (nreq + k.g - 1) / k.g;
                // one workgroup per CU = one wave per SIMD (measured: a second wave per SIMD at half the registers
                // spills in the production phase and gains nothing, 1.45 -> 1.48 ms: the kernel is bound by the
                // shared fp64 MFMA/VALU pipe, not by latencies)
                // (the request-per-workgroup kernel's one-row-tile instances -- values-only requests of up to 64 rows -- are the
                // exception: 256 registers, two workgroups a CU, 320 -> 298 us for degree-5 tetrahedra at 74 points)
                const int wgs = wgk ? fxwg::workgroups_per_cu(k.sd, k.n, wg_ctw, wg_odd, wg_mix1 ? 1 : 0, RT) : 1;
                L.kgrid = (int)std::max<long long>(1, std::min<long long>((groups + STACKED_NW - 1) / STACKED_NW, (long long)ctx->num_cu * wgs));
                L.ncu = ctx->num_cu;
                L.trash = ctx->d_trash;
                L.queue = ctx->d_queue + (size_t)(ctx->launch_seq++ % FX_QUEUE_SLOTS) * 16;
                L.kmix_order = order;
                L.kodd = wgk ? wg_odd : ((k.rtc == 0 && !even) || mix_odd);
                L.kpiola = pio ? mapping : 0;
                if (pio) L.fused_mapping = true;
                L.stacked_id = (int)i;
                L.small_id = -1;
                if (fixed_yields) L.fixed_id = -1;
                break;
            }
        }
    }
    int per_cu = std::max(1, std::min(16, ctx->lds_per_cu / std::max(1, L.lds_bytes)));
    long long want = (long long)ctx->num_cu * per_cu * 4;
    L.grid = (int)std::max<long long>(1, std::min<long long>(a.nitems, want));
    return FX_OK;
}

int run_tabulate(fx_ctx* ctx, const fx_element* e, int order, const Launch& L, hipStream_t s) {
    if (L.args.nitems == 0 || L.args.npts == 0) return FX_OK;
    if (L.fixed_id >= 0) return run_fixed(L, s);
    if (L.stacked_id >= 0) return run_stacked(L, s);
    if (L.coop_id >= 0) return run_coop(L, s);
    if (L.small_id >= 0) return run_small(order, L, s);
    switch (e->sd) {
        case 1: return launch_sd<1>(order, L, s);
        case 2: return launch_sd<2>(order, L, s);
        case 3: return launch_sd<3>(order, L, s);
    }
    return fail(FX_EINVAL, "Invalid number of spatial dimensions");
}

}  // namespace

namespace {
// register-resident shared-point kernel: NP units (pairs, or single doubles for odd tables) per thread and table slice; NP is
// capped so that the reference values stay in registers (NP * EL * NTAB * NE doubles), larger tables take more slices
template <int SD, int ORDER, bool PIOLA, int EL>
bool launch_shared_reg_sliced(int units, const fxk::SharedArgs& sa_in, int ncu, hipStream_t s) {
    constexpr int NTAB = fxk::NTab<SD, ORDER>::value;
    constexpr int PER = EL * NTAB * (PIOLA ? SD : 1);                       // doubles per unit slot
    constexpr int NPMAX = PER >= 80 ? 1 : PER >= 40 ? 2 : PER >= 27 ? 3 : 4;  // <= ~120 doubles of reference values per thread
    const int need = (units + 255) / 256;
    const int np = std::min(need, NPMAX);
    const int slices = (need + np - 1) / np;
    // requests per block: 64 for large batches (their K in LDS at once); batches of few large requests -- 976 requests of
    // 820 KB, six slices -- left most CUs idle at 64 a block (96 workgroups: 12 % of the HBM peak), so a block shrinks until
    // there are ~4 workgroups per CU
    fxk::SharedArgs sa = sa_in;
    const long long want = 4LL * ncu;
    sa.rb = (int)std::max<long long>(1, std::min<long long>(fxk::SHARED_RB, sa.nreq * slices / want));
    const long long blocks = (sa.nreq + sa.rb - 1) / sa.rb;
    const dim3 g((unsigned)std::max<long long>(1, std::min<long long>(blocks, std::max<long long>(1, 8LL * ncu / slices))), (unsigned)slices);
    switch (np) {
        case 1: hipLaunchKernelGGL((fxk::shared_points_reg_kernel<SD, ORDER, 1, PIOLA, EL>), g, dim3(256), 0, s, sa); return true;
        case 2: if constexpr (NPMAX >= 2) { hipLaunchKernelGGL((fxk::shared_points_reg_kernel<SD, ORDER, 2, PIOLA, EL>), g, dim3(256), 0, s, sa); return true; } break;
        case 3: if constexpr (NPMAX >= 3) { hipLaunchKernelGGL((fxk::shared_points_reg_kernel<SD, ORDER, 3, PIOLA, EL>), g, dim3(256), 0, s, sa); return true; } break;
        case 4: if constexpr (NPMAX >= 4) { hipLaunchKernelGGL((fxk::shared_points_reg_kernel<SD, ORDER, 4, PIOLA, EL>), g, dim3(256), 0, s, sa); return true; } break;
    }
    return false;
}

#ifndef FX_SHARED_FLAT_MAX
#define FX_SHARED_FLAT_MAX 12   // doubles of tables per request up to which the flat small-request kernel takes the launch (odd tables: 32)
#endif
static inline bool noflat_shared(unsigned policy) { return (policy & FX_POLICY_NO_SHARED_REG) != 0 && (policy & FX_POLICY_NO_SHARED_WAVE) != 0; }

template <int SD>
int launch_shared(int order, const fxk::SharedArgs& sa, int ncu, hipStream_t s, unsigned policy) {
    const int table = sa.rows * sa.npts;
    const int grid = (int)std::max<long long>(1, std::min<long long>((sa.nreq + fxk::SHARED_RB - 1) / fxk::SHARED_RB, (long long)ncu * 8));
    // tiny requests (<= FX_SHARED_FLAT_MAX doubles of tables): a wave per 64 requests, flat contiguous output
    {
        const long long total = (long long)fx::binom(SD + order, SD) * table;
        // (tools/coverage_map_cells.py, flat kernel for <= 64 doubles against the kernels below: P0 / DG0 with derivatives 4-15 ->
        // 20-64 % of the HBM peak, P1 triangles at 3 points 30-32 -> 36-57 %; but P1 tetrahedra at 4 points 50-55 -> 30-43 % and
        // P2 triangles 52 -> 41 %: those keep the wave / register-resident kernels)
        if (!noflat_shared(policy) && (total <= FX_SHARED_FLAT_MAX || ((table & 1) && total <= 32)) && (sa.kind == 0 || sa.vdim == SD)) {
            const long long teams = (sa.nreq + 63) / 64;
            const int fgrid = (int)std::max<long long>(1, std::min<long long>((teams + 3) / 4, (long long)grid));
            switch (order) {
                case 0: hipLaunchKernelGGL((fxk::shared_points_flat_kernel<SD, 0>), dim3(fgrid), dim3(256), 0, s, sa); break;
                case 1: hipLaunchKernelGGL((fxk::shared_points_flat_kernel<SD, 1>), dim3(fgrid), dim3(256), 0, s, sa); break;
                default: hipLaunchKernelGGL((fxk::shared_points_flat_kernel<SD, 2>), dim3(fgrid), dim3(256), 0, s, sa); break;
            }
            HIP_TRY(hipGetLastError());
            return FX_OK;
        }
    }
    // small affine requests, order <= 1: one wave per request, line-aligned 1 KB stores
    const bool nowave = (policy & FX_POLICY_NO_SHARED_WAVE) != 0;
    if (!nowave && sa.kind == 0 && order == 1 && (table & 1) == 0) {
        const int npairs = (1 + SD) * table / 2;
        const int ns = (npairs + 63) / 64;       // 1 KB slots per request
        const int nwr = ns <= 8 ? 1 : ns <= 16 ? 2 : 4;  // waves per request, 8 slots each
        const long long teams = (sa.nreq + 63) / 64;     // a team takes 64 requests at a time
        const int wgrid = (int)std::max<long long>(1, std::min<long long>((teams * nwr + 3) / 4, (long long)grid));
        bool ok = ns <= 32;
        if (ok) hipLaunchKernelGGL((fxk::shared_points_wave_kernel<SD, 8>), dim3(wgrid), dim3(256), 0, s, sa, nwr);
        if (ok) {
            HIP_TRY(hipGetLastError());
            return FX_OK;
        }
    }
    const bool noreg = (policy & FX_POLICY_NO_SHARED_REG) != 0;
    // odd tables: register-resident kernel with one double per slot (order <= 1: NTAB sources per slot)
    // register-resident kernel: reference values of each thread's fixed output positions in registers; tables of odd size go
    // double by double (order <= 1), even ones pair by pair; any table size (sliced)
    if (!noreg && (sa.kind == 0 || sa.vdim == SD)) {
        const bool piola = sa.kind != 0;
        bool ok = false;
        if (table & 1) {
            if (order == 0) ok = piola ? launch_shared_reg_sliced<SD, 0, true, 1>(table, sa, ncu, s) : launch_shared_reg_sliced<SD, 0, false, 1>(table, sa, ncu, s);
            if (order == 1) ok = piola ? launch_shared_reg_sliced<SD, 1, true, 1>(table, sa, ncu, s) : launch_shared_reg_sliced<SD, 1, false, 1>(table, sa, ncu, s);
            // (odd tables with Hessians -- N3 / RT2 tetrahedra, P5 triangles at their default rules -- ran on the
            // one-workgroup-per-request fallback at 1-20 % of the HBM peak: tools/coverage_map_cells.py)
            if (order == 2) ok = piola ? launch_shared_reg_sliced<SD, 2, true, 1>(table, sa, ncu, s) : launch_shared_reg_sliced<SD, 2, false, 1>(table, sa, ncu, s);
        } else {
            if (order == 0) ok = piola ? launch_shared_reg_sliced<SD, 0, true, 2>(table / 2, sa, ncu, s) : launch_shared_reg_sliced<SD, 0, false, 2>(table / 2, sa, ncu, s);
            if (order == 1) ok = piola ? launch_shared_reg_sliced<SD, 1, true, 2>(table / 2, sa, ncu, s) : launch_shared_reg_sliced<SD, 1, false, 2>(table / 2, sa, ncu, s);
            if (order == 2) ok = piola ? launch_shared_reg_sliced<SD, 2, true, 2>(table / 2, sa, ncu, s) : launch_shared_reg_sliced<SD, 2, false, 2>(table / 2, sa, ncu, s);
        }
        if (ok) {
            HIP_TRY(hipGetLastError());
            return FX_OK;
        }
    }
    // one workgroup per request, persistent (the fallback: reads the reference tables from L2 for every request)
    const int rgrid = (int)std::max<long long>(1, std::min<long long>(sa.nreq, (long long)ncu * 8));
    switch (order) {
        case 0: hipLaunchKernelGGL((fxk::shared_points_kernel<SD, 0>), dim3(rgrid), dim3(256), 0, s, sa); break;
        case 1: hipLaunchKernelGGL((fxk::shared_points_kernel<SD, 1>), dim3(rgrid), dim3(256), 0, s, sa); break;
        default: hipLaunchKernelGGL((fxk::shared_points_kernel<SD, 2>), dim3(rgrid), dim3(256), 0, s, sa); break;
    }
    HIP_TRY(hipGetLastError());
    return FX_OK;
}
}  // namespace

namespace {
// ---- fused prism kernel (prism_small.hpp): <N (degree of the triangle factor's expansion set), NN (nodes of the interval factor), ORDER> ----
template <int N, int NN, int ORDER>
void launch_prism_small(const fxk::PrismArgs& a, int grid, size_t lds, hipStream_t s) {
    auto kern = fxk::prism_small_kernel<N, NN, ORDER, 4>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
}
template <int N, int NN>
bool launch_prism_small_order(int order, const fxk::PrismArgs& a, int grid, size_t lds, hipStream_t s) {
    if (order == 0) launch_prism_small<N, NN, 0>(a, grid, lds, s);
    else if (order == 1) launch_prism_small<N, NN, 1>(a, grid, lds, s);
    else if (order == 2 && N <= 2 && NN <= 3) {
        if constexpr (N <= 2 && NN <= 3) launch_prism_small<N, NN, 2>(a, grid, lds, s);
    } else return false;
    return true;
}
template <int N>
bool launch_prism_small_nn(int nn, int order, const fxk::PrismArgs& a, int grid, size_t lds, hipStream_t s) {
    switch (nn) {
        case 1: return launch_prism_small_order<N, 1>(order, a, grid, lds, s);
        case 2: return launch_prism_small_order<N, 2>(order, a, grid, lds, s);
        case 3: return launch_prism_small_order<N, 3>(order, a, grid, lds, s);
        case 4: return launch_prism_small_order<N, 4>(order, a, grid, lds, s);
    }
    return false;
}
}  // namespace

namespace {
// ---- lane-local kernel for small tensor-product requests (tensor_small.hpp): <NF, NN, ORDER> ----
template <int NF, int NN, int ORDER>
static void launch_tensor_small(bool grid_mode, const fxk::TensorArgs& a, int P, int img_doubles, int grid, size_t lds, hipStream_t s) {
    if (grid_mode) {
        auto kern = fxk::tensor_small_kernel<NF, NN, ORDER, true>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, P, img_doubles);
    } else {
        auto kern = fxk::tensor_small_kernel<NF, NN, ORDER, false>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, P, img_doubles);
    }
}

// registered shapes: every factor with the same NN nodes; true if launched
static bool try_tensor_small(fx_ctx* ctx, bool grid_mode, const fxk::TensorArgs& a, hipStream_t s) {
    if ((ctx->policy & FX_POLICY_NO_SMALL) != 0 || a.nf < 2 || a.npts > 64) return false;
    const int nn = a.L[0].nn;
    for (int f = 1; f < a.nf; ++f)
        if (a.L[f].nn != nn) return false;
    long long ndof = 1;
    for (int f = 0; f < a.nf; ++f) ndof *= nn;
    const long long total = (long long)a.ntab * ndof * a.npts;  // doubles per request
    constexpr long long IMG_BYTES = 16 * 1024;                  // per wave: four waves per workgroup, two workgroups per CU
    if (total * 8 > IMG_BYTES) return false;
    const int P = (int)std::max<long long>(1, std::min<long long>(64 / a.npts, IMG_BYTES / (total * 8)));
    const int img_doubles = (int)((P * total + 1) & ~1LL);
    const size_t lds = (size_t)img_doubles * 8 * 4;
    const long long items = (a.nreq + P - 1) / P;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)ctx->lds_per_cu / std::max<size_t>(lds, 1)));
    const int grid = (int)std::max<long long>(1, std::min<long long>((items + 3) / 4, (long long)ctx->num_cu * per_cu * 2));
    const int key = a.nf * 100 + nn * 10 + a.order;
    switch (key) {
        case 220: launch_tensor_small<2, 2, 0>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 221: launch_tensor_small<2, 2, 1>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 222: launch_tensor_small<2, 2, 2>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 230: launch_tensor_small<2, 3, 0>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 231: launch_tensor_small<2, 3, 1>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 232: launch_tensor_small<2, 3, 2>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 240: launch_tensor_small<2, 4, 0>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 241: launch_tensor_small<2, 4, 1>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 242: launch_tensor_small<2, 4, 2>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 250: launch_tensor_small<2, 5, 0>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 251: launch_tensor_small<2, 5, 1>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 320: launch_tensor_small<3, 2, 0>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 321: launch_tensor_small<3, 2, 1>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 322: launch_tensor_small<3, 2, 2>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
        case 330: launch_tensor_small<3, 3, 0>(grid_mode, a, P, img_doubles, grid, lds, s); return true;
    }
    return false;
}

}  // namespace

extern "C" {

int fx_tabulate_batch(fx_ctx* ctx, const fx_element* e, int order, int64_t nreq, int npts, const double* pts,
                      const double* verts, double* out, void* stream) {
    if (ctx && e && order > 2) {  // differentiation-matrix route (ensure_high_order)
        if (order > FX_MAX_ORDER) return fail(FX_ENOTIMPL, "derivative order %d > %d is not implemented on the device", order, FX_MAX_ORDER);
        if (nreq < 0 || npts < 0) return fail(FX_EINVAL, "negative batch size");
        if (nreq == 0 || npts == 0) return FX_OK;
        int rc = ensure_high_order(ctx, const_cast<fx_element*>(e), order);
        if (rc != FX_OK) return rc;
        // with per-request cells the order-0 kernels map the points into the element's cell: the tables are derivatives
        // with respect to the ELEMENT's coordinates, and the chain rule follows as a pass over the tables
        rc = fx_tabulate_batch(ctx, e->high[order], 0, nreq, npts, pts, verts, out, stream);
        if (rc != FX_OK || !verts) return rc;
        return mix_high_order(ctx, e, order, nreq, npts, verts, out, (hipStream_t)stream);
    }
    Launch L;
    int rc = plan_launch(ctx, e, order, nreq, npts, pts, verts, out, L);
    if (rc != FX_OK) return rc;
    return run_tabulate(ctx, e, order, L, (hipStream_t)stream);
}

int fx_tabulate_batch_mapped(fx_ctx* ctx, const fx_element* e, int mapping, int order, int64_t nreq, int npts,
                             const double* pts, const double* verts, double* out, void* stream) {
    if (mapping < FX_MAP_AFFINE || mapping > FX_MAP_COVARIANT_CONTRAVARIANT_PIOLA) return fail(FX_EINVAL, "unknown mapping %d", mapping);
    if (mapping == FX_MAP_AFFINE) return fx_tabulate_batch(ctx, e, order, nreq, npts, pts, verts, out, stream);
    if (!e) return fail(FX_EINVAL, "null context/element");
    if (mapping >= FX_MAP_DOUBLE_COVARIANT_PIOLA) {
        if (e->vdim != e->sd * e->sd || e->sd < 2)
            return fail(FX_EINVAL, "double Piola maps need matrix-valued functions with value shape (%d, %d), got %d components",
                        e->sd, e->sd, e->vdim);
    } else if (e->vdim != e->sd || e->sd < 2)
        return fail(FX_EINVAL, "Piola maps need vector-valued functions with value shape (%d,), got %d components", e->sd, e->vdim);
    if (!verts && nreq > 0 && npts > 0) return fail(FX_EINVAL, "a Piola push-forward needs the physical cells (verts)");
    Launch L;
    int rc = plan_launch(ctx, e, order, nreq, npts, pts, verts, out, L, mapping);
    if (rc != FX_OK) return rc;
    rc = run_tabulate(ctx, e, order, L, (hipStream_t)stream);
    if (rc != FX_OK || L.fused_mapping) return rc;
    return fx_pushforward_batch(ctx, e, mapping, order, nreq, npts, verts, out, stream);
}

int fx_pushforward_batch(fx_ctx* ctx, const fx_element* e, int mapping, int order, int64_t nreq, int npts,
                         const double* verts, double* out, void* stream) {
    if (!ctx || !e) return fail(FX_EINVAL, "null context/element");
    if (mapping < FX_MAP_AFFINE || mapping > FX_MAP_COVARIANT_CONTRAVARIANT_PIOLA) return fail(FX_EINVAL, "unknown mapping %d", mapping);
    if (order < 0) return fail(FX_EINVAL, "negative derivative order");
    if (order > 2) return fail(FX_ENOTIMPL, "derivative order %d > 2 is not implemented on the device", order);
    if (nreq < 0 || npts < 0) return fail(FX_EINVAL, "negative batch size");
    if (mapping == FX_MAP_AFFINE || nreq == 0 || npts == 0) return FX_OK;  // derivatives are w.r.t. physical x already
    if (mapping >= FX_MAP_DOUBLE_COVARIANT_PIOLA) {
        if (e->vdim != e->sd * e->sd || e->sd < 2)
            return fail(FX_EINVAL, "double Piola maps need matrix-valued functions with value shape (%d, %d), got %d components",
                        e->sd, e->sd, e->vdim);
    } else if (e->vdim != e->sd || e->sd < 2)
        return fail(FX_EINVAL, "Piola maps need vector-valued functions with value shape (%d,), got %d components", e->sd, e->vdim);
    if (!verts || !out) return fail(FX_EINVAL, "null device pointer");
    fxk::PiolaArgs pa;
    pa.verts = verts;
    pa.out = out;
    for (int i = 0; i < 9; ++i) pa.G[i] = 0.5 * e->A0[i];
    pa.ntab = fx::binom(e->sd + order, e->sd);
    pa.ndof = e->ndof;
    pa.npts = npts;
    pa.kind = mapping;
    pa.nreq = nreq;
    const long long per = (long long)pa.ntab * pa.ndof * npts;  // elements (vectors / matrices) per request
    if (per > 0x3fffffffLL / fxk::PIOLA_RB) return fail(FX_EINVAL, "request too large for the push-forward pass");
    pa.rb = (int)std::max<long long>(1, std::min<long long>(fxk::PIOLA_RB, 2048 / std::max<long long>(per, 1)));
    const long long blocks = (nreq + pa.rb - 1) / pa.rb;
    const unsigned pgrid = (unsigned)std::max<long long>(1, std::min<long long>(blocks, (long long)ctx->num_cu * 16));
    if (e->sd == 2)
        hipLaunchKernelGGL(fxk::piola_apply_kernel<2>, dim3(pgrid), dim3(256), 0, (hipStream_t)stream, pa);
    else
        hipLaunchKernelGGL(fxk::piola_apply_kernel<3>, dim3(pgrid), dim3(256), 0, (hipStream_t)stream, pa);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

int fx_tabulate_batch_shared(fx_ctx* ctx, const fx_element* e, int mapping, int order, int64_t nreq, int npts,
                             const double* ref_pts, const double* verts, double* out, void* stream) {
    if (!ctx || !e) return fail(FX_EINVAL, "null context/element");
    if (mapping < FX_MAP_AFFINE || mapping > FX_MAP_CONTRAVARIANT_PIOLA) return fail(FX_EINVAL, "unknown mapping %d", mapping);
    if (order < 0) return fail(FX_EINVAL, "negative derivative order");
    if (order > 2) return fail(FX_ENOTIMPL, "derivative order %d > 2 is not implemented on the device", order);
    if (nreq < 0 || npts < 0) return fail(FX_EINVAL, "negative batch size");
    if (nreq == 0 || npts == 0) return FX_OK;
    if (!ref_pts || !verts || !out) return fail(FX_EINVAL, "null device pointer");
    if (mapping != FX_MAP_AFFINE && (e->vdim != e->sd || e->sd < 2))
        return fail(FX_EINVAL, "Piola maps need vector-valued functions with value shape (%d,), got %d components", e->sd, e->vdim);
    const int ntab = fx::binom(e->sd + order, e->sd);
    const int rows = e->ndof * e->vdim;
    const size_t need = (size_t)ntab * rows * npts * sizeof(double);
    hipStream_t s = (hipStream_t)stream;
    // Tiny requests (<= 2 KB of tables each: P0-P2, N1 / RT1 / BDM1 at their small rules, with derivatives or a Piola map): the
    // streaming kernels below spend a block pass on K, two divisions per output double and stores of a few hundred bytes -- 20-46 % of
    // the HBM peak (tools/coverage_map_cells.py) where the lane-local kernel, which RE-COMPUTES the tables per request from the
    // point's jets, reaches 60-90 % on per-request points (tools/coverage_map.py --verts --pushforward).  It takes them here too:
    // same kernel, `pts` = the one reference point set (SmallArgs::shared_pts), chain rule and Piola map through the request's cell
    // as for per-request points.  Scalar values-only requests stay (a copy of the reference table: 0.8-1.2 x), as does everything
    // the planner would not give to the lane-local kernel.
    // (triangles: up to 3 KB, and scalar values of degree >= 2 too -- P2 / P3 / P4 at their default rules 0.78-0.88 of the streaming
    // kernels' time in the two maps; P1 values, P2 tetrahedra values and N1 tetrahedra with gradients at 2.3 KB are 1.12-1.23: they stay)
    const bool scalar_values = order == 0 && mapping == FX_MAP_AFFINE;
    if (need <= (e->sd == 2 ? 3072u : 2048u) && !(ctx->policy & FX_POLICY_NO_SMALL) && !(scalar_values && !(e->sd == 2 && e->n >= 2)) && e->sd >= 2) {
        Launch L;
        const int prc = plan_launch(ctx, e, order, nreq, npts, ref_pts, verts, out, L, mapping);
        if (prc == FX_OK && L.small_id >= 0 && L.fixed_id < 0 && L.stacked_id < 0 && L.coop_id < 0 &&
            (mapping == FX_MAP_AFFINE || L.fused_mapping)) {
            L.sargs.shared_pts = 1;
            return run_small(order, L, s);
        }
    }
    // Reference-cell tables: scratch of THIS call, allocated and released in stream order (hipMallocAsync /
    // hipFreeAsync on the caller's stream), so calls on different streams never share a buffer and the memory
    // returns to the pool only after the streaming kernel below has read it.
    HIP_TRY(hipSetDevice(ctx->device));
    double* d_ref = nullptr;
    HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&d_ref), need, s));
    struct Release {
        double* p;
        hipStream_t s;
        ~Release() { (void)hipFreeAsync(p, s); }
    } release{d_ref, s};
    // the element on its own cell at the shared points, once
    int rc = fx_tabulate_batch(ctx, e, order, 1, npts, ref_pts, nullptr, d_ref, stream);
    if (rc != FX_OK) return rc;
    fxk::SharedArgs sa;
    sa.ref = d_ref;
    sa.verts = verts;
    sa.out = out;
    if (!invert_small(e->sd, e->A0, sa.A0inv)) return fail(FX_EINVAL, "degenerate cell");
    sa.nreq = nreq;
    sa.rows = rows;
    sa.vdim = e->vdim;
    sa.npts = npts;
    sa.kind = mapping;
    // persistent workgroups; the register-resident kernel takes blocks of 256 requests
    sa.rb = fxk::SHARED_RB;
    switch (e->sd) {
        case 1: return launch_shared<1>(order, sa, ctx->num_cu, s, ctx->policy);
        case 2: return launch_shared<2>(order, sa, ctx->num_cu, s, ctx->policy);
        case 3: return launch_shared<3>(order, sa, ctx->num_cu, s, ctx->policy);
    }
    return fail(FX_EINVAL, "Invalid number of spatial dimensions");
}

int fx_collapsed_quadrature(fx_ctx* ctx, int sd, int m, const double* verts, double* pts, double* wts, void* stream) {
    if (!ctx) return fail(FX_EINVAL, "null context");
    if (sd < 1 || sd > 3) return fail(FX_EINVAL, "Invalid number of spatial dimensions");
    if (m < 1 || m > fxk::GJ_MAX) return fail(FX_EINVAL, "points per direction must be in [1, %d]", fxk::GJ_MAX);
    if (!pts || !wts) return fail(FX_EINVAL, "null device pointer");
    static const double UFC[3][12] = {{0, 1}, {0, 0, 1, 0, 0, 1}, {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1}};
    fxk::RuleArgs ra;
    double A[9], b[3];
    if (!host_cell_map(sd, verts ? verts : UFC[sd - 1], A, b) || !invert_small(sd, A, ra.Ainv)) return fail(FX_EINVAL, "degenerate cell");
    for (int i = 0; i < 3; ++i) ra.b[i] = i < sd ? b[i] : 0.0;
    double det = sd == 1 ? ra.Ainv[0]
                 : sd == 2 ? ra.Ainv[0] * ra.Ainv[3] - ra.Ainv[1] * ra.Ainv[2]
                           : ra.Ainv[0] * (ra.Ainv[4] * ra.Ainv[8] - ra.Ainv[5] * ra.Ainv[7]) -
                                 ra.Ainv[1] * (ra.Ainv[3] * ra.Ainv[8] - ra.Ainv[5] * ra.Ainv[6]) +
                                 ra.Ainv[2] * (ra.Ainv[3] * ra.Ainv[7] - ra.Ainv[4] * ra.Ainv[6]);
    ra.jac = std::fabs(det);
    ra.m = m;
    const size_t need = (size_t)2 * 3 * fxk::GJ_MAX * sizeof(double);
    hipStream_t s = (hipStream_t)stream;
    // the 1-D rules: per-call scratch in stream order (see fx_tabulate_batch_shared)
    HIP_TRY(hipSetDevice(ctx->device));
    double* d_rules = nullptr;
    HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&d_rules), need, s));
    struct Release {
        double* p;
        hipStream_t s;
        ~Release() { (void)hipFreeAsync(p, s); }
    } release{d_rules, s};
    ra.rules = d_rules;
    ra.pts = pts;
    ra.wts = wts;
    hipLaunchKernelGGL(fxk::gauss_jacobi_kernel, dim3(1), dim3(64), 0, s, m, sd, d_rules);
    int total = 1;
    for (int d = 0; d < sd; ++d) total *= m;
    const int grid = (total + 255) / 256;
    if (sd == 1) hipLaunchKernelGGL(fxk::collapsed_rule_kernel<1>, dim3(grid), dim3(256), 0, s, ra);
    if (sd == 2) hipLaunchKernelGGL(fxk::collapsed_rule_kernel<2>, dim3(grid), dim3(256), 0, s, ra);
    if (sd == 3) hipLaunchKernelGGL(fxk::collapsed_rule_kernel<3>, dim3(grid), dim3(256), 0, s, ra);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

int fx_classify_tables(fx_ctx* ctx, int64_t ntables, int rows, int npts, double rtol, const double* tables, double* stats,
                       void* stream) {
    if (!ctx) return fail(FX_EINVAL, "null context");
    if (ntables < 0 || rows < 1 || npts < 1) return fail(FX_EINVAL, "bad table shape (%lld, %d, %d)", (long long)ntables, rows, npts);
    if (ntables > 0x7fffffffLL) return fail(FX_EINVAL, "too many tables for one launch");
    if ((long long)rows * npts > 0x7fffffffLL) return fail(FX_EINVAL, "table too large");
    if (ntables == 0) return FX_OK;
    if (!tables || !stats) return fail(FX_EINVAL, "null device pointer");
    fxk::ClassifyArgs ca{tables, stats, rows, npts, rtol};
    hipLaunchKernelGGL(fxk::classify_tables_kernel, dim3((unsigned)ntables), dim3(256), 0, (hipStream_t)stream, ca);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

int fx_tables_squared_norm(fx_ctx* ctx, int64_t ntables, int rows, int vdim, int npts, const double* tables, const double* weights,
                           double* out, void* stream) {
    if (!ctx) return fail(FX_EINVAL, "null context");
    if (ntables < 0 || rows < 1 || vdim < 1 || npts < 1)
        return fail(FX_EINVAL, "bad table shape (%lld, %d, %d, %d)", (long long)ntables, rows, vdim, npts);
    if ((long long)vdim * npts > 0x7fffffffLL) return fail(FX_EINVAL, "table rows too long");
    const long long nrows = (long long)ntables * rows;
    if ((nrows + 3) / 4 > 0x7fffffffLL) return fail(FX_EINVAL, "too many rows for one launch");
    if (ntables == 0) return FX_OK;
    if (!tables || !weights || !out) return fail(FX_EINVAL, "null device pointer");
    fxk::SquaredNormArgs sa{tables, weights, out, rows, vdim, npts};
    hipLaunchKernelGGL(fxk::squared_norm_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, sa, nrows);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

int fx_tables_point_major(fx_ctx* ctx, int64_t ntables, int rows, int npts, const double* in, double* out, void* stream) {
    if (!ctx) return fail(FX_EINVAL, "null context");
    if (ntables < 0 || rows < 1 || npts < 1) return fail(FX_EINVAL, "bad table shape (%lld, %d, %d)", (long long)ntables, rows, npts);
    if (ntables == 0) return FX_OK;
    if (!in || !out || in == out) return fail(FX_EINVAL, "null or aliased device pointers");
    const long long tiles = (long long)((rows + 31) / 32) * ((npts + 31) / 32);
    if (ntables * tiles > 0x7fffffffLL) return fail(FX_EINVAL, "too many tiles for one launch");
    fxk::PointMajorArgs pa{in, out, rows, npts};
    hipLaunchKernelGGL(fxk::point_major_kernel, dim3((unsigned)(ntables * tiles)), dim3(256), 0, (hipStream_t)stream, pa);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

int fx_plan_kernel(fx_ctx* ctx, const fx_element* e, int order, int64_t nreq, int npts, int has_verts, char* name,
                   int name_len) {
    if (!name || name_len < 1) return fail(FX_EINVAL, "fx_plan_kernel: bad name buffer");
    Launch L;
    // plan_launch only records the pointers; a non-null dummy stands for "per-request cells given"
    static double dummy;
    int rc = plan_launch(ctx, e, order, nreq, npts, &dummy, (has_verts & 1) ? &dummy : nullptr, &dummy, L, (has_verts >> 2) & 3);
    if (rc != FX_OK) return rc;
    if ((has_verts & 2) && L.stacked_id >= 0 && L.fixed_id < 0) {  // with the instance of the stacked-matrix registry
        const StackedShape& k = kStackedShapes[L.stacked_id];
        if (k.rtc == -7) snprintf(name, (size_t)name_len, "fxk::tabulate_simplex_wg<%d,%d,%d>%s", k.sd, k.n, L.wg_ct,
                                  ((L.khead.gslab > 1 ? "x" + std::to_string(L.khead.gslab) : std::string()) + (L.wg_mix ? "+mix" : "")).c_str());
        else snprintf(name, (size_t)name_len, "fxk::tabulate_simplex_stacked<%d,%d,%d,%d,%d>%s", k.sd, k.n, k.ct, k.g, k.rtc,
                      L.kpiola ? "+piola" : "");
        return FX_OK;
    }
    const char* k = "fxk::tabulate_simplex_kernel";
    if (L.fixed_id >= 0)
        k = L.fkind == 0   ? "fxk::tabulate_simplex_fixed"
            : L.fkind == 1 ? "fxk::tabulate_simplex_stream"
                           : "fxk::tabulate_simplex_pair";
    else if (L.stacked_id >= 0)
        k = kStackedShapes[L.stacked_id].rtc == -7 ? "fxk::tabulate_simplex_wg" : "fxk::tabulate_simplex_stacked";
    else if (L.coop_id >= 0) k = "fxk::tabulate_simplex_coop";
    else if (L.small_id >= 0) k = "fxk::tabulate_simplex_small";
    snprintf(name, (size_t)name_len, "%s", k);
    return FX_OK;
}

namespace {
// after a synchronisation point: did a dynamically scheduled kernel give up waiting for a chunk id
// (work_queue.hpp)?  Its output is then incomplete.
int check_work_queues(fx_ctx* ctx) {
    unsigned int flags[FX_QUEUE_SLOTS * 32];
    HIP_TRY(hipMemcpy(flags, ctx->d_queue, sizeof flags, hipMemcpyDeviceToHost));
    for (int i = 0; i < FX_QUEUE_SLOTS; ++i)
        if (flags[i * 32 + 2]) {
            (void)hipMemset(ctx->d_queue, 0, FX_QUEUE_SLOTS * 128);
            return fail(FX_EHIP, "work queue protocol error in a tabulation kernel: the output is incomplete");
        }
    return FX_OK;
}
}  // namespace

int fx_time_tabulate_batch(fx_ctx* ctx, const fx_element* e, int order, int64_t nreq, int npts, const double* pts,
                           const double* verts, double* out, void* stream, int reps, float* ms) {
    if (reps < 1 || !ms) return fail(FX_EINVAL, "fx_time_tabulate_batch: bad reps/ms");
    Launch L;
    int rc = plan_launch(ctx, e, order, nreq, npts, pts, verts, out, L);
    if (rc != FX_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t t0, t1;
    HIP_TRY(hipEventCreate(&t0));
    HIP_TRY(hipEventCreate(&t1));
    HIP_TRY(hipEventRecord(t0, s));
    for (int i = 0; i < reps; ++i) {
        rc = run_tabulate(ctx, e, order, L, s);
        if (rc != FX_OK) break;
    }
    HIP_TRY(hipEventRecord(t1, s));
    HIP_TRY(hipEventSynchronize(t1));
    float total = 0.f;
    HIP_TRY(hipEventElapsedTime(&total, t0, t1));
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    *ms = total / reps;
    if (rc == FX_OK) rc = check_work_queues(ctx);
    return rc;
}

int fx_tabulate_batch_host(fx_ctx* ctx, const fx_element* e, int order, int64_t nreq, int npts, const double* pts,
                           const double* verts, double* out) {
    if (!ctx || !e) return fail(FX_EINVAL, "null context/element");
    if (order < 0 || order > 2) return fail(order < 0 ? FX_EINVAL : FX_ENOTIMPL, "unsupported derivative order %d", order);
    HIP_TRY(hipSetDevice(ctx->device));
    const int ntab = fx::binom(e->sd + order, e->sd);
    size_t pbytes = (size_t)nreq * npts * e->sd * 8, vbytes = (size_t)nreq * (e->sd + 1) * e->sd * 8;
    size_t obytes = (size_t)nreq * ntab * e->ndof * e->vdim * npts * 8;
    if (pbytes == 0 || obytes == 0) return FX_OK;
    double *dp = nullptr, *dv = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&dp, pbytes));
    HIP_TRY(hipMalloc(&dout, obytes));
    HIP_TRY(hipMemcpy(dp, pts, pbytes, hipMemcpyHostToDevice));
    if (verts) {
        HIP_TRY(hipMalloc(&dv, vbytes));
        HIP_TRY(hipMemcpy(dv, verts, vbytes, hipMemcpyHostToDevice));
    }
    int rc = fx_tabulate_batch(ctx, e, order, nreq, npts, dp, dv, dout, nullptr);
    if (rc == FX_OK) {
        hipError_t he = hipDeviceSynchronize();
        if (he != hipSuccess) rc = fail(FX_EHIP, "tabulate kernel: %s", hipGetErrorString(he));
    }
    if (rc == FX_OK) rc = check_work_queues(ctx);
    if (rc == FX_OK) {
        hipError_t he = hipMemcpy(out, dout, obytes, hipMemcpyDeviceToHost);
        if (he != hipSuccess) rc = fail(FX_EHIP, "copy back: %s", hipGetErrorString(he));
    }
    (void)hipFree(dp);
    (void)hipFree(dout);
    if (dv) (void)hipFree(dv);
    return rc;
}

// ---------------------------------------------------------------------------------
int fx_riesz_assemble(fx_ctx* ctx, int nrows, int nq, int nexp, const double* wts, const double* ev, double* mat,
                      void* stream) {
    if (!ctx || nrows < 0 || nq < 0 || nexp < 0) return fail(FX_EINVAL, "fx_riesz_assemble: bad argument");
    long long total = (long long)nrows * nexp;
    if (total == 0) return FX_OK;
    int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(fxk::riesz_assemble_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, nrows, nq, nexp, wts,
                       ev, mat);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

int fx_vandermonde_solve_batch(fx_ctx* ctx, int64_t nsys, int ndof, int m, const double* A, const double* B, double* X,
                               double* Vout, int* info, void* stream) {
    if (!ctx || nsys < 0 || ndof < 1 || m < 1 || !info) return fail(FX_EINVAL, "fx_vandermonde_solve_batch: bad argument");
    if (nsys == 0) return FX_OK;
    size_t lds = ((size_t)ndof * ndof + (size_t)ndof * m) * 8;
    if (lds > 150 * 1024) {
        // beyond the LDS: the same elimination with M and the right-hand sides in a global workspace
        if (nsys > 65535) return fail(FX_EINVAL, "fx_vandermonde_solve_batch: too many large systems in one call");
        double* ws = nullptr;
        HIP_TRY(hipMalloc(&ws, (size_t)nsys * lds));
        hipLaunchKernelGGL(fxk::vandermonde_solve_kernel<true>, dim3((unsigned)nsys), dim3(256), 0, (hipStream_t)stream, ndof, m,
                           A, B, X, Vout, info, ws);
        hipError_t he = hipGetLastError();
        if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream);
        (void)hipFree(ws);
        if (he != hipSuccess) return fail(FX_EHIP, "fx_vandermonde_solve_batch: %s", hipGetErrorString(he));
        return FX_OK;
    }
    auto kern = fxk::vandermonde_solve_kernel<false>;
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)nsys), dim3(256), lds, (hipStream_t)stream, ndof, m, A, B, X, Vout, info,
                       (double*)nullptr);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

// ---------------------------------------------------------------------------------
int fx_line_element_create(fx_ctx* ctx, int nn, const double* nodes, fx_line_element** out) {
    if (!ctx || !nodes || !out) return fail(FX_EINVAL, "fx_line_element_create: null argument");
    // (up to NN_MAX nodes: register-resident kernels, tensor products, prisms; beyond: fx_line_tabulate_batch only)
    if (nn < 1 || nn > fxk::NN_BIG_MAX) return fail(FX_ENOTIMPL, "1-D Lagrange with %d nodes (max %d)", nn, fxk::NN_BIG_MAX);
    HIP_TRY(hipSetDevice(ctx->device));
    fx_line_element* e = new fx_line_element;
    e->ctx = ctx;
    e->nn = nn;
    e->nodes.assign(nodes, nodes + nn);
    // make_dmat (barycentric_interpolation.py:50-59)
    std::vector<double> D((size_t)nn * nn), w(nn);
    for (int i = 0; i < nn; ++i)
        for (int j = 0; j < nn; ++j) D[(size_t)i * nn + j] = (i == j) ? 1.0 : nodes[j] - nodes[i];
    for (int j = 0; j < nn; ++j) {
        double p = 1.0;
        for (int i = 0; i < nn; ++i) p *= D[(size_t)i * nn + j];
        if (p == 0.0) {
            delete e;
            return fail(FX_EINVAL, "repeated interpolation node");
        }
        w[j] = 1.0 / p;
    }
    for (int i = 0; i < nn; ++i)
        for (int j = 0; j < nn; ++j) D[(size_t)i * nn + j] = (w[i] / w[j]) / D[(size_t)i * nn + j];
    std::vector<double> colsum(nn, 0.0);
    for (int i = 0; i < nn; ++i)
        for (int j = 0; j < nn; ++j) colsum[j] += D[(size_t)i * nn + j];
    for (int i = 0; i < nn; ++i) D[(size_t)i * nn + i] -= colsum[i];
    e->wts = w;
    e->dmat = D;
    std::vector<double> buf;
    buf.insert(buf.end(), e->nodes.begin(), e->nodes.end());
    buf.insert(buf.end(), w.begin(), w.end());
    buf.insert(buf.end(), D.begin(), D.end());
    hipError_t he = hipMalloc(&e->d_buf, buf.size() * 8);
    if (he == hipSuccess) he = hipMemcpy(e->d_buf, buf.data(), buf.size() * 8, hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        fx_line_element_destroy(e);
        return fail(FX_EHIP, "fx_line_element_create: %s", hipGetErrorString(he));
    }
    *out = e;
    return FX_OK;
}

int fx_line_element_destroy(fx_line_element* e) {
    if (!e) return FX_OK;
    if (e->d_buf) (void)hipFree(e->d_buf);
    delete e;
    return FX_OK;
}

static fxk::LineDesc line_desc(const fx_line_element* e) {
    fxk::LineDesc L;
    L.nodes = e->d_buf;
    L.wts = e->d_buf + e->nn;
    L.dmat = e->d_buf + 2 * e->nn;
    L.nn = e->nn;
    return L;
}

int fx_line_tabulate_batch(fx_ctx* ctx, const fx_line_element* e, int order, int64_t nreq, int npts, const double* pts,
                           double* out, void* stream) {
    if (!ctx || !e || order < 0 || nreq < 0 || npts < 0) return fail(FX_EINVAL, "fx_line_tabulate_batch: bad argument");
    long long total = (long long)nreq * npts;
    if (total == 0) return FX_OK;
    int grid = (int)((total + 127) / 128);
    if (e->nn > fxk::NN_MAX)
        hipLaunchKernelGGL(fxk::line_tabulate_big_kernel, dim3(grid), dim3(128), 0, (hipStream_t)stream, line_desc(e), order,
                           (long long)nreq, npts, pts, out);
    else
        hipLaunchKernelGGL(fxk::line_tabulate_kernel, dim3(grid), dim3(128), 0, (hipStream_t)stream, line_desc(e), order,
                           (long long)nreq, npts, pts, out);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

static void fill_alpha(int nf, int order, fxk::TensorArgs& a) {
    // mis(nf, k), k = 0..order (polynomial_set.py:23-32)
    int t = 0;
    for (int k = 0; k <= order; ++k) {
        if (nf == 1) {
            a.alpha[t][0] = k;
            ++t;
        } else if (nf == 2) {
            for (int i = 0; i <= k; ++i) {
                a.alpha[t][0] = k - i;
                a.alpha[t][1] = i;
                ++t;
            }
        } else {
            for (int i = 0; i <= k; ++i)
                for (int j = 0; j <= i; ++j) {
                    a.alpha[t][0] = k - i;
                    a.alpha[t][1] = i - j;
                    a.alpha[t][2] = j;
                    ++t;
                }
        }
    }
    a.ntab = t;
}

static int tensor_launch(fx_ctx* ctx, int nf, const fx_line_element* const* factors, int order, int64_t nreq, int npts,
                         int q, const double* pts, double* out, void* stream, bool grid_mode) {
    if (!ctx || !factors || nf < 1 || nf > 3) return fail(FX_EINVAL, "tensor tabulate: 1..3 interval factors supported");
    if (order < 0) return fail(FX_EINVAL, "negative derivative order");
    if (order > 2) return fail(FX_ENOTIMPL, "derivative order %d > 2 is not implemented on the device", order);
    if (nreq < 0 || npts < 0) return fail(FX_EINVAL, "negative batch size");
    if (nreq == 0 || npts == 0) return FX_OK;
    fxk::TensorArgs a;
    memset(&a, 0, sizeof a);
    size_t lds = 0;
    const int w = grid_mode ? q : npts;
    for (int f = 0; f < nf; ++f) {
        if (!factors[f]) return fail(FX_EINVAL, "null factor");
        if (factors[f]->nn > fxk::NN_MAX)
            return fail(FX_ENOTIMPL, "tensor products of 1-D Lagrange factors with %d nodes (max %d)", factors[f]->nn, fxk::NN_MAX);
        a.L[f] = line_desc(factors[f]);
        lds += (size_t)(order + 1) * factors[f]->nn * w * 8;
    }
    {   // + row table: 4 ints per output row
        long long nbf = 1;
        for (int f = 0; f < nf; ++f) nbf *= factors[f]->nn;
        lds += (size_t)fx::binom(nf + order, nf) * nbf * 16;
    }
    a.nf = nf;
    a.order = order;
    fill_alpha(nf, order, a);
    a.nreq = nreq;
    a.npts = npts;
    a.q = q;
    a.pts = pts;
    a.out = out;
    if (try_tensor_small(ctx, grid_mode, a, (hipStream_t)stream)) {
        HIP_TRY(hipGetLastError());
        return FX_OK;
    }
    if (lds > 150 * 1024) return fail(FX_ENOTIMPL, "factor tables exceed LDS (%zu bytes)", lds);
    int block = 256;
    if (const char* e = ab_env("FIAT_AMD_TENSOR_BLOCK")) block = atoi(e);
    int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)(8 * 256 / block), (size_t)ctx->lds_per_cu / std::max<size_t>(lds, 1)));
    int grid = (int)std::min<long long>(nreq, (long long)ctx->num_cu * per_cu * 2);
    // every factor with the same node count (Q_k elements): instances with compile-time loops in the factor phase
    int nnc = factors[0]->nn;
    for (int f = 1; f < nf; ++f)
        if (factors[f]->nn != nnc) nnc = 0;
    using TensorKern = void (*)(fxk::TensorArgs);
    TensorKern kern = grid_mode ? (TensorKern)fxk::tensor_tabulate_kernel<true> : (TensorKern)fxk::tensor_tabulate_kernel<false>;
    switch (nnc) {
        case 2: kern = grid_mode ? (TensorKern)fxk::tensor_tabulate_kernel<true, 2> : (TensorKern)fxk::tensor_tabulate_kernel<false, 2>; break;
        case 3: kern = grid_mode ? (TensorKern)fxk::tensor_tabulate_kernel<true, 3> : (TensorKern)fxk::tensor_tabulate_kernel<false, 3>; break;
        case 4: kern = grid_mode ? (TensorKern)fxk::tensor_tabulate_kernel<true, 4> : (TensorKern)fxk::tensor_tabulate_kernel<false, 4>; break;
        case 5: kern = grid_mode ? (TensorKern)fxk::tensor_tabulate_kernel<true, 5> : (TensorKern)fxk::tensor_tabulate_kernel<false, 5>; break;
    }
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

int fx_tensor_tabulate_batch(fx_ctx* ctx, int nf, const fx_line_element* const* factors, int order, int64_t nreq,
                             int npts, const double* pts, double* out, void* stream) {
    return tensor_launch(ctx, nf, factors, order, nreq, npts, 0, pts, out, stream, false);
}

// (element on a triangle) x (1-D Lagrange element), fused (prism_small.hpp).  FX_ENOTIMPL: shape not registered -- the caller
// takes the general route (factor tables + fx_table_outer_batch).
int fx_prism_tabulate_batch(fx_ctx* ctx, const fx_element* tri, const fx_line_element* line, int order, int64_t nreq, int npts,
                            const double* pts, double* out, void* stream) {
    if (!ctx || !tri || !line) return fail(FX_EINVAL, "fx_prism_tabulate_batch: null context/element");
    if (order < 0 || nreq < 0 || npts < 0) return fail(FX_EINVAL, "fx_prism_tabulate_batch: bad argument");
    if (tri->sd != 2) return fail(FX_EINVAL, "fx_prism_tabulate_batch: the first factor must live on a triangle");
    if (line->nn > fxk::NN_MAX) return fail(FX_ENOTIMPL, "prisms with a 1-D Lagrange factor of %d nodes (max %d)", line->nn, fxk::NN_MAX);
    if (nreq == 0 || npts == 0) return FX_OK;
    if (!pts || !out) return fail(FX_EINVAL, "fx_prism_tabulate_batch: null device pointer");
    const int rows = (int)(tri->hC.size() / (size_t)tri->nexp);
    const int ntab = fx::binom(3 + order, 3);
    const long long reqsize = (long long)ntab * rows * line->nn * npts;
    if ((ctx->policy & FX_POLICY_NO_SMALL) != 0 || order > 2 || npts > 64 || tri->n < 1 || tri->n > 3 || line->nn < 1 || line->nn > 4 ||
        !tri->d_cmat || tri->raw_expansion || (int)tri->prog.steps.size() > fxk::SMALL_MAXSTEPS || reqsize * 8 > 20 * 1024 || rows > 24 ||
        (order == 2 && (tri->n > 2 || line->nn > 3)))
        return fail(FX_ENOTIMPL, "fx_prism_tabulate_batch: shape not registered (degree %d x %d nodes, order %d, %d points, %d rows)",
                    tri->n, line->nn, order, npts, rows);
    bool match = false;
    if (tri->n == 1) match = table_matches<2, 1>(tri->prog);
    if (tri->n == 2) match = table_matches<2, 2>(tri->prog);
    if (tri->n == 3) match = table_matches<2, 3>(tri->prog);
    if (!match) return fail(FX_ENOTIMPL, "fx_prism_tabulate_batch: the triangle factor's recurrence is not the registered one");
    fxk::PrismArgs a;
    memset(&a, 0, sizeof a);
    a.pts = pts;
    a.out = out;
    a.cmat = tri->d_cmat;
    for (size_t k = 0; k < tri->prog.steps.size(); ++k) {
        a.coef[3 * k + 0] = tri->prog.steps[k].A;
        a.coef[3 * k + 1] = tri->prog.steps[k].B;
        a.coef[3 * k + 2] = tri->prog.steps[k].C;
    }
    a.phi0 = tri->prog.phi0;
    for (int i = 0; i < 4; ++i) a.A0[i] = tri->A0[i];
    a.b0[0] = tri->b0[0];
    a.b0[1] = tri->b0[1];
    a.L = line_desc(line);
    a.nreq = nreq;
    a.npts = npts;
    a.rowsA = rows;
    a.vdimA = tri->vdim;
    int P = std::max(1, 64 / npts);
    while (P > 1 && P * reqsize * 8 > 20 * 1024) --P;  // 20 KB per wave: two four-wave workgroups per CU
    a.P = P;
    a.nitems = (nreq + P - 1) / P;
    a.stage_doubles = (int)((P * reqsize + 1) & ~1LL);
    const size_t lds = (size_t)a.stage_doubles * 8 * 4;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)ctx->lds_per_cu / std::max<size_t>(lds, 1)));
    const int grid = (int)std::max<long long>(1, std::min<long long>((a.nitems + 3) / 4, (long long)ctx->num_cu * per_cu * 2));
    bool ok = false;
    if (tri->n == 1) ok = launch_prism_small_nn<1>(line->nn, order, a, grid, lds, (hipStream_t)stream);
    if (tri->n == 2) ok = launch_prism_small_nn<2>(line->nn, order, a, grid, lds, (hipStream_t)stream);
    if (tri->n == 3) ok = launch_prism_small_nn<3>(line->nn, order, a, grid, lds, (hipStream_t)stream);
    if (!ok) return fail(FX_ENOTIMPL, "fx_prism_tabulate_batch: no instance for degree %d x %d nodes, order %d", tri->n, line->nn, order);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

int fx_tensor_tabulate_grid_batch(fx_ctx* ctx, int nf, const fx_line_element* const* factors, int order, int64_t nreq,
                                  int q, const double* grid, double* out, void* stream) {
    if (q < 0) return fail(FX_EINVAL, "negative grid size");
    long long npts = 1;
    for (int f = 0; f < nf; ++f) npts *= q;
    return tensor_launch(ctx, nf, factors, order, nreq, (int)npts, q, grid, out, stream, true);
}

}  // extern "C"

namespace {
// Solve M Y = B in place (B: n x nrhs, row-major) by Gaussian elimination with partial pivoting.
bool solve_dense(int n, std::vector<double> M, int nrhs, std::vector<double>& B) {
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(M[(size_t)r * n + c]) > std::fabs(M[(size_t)piv * n + c])) piv = r;
        if (M[(size_t)piv * n + c] == 0.0) return false;
        if (piv != c) {
            for (int k = 0; k < n; ++k) std::swap(M[(size_t)c * n + k], M[(size_t)piv * n + k]);
            for (int k = 0; k < nrhs; ++k) std::swap(B[(size_t)c * nrhs + k], B[(size_t)piv * nrhs + k]);
        }
        const double d = 1.0 / M[(size_t)c * n + c];
        for (int r = c + 1; r < n; ++r) {
            const double f = M[(size_t)r * n + c] * d;
            if (f == 0.0) continue;
            for (int k = c + 1; k < n; ++k) M[(size_t)r * n + k] -= f * M[(size_t)c * n + k];
            for (int k = 0; k < nrhs; ++k) B[(size_t)r * nrhs + k] -= f * B[(size_t)c * nrhs + k];
        }
    }
    for (int c = n - 1; c >= 0; --c) {
        const double d = 1.0 / M[(size_t)c * n + c];
        for (int k = 0; k < nrhs; ++k) {
            double t = B[(size_t)c * nrhs + k];
            for (int j = c + 1; j < n; ++j) t -= M[(size_t)c * n + j] * B[(size_t)j * nrhs + k];
            B[(size_t)c * nrhs + k] = t * d;
        }
    }
    return true;
}

// A fragments of the stacked matrix [C; C D^alpha ...] of derivative order `order` for the stacked-matrix
// kernel.  The derivative matrices D^alpha (d^alpha phi_j = sum_k D^alpha[j][k] phi_k on the element's cell:
// what FIAT's dmats hold, FIAT/expansions.py:438-446) are obtained by L2 projection: the raw expansion set
// and its derivatives are tabulated ON THE DEVICE (the parity-checked recurrence kernels) at a collapsed
// Gauss-Jacobi rule of n + 1 points per direction (exact for the degree-2n products), D^alpha = R_alpha M^-1
// with the mass matrix M = sum w phi phi^T and R_alpha = sum w (d^alpha phi) phi^T.
// D^alpha of the RAW recurrence basis on the element's own cell, |alpha| = 1..order (order <= 2), by L2 projection:
// the expansion set and its derivatives are tabulated on the device by the parity-checked recurrence kernels at a
// collapsed Gauss-Jacobi rule exact for the degree-2n products, D_t = R_t M^-1.  On return (ok) B[k][(t-1) nexp + j]
// = D_t[j][k], i.e. d^alpha_t phi_j = sum_k D_t[j][k] phi_k.
int project_derivative_matrices(fx_ctx* ctx, fx_element* e, int order, std::vector<double>& B, bool& ok) {
    ok = false;
    const int sd = e->sd, nexp = e->nexp;
    const int ntab = fx::binom(sd + order, sd);
    // vertices of the element's cell: preimages of the default simplex's vertices under x -> A0 x + b0
    double inv[9], verts[12];
    if (!invert_small(sd, e->A0, inv)) return FX_OK;
    for (int v = 0; v <= sd; ++v)
        for (int i = 0; i < sd; ++i) {
            double t = 0.0;
            for (int d = 0; d < sd; ++d) t += inv[i * sd + d] * ((v == d + 1 ? 1.0 : -1.0) - e->b0[d]);
            verts[v * sd + i] = t;
        }
    const int m = e->n + 1;
    int nq = 1;
    for (int d = 0; d < sd; ++d) nq *= m;
    double *d_pts = nullptr, *d_wts = nullptr, *d_tab = nullptr;
    fx_element* raw = new fx_element;
    raw->ctx = ctx;
    raw->sd = sd;
    raw->n = e->n;
    raw->variant = e->variant;
    raw->nexp = nexp;
    raw->scale = e->scale;
    memcpy(raw->A0, e->A0, sizeof raw->A0);
    memcpy(raw->b0, e->b0, sizeof raw->b0);
    raw->prog = e->prog;
    raw->raw_expansion = true;  // (no C0 transform: the projection is onto the raw recurrence basis)
    std::vector<double> tab((size_t)ntab * nexp * nq), wts(nq);
    auto cleanup = [&]() {
        if (d_pts) (void)hipFree(d_pts);
        if (d_wts) (void)hipFree(d_wts);
        if (d_tab) (void)hipFree(d_tab);
        fx_element_destroy(raw);
    };
    int rc = FX_OK;
    hipError_t he = hipMalloc(&raw->d_steps, std::max<size_t>(1, raw->prog.steps.size()) * sizeof(fxk::Step));
    if (he == hipSuccess && !raw->prog.steps.empty())
        he = hipMemcpy(raw->d_steps, raw->prog.steps.data(), raw->prog.steps.size() * sizeof(fxk::Step), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMalloc(&d_pts, (size_t)nq * sd * sizeof(double));
    if (he == hipSuccess) he = hipMalloc(&d_wts, (size_t)nq * sizeof(double));
    if (he == hipSuccess) he = hipMalloc(&d_tab, tab.size() * sizeof(double));
    if (he != hipSuccess) {
        cleanup();
        return fail(FX_EHIP, "stacked matrix: %s", hipGetErrorString(he));
    }
    rc = upload_coeffs(raw, nexp, 1, nullptr);
    if (rc == FX_OK) rc = fx_collapsed_quadrature(ctx, sd, m, verts, d_pts, d_wts, nullptr);
    if (rc == FX_OK) rc = fx_tabulate_batch(ctx, raw, order, 1, nq, d_pts, nullptr, d_tab, nullptr);
    if (rc == FX_OK) {
        he = hipMemcpy(tab.data(), d_tab, tab.size() * sizeof(double), hipMemcpyDeviceToHost);
        if (he == hipSuccess) he = hipMemcpy(wts.data(), d_wts, wts.size() * sizeof(double), hipMemcpyDeviceToHost);
        if (he != hipSuccess) rc = fail(FX_EHIP, "stacked matrix: %s", hipGetErrorString(he));
    }
    cleanup();
    if (rc != FX_OK) return rc;
    // M (symmetric) and the right-hand sides R_alpha^T, side by side
    const int nrhs = (ntab - 1) * nexp;
    std::vector<double> M((size_t)nexp * nexp, 0.0), wphi((size_t)nexp * nq);
    B.assign((size_t)nexp * nrhs, 0.0);
    for (int k = 0; k < nexp; ++k)
        for (int q = 0; q < nq; ++q) wphi[(size_t)k * nq + q] = wts[q] * tab[(size_t)k * nq + q];
    for (int k = 0; k < nexp; ++k)
        for (int l = 0; l < nexp; ++l) {
            double t = 0.0;
            for (int q = 0; q < nq; ++q) t += wphi[(size_t)k * nq + q] * tab[(size_t)l * nq + q];
            M[(size_t)k * nexp + l] = t;
        }
    for (int t = 1; t < ntab; ++t)
        for (int j = 0; j < nexp; ++j) {
            const double* dj = &tab[((size_t)t * nexp + j) * nq];
            for (int k = 0; k < nexp; ++k) {
                double acc = 0.0;
                for (int q = 0; q < nq; ++q) acc += dj[q] * wphi[(size_t)k * nq + q];
                B[(size_t)k * nrhs + (size_t)(t - 1) * nexp + j] = acc;  // R_t^T[k][j]
            }
        }
    ok = solve_dense(nexp, M, nrhs, B);
    return FX_OK;
}

// chain rule across the tables of orders 1..order (3 or 4) for per-request cells (table_mix_high_kernel)
// ... and of any order up to FX_MAX_ORDER (table_mix_any_kernel): orders 5..8
int mix_any_order(fx_ctx* ctx, const fx_element* e, int order, int64_t nreq, int npts, const double* verts, double* out, hipStream_t s) {
    const int sd = e->sd;
    fxk::TableMixAnyArgs ma;
    memset(&ma, 0, sizeof ma);
    ma.out = out;
    ma.verts = verts;
    if (!invert_small(sd, e->A0, ma.A0inv)) return fail(FX_EINVAL, "degenerate cell");
    ma.n = e->ndof * e->vdim * npts;
    ma.slices = std::max(1, std::min(16, (ma.n + 1023) / 1024));
    ma.nreq = nreq;
    ma.order = order;
    if (order > 9 || nreq * ma.slices > 0x7fffffffLL) return fail(FX_EINVAL, "batch too large for the table-mixing pass");
    std::vector<std::vector<int>> prev = fx::multi_indices(sd, 0);
    int t = 1, moff = 0;
    ma.first[0] = 0;
    ma.cnt[0] = 1;
    for (int k = 1; k <= order; ++k) {
        const std::vector<std::vector<int>> cur = fx::multi_indices(sd, k);
        ma.first[k] = t;
        ma.cnt[k] = (int)cur.size();
        ma.moff[k] = moff;
        moff += (int)(cur.size() * cur.size());
        for (const std::vector<int>& al : cur) {
            if (t >= fxk::MIXA_MAXT) return fail(FX_ENOTIMPL, "too many derivative tables for the mixing pass");
            int lead = 0;
            while (al[lead] == 0) ++lead;
            ma.lead[t] = (unsigned char)lead;
            for (int c = 0; c < 3; ++c) {
                ma.down[t][c] = -1;
                if (c < sd && al[c] > 0) {
                    std::vector<int> be = al;
                    be[c] -= 1;
                    ma.down[t][c] = (signed char)(std::find(prev.begin(), prev.end(), be) - prev.begin());
                }
            }
            ++t;
        }
        prev = cur;
    }
    const int lds = (16 + moff) * 8;
    const dim3 grid((unsigned)std::max<long long>(1, nreq * ma.slices));
    if (sd == 1) hipLaunchKernelGGL((fxk::table_mix_any_kernel<1>), grid, dim3(256), lds, s, ma);
    else if (sd == 2) hipLaunchKernelGGL((fxk::table_mix_any_kernel<2>), grid, dim3(256), lds, s, ma);
    else hipLaunchKernelGGL((fxk::table_mix_any_kernel<3>), grid, dim3(256), lds, s, ma);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

int mix_high_order(fx_ctx* ctx, const fx_element* e, int order, int64_t nreq, int npts, const double* verts, double* out, hipStream_t s) {
    if (order > 4) return mix_any_order(ctx, e, order, nreq, npts, verts, out, s);
    const int sd = e->sd;
    fxk::TableMixHighArgs ma;
    memset(&ma, 0, sizeof ma);
    ma.out = out;
    ma.verts = verts;
    if (!invert_small(sd, e->A0, ma.A0inv)) return fail(FX_EINVAL, "degenerate cell");
    ma.n = e->ndof * e->vdim * npts;
    ma.slices = std::max(1, std::min(8, (ma.n + 2047) / 2048));
    ma.nreq = nreq;
    ma.rb = ma.n >= 1024 ? 1 : std::max(1, std::min(fxk::MIXH_RB, 2048 / std::max(1, ma.n)));   // small requests: blocks per workgroup
    if (nreq * ma.slices > 0x7fffffffLL) return fail(FX_EINVAL, "batch too large for the table-mixing pass");
    // where alpha - e_c sits within the previous order (mis() order inside each order)
    std::vector<std::vector<int>> prev = fx::multi_indices(sd, 0);
    int t = 1;
    for (int k = 1; k <= order; ++k) {
        const std::vector<std::vector<int>> cur = fx::multi_indices(sd, k);
        for (const std::vector<int>& al : cur) {
            if (t >= fxk::MIXH_MAXT) return fail(FX_ENOTIMPL, "too many derivative tables for the mixing pass");
            int lead = 0;
            while (al[lead] == 0) ++lead;
            ma.lead[t] = (unsigned char)lead;
            for (int c = 0; c < 3; ++c) {
                ma.down[t][c] = -1;
                if (c < sd && al[c] > 0) {
                    std::vector<int> be = al;
                    be[c] -= 1;
                    ma.down[t][c] = (signed char)(std::find(prev.begin(), prev.end(), be) - prev.begin());
                }
            }
            ++t;
        }
        prev = cur;
    }
    const long long nblk = ma.rb > 1 ? std::min<long long>((nreq + ma.rb - 1) / ma.rb, (long long)ctx->num_cu * 16) : nreq * ma.slices;
    const dim3 grid((unsigned)std::max<long long>(1, nblk));
    if (sd == 1 && order == 3) hipLaunchKernelGGL((fxk::table_mix_high_kernel<1, 3>), grid, dim3(256), 0, s, ma);
    else if (sd == 1) hipLaunchKernelGGL((fxk::table_mix_high_kernel<1, 4>), grid, dim3(256), 0, s, ma);
    else if (sd == 2 && order == 3) hipLaunchKernelGGL((fxk::table_mix_high_kernel<2, 3>), grid, dim3(256), 0, s, ma);
    else if (sd == 2) hipLaunchKernelGGL((fxk::table_mix_high_kernel<2, 4>), grid, dim3(256), 0, s, ma);
    else if (order == 3) hipLaunchKernelGGL((fxk::table_mix_high_kernel<3, 3>), grid, dim3(256), 0, s, ma);
    else hipLaunchKernelGGL((fxk::table_mix_high_kernel<3, 4>), grid, dim3(256), 0, s, ma);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

// Derivative orders > 2 (FIAT/expansions.py:66-137 carries the Leibniz rule to any order; above its recurrence order the
// reference itself switches to differentiation matrices, :438-446, 577-599).  On the element's own cell
// d^alpha phi = D^alpha phi with constant matrices, D^(beta + e_d) = D^beta D^(e_d): the first-order matrices come from
// the parity-checked order-1 recurrence tables (project_derivative_matrices), the stacked matrix [C D^alpha] over all
// |alpha| <= order becomes the coefficient matrix of an internal element, and one order-0 tabulation of that element
// writes every table in place -- [ntab][rows][npts] IS [ntab * rows][npts].
int ensure_high_order(fx_ctx* ctx, fx_element* e, int order) {
    static std::mutex build_mutex;
    std::lock_guard<std::mutex> lock(build_mutex);
    if (e->high_state[order] == 1) return FX_OK;
    // -1 is recorded for STRUCTURAL failures only (a singular mass matrix: retrying cannot help); transient ones (hipMalloc,
    // a failed copy) leave the state at 0 so that the next call builds again.  The first order > 2 call per element runs
    // null-stream launches and synchronous copies in project_derivative_matrices: it synchronises the device once.
    if (e->high_state[order] < 0) return fail(FX_ENOTIMPL, "derivative order %d: differentiation matrices could not be built (singular mass matrix of the expansion set)", order);
    const int sd = e->sd, nexp = e->nexp;
    const int rows = (int)(e->hC.size() / (size_t)nexp);
    std::vector<double> D1((size_t)sd * nexp * nexp, 0.0);   // D1[d][j][k]
    if (e->n >= 1) {
        std::vector<double> B;
        bool ok = false;
        int rc = project_derivative_matrices(ctx, e, 1, B, ok);
        if (rc != FX_OK) return rc;      // transient (HIP / allocation): state stays 0, the next call retries
        if (!ok) {
            e->high_state[order] = -1;
            return fail(FX_ENOTIMPL, "derivative order %d: singular mass matrix of the expansion set", order);
        }
        const int nrhs = sd * nexp;
        // members are graded by total degree (Morton order: degree s starts at C(s - 1 + sd, sd)) and differentiation
        // lowers the degree: entries with deg(k) >= deg(j) are structural zeros -- whatever the projection left there
        // is round-off, and dropping it makes D^alpha vanish exactly for |alpha| > degree
        std::vector<int> deg(nexp, 0);
        for (int j = 0, s = 0; j < nexp; ++j) {
            while (fx::binom(s + sd, sd) <= j) ++s;
            deg[j] = s;
        }
        for (int d = 0; d < sd; ++d)
            for (int j = 0; j < nexp; ++j)
                for (int k = 0; k < nexp; ++k)
                    D1[((size_t)d * nexp + j) * nexp + k] = deg[k] < deg[j] ? B[(size_t)k * nrhs + (size_t)d * nexp + j] : 0.0;
    }
    // all multi-indices |alpha| <= order in mis() order, each D^alpha from its predecessor
    std::vector<std::vector<int>> alphas;
    for (int k = 0; k <= order; ++k) {
        std::vector<std::vector<int>> level = fx::multi_indices(sd, k);
        alphas.insert(alphas.end(), level.begin(), level.end());
    }
    const int ntab = (int)alphas.size();
    std::vector<std::vector<double>> Dm(ntab);
    std::vector<double> S((size_t)ntab * rows * nexp, 0.0);
    for (int t = 0; t < ntab; ++t) {
        std::vector<double>& D = Dm[t];
        D.assign((size_t)nexp * nexp, 0.0);
        int d = 0;
        while (d < sd && alphas[t][d] == 0) ++d;
        if (d == sd) {
            for (int i = 0; i < nexp; ++i) D[(size_t)i * nexp + i] = 1.0;
        } else {
            std::vector<int> beta = alphas[t];
            beta[d] -= 1;
            int tb = 0;
            while (alphas[tb] != beta) ++tb;
            const std::vector<double>& P = Dm[tb];
            const double* E = &D1[(size_t)d * nexp * nexp];
            for (int i = 0; i < nexp; ++i)
                for (int l = 0; l < nexp; ++l) {
                    const double pil = P[(size_t)i * nexp + l];
                    if (pil == 0.0) continue;
                    for (int k = 0; k < nexp; ++k) D[(size_t)i * nexp + k] += pil * E[(size_t)l * nexp + k];
                }
        }
        for (int r = 0; r < rows; ++r) {
            double* srow = &S[((size_t)t * rows + r) * nexp];
            for (int j = 0; j < nexp; ++j) {
                const double c = e->hC[(size_t)r * nexp + j];
                if (c == 0.0) continue;
                for (int k = 0; k < nexp; ++k) srow[k] += c * D[(size_t)j * nexp + k];
            }
        }
    }
    // the internal element: same cell, same recurrence, no C0 transform (hC already carries it), rows = all tables
    fx_element* h = new fx_element;
    h->ctx = ctx;
    h->sd = sd;
    h->n = e->n;
    h->variant = e->variant;
    h->nexp = nexp;
    h->scale = e->scale;
    memcpy(h->A0, e->A0, sizeof h->A0);
    memcpy(h->b0, e->b0, sizeof h->b0);
    h->prog = e->prog;
    hipError_t he = hipMalloc(&h->d_steps, std::max<size_t>(1, h->prog.steps.size()) * sizeof(fxk::Step));
    if (he == hipSuccess && !h->prog.steps.empty())
        he = hipMemcpy(h->d_steps, h->prog.steps.data(), h->prog.steps.size() * sizeof(fxk::Step), hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        fx_element_destroy(h);
        return fail(FX_EHIP, "derivative order %d: %s", order, hipGetErrorString(he));
    }
    int rc = upload_coeffs(h, ntab * rows, 1, S.data());
    if (rc != FX_OK) {
        fx_element_destroy(h);
        return rc;
    }
    e->high[order] = h;
    e->high_state[order] = 1;
    return FX_OK;
}

int ensure_stacked(fx_ctx* ctx, fx_element* e, int order) {
    if (order < 0 || order > 2) return FX_OK;
    // built lazily from plan_launch, which may run on several host threads for the same element
    static std::mutex build_mutex;
    std::lock_guard<std::mutex> lock(build_mutex);
    if (e->stack_state[order] != 0) return FX_OK;
    // -1 ("other kernels serve the element") is recorded for the STRUCTURAL failure only, a singular mass matrix; a transient one
    // (allocation, copy) returns its error and leaves the state at 0, so that the next call builds again (as ensure_high_order)
    const int sd = e->sd, nexp = e->nexp;
    const int rows = (int)(e->hC.size() / (size_t)nexp);
    const int ntab = fx::binom(sd + order, sd);
    const long long R = (long long)ntab * rows;
    std::vector<double> S((size_t)R * nexp, 0.0);  // stacked matrix
    std::copy(e->hC.begin(), e->hC.end(), S.begin());
    if (ntab > 1) {
        std::vector<double> B;
        bool ok = false;
        int rc = project_derivative_matrices(ctx, e, order, B, ok);
        if (rc != FX_OK) return rc;
        if (!ok) {
            e->stack_state[order] = -1;
            return FX_OK;  // other kernels serve the element
        }
        const int nrhs = (ntab - 1) * nexp;
        // B[k][(t-1) nexp + j] = D_t[j][k];  rows of table t: C D_t
        for (int t = 1; t < ntab; ++t)
            for (int r = 0; r < rows; ++r) {
                double* srow = &S[((size_t)t * rows + r) * nexp];
                for (int j = 0; j < nexp; ++j) {
                    const double c = e->hC[(size_t)r * nexp + j];
                    if (c == 0.0) continue;
                    for (int k = 0; k < nexp; ++k) srow[k] += c * B[(size_t)k * nrhs + (size_t)(t - 1) * nexp + j];
                }
            }
    }
    // 16x16x4 A fragments, K in production order (slot 0 = member 0, slot s = destination of step s-1)
    const int KS = (nexp + 3) / 4, RT = (int)((R + 15) / 16);
    std::vector<int> member(4 * KS, -1);
    member[0] = 0;
    for (size_t i = 0; i < e->prog.steps.size() && (int)i + 1 < nexp; ++i) member[i + 1] = e->prog.steps[i].dst;
    std::vector<double> F((size_t)(RT + 1) * KS * 64, 0.0);
    for (int rt = 0; rt < RT; ++rt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const long long row = 16LL * rt + (lane & 15);
                const int mem = member[4 * ks + (lane >> 4)];
                if (row < R && mem >= 0) F[((size_t)rt * KS + ks) * 64 + lane] = S[(size_t)row * nexp + mem];
            }
    double* d_f = nullptr;
    HIP_TRY(hipMalloc(&d_f, F.size() * sizeof(double)));
    if (hipMemcpy(d_f, F.data(), F.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d_f);
        (void)hipGetLastError();  // (the failed call left HIP's sticky error set: the next launch check must not report it)
        return fail(FX_EHIP, "ensure_stacked: copy of the stacked matrix failed");
    }
    if (order >= 1) {  // dof-major tiles for the instances that apply the chain rule across the tables themselves (per-request cells)
        const int RTd = (rows + 15) / 16;
        std::vector<double> Fd((size_t)(RTd * ntab + 1) * KS * 64, 0.0);
        for (int i = 0; i < RTd; ++i)
            for (int t = 0; t < ntab; ++t)
                for (int ks = 0; ks < KS; ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int r = 16 * i + (lane & 15);
                        const int mem = member[4 * ks + (lane >> 4)];
                        if (r < rows && mem >= 0)
                            Fd[(((size_t)i * ntab + t) * KS + ks) * 64 + lane] = S[((size_t)t * rows + r) * nexp + mem];
                    }
        double* d_fd = nullptr;
        if (hipMalloc(&d_fd, Fd.size() * sizeof(double)) != hipSuccess ||
            hipMemcpy(d_fd, Fd.data(), Fd.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            if (d_fd) (void)hipFree(d_fd);
            (void)hipFree(d_f);
            (void)hipGetLastError();  // (the failed call left HIP's sticky error set: the next launch check must not report it)
            return fail(FX_EHIP, "ensure_stacked: allocation or copy of the dof-major stacked matrix failed");
        }
        e->d_astack_dm[order] = d_fd;
    }
    if (e->vdim == sd && sd >= 2) {  // vector-valued: dof-major tiles with the components of a dof in one lane (fused Piola map)
        const int TR = fxk::stacked_tile_rows(sd, 1);
        const int RTd = (rows + TR - 1) / TR;
        std::vector<double> Fp((size_t)(RTd * ntab + 1) * KS * 64, 0.0);
        for (int i = 0; i < RTd; ++i)
            for (int t = 0; t < ntab; ++t)
                for (int ks = 0; ks < KS; ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int pos = lane & 15;  // tile position 4 jj + kk
                        const int tr = fxk::stacked_pio_row(sd, pos >> 2, pos & 3);
                        const int mem = member[4 * ks + (lane >> 4)];
                        if (tr >= 0 && TR * i + tr < rows && mem >= 0)
                            Fp[(((size_t)i * ntab + t) * KS + ks) * 64 + lane] = S[((size_t)t * rows + TR * i + tr) * nexp + mem];
                    }
        double* d_fp = nullptr;
        if (hipMalloc(&d_fp, Fp.size() * sizeof(double)) != hipSuccess ||
            hipMemcpy(d_fp, Fp.data(), Fp.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            if (d_fp) (void)hipFree(d_fp);
            if (e->d_astack_dm[order]) (void)hipFree(e->d_astack_dm[order]);
            e->d_astack_dm[order] = nullptr;
            (void)hipFree(d_f);
            (void)hipGetLastError();  // (the failed call left HIP's sticky error set: the next launch check must not report it)
            return fail(FX_EHIP, "ensure_stacked: allocation or copy of the component-major stacked matrix failed");
        }
        e->d_astack_dmp[order] = d_fp;
    }
    e->d_astack[order] = d_f;
    e->stack_state[order] = 1;
    return FX_OK;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// Macro elements: the element's cell is a simplicial complex (FIAT/expansions.py:449-490).
// One MACRO instance of the generic kernel; the members of sub-cell c are the K rows
// [c*nexp, (c+1)*nexp) of the contraction, A = [C[:, map[0]] T s_0 | C[:, map[1]] T s_1 | ...].
// ---------------------------------------------------------------------------------------------
struct fx_macro_element {
    fx_ctx* ctx = nullptr;
    int sd = 0, n = 0, variant = 0, nexp = 0, ncell = 0, nmacro = 0, ndof = 0, vdim = 1;
    double A0[9] = {0}, b0[3] = {0};
    fx::Program prog;
    std::vector<double> T;       // C0 transform of one sub-cell (bubble) or empty
    std::vector<int> map;        // [ncell][nexp]
    std::vector<double> cscale;  // [ncell]
    int KS = 0, MT = 0;
    fxk::Step* d_steps = nullptr;
    double* d_cells = nullptr;
    double* d_afrag = nullptr;
    double* d_cmat = nullptr;  // [ncell][rows][nexp] blocks of the lane-local kernel (macro_small.hpp)
};

namespace {

// rescaled barycentric coordinates of the simplex whose map to (-1,1)^sd is (Ac, bc), as functions of the
// parent's (-1,1)^sd coordinates xi (x = A0inv (xi - b0)): rows L[i*3+d], offsets l[i]
void barycentric_rows(int sd, const double* Ac, const double* bc, const double* A0inv, const double* b0, double* L,
                      double* l) {
    double g[4][3] = {{0}}, g0[4] = {0};
    g0[0] = 1.0;
    for (int i = 1; i <= sd; ++i) {  // lambda_i = (X_{i-1} + 1) / 2, lambda_0 = 1 - sum
        for (int d = 0; d < sd; ++d) g[i][d] = 0.5 * Ac[(i - 1) * sd + d];
        g0[i] = 0.5 * (bc[i - 1] + 1.0);
        for (int d = 0; d < sd; ++d) g[0][d] -= g[i][d];
        g0[0] -= g0[i];
    }
    for (int i = 0; i <= sd; ++i) {
        double nrm = 0.0;
        for (int d = 0; d < sd; ++d) nrm += g[i][d] * g[i][d];
        const double h = 1.0 / std::sqrt(nrm);  // height over the facet (reference_element.py:638-642)
        double shift = 0.0;
        for (int d = 0; d < sd; ++d) {
            double t = 0.0;
            for (int e = 0; e < sd; ++e) t += g[i][e] * A0inv[e * sd + d];
            L[i * 3 + d] = h * t;
            shift += t * b0[d];
        }
        l[i] = h * (g0[i] - shift);
    }
}

int macro_upload_coeffs(fx_macro_element* e, int ndof, int vdim, const double* coeffs) {
    const int rows = ndof * vdim, nexp = e->nexp, K = e->ncell * nexp;
    if (!coeffs && rows != e->nmacro) return fail(FX_EINVAL, "identity coefficients need ndof*vdim == nmacro");
    std::vector<double> S((size_t)rows * K, 0.0);
    for (int i = 0; i < rows; ++i)
        for (int c = 0; c < e->ncell; ++c)
            for (int j = 0; j < nexp; ++j) {
                const int m = e->map[(size_t)c * nexp + j];
                const double cij = (coeffs ? coeffs[(size_t)i * e->nmacro + m] : (i == m ? 1.0 : 0.0)) * e->cscale[c];
                if (cij == 0.0) continue;
                double* dst = &S[(size_t)i * K + (size_t)c * nexp];
                if (e->T.empty()) {
                    dst[j] += cij;
                } else {  // member j of the C0 set = sum_k T[j][k] raw_k
                    const double* trow = &e->T[(size_t)j * nexp];
                    for (int k = 0; k < nexp; ++k) dst[k] += cij * trow[k];
                }
            }
    std::vector<double> F = fx::pack_a_fragments(S, rows, K);
    if (e->d_afrag) {
        (void)hipFree(e->d_afrag);
        e->d_afrag = nullptr;
    }
    HIP_TRY(hipMalloc(&e->d_afrag, F.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(e->d_afrag, F.data(), F.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> CM((size_t)e->ncell * rows * nexp);
    for (int c = 0; c < e->ncell; ++c)
        for (int i = 0; i < rows; ++i)
            for (int k = 0; k < nexp; ++k) CM[((size_t)c * rows + i) * nexp + k] = S[(size_t)i * K + (size_t)c * nexp + k];
    if (e->d_cmat) {
        (void)hipFree(e->d_cmat);
        e->d_cmat = nullptr;
    }
    HIP_TRY(hipMalloc(&e->d_cmat, CM.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(e->d_cmat, CM.data(), CM.size() * sizeof(double), hipMemcpyHostToDevice));
    e->ndof = ndof;
    e->vdim = vdim;
    e->MT = (rows + 15) / 16;
    e->KS = (K + 3) / 4;
    return FX_OK;
}

template <int SD, int ORDER>
int launch_macro_one(const fxk::TabArgs& a, int grid, int lds_bytes, hipStream_t s) {
    auto kern = fxk::tabulate_simplex_kernel<SD, ORDER, 1, 0, 0, true>;
    if (lds_bytes > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds_bytes, s, a);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

template <int SD>
int launch_macro_sd(int order, const fxk::TabArgs& a, int grid, int lds_bytes, hipStream_t s) {
    switch (order) {
        case 0: return launch_macro_one<SD, 0>(a, grid, lds_bytes, s);
        case 1: return launch_macro_one<SD, 1>(a, grid, lds_bytes, s);
        case 2: return launch_macro_one<SD, 2>(a, grid, lds_bytes, s);
    }
    return fail(FX_ENOTIMPL, "derivative order %d > 2 is not implemented on the device", order);
}

// ---- lane-local kernel for low-order macro elements (macro_small.hpp): (sd, n, highest order) ----
struct MacroSmallShape {
    int sd, n, max_order;
};
const MacroSmallShape kMacroSmallShapes[] = {{2, 1, 2}, {2, 2, 2}, {2, 3, 2}, {3, 1, 2}, {3, 2, 2}, {3, 3, 1}};
constexpr int MACRO_SMALL_NW = 4;

template <int SD, int N, int ORDER>
int launch_macro_small_one(const fxk::MacroSmallArgs& a, int grid, int lds_bytes, hipStream_t s) {
    auto kern = fxk::tabulate_macro_small<SD, N, ORDER, MACRO_SMALL_NW>;
    if (lds_bytes > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * MACRO_SMALL_NW), lds_bytes, s, a);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

template <int SD, int N, int MAXORDER>
int launch_macro_small(int order, const fxk::MacroSmallArgs& a, int grid, int lds_bytes, hipStream_t s) {
    if (order == 0) return launch_macro_small_one<SD, N, 0>(a, grid, lds_bytes, s);
    if (order == 1) return launch_macro_small_one<SD, N, 1>(a, grid, lds_bytes, s);
    if constexpr (MAXORDER >= 2) {
        if (order == 2) return launch_macro_small_one<SD, N, 2>(a, grid, lds_bytes, s);
    }
    return fail(FX_EINVAL, "internal: order %d not instantiated for the lane-local macro kernel", order);
}

int run_macro_small(int id, int order, const fxk::MacroSmallArgs& a, int grid, int lds_bytes, hipStream_t s) {
    switch (id) {
        case 0: return launch_macro_small<2, 1, 2>(order, a, grid, lds_bytes, s);
        case 1: return launch_macro_small<2, 2, 2>(order, a, grid, lds_bytes, s);
        case 2: return launch_macro_small<2, 3, 2>(order, a, grid, lds_bytes, s);
        case 3: return launch_macro_small<3, 1, 2>(order, a, grid, lds_bytes, s);
        case 4: return launch_macro_small<3, 2, 2>(order, a, grid, lds_bytes, s);
        case 5: return launch_macro_small<3, 3, 1>(order, a, grid, lds_bytes, s);
    }
    return fail(FX_EINVAL, "internal: unknown lane-local macro kernel %d", id);
}

bool macro_small_table_matches(int id, const fx::Program& P) {
    switch (id) {
        case 0: return table_matches<2, 1>(P);
        case 1: return table_matches<2, 2>(P);
        case 2: return table_matches<2, 3>(P);
        case 3: return table_matches<3, 1>(P);
        case 4: return table_matches<3, 2>(P);
        case 5: return table_matches<3, 3>(P);
    }
    return false;
}

}  // namespace

extern "C" {

int fx_macro_element_destroy(fx_macro_element* e) {
    if (!e) return FX_OK;
    if (e->d_steps) (void)hipFree(e->d_steps);
    if (e->d_cells) (void)hipFree(e->d_cells);
    if (e->d_afrag) (void)hipFree(e->d_afrag);
    if (e->d_cmat) (void)hipFree(e->d_cmat);
    delete e;
    return FX_OK;
}

int fx_macro_element_create(fx_ctx* ctx, int sd, int n, int variant, double scale, const double* parent_verts, int ncell,
                            const double* cell_verts, int nmacro, const int* cell_node_map, const double* cell_scale,
                            int ndof, int vdim, const double* coeffs, fx_macro_element** out) {
    if (!ctx || !out || !cell_verts || !cell_node_map) return fail(FX_EINVAL, "fx_macro_element_create: null argument");
    if (sd < 1 || sd > 3) return fail(FX_EINVAL, "Invalid number of spatial dimensions");
    if (n < 0) return fail(FX_EINVAL, "negative degree");
    if (variant < 0 || variant > 2) return fail(FX_EINVAL, "Invalid variant %d", variant);
    if (variant == FX_VARIANT_BUBBLE && n < 1) return fail(FX_EINVAL, "bubble variant needs degree >= 1");
    if (ncell < 1 || ncell > 32) return fail(FX_EINVAL, "a macro cell needs 1..32 sub-cells, got %d", ncell);
    if (ndof < 1 || vdim < 1 || nmacro < 1) return fail(FX_EINVAL, "bad ndof/vdim/nmacro");
    if (!(scale > 0.0)) return fail(FX_EINVAL, "a macro expansion set needs an explicit positive scale");
    HIP_TRY(hipSetDevice(ctx->device));
    fx_macro_element* e = new fx_macro_element;
    e->ctx = ctx;
    e->sd = sd;
    e->n = n;
    e->variant = variant;
    e->nexp = fx::binom(n + sd, sd);
    e->ncell = ncell;
    e->nmacro = nmacro;
    e->map.assign(cell_node_map, cell_node_map + (size_t)ncell * e->nexp);
    for (int m : e->map)
        if (m < 0 || m >= nmacro) {
            delete e;
            return fail(FX_EINVAL, "cell_node_map entry %d outside [0, %d)", m, nmacro);
        }
    e->cscale.assign(ncell, 1.0);
    if (cell_scale) e->cscale.assign(cell_scale, cell_scale + ncell);
    if (!host_cell_map(sd, parent_verts ? parent_verts : UFC[sd - 1], e->A0, e->b0)) {
        delete e;
        return fail(FX_EINVAL, "degenerate cell");
    }
    double A0inv[9];
    invert_small(sd, e->A0, A0inv);
    std::vector<double> cells(16 + (size_t)ncell * 28, 0.0);
    barycentric_rows(sd, e->A0, e->b0, A0inv, e->b0, &cells[0], &cells[12]);
    for (int c = 0; c < ncell; ++c) {
        double Ac[9], bc[3];
        if (!host_cell_map(sd, cell_verts + (size_t)c * (sd + 1) * sd, Ac, bc)) {
            delete e;
            return fail(FX_EINVAL, "degenerate sub-cell %d", c);
        }
        double* cd = &cells[16 + (size_t)c * 28];
        for (int i = 0; i < sd; ++i) {  // X_c = Ac A0inv (xi - b0) + bc
            double shift = 0.0;
            for (int d = 0; d < sd; ++d) {
                double t = 0.0;
                for (int k = 0; k < sd; ++k) t += Ac[i * sd + k] * A0inv[k * sd + d];
                cd[i * 3 + d] = t;
                shift += t * e->b0[d];
            }
            cd[9 + i] = bc[i] - shift;
        }
        barycentric_rows(sd, Ac, bc, A0inv, e->b0, cd + 12, cd + 24);
    }
    e->prog = fx::build_program(sd, n, variant, scale);
    if (variant == FX_VARIANT_BUBBLE) e->T = fx::c0_transform(sd, n);
    size_t sbytes = std::max<size_t>(1, e->prog.steps.size()) * sizeof(fxk::Step);
    hipError_t he = hipMalloc(&e->d_steps, sbytes);
    if (he == hipSuccess && !e->prog.steps.empty())
        he = hipMemcpy(e->d_steps, e->prog.steps.data(), e->prog.steps.size() * sizeof(fxk::Step), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMalloc(&e->d_cells, cells.size() * sizeof(double));
    if (he == hipSuccess) he = hipMemcpy(e->d_cells, cells.data(), cells.size() * sizeof(double), hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        fx_macro_element_destroy(e);
        return fail(FX_EHIP, "fx_macro_element_create: %s", hipGetErrorString(he));
    }
    int rc = macro_upload_coeffs(e, ndof, vdim, coeffs);
    if (rc != FX_OK) {
        fx_macro_element_destroy(e);
        return rc;
    }
    *out = e;
    return FX_OK;
}

int fx_macro_element_set_coeffs(fx_macro_element* e, int ndof, int vdim, const double* coeffs) {
    if (!e || ndof < 1 || vdim < 1) return fail(FX_EINVAL, "fx_macro_element_set_coeffs: bad argument");
    HIP_TRY(hipSetDevice(e->ctx->device));
    return macro_upload_coeffs(e, ndof, vdim, coeffs);
}

int fx_macro_tabulate_batch(fx_ctx* ctx, const fx_macro_element* e, int order, int64_t nreq, int npts, const double* pts,
                            const double* verts, double* out, void* stream) {
    if (!ctx || !e) return fail(FX_EINVAL, "null context/element");
    if (order < 0) return fail(FX_EINVAL, "negative derivative order");
    if (order > 2) return fail(FX_ENOTIMPL, "derivative order %d > 2 is not implemented on the device", order);
    if (nreq < 0 || npts < 0) return fail(FX_EINVAL, "negative batch size");
    if (nreq == 0 || npts == 0) return FX_OK;
    if (!pts || !out) return fail(FX_EINVAL, "null device pointer");
    const int ntab = fx::binom(e->sd + order, e->sd);
    const int rows = e->ndof * e->vdim;
    HIP_TRY(hipSetDevice(ctx->device));
    {   // lane-local kernel for the registered low-order shapes
        const bool nosmall = (ctx->policy & FX_POLICY_NO_MACRO_SMALL) != 0;
        const long long cmat_doubles = ((long long)e->ncell * rows * e->nexp + 1) & ~1LL;
        for (size_t i = 0; !nosmall && i < sizeof(kMacroSmallShapes) / sizeof(kMacroSmallShapes[0]); ++i) {
            const MacroSmallShape& m = kMacroSmallShapes[i];
            if (m.sd != e->sd || m.n != e->n || order > m.max_order || npts > 64) continue;
            if (!macro_small_table_matches((int)i, e->prog)) continue;
            const int P = std::max(1, 64 / npts);
            // rows per image round: <= 16 KB per wave, an even number of doubles per run where possible
            const long long row_bytes = (long long)P * ntab * npts * 8;
            int RC = (int)std::max<long long>(1, std::min<long long>(rows, 16 * 1024 / row_bytes));
            if (RC < rows) {  // rounds of equal size
                const int rounds = (rows + RC - 1) / RC;
                RC = (rows + rounds - 1) / rounds;
                if (((RC * npts) & 1) && (long long)(RC + 1) * row_bytes <= 16 * 1024) ++RC;
                if (RC > 1 && ((RC * npts) & 1)) --RC;
            }
            const int table = rows * npts;
            const int vec2 = ((table & 1) == 0 && (RC >= rows || ((RC * npts) & 1) == 0)) ? 1 : 0;
            const int Ls = (RC * npts + 1) & ~1;
            const long long stage_doubles = (long long)P * ntab * Ls;
            const long long lds_bytes = (cmat_doubles + stage_doubles * MACRO_SMALL_NW) * 8;
            if (lds_bytes > ctx->lds_per_cu - 1024 || lds_bytes > 150 * 1024) continue;
            fxk::MacroSmallArgs sa;
            memset(&sa, 0, sizeof sa);
            sa.pts = pts;
            sa.verts = verts;
            sa.out = out;
            sa.cmat = e->d_cmat;
            sa.cells = e->d_cells;
            for (size_t k = 0; k < e->prog.steps.size(); ++k) {
                sa.coef[3 * k + 0] = e->prog.steps[k].A;
                sa.coef[3 * k + 1] = e->prog.steps[k].B;
                sa.coef[3 * k + 2] = e->prog.steps[k].C;
            }
            sa.phi0 = e->prog.phi0;
            memcpy(sa.A0, e->A0, sizeof sa.A0);
            memcpy(sa.b0, e->b0, sizeof sa.b0);
            sa.nreq = nreq;
            sa.nitems = (nreq + P - 1) / P;
            sa.npts = npts;
            sa.rows = rows;
            sa.ncell = e->ncell;
            sa.unique = (e->variant == FX_VARIANT_BUBBLE && order == 0) ? 1 : 0;
            sa.P = P;
            sa.RC = RC;
            sa.Ls = Ls;
            sa.vec2 = vec2;
            sa.stage_doubles = (int)stage_doubles;
            sa.cmat_doubles = (int)cmat_doubles;
            if (const char* dbg = ab_env("FIAT_AMD_DEBUG")) sa.debug = atoi(dbg);  // ablation switches (measurement only)
            const int wg_per_cu = std::max(1, std::min(8, ctx->lds_per_cu / (int)lds_bytes));
            const long long nwg = (sa.nitems + MACRO_SMALL_NW - 1) / MACRO_SMALL_NW;
            const int grid = (int)std::max<long long>(1, std::min<long long>(nwg, (long long)ctx->num_cu * wg_per_cu * 2));
            return run_macro_small((int)i, order, sa, grid, (int)lds_bytes, (hipStream_t)stream);
        }
    }
    fxk::TabArgs a;
    memset(&a, 0, sizeof a);
    a.pts = pts;
    a.verts = verts;
    a.out = out;
    a.afrag = e->d_afrag;
    a.steps = e->d_steps;
    a.phi0 = e->prog.phi0;
    memcpy(a.A0, e->A0, sizeof a.A0);
    memcpy(a.b0, e->b0, sizeof a.b0);
    a.nreq = nreq;
    a.npts = npts;
    a.rows = rows;
    a.nexp = e->nexp;
    a.nsteps = (int)e->prog.steps.size();
    a.KS = e->KS;
    a.MT = e->MT;
    a.ntab = ntab;
    a.cells = e->d_cells;
    a.ncell = e->ncell;
    a.unique = (e->variant == FX_VARIANT_BUBBLE && order == 0) ? 1 : 0;
    // work-item shape under a per-wave LDS budget (as for the generic kernel, with the K rows of all sub-cells)
    const long long budget = 40 * 1024, hard = 64 * 1024;
    auto phi_bytes = [&](long long cols) { return ((cols + 15) / 16) * (long long)e->KS * 64 * 8; };
    const long long reqbytes = (long long)ntab * rows * npts * 8;
    int P = 1, pc = npts, nchunk = 1;
    long long stage = 0;
    if (npts <= 64 && phi_bytes((long long)ntab * npts) + reqbytes <= budget) {
        const int pmax = 64 / npts;
        for (int p = 2; p <= pmax; ++p)
            if (phi_bytes((long long)p * ntab * npts) + p * reqbytes <= budget) P = p;
        stage = P * reqbytes;
    } else {
        pc = std::min(npts, 64);
        while (pc > 1 && phi_bytes((long long)ntab * pc) > budget) --pc;
        if (phi_bytes((long long)ntab * pc) > hard)
            return fail(FX_ENOTIMPL, "macro expansion set too large for the LDS tile (ncell=%d, nexp=%d, ntab=%d)", e->ncell,
                        e->nexp, ntab);
        nchunk = (npts + pc - 1) / pc;
    }
    a.P = P;
    a.pc = pc;
    a.nchunk = nchunk;
    a.nitems = nchunk > 1 ? nreq * nchunk : (nreq + P - 1) / P;
    a.phi_doubles = (int)(phi_bytes((long long)P * ntab * pc) / 8);
    a.stage_doubles = stage ? (int)((stage / 8 + 1) & ~1LL) : 0;
    const int lds_bytes = (a.phi_doubles + a.stage_doubles) * 8;
    const int per_cu = std::max(1, std::min(16, ctx->lds_per_cu / std::max(1, lds_bytes)));
    const int grid = (int)std::max<long long>(1, std::min<long long>(a.nitems, (long long)ctx->num_cu * per_cu * 4));
    HIP_TRY(hipSetDevice(ctx->device));
    switch (e->sd) {
        case 1: return launch_macro_sd<1>(order, a, grid, lds_bytes, (hipStream_t)stream);
        case 2: return launch_macro_sd<2>(order, a, grid, lds_bytes, (hipStream_t)stream);
        case 3: return launch_macro_sd<3>(order, a, grid, lds_bytes, (hipStream_t)stream);
    }
    return fail(FX_EINVAL, "Invalid number of spatial dimensions");
}

// ---------------------------------------------------------------------------------
// multi-GPU: RCCL over xGMI behind the C ABI (comm.hpp; SURVEY.md 8b fx_allgather_tables, 8e)
#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) return fail(FX_EHIP, "%s: %s", #expr, fxcomm::api().GetErrorString(r_)); \
    } while (0)

int fx_comm_available(void) {
    const fxcomm::Api& a = fxcomm::api();
    if (!a.handle) return fail(FX_EHIP, "RCCL is not available: %s", a.why);
    return FX_OK;
}

int fx_comm_unique_id(unsigned char* id) {
    if (!id) return fail(FX_EINVAL, "fx_comm_unique_id: null buffer");
    if (fx_comm_available() != FX_OK) return FX_EHIP;
    static_assert(sizeof(ncclUniqueId) == FX_COMM_ID_BYTES, "FX_COMM_ID_BYTES must match ncclUniqueId");
    ncclUniqueId u;
    NCCL_TRY(fxcomm::api().GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return FX_OK;
}

int fx_comm_create(fx_ctx* ctx, int nranks, int rank, const unsigned char* id, fx_comm** out) {
    if (!ctx || !id || !out) return fail(FX_EINVAL, "fx_comm_create: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(FX_EINVAL, "fx_comm_create: rank %d of %d", rank, nranks);
    if (fx_comm_available() != FX_OK) return FX_EHIP;
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t c = nullptr;
    NCCL_TRY(fxcomm::api().CommInitRank(&c, nranks, u, rank));
    fx_comm* fc = new fx_comm;
    fc->ctx = ctx;
    fc->comm = c;
    fc->nranks = nranks;
    fc->rank = rank;
    *out = fc;
    return FX_OK;
}

int fx_comm_destroy(fx_comm* c) {
    if (!c) return FX_OK;
    int rc = FX_OK;
    if (c->comm) {
        // exchanges are enqueued, not waited for: drain the streams they went to before the communicator is destroyed
        if (hipSetDevice(c->ctx->device) != hipSuccess) rc = FX_EHIP;
        if (c->used_null_stream) {
            if (hipDeviceSynchronize() != hipSuccess) rc = FX_EHIP;
        } else {
            for (int i = 0; i < c->nused; ++i)
                if (hipStreamSynchronize(c->used[i]) != hipSuccess) rc = FX_EHIP;
        }
        (void)hipGetLastError();
        if (fxcomm::api().CommDestroy(c->comm) != ncclSuccess) rc = FX_EHIP;
    }
    delete c;
    return rc == FX_OK ? FX_OK : fail(FX_EHIP, "fx_comm_destroy: draining or destroying the communicator failed");
}

int fx_allgather_tables(fx_comm* c, const double* send, double* recv, int64_t count, int64_t stride, int64_t offset,
                        int64_t recv_count, int algo, void* stream) {
    if (!c || !c->comm) return fail(FX_EINVAL, "fx_allgather_tables: null communicator");
    if (count < 0 || stride < 0 || offset < 0 || offset + count > stride)
        return fail(FX_EINVAL, "fx_allgather_tables: block [%lld, %lld) does not fit the stride %lld", (long long)offset,
                    (long long)(offset + count), (long long)stride);
    // the receive extent the caller owns: nranks blocks at `stride` must fit it (the last block may end early)
    if (recv_count < 0 || (c->nranks > 0 && (int64_t)(c->nranks - 1) * stride + offset + count > recv_count))
        return fail(FX_EINVAL, "fx_allgather_tables: %d blocks at stride %lld (+ offset %lld, count %lld) exceed the receive buffer of %lld doubles",
                    c->nranks, (long long)stride, (long long)offset, (long long)count, (long long)recv_count);
    if (count == 0) return FX_OK;
    if (!send || !recv) return fail(FX_EINVAL, "fx_allgather_tables: null device pointer");
    const fxcomm::Api& a = fxcomm::api();
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->ctx->device));
    c->remember(s);
    if (algo == FX_GATHER_RING) {
        if (stride != count || offset != 0)
            return fail(FX_EINVAL, "fx_allgather_tables: FX_GATHER_RING needs blocks that tile the buffer (stride == count, offset == 0)");
        NCCL_TRY(a.AllGather(send, recv, (size_t)count, ncclDouble, c->comm, s));
        return FX_OK;
    }
    if (algo != FX_GATHER_DIRECT) return fail(FX_EINVAL, "fx_allgather_tables: unknown algorithm %d", algo);
    double* mine = recv + (size_t)c->rank * stride + offset;
    if (mine != send) HIP_TRY(hipMemcpyAsync(mine, send, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (c->nranks == 1) return FX_OK;
    NCCL_TRY(a.GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int d = 1; d < c->nranks && bad == ncclSuccess; ++d) {
        // peer order rotated by the rank: at step d every GPU sends to rank + d and receives from rank - d,
        // so the 7 steps use 7 different links on every GPU
        const int to = (c->rank + d) % c->nranks, from = (c->rank - d + c->nranks) % c->nranks;
        bad = a.Send(send, (size_t)count, ncclDouble, to, c->comm, s);
        if (bad == ncclSuccess) bad = a.Recv(recv + (size_t)from * stride + offset, (size_t)count, ncclDouble, from, c->comm, s);
    }
    const ncclResult_t end = a.GroupEnd();  // always closed, also after a failed call inside the group
    if (bad != ncclSuccess) return fail(FX_EHIP, "ncclSend/ncclRecv: %s", a.GetErrorString(bad));
    if (end != ncclSuccess) return fail(FX_EHIP, "ncclGroupEnd: %s", a.GetErrorString(end));
    return FX_OK;
}

// TensorProductElement.tabulate for ANY two tabulated factors (table_kernels.hpp table_outer_kernel)
int fx_table_outer_batch(fx_ctx* ctx, int order, int sdA, int sdB, int64_t nreq, int npts, int rowsA, int vdimA, int rowsB,
                         int vdimB, const double* tabA, const double* tabB, double* out, void* stream) {
    if (!ctx) return fail(FX_EINVAL, "fx_table_outer_batch: null context");
    if (order < 0 || sdA < 0 || sdB < 0 || sdA + sdB < 1 || nreq < 0 || npts < 0 || rowsA < 1 || rowsB < 1 || vdimA < 1 || vdimB < 1)
        return fail(FX_EINVAL, "fx_table_outer_batch: bad argument");
    if (vdimA > 1 && vdimB > 1) return fail(FX_ENOTIMPL, "tabulate does not support two vector-valued inputs");
    if (nreq == 0 || npts == 0) return FX_OK;
    if (!tabA || !tabB || !out) return fail(FX_EINVAL, "fx_table_outer_batch: null device pointer");
    const int vdim = std::max(vdimA, vdimB);
    if ((long long)rowsA * rowsB * vdim * npts >= (1LL << 24)) return fail(FX_ENOTIMPL, "fx_table_outer_batch: table of %lld entries is too large", (long long)rowsA * rowsB * vdim * npts);
    // tables of a factor: all multi-indices of order <= `order` in mis() order (a 0-dimensional factor has one table)
    auto tables = [&](int sd) {
        std::vector<std::vector<int>> all;
        for (int k = 0; k <= order; ++k) {
            if (sd == 0) {
                if (k == 0) all.push_back({});
                continue;
            }
            std::vector<std::vector<int>> level = fx::multi_indices(sd, k);
            all.insert(all.end(), level.begin(), level.end());
        }
        return all;
    };
    const std::vector<std::vector<int>> TA = tables(sdA), TB = tables(sdB), T = tables(sdA + sdB);
    if ((int)T.size() > fxk::OUTER_MAXTAB || TA.size() > 255 || TB.size() > 255) return fail(FX_ENOTIMPL, "fx_table_outer_batch: too many derivative tables (%d)", (int)T.size());
    fxk::OuterArgs a;
    memset(&a, 0, sizeof a);
    a.A = tabA;
    a.B = tabB;
    a.out = out;
    a.nreq = nreq;
    a.ntab = (int)T.size();
    a.ntabA = (int)TA.size();
    a.ntabB = (int)TB.size();
    a.rowsA = rowsA;
    a.rowsB = rowsB;
    a.vdimA = vdimA;
    a.vdimB = vdimB;
    a.npts = npts;
    for (size_t t = 0; t < T.size(); ++t) {
        const std::vector<int> aa(T[t].begin(), T[t].begin() + sdA), bb(T[t].begin() + sdA, T[t].end());
        a.tA[t] = (unsigned char)(std::find(TA.begin(), TA.end(), aa) - TA.begin());
        a.tB[t] = (unsigned char)(std::find(TB.begin(), TB.end(), bb) - TB.begin());
    }
    HIP_TRY(hipSetDevice(ctx->device));
    // factor tables of a group of requests staged in LDS when they fit a 16 KB tile (and the group's output < 2^24 entries)
    const long long sizeA = (long long)a.ntabA * rowsA * vdimA * npts, sizeB = (long long)a.ntabB * rowsB * vdimB * npts;
    const long long nrows = (long long)a.ntab * rowsA * rowsB * vdim, S = nrows * npts;
    if (sizeA + sizeB <= fxk::OUTER_TILE && nrows <= 8192 && S < (1LL << 22)) {
        fxk::OuterLdsArgs la;
        la.o = a;
        la.sizeA = (int)sizeA;
        la.sizeB = (int)sizeB;
        la.nrows = (int)nrows;
        const long long by_tile = fxk::OUTER_TILE / (sizeA + sizeB), by_out = std::max<long long>(1, 16384 / S);
        la.G = (int)std::max<long long>(1, std::min<long long>(std::min(by_tile, by_out), 64));
        const size_t lds = ((size_t)((2 * nrows + 1) / 2 + 1) / 2 * 2 + 2 * fxk::OUTER_TILE) * sizeof(double);
        const long long ngroups = (nreq + la.G - 1) / la.G;
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)ctx->lds_per_cu / lds));
        const int grid = (int)std::max<long long>(1, std::min<long long>(ngroups, (long long)ctx->num_cu * per_cu));
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fxk::table_outer_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fxk::table_outer_lds_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, la);
        HIP_TRY(hipGetLastError());
        return FX_OK;
    }
    const long long units = nreq * a.ntab;
    const int grid = (int)std::max<long long>(1, std::min<long long>(units, (long long)ctx->num_cu * 16));
    hipLaunchKernelGGL(fxk::table_outer_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

// entity coordinates -> cell coordinates for a whole batch of points (table_kernels.hpp map_points_kernel)
int fx_map_points(fx_ctx* ctx, int din, int dout, const double* M, const double* b, int64_t n, const double* in, double* out,
                  void* stream) {
    if (!ctx) return fail(FX_EINVAL, "fx_map_points: null context");
    if (din < 0 || din > 3 || dout < 1 || dout > 3 || n < 0) return fail(FX_EINVAL, "fx_map_points: bad dimensions (%d -> %d)", din, dout);
    if (n == 0) return FX_OK;
    if (!b || (din > 0 && (!M || !in)) || !out) return fail(FX_EINVAL, "fx_map_points: null pointer");
    fxk::MapPointsArgs a;
    memset(&a, 0, sizeof a);
    a.in = in;
    a.out = out;
    a.n = n;
    a.din = din;
    a.dout = dout;
    for (int i = 0; i < dout * din; ++i) a.M[i] = M[i];
    for (int i = 0; i < dout; ++i) a.b[i] = b[i];
    HIP_TRY(hipSetDevice(ctx->device));
    const int grid = (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)ctx->num_cu * 16));
    hipLaunchKernelGGL(fxk::map_points_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

// FIAT/jacobi.py eval_jacobi_batch / eval_jacobi_deriv_batch on the device (jacobi_kernel.hpp)
int fx_jacobi_batch(fx_ctx* ctx, double a, double b, int n, int order, int64_t npts, const double* xs, double* out,
                    void* stream) {
    if (!ctx) return fail(FX_EINVAL, "fx_jacobi_batch: null context");
    if (n < 0 || order < 0 || npts < 0) return fail(FX_EINVAL, "fx_jacobi_batch: negative degree, order or point count");
    if (n > fxk::JACOBI_MAXN) return fail(FX_ENOTIMPL, "fx_jacobi_batch: degree %d > %d", n, fxk::JACOBI_MAXN);
    if (npts == 0) return FX_OK;
    if (!xs || !out) return fail(FX_EINVAL, "fx_jacobi_batch: null device pointer");
    fxk::JacobiArgs A;
    memset(&A, 0, sizeof A);
    A.xs = xs;
    A.out = out;
    A.npts = npts;
    A.n = n;
    A.order = order;
    const double as = a + order, bs = b + order, apb = as + bs;  // weights of the shifted family
    A.p1c = 0.5 * (as - bs);
    A.p1x = 0.5 * (apb + 2.0);
    for (int k = 2; k <= n - order; ++k) {  // jacobi.py:62-71, same expression order
        const double a1 = 2.0 * k * (k + apb) * (2.0 * k + apb - 2.0);
        A.a2[k] = (2.0 * k + apb - 1.0) * (as * as - bs * bs) / a1;
        A.a3[k] = (2.0 * k + apb - 2.0) * (2.0 * k + apb - 1.0) * (2.0 * k + apb) / a1;
        A.a4[k] = 2.0 * (k + as - 1.0) * (k + bs - 1.0) * (2.0 * k + apb) / a1;
    }
    for (int j = order; j <= n; ++j) {      // jacobi.py:96-101
        double z = 1.0;
        const double f = a + b + j + 1;
        for (int l = 0; l < order; ++l) z *= 0.5 * (f + l);
        A.z[j] = z;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const long long want = (npts + 255) / 256;
    const int grid = (int)std::max<long long>(1, std::min<long long>(want, (long long)ctx->num_cu * 16));
    hipLaunchKernelGGL(fxk::jacobi_batch_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, A);
    HIP_TRY(hipGetLastError());
    return FX_OK;
}

// after a synchronisation: did a dynamically scheduled kernel of this context give up waiting for a chunk
// id (work_queue.hpp)?  Synchronises `stream` first.
int fx_ctx_check(fx_ctx* ctx, void* stream) {
    if (!ctx) return fail(FX_EINVAL, "fx_ctx_check: null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return check_work_queues(ctx);
}

}  // extern "C"
