// Host side of the request-per-workgroup kernel (simplex_wg.hpp): its own translation unit, so that the ~60 instances
// compile beside api.hip instead of inside it.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "simplex_wg.hpp"
#include "wg_launch.hpp"

namespace {

// (two waves per row tile wherever the column tiles split evenly: DESIGN.md 4.5c)
constexpr int wg_pc(int ct) { return ct % 2 == 0 ? 2 : 1; }

template <int SD, int N, int CT, bool ODD, int PC, int MIX, bool FAST = false>
hipError_t launch(const fxk::StackedArgs<0>& h, const double* coef, int ncoef, int lds_bytes, int grid, double* trash, unsigned int* queue,
                  hipStream_t s) {
    constexpr int NC = fxk::FixedNC<SD, N>::value;
    constexpr int KS = (fxk::StepTable<SD, N>::NEXP + 3) / 4;
    if (ncoef != NC || lds_bytes != fxk::wg_lds_doubles(CT, KS) * 8 || h.npts < 1 || h.gslab < 1 || (long long)h.gslab * h.npts > 16 * CT ||
        (long long)grid * h.gslab > h.nreq + h.gslab - 1 || (PC == 1 && h.gslab != 1) || (FAST && h.RT > fxk::WG_NW))
        return hipErrorInvalidValue;
    fxk::StackedArgs<NC> ka;
    memset(&ka, 0, sizeof ka);
    ka.pts = h.pts;
    ka.verts = h.verts;
    ka.out = h.out;
    ka.afrag = h.afrag;
    ka.phi0 = h.phi0;
    memcpy(ka.A0, h.A0, sizeof ka.A0);
    memcpy(ka.b0, h.b0, sizeof ka.b0);
    memcpy(ka.A0inv, h.A0inv, sizeof ka.A0inv);
    ka.nreq = h.nreq;
    ka.npts = h.npts;
    ka.R = h.R;
    ka.RT = h.RT;
    ka.gslab = h.gslab;
    ka.lim_pts = h.lim_pts;
    ka.lim_verts = h.lim_verts;
    ka.lim_out = h.lim_out;
    ka.lim_afrag = h.lim_afrag;
    memcpy(ka.coef, coef, NC * sizeof(double));
    auto kern = fxk::tabulate_simplex_wg<SD, N, CT, ODD, PC, MIX, FAST>;
    static thread_local bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr = true;
    }
#if FX_WG_DBG == 3
    {   // is the counter slot clean when the launch takes it?
        unsigned int q4[4];
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(q4, queue, sizeof q4, hipMemcpyDeviceToHost) == hipSuccess && (q4[0] | q4[1] | q4[2] | q4[3]))
            fprintf(stderr, "[fiat_amd] WG DIRTY SLOT %p: %u %u %u %u (nreq %lld grid %d)\n", (void*)queue, q4[0], q4[1], q4[2], q4[3], (long long)h.nreq, grid);
    }
#endif
#if FX_WG_DBG
    static int dbg_launches = 0;
    if (dbg_launches++ == 0) (void)hipMemsetAsync(trash + 4096, 0, 24 * sizeof(double), s);
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * fxk::WG_NW), lds_bytes, s, ka, trash, queue);
#if FX_WG_DBG
    if (FX_WG_DBG == 1 || dbg_launches % 16 == 0) {   // range-check build: report accesses that left their buffers (redirected to the scratch area by the kernel); 2: every 16th launch only (launches stay back to back)
        double rep[24];
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(rep, trash + 4096, sizeof rep, hipMemcpyDeviceToHost) != hipSuccess) return hipErrorUnknown;
        static const char* site[] = {"", "pts", "", "afrag", "out", "out-address"};
        for (int k = 1; k <= 5; ++k) {
            unsigned long long cnt;
            memcpy(&cnt, &rep[4 * k], sizeof cnt);
            if (cnt)
                fprintf(stderr, "[fiat_amd] WG RANGE CHECK <%d,%d,%d>: %llu accesses outside `%s`: first index %.0f, limit %.0f, request %.0f\n", SD, N, CT,
                        cnt, site[k], rep[4 * k + 1], rep[4 * k + 2], rep[4 * k + 3]);
        }
        (void)hipMemset(trash + 4096, 0, sizeof rep);
    }
#endif
#if FX_WG_TIME
    {
        static int tl = 0;
        if (h.nreq > 8 && (tl++ % 64) == 40) {
            double rep[32];
            if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(rep, trash + 8192, sizeof rep, hipMemcpyDeviceToHost) != hipSuccess) return hipErrorUnknown;
            for (int w = 0; w < 4; ++w)
                fprintf(stderr, "[fiat_amd] WG TIME <%d,%d,%d,pc%d,mix%d> wave %d: %.0f groups; ticks per group: top %.0f production %.0f barrier %.0f operands %.0f sweep %.0f tail %.0f\n",
                        SD, N, CT, PC, MIX, w, rep[8 * w + 6], rep[8 * w + 5] / rep[8 * w + 6], rep[8 * w] / rep[8 * w + 6], rep[8 * w + 1] / rep[8 * w + 6], rep[8 * w + 2] / rep[8 * w + 6],
                        rep[8 * w + 3] / rep[8 * w + 6], rep[8 * w + 4] / rep[8 * w + 6]);
        }
    }
#endif
    return hipGetLastError();
}

// (the chain-rule instances: tetrahedra on 4 column tiles with two waves per row tile, on 6 up to degree 5 (degree 6 spills 98
// registers there), on 8 with four waves per row tile; triangles on 4 / 6 / 8 with two)
constexpr int wg_mix_pc(int sd, int ct, bool odd) { return ct == 8 && (sd == 3 || odd) ? 4 : 2; }
constexpr bool wg_has_mix(int sd, int n, int ct) { return sd == 3 ? (ct == 4 || ct == 8 || (ct == 6 && n <= 5)) : (ct == 4 || ct == 6 || ct == 8); }

template <int SD, int N, int CT>
hipError_t launch_odd(bool odd, int mix, const fxk::StackedArgs<0>& h, const double* coef, int ncoef, int lds_bytes, int grid, double* trash,
                      unsigned int* queue, hipStream_t s) {
    if (mix) {
        if constexpr (wg_has_mix(SD, N, CT)) {
            if (!odd) return launch<SD, N, CT, false, wg_mix_pc(SD, CT, false), 1>(h, coef, ncoef, lds_bytes, grid, trash, queue, s);
            // (degree-6 tetrahedra on eight column tiles: the 8-byte twin spills 96 registers -- not instantiated; scalar elements of
            // that degree have 84 rows a table, their requests are never odd)
            if constexpr (!(SD == 3 && N == 6 && CT == 8))
                return launch<SD, N, CT, true, wg_mix_pc(SD, CT, true), 1>(h, coef, ncoef, lds_bytes, grid, trash, queue, s);
        }
        return hipErrorInvalidValue;
    }
    if constexpr (wg_pc(CT) == 1) {   // at most one row tile per wave (values-only requests of up to 64 rows): the FAST instances
        if (h.RT <= fxk::WG_NW)
            return odd ? launch<SD, N, CT, true, 1, 0, true>(h, coef, ncoef, lds_bytes, grid, trash, queue, s)
                       : launch<SD, N, CT, false, 1, 0, true>(h, coef, ncoef, lds_bytes, grid, trash, queue, s);
    }
    return odd ? launch<SD, N, CT, true, wg_pc(CT), 0>(h, coef, ncoef, lds_bytes, grid, trash, queue, s)
               : launch<SD, N, CT, false, wg_pc(CT), 0>(h, coef, ncoef, lds_bytes, grid, trash, queue, s);
}

template <int SD, int N>
hipError_t launch_ct(int ct, bool odd, int mix, const fxk::StackedArgs<0>& h, const double* coef, int ncoef, int lds_bytes, int grid, double* trash,
                     unsigned int* queue, hipStream_t s) {
    constexpr int KS = (fxk::StepTable<SD, N>::NEXP + 3) / 4;
    switch (ct) {
        case 4: return launch_odd<SD, N, 4>(odd, mix, h, coef, ncoef, lds_bytes, grid, trash, queue, s);
        case 5: return launch_odd<SD, N, 5>(odd, mix, h, coef, ncoef, lds_bytes, grid, trash, queue, s);
        case 6: return launch_odd<SD, N, 6>(odd, mix, h, coef, ncoef, lds_bytes, grid, trash, queue, s);
        case 8: if constexpr (fxk::wg_lds_doubles(8, KS) * 8 <= 160 * 1024) return launch_odd<SD, N, 8>(odd, mix, h, coef, ncoef, lds_bytes, grid, trash, queue, s); break;
    }
    return hipErrorInvalidValue;
}

}  // namespace

namespace fxwg {

int lds_bytes(int sd, int n, int ct) {
    const int nexp = sd == 3 ? (n + 1) * (n + 2) * (n + 3) / 6 : (n + 1) * (n + 2) / 2;
    return fxk::wg_lds_doubles(ct, (nexp + 3) / 4) * 8;
}

bool has_instance(int sd, int n, int ct, bool odd) {
    // (seven column tiles: the one-wave-per-row-tile layout spills there -- the planner takes eight, two waves per row tile)
    if (ct < 4 || ct > 8 || ct == 7 || lds_bytes(sd, n, ct) > 160 * 1024) return false;
    (void)odd;
    return (sd == 3 && n >= 3 && n <= 6) || (sd == 2 && (n == 5 || n == 6));
}

// (the one-row-tile instances -- one wave per row tile, at most four row tiles, no chain rule -- with 16-byte flush pieces are compiled for 256 registers: two
// workgroups share a CU where two slabs fit its LDS, one's recurrence phase under the other's MFMAs)
int workgroups_per_cu(int sd, int n, int ct, bool odd, int mix, int rt) {
    return (wg_pc(ct) == 1 && !odd && !mix && rt <= fxk::WG_NW && 2 * lds_bytes(sd, n, ct) <= 160 * 1024) ? 2 : 1;
}

bool has_mix_instance(int sd, int n, int ct, bool odd) { return has_instance(sd, n, ct, odd) && wg_has_mix(sd, n, ct) && !(odd && sd == 3 && n == 6 && ct == 8); }

int mix_ct(int sd, int n, int ctn) {
    if (ctn <= 4) return 4;
    if (sd == 2) return (ctn + 1) & ~1;
    if (ctn <= 6) return n <= 5 ? 6 : 0;   // (degree 6 on six column tiles spills: the point-chunked instances keep 65..96 points)
    return 8;
}

hipError_t launch_simplex_wg(int sd, int n, int ct, bool odd, int mix, const fxk::StackedArgs<0>& h, const double* coef, int ncoef, int lds, int grid,
                             double* trash, unsigned int* queue, hipStream_t s) {
    if (!has_instance(sd, n, ct, odd) || (mix && !has_mix_instance(sd, n, ct, odd))) return hipErrorInvalidValue;
    if (sd == 3 && n == 6) return launch_ct<3, 6>(ct, odd, mix, h, coef, ncoef, lds, grid, trash, queue, s);
    if (sd == 3 && n == 5) return launch_ct<3, 5>(ct, odd, mix, h, coef, ncoef, lds, grid, trash, queue, s);
    if (sd == 3 && n == 4) return launch_ct<3, 4>(ct, odd, mix, h, coef, ncoef, lds, grid, trash, queue, s);
    if (sd == 3 && n == 3) return launch_ct<3, 3>(ct, odd, mix, h, coef, ncoef, lds, grid, trash, queue, s);
    if (sd == 2 && n == 6) return launch_ct<2, 6>(ct, odd, mix, h, coef, ncoef, lds, grid, trash, queue, s);
    if (sd == 2 && n == 5) return launch_ct<2, 5>(ct, odd, mix, h, coef, ncoef, lds, grid, trash, queue, s);
    return hipErrorInvalidValue;
}

}  // namespace fxwg
