// Dynamic distribution of work units (requests, pairs of requests) over the persistent waves
// of the specialised tabulation kernels (gfx950).
//
// Measured on MI355X with static, equal shares (tools/placement_probe*.py, FX_DBG=512 builds):
//   * the SIMD arbiter favours the older of its resident waves: the first workgroup of a CU
//     finished after 267 us, the second after 336 us;
//   * under saturated HBM writes the XCDs do not progress at the same rate (odd-numbered XCDs
//     10-30 % slower, varying from launch to launch): equal shares per XCD end in a tail of up
//     to a third of the launch.
// Both are removed by handing out the batch in chunks of 8 consecutive units from ONE counter in
// global memory (a device-scope atomic costs ~12 ns of serialised time and ~0.5 us of latency,
// so it is paid once per chunk and two chunks ahead), and the 8 units of a chunk to the waves
// of the workgroup through a counter in LDS (ds_add_rtn, no vmcnt traffic).
//
// Round 3 (wave timelines of the FX_DBG=512 build, tools/batch_scaling.py: a launch costs 22-33 us more than its
// per-request rate explains -- RT2, 25 000 requests: 7 us from kernel entry to the first unit, and at the end the
// average wave idles 13-17 us, a whole unit, while the last ones finish):
//   * START: a workgroup's first WQ_AHEAD chunks are STATIC (chunk b, b + G, .. of workgroup b of G): the 768 atomics of
//     256 workgroups on one address that used to open every launch (two at init, one at the first claim, each waiting
//     for the previous one's return) are gone; the counter hands out chunk WQ_AHEAD G + n.
//   * END: when the counter runs dry a workgroup still holds its look-ahead -- with two chunks ahead 16-24 units, two
//     to three rounds of its 8 waves, which the slow XCDs work off 10-30 % slower than the fast ones.  The finishing
//     times of the workgroups are spread over one unit by their phase in any case (a chunk is one round of a workgroup);
//     what the look-ahead adds is (held work) x (speed difference).  One chunk ahead (WQ_AHEAD = 1) halves it; the id is
//     requested half a unit after the chunk before it was opened and needed a whole unit after (8 claims = one claim of
//     every wave), so the ~1 us of the atomic stays hidden.
//
// The global counter cleans up after itself: the last wave of the last workgroup to finish sets
// it back to zero (gctr[0] chunk counter, gctr[1] finished workgroups; gctr[2] = protocol-error mark), so a launch costs no
// extra memset node; the host hands concurrent launches different counters (api.hip).
//
// LDS control block (64 bytes at the start of dynamic LDS):
//   ctl[0]            claims of this workgroup so far (k); unit = chunk(k >> 3) * 8 + (k & 7)
//   ctl[1]            waves of this workgroup that have finished
//   ctl[10]           (TAIL) newest chunk id the workgroup holds (byte 40, behind the slot ring)
//   slot[4] (u64)     ring of chunk ids, slot[j & 3] = (j << 32) | chunk id of the workgroup's
//                     j-th chunk, written by the wave that claimed the first unit of chunk j - WQ_AHEAD
#pragma once
#include <hip/hip_runtime.h>

namespace fxk {

constexpr int WQ_CTL_DOUBLES = 8;  // control block size in doubles
constexpr unsigned int WQ_END = 0x7fffffffu;
#ifndef FX_WQ_AHEAD
#define FX_WQ_AHEAD 1
#endif
constexpr unsigned int WQ_AHEAD = FX_WQ_AHEAD;  // chunks requested ahead of the one being opened (1 or 2; the slot ring holds 4)
static_assert(WQ_AHEAD == 1 || WQ_AHEAD == 2, "look-ahead of one or two chunks");

constexpr int WQ_NEWEST = 10;                   // index into ctl (32-bit words) of the newest chunk id, behind the slot ring
constexpr unsigned int WQ_DEFER = 0x7ffffffeu;  // slot value: "fetch this chunk's id when you open it" (TAIL)

// TAIL: near the end of the batch the look-ahead is switched off.  When the counter runs dry a workgroup still holds its
// look-ahead chunk -- a whole round of its waves that nobody else can take, while workgroups that found the counter empty
// sit idle for up to two unit times (wave timelines: the average wave idles 13-17 us before the last one exits, a unit of the
// benchmark kernels takes 11).  So a workgroup whose newest chunk id is within two rounds of the end (ids advance by about
// gridDim.x per round) no longer asks ahead: it marks the slot WQ_DEFER, and the wave that opens that chunk fetches the id
// then (its ~1 us is exposed, but only in the last round or two); every chunk not yet started stays in the counter for
// whoever is free first.
// MEASURED (round 3, FX_DBG=512 timelines + tools/queue_ab.sh, RT2 25 000 / P3 100 000 requests): mean idle before the last
// exit 12.0 -> 10.5-11.5 us, p90 17 -> 15, launch times 151.3 vs 152.4 us (RT2), 205.8 vs 206.6 (N2), 270.3 vs 268.5 (P3):
// within noise, so it is OFF (FX_WQ_TAIL=0 in simplex_pair.hpp).  The exits still spread over two unit times: the last
// chunk of a workgroup is 8 units for its 8 waves, but the waves free up over a whole unit time, so an early wave takes two
// of them while idle workgroups cannot help -- the granularity at the end is a workgroup's chunk, not a unit.
template <bool TAIL> struct WorkQueueT {
    unsigned int* ctl;            // LDS
    unsigned long long* slot;     // LDS, 4 entries
    unsigned int* gctr;           // global chunk counter (zeroed before the launch)
    long long nunits;
    unsigned int nchunks;
    bool pending;                 // this wave owes the workgroup the chunk id of ordinal `pord`
    unsigned int pord;

    // all threads of the workgroup; followed by __syncthreads() in the caller
    __device__ __forceinline__ void init(double* lds, unsigned int* global_counter, long long units) {
        ctl = reinterpret_cast<unsigned int*>(lds);
        slot = reinterpret_cast<unsigned long long*>(lds) + 1;
        gctr = global_counter;
        nunits = units;
        nchunks = (unsigned int)((units + 7) >> 3);
        pending = false;
        pord = 0;
        if (threadIdx.x == 0) {
            ctl[0] = 0;
            ctl[1] = 0;
            ctl[WQ_NEWEST] = blockIdx.x + (WQ_AHEAD - 1) * gridDim.x;   // (TAIL) newest chunk id this workgroup holds
            // the first WQ_AHEAD chunks of a workgroup are static: no atomic before the first unit
            for (unsigned int i = 0; i < 4; ++i)
                slot[i] = i < WQ_AHEAD ? ((unsigned long long)i << 32) | (blockIdx.x + i * gridDim.x) : ~0ULL;
        }
    }

    // wave-uniform: next unit of this wave, or nunits when the batch is exhausted
    __device__ __forceinline__ long long claim() {
        unsigned int r = 0;
        if ((threadIdx.x & 63) == 0) r = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned int k = __builtin_amdgcn_readfirstlane(r);
        const unsigned int j = k >> 3, off = k & 7;
        if (off == 0) {  // first unit of chunk j: this wave fetches the id of chunk j + WQ_AHEAD (see service())
            pending = true;
            pord = j + WQ_AHEAD;
        }
        // the id of chunk j was requested WQ_AHEAD chunks ago; normally it is there.  The wait is
        // bounded (~1 s): a protocol error must not hang the GPU, it ends the batch early instead
        // (and the parity tests fail).
        unsigned long long s = 0;
        bool ok = false;
        for (int spin = 0; spin < (1 << 22); ++spin) {
            s = __hip_atomic_load(&slot[j & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((unsigned int)(s >> 32) == j) {
                ok = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) {  // leave a mark the host finds at its next synchronisation point (gctr[2], never reset by kernels)
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(gctr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return nunits;
        }
        unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)s);
        if constexpr (TAIL) {
            if (lo == WQ_DEFER) {  // (wave-uniform) the id was not asked for ahead of time
                if (off == 0) {    // this wave opens the chunk: fetch the id now and publish it
                    unsigned int c = 0;
                    if ((threadIdx.x & 63) == 0) {
                        c = WQ_AHEAD * gridDim.x + __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        c = c < WQ_END ? c : WQ_END;
                        __hip_atomic_store(&slot[j & 3], ((unsigned long long)j << 32) | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(ctl + WQ_NEWEST, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    lo = __builtin_amdgcn_readfirstlane(c);
                } else {           // the others wait for it (bounded, as above)
                    ok = false;
                    for (int spin = 0; spin < (1 << 22); ++spin) {
                        s = __hip_atomic_load(&slot[j & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if ((unsigned int)(s >> 32) == j && (unsigned int)s != WQ_DEFER) {
                            ok = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (!ok) {
                        if ((threadIdx.x & 63) == 0) __hip_atomic_store(gctr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return nunits;
                    }
                    lo = __builtin_amdgcn_readfirstlane((unsigned int)s);
                }
            }
        }
        if (lo >= nchunks) return nunits;
        const long long u = (long long)lo * 8 + off;
        return u < nunits ? u : nunits;
    }

    // Executes the fetch this wave owes (one device-scope atomic, its return is waited for:
    // call it where few vector-memory operations are outstanding).  Must run before the wave
    // claims again (a second claim would overwrite the debt) or exits.
    __device__ __forceinline__ void service() {
        if (!pending) return;
        pending = false;
        if ((threadIdx.x & 63) == 0) {
            if constexpr (TAIL) {
                const unsigned int last = __hip_atomic_load(ctl + WQ_NEWEST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((unsigned long long)last + 2ull * gridDim.x >= nchunks) {  // within two rounds of the end: no look-ahead
                    __hip_atomic_store(&slot[pord & 3], ((unsigned long long)pord << 32) | WQ_DEFER, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                    return;
                }
            }
            const unsigned int c = WQ_AHEAD * gridDim.x + __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&slot[pord & 3], ((unsigned long long)pord << 32) | (c < WQ_END ? c : WQ_END), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (TAIL) __hip_atomic_store(ctl + WQ_NEWEST, c < WQ_END ? c : WQ_END, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }

    // every wave, once, when it is done (after its last service()): leaves the global counter
    // zeroed for the next launch
    __device__ __forceinline__ void finish() {
        if ((threadIdx.x & 63) == 0) {
            const unsigned int nw = blockDim.x >> 6;
            if (__hip_atomic_fetch_add(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == nw - 1) {
                if (__hip_atomic_fetch_add(gctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
                    __hip_atomic_store(gctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(gctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
};

using WorkQueue = WorkQueueT<false>;

}  // namespace fxk
