// Multi-GPU reassembly of the tables: one process per GPU, rank g owns a contiguous block of the
// requests and -- only when the consumer needs every table on every GPU -- the blocks are exchanged over xGMI with
// RCCL.  RCCL is bound at run time (dlopen: the copy PyTorch-ROCm already loaded when there is one, so that the
// process holds ONE RCCL instance), the library itself has no link dependency on it.
//
// Two exchange patterns behind fx_allgather_tables:
//   RING    ncclAllGather -- the library's choice of algorithm; blocks must tile the receive buffer.
//   DIRECT  one grouped ncclSend/ncclRecv pair per peer: every GPU pushes its block over all 7 xGMI links at once
//           (the node is fully connected, 7 x ~153 GB/s per GPU; a ring is bound by ONE link), and the blocks may
//           land at a stride and offset in the receive buffer -- what the chunked, compute-overlapped gather needs
//           (chunk c of rank p goes to recv[p * stride + offset]).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only

#include <mutex>

namespace fxcomm {

struct Api {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    const char* why = "not loaded";
};

inline const Api& api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)  // already in the process (PyTorch-ROCm)?
            if (!a.handle) a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names)
            if (!a.handle) a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!a.handle) a.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!a.handle) {
            a.why = "librccl.so not found";
            return;
        }
#define FX_SYM(field, name)                                             \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, name)); \
    if (!a.field) {                                                       \
        a.why = "symbol " name " missing from librccl";                   \
        a.handle = nullptr;                                               \
        return;                                                           \
    }
        FX_SYM(GetUniqueId, "ncclGetUniqueId")
        FX_SYM(CommInitRank, "ncclCommInitRank")
        FX_SYM(CommDestroy, "ncclCommDestroy")
        FX_SYM(AllGather, "ncclAllGather")
        FX_SYM(Send, "ncclSend")
        FX_SYM(Recv, "ncclRecv")
        FX_SYM(GroupStart, "ncclGroupStart")
        FX_SYM(GroupEnd, "ncclGroupEnd")
        FX_SYM(GetErrorString, "ncclGetErrorString")
#undef FX_SYM
        a.why = "ok";
    });
    return a;
}

}  // namespace fxcomm

struct fx_comm {
    fx_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
    // streams an exchange was enqueued on: fx_comm_destroy waits for them before the communicator goes away
    // (ncclCommDestroy with work in flight is undefined)
    hipStream_t used[4] = {nullptr, nullptr, nullptr, nullptr};
    int nused = 0;
    bool used_null_stream = false;
    void remember(hipStream_t s) {
        if (!s) {
            used_null_stream = true;
            return;
        }
        for (int i = 0; i < nused; ++i)
            if (used[i] == s) return;
        if (nused < 4) used[nused++] = s;
        else used_null_stream = true;  // more streams than slots: fall back to a device-wide wait at destroy time
    }
};
