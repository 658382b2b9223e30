// xpbd_rccl.h -- RCCL bound at run time (dlopen), so that libxpbd_hip.so loads on hosts without RCCL and, inside a process
// that already carries a copy of RCCL (PyTorch ships its own), uses THAT copy instead of bringing in a second one.
#pragma once

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h> // types and prototypes only: nothing is linked

namespace xpbd {

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr; // optional: used when a collective could not be enqueued
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    const char *path = ""; // what was opened
};

// The process-wide binding, or nullptr with *why set.  Search order: $XPBD_RCCL_LIB, librccl next to the HIP runtime this
// library runs on (see xpbd_rccl.cpp: it must come from the same ROCm tree), a copy already loaded into the process, then
// librccl.so.1 / librccl.so on the loader path, then /opt/rocm/lib.
const RcclApi *rccl_api(const char **why);

} // namespace xpbd
