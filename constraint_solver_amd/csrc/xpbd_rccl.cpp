// xpbd_rccl.cpp -- see xpbd_rccl.h.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE // dladdr
#endif
#include "xpbd_rccl.h"

#include <cstdlib>
#include <dlfcn.h>
#include <mutex>
#include <string>

namespace xpbd {
namespace {

RcclApi g_api;
std::string g_why, g_path;
bool g_ok = false;

bool bind(void *h, const char *path)
{
    RcclApi a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(h, "ncclCommAbort"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GroupStart || !a.GroupEnd || !a.GetErrorString)
        return false;
    g_path = path;
    a.path = g_path.c_str();
    g_api = a;
    return true;
}

// Directory of the HIP runtime this library is running on.  RCCL opens the HSA runtime by its unversioned file name, found
// through RCCL's own run path: a process can hold two ROCm trees (the system's and the one PyTorch bundles) with only one
// of them initialised -- whichever was loaded first -- and an RCCL from the OTHER tree then opens a second, uninitialised
// copy of the HSA runtime and fails with "no ROCm-capable device".  So RCCL is taken from the tree of the active runtime.
std::string active_runtime_dir()
{
    Dl_info info;
    if (!dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) || !info.dli_fname)
        return "";
    const std::string path = info.dli_fname;
    const size_t slash = path.rfind('/');
    return slash == std::string::npos ? "" : path.substr(0, slash);
}

void load_once()
{
    struct Try {
        const char *name;
        int flags;
    };
    const char *env = std::getenv("XPBD_RCCL_LIB");
    const std::string dir = active_runtime_dir();
    const std::string beside1 = dir.empty() ? "" : dir + "/librccl.so.1", beside = dir.empty() ? "" : dir + "/librccl.so";
    const Try tries[] = {
        {env, RTLD_NOW | RTLD_LOCAL},
        {beside1.c_str(), RTLD_NOW | RTLD_LOCAL},
        {beside.c_str(), RTLD_NOW | RTLD_LOCAL},
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
        {"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL},
        {"librccl.so", RTLD_NOW | RTLD_LOCAL},
        {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL},
    };
    for (const Try &t : tries) {
        if (!t.name || !*t.name)
            continue;
        void *h = dlopen(t.name, t.flags);
        if (!h) {
            if (!(t.flags & RTLD_NOLOAD)) {
                const char *e = dlerror();
                g_why += std::string(t.name) + ": " + (e ? e : "not found") + "; ";
            }
            continue;
        }
        if (bind(h, t.name)) {
            g_ok = true;
            return;
        }
        g_why += std::string(t.name) + ": RCCL symbols missing; ";
    }
    if (g_why.empty())
        g_why = "no RCCL library found";
}

} // namespace

const RcclApi *rccl_api(const char **why)
{
    static std::once_flag once;
    std::call_once(once, load_once);
    if (why)
        *why = g_why.c_str();
    return g_ok ? &g_api : nullptr;
}

} // namespace xpbd
