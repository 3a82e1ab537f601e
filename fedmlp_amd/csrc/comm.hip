// RCCL binding of the engine (host code).  FedAvg (utils/FedAvg.py:7-14) and its FedMLP tails
// (FedAvg_tao :51-70, FedAvg_proto :72-93) are sums over clients; with one client per GPU they are
// ncclAllReduce(SUM) calls over xGMI on the engine's stream.  RCCL is resolved at run time (dlopen):
// a process that already holds an RCCL (torch bundles one next to its HIP runtime) keeps using that
// copy, a plain C caller gets /opt/rocm/lib/librccl.so.1; the library itself has no link-time
// dependency, so single-GPU users never load RCCL at all.
#include <dlfcn.h>
#include <link.h>
#include <string.h>

#include <string>

#include "comm.h"

namespace {

struct UniqueId { char internal[128]; };              // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);

struct Api {
    void* lib = nullptr;
    GetUniqueIdFn get_id = nullptr;
    CommInitRankFn init = nullptr;
    CommDestroyFn destroy = nullptr;
    AllReduceFn allreduce = nullptr;
    GetErrorStringFn errstr = nullptr;
    std::string err;
};
Api g_api;

int find_loaded(struct dl_phdr_info* info, size_t, void* out)
{
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
        *static_cast<std::string*>(out) = info->dlpi_name;
        return 1;
    }
    return 0;
}

bool load_api()
{
    if (g_api.lib) return true;
    std::string loaded;
    dl_iterate_phdr(find_loaded, &loaded);
    const char* cands[] = {loaded.empty() ? nullptr : loaded.c_str(), "librccl.so.1", "librccl.so",
                           "/opt/rocm/lib/librccl.so.1"};
    for (const char* c : cands) {
        if (!c) continue;
        g_api.lib = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (g_api.lib) break;
    }
    if (!g_api.lib) { g_api.err = std::string("cannot load librccl: ") + dlerror(); return false; }
    g_api.get_id = (GetUniqueIdFn)dlsym(g_api.lib, "ncclGetUniqueId");
    g_api.init = (CommInitRankFn)dlsym(g_api.lib, "ncclCommInitRank");
    g_api.destroy = (CommDestroyFn)dlsym(g_api.lib, "ncclCommDestroy");
    g_api.allreduce = (AllReduceFn)dlsym(g_api.lib, "ncclAllReduce");
    g_api.errstr = (GetErrorStringFn)dlsym(g_api.lib, "ncclGetErrorString");
    if (!g_api.get_id || !g_api.init || !g_api.destroy || !g_api.allreduce) {
        g_api.err = "librccl lacks ncclGetUniqueId/ncclCommInitRank/ncclCommDestroy/ncclAllReduce";
        g_api.lib = nullptr;
        return false;
    }
    return true;
}

bool chk(int rc, const char* what)
{
    if (rc == 0) return true;
    g_api.err = std::string(what) + ": " + (g_api.errstr ? g_api.errstr(rc) : "rccl error");
    return false;
}

}  // namespace

const char* fmcomm_error() { return g_api.err.c_str(); }
bool fmcomm_preflight() { return load_api(); }

bool fmcomm_unique_id(unsigned char id[128])
{
    if (!load_api()) return false;
    UniqueId u;
    if (!chk(g_api.get_id(&u), "ncclGetUniqueId")) return false;
    memcpy(id, u.internal, 128);
    return true;
}

bool fmcomm_init(void** comm, const unsigned char id[128], int rank, int world)
{
    if (!load_api()) return false;
    UniqueId u;
    memcpy(u.internal, id, 128);
    return chk(g_api.init(comm, world, u, rank), "ncclCommInitRank");
}

bool fmcomm_destroy(void* comm)
{
    if (!comm || !g_api.lib) return true;
    return chk(g_api.destroy(comm), "ncclCommDestroy");
}

bool fmcomm_allreduce_sum(void* comm, void* buf, size_t n, bool f64, hipStream_t s)
{
    // ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0 (rccl.h)
    return chk(g_api.allreduce(buf, buf, n, f64 ? 8 : 7, 0, comm, s), "ncclAllReduce");
}
