// The training step's one collective as C entry points: an in-place fp32 sum all-reduce of the flat gradient
// bucket over RCCL, one process per GPU.  The RCCL library is not a link-time dependency of libunetmi.so: it
// is looked up when the communicator is created -- first the copy the process already has loaded (a PyTorch
// process carries one), then the system's -- so the convolution entry points load and run where no RCCL exists.
//
// Rendezvous is the caller's: rank 0 asks for an id (unetmi_comm_unique_id), hands its 128 bytes to the other
// ranks through whatever channel the job already has (a TCP store, MPI, a file), and every rank calls
// unetmi_comm_init with the GPU it owns current.  One communicator per process.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "../../include/unetmi.h"

namespace unetmi {

int api_fail(int code, const std::string& msg);

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
};

std::mutex g_mutex;            // guards g_rccl, g_comm, g_world
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_world = 0;

// with g_mutex held
bool load_rccl(std::string* err) {
    if (g_rccl.handle) return true;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);          // the copy this process already uses, if any
        if (h) break;
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (h) break;
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) { *err = std::string("RCCL library not found: ") + dlerror(); return false; }
    Rccl r;
    r.handle = h;
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(h, "ncclAllReduce"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(h, "ncclGetErrorString"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy || !r.error_string) {
        *err = "the RCCL library lacks an expected symbol";
        dlclose(h);
        return false;
    }
    g_rccl = r;
    return true;
}

int rccl_fail(const char* what, ncclResult_t r) {
    return api_fail(UNETMI_E_HIP, std::string(what) + ": " + g_rccl.error_string(r));
}

}  // namespace
}  // namespace unetmi

using namespace unetmi;

extern "C" {

int unetmi_comm_unique_id(void* id128) {
    if (!id128) return api_fail(UNETMI_E_BAD_ARG, "comm_unique_id: null buffer");
    static_assert(sizeof(ncclUniqueId) == UNETMI_COMM_ID_BYTES, "the id is handed over as 128 opaque bytes");
    std::lock_guard<std::mutex> lock(g_mutex);
    std::string err;
    if (!load_rccl(&err)) return api_fail(UNETMI_E_UNSUPPORTED, err);
    const ncclResult_t r = g_rccl.get_unique_id(static_cast<ncclUniqueId*>(id128));
    return r == ncclSuccess ? UNETMI_OK : rccl_fail("ncclGetUniqueId", r);
}

int unetmi_comm_init(int rank, int world, const void* id128) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) return api_fail(UNETMI_E_BAD_ARG, "comm_init: bad rank / world / id");
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_comm) return api_fail(UNETMI_E_BAD_ARG, "comm_init: this process already has a communicator (unetmi_comm_destroy first)");
    std::string err;
    if (!load_rccl(&err)) return api_fail(UNETMI_E_UNSUPPORTED, err);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.comm_init_rank(&comm, world, id, rank);
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    g_comm = comm;
    g_world = world;
    return UNETMI_OK;
}

int unetmi_comm_world(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    return g_comm ? g_world : 0;
}

int unetmi_allreduce_f32(float* buf, size_t n, void* stream) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_comm) return api_fail(UNETMI_E_BAD_ARG, "allreduce_f32: no communicator (unetmi_comm_init first)");
    if (n == 0) return UNETMI_OK;
    if (!buf) return api_fail(UNETMI_E_BAD_ARG, "allreduce_f32: null buffer");
    const ncclResult_t r = g_rccl.all_reduce(buf, buf, n, ncclFloat32, ncclSum, g_comm, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? UNETMI_OK : rccl_fail("ncclAllReduce", r);
}

int unetmi_comm_destroy(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_comm) return UNETMI_OK;
    const ncclResult_t r = g_rccl.comm_destroy(g_comm);
    g_comm = nullptr;
    g_world = 0;
    return r == ncclSuccess ? UNETMI_OK : rccl_fail("ncclCommDestroy", r);
}

}  // extern "C"
