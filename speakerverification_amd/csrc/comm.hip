// comm.hip — the path's one exchange step under the C ABI: an RCCL all-gather of embedding rows on the handle's stream.
//
// Replaces the reference's pickled `all_gather_object` of per-rank feature dicts (src/model.py:400-411).  One process per
// GPU, one communicator per handle (ncclCommInitRank); the 128-byte unique id travels between the processes by whatever
// side channel the host has (speakerverification_amd/distributed.py uses the torch.distributed store; a file or an
// environment variable works as well) — the data path itself never touches torch.
//
// RCCL is bound lazily (dlopen "librccl.so.1"): a process that already mapped an RCCL with that SONAME (PyTorch-ROCm ships
// one) gets that same copy, so there is exactly one RCCL per process; a process that never calls svhip_comm_* needs none.
#include "../../include/svhip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "kernels.h"

namespace {

struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // SVHIP_RCCL_LIB names the library instead of the default search (deployments with a private RCCL; tests of the failure path)
        const char* forced = getenv("SVHIP_RCCL_LIB");
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::string why;
        for (const char* n : names) {
            if (forced) n = forced;
            r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
            const char* m = dlerror();                  // (one call: dlerror() clears the message it returns)
            if (why.empty()) why = m ? m : "?";
            if (forced) break;
        }
        if (!r.so) { r.err = "cannot load librccl: " + why; return; }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.so, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.so, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.so, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.so, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) {
            r.err = "librccl lacks a required symbol";
            r.so = nullptr;
        }
    });
    return &r;
}

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    float* stage_in = nullptr;       // device staging for host-pointer calls, grown on demand
    float* stage_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
};

int fail(svhip_handle* h, int code, const char* what, const char* detail) {
    char b[512];
    snprintf(b, sizeof(b), "%s: %s", what, detail ? detail : "?");
    svhip::handle_set_error(h, b);
    return code;
}

thread_local std::string g_id_error;

}  // namespace

extern "C" {

int svhip_comm_unique_id(void* id_out) {
    if (!id_out) return SVHIP_ERR_INVALID;
    Rccl* r = rccl();
    if (!r->so) { g_id_error = r->err; return SVHIP_ERR_UNSUPPORTED; }
    ncclUniqueId id;
    const ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) { g_id_error = r->GetErrorString(e); return SVHIP_ERR_HIP; }
    static_assert(sizeof(id) == SVHIP_COMM_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof(id));
    return SVHIP_OK;
}

int svhip_comm_init(svhip_handle* h, const void* id_bytes, int32_t rank, int32_t world) {
    if (!h || !id_bytes || world <= 0 || rank < 0 || rank >= world) return SVHIP_ERR_INVALID;
    if (svhip::handle_comm(h)) return fail(h, SVHIP_ERR_STATE, "svhip_comm_init", "the handle already owns a communicator");
    Rccl* r = rccl();
    if (!r->so) return fail(h, SVHIP_ERR_UNSUPPORTED, "svhip_comm_init", r->err.c_str());
    hipError_t he = hipSetDevice(svhip::handle_device(h));
    if (he != hipSuccess) return fail(h, SVHIP_ERR_HIP, "hipSetDevice", hipGetErrorString(he));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    Comm* c = new Comm();
    c->rank = rank; c->world = world;
    const ncclResult_t e = r->CommInitRank(&c->comm, world, id, rank);
    if (e != ncclSuccess) { delete c; return fail(h, SVHIP_ERR_HIP, "ncclCommInitRank", r->GetErrorString(e)); }
    svhip::handle_comm(h) = c;
    return SVHIP_OK;
}

int svhip_comm_rank(const svhip_handle* h, int32_t* rank, int32_t* world) {
    if (!h) return SVHIP_ERR_INVALID;
    const Comm* c = static_cast<const Comm*>(svhip::handle_comm(const_cast<svhip_handle*>(h)));
    if (rank) *rank = c ? c->rank : 0;
    if (world) *world = c ? c->world : 1;
    return c ? SVHIP_OK : SVHIP_ERR_STATE;
}

int svhip_comm_destroy(svhip_handle* h) {
    if (!h) return SVHIP_ERR_INVALID;
    Comm* c = static_cast<Comm*>(svhip::handle_comm(h));
    if (!c) return SVHIP_OK;
    (void)hipSetDevice(svhip::handle_device(h));
    (void)hipStreamSynchronize(svhip::handle_stream(h));
    Rccl* r = rccl();
    if (r->so && c->comm) (void)r->CommDestroy(c->comm);
    if (c->stage_in) (void)hipFree(c->stage_in);
    if (c->stage_out) (void)hipFree(c->stage_out);
    delete c;
    svhip::handle_comm(h) = nullptr;
    return SVHIP_OK;
}

int svhip_allgather_rows(svhip_handle* h, const float* local, int64_t rows, int32_t D, float* out, int32_t flags) {
    if (!h || !local || !out || rows < 0 || D <= 0) return SVHIP_ERR_INVALID;
    Comm* c = static_cast<Comm*>(svhip::handle_comm(h));
    if (!c) return fail(h, SVHIP_ERR_STATE, "svhip_allgather_rows", "no communicator (call svhip_comm_init first)");
    const bool din = flags & SVHIP_IN_DEVICE, dout = flags & SVHIP_OUT_DEVICE;
    if ((flags & SVHIP_ASYNC) && !(din && dout)) return fail(h, SVHIP_ERR_INVALID, "svhip_allgather_rows", "SVHIP_ASYNC needs device pointers");
    if (rows == 0) return SVHIP_OK;
    hipError_t he = hipSetDevice(svhip::handle_device(h));
    if (he != hipSuccess) return fail(h, SVHIP_ERR_HIP, "hipSetDevice", hipGetErrorString(he));
    hipStream_t st = svhip::handle_stream(h);
    const size_t n_in = (size_t)rows * D, n_out = n_in * c->world;
    const float* d_in = local;
    float* d_out = out;
    if (!din) {
        if (c->cap_in < n_in) {
            if (c->stage_in) (void)hipFree(c->stage_in);
            c->stage_in = nullptr; c->cap_in = 0;
            if ((he = hipMalloc((void**)&c->stage_in, n_in * 4)) != hipSuccess) return fail(h, SVHIP_ERR_NOMEM, "hipMalloc", hipGetErrorString(he));
            c->cap_in = n_in;
        }
        if ((he = hipMemcpyAsync(c->stage_in, local, n_in * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(h, SVHIP_ERR_HIP, "hipMemcpyAsync", hipGetErrorString(he));
        d_in = c->stage_in;
    }
    if (!dout) {
        if (c->cap_out < n_out) {
            if (c->stage_out) (void)hipFree(c->stage_out);
            c->stage_out = nullptr; c->cap_out = 0;
            if ((he = hipMalloc((void**)&c->stage_out, n_out * 4)) != hipSuccess) return fail(h, SVHIP_ERR_NOMEM, "hipMalloc", hipGetErrorString(he));
            c->cap_out = n_out;
        }
        d_out = c->stage_out;
    }
    Rccl* r = rccl();
    const int rc = svhip::handle_run(h, "allgather_rows", [&]() -> hipError_t {
        const ncclResult_t e = r->AllGather(d_in, d_out, n_in, ncclFloat32, c->comm, st);
        if (e != ncclSuccess) { svhip::handle_set_error(h, r->GetErrorString(e)); return hipErrorUnknown; }
        return hipSuccess;
    });
    if (rc) return rc;
    if (!dout && (he = hipMemcpyAsync(out, d_out, n_out * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(h, SVHIP_ERR_HIP, "hipMemcpyAsync", hipGetErrorString(he));
    if (!(flags & SVHIP_ASYNC) && (he = hipStreamSynchronize(st)) != hipSuccess) return fail(h, SVHIP_ERR_HIP, "hipStreamSynchronize", hipGetErrorString(he));
    return SVHIP_OK;
}

const char* svhip_comm_last_error(void) { return g_id_error.c_str(); }

}  // extern "C"
