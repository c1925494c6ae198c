// comm.hip -- morec_comm_*: the collectives of the data-parallel step enqueued on the CALLER's stream through
// RCCL (ncclAllGather / ncclReduceScatter / ncclAllReduce over xGMI).  Host code only.
//
// RCCL is resolved at run time with dlopen / dlsym and never linked: PyTorch-ROCm ships its own librccl.so.1 and has it mapped
// before this library is loaded; a second copy of RCCL in the process (link-time dependency on /opt/rocm/lib) would carry its own
// topology / IPC state.  RTLD_NOLOAD first picks up whatever copy is already mapped, a plain dlopen is the fallback for a process
// without PyTorch (a C++ host).
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>
#include "common.hpp"

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

template <typename F>
bool sym(void* h, const char* name, F& out) {
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}

const Rccl& rccl() {
    std::call_once(g_rccl_once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if (!g_rccl.h) g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names)
            if (!g_rccl.h) g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl.h) return;
        g_rccl.ok = sym(g_rccl.h, "ncclGetUniqueId", g_rccl.GetUniqueId) && sym(g_rccl.h, "ncclCommInitRank", g_rccl.CommInitRank) &&
                    sym(g_rccl.h, "ncclCommDestroy", g_rccl.CommDestroy) && sym(g_rccl.h, "ncclAllGather", g_rccl.AllGather) &&
                    sym(g_rccl.h, "ncclReduceScatter", g_rccl.ReduceScatter) && sym(g_rccl.h, "ncclAllReduce", g_rccl.AllReduce) &&
                    sym(g_rccl.h, "ncclGetErrorString", g_rccl.GetErrorString);
    });
    return g_rccl;
}
}  // namespace

struct morec_comm {
    ncclComm_t c;
    int rank, world;
    int last_nccl;      // last ncclResult_t that was not ncclSuccess (diagnostics)
};

static int nccl_rc(morec_comm* cm, ncclResult_t r) {
    if (r == ncclSuccess) return MOREC_OK;
    if (cm) cm->last_nccl = (int)r;
    return MOREC_E_COMM;
}

extern "C" int morec_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int morec_comm_unique_id(void* h_id128) {
    if (!h_id128) return MOREC_E_ARG;
    const Rccl& R = rccl();
    if (!R.ok) return MOREC_E_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    return nccl_rc(nullptr, R.GetUniqueId(reinterpret_cast<ncclUniqueId*>(h_id128)));
}

extern "C" int morec_comm_create(morec_comm** out, const void* h_id128, int rank, int world) {
    if (!out || !h_id128 || world < 1 || rank < 0 || rank >= world) return MOREC_E_ARG;
    const Rccl& R = rccl();
    if (!R.ok) return MOREC_E_UNSUPPORTED;
    ncclUniqueId id;
    memcpy(&id, h_id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t r = R.CommInitRank(&c, world, id, rank);      // joins the other ranks: blocks until all have called it
    if (r != ncclSuccess) return MOREC_E_COMM;
    *out = new morec_comm{c, rank, world, 0};
    return MOREC_OK;
}

extern "C" int morec_comm_destroy(morec_comm* cm) {
    if (!cm) return MOREC_E_ARG;
    const ncclResult_t r = rccl().CommDestroy(cm->c);
    delete cm;
    return r == ncclSuccess ? MOREC_OK : MOREC_E_COMM;
}

extern "C" const char* morec_comm_last_error(const morec_comm* cm) {
    const Rccl& R = rccl();
    if (!R.ok) return "librccl.so.1 not found (dlopen)";
    if (!cm) return "no communicator";
    return R.GetErrorString((ncclResult_t)cm->last_nccl);
}

extern "C" int morec_comm_all_gather(morec_comm* cm, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
    if (!cm || !send || !recv) return MOREC_E_ARG;
    if (bytes_per_rank == 0) return MOREC_OK;
    return nccl_rc(cm, rccl().AllGather(send, recv, bytes_per_rank, ncclInt8, cm->c, reinterpret_cast<hipStream_t>(stream)));
}

extern "C" int morec_comm_reduce_scatter_f32(morec_comm* cm, const float* send, float* recv, size_t count_per_rank, void* stream) {
    if (!cm || !send || !recv) return MOREC_E_ARG;
    if (count_per_rank == 0) return MOREC_OK;
    return nccl_rc(cm, rccl().ReduceScatter(send, recv, count_per_rank, ncclFloat32, ncclSum, cm->c, reinterpret_cast<hipStream_t>(stream)));
}

extern "C" int morec_comm_all_reduce_f32(morec_comm* cm, float* buf, size_t count, void* stream) {
    if (!cm || !buf) return MOREC_E_ARG;
    if (count == 0) return MOREC_OK;
    return nccl_rc(cm, rccl().AllReduce(buf, buf, count, ncclFloat32, ncclSum, cm->c, reinterpret_cast<hipStream_t>(stream)));
}
