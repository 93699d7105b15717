// Thin C-ABI wrappers over an RCCL communicator (gs_comm_init / gs_allreduce_*).
// The reference is single-GPU (gan_synth_main.py:91-98): data parallelism is new.  One process per GPU; the flat fp32 gradient
// of a network is summed over ranks with ncclAllReduce ON THE CALLER'S STREAM -- the stream the backward ran on -- so the
// collective is ordered behind the gradients and ahead of the optimizer update without any cross-stream event (on this ROCm
// stack an event hop between a hipGraph replay and another stream costs ~0.25 ms, more than the collective itself).
// RCCL is resolved at run time (dlopen of the already-loaded librccl.so.1 when the host process has one, e.g. PyTorch's), so the
// library carries no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "gs_common.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;   // the copy the process already uses
    if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return gs::fail(GS_ERR_UNSUPPORTED, "gs_comm: librccl.so.1 not found (%s)", dlerror());
#define GS_SYM(field, name)                                                                                    \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                                   \
    if (!g_rccl.field) return gs::fail(GS_ERR_UNSUPPORTED, "gs_comm: symbol %s missing in librccl", name)
    GS_SYM(GetUniqueId, "ncclGetUniqueId");
    GS_SYM(CommInitRank, "ncclCommInitRank");
    GS_SYM(CommDestroy, "ncclCommDestroy");
    GS_SYM(CommCount, "ncclCommCount");
    GS_SYM(AllReduce, "ncclAllReduce");
    GS_SYM(Broadcast, "ncclBroadcast");
    GS_SYM(GetErrorString, "ncclGetErrorString");
#undef GS_SYM
    g_rccl.lib = h;
    return 0;
}

#define GS_NCCL_OK(expr)                                                                                      \
    do {                                                                                                      \
        ncclResult_t r__ = (expr);                                                                            \
        if (r__ != ncclSuccess) return gs::fail(GS_ERR_HIP, "%s: %s", #expr, g_rccl.GetErrorString(r__));     \
    } while (0)

}  // namespace

struct gs_comm {
    ncclComm_t comm;
    int rank, world;
    double marker_us = -1.0;   // (test hook of ONE-rank communicators, gs_comm_set_marker_us; < 0: off)
};

extern "C" int gs_comm_available(void) { return load_rccl(); }

extern "C" int gs_comm_unique_id(void* id128) {
    GS_CHECK_ARG(id128, "gs_comm_unique_id: null buffer");
    if (int e = load_rccl()) return e;
    static_assert(sizeof(ncclUniqueId) == GS_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    GS_NCCL_OK(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)));
    return 0;
}

extern "C" int gs_comm_init(gs_comm** out, int rank, int world, const void* id128) {
    GS_CHECK_ARG(out && id128 && world >= 1 && rank >= 0 && rank < world, "gs_comm_init: bad arguments (rank %d of %d)", rank, world);
    if (int e = load_rccl()) return e;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c;
    GS_NCCL_OK(g_rccl.CommInitRank(&c, world, id, rank));   // (binds to the calling thread's current HIP device)
    gs_comm* p = new gs_comm();
    p->comm = c; p->rank = rank; p->world = world;
    *out = p;
    return 0;
}

extern "C" int gs_comm_count(gs_comm* c, int* ranks) {
    GS_CHECK_ARG(c && ranks, "gs_comm_count: bad arguments");
    GS_NCCL_OK(g_rccl.CommCount(c->comm, ranks));   // what the communicator itself says, not what the launcher's environment says
    return 0;
}

extern "C" int gs_comm_destroy(gs_comm* c) {
    if (!c) return 0;
    if (g_rccl.lib) g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}

// World size 1 is the only one the development box has, and RCCL short-cuts a one-rank in-place all-reduce to nothing -- no node in a
// captured graph, nothing to see in a timeline.  gs_comm_set_marker_us (tests / profiles only, refused on a communicator with peers) puts a
// stand-in there: one block that occupies the stream for n microseconds (0: a no-op kernel), so that where the collective sits in a graph --
// beside other work, or on the critical path -- shows up as time.  The all-reduce itself reads no environment.
static __global__ void comm_marker_kernel(const float* data, long long ticks) {
    const long long t0 = wall_clock64();   // (100 MHz constant clock)
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (ticks < 0) const_cast<float*>(data)[0] = 0.f;   // (never: keeps the argument alive)
}

extern "C" int gs_comm_set_marker_us(gs_comm* c, double us) {
    GS_CHECK_ARG(c, "gs_comm_set_marker_us: null communicator");
    GS_CHECK_ARG(c->world == 1 || us < 0.0, "gs_comm_set_marker_us: a stand-in only replaces the all-reduce of a ONE-rank communicator (this one has %d)", c->world);
    c->marker_us = us < 0.0 ? -1.0 : (us > 10000.0 ? 10000.0 : us);   // (clamped: a typo cannot park the stream)
    return 0;
}

extern "C" int gs_allreduce_sum_f32(gs_comm* c, float* data, int64_t count, void* stream) {
    GS_CHECK_ARG(c && data && count >= 0, "gs_allreduce_sum_f32: bad arguments");
    if (count == 0) return 0;
    if (c->world == 1 && c->marker_us >= 0.0) {
        hipLaunchKernelGGL(comm_marker_kernel, dim3(1), dim3(64), 0, gs::as_stream(stream), data, (long long)(c->marker_us * 100.0));
        GS_CHECK_LAUNCH();
        return 0;
    }
    GS_NCCL_OK(g_rccl.AllReduce(data, data, (size_t)count, ncclFloat32, ncclSum, c->comm, gs::as_stream(stream)));
    return 0;
}

extern "C" int gs_broadcast_f32(gs_comm* c, float* data, int64_t count, int root, void* stream) {
    GS_CHECK_ARG(c && data && count >= 0 && root >= 0 && root < c->world, "gs_broadcast_f32: bad arguments");
    if (count == 0) return 0;
    GS_NCCL_OK(g_rccl.Broadcast(data, data, (size_t)count, ncclFloat32, root, c->comm, gs::as_stream(stream)));
    return 0;
}
