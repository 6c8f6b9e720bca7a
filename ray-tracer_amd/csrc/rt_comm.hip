// Multi-GPU entry points of the C ABI: one process per GPU, RCCL over xGMI.
//
// The reference renders on one GPU (one dispatch covers the frame, src/renderer.rs:865-867); nothing here has a
// counterpart there. A frame shards by image tile (rt_shard), so the path has exactly two exchanges:
//   scene        rank `root` holds the host arrays; every rank ends up with them in its scene buffers. The bytes cross
//                PCIe once (on the root) and then travel GPU to GPU: ncclBroadcast writes straight into the buffers
//                the unpack kernel reads, no host copy on the receiving ranks.
//   framebuffer  every rank's packed tiles go to `root`: grouped ncclSend / ncclRecv, i.e. N-1 point-to-point
//                transfers that use the N-1 xGMI links into the root concurrently - no ring, no all-reduce.
// Both are queued on the context's current stream like a kernel launch; nothing blocks the host except where the
// scene's sizes have to be known (scene load, not the frame path).
//
// RCCL is bound at run time (dlopen), not at link time: a single-GPU caller needs no RCCL at all, and a process that
// already holds an RCCL (PyTorch ships its own copy next to its own HIP runtime) must use THAT copy - a second one
// would bring a second HIP runtime into the process.
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>

#include <rccl/rccl.h>

#include "rt_internal.hpp"

using namespace rt;

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

RcclApi g_rccl;
std::mutex g_rccl_lock;

// nullptr when RCCL cannot be bound; g_rccl.error says why.
const RcclApi* rccl() {
    std::lock_guard<std::mutex> hold(g_rccl_lock);
    if (g_rccl.handle) return &g_rccl;
    if (!g_rccl.error.empty()) return nullptr;
    // a copy that is already in the process first (RTLD_NOLOAD), then the system's. RT_RCCL_LIBRARY names the one
    // library to bind instead (a site's own build; the tests use it to reach the not-found path).
    const struct {
        const char* name;
        int flags;
    } tries[] = {{"librccl.so", RTLD_NOW | RTLD_NOLOAD},   {"librccl.so.1", RTLD_NOW | RTLD_NOLOAD},
                 {"librccl.so.1", RTLD_NOW | RTLD_GLOBAL}, {"librccl.so", RTLD_NOW | RTLD_GLOBAL},
                 {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL}};
    void* h = nullptr;
    const char* only = getenv("RT_RCCL_LIBRARY");
    if (only && *only) {
        h = dlopen(only, RTLD_NOW | RTLD_GLOBAL);
    } else {
        for (const auto& t : tries)
            if ((h = dlopen(t.name, t.flags))) break;
    }
    if (!h) {
        const char* why = dlerror();  // one call: it hands the message over and clears it
        g_rccl.error = std::string("RCCL not found (") + (only && *only ? only : "librccl.so / librccl.so.1") + "): " + (why ? why : "");
        return nullptr;
    }
    bool ok = true;
    auto bind = [&](auto& fn, const char* name) {
        fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, name));
        if (!fn) {
            ok = false;
            g_rccl.error = std::string("RCCL lacks ") + name;
        }
    };
    bind(g_rccl.GetUniqueId, "ncclGetUniqueId");
    bind(g_rccl.CommInitRank, "ncclCommInitRank");
    bind(g_rccl.CommDestroy, "ncclCommDestroy");
    bind(g_rccl.Broadcast, "ncclBroadcast");
    bind(g_rccl.Send, "ncclSend");
    bind(g_rccl.Recv, "ncclRecv");
    bind(g_rccl.GroupStart, "ncclGroupStart");
    bind(g_rccl.GroupEnd, "ncclGroupEnd");
    bind(g_rccl.GetErrorString, "ncclGetErrorString");
    if (!ok) return nullptr;
    g_rccl.handle = h;
    return &g_rccl;
}

#define RT_NCCL(c, api, expr)                                                                              \
    do {                                                                                                   \
        ncclResult_t r__ = (expr);                                                                         \
        if (r__ != ncclSuccess)                                                                            \
            return rt::fail((c), RT_ERR_COMM, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r__), \
                            __FILE__, __LINE__);                                                           \
    } while (0)

int with_comm(rt_ctx* c, const char* who, const RcclApi** api) {
    if (!c) return RT_ERR_INVALID;
    c->err.clear();
    RT_HIP(c, hipSetDevice(c->device));
    if (!c->comm) return fail(c, RT_ERR_STATE, "%s: no communicator (call rt_comm_init)", who);
    *api = rccl();
    if (!*api) return fail(c, RT_ERR_COMM, "%s: %s", who, g_rccl.error.c_str());
    return RT_OK;
}

}  // namespace

static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "RT_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

extern "C" int rt_comm_unique_id(uint8_t* id) {
    if (!id) return RT_ERR_INVALID;
    const RcclApi* api = rccl();
    if (!api) return RT_ERR_COMM;
    ncclUniqueId u;
    if (api->GetUniqueId(&u) != ncclSuccess) return RT_ERR_COMM;
    memcpy(id, &u, sizeof(u));
    return RT_OK;
}

extern "C" int rt_comm_init(rt_ctx* c, const uint8_t* id, int rank, int n_ranks) {
    if (!c) return RT_ERR_INVALID;
    c->err.clear();
    RT_HIP(c, hipSetDevice(c->device));
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, RT_ERR_INVALID, "rt_comm_init: bad id, rank %d or size %d", rank, n_ranks);
    if (c->comm) return fail(c, RT_ERR_STATE, "rt_comm_init: this context already has a communicator");
    const RcclApi* api = rccl();
    if (!api) return fail(c, RT_ERR_COMM, "rt_comm_init: %s", g_rccl.error.c_str());
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    RT_NCCL(c, api, api->CommInitRank(&comm, n_ranks, u, rank));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_size = n_ranks;
    return RT_OK;
}

extern "C" int rt_comm_destroy(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    c->err.clear();
    if (!c->comm) return RT_OK;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipDeviceSynchronize());  // nothing queued may still use the communicator
    const RcclApi* api = rccl();
    if (api) (void)api->CommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    c->comm_size = 0;
    return RT_OK;
}

extern "C" int rt_comm_info(rt_ctx* c, int* rank, int* n_ranks) {
    if (!c) return RT_ERR_INVALID;
    c->err.clear();
    if (!c->comm) return fail(c, RT_ERR_STATE, "rt_comm_info: no communicator (call rt_comm_init)");
    if (rank) *rank = c->comm_rank;
    if (n_ranks) *n_ranks = c->comm_size;
    return RT_OK;
}

// A collective: every rank must reach the first broadcast whatever its own arguments look like, or the others wait in it
// for ever. So the root's argument checks do not return early: a root that cannot go on says so IN the counts word
// ({0xFFFFFFFF, -status}), and every rank returns that status after the broadcast.
extern "C" int rt_broadcast_scene(rt_ctx* c, int root, const BoxData* boxes, uint32_t nb, const TriangleData* tris, uint32_t nt) {
    const RcclApi* api = nullptr;
    if (int r = with_comm(c, "rt_broadcast_scene", &api)) return r;
    if (root < 0 || root >= c->comm_size) return fail(c, RT_ERR_INVALID, "rt_broadcast_scene: root %d of %d ranks", root, c->comm_size);
    const bool is_root = c->comm_rank == root;
    int root_status = RT_OK;
    if (is_root && ((nb && !boxes) || (nt && !tris))) root_status = RT_ERR_INVALID;
    if (is_root && (uint64_t)nb + nt >= 0xFFFFFFF0ull) root_status = RT_ERR_LIMIT;
    if (int r = begin_scene(c)) root_status = is_root ? r : root_status;
    ncclComm_t comm = static_cast<ncclComm_t>(c->comm);
    // sizes first: two words, device to device, read back by the ranks that do not know them yet
    RT_HIP(c, c->stats.reserve(16 * sizeof(unsigned long long)));
    uint32_t counts[2] = {nb, nt};
    if (root_status != RT_OK) counts[0] = 0xFFFFFFFFu, counts[1] = (uint32_t)(-root_status);
    if (is_root) RT_HIP(c, hipMemcpyAsync(c->stats.p, counts, sizeof(counts), hipMemcpyHostToDevice, c->stream));
    RT_NCCL(c, api, api->Broadcast(c->stats.p, c->stats.p, sizeof(counts), ncclUint8, root, comm, c->stream));
    if (!is_root) {
        RT_HIP(c, hipMemcpyAsync(counts, c->stats.p, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
        nb = counts[0];
        nt = counts[1];
    }
    if (counts[0] == 0xFFFFFFFFu) {
        const int status = -(int)counts[1];
        if (!is_root) return fail(c, status, "rt_broadcast_scene: rank %d (the root) could not send its scene (status %d)", root, status);
        if (status == RT_ERR_LIMIT) return fail(c, status, "rt_broadcast_scene: more than 2^32-16 objects");
        if (status == RT_ERR_INVALID) return fail(c, status, "rt_broadcast_scene: NULL array with non-zero count");
        return status;  // begin_scene's own message stands
    }
    if (int r = reserve_scene(c, nb, nt)) return r;
    // the root's arrays cross PCIe once, into the very buffers the broadcast then replicates
    if (is_root && nb) RT_HIP(c, hipMemcpyAsync(c->boxes.p, boxes, (size_t)nb * sizeof(BoxData), hipMemcpyHostToDevice, c->stream));
    if (is_root && nt) RT_HIP(c, hipMemcpyAsync(c->misc.p, tris, (size_t)nt * sizeof(TriangleData), hipMemcpyHostToDevice, c->stream));
    if (nb) RT_NCCL(c, api, api->Broadcast(c->boxes.p, c->boxes.p, (size_t)nb * sizeof(BoxData), ncclUint8, root, comm, c->stream));
    if (nt) RT_NCCL(c, api, api->Broadcast(c->misc.p, c->misc.p, (size_t)nt * sizeof(TriangleData), ncclUint8, root, comm, c->stream));
    return finish_scene(c, nb, nt);
}

extern "C" int rt_gather_frames(rt_ctx* c, int root, const void* local, size_t bytes, void* gathered) {
    const RcclApi* api = nullptr;
    if (int r = with_comm(c, "rt_gather_frames", &api)) return r;
    if (root < 0 || root >= c->comm_size) return fail(c, RT_ERR_INVALID, "rt_gather_frames: root %d of %d ranks", root, c->comm_size);
    if (bytes == 0) return RT_OK;
    const bool is_root = c->comm_rank == root;
    if (!local || (is_root && !gathered)) return fail(c, RT_ERR_INVALID, "rt_gather_frames: NULL buffer");
    if (!is_device_ptr(local) || (is_root && !is_device_ptr(gathered)))
        return fail(c, RT_ERR_INVALID, "rt_gather_frames: buffers must be device memory");
    ncclComm_t comm = static_cast<ncclComm_t>(c->comm);
    if (!is_root) {
        RT_NCCL(c, api, api->Send(local, bytes, ncclUint8, root, comm, c->stream));
        return RT_OK;
    }
    uint8_t* out = static_cast<uint8_t*>(gathered);
    if (c->comm_size > 1) {
        // one group: the N-1 receives progress together, each over its own link into this GPU
        RT_NCCL(c, api, api->GroupStart());
        for (int r = 0; r < c->comm_size; ++r) {
            if (r == root) continue;
            const ncclResult_t e = api->Recv(out + (size_t)r * bytes, bytes, ncclUint8, r, comm, c->stream);
            if (e != ncclSuccess) {
                (void)api->GroupEnd();
                return fail(c, RT_ERR_COMM, "ncclRecv from rank %d failed: %s", r, api->GetErrorString(e));
            }
        }
        RT_NCCL(c, api, api->GroupEnd());
    }
    if (out + (size_t)root * bytes != local)
        RT_HIP(c, hipMemcpyAsync(out + (size_t)root * bytes, local, bytes, hipMemcpyDeviceToDevice, c->stream));
    return RT_OK;
}
