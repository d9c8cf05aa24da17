// One-time weight distribution over RCCL / xGMI (included at the end of engine.cpp).
//
// The hot path has no data-path collective: frames shard across ranks (frame f -> rank f mod R) and every rank
// runs the whole pipeline on its own GPU.  The only exchange is this broadcast of the packed network programs
// from rank 0 at start-up.  For contrast, the reference's only collectives are its DDP training
// all-reduces (TRAIN/face_landmark/lib/core/base_trainer/net_work.py:30,131-137); its inference path
// (Skps/core/api/face_landmark.py:40-48) has no parallelism at all.
//
// librccl is bound lazily with dlopen/dlsym, so single-GPU users need no RCCL at all and a missing library is a
// loud, specific error of THIS call rather than a load failure of the engine.  Only the five entry points used
// here are declared (ABI of rccl.h, ROCm 7.x: ncclUniqueId is a 128-byte struct passed by value).
#include <dlfcn.h>

namespace {

struct PfNcclUid { char internal[128]; };
typedef void* PfNcclComm;
enum { kNcclUint8 = 1 };   // ncclDataType_t

struct RcclApi {
    void* lib = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*GetUniqueId)(PfNcclUid*) = nullptr;
    int (*CommInitRank)(PfNcclComm*, int, PfNcclUid, int) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, PfNcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(PfNcclComm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
    bool ok = false;
};

RcclApi& rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    const char* names[] = {getenv("PEPPA_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
        const char* de = dlerror();
        api.err = de ? de : "";
    }
    if (!api.lib) { api.err = "librccl not found (tried librccl.so.1, librccl.so): " + api.err; return api; }
#define PF_RCCL_SYM(field, name)                                                     \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name));          \
    if (!api.field) { api.err = std::string("librccl lacks symbol ") + name; return api; }
    PF_RCCL_SYM(GetVersion, "ncclGetVersion")
    PF_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    PF_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    PF_RCCL_SYM(Broadcast, "ncclBroadcast")
    PF_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    PF_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef PF_RCCL_SYM
    api.ok = true;
    return api;
}

#define PF_NCCL(h, call)                                                                                     \
    do {                                                                                                     \
        const int _r = (call);                                                                               \
        if (_r != 0) PF_FAIL(h, "%s failed: %s (%s:%d)", #call, rccl_api().GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

void comm_release(pf_handle* h) {
    if (h->comm && rccl_api().ok) (void)rccl_api().CommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_rank = -1;
    h->comm_world = 0;
}

int comm_ensure(pf_handle* h, const void* id, int rank, int world) {
    RcclApi& api = rccl_api();
    if (!api.ok) PF_FAIL(h, "pf_broadcast_weights: %s", api.err.c_str());
    if (h->comm && h->comm_rank == rank && h->comm_world == world && memcmp(h->comm_id, id, sizeof(h->comm_id)) == 0) return 0;
    comm_release(h);
    PfNcclUid uid;
    memcpy(uid.internal, id, sizeof(uid.internal));
    PF_HIP(h, hipSetDevice(h->device));
    PfNcclComm c = nullptr;
    PF_NCCL(h, api.CommInitRank(&c, world, uid, rank));
    h->comm = c;
    memcpy(h->comm_id, id, sizeof(h->comm_id));
    h->comm_rank = rank;
    h->comm_world = world;
    return 0;
}

}  // namespace

extern "C" {

int pf_comm_unique_id(void* id_out, size_t id_bytes) {
    if (!id_out || id_bytes < PF_COMM_ID_BYTES) { g_create_error = "pf_comm_unique_id: need a 128-byte buffer"; return 1; }
    RcclApi& api = rccl_api();
    if (!api.ok) { g_create_error = "pf_comm_unique_id: " + api.err; return 1; }
    PfNcclUid uid;
    memset(&uid, 0, sizeof(uid));
    const int r = api.GetUniqueId(&uid);
    if (r != 0) { g_create_error = std::string("ncclGetUniqueId failed: ") + api.GetErrorString(r); return 1; }
    memcpy(id_out, uid.internal, sizeof(uid.internal));
    return 0;
}

int pf_rccl_version(int* version) {
    RcclApi& api = rccl_api();
    if (!api.ok || !version) { g_create_error = "pf_rccl_version: " + api.err; return 1; }
    return api.GetVersion(version) == 0 ? 0 : 1;
}

int pf_broadcast_weights(pf_handle* h, const void* rccl_unique_id, int rank, int world, int slot,
                         void* blob, size_t capacity, size_t* bytes, int max_batch, float* bcast_ms) {
    if (!h) return 1;
    if (!rccl_unique_id || world < 1 || rank < 0 || rank >= world) PF_FAIL(h, "pf_broadcast_weights: bad rank %d / world %d", rank, world);
    if (slot < 0 || slot >= PF_NET_SLOTS || !blob || !bytes || max_batch < 1) PF_FAIL(h, "pf_broadcast_weights: bad arguments");
    if (rank == 0 && (*bytes < sizeof(PfHeader) || *bytes > capacity)) PF_FAIL(h, "pf_broadcast_weights: root blob size %zu (capacity %zu)", *bytes, capacity);
    PF_HIP(h, hipSetDevice(h->device));
    if (comm_ensure(h, rccl_unique_id, rank, world)) return 1;
    RcclApi& api = rccl_api();
    // 1. size: 8 bytes through the start of the staging buffer (also warms the communicator's channels up)
    if (ensure_stage(h, 256)) return 1;
    unsigned long long sz = rank == 0 ? (unsigned long long)*bytes : 0ull;
    PF_HIP(h, hipMemcpyAsync(h->d_stage, &sz, sizeof(sz), hipMemcpyHostToDevice, h->stream));
    PF_NCCL(h, api.Broadcast(h->d_stage, h->d_stage, sizeof(sz), kNcclUint8, 0, h->comm, h->stream));
    PF_HIP(h, hipMemcpyAsync(&sz, h->d_stage, sizeof(sz), hipMemcpyDeviceToHost, h->stream));
    PF_HIP(h, hipStreamSynchronize(h->stream));
    if (sz < sizeof(PfHeader)) PF_FAIL(h, "pf_broadcast_weights: root announced %llu bytes", sz);
    // A rank whose receive buffer is too small must still take part in the payload broadcast (it lands in the engine's
    // own staging buffer, not in `blob`): returning here would leave the root and the other ranks waiting in a collective
    // this rank never enters.  The call fails on this rank AFTER the exchange; every other rank completes normally.
    const bool too_small = rank != 0 && sz > capacity;
    // 2. payload: rank 0's packed program, HBM to HBM over xGMI, timed with HIP events on the engine's stream
    if (ensure_stage(h, (size_t)sz)) return 1;
    if (rank == 0) PF_HIP(h, hipMemcpyAsync(h->d_stage, blob, (size_t)sz, hipMemcpyHostToDevice, h->stream));
    PF_HIP(h, hipEventRecord(h->ev0, h->stream));
    PF_NCCL(h, api.Broadcast(h->d_stage, h->d_stage, (size_t)sz, kNcclUint8, 0, h->comm, h->stream));
    PF_HIP(h, hipEventRecord(h->ev1, h->stream));
    if (rank != 0 && !too_small) PF_HIP(h, hipMemcpyAsync(blob, h->d_stage, (size_t)sz, hipMemcpyDeviceToHost, h->stream));
    PF_HIP(h, hipStreamSynchronize(h->stream));
    if (too_small) PF_FAIL(h, "pf_broadcast_weights: blob of %llu bytes exceeds the receive capacity %zu", sz, capacity);
    float ms = 0.f;
    PF_HIP(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (bcast_ms) *bcast_ms = ms;
    *bytes = (size_t)sz;
    // 3. every rank (the root included) loads the same bytes
    return pf_load_program(h, slot, blob, (size_t)sz, max_batch);
}

int pf_comm_destroy(pf_handle* h) {
    if (!h) return 1;
    comm_release(h);
    return 0;
}

}  // extern "C"
