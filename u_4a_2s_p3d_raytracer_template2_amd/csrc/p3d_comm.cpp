// p3d_comm.cpp -- the one collective of the multi-GPU frame behind the C-ABI: every
// rank's compact shard of row blocks is moved to rank 0 over RCCL (grouped ncclSend / ncclRecv: direct
// peer -> root transfers, one xGMI link per peer, not a ring), where p3d_deinterleave() restores row
// order.  Two ways to form the communicator, same gather:
//   * one process per GPU (torchrun, MPI, ...): p3d_comm_unique_id() on rank 0, the 128 bytes carried to
//     the other ranks by whatever the launcher offers, p3d_comm_create() everywhere;
//   * one process driving all GPUs (the C++ host layer / p3d_render --gpus N): p3d_comm_create_all().
// The reference has no counterpart (single-threaded CPU loop, RT/main.cpp:749-805); the call site this
// serves is main()'s "renderScene(); save image" (RT/main.cpp:966-970).
#include "p3d_hip.h"
#include "p3d_pathtracer.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

extern "C" int p3d_internal_set_error(int code, const char* msg);
extern "C" int p3d_internal_scene_binding(p3d_scene* s, int* device, void** stream);
extern "C" int p3d_internal_pt_binding(p3d_pt* h, int* device, void** stream);

struct p3d_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

namespace {

int fail(int code, const std::string& msg) { return p3d_internal_set_error(code, msg.c_str()); }

#define NCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess) return fail(P3D_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail(P3D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// this rank's part of the gather, to be called inside an open group when several ranks share a thread
int enqueue_gather(p3d_comm* c, hipStream_t stream, const void* tile, void* gathered, uint64_t tile_bytes) {
    if (c->rank == 0) {
        for (int r = 1; r < c->world; r++)
            NCCL_TRY(ncclRecv((char*)gathered + (size_t)r * tile_bytes, (size_t)tile_bytes, ncclUint8, r, c->comm, stream));
    } else {
        NCCL_TRY(ncclSend(tile, (size_t)tile_bytes, ncclUint8, 0, c->comm, stream));
    }
    return P3D_OK;
}

int check_gather_args(p3d_comm* c, p3d_scene* s, const void* tile, void* gathered, uint64_t tile_bytes,
                      int* device, void** stream) {
    if (!c || !s) return fail(P3D_ERR_ARG, "comm/scene is NULL");
    if (!tile) return fail(P3D_ERR_ARG, "tile is NULL");
    if (tile_bytes == 0) return fail(P3D_ERR_ARG, "tile_bytes is 0");
    if (c->rank == 0 && !gathered) return fail(P3D_ERR_ARG, "rank 0 needs the gathered buffer");
    int rc = p3d_internal_scene_binding(s, device, stream);
    if (rc) return rc;
    if (*device != c->device) return fail(P3D_ERR_ARG, "scene and communicator are bound to different devices");
    return P3D_OK;
}

}  // namespace

extern "C" {

int p3d_comm_unique_id(void* id_out) {
    if (!id_out) return fail(P3D_ERR_ARG, "id_out is NULL");
    static_assert(P3D_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id_out, id.internal, P3D_COMM_ID_BYTES);
    return P3D_OK;
}

int p3d_comm_create(const void* id_bytes, int rank, int world, int device, p3d_comm** out) {
    if (!out) return fail(P3D_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (world < 1 || world > 64) return fail(P3D_ERR_ARG, "world must be in 1..64");
    if (rank < 0 || rank >= world) return fail(P3D_ERR_ARG, "rank outside [0, world)");
    if (!id_bytes) return fail(P3D_ERR_ARG, "id is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(P3D_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(P3D_ERR_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(id.internal, id_bytes, P3D_COMM_ID_BYTES);
    p3d_comm* c = new p3d_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return fail(P3D_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    *out = c;
    return P3D_OK;
}

int p3d_comm_create_all(const int* devices, int n, p3d_comm** out) {
    if (!out) return fail(P3D_ERR_ARG, "out is NULL");
    if (n < 1 || n > 64) return fail(P3D_ERR_ARG, "n must be in 1..64");
    for (int i = 0; i < n; i++) out[i] = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(P3D_ERR_NO_DEVICE, "no HIP device visible");
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) {
        devs[i] = devices ? devices[i] : i;
        if (devs[i] < 0 || devs[i] >= ndev) return fail(P3D_ERR_ARG, "device index out of range");
        for (int j = 0; j < i; j++) if (devs[j] == devs[i]) return fail(P3D_ERR_ARG, "a device is listed twice");
    }
    std::vector<ncclComm_t> comms(n);
    NCCL_TRY(ncclCommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; i++) {
        p3d_comm* c = new p3d_comm();
        c->comm = comms[i]; c->rank = i; c->world = n; c->device = devs[i];
        out[i] = c;
    }
    return P3D_OK;
}

int p3d_comm_destroy(p3d_comm* c) {
    if (!c) return P3D_OK;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
    return P3D_OK;
}

int p3d_comm_info(const p3d_comm* c, int* rank, int* world, int* device) {
    if (!c) return fail(P3D_ERR_ARG, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    return P3D_OK;
}

int p3d_gather(p3d_comm* c, p3d_scene* s, const void* tile, void* gathered, uint64_t tile_bytes) {
    int device = 0; void* stream = nullptr;
    int rc = check_gather_args(c, s, tile, gathered, tile_bytes, &device, &stream);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    if (c->rank == 0 && gathered != tile)
        HIP_TRY(hipMemcpyAsync(gathered, tile, (size_t)tile_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (c->world == 1) return P3D_OK;
    NCCL_TRY(ncclGroupStart());
    rc = enqueue_gather(c, (hipStream_t)stream, tile, gathered, tile_bytes);
    ncclResult_t r = ncclGroupEnd();
    if (rc) return rc;
    if (r != ncclSuccess) return fail(P3D_ERR_COMM, std::string("ncclGroupEnd: ") + ncclGetErrorString(r));
    return P3D_OK;
}

int p3d_gather_all(p3d_comm* const* comms, p3d_scene* const* scenes, const void* const* tiles, int n,
                   void* gathered, uint64_t tile_bytes) {
    if (!comms || !scenes || !tiles) return fail(P3D_ERR_ARG, "NULL argument");
    if (n < 1) return fail(P3D_ERR_ARG, "n must be >= 1");
    std::vector<int> device(n); std::vector<void*> stream(n);
    for (int i = 0; i < n; i++) {
        if (!comms[i]) return fail(P3D_ERR_ARG, "comm is NULL");
        if (comms[i]->rank != i || comms[i]->world != n) return fail(P3D_ERR_ARG, "comms must be ranks 0..n-1 of one group, in order");
        int rc = check_gather_args(comms[i], scenes[i], tiles[i], gathered, tile_bytes, &device[i], &stream[i]);
        if (rc) return rc;
    }
    HIP_TRY(hipSetDevice(device[0]));
    if (gathered != tiles[0])
        HIP_TRY(hipMemcpyAsync(gathered, tiles[0], (size_t)tile_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream[0]));
    if (n == 1) return P3D_OK;
    // one thread, several ranks: all sends and receives inside ONE group, or the first would block
    NCCL_TRY(ncclGroupStart());
    int rc = P3D_OK;
    for (int i = 0; i < n && !rc; i++) {
        if (hipSetDevice(device[i]) != hipSuccess) rc = fail(P3D_ERR_HIP, "hipSetDevice failed");
        else rc = enqueue_gather(comms[i], (hipStream_t)stream[i], tiles[i], gathered, tile_bytes);
    }
    ncclResult_t r = ncclGroupEnd();
    (void)hipSetDevice(device[0]);
    if (rc) return rc;
    if (r != ncclSuccess) return fail(P3D_ERR_COMM, std::string("ncclGroupEnd: ") + ncclGetErrorString(r));
    return P3D_OK;
}

// the path tracer's collective: per-rank sums of linear sample colours -> rank 0
int p3d_pt_reduce_sum(p3d_comm* c, p3d_pt* pt, float* linear, uint64_t count) {
    if (!c || !pt || !linear) return fail(P3D_ERR_ARG, "NULL argument");
    int device = 0; void* stream = nullptr;
    int rc = p3d_internal_pt_binding(pt, &device, &stream);
    if (rc) return rc;
    if (device != c->device) return fail(P3D_ERR_ARG, "handle and communicator are bound to different devices");
    if (c->world == 1 || count == 0) return P3D_OK;
    HIP_TRY(hipSetDevice(device));
    NCCL_TRY(ncclReduce(linear, linear, (size_t)count, ncclFloat, ncclSum, 0, c->comm, (hipStream_t)stream));
    return P3D_OK;
}

}  // extern "C"
