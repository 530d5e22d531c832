// rpt_comm.cpp — the frame exchange between the GPUs of one node behind the C ABI (include/rpt_hip.h, "frame exchange").
//
// The reference parallelises Renderer::sample over image rows on one host (rayon, src/renderer.rs:158-171); here pixel
// tiles are sharded over one process per GPU and rank 0 assembles the frame.  What crosses xGMI is the set of tiles a
// rank owns, packed, f64: one ncclSend per rank and one ncclGroup of ncclRecv on rank 0 -- a gather, not a reduce of
// zero-padded full frames.  RCCL is bound at run time with dlopen/dlsym, so librpt_hip.so does not depend on it and a
// process that already holds a copy (PyTorch bundles one under the same SONAME) shares that copy instead of loading
// a second one.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>   // types and prototypes only: nothing here links against librccl

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "host_internal.h"

using rpti::fail;

namespace {
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    // the copy already in the process first (RTLD_NOLOAD), then the loader's search path, then the ROCm tree
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW);
    if (!h) {
        const char* why = dlerror();   // (read once: dlerror() clears the message it returns)
        g_rccl.error = std::string("librccl.so.1 cannot be loaded: ") + (why ? why : "?");
        return;
    }
    auto sym = [&](const char* name) -> void* {
        void* p = dlsym(h, name);
        if (!p && g_rccl.error.empty()) g_rccl.error = std::string("librccl.so.1 lacks ") + name;
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (g_rccl.error.empty()) g_rccl.handle = h;
}
int need_rccl() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.handle) return fail(RPT_ERR_UNSUPPORTED, g_rccl.error);
    return RPT_OK;
}
#define RCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r__ = (expr);                                                                             \
        if (r__ != ncclSuccess) return fail(RPT_ERR_DEVICE, std::string(#expr) + ": " + g_rccl.GetErrorString(r__)); \
    } while (0)

constexpr uint64_t kTileDoubles = 32u * 32u * 3u;

// Tile lists of every rank for one frame size: rank r's tiles are [offsets[r], offsets[r + 1]) of `all`.
struct TileLayout {
    uint32_t width = 0, height = 0, n_ranks = 0, tiles_x = 0;
    std::vector<uint32_t> all;
    std::vector<uint64_t> offsets;
};
int build_layout(uint32_t width, uint32_t height, uint32_t n_ranks, TileLayout& L) {
    if (width == 0 || height == 0 || n_ranks == 0) return fail(RPT_ERR_INVALID, "bad frame layout arguments");
    L.width = width; L.height = height; L.n_ranks = n_ranks; L.tiles_x = (width + 31u) / 32u;
    const uint64_t n_tiles = uint64_t(L.tiles_x) * ((height + 31u) / 32u);
    L.all.assign(n_tiles, 0u);
    L.offsets.assign(size_t(n_ranks) + 1u, 0u);
    uint64_t at = 0;
    for (uint32_t r = 0; r < n_ranks; r++) {
        L.offsets[r] = at;
        const int64_t n = rpt_shard_tiles(width, height, r, n_ranks, L.all.data() + at, n_tiles - at);
        if (n < 0) return int(n);
        at += uint64_t(n);
    }
    L.offsets[n_ranks] = at;
    return RPT_OK;
}
}  // namespace

struct rpt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1, device = 0;
    // per frame size: the tile lists on the device and the packed staging buffer
    TileLayout layout;
    uint32_t* d_tiles = nullptr;   // layout.all
    double* d_stage = nullptr;     // rank 0: every rank's block (whole frame worth of tiles); others: their own block
    size_t stage_cap = 0, tiles_cap = 0;
    // photon records: counts of every rank, and the padded blocks of the all-gather
    unsigned long long* d_counts = nullptr;
    void* d_rec_stage = nullptr;
    size_t rec_stage_cap = 0;
};

extern "C" {

int rpt_comm_unique_id(void* id_out) {
    if (!id_out) return fail(RPT_ERR_INVALID, "null id");
    if (int rc = need_rccl()) return rc;
    static_assert(sizeof(ncclUniqueId) == RPT_COMM_ID_BYTES, "RPT_COMM_ID_BYTES is the size of an ncclUniqueId");
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return RPT_OK;
}

int rpt_comm_create(const void* id, int rank, int n_ranks, int device, rpt_comm** out) {
    if (!id || !out) return fail(RPT_ERR_INVALID, "null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RPT_ERR_INVALID, "rank outside [0, n_ranks)");
    if (int rc = need_rccl()) return rc;
    RPTI_HIP_TRY(hipSetDevice(device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&c, n_ranks, uid, rank));
    rpt_comm* cm = new rpt_comm;
    cm->comm = c; cm->rank = rank; cm->n_ranks = n_ranks; cm->device = device;
    *out = cm;
    return RPT_OK;
}

void rpt_comm_destroy(rpt_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_tiles) (void)hipFree(c->d_tiles);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->d_rec_stage) (void)hipFree(c->d_rec_stage);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

int rpt_comm_rank(const rpt_comm* c, int* rank, int* n_ranks) {
    if (!c) return fail(RPT_ERR_INVALID, "null communicator");
    if (rank) *rank = c->rank;
    if (n_ranks) *n_ranks = c->n_ranks;
    return RPT_OK;
}

int rpt_frame_pack_layout(uint32_t width, uint32_t height, uint32_t n_ranks, uint64_t* tile_offsets) {
    if (!tile_offsets) return fail(RPT_ERR_INVALID, "null output");
    TileLayout L;
    if (int rc = build_layout(width, height, n_ranks, L)) return rc;
    for (uint32_t r = 0; r <= n_ranks; r++) tile_offsets[r] = L.offsets[r];
    return RPT_OK;
}

}  // extern "C"

// One rank's tile list on the device (uncached: the test hooks below are not on any hot path).
static int with_rank_tiles(uint32_t width, uint32_t height, uint32_t rank, uint32_t n_ranks, hipStream_t st,
                           const std::function<hipError_t(const uint32_t*, uint32_t, uint32_t)>& fn) {
    if (n_ranks == 0 || rank >= n_ranks) return fail(RPT_ERR_INVALID, "rank outside [0, n_ranks)");
    TileLayout L;
    if (int rc = build_layout(width, height, n_ranks, L)) return rc;
    const uint32_t n = uint32_t(L.offsets[rank + 1] - L.offsets[rank]);
    if (n == 0) return RPT_OK;
    uint32_t* d = nullptr;
    RPTI_HIP_TRY(hipMalloc((void**)&d, size_t(n) * 4));
    hipError_t e = hipMemcpyAsync(d, L.all.data() + L.offsets[rank], size_t(n) * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);   // (the host vector goes away with this frame)
    if (e == hipSuccess) e = fn(d, n, L.tiles_x);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RPT_ERR_DEVICE, std::string("frame tiles: ") + hipGetErrorString(e));
    return RPT_OK;
}

extern "C" {

int rpt_frame_pack_device(uint32_t width, uint32_t height, uint32_t rank, uint32_t n_ranks, const void* d_frame, void* d_packed,
                          void* hip_stream) {
    if (!d_frame || !d_packed) return fail(RPT_ERR_INVALID, "null device pointer");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    return with_rank_tiles(width, height, rank, n_ranks, st, [&](const uint32_t* t, uint32_t n, uint32_t tiles_x) {
        return rptg::launch_frame_pack(static_cast<const double*>(d_frame), static_cast<double*>(d_packed), t, n, tiles_x, width, height, st);
    });
}

int rpt_frame_unpack_device(uint32_t width, uint32_t height, uint32_t rank, uint32_t n_ranks, const void* d_packed, void* d_frame,
                            void* hip_stream) {
    if (!d_frame || !d_packed) return fail(RPT_ERR_INVALID, "null device pointer");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    return with_rank_tiles(width, height, rank, n_ranks, st, [&](const uint32_t* t, uint32_t n, uint32_t tiles_x) {
        return rptg::launch_frame_unpack(static_cast<const double*>(d_packed), static_cast<double*>(d_frame), t, n, tiles_x, width, height, st);
    });
}

int rpt_gather_frame_device(rpt_comm* c, uint32_t width, uint32_t height, const void* d_shard, void* d_frame, uint32_t flags,
                            void* hip_stream) {
    if (!c || !d_shard) return fail(RPT_ERR_INVALID, "null argument");
    if (c->rank == 0 && !d_frame) return fail(RPT_ERR_INVALID, "rank 0 needs the frame to assemble into");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    RPTI_HIP_TRY(hipSetDevice(c->device));
    if (c->layout.width != width || c->layout.height != height || c->layout.n_ranks != uint32_t(c->n_ranks)) {
        // a new frame size: tile lists to the device, staging buffer (rank 0: the whole frame's tiles).  The cached layout is
        // replaced only once every allocation and upload has succeeded: a failure below leaves it invalid, so that the next
        // call with the same size sets everything up again instead of launching on buffers that were never filled.
        RPTI_HIP_TRY(hipStreamSynchronize(st));   // nothing in flight may still read the old lists
        c->layout.width = 0;
        TileLayout fresh;
        if (int rc = build_layout(width, height, uint32_t(c->n_ranks), fresh)) return rc;
        if (fresh.all.size() > c->tiles_cap) {
            if (c->d_tiles) RPTI_HIP_TRY(hipFree(c->d_tiles));
            c->d_tiles = nullptr; c->tiles_cap = 0;
            RPTI_HIP_TRY(hipMalloc((void**)&c->d_tiles, fresh.all.size() * 4));
            c->tiles_cap = fresh.all.size();
        }
        RPTI_HIP_TRY(hipMemcpy(c->d_tiles, fresh.all.data(), fresh.all.size() * 4, hipMemcpyHostToDevice));
        // rank 0: every rank's block + room to receive its own once more (loopback); others: their own block
        const uint64_t tiles = c->rank == 0 ? fresh.offsets[fresh.n_ranks] + fresh.offsets[1] : fresh.offsets[c->rank + 1] - fresh.offsets[c->rank];
        const size_t bytes = std::max<size_t>(size_t(tiles) * kTileDoubles * 8, 8);
        if (bytes > c->stage_cap) {
            if (c->d_stage) RPTI_HIP_TRY(hipFree(c->d_stage));
            c->d_stage = nullptr; c->stage_cap = 0;
            RPTI_HIP_TRY(hipMalloc((void**)&c->d_stage, bytes));
            c->stage_cap = bytes;
        }
        c->layout = std::move(fresh);
    }
    TileLayout& L = c->layout;
    const bool loop = (flags & RPT_GATHER_LOOPBACK) != 0;
    const double* shard = static_cast<const double*>(d_shard);
    double* frame = static_cast<double*>(d_frame);
    auto n_of = [&](int r) { return uint32_t(L.offsets[r + 1] - L.offsets[r]); };
    if (c->rank != 0) {
        const uint32_t n = n_of(c->rank);
        RPTI_HIP_TRY(rptg::launch_frame_pack(shard, c->d_stage, c->d_tiles + L.offsets[c->rank], n, L.tiles_x, width, height, st));
        if (n) RCCL_TRY(g_rccl.Send(c->d_stage, size_t(n) * kTileDoubles, ncclDouble, 0, c->comm, st));
        return RPT_OK;
    }
    // rank 0: its own tiles are in place already (or are copied tile by tile, or -- loopback -- travel like the others)
    if (loop) RPTI_HIP_TRY(rptg::launch_frame_pack(shard, c->d_stage, c->d_tiles, n_of(0), L.tiles_x, width, height, st));
    const bool any_transfer = c->n_ranks > 1 || (loop && n_of(0));
    if (any_transfer) {
        // loopback: rank 0's packed block (the start of the staging buffer) is sent to itself and received behind the last block
        double* const self_recv = c->d_stage + L.offsets[L.n_ranks] * kTileDoubles;
        RCCL_TRY(g_rccl.GroupStart());
        ncclResult_t bad = ncclSuccess;
        if (loop && n_of(0)) {
            ncclResult_t r1 = g_rccl.Send(c->d_stage, size_t(n_of(0)) * kTileDoubles, ncclDouble, 0, c->comm, st);
            ncclResult_t r2 = g_rccl.Recv(self_recv, size_t(n_of(0)) * kTileDoubles, ncclDouble, 0, c->comm, st);
            if (r1 != ncclSuccess) bad = r1; else if (r2 != ncclSuccess) bad = r2;
        }
        for (int r = 1; r < c->n_ranks; r++) {
            if (!n_of(r)) continue;
            ncclResult_t rr = g_rccl.Recv(c->d_stage + L.offsets[r] * kTileDoubles, size_t(n_of(r)) * kTileDoubles, ncclDouble, r, c->comm, st);
            if (rr != ncclSuccess && bad == ncclSuccess) bad = rr;
        }
        RCCL_TRY(g_rccl.GroupEnd());
        if (bad != ncclSuccess) return fail(RPT_ERR_DEVICE, std::string("ncclSend/ncclRecv: ") + g_rccl.GetErrorString(bad));
        if (loop) RPTI_HIP_TRY(rptg::launch_frame_unpack(self_recv, frame, c->d_tiles, n_of(0), L.tiles_x, width, height, st));
        // every other rank's block with one launch: their tiles are contiguous in the list and in the staging buffer
        const uint32_t n_others = uint32_t(L.offsets[L.n_ranks] - L.offsets[1]);
        RPTI_HIP_TRY(rptg::launch_frame_unpack(c->d_stage + L.offsets[1] * kTileDoubles, frame, c->d_tiles + L.offsets[1], n_others, L.tiles_x,
                                                width, height, st));
    }
    if (!loop && frame != shard) {   // own tiles into a separate frame: pack + unpack on the device
        RPTI_HIP_TRY(rptg::launch_frame_pack(shard, c->d_stage, c->d_tiles, n_of(0), L.tiles_x, width, height, st));
        RPTI_HIP_TRY(rptg::launch_frame_unpack(c->d_stage, frame, c->d_tiles, n_of(0), L.tiles_x, width, height, st));
    }
    return RPT_OK;
}

// The photon maps' exchange step (src/photon.rs:656-690 sharded by photon index): every rank needs every rank's records, in
// rank order.  Two ncclAllGathers -- the counts, then blocks padded to the largest count -- and one device copy per rank
// that closes the gaps.  The counts come to the host in between (the second collective is sized by them).  With d_out = NULL and
// capacity = 0 only the counts are exchanged (n_per_rank, n_total): how a caller learns the size of the buffer to bring.
int rpt_allgather_records_device(rpt_comm* c, const void* d_local, uint64_t n_local, void* d_out, uint64_t capacity,
                                 uint64_t* n_per_rank, uint64_t* n_total, void* hip_stream) {
    if (!c || !n_total) return fail(RPT_ERR_INVALID, "null argument");
    if (n_local && !d_local) return fail(RPT_ERR_INVALID, "null records");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    RPTI_HIP_TRY(hipSetDevice(c->device));
    const size_t n = size_t(c->n_ranks);
    if (!c->d_counts) RPTI_HIP_TRY(hipMalloc((void**)&c->d_counts, (n + 1) * sizeof(unsigned long long)));
    // [n]: this rank's count (send buffer), [0, n): everybody's
    unsigned long long mine = n_local;
    RPTI_HIP_TRY(hipMemcpyAsync(c->d_counts + n, &mine, sizeof(mine), hipMemcpyHostToDevice, st));
    RCCL_TRY(g_rccl.AllGather(c->d_counts + n, c->d_counts, 1, ncclUint64, c->comm, st));
    std::vector<unsigned long long> counts(n);
    RPTI_HIP_TRY(hipMemcpyAsync(counts.data(), c->d_counts, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    RPTI_HIP_TRY(hipStreamSynchronize(st));
    uint64_t total = 0, largest = 0;
    for (size_t r = 0; r < n; r++) {
        if (n_per_rank) n_per_rank[r] = counts[r];
        total += counts[r];
        largest = std::max<uint64_t>(largest, counts[r]);
    }
    *n_total = total;
    if (!d_out && capacity == 0) return RPT_OK;   // a counts-only call: the caller sizes its buffer from n_total and calls again (every rank takes this exit together)
    if (total > capacity) return fail(RPT_ERR_INVALID, "rpt_allgather_records_device: the output holds fewer records than the ranks have shot");
    if (total == 0) return RPT_OK;
    if (!d_out) return fail(RPT_ERR_INVALID, "null output");
    const size_t block = size_t(largest) * RPT_PHOTON_RECORD_BYTES;
    const size_t need = block * (n + 1);   // the padded copy of this rank's records + every rank's block
    if (need > c->rec_stage_cap) {
        if (c->d_rec_stage) RPTI_HIP_TRY(hipFree(c->d_rec_stage));
        c->d_rec_stage = nullptr; c->rec_stage_cap = 0;
        RPTI_HIP_TRY(hipMalloc(&c->d_rec_stage, need));
        c->rec_stage_cap = need;
    }
    char* const send = static_cast<char*>(c->d_rec_stage), *const recv = send + block;
    if (n_local) RPTI_HIP_TRY(hipMemcpyAsync(send, d_local, size_t(n_local) * RPT_PHOTON_RECORD_BYTES, hipMemcpyDeviceToDevice, st));
    RCCL_TRY(g_rccl.AllGather(send, recv, block, ncclUint8, c->comm, st));
    size_t at = 0;
    for (size_t r = 0; r < n; r++) {
        const size_t bytes = size_t(counts[r]) * RPT_PHOTON_RECORD_BYTES;
        if (bytes) RPTI_HIP_TRY(hipMemcpyAsync(static_cast<char*>(d_out) + at, recv + r * block, bytes, hipMemcpyDeviceToDevice, st));
        at += bytes;
    }
    return RPT_OK;
}

}  // extern "C"
