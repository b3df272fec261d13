// Multi-GPU helpers of the C-ABI: one process per GPU, the output grid cut into contiguous row tiles, the
// observation block of a step broadcast from rank 0, halo rows of the neighbourhood filters exchanged between neighbouring
// ranks -- RCCL over xGMI, called directly (ncclBroadcast / ncclSend / ncclRecv on the library's stream), so that a C++ caller
// of host/gridpp.hpp tiles a call over the GPUs of a node without torch.  RCCL is loaded on first use (dlopen): a single-GPU
// user of the library never maps it.  The reference has no counterpart (its parallelism is OpenMP, oi.cpp:221).
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

using namespace gpp;

namespace {
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;

template <class F> void sym(F& f, const char* name) {
    f = reinterpret_cast<F>(dlsym(g_rccl.so, name));
    if(!f) throw Error{GPP_ERUNTIME, std::string("librccl.so lacks ") + name};
}
void load_rccl() {
    if(g_rccl.so) return;
    g_rccl.so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if(!g_rccl.so) g_rccl.so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if(!g_rccl.so) g_rccl.so = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if(!g_rccl.so) throw Error{GPP_ERUNTIME, std::string("cannot load librccl.so: ") + dlerror()};
    sym(g_rccl.GetUniqueId, "ncclGetUniqueId"); sym(g_rccl.CommInitRank, "ncclCommInitRank"); sym(g_rccl.CommDestroy, "ncclCommDestroy");
    sym(g_rccl.Broadcast, "ncclBroadcast"); sym(g_rccl.Send, "ncclSend"); sym(g_rccl.Recv, "ncclRecv");
    sym(g_rccl.GroupStart, "ncclGroupStart"); sym(g_rccl.GroupEnd, "ncclGroupEnd"); sym(g_rccl.GetErrorString, "ncclGetErrorString");
}
void nccl_check(ncclResult_t r, const char* what) {
    if(r != ncclSuccess) throw Error{GPP_ERUNTIME, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error")};
}
}  // namespace

// rows [row0, row1) of a ny-row grid owned by `rank` of `world` (contiguous, balanced to within one row) -- the partition of
// gridpp_amd.dist.row_tile and bench.py
extern "C" int gpp_row_tile(int ny, int rank, int world, int* row0, int* row1) {
    GPP_TRY
    if(world < 1 || rank < 0 || rank >= world || ny < 0) invalid("gpp_row_tile: rank / world / ny out of range");
    if(!row0 || !row1) invalid("gpp_row_tile: NULL output");
    *row0 = (int)((long)ny * rank / world); *row1 = (int)((long)ny * (rank + 1) / world);
    return GPP_OK;
    GPP_CATCH
}
// rank 0 creates the 128-byte id; the caller hands it to the other processes (file, environment, MPI, socket ...)
extern "C" int gpp_comm_unique_id(char* id128) {
    GPP_TRY
    if(!id128) invalid("gpp_comm_unique_id: NULL id");
    ensure_device();
    load_rccl();
    ncclUniqueId id;
    nccl_check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return GPP_OK;
    GPP_CATCH
}
// collective over the `world` processes; the communicator is bound to the library's device (gpp_set_device first)
extern "C" int gpp_comm_init(int rank, int world, const char* id128) {
    GPP_TRY
    if(world < 1 || rank < 0 || rank >= world) invalid("gpp_comm_init: rank / world out of range");
    if(!id128) invalid("gpp_comm_init: NULL id");
    if(g_comm) invalid("gpp_comm_init: a communicator exists already (gpp_comm_destroy first)");
    ensure_device();
    load_rccl();
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    nccl_check(g_rccl.CommInitRank(&g_comm, world, id, rank), "ncclCommInitRank");
    g_rank = rank; g_world = world;
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_comm_rank(int* rank, int* world) {
    GPP_TRY
    if(!rank || !world) invalid("gpp_comm_rank: NULL output");
    *rank = g_comm ? g_rank : 0; *world = g_comm ? g_world : 1;
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_comm_destroy(void) {
    GPP_TRY
    if(g_comm) { GPP_HIP(hipStreamSynchronize(stream())); nccl_check(g_rccl.CommDestroy(g_comm), "ncclCommDestroy"); g_comm = nullptr; g_rank = 0; g_world = 1; }
    return GPP_OK;
    GPP_CATCH
}
// in-place broadcast of `bytes` bytes of a DEVICE buffer from `root` (the packed observation block of a step: a few hundred KB,
// latency bound), on the library stream: the kernels of the following gpp_* call are ordered behind it without a host wait
extern "C" int gpp_comm_broadcast(void* device_buf, size_t bytes, int root) {
    GPP_TRY
    if(!g_comm) { if(root != 0) invalid("gpp_comm_broadcast: no communicator"); return GPP_OK; }   // single process: nothing to do
    if(!device_buf && bytes) invalid("gpp_comm_broadcast: NULL buffer");
    if(root < 0 || root >= g_world) invalid("gpp_comm_broadcast: root out of range");
    nccl_check(g_rccl.Broadcast(device_buf, device_buf, bytes, ncclChar, root, g_comm, stream()), "ncclBroadcast");
    return GPP_OK;
    GPP_CATCH
}
// the same for a HOST buffer (std::vector data of the C++ caller): staged through a device scratch buffer
extern "C" int gpp_comm_broadcast_host(void* host_buf, size_t bytes, int root) {
    GPP_TRY
    if(!g_comm) { if(root != 0) invalid("gpp_comm_broadcast_host: no communicator"); return GPP_OK; }
    if(!host_buf && bytes) invalid("gpp_comm_broadcast_host: NULL buffer");
    if(root < 0 || root >= g_world) invalid("gpp_comm_broadcast_host: root out of range");
    static thread_local DevBuf<char> scratch;
    char* d = scratch.get(bytes ? bytes : 1);
    if(g_rank == root) GPP_HIP(hipMemcpyAsync(d, host_buf, bytes, hipMemcpyHostToDevice, stream()));
    nccl_check(g_rccl.Broadcast(d, d, bytes, ncclChar, root, g_comm, stream()), "ncclBroadcast");
    if(g_rank != root) GPP_HIP(hipMemcpyAsync(host_buf, d, bytes, hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}
// Halo of a row-tiled field for the neighbourhood filters: `tile` holds this rank's `rows` rows of `row_floats` floats each
// (row_floats = X or X * E); the rank sends its first / last `halfwidth` rows to the ranks above / below and receives theirs.
// `padded` (device, (top + rows + bottom) * row_floats floats with top = halfwidth if rank > 0, bottom = halfwidth if rank <
// world - 1) receives [halo above | tile | halo below]; *top_rows tells where the tile starts.  ncclSend / ncclRecv in one
// group on the library stream; with a single process it is the copy of the tile.
extern "C" int gpp_comm_halo_exchange(const float* tile, int rows, size_t row_floats, int halfwidth, float* padded, int* top_rows) {
    GPP_TRY
    if(rows < 0 || halfwidth < 0) invalid("gpp_comm_halo_exchange: negative size");
    if((!tile || !padded) && rows > 0 && row_floats > 0) invalid("gpp_comm_halo_exchange: NULL buffer");
    ensure_device();
    const int top = (g_comm && g_rank > 0) ? halfwidth : 0, bot = (g_comm && g_rank < g_world - 1) ? halfwidth : 0;
    if(g_comm && g_world > 1 && rows < halfwidth) invalid("gpp_comm_halo_exchange: a row tile must hold at least `halfwidth` rows");
    if(top_rows) *top_rows = top;
    float* const mid = padded + (size_t)top * row_floats;
    if(mid != tile) GPP_HIP(hipMemcpyAsync(mid, tile, (size_t)rows * row_floats * sizeof(float), hipMemcpyDeviceToDevice, stream()));
    if(top || bot) {
        const size_t nh = (size_t)halfwidth * row_floats;
        nccl_check(g_rccl.GroupStart(), "ncclGroupStart");
        if(top) {
            nccl_check(g_rccl.Send(tile, nh, ncclFloat, g_rank - 1, g_comm, stream()), "ncclSend");
            nccl_check(g_rccl.Recv(padded, nh, ncclFloat, g_rank - 1, g_comm, stream()), "ncclRecv");
        }
        if(bot) {
            nccl_check(g_rccl.Send(tile + (size_t)(rows - halfwidth) * row_floats, nh, ncclFloat, g_rank + 1, g_comm, stream()), "ncclSend");
            nccl_check(g_rccl.Recv(mid + (size_t)rows * row_floats, nh, ncclFloat, g_rank + 1, g_comm, stream()), "ncclRecv");
        }
        nccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
    }
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}
