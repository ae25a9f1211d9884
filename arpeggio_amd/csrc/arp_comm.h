// arp_comm.h — the exchange between the shards of a distributed structure, behind the C ABI.
//
// the grid is cut into x-slabs, one rank per GPU; neighbours exchange the records of a one-cell halo once
// per structure and, when a selection is given, the selection_plus bits of the halo atoms and one all-reduce (MAX) of
// the residue sets per pass.  The transport is RCCL (`librccl.so`: ncclSend / ncclRecv grouped per neighbour, one
// ncclAllReduce) on the context's own stream, so kernels and transfers are ordered without host synchronisation.
// The library is loaded on first use (dlopen): a process that never shards does not need it.  How the 128-byte unique id
// reaches the other ranks (a file, MPI, a TCP store) is the caller's business: arp_comm_unique_id / arp_comm_init.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string>

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;       // optional: what the communicator itself reports (arp_comm_info)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    std::string error;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { error = std::string("librccl.so could not be loaded: ") + dlerror(); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) error = std::string("librccl.so lacks ") + n; return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        CommCount = (decltype(CommCount))dlsym(lib, "ncclCommCount");
        CommUserRank = (decltype(CommUserRank))dlsym(lib, "ncclCommUserRank");
        if (!(GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv && AllReduce && GetErrorString)) {
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        return true;
    }
};
inline RcclApi& rccl() {
    static RcclApi api;
    return api;
}

// selection_plus bits of the atoms a neighbour needs (gather) / of the halo atoms it sent (scatter)
__global__ __launch_bounds__(256) void k_gather_u8(int n, const int* __restrict__ idx, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}
__global__ __launch_bounds__(256) void k_scatter_u8(int n, const int* __restrict__ idx, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[idx[i]] = src[i];
}
