// HipGlue.h — the only place the host facade touches the HIP runtime (kept in its own translation unit so
// that the facade's float3/uchar3 never meet HIP's vector types).
#pragma once
#include <cstddef>
namespace hipglue {
void* streamCreate();
void  streamDestroy(void* stream);
void  streamSync(void* stream);
void* deviceAlloc(size_t bytes);
void  deviceFree(void* p);
void  copyToHost(void* dst, const void* src, size_t bytes);
void* hostAllocPinned(size_t bytes);
void  hostFreePinned(void* p);
void  copyToHostAsync(void* dst, const void* src, size_t bytes, void* stream);
// several GPUs in one process (the CLI's --gpus N: one GaussianTracer per device, each driven by its own host thread)
int   deviceCount();
void  setDevice(int device);
void  copyPeerAsync(void* dst, int dst_device, const void* src, int src_device, size_t bytes, void* stream);
// RCCL over xGMI ("one RCCL gather to rank 0 per frame — ncclGather, rccl.h:745, or grouped ncclSend /
// ncclRecv"): one communicator per rank of this process (ncclCommInitAll); every rank's host thread calls gatherToRoot
// for its own rank, rank 0's call also posts the receives into the slices of its gather buffer.  Distinct devices only.
struct Rccl;
Rccl* rcclInitAll(const int* devices, int n);
void  rcclGatherToRoot(Rccl* r, int rank, const void* send, size_t send_bytes, void* recv_base, const size_t* bytes_of_rank,
                       size_t slice_bytes, void* stream);
void  rcclDestroy(Rccl* r);
}
