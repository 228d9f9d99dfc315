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
}
