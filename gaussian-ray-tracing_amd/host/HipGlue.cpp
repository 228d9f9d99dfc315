#include "HipGlue.h"

#include <hip/hip_runtime_api.h>

#include <stdexcept>
#include <string>

namespace hipglue {
static void chk(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void* streamCreate() { hipStream_t s; chk(hipStreamCreate(&s), "hipStreamCreate"); return s; }
void streamDestroy(void* s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
void streamSync(void* s) { chk(hipStreamSynchronize((hipStream_t)s), "hipStreamSynchronize"); }
void* deviceAlloc(size_t bytes) { void* p = nullptr; chk(hipMalloc(&p, bytes), "hipMalloc"); return p; }
void deviceFree(void* p) { if (p) (void)hipFree(p); }
void copyToHost(void* dst, const void* src, size_t bytes) { chk(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), "hipMemcpy"); }
void* hostAllocPinned(size_t bytes) { void* p = nullptr; chk(hipHostMalloc(&p, bytes, hipHostMallocDefault), "hipHostMalloc"); return p; }
void hostFreePinned(void* p) { if (p) (void)hipHostFree(p); }
int deviceCount() { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
void setDevice(int device) { chk(hipSetDevice(device), "hipSetDevice"); }
void copyPeerAsync(void* dst, int dst_device, const void* src, int src_device, size_t bytes, void* stream)
{
    // over xGMI between two devices of one node (a plain copy when both are the same device)
    chk(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, (hipStream_t)stream), "hipMemcpyPeerAsync");
}
void copyToHostAsync(void* dst, const void* src, size_t bytes, void* stream) { chk(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), "hipMemcpyAsync"); }
}
