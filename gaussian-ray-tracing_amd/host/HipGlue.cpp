#include "HipGlue.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <vector>

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

struct Rccl { std::vector<ncclComm_t> comms; };
static void nchk(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}
Rccl* rcclInitAll(const int* devices, int n)
{
    Rccl* r = new Rccl();
    r->comms.resize((size_t)n);
    const ncclResult_t e = ncclCommInitAll(r->comms.data(), n, devices);
    if (e != ncclSuccess) { delete r; nchk(e, "ncclCommInitAll"); }
    return r;
}
void rcclGatherToRoot(Rccl* r, int rank, const void* send, size_t send_bytes, void* recv_base, const size_t* bytes_of_rank, size_t slice_bytes,
                      void* stream)
{
    // grouped point-to-point: 7 peers -> 7 distinct xGMI links into rank 0 (no ring; a rank with nothing to send still takes
    // part with a zero-byte message so that the group matches on every rank)
    const int n = (int)r->comms.size();
    nchk(ncclGroupStart(), "ncclGroupStart");
    nchk(ncclSend(send, send_bytes, ncclUint8, 0, r->comms[(size_t)rank], (hipStream_t)stream), "ncclSend");
    if (rank == 0)
        for (int k = 0; k < n; k++)
            nchk(ncclRecv((char*)recv_base + (size_t)k * slice_bytes, bytes_of_rank[k], ncclUint8, k, r->comms[0], (hipStream_t)stream), "ncclRecv");
    nchk(ncclGroupEnd(), "ncclGroupEnd");
}
void rcclDestroy(Rccl* r)
{
    if (!r) return;
    for (ncclComm_t c : r->comms) (void)ncclCommDestroy(c);
    delete r;
}
}
