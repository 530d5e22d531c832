// Test infrastructure: the HIP runtime calls of rpt_amd/csrc/rpt_capi.cpp backed by malloc / memcpy, so that the
// host half of the library (scene store, flattening, tree builds, box shell, instancing, error paths) runs on a CPU,
// under sanitizers and at different optimisation levels.  Kernel launches are no-ops.  Never part of the product.
#pragma once
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime_api.h>
extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
    std::memset(p, 0, sizeof(*p));
    std::strcpy(p->gcnArchName, "gfx950");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(n ? n : 1, 1); return hipSuccess; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t n, hipStream_t) {
    for (size_t i = 0; i < n; i++) std::memcpy(static_cast<char*>(d) + 4 * i, &v, 4);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* f, hipEvent_t, hipEvent_t) { *f = 0; return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
}
