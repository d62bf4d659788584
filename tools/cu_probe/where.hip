// Probe kernel for tools/cu_partition_probe.py: every workgroup reports the XCC / SE / SH / CU it runs on.
// Not part of libswiftk; built by the tool into tools/cu_probe/libwhere.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void where_kernel(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // hold the CU for a while so that the workgroups of one launch spread over every CU the queue may use
    const uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (uint64_t)spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}

extern "C" int where_launch(uint32_t* out, int nblocks, int threads, int lds_bytes, int spin, void* stream) {
    hipLaunchKernelGGL(where_kernel, dim3(nblocks), dim3(threads), lds_bytes, static_cast<hipStream_t>(stream), out, spin);
    return (int)hipGetLastError();
}
