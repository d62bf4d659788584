// Standalone reproducer attempt (no torch, no RCCL) for the round-5 data-parallel gradient overflow (DESIGN, "open questions"):
// is a hipMemsetAsync reliably ordered in front of an atomically accumulating kernel on the same stream when cross-stream
// event waits sit on that stream?  VERDICT r5 item 3a.
//
// One round, on the MAIN stream:
//   busy kernel (keeps the queue deep)  ->  [event ping-pong with a SIDE stream, as a collective's stream would do]
//   -> poison the workspace (so a skipped / late clear is visible, whatever the old contents)
//   -> clear (mode 0: hipMemsetAsync, mode 1: an ordinary kernel)
//   -> accumulate: `blocks` workgroups each atomicAdd 1.0f into every one of n floats
//   -> check: a kernel compares every float with `blocks` and counts mismatches into a device counter.
// Configurations: main stream = created / null / non-blocking; cross-stream waits on / off; a captured-graph replay of the same
// sequence interleaved with the eager rounds on / off (the training loop mixes both on one workspace).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/memset_order_repro tools/memset_order_repro.hip     run: /tmp/memset_order_repro [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

__global__ void busy_kernel(float* p, int n, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int k = 0; k < iters; ++k) v = v * 1.0000001f + 1e-7f;
    p[i] = v;
}
__global__ void poison_kernel(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 3e19f;
}
__global__ void zero_kernel(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}
__global__ void accumulate_kernel(float* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(p + i, 1.0f);
}
__global__ void check_kernel(const float* p, int n, float expect, unsigned long long* bad, float* worst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i] != expect) {
        atomicAdd(bad, 1ull);
        atomicMax(reinterpret_cast<unsigned int*>(worst), __float_as_uint(fabsf(p[i] - expect)));
    }
}
__global__ void side_kernel(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int k = 0; k < iters; ++k) v = v * 1.0000001f + 1e-7f;
    p[threadIdx.x] = v;
}

struct Cfg {
    int mode;       // 0 hipMemsetAsync, 1 zero kernel
    int stream;     // 0 created (blocking), 1 null stream, 2 created non-blocking
    int cross;      // event ping-pong with the side stream
    int graph;      // interleave a captured-graph replay of the same sequence
};

static void sequence(hipStream_t st, const Cfg& c, float* ws, int n, int blocks, float* scratch, int ns, unsigned long long* bad,
                     float* worst, hipStream_t side, hipEvent_t e0, hipEvent_t e1, float* sidebuf, bool allow_cross) {
    hipLaunchKernelGGL(busy_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, scratch, ns, 200);
    if (c.cross && allow_cross) {
        CK(hipEventRecord(e0, st));
        CK(hipStreamWaitEvent(side, e0, 0));
        hipLaunchKernelGGL(side_kernel, dim3(1), dim3(64), 0, side, sidebuf, 2000);
        CK(hipEventRecord(e1, side));
        CK(hipStreamWaitEvent(st, e1, 0));
    }
    hipLaunchKernelGGL(poison_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws, n);
    if (c.mode == 0) CK(hipMemsetAsync(ws, 0, sizeof(float) * n, st));
    else hipLaunchKernelGGL(zero_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws, n);
    hipLaunchKernelGGL(accumulate_kernel, dim3(blocks), dim3(256), 0, st, ws, n);
    hipLaunchKernelGGL(check_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws, n, (float)blocks, bad, worst);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 1000;
    const int n = 2 * 8 * 1056;   // the psum workspace of swiftk_modnorm_bwd at local batch 8: [2][8][1056] floats
    const int blocks = 256;       // one accumulating workgroup per CU
    const int ns = 1 << 22;
    float *ws, *scratch, *sidebuf, *worst;
    unsigned long long* bad;
    CK(hipMalloc(&ws, sizeof(float) * n));
    CK(hipMalloc(&scratch, sizeof(float) * ns));
    CK(hipMalloc(&sidebuf, 256));
    CK(hipMalloc(&bad, 8));
    CK(hipMalloc(&worst, 4));
    CK(hipMemset(scratch, 0, sizeof(float) * ns));
    CK(hipMemset(sidebuf, 0, 256));
    hipStream_t side;
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int rt = 0;
    CK(hipRuntimeGetVersion(&rt));
    printf("device %s, HIP runtime %d, %d rounds per configuration, workspace %d floats, %d accumulating workgroups\n", prop.name, rt,
           rounds, n, blocks);
    int total_bad_cfgs = 0;
    for (int mode = 0; mode < 2; ++mode)
        for (int stream = 0; stream < 3; ++stream)
            for (int cross = 0; cross < 2; ++cross)
                for (int graph = 0; graph < 2; ++graph) {
                    Cfg c{mode, stream, cross, graph};
                    hipStream_t st = nullptr;
                    if (stream == 0) CK(hipStreamCreate(&st));
                    if (stream == 2) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                    CK(hipMemset(bad, 0, 8));
                    CK(hipMemset(worst, 0, 4));
                    hipGraphExec_t exec = nullptr;
                    hipStream_t cap = nullptr;
                    if (graph) {  // (capture needs a created stream; the replay is launched on the main stream)
                        CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
                        hipGraph_t g;
                        CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
                        sequence(cap, c, ws, n, blocks, scratch, ns, bad, worst, side, e0, e1, sidebuf, false);
                        CK(hipStreamEndCapture(cap, &g));
                        CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
                        CK(hipGraphDestroy(g));
                    }
                    for (int r = 0; r < rounds; ++r) {
                        if (graph && (r & 1)) CK(hipGraphLaunch(exec, st));
                        else sequence(st, c, ws, n, blocks, scratch, ns, bad, worst, side, e0, e1, sidebuf, true);
                        if ((r & 63) == 63) CK(hipStreamSynchronize(st));  // (the host stays at most 64 rounds ahead)
                    }
                    CK(hipDeviceSynchronize());
                    unsigned long long hb = 0;
                    float hw = 0;
                    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(&hw, worst, 4, hipMemcpyDeviceToHost));
                    printf("clear=%-14s stream=%-12s cross_waits=%d graph_mix=%d : %llu wrong sums of %lld (worst |error| %.3g)\n",
                           mode ? "kernel" : "hipMemsetAsync", stream == 0 ? "created" : stream == 1 ? "null" : "non-blocking", cross,
                           graph, hb, (long long)rounds * n, hw);
                    fflush(stdout);
                    total_bad_cfgs += hb != 0;
                    if (exec) CK(hipGraphExecDestroy(exec));
                    if (cap) CK(hipStreamDestroy(cap));
                    if (st) CK(hipStreamDestroy(st));
                }
    printf("%d of 24 configurations saw a wrong sum\n", total_bad_cfgs);
    return 0;
}
