// Standalone reproducer (no torch, no RCCL) for what round 5 saw as an intermittent gradient overflow in data-parallel CRPS runs
// and round 6 narrowed down on the training loop itself (tools/overflow_campaign.sh, profiles/r06d_overflow_campaign.txt):
//   * clears done with hipMemsetAsync overflow the sums in 8 of 8 runs -- but ONLY when the launch sequences replay as HIP graphs
//     (0 of 4 with SWIFTK_TRAIN_GRAPHS=0), with or without a process group;
//   * the wrong values sit in every 4th column, or in columns 2,3 mod 4: a 16-BYTE PERIOD -- the period of the fill kernel's pattern.
// So the suspect is not ordering but the memset NODE of a captured graph: does a replayed hipMemsetAsync write its pattern
// (zeros) or something else, once other work has gone through the runtime between instantiation and replay?
//
// The graph: memset(ws, 0) -> check kernel that counts the non-zero dwords of ws and the largest |value| found (by dword index
// mod 4) -> poison kernel (ws := 1.0, so that a memset that does NOTHING is also caught).
// Between replays (the "churn", selectable): eager hipMemsetAsync of another buffer with the byte 0x7f, eager kernels with
// large by-value arguments, pageable host-to-device copies, other graphs with their own memset nodes.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/memset_graph_repro tools/memset_graph_repro.hip
// run:   /tmp/memset_graph_repro [replays]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

struct Big {
    unsigned long long v[440];  // 3,520 bytes of by-value kernel arguments: 32 such launches per replay wrap a MB-sized kernarg ring quickly
};

__global__ void check_kernel(const unsigned int* ws, int n, unsigned long long* bad, unsigned int* worst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ws[i] != 0u) {
        atomicAdd(bad + (i & 3), 1ull);
        atomicMax(worst + (i & 3), ws[i] & 0x7fffffffu);
    }
}
__global__ void poison_kernel(float* ws, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ws[i] = 1.0f;
}
__global__ void churn_kernel(float* p, Big b) {
    unsigned long long s = 0;
    for (int k = 0; k < 440; k += 55) s += b.v[k];
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = (float)(s & 0xff);
}

__global__ void check_copy_kernel(const unsigned int* ws, int n, unsigned long long* bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ws[i] != 0x3f800000u) atomicAdd(bad, 1ull);
}
__global__ void poison2_kernel(float* ws, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ws[i] = 2.0f;
}
__global__ void ptr_kernel(float* a, float* b, float* c, float* d, int k) {
    if (threadIdx.x == 0) a[0] = b[0] + c[0] + d[0] + (float)k;
}

static hipGraphExec_t capture(hipStream_t cap, float* ws, int n, unsigned long long* bad, unsigned int* worst) {
    hipGraph_t g;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(ws, 0, sizeof(float) * n, cap));
    hipLaunchKernelGGL(check_kernel, dim3((n + 255) / 256), dim3(256), 0, cap, reinterpret_cast<const unsigned int*>(ws), n, bad, worst);
    hipLaunchKernelGGL(poison_kernel, dim3((n + 255) / 256), dim3(256), 0, cap, ws, n);
    CK(hipStreamEndCapture(cap, &g));
    CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    return exec;
}

int main(int argc, char** argv) {
    const int replays = argc > 1 ? atoi(argv[1]) : 2000;
    const int n = 2 * 8 * 1056;  // swiftk_modnorm_bwd's column-sum workspace at local batch 8
    float *ws, *other, *scratch;
    unsigned long long* bad;
    unsigned int* worst;
    CK(hipMalloc(&ws, sizeof(float) * n));
    CK(hipMalloc(&other, 1 << 20));
    CK(hipMalloc(&scratch, 1 << 20));
    CK(hipMalloc(&bad, 32));
    CK(hipMalloc(&worst, 16));
    std::vector<char> pageable(1 << 16, 0x5a);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int rt = 0;
    CK(hipRuntimeGetVersion(&rt));
    printf("device %s, HIP runtime %d, %d replays per configuration, memset node of %d floats\n", prop.name, rt, replays, n);
    hipStream_t cap;
    CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    int failing = 0;
    // churn bits: 1 = eager hipMemsetAsync(other, 0x7f), 2 = eager kernels with 3.5 KB of by-value arguments (0x7f bytes), 4 = pageable H2D copies,
    // 8 = a second graph with its own (0x7f) memset node replayed in between
    for (int stream_kind = 0; stream_kind < 2; ++stream_kind)
        for (int churn = 0; churn < 16; ++churn) {
            hipStream_t st = nullptr;
            if (stream_kind == 1) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            CK(hipMemset(bad, 0, 32));
            CK(hipMemset(worst, 0, 16));
            CK(hipMemset(ws, 0, sizeof(float) * n));
            hipGraphExec_t exec = capture(cap, ws, n, bad, worst);
            hipGraphExec_t exec2 = nullptr;
            if (churn & 8) {
                hipGraph_t g;
                CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
                CK(hipMemsetAsync(other, 0x7f, 1 << 16, cap));
                CK(hipStreamEndCapture(cap, &g));
                CK(hipGraphInstantiate(&exec2, g, nullptr, nullptr, 0));
                CK(hipGraphDestroy(g));
            }
            Big b;
            for (int r = 0; r < replays; ++r) {
                CK(hipGraphLaunch(exec, st));
                for (int k = 0; k < 32; ++k) {
                    if (churn & 1) CK(hipMemsetAsync(other + 64 * k, 0x7f, 4096 + 16 * k, st));
                    if (churn & 2) {
                        for (int q = 0; q < 440; ++q) b.v[q] = 0x7f7f7f7f7f7f7f7full;
                        b.v[0] ^= (unsigned long long)(r * 32 + k);
                        hipLaunchKernelGGL(churn_kernel, dim3(1), dim3(64), 0, st, scratch, b);
                    }
                    if (churn & 4) CK(hipMemcpyAsync(scratch, pageable.data() + 64 * k, 4096, hipMemcpyHostToDevice, st));
                }
                if (churn & 8) CK(hipGraphLaunch(exec2, st));
                if ((r & 63) == 63) CK(hipStreamSynchronize(st));
            }
            CK(hipDeviceSynchronize());
            unsigned long long hb[4];
            unsigned int hw[4];
            CK(hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hw, worst, 16, hipMemcpyDeviceToHost));
            float fw[4];
            memcpy(fw, hw, 16);
            const unsigned long long tot = hb[0] + hb[1] + hb[2] + hb[3];
            printf("stream=%-12s churn=%2d [%s%s%s%s]: non-zero dwords after the replayed memset, by index mod 4: %llu %llu %llu %llu "
                   "(largest |value| %.3g %.3g %.3g %.3g)\n", stream_kind ? "non-blocking" : "null", churn, churn & 1 ? "memset7f " : "",
                   churn & 2 ? "bigargs " : "", churn & 4 ? "h2d " : "", churn & 8 ? "graph2 " : "", hb[0], hb[1], hb[2], hb[3], fw[0], fw[1],
                   fw[2], fw[3]);
            fflush(stdout);
            failing += tot != 0;
            CK(hipGraphExecDestroy(exec));
            if (exec2) CK(hipGraphExecDestroy(exec2));
            if (st) CK(hipStreamDestroy(st));
        }
    printf("%d of 32 configurations saw a replayed memset node leave non-zero data behind\n", failing);

    // ---- second experiment: the shape of the training loop's graphs.  Graph A = `before` kernel nodes with pointer arguments,
    // the memset node, the check kernel, the poison kernel, `before` more kernel nodes; then `others` further graphs of 300 kernel
    // nodes each are captured and instantiated AFTER A (what a training loop does: forward slots, backward passes, the tangent
    // pass), and all of them are replayed in turn.
    int failing2 = 0;
    for (int flags = 0; flags < 2; ++flags)
        for (int before = 0; before <= 300; before += 300)
            for (int others = 0; others <= 6; others += 6) {
                CK(hipMemset(bad, 0, 32));
                CK(hipMemset(worst, 0, 16));
                CK(hipMemset(ws, 0, sizeof(float) * n));
                auto many = [&](hipStream_t s_, int count, int salt) {
                    for (int k = 0; k < count; ++k)
                        hipLaunchKernelGGL(ptr_kernel, dim3(1), dim3(64), 0, s_, scratch + 16 * ((k + salt) & 1023), other + 16 * ((k + salt) & 1023),
                                           scratch + 8 * ((k + salt) & 1023), other + 8 * ((k + salt) & 1023), k);
                };
                auto inst = [&](hipGraph_t g) {
                    hipGraphExec_t e;
                    if (flags) CK(hipGraphInstantiateWithFlags(&e, g, hipGraphInstantiateFlagAutoFreeOnLaunch));
                    else CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
                    CK(hipGraphDestroy(g));
                    return e;
                };
                hipGraph_t g;
                CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
                many(cap, before, 0);
                CK(hipMemsetAsync(ws, 0, sizeof(float) * n, cap));
                hipLaunchKernelGGL(check_kernel, dim3((n + 255) / 256), dim3(256), 0, cap, reinterpret_cast<const unsigned int*>(ws), n, bad, worst);
                hipLaunchKernelGGL(poison_kernel, dim3((n + 255) / 256), dim3(256), 0, cap, ws, n);
                many(cap, before, 7);
                CK(hipStreamEndCapture(cap, &g));
                hipGraphExec_t A = inst(g);
                std::vector<hipGraphExec_t> Bs;
                for (int o = 0; o < others; ++o) {
                    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
                    many(cap, 300, 13 * o);
                    if (o & 1) CK(hipMemsetAsync(other, 0x7f, 4096, cap));
                    CK(hipStreamEndCapture(cap, &g));
                    Bs.push_back(inst(g));
                }
                for (int r = 0; r < replays / 4; ++r) {
                    CK(hipGraphLaunch(A, nullptr));
                    for (auto e : Bs) CK(hipGraphLaunch(e, nullptr));
                    many(nullptr, 16, r);
                    if ((r & 31) == 31) CK(hipStreamSynchronize(nullptr));
                }
                CK(hipDeviceSynchronize());
                unsigned long long hb[4];
                unsigned int hw[4];
                CK(hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hw, worst, 16, hipMemcpyDeviceToHost));
                printf("big graphs: instantiate flags=%s, %3d kernel nodes either side of the memset node, %d other graphs: non-zero dwords by index "
                       "mod 4: %llu %llu %llu %llu (largest bits %#x %#x %#x %#x)\n", flags ? "AutoFreeOnLaunch" : "default", before, others, hb[0],
                       hb[1], hb[2], hb[3], hw[0], hw[1], hw[2], hw[3]);
                fflush(stdout);
                failing2 += (hb[0] + hb[1] + hb[2] + hb[3]) != 0;
                CK(hipGraphExecDestroy(A));
                for (auto e : Bs) CK(hipGraphExecDestroy(e));
            }
    printf("%d of 8 big-graph configurations saw a replayed memset node leave non-zero data behind\n", failing2);

    // ---- third experiment: the OTHER runtime-owned node a captured launch sequence contains -- a device-to-device hipMemcpyAsync
    // (what torch's copy_ / clone() capture as).  Graph: memcpy(ws <- src, src all 1.0f) -> check (every dword must be 0x3f800000)
    // -> poison (ws := 2.0f), replayed on the null stream under the same eager churn.
    {
        float* src;
        CK(hipMalloc(&src, sizeof(float) * n));
        hipLaunchKernelGGL(poison_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, src, n);  // src := 1.0f
        CK(hipDeviceSynchronize());
        int failing3 = 0;
        for (int churn = 0; churn < 8; ++churn) {
            CK(hipMemset(bad, 0, 32));
            CK(hipMemset(worst, 0, 16));
            hipGraph_t g;
            hipGraphExec_t exec;
            CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
            CK(hipMemcpyAsync(ws, src, sizeof(float) * n, hipMemcpyDeviceToDevice, cap));
            hipLaunchKernelGGL(check_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, cap, reinterpret_cast<const unsigned int*>(ws), n, bad);
            hipLaunchKernelGGL(poison2_kernel, dim3((n + 255) / 256), dim3(256), 0, cap, ws, n);
            CK(hipStreamEndCapture(cap, &g));
            CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
            Big b;
            for (int r = 0; r < replays; ++r) {
                CK(hipGraphLaunch(exec, nullptr));
                for (int k = 0; k < 32; ++k) {
                    if (churn & 1) CK(hipMemsetAsync(other + 64 * k, 0x7f, 4096 + 16 * k, nullptr));
                    if (churn & 2) {
                        for (int q = 0; q < 440; ++q) b.v[q] = 0x7f7f7f7f7f7f7f7full;
                        hipLaunchKernelGGL(churn_kernel, dim3(1), dim3(64), 0, nullptr, scratch, b);
                    }
                    if (churn & 4) CK(hipMemcpyAsync(scratch, pageable.data() + 64 * k, 4096, hipMemcpyHostToDevice, nullptr));
                }
                if ((r & 63) == 63) CK(hipStreamSynchronize(nullptr));
            }
            CK(hipDeviceSynchronize());
            unsigned long long hb[4];
            CK(hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost));
            printf("memcpy node, null stream, churn=%d: dwords that are not the source's 1.0f after the replayed copy: %llu\n", churn, hb[0]);
            failing3 += hb[0] != 0;
            CK(hipGraphExecDestroy(exec));
        }
        printf("%d of 8 configurations saw a replayed device-to-device memcpy node deliver wrong data\n", failing3);
    }
    return 0;
}
