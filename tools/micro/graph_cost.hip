// Host cost of enqueuing a frame-like pattern (40 small kernels over 4 streams with event edges):
// direct launches against one hipGraphLaunch of the captured pattern.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
__global__ void work(float *p, int n, int spin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? p[i] : 0.f;
    for (int k = 0; k < spin; ++k) v = v * 1.0001f + 0.5f;
    if (i < n) p[i] = v;
}
static hipStream_t s[4];
static hipEvent_t ev[16];
static float *buf[4];
static int pattern()
{
    // origin s[0]; forks to s[1..3], joins back
    CK(hipEventRecord(ev[0], s[0]));
    for (int q = 1; q < 4; ++q) CK(hipStreamWaitEvent(s[q], ev[0], 0));
    for (int k = 0; k < 10; ++k) {
        for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(work, dim3(8), dim3(256), 0, s[q], buf[q], 2048, 200);
        if (k == 4) {   // a cross edge in the middle
            CK(hipEventRecord(ev[1], s[1]));
            CK(hipStreamWaitEvent(s[0], ev[1], 0));
            CK(hipEventRecord(ev[2], s[0]));
            CK(hipStreamWaitEvent(s[2], ev[2], 0));
        }
    }
    for (int q = 1; q < 4; ++q) {
        CK(hipEventRecord(ev[3 + q], s[q]));
        CK(hipStreamWaitEvent(s[0], ev[3 + q], 0));
    }
    return 0;
}
int main()
{
    for (int q = 0; q < 4; ++q) {
        CK(hipStreamCreateWithFlags(&s[q], hipStreamNonBlocking));
        CK(hipMalloc(&buf[q], 2048 * sizeof(float)));
        CK(hipMemset(buf[q], 0, 2048 * sizeof(float)));
    }
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    using clk = std::chrono::steady_clock;
    const int reps = 200;
    for (int w = 0; w < 20; ++w) if (pattern()) return 1;
    CK(hipDeviceSynchronize());
    auto t0 = clk::now();
    for (int r = 0; r < reps; ++r) if (pattern()) return 1;
    auto t1 = clk::now();
    CK(hipDeviceSynchronize());
    auto t2 = clk::now();
    std::printf("direct: host enqueue %.1f us per pattern, end to end %.1f us\n",
                std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
                std::chrono::duration<double, std::micro>(t2 - t0).count() / reps);
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
    if (pattern()) return 1;
    CK(hipStreamEndCapture(s[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nn = 0;
    CK(hipGraphGetNodes(g, nullptr, &nn));
    for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, s[0]));
    CK(hipDeviceSynchronize());
    t0 = clk::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s[0]));
    t1 = clk::now();
    CK(hipDeviceSynchronize());
    t2 = clk::now();
    std::printf("graph (%zu nodes): host launch %.1f us per pattern, end to end %.1f us\n", nn,
                std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
                std::chrono::duration<double, std::micro>(t2 - t0).count() / reps);
    // one kernel node's parameters changed before every launch
    hipGraphNode_t nodes[64];
    nn = 64;
    CK(hipGraphGetNodes(g, nodes, &nn));
    hipGraphNode_t kn = nullptr;
    for (size_t q = 0; q < nn; ++q) {
        hipGraphNodeType ty;
        CK(hipGraphNodeGetType(nodes[q], &ty));
        if (ty == hipGraphNodeTypeKernel) { kn = nodes[q]; break; }
    }
    hipKernelNodeParams kp;
    CK(hipGraphKernelNodeGetParams(kn, &kp));
    t0 = clk::now();
    for (int r = 0; r < reps; ++r) {
        kp.gridDim = dim3(8 + (r & 1));
        CK(hipGraphExecKernelNodeSetParams(ge, kn, &kp));
        CK(hipGraphLaunch(ge, s[0]));
    }
    t1 = clk::now();
    CK(hipDeviceSynchronize());
    t2 = clk::now();
    std::printf("graph + 1 node update: host %.1f us per pattern, end to end %.1f us\n",
                std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
                std::chrono::duration<double, std::micro>(t2 - t0).count() / reps);
    return 0;
}
