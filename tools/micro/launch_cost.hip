// Host cost of one kernel launch by the three ways the runtime offers: the <<<>>> form
// (hipLaunchKernelGGL), hipLaunchKernel with an argument array, and hipModuleLaunchKernel on a function
// handle resolved once (hipGetFuncBySymbol).  A small kernel with a handful of arguments, launches
// enqueued back to back on one stream, host time per launch; then the same with an event record and a
// stream wait between the launches (what a frame's enqueue is made of).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void tiny_kernel(const double *a, double *b, int n, double s, const int *c, int *d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && c) b[i] = a[i] * s + (d ? d[0] : 0);
}

static double us_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

int main()
{
    double *a, *b;
    int *c;
    (void)hipMalloc(&a, 8 * 256);
    (void)hipMalloc(&b, 8 * 256);
    (void)hipMalloc(&c, 4 * 16);
    hipStream_t s, s2;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t ev;
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    hipFunction_t fn = nullptr;
    const hipError_t e = hipGetFuncBySymbol(&fn, (const void *)tiny_kernel);
    printf("hipGetFuncBySymbol: %s\n", hipGetErrorString(e));
    int n = 256, *d = nullptr;
    double sc = 2.0;
    const int N = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(256), 0, s, a, b, n, sc, c, d);
        const double t_ggl = us_since(t0) / N;
        (void)hipStreamSynchronize(s);
        void *args[] = {&a, &b, &n, &sc, &c, &d};
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) (void)hipLaunchKernel((const void *)tiny_kernel, dim3(1), dim3(256), args, 0, s);
        const double t_lk = us_since(t0) / N;
        (void)hipStreamSynchronize(s);
        double t_mod = -1.0;
        if (fn) {
            t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) (void)hipModuleLaunchKernel(fn, 1, 1, 1, 256, 1, 1, 0, s, args, nullptr);
            t_mod = us_since(t0) / N;
            (void)hipStreamSynchronize(s);
        }
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(256), 0, s, a, b, n, sc, c, d);
            (void)hipEventRecord(ev, s);
            (void)hipStreamWaitEvent(s2, ev, 0);
            hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(256), 0, s2, a, b, n, sc, c, d);
        }
        const double t_mix = us_since(t0) / N;
        (void)hipStreamSynchronize(s);
        (void)hipStreamSynchronize(s2);
        printf("per launch: <<<>>> %.2f us, hipLaunchKernel %.2f us, hipModuleLaunchKernel %.2f us; launch + record + wait + launch %.2f us\n",
               t_ggl, t_lk, t_mod, t_mix);
    }
    return 0;
}
