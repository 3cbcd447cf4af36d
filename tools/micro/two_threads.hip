// Host cost of enqueuing small kernels on four streams from one thread against two threads (two streams each):
// does the runtime let two threads enqueue at once?
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
__global__ void work(float *p, int n, int spin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? p[i] : 0.f;
    for (int k = 0; k < spin; ++k) v = v * 1.0001f + 0.5f;
    if (i < n) p[i] = v;
}
static hipStream_t s[4];
static hipEvent_t ev[8];
static float *buf[4];
// 10 launches, one record and one wait per stream
static int half(int a, int b)
{
    for (int k = 0; k < 10; ++k) {
        hipLaunchKernelGGL(work, dim3(8), dim3(256), 0, s[a], buf[a], 2048, 50);
        hipLaunchKernelGGL(work, dim3(8), dim3(256), 0, s[b], buf[b], 2048, 50);
    }
    CK(hipEventRecord(ev[a], s[a]));
    CK(hipStreamWaitEvent(s[b], ev[a], 0));
    CK(hipEventRecord(ev[b], s[b]));
    CK(hipStreamWaitEvent(s[a], ev[b], 0));
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
    const int reps = 300;
    for (int w = 0; w < 20; ++w) { if (half(0, 1)) return 1; if (half(2, 3)) return 1; }
    CK(hipDeviceSynchronize());
    auto t0 = clk::now();
    for (int r = 0; r < reps; ++r) { if (half(0, 1)) return 1; if (half(2, 3)) return 1; }
    auto t1 = clk::now();
    CK(hipDeviceSynchronize());
    std::printf("one thread: %.1f us per pattern (40 launches, 4 records, 4 waits)\n",
                std::chrono::duration<double, std::micro>(t1 - t0).count() / reps);
    // two threads, lock step per pattern (a spin barrier), as a frame loop would use a helper
    std::atomic<int> go{0}, done{0};
    std::atomic<bool> quit{false};
    std::thread helper([&] {
        (void)hipSetDevice(0);
        int seen = 0;
        while (true) {
            while (go.load(std::memory_order_acquire) == seen) {
                if (quit.load()) return;
            }
            ++seen;
            half(2, 3);
            done.store(seen, std::memory_order_release);
        }
    });
    t0 = clk::now();
    for (int r = 0; r < reps; ++r) {
        go.store(r + 1, std::memory_order_release);
        if (half(0, 1)) return 1;
        while (done.load(std::memory_order_acquire) != r + 1) {}
    }
    t1 = clk::now();
    quit.store(true);
    helper.join();
    CK(hipDeviceSynchronize());
    std::printf("two threads: %.1f us per pattern\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps);
    return 0;
}
