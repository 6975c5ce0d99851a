// Probe: GPU-side time per dependent tiny kernel, enqueued on a stream vs launched as a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(double* p, int k) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += (double)k; }
__global__ void mid(double* p, int k) { p[blockIdx.x * 256 + threadIdx.x] += (double)k; }
int main()
{
    double* d; CK(hipMalloc(&d, 1024 * 256 * sizeof(double))); CK(hipMemset(d, 0, 1024 * 256 * sizeof(double)));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 200;
    for (int variant = 0; variant < 2; ++variant)
    {
        auto enqueue = [&]() {
            for (int k = 0; k < N; ++k)
            {
                if (variant == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, s, d, k);
                else hipLaunchKernelGGL(mid, dim3(1024), dim3(256), 0, s, d, k);
            }
        };
        for (int rep = 0; rep < 3; ++rep)
        {
            CK(hipEventRecord(e0, s)); enqueue(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("variant %d stream: %.2f us per kernel\n", variant, 1e3 * ms / N);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); enqueue(); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep)
        {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("variant %d graph:  %.2f us per kernel\n", variant, 1e3 * ms / N);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
