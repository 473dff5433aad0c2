// How long the GPU idles between two dependent kernels of one stream, launched one by one or replayed as a HIP graph:
// a chain of N kernels that each keep every CU busy for about `us` microseconds, timed as a whole.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gap_probe.hip -o tools/probes/gap_probe && tools/probes/gap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_busy(float *out, int iters)
{
    float x = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.678f) out[blockIdx.x] = x;
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main()
{
    float *out = nullptr;
    CK(hipMalloc(&out, 1 << 20));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 26, reps = 20;
    for (int iters : {1000, 8000}) {
        for (int blocks : {2048, 16384}) {
            auto chain = [&]() { for (int k = 0; k < N; ++k) k_busy<<<blocks, 256, 0, s>>>(out, iters); };
            // one kernel alone
            chain(); CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) k_busy<<<blocks, 256, 0, s>>>(out, iters * N);
            CK(hipStreamSynchronize(s));
            double one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) chain();
            CK(hipStreamSynchronize(s));
            double stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            chain();
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            printf("iters %6d blocks %5d: one kernel of the same work %8.1f us, %d launches %8.1f us (+%.1f per launch), graph %8.1f us (+%.1f per launch)\n",
                   iters, blocks, one, N, stream, (stream - one) / N, graph, (graph - one) / N);
            fflush(stdout);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
