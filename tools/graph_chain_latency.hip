// How long does a dependent kernel in a hipGraph take as a function of the straight-line code
// it executes?  One wave per kernel, N unrolled dependent FMAs (8 bytes of code each, ~8 cycles
// of execution each), 256 kernels per graph.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/graph_chain_latency tools/graph_chain_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

template <int N, bool LOOP>
__global__ void chain_kernel(float *x, float a, float b)
{
    float v = x[threadIdx.x];
    if (LOOP) {
#pragma nounroll
        for (int i = 0; i < N; i++)
            v = __builtin_fmaf(v, a, b);
    } else {
#pragma unroll
        for (int i = 0; i < N; i++)
            v = __builtin_fmaf(v, a, b + (float) i);      // distinct literal: no loop re-rolling
    }
    x[threadIdx.x] = v;
}

template <int N, bool LOOP>
void run(float *x, hipStream_t s)
{
    hipGraph_t graph;
    hipGraphExec_t exec;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 256; i++)
        chain_kernel<N, LOOP><<<1, 64, 0, s>>>(x, 0.999f, 0.001f);
    hipStreamEndCapture(s, &graph);
    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphLaunch(exec, s);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 8; r++)
        hipGraphLaunch(exec, s);
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("N=%5d %s: %.2f us per kernel\n", N, LOOP ? "rolled  " : "unrolled", us / (8 * 256));
    hipGraphExecDestroy(exec);
    hipGraphDestroy(graph);
}

int main()
{
    float *x;
    hipMalloc(&x, 64 * sizeof(float));
    hipMemset(x, 0, 64 * sizeof(float));
    hipStream_t s;
    hipStreamCreate(&s);
    run<16, false>(x, s);
    run<128, false>(x, s);
    run<512, false>(x, s);
    run<1024, false>(x, s);
    run<2048, false>(x, s);
    run<16, true>(x, s);
    run<512, true>(x, s);
    run<2048, true>(x, s);
    return 0;
}
