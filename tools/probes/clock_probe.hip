// Does a wave slow down when the other SIMDs of its CU are busy, and is it the clock or the issue?
// G one-wave workgroups (39 KB of LDS each: at most 4 per CU, one per SIMD) run a chain of fp64 FMAs;
// reports wall time per FMA instruction and shader-clock ticks per wall microsecond.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/clock_probe.hip -o tests/_build/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ __launch_bounds__(64) void chain(int iters, int lanes, double *sink, unsigned long long *stamps) {
    __shared__ double pad[39 * 128];
    pad[threadIdx.x] = 1.0;
    if ((int)threadIdx.x >= lanes) return;
    double a[ILP];
    for (int k = 0; k < ILP; k++) a[k] = 1.0 + 1e-9 * (threadIdx.x + k);
    const double m = 1.0000001, c = pad[threadIdx.x] * 1e-12;
    const unsigned long long w0 = wall_clock64(), s0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int k = 0; k < ILP; k++) a[k] = __builtin_fma(a[k], m, c);
    }
    const unsigned long long s1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int k = 0; k < ILP; k++) s += a[k];
    if (s == 12345.678) sink[0] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = s1 - s0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int ILP>
void run(int G, int lanes) {
    const int iters = 20000;
    double *sink; unsigned long long *st;
    hipMalloc(&sink, 8); hipMalloc(&st, 16 * G);
    hipLaunchKernelGGL(chain<ILP>, dim3(G), dim3(64), 0, 0, 100, lanes, sink, st);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<ILP>, dim3(G), dim3(64), 0, 0, iters, lanes, sink, st);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * G);
    hipMemcpy(h.data(), st, 16 * G, hipMemcpyDeviceToHost);
    double sc = 0, wc = 0;
    for (int i = 0; i < G; i++) { sc += h[2 * i]; wc += h[2 * i + 1]; }
    sc /= G; wc /= G;
    const double n = double(iters) * 16 * ILP;
    printf("ILP %d  waves %5d  lanes %2d : kernel %.3f ms, %.3f ns per FMA instr per wave, shader ticks/FMA %.2f, "
           "shader ticks per wall tick %.2f\n", ILP, G, lanes, ms, ms * 1e6 / n, sc / n, sc / wc);
    hipFree(sink); hipFree(st);
}

int main() {
    for (int lanes : {1, 64})
        for (int G : {64, 256, 512, 1024, 2048}) run<1>(G, lanes);
    for (int lanes : {1, 64})
        for (int G : {256, 1024}) run<4>(G, lanes);
    return 0;
}
