// Microbenchmark (not part of the product): what one "v_cmp_f64 -> s_and_saveexec -> s_cbranch_execz -> ... -> s_or exec"
// sequence costs a wave that has its SIMD to itself, against the same fp64 work without it.
// build: hipcc --offload-arch=gfx950 -O3 -o branch_cost tools/micro/branch_cost.hip ; run: ./branch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

// K fp64 FMAs per iteration in NCH independent chains; every iteration optionally tests a value no lane ever satisfies
template <int K, int NCH, int NBR>
__global__ __launch_bounds__(64) void k(double *out, const double *in, int iters, long long *clk) {
    double a = in[0], b = in[1], thr = in[2];
    double x[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) x[c] = in[3 + c] + threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < K; j++) {
            x[j % NCH] = __builtin_fma(x[j % NCH], a, b);
            if (NBR > 0 && (j % (K / (NBR > 0 ? NBR : 1))) == 0) {
                if (__builtin_expect(x[j % NCH] > thr, 0)) {        // never true: thr is huge
                    out[threadIdx.x + 64 * (j + 1)] = x[j % NCH];
                    x[j % NCH] = 0.0;
                }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) s += x[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int K, int NCH, int NBR>
static void run(const char *name, double *out, double *in, long long *clk) {
    const int iters = 20000;
    hipLaunchKernelGGL((k<K, NCH, NBR>), dim3(1), dim3(64), 0, 0, out, in, iters, clk);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<K, NCH, NBR>), dim3(1), dim3(64), 0, 0, out, in, iters, clk);
    CHECK(hipDeviceSynchronize());
    long long c;
    CHECK(hipMemcpy(&c, clk, sizeof c, hipMemcpyDeviceToHost));
    // s_memtime ticks at 100 MHz: convert with the measured ratio of the plain FMA loop instead of trusting a clock rate
    printf("%-34s %8.2f ticks/iter  (%d fma, %d chains, %d tests per iter)\n", name, double(c) / iters, K, NCH, NBR);
}

int main() {
    double *out, *in;
    long long *clk;
    CHECK(hipMalloc(&out, 64 * 64 * sizeof(double)));
    CHECK(hipMalloc(&in, 64 * sizeof(double)));
    CHECK(hipMalloc(&clk, 64 * sizeof(long long)));
    double h[64];
    for (int i = 0; i < 64; i++) h[i] = 0.5 + 0.001 * i;
    h[0] = 0.999; h[1] = 1e-3; h[2] = 1e300;
    CHECK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    run<32, 8, 0>("32 fma, 8 chains, no test", out, in, clk);
    run<32, 8, 4>("32 fma, 8 chains, 4 tests", out, in, clk);
    run<32, 8, 8>("32 fma, 8 chains, 8 tests", out, in, clk);
    run<32, 8, 16>("32 fma, 8 chains, 16 tests", out, in, clk);
    run<32, 1, 0>("32 fma, 1 chain, no test", out, in, clk);
    run<32, 1, 8>("32 fma, 1 chain, 8 tests", out, in, clk);
    run<32, 2, 0>("32 fma, 2 chains, no test", out, in, clk);
    run<32, 4, 0>("32 fma, 4 chains, no test", out, in, clk);
    return 0;
}
