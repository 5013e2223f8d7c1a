// Microbenchmark 2 (not part of the product): which part of the "v_cmp -> s_and_saveexec -> s_cbranch -> s_or exec"
// sequence costs a lone wave its ~40 clocks.  Hand-written blocks, 8 per loop iteration, never-taken branches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

#define FMA4 "v_fma_f64 %0, %4, %0, %5\n v_fma_f64 %1, %4, %1, %5\n v_fma_f64 %2, %4, %2, %5\n v_fma_f64 %3, %4, %3, %5\n"
#define OPS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "v"(thr) : "vcc", "s6", "s7", "s8", "s9", "scc"

template <int V>
__global__ __launch_bounds__(64) void k(double *out, const double *in, int iters, long long *clk) {
    double a = in[0], b = in[1], thr = in[2];
    double x0 = in[3] + threadIdx.x, x1 = in[4] + threadIdx.x, x2 = in[5] + threadIdx.x, x3 = in[6] + threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (V == 0) asm volatile(FMA4 OPS);
            if (V == 1) asm volatile(FMA4 "v_cmp_lt_f64 vcc, %6, %3\n s_and_saveexec_b64 s[6:7], vcc\n s_cbranch_execnz 1f\n 1: s_or_b64 exec, exec, s[6:7]\n" OPS);
            if (V == 2) asm volatile(FMA4 "v_cmp_lt_f64 vcc, %6, %3\n s_cbranch_vccnz 1f\n 1:\n" OPS);
            if (V == 3) asm volatile("v_cmp_lt_f64 vcc, %6, %3\n" FMA4 "s_and_saveexec_b64 s[6:7], vcc\n s_cbranch_execnz 1f\n 1: s_or_b64 exec, exec, s[6:7]\n" OPS);
            if (V == 4) asm volatile("v_cmp_lt_f64 vcc, %6, %3\n" FMA4 "s_cbranch_vccnz 1f\n 1:\n" OPS);
            if (V == 5) asm volatile(FMA4 "v_cmp_lt_f64 vcc, %6, %3\n" OPS);                                  // compare only
            if (V == 6) asm volatile(FMA4 "v_cmp_lt_f64 vcc, %6, %3\n s_and_saveexec_b64 s[6:7], vcc\n s_or_b64 exec, exec, s[6:7]\n" OPS);   // no branch
            if (V == 7) asm volatile(FMA4 "v_cmp_lt_f64 vcc, %6, %3\n s_and_b64 s[8:9], vcc, exec\n s_cbranch_scc1 1f\n 1:\n" OPS);   // scalar test, exec untouched
            if (V == 8) asm volatile("v_cmp_lt_f64 s[8:9], %6, %3\n" FMA4 "s_and_saveexec_b64 s[6:7], s[8:9]\n s_cbranch_execnz 1f\n 1: s_or_b64 exec, exec, s[6:7]\n" OPS);
            if (V == 9) asm volatile(FMA4 "s_cbranch_execz 1f\n 1:\n" OPS);                                   // a branch on a flag nothing just wrote
            if (V == 10) asm volatile(FMA4 "s_nop 0\n s_nop 0\n s_nop 0\n" OPS);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int V>
static double run(const char *name, double *out, double *in, long long *clk, double base) {
    const int iters = 20000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k<V>), dim3(1), dim3(64), 0, 0, out, in, iters, clk);
        CHECK(hipDeviceSynchronize());
    }
    long long c;
    CHECK(hipMemcpy(&c, clk, sizeof c, hipMemcpyDeviceToHost));
    const double per = double(c) / iters / 8.0;
    printf("%-64s %6.2f ticks per block  (+%.2f over 4 fma)\n", name, per, per - base);
    return per;
}

int main() {
    double *out, *in;
    long long *clk;
    CHECK(hipMalloc(&out, 64 * sizeof(double)));
    CHECK(hipMalloc(&in, 64 * sizeof(double)));
    CHECK(hipMalloc(&clk, 64 * sizeof(long long)));
    double h[64];
    for (int i = 0; i < 64; i++) h[i] = 0.5 + 0.001 * i;
    h[0] = 0.999; h[1] = 1e-3; h[2] = 1e300;
    CHECK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    double b = run<0>("4 fma", out, in, clk, 0);
    run<5>("4 fma, v_cmp", out, in, clk, b);
    run<1>("4 fma, v_cmp, saveexec, cbranch_execnz, s_or exec   (compiler's)", out, in, clk, b);
    run<6>("4 fma, v_cmp, saveexec, s_or exec                   (no branch)", out, in, clk, b);
    run<2>("4 fma, v_cmp, cbranch_vccnz", out, in, clk, b);
    run<7>("4 fma, v_cmp, s_and vcc exec, cbranch_scc1", out, in, clk, b);
    run<3>("v_cmp, 4 fma, saveexec, cbranch_execnz, s_or exec   (compare early)", out, in, clk, b);
    run<8>("v_cmp->sgpr, 4 fma, saveexec, cbranch_execnz, s_or exec", out, in, clk, b);
    run<4>("v_cmp, 4 fma, cbranch_vccnz                         (compare early)", out, in, clk, b);
    run<9>("4 fma, cbranch_execz", out, in, clk, b);
    run<10>("4 fma, 3 s_nop", out, in, clk, b);
    return 0;
}
