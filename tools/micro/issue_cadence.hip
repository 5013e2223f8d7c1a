// Microbenchmark 3 (not part of the product): issue cadence of the instruction kinds the rollout kernel is made of,
// for a wave that has its SIMD to itself and for two / four waves sharing one SIMD.  One workgroup on one CU:
// 64 threads = 1 wave; 512 threads = 8 waves = 2 per SIMD; 1024 threads = 16 waves = 4 per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

#define R8(s) s s s s s s s s
#define OPS32 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : : "vcc", "a0", "a1", "a2", "a3"
#define OPS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(sa) : "vcc", "a0", "a1", "a2", "a3", "s20", "s21"

template <int V>
__global__ __launch_bounds__(1024) void k(double *out, const double *in, int iters, long long *clk) {
    double a = in[0], b = in[1];
    double sa = in[0];
    double x0 = in[3] + threadIdx.x, x1 = in[4] + threadIdx.x, x2 = in[5] + threadIdx.x, x3 = in[6] + threadIdx.x;
    double x4 = in[7] + threadIdx.x, x5 = in[8] + threadIdx.x, x6 = in[9] + threadIdx.x, x7 = in[10] + threadIdx.x;
    float f0 = float(x0), f1 = float(x1), f2 = float(x2), f3 = float(x3), f4 = float(x4), f5 = float(x5), f6 = float(x6), f7 = float(x7);
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // 8 instructions per block, 8 independent destinations
            if (V == 0) asm volatile("v_fma_f64 %0, %8, %0, %9\n v_fma_f64 %1, %8, %1, %9\n v_fma_f64 %2, %8, %2, %9\n v_fma_f64 %3, %8, %3, %9\n"
                                     "v_fma_f64 %4, %8, %4, %9\n v_fma_f64 %5, %8, %5, %9\n v_fma_f64 %6, %8, %6, %9\n v_fma_f64 %7, %8, %7, %9\n" OPS);
            if (V == 1) asm volatile("v_fma_f64 %0, %10, %0, %9\n v_fma_f64 %1, %10, %1, %9\n v_fma_f64 %2, %10, %2, %9\n v_fma_f64 %3, %10, %3, %9\n"
                                     "v_fma_f64 %4, %10, %4, %9\n v_fma_f64 %5, %10, %5, %9\n v_fma_f64 %6, %10, %6, %9\n v_fma_f64 %7, %10, %7, %9\n" OPS);
            if (V == 2) asm volatile("v_mul_f64 %0, %8, %0\n v_mul_f64 %1, %8, %1\n v_mul_f64 %2, %8, %2\n v_mul_f64 %3, %8, %3\n"
                                     "v_mul_f64 %4, %8, %4\n v_mul_f64 %5, %8, %5\n v_mul_f64 %6, %8, %6\n v_mul_f64 %7, %8, %7\n" OPS);
            if (V == 3) asm volatile("v_add_f64 %0, %9, %0\n v_add_f64 %1, %9, %1\n v_add_f64 %2, %9, %2\n v_add_f64 %3, %9, %3\n"
                                     "v_add_f64 %4, %9, %4\n v_add_f64 %5, %9, %5\n v_add_f64 %6, %9, %6\n v_add_f64 %7, %9, %7\n" OPS);
            if (V == 4) asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                                     "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n" OPS32);
            if (V == 5) asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n"
                                     "v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0\n" OPS);
            if (V == 6) asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3\n"
                                     "v_accvgpr_read_b32 %4, a0\n v_accvgpr_read_b32 %5, a1\n v_accvgpr_read_b32 %6, a2\n v_accvgpr_read_b32 %7, a3\n" OPS32);
            if (V == 7) asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n"
                                     "v_cndmask_b32 %4, %5, %6, vcc\n v_cndmask_b32 %5, %6, %7, vcc\n v_cndmask_b32 %6, %7, %0, vcc\n v_cndmask_b32 %7, %0, %1, vcc\n" OPS32);
            if (V == 8) asm volatile("s_mov_b32 s20, 0x3f800000\n s_mov_b32 s21, 0x3f800001\n s_mov_b32 s20, 0x3f800002\n s_mov_b32 s21, 0x3f800003\n"
                                     "s_mov_b32 s20, 0x3f800004\n s_mov_b32 s21, 0x3f800005\n s_mov_b32 s20, 0x3f800006\n s_mov_b32 s21, 0x3f800007\n" ::: "s20", "s21");
            if (V == 9) asm volatile("v_fma_f64 %0, %8, %0, %9\n s_mov_b32 s20, 0x3f800001\n v_fma_f64 %2, %8, %2, %9\n s_mov_b32 s21, 0x3f800003\n"
                                     "v_fma_f64 %4, %8, %4, %9\n s_mov_b32 s20, 0x3f800005\n v_fma_f64 %6, %8, %6, %9\n s_mov_b32 s21, 0x3f800007\n" OPS);
            if (V == 10) asm volatile("v_fma_f64 %0, %8, %0, %9\n v_fma_f64 %0, %8, %0, %9\n v_fma_f64 %0, %8, %0, %9\n v_fma_f64 %0, %8, %0, %9\n"
                                      "v_fma_f64 %0, %8, %0, %9\n v_fma_f64 %0, %8, %0, %9\n v_fma_f64 %0, %8, %0, %9\n v_fma_f64 %0, %8, %0, %9\n" OPS);
            if (V == 11) asm volatile("v_fma_f64 %0, %8, %0, %9\n v_mov_b64 %1, %2\n v_fma_f64 %3, %8, %3, %9\n v_mov_b64 %4, %5\n"
                                      "v_fma_f64 %6, %8, %6, %9\n v_mov_b64 %7, %2\n v_fma_f64 %0, %8, %0, %9\n v_mov_b64 %1, %5\n" OPS);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + double(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if ((threadIdx.x & 63) == 0) clk[threadIdx.x >> 6] = t1 - t0;
}

template <int V>
static void run(const char *name, double *out, double *in, long long *clk) {
    const int iters = 20000;
    printf("%-44s", name);
    for (int threads : {64, 256, 512, 1024}) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL((k<V>), dim3(1), dim3(threads), 0, 0, out, in, iters, clk);
            CHECK(hipDeviceSynchronize());
        }
        long long c[16];
        CHECK(hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost));
        long long m = 0;
        for (int w = 0; w < threads / 64; w++) m = c[w] > m ? c[w] : m;
        printf("  %2d wave%s %6.2f", threads / 64, threads == 64 ? " " : "s", double(m) / iters / 32.0);
    }
    printf("   clocks per instruction per wave\n");
}

int main() {
    double *out, *in;
    long long *clk;
    CHECK(hipMalloc(&out, 1024 * sizeof(double)));
    CHECK(hipMalloc(&in, 64 * sizeof(double)));
    CHECK(hipMalloc(&clk, 64 * sizeof(long long)));
    double h[64];
    for (int i = 0; i < 64; i++) h[i] = 0.5 + 0.001 * i;
    h[0] = 0.999; h[1] = 1e-3;
    CHECK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    printf("one workgroup on one CU: 1 wave = a SIMD to itself, 4 waves = one per SIMD, 8 = two per SIMD, 16 = four per SIMD\n");
    run<0>("v_fma_f64 (3 vgpr sources), 8 chains", out, in, clk);
    run<1>("v_fma_f64 (sgpr, vgpr, vgpr), 8 chains", out, in, clk);
    run<10>("v_fma_f64, 1 dependent chain", out, in, clk);
    run<2>("v_mul_f64", out, in, clk);
    run<3>("v_add_f64", out, in, clk);
    run<4>("v_mov_b32", out, in, clk);
    run<5>("v_mov_b64", out, in, clk);
    run<6>("v_accvgpr_write/read_b32", out, in, clk);
    run<7>("v_cndmask_b32", out, in, clk);
    run<8>("s_mov_b32 literal", out, in, clk);
    run<9>("v_fma_f64 / s_mov_b32 alternating", out, in, clk);
    run<11>("v_fma_f64 / v_mov_b64 alternating", out, in, clk);
    return 0;
}
