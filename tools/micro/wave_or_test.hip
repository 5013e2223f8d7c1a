// Check (not part of the product's tests): the wave-wide OR used by the rollout kernels' row tests, under arbitrary EXEC
// masks, against the host's OR.  Prints "ok" or the first mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)
#define MBRL_WAVE_OR_ONLY
#include "../../model-based-rl_amd/csrc/wave_or.hpp"

__global__ __launch_bounds__(64) void k(const uint32_t *x, const unsigned long long *active, uint32_t *out, long long *clk) {
    const int t = threadIdx.x;
    const unsigned long long a = active[blockIdx.x];
    uint32_t r = 0xdeadbeefu;
    long long t0 = 0, t1 = 0;
    if ((a >> t) & 1ull) {
        t0 = __builtin_amdgcn_s_memtime();
        r = mbrl::wave_or(x[blockIdx.x * 64 + t]);
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[blockIdx.x * 64 + t] = r;
    if (t == __builtin_ctzll(a)) clk[blockIdx.x] = t1 - t0;
}

int main() {
    const int B = 4096;
    uint32_t *hx = (uint32_t *)malloc(B * 64 * 4), *ho = (uint32_t *)malloc(B * 64 * 4);
    unsigned long long *ha = (unsigned long long *)malloc(B * 8);
    long long *hc = (long long *)malloc(B * 8);
    srand(1);
    for (int b = 0; b < B; b++) {
        unsigned long long a = 0;
        const int kind = b % 8;
        for (int t = 0; t < 64; t++) {
            hx[b * 64 + t] = (rand() % 4 == 0) ? (1u << (rand() % 32)) | (1u << (rand() % 32)) : 0u;
            bool on = kind == 0 ? true : kind == 1 ? t < 32 : kind == 2 ? t >= 32 : kind == 3 ? t <= b % 64 : kind == 4 ? t == b % 64
                      : kind == 5 ? (rand() & 1) : kind == 6 ? (rand() % 8 == 0) : t < 63;
            if (on) a |= 1ull << t;
        }
        if (!a) a = 1ull << (b % 64);
        ha[b] = a;
    }
    uint32_t *dx, *dout; unsigned long long *da; long long *dc;
    CHECK(hipMalloc(&dx, B * 64 * 4)); CHECK(hipMalloc(&dout, B * 64 * 4)); CHECK(hipMalloc(&da, B * 8)); CHECK(hipMalloc(&dc, B * 8));
    CHECK(hipMemcpy(dx, hx, B * 64 * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(da, ha, B * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, dx, da, dout, dc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(ho, dout, B * 64 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hc, dc, B * 8, hipMemcpyDeviceToHost));
    long long csum = 0;
    for (int b = 0; b < B; b++) {
        uint32_t want = 0;
        for (int t = 0; t < 64; t++) if ((ha[b] >> t) & 1ull) want |= hx[b * 64 + t];
        for (int t = 0; t < 64; t++) {
            const uint32_t w = ((ha[b] >> t) & 1ull) ? want : 0xdeadbeefu;
            if (ho[b * 64 + t] != w) { printf("MISMATCH block %d lane %d active %016llx: got %08x want %08x\n", b, t, ha[b], ho[b * 64 + t], w); return 1; }
        }
        csum += hc[b];
    }
    printf("ok: %d waves, arbitrary EXEC masks; %.1f ticks per call\n", B, double(csum) / B);
    return 0;
}
