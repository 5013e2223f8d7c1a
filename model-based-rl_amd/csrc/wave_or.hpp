// wave_or(x): bitwise OR of x over the ACTIVE lanes of the wave, returned as a wave-uniform (scalar) value.
// The rollout kernels use it once per dynamics evaluation to learn which constraint rows are active in
// ANY lane, so that the per-row tests of the solver passes become scalar branches (s_bitcmp / s_cbranch_scc, ~11 clocks
// for a row nobody has) instead of vector compares feeding s_and_saveexec (~50 clocks each for a wave that has its SIMD
// to itself: tools/micro/branch_cost2.hip).
// Through one LDS word per workgroup (the rollout kernels' workgroups are single waves): ds_write 0, ds_or, ds_read.  A
// wave's LDS instructions execute in program order, so no barrier is needed; it works under any EXEC mask and writes no
// register of a lane that is not executing (a whole-wave DPP scan would: inactive lanes keep live values in registers the
// allocator considers free inside a divergent region).
#pragma once
#include <cstdint>

namespace mbrl {
#if defined(__HIP_DEVICE_COMPILE__)
// (written as instructions: the compiler's atomic optimiser would turn a wave-wide atomicOr on one address into a scalar
// loop over all 64 lanes followed by one lane's atomic -- 400 instructions)
// Split in two so that most of the LDS round trip runs under other work: wave_or_begin issues the store and the OR (no
// result, so nothing the register allocator could copy before it has arrived), wave_or_end reads the word, waits and
// broadcasts.  The compiler does not know about these LDS operations; its own s_waitcnt lgkmcnt(N) stay correct because
// LDS operations complete in order (more outstanding operations only make such a wait cover more than it needs to).
__device__ __forceinline__ uint32_t wave_or_word() {
    __shared__ uint32_t word;
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(&word);
}
__device__ __forceinline__ void wave_or_begin(uint32_t x) {
    asm volatile("ds_write_b32 %0, %1\n\tds_or_b32 %0, %2" : : "v"(wave_or_word()), "v"(0u), "v"(x) : "memory");
}
__device__ __forceinline__ uint32_t wave_or_end() {
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(wave_or_word()) : "memory");
    return __builtin_amdgcn_readfirstlane(r);
}
__device__ __forceinline__ uint32_t wave_or(uint32_t x) { wave_or_begin(x); return wave_or_end(); }
#else
inline uint32_t wave_or(uint32_t x) { return x; }   // host twin: one rollout per "wave" (see Lean::forward for begin / end)
#endif
}  // namespace mbrl
