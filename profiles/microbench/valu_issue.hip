// valu_issue.hip -- issue cost (cycles per wave64 instruction on one SIMD) of the VALU opcodes the Philox /
// Box-Muller code of the noise variants is made of.  One wave per SIMD, 8 independent chains per opcode.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_issue profiles/microbench/valu_issue.hip && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BENCH(NAME, ASM)                                                                              \
  __global__ void k_##NAME(uint64_t *out, uint32_t seed) {                                            \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15; \
    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3, w4 = a4, w5 = a5, w6 = a6, w7 = a7;                  \
    const uint32_t m = 0xD2511F53u;                                                                   \
    uint64_t t0 = __builtin_amdgcn_s_memtime();                                                       \
    for (int i = 0; i < 256; ++i) { ASM }                                                             \
    uint64_t t1 = __builtin_amdgcn_s_memtime();                                                       \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                  \
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7)) == 0x12345u) out[1000] = 1; \
  }

#define MUL_LO(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##j) : "v"(m));
#define MUL_HI(j) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##j) : "v"(m));
#define MAD64(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(w##j) : "v"(a##j), "v"(m) : "vcc");
#define MUL24(j) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##j) : "v"(m));
#define MULHI24(j) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a##j) : "v"(m));
#define XOR(j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##j) : "v"(m));
#define FMA(j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##j) : "v"(m));
#define PKFMA(j) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w##j));
#define SIN(j) asm volatile("v_sin_f32 %0, %0" : "+v"(a##j));
#define LOG(j) asm volatile("v_log_f32 %0, %0" : "+v"(a##j));
#define SQRT(j) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##j));
#define CVT(j) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##j));

BENCH(mul_lo_u32, REP8(MUL_LO))
BENCH(mul_hi_u32, REP8(MUL_HI))
BENCH(mad_u64_u32, REP8(MAD64))
BENCH(mul_u32_u24, REP8(MUL24))
BENCH(mul_hi_u32_u24, REP8(MULHI24))
BENCH(xor_b32, REP8(XOR))
BENCH(fma_f32, REP8(FMA))
BENCH(pk_fma_f32, REP8(PKFMA))
BENCH(sin_f32, REP8(SIN))
BENCH(log_f32, REP8(LOG))
BENCH(sqrt_f32, REP8(SQRT))
BENCH(cvt_f32_u32, REP8(CVT))

#define RUN(NAME, WAVES)                                                                              \
  {                                                                                                   \
    hipLaunchKernelGGL(k_##NAME, dim3(256), dim3(64 * WAVES), 0, 0, d, 1u);                           \
    hipLaunchKernelGGL(k_##NAME, dim3(256), dim3(64 * WAVES), 0, 0, d, 1u);                           \
    hipDeviceSynchronize();                                                                           \
    uint64_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);                               \
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];                                    \
    printf("%-18s %d waves/CU: %6.2f s_memtime ticks per instruction per wave\n", #NAME, WAVES, s / 256 / (256.0 * 8)); \
  }

int main() {
  uint64_t *d; hipMalloc(&d, 2048 * 8);
  // (s_memtime ticks are not exactly shader cycles: read the rows relative to v_fma_f32 = 4 cycles)
  RUN(fma_f32, 4) RUN(xor_b32, 4) RUN(pk_fma_f32, 4) RUN(cvt_f32_u32, 4)
  RUN(mul_lo_u32, 4) RUN(mul_hi_u32, 4) RUN(mad_u64_u32, 4) RUN(mul_u32_u24, 4) RUN(mul_hi_u32_u24, 4)
  RUN(sin_f32, 4) RUN(log_f32, 4) RUN(sqrt_f32, 4)
  RUN(fma_f32, 8) RUN(mul_lo_u32, 8) RUN(mad_u64_u32, 8)
  return 0;
}
