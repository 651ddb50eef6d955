// Instruction-issue-rate microbenchmark for the integer ops the field arithmetic is built from
// (gfx950).  Not part of the product; its numbers feed DESIGN.md's integer-ALU roofline.
//   hipcc --offload-arch=gfx950 -O3 tools/instr_rate.hip -o tools/instr_rate && tools/instr_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// every kernel: `iters` x 64 instructions per lane, 8 independent chains
__global__ void k_mad_u64(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
  uint32_t x = a + threadIdx.x, y = b;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
                      "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
                      "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\t"
                      "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                      "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\t"
                      "v_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                      "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\t"
                      "v_mad_u64_u32 %7, vcc, %8, %9, %7\n\t"
                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6),
                        "+v"(r7)
                      : "v"(x), "v"(y)
                      : "vcc");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

#define K32(NAME, INSTR)                                                                       \
  __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b, int iters) {                     \
    uint32_t r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;         \
    uint32_t x = a + threadIdx.x, y = b;                                                       \
    for (int i = 0; i < iters; i++) {                                                          \
      REP8(asm volatile(INSTR " %0, %8, %0\n\t" INSTR " %1, %9, %1\n\t" INSTR " %2, %8, %2\n\t" \
                        INSTR " %3, %9, %3\n\t" INSTR " %4, %8, %4\n\t" INSTR " %5, %9, %5\n\t" \
                        INSTR " %6, %8, %6\n\t" INSTR " %7, %9, %7\n\t"                        \
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), \
                          "+v"(r7)                                                             \
                        : "v"(x), "v"(y));)                                                    \
    }                                                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;        \
  }
K32(k_add_u32, "v_add_u32")
K32(k_mul_lo, "v_mul_lo_u32")
K32(k_mul_hi, "v_mul_hi_u32")
K32(k_and, "v_and_b32")
K32(k_xor, "v_xor_b32")

__global__ void k_mov(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint32_t r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\t"
                      "v_mov_b32 %3, %4\n\tv_mov_b32 %4, %5\n\tv_mov_b32 %5, %6\n\t"
                      "v_mov_b32 %6, %7\n\tv_mov_b32 %7, %0\n\t"
                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6),
                        "+v"(r7));)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

#define K64(NAME, BODY)                                                                        \
  __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b, int iters) {                     \
    uint64_t r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;         \
    uint64_t x = a + threadIdx.x;                                                              \
    uint32_t y = b;                                                                            \
    for (int i = 0; i < iters; i++) {                                                          \
      REP8(asm volatile(BODY : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),     \
                               "+v"(r6), "+v"(r7)                                             \
                             : "v"(x), "v"(y));)                                               \
    }                                                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;        \
  }
K64(k_lshl_add_u64,
    "v_lshl_add_u64 %0, %0, 0, %8\n\tv_lshl_add_u64 %1, %1, 0, %8\n\t"
    "v_lshl_add_u64 %2, %2, 0, %8\n\tv_lshl_add_u64 %3, %3, 0, %8\n\t"
    "v_lshl_add_u64 %4, %4, 0, %8\n\tv_lshl_add_u64 %5, %5, 0, %8\n\t"
    "v_lshl_add_u64 %6, %6, 0, %8\n\tv_lshl_add_u64 %7, %7, 0, %8\n\t")
K64(k_lshr_b64,
    "v_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %1, 1, %1\n\tv_lshrrev_b64 %2, 1, %2\n\t"
    "v_lshrrev_b64 %3, 1, %3\n\tv_lshrrev_b64 %4, 1, %4\n\tv_lshrrev_b64 %5, 1, %5\n\t"
    "v_lshrrev_b64 %6, 1, %6\n\tv_lshrrev_b64 %7, 1, %7\n\t")

// add with carry chain: v_add_co_u32 / v_addc_co_u32 alternating (8-limb add shape)
__global__ void k_addc(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint32_t r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
  uint32_t x = a + threadIdx.x;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("v_add_co_u32 %0, vcc, %8, %0\n\tv_addc_co_u32 %1, vcc, %8, %1, vcc\n\t"
                      "v_addc_co_u32 %2, vcc, %8, %2, vcc\n\tv_addc_co_u32 %3, vcc, %8, %3, vcc\n\t"
                      "v_addc_co_u32 %4, vcc, %8, %4, vcc\n\tv_addc_co_u32 %5, vcc, %8, %5, vcc\n\t"
                      "v_addc_co_u32 %6, vcc, %8, %6, vcc\n\tv_addc_co_u32 %7, vcc, %8, %7, vcc\n\t"
                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6),
                        "+v"(r7)
                      : "v"(x)
                      : "vcc");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

// f64 FMA rate (the 52-bit-limb alternative)
__global__ void k_fma_f64(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  double r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
  double x = 1.0 + a * 1e-9, y = b * 1e-9;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\t"
                      "v_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                      "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\t"
                      "v_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t"
                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6),
                        "+v"(r7)
                      : "v"(x), "v"(y));)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] =
      (uint64_t)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);
}

// signed variant, and dependent chains (all 8 instructions of a group on one / two accumulators):
// what a single-accumulator Montgomery column chain looks like to the SIMD
#define KMAD(NAME, OP, R0, R1, R2, R3, R4, R5, R6, R7)                                          \
  __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b, int iters) {                     \
    uint64_t r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;         \
    uint32_t x = a + threadIdx.x, y = b;                                                       \
    for (int i = 0; i < iters; i++) {                                                          \
      REP8(asm volatile(OP " %" #R0 ", vcc, %8, %9, %" #R0 "\n\t" OP " %" #R1 ", vcc, %8, %9, %" #R1 "\n\t" \
                        OP " %" #R2 ", vcc, %8, %9, %" #R2 "\n\t" OP " %" #R3 ", vcc, %8, %9, %" #R3 "\n\t" \
                        OP " %" #R4 ", vcc, %8, %9, %" #R4 "\n\t" OP " %" #R5 ", vcc, %8, %9, %" #R5 "\n\t" \
                        OP " %" #R6 ", vcc, %8, %9, %" #R6 "\n\t" OP " %" #R7 ", vcc, %8, %9, %" #R7 "\n\t" \
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), \
                          "+v"(r7)                                                             \
                        : "v"(x), "v"(y)                                                       \
                        : "vcc");)                                                             \
    }                                                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;        \
  }
KMAD(k_mad_i64, "v_mad_i64_i32", 0, 1, 2, 3, 4, 5, 6, 7)
KMAD(k_mad_i64_dep1, "v_mad_i64_i32", 0, 0, 0, 0, 0, 0, 0, 0)
KMAD(k_mad_i64_dep2, "v_mad_i64_i32", 0, 1, 0, 1, 0, 1, 0, 1)
KMAD(k_mad_u64_dep1, "v_mad_u64_u32", 0, 0, 0, 0, 0, 0, 0, 0)
K64(k_ashr_i64,
    "v_ashrrev_i64 %0, 1, %0\n\tv_ashrrev_i64 %1, 1, %1\n\tv_ashrrev_i64 %2, 1, %2\n\t"
    "v_ashrrev_i64 %3, 1, %3\n\tv_ashrrev_i64 %4, 1, %4\n\tv_ashrrev_i64 %5, 1, %5\n\t"
    "v_ashrrev_i64 %6, 1, %6\n\tv_ashrrev_i64 %7, 1, %7\n\t")

template <class K>
static void run(const char* name, K kern, int waves_per_simd) {
  const int iters = 2000;
  const int threads = 256, blocks = 256 * waves_per_simd;  // 4 waves per block -> one per SIMD
  uint64_t* out;
  hipMalloc(&out, (size_t)threads * blocks * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double instr = (double)iters * 64.0 * threads * blocks;  // lane-instructions
  double per_s = instr / (ms * 1e-3);
  // cycles per wave-instruction per SIMD at 2.4 GHz, 1024 SIMDs
  double wave_instr_per_simd_per_s = per_s / 64.0 / 1024.0;
  printf("%-16s waves/SIMD=%d  %.3e lane-ops/s  %.2f cycles/wave-instr/SIMD @2.4GHz\n", name,
         waves_per_simd, per_s, 2.4e9 / wave_instr_per_simd_per_s);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run("v_mad_u64_u32", k_mad_u64, w);
    run("v_mad_i64_i32", k_mad_i64, w);
    run("v_mad_i64 1 chain", k_mad_i64_dep1, w);
    run("v_mad_i64 2 chains", k_mad_i64_dep2, w);
    run("v_mad_u64 1 chain", k_mad_u64_dep1, w);
    run("v_ashrrev_i64", k_ashr_i64, w);
    run("v_mul_lo_u32", k_mul_lo, w);
    run("v_mul_hi_u32", k_mul_hi, w);
    run("v_add_u32", k_add_u32, w);
    run("v_and_b32", k_and, w);
    run("v_mov_b32", k_mov, w);
    run("v_lshl_add_u64", k_lshl_add_u64, w);
    run("v_lshrrev_b64", k_lshr_b64, w);
    run("v_add(c)_co_u32", k_addc, w);
    run("v_fma_f64", k_fma_f64, w);
  }
  return 0;
}
