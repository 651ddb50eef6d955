// Experiment bench (not product code): scheduling variants of the 9 x 29-bit Montgomery product of
// csrc/ff29.h.  hipcc -O3 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000
//   tools/mul_variants.hip -o tools/mul_variants && tools/mul_variants
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../gnark_crypto_primitives_amd/csrc/ff29.h"
using namespace zk;

__device__ __forceinline__ int64_t opq(int64_t x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ int64_t opq_nv(int64_t x) {
  asm("" : "+v"(x));
  return x;
}

// V = 0: ff29.h mul.  1: barrier after every accumulate.  2: barrier after the first product of
// each column (forces the carried accumulator into a mad addend).  3: as 2, non-volatile asm.
// 4: mul_ilp
template <int V>
__device__ __forceinline__ Fq29 mulv(const Fq29& a, const Fq29& b) {
  if (V == 0) return mul(a, b);
  if (V == 4) return mul_ilp(a, b);
  typedef Fq29Params P;
  int64_t acc = 0;
  int32_t m[9];
  Fq29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      acc += (int64_t)a.v[i] * b.v[k - i];
      if (V == 1 || (V == 2 && i == 0)) acc = opq(acc);
      if (V == 3 && i == 0) acc = opq_nv(acc);
    }
#pragma unroll
    for (int i = 0; i < k; i++) {
      acc += (int64_t)m[i] * P::p(k - i);
      if (V == 1) acc = opq(acc);
    }
    m[k] = (int32_t)(((uint32_t)acc * P::inv) & (uint32_t)Fq29::MASK);
    acc += (int64_t)m[k] * P::p(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) {
      acc += (int64_t)a.v[i] * b.v[k - i];
      if (V == 1 || (V == 2 && i == k - 8)) acc = opq(acc);
      if (V == 3 && i == k - 8) acc = opq_nv(acc);
    }
#pragma unroll
    for (int i = k - 8; i < 9; i++) {
      acc += (int64_t)m[i] * P::p(k - i);
      if (V == 1) acc = opq(acc);
    }
    r.v[k - 9] = (int32_t)acc & Fq29::MASK;
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
}

template <int V>
__global__ __launch_bounds__(256) void kern(int32_t* io, int iters) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 x, y;
  for (int i = 0; i < 9; i++) {
    x.v[i] = io[t * 18 + i] & Fq29::MASK;
    y.v[i] = io[t * 18 + 9 + i] & Fq29::MASK;
  }
  for (int k = 0; k < iters; k++) {
    Fq29 z = mulv<V>(x, y);
    x = y;
    y = z;
  }
  for (int i = 0; i < 9; i++) io[t * 18 + i] = y.v[i];
}

template <int V>
static void run(int32_t* d, int32_t* h, size_t n, int iters, const char* name, int32_t* ref) {
  hipMemcpy(d, h, n * 18 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern<V>, dim3(n / 256), dim3(256), 0, 0, d, 8);
  hipMemcpy(d, h, n * 18 * 4, hipMemcpyHostToDevice);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern<V>, dim3(n / 256), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  int32_t out[9];
  hipMemcpy(out, d + 18 * 1234, 36, hipMemcpyDeviceToHost);
  bool same = true;
  if (V == 0) for (int i = 0; i < 9; i++) ref[i] = out[i];
  else for (int i = 0; i < 9; i++) same &= ref[i] == out[i];
  printf("%-28s %8.2f Gmul/s  %s\n", name, (double)n * iters / ms / 1e6, same ? "same" : "DIFFERENT");
}

int main() {
  const size_t n = 1 << 20;
  const int iters = 512;
  int32_t* h = (int32_t*)malloc(n * 18 * 4);
  uint32_t s = 12345;
  for (size_t i = 0; i < n * 18; i++) { s = s * 1664525u + 1013904223u; h[i] = (int32_t)(s >> 3); }
  int32_t* d;
  hipMalloc(&d, n * 18 * 4);
  int32_t ref[9];
  run<0>(d, h, n, iters, "ff29 mul", ref);
  run<1>(d, h, n, iters, "barrier every accumulate", ref);
  run<2>(d, h, n, iters, "barrier first product", ref);
  run<3>(d, h, n, iters, "barrier first, non-volatile", ref);
  run<4>(d, h, n, iters, "mul_ilp", ref);
  return 0;
}
