// Big-integer arithmetic of the OP_EMUL unit (solve.hip): product of two multi-limb integers and its
// quotient and remainder by a 4 x 64-bit modulus -- the mulHint of gnark's std/math/emulated
// [UPSTREAM-RECALL], frontend/api.py::NewHintEmulMul.  Plain 32-bit word arrays with compile-time
// indices only (registers on the device, no scratch).  Host-compilable: tests/native/test_emul.cpp
// runs the same code against Python integers (tests/test_native_emul.py).
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define ZK_EMUL_FN __host__ __device__ __forceinline__
#else
#define ZK_EMUL_FN inline
#endif

namespace zk {

// X += v << (32 OFF): the limb offset of an operand is wave-uniform, so the add-at-offset is a
// switch over four instances
template <int OFF>
ZK_EMUL_FN void emul_acc(uint32_t (&X)[12], const uint32_t (&v)[8]) {
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j + OFF < 12; j++) {
    const uint64_t t = (uint64_t)X[OFF + j] + (j < 8 ? v[j] : 0u) + carry;
    X[OFF + j] = (uint32_t)t;
    carry = t >> 32;
  }
}
ZK_EMUL_FN void emul_acc_at(uint32_t (&X)[12], const uint32_t (&v)[8], uint32_t limb) {
  switch (limb) {
    case 0: emul_acc<0>(X, v); break;
    case 1: emul_acc<2>(X, v); break;
    case 2: emul_acc<4>(X, v); break;
    default: emul_acc<6>(X, v); break;
  }
}
// T <- floor(T / P) (24 words), Rm <- T mod P: schoolbook division in base 2^32 (Knuth's algorithm
// D), every index a compile-time constant.  P is normalised to a set top bit; a quotient digit is
// estimated from the two top words of the running remainder and the top word of P, which is at most
// two too large -- two masked add-backs.  Needs P >= 2^224 (the frontend only builds such units);
// anything else yields garbage, never a fault.
ZK_EMUL_FN void emul_divmod(uint32_t (&T)[24], uint32_t (&Rm)[9], const uint32_t (&P)[8]) {
  const uint32_t s = P[7] ? (uint32_t)__builtin_clz(P[7]) : 0u;
  uint32_t V[8], U[25], Q[17];
#pragma unroll
  for (int i = 7; i >= 0; i--)
    V[i] = (uint32_t)((((uint64_t)P[i] << 32) | (i ? P[i - 1] : 0u)) >> (32 - s));
  U[24] = s ? (T[23] >> (32 - s)) : 0u;
#pragma unroll
  for (int i = 23; i >= 0; i--)
    U[i] = (uint32_t)((((uint64_t)T[i] << 32) | (i ? T[i - 1] : 0u)) >> (32 - s));
  const uint32_t vt = V[7] | (P[7] ? 0u : 0x80000000u);
#pragma unroll
  for (int j = 16; j >= 0; j--) {
    const uint64_t num = ((uint64_t)U[j + 8] << 32) | U[j + 7];
    uint64_t qh = num / vt;
    if (qh > 0xffffffffull) qh = 0xffffffffull;
    uint32_t q = (uint32_t)qh;
    uint64_t carry = 0, borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint64_t pr = (uint64_t)q * V[i] + carry;
      carry = pr >> 32;
      const uint64_t d = (uint64_t)U[j + i] - (uint32_t)pr - borrow;
      U[j + i] = (uint32_t)d;
      borrow = (d >> 32) & 1;
    }
    const uint64_t d = (uint64_t)U[j + 8] - carry - borrow;
    U[j + 8] = (uint32_t)d;
    uint32_t neg = (uint32_t)(d >> 32) & 1u;
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
      const uint32_t mask = 0u - neg;
      q -= neg;
      uint64_t c = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint64_t t = (uint64_t)U[j + i] + (V[i] & mask) + c;
        U[j + i] = (uint32_t)t;
        c = t >> 32;
      }
      const uint64_t t = (uint64_t)U[j + 8] + c;
      U[j + 8] = (uint32_t)t;
      neg &= 1u - (uint32_t)(t >> 32);
    }
    Q[j] = q;
  }
#pragma unroll
  for (int i = 0; i < 8; i++) Rm[i] = (uint32_t)((((uint64_t)U[i + 1] << 32) | U[i]) >> s);
  Rm[8] = 0;
#pragma unroll
  for (int i = 0; i < 24; i++) T[i] = i < 17 ? Q[i] : 0u;
}

// T = A B (12 x 12 words)
ZK_EMUL_FN void emul_mul(uint32_t (&T)[24], const uint32_t (&A)[12], const uint32_t (&B)[12]) {
#pragma unroll
  for (int i = 0; i < 24; i++) T[i] = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
      const uint64_t t = (uint64_t)A[i] * B[j] + T[i + j] + carry;
      T[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    T[i + 12] = (uint32_t)carry;   // untouched so far: earlier rows end at word i + 11
  }
}

}  // namespace zk
