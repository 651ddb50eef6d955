// Host driver for csrc/emul.h (tests/test_native_emul.py): lines "a b p" in hex -> "q r" in hex,
// a, b < 2^384, 2^224 <= p < 2^256.  The limbs of a and b also go through emul_acc_at (64-bit limb
// offsets with overflowing limbs) when the line starts with "L": "L na nb limb... p".
#include <cstdio>
#include <cstring>
#include <string>
#include <iostream>
#include <sstream>
#include <vector>
#include "emul.h"

static void parse(const std::string& h, uint32_t* w, int n) {
  for (int i = 0; i < n; i++) w[i] = 0;
  int pos = 0;
  for (int i = (int)h.size() - 1; i >= 0; i--, pos++) {
    char c = h[i];
    uint32_t v = c <= '9' ? c - '0' : (c | 32) - 'a' + 10;
    if (pos / 8 < n) w[pos / 8] |= v << (4 * (pos % 8));
  }
}
static void print(const uint32_t* w, int n) {
  for (int i = n - 1; i >= 0; i--) printf("%08x", w[i]);
}

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream is(line);
    std::string tok;
    std::vector<std::string> t;
    while (is >> tok) t.push_back(tok);
    if (t.empty()) continue;
    uint32_t A[12], B[12], T[24], Rm[9], P[8];
    if (t[0] == "L") {
      int na = std::stoi(t[1]), nb = std::stoi(t[2]);
      memset(A, 0, sizeof A);
      memset(B, 0, sizeof B);
      for (int i = 0; i < na + nb; i++) {
        uint32_t v[8];
        parse(t[3 + i], v, 8);
        if (i < na)
          zk::emul_acc_at(A, v, i);
        else
          zk::emul_acc_at(B, v, i - na);
      }
      parse(t[3 + na + nb], P, 8);
    } else {
      parse(t[0], A, 12);
      parse(t[1], B, 12);
      parse(t[2], P, 8);
    }
    zk::emul_mul(T, A, B);
    zk::emul_divmod(T, Rm, P);
    print(T, 24);
    printf(" ");
    print(Rm, 8);
    printf("\n");
  }
  return 0;
}
