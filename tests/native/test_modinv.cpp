// Host build of csrc/modinv30.h: reads "<field> <64 hex digits>" lines, prints the inverse.
// Driven by tests/test_native_modinv.py, which checks the answers with Python integers.
#include <cstdio>
#include <cstring>

#include "modinv30.h"

int main() {
  char which[8], hex[80];
  while (scanf("%7s %79s", which, hex) == 2) {
    uint32_t x[8], out[8];
    for (int w = 0; w < 8; w++) {
      unsigned v;
      char buf[9];
      memcpy(buf, hex + 8 * (7 - w), 8);
      buf[8] = 0;
      sscanf(buf, "%x", &v);
      x[w] = v;
    }
    if (which[0] == 'r')
      zk::modinv30<zk::ModInvFr>(out, x);
    else
      zk::modinv30<zk::ModInvFq>(out, x);
    for (int w = 7; w >= 0; w--) printf("%08x", out[w]);
    printf("\n");
  }
  return 0;
}
