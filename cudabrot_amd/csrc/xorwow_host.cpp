// xorwow_host.cpp -- host-side preparation for the device RNG init kernel.
//
// rocRAND's XORWOW (4.2, /opt/rocm/include/rocrand/rocrand_xorwow.h) advances 160 xorshift bits
// linearly over GF(2); "subsequence s" is the state after s * 2^67 steps (:149-158).  rocRAND ships
// 32 precomputed matrices A^(2^67 * 4^i) and applies each up to three times per base-4 digit.  Here
// the matrices for every BINARY digit, A^(2^(67+b)), are derived from the step function by repeated
// squaring (about 2 ms), so a jump costs one matrix-vector product per set bit and nothing depends
// on rocRAND's tables at run time (tests compare the two).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "kernels.h"

namespace cb {

namespace {

struct Gf2Mat {
  uint32_t img[160][5];  // img[word*32+bit] = image of that basis bit (rocrand_xorwow.h:49-66)
};

// rocrand_xorwow.h:165-177, xorshift part only
void xorshift_step(uint32_t x[5]) {
  const uint32_t t = x[0] ^ (x[0] >> 2);
  x[0] = x[1];
  x[1] = x[2];
  x[2] = x[3];
  x[3] = x[4];
  x[4] = (x[4] ^ (x[4] << 4)) ^ (t ^ (t << 1));
}

void mat_vec(const Gf2Mat &m, uint32_t v[5]) {
  uint32_t r[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < 160; ++b) {
    if (v[b >> 5] & (1u << (b & 31))) {
      for (int k = 0; k < 5; ++k) r[k] ^= m.img[b][k];
    }
  }
  memcpy(v, r, sizeof(r));
}

void mat_square(Gf2Mat &out, const Gf2Mat &a) {
  for (int b = 0; b < 160; ++b) {
    uint32_t v[5];
    memcpy(v, a.img[b], sizeof(v));
    mat_vec(a, v);
    memcpy(out.img[b], v, sizeof(v));
  }
}

}  // namespace

void build_sequence_jump_matrices(uint32_t *out) {
  Gf2Mat cur, next;
  for (int b = 0; b < 160; ++b) {
    uint32_t v[5] = {0, 0, 0, 0, 0};
    v[b >> 5] = 1u << (b & 31);
    xorshift_step(v);
    memcpy(cur.img[b], v, sizeof(v));
  }
  // cur = A^(2^0); square 67 times -> A^(2^67)
  for (int e = 0; e < 67; ++e) {
    mat_square(next, cur);
    cur = next;
  }
  for (int b = 0; b < kSeqJumpMatrices; ++b) {
    memcpy(out + (size_t) b * kMatrixWords, cur.img, sizeof(cur.img));
    mat_square(next, cur);
    cur = next;
  }
}

// rocrand_xorwow.h:104-123
void seed_state(uint64_t seed, uint32_t x[5], uint32_t *d) {
  x[0] = 123456789u;
  x[1] = 362436069u;
  x[2] = 521288629u;
  x[3] = 88675123u;
  x[4] = 5783321u;
  *d = 6615241u;
  const uint32_t s0 = ((uint32_t) seed) ^ 0x2c7f967fu;
  const uint32_t s1 = ((uint32_t) (seed >> 32)) ^ 0xa03697cbu;
  const uint32_t t0 = 1228688033u * s0;
  const uint32_t t1 = 2073658381u * s1;
  x[0] += t0;
  x[1] ^= t0;
  x[2] += t1;
  x[3] ^= t1;
  x[4] += t0;
  *d += t1 + t0;
}

}  // namespace cb
