// periodicity_model.c -- CPU model of the LONG stage's exact-periodicity check: how many iterations
// the never-escaping samples outside the cardioid and the period-2 disc cost per sample drawn, under several
// schemes of saved points, against iterating them to max_iter and against the best any scheme that compares chunk
// ends could do (the first exact repeat of ANY earlier chunk end).  Same arithmetic as the kernels (fma contraction
// of device_math.h), uniform samples on [-2,2]^2, max_iter 20000, LONG from iteration 20.
//   gcc -O2 -march=native -ffp-contract=off -o periodicity_model tools/periodicity_model.c -lm
//   ./periodicity_model [samples] [chunk]
// 600000 samples, chunk 60:  to max_iter 163.0 | Brent, top 2 bits (the kernel) 17.64 | + the previous chunk end 17.14 |
// two Brent points in turn 17.08 | top 3 bits 18.54 | powers of two 19.02 | ideal 15.43 iterations per sample.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static int sched(unsigned c, int nbits) {  // chunk count has no set bit below its top nbits
  int top = 31 - __builtin_clz(c);
  int keep = nbits - 1;
  unsigned low = top >= keep ? c & ((1u << (top - keep)) - 1u) : 0;
  return low == 0;
}
// modes: 0 none; 1 Brent(2 bits); 2 Brent + previous-chunk point (two points); 3 two Brent points replaced in turn;
// 4 Brent(3 bits); 5 Brent(1 bit); 6 Brent + point of two chunks ago (p | 120); 7 ideal: first exact repeat of ANY earlier chunk end
int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 400000;
  const int max_iter = 20000, start = 20, chunk = argc > 2 ? atoi(argv[2]) : 60;
  enum { M = 8 };
  uint64_t s = 88172645463325252ull;
  double tot[M] = {0};
  long never = 0, samples = 0;
  static double hr[400], hi_[400];
  for (long k = 0; k < n; ++k) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double cr = (double) (s >> 11) * (4.0 / 9007199254740992.0) - 2.0;
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double ci = (double) (s >> 11) * (4.0 / 9007199254740992.0) - 2.0;
    ++samples;
    const double x = cr - 0.25, q = x * x + ci * ci;
    if (q * (q + x) < 0.25 * ci * ci) continue;
    if ((cr + 1.0) * (cr + 1.0) + ci * ci < 0.0625) continue;
    for (int mode = 0; mode < M; ++mode) {
      double r = cr, i = ci, sr = 0, si = 0, pr = 0, pi = 0, p2r = 0, p2i = 0, tr = 0, ti = 0;
      int it = 0, escaped = 0, turn = 0, nh = 0;
      unsigned c = 0;
      while (it < max_iter) {
        const double ii = i * i, t = fma(r, r, -ii);
        const double ni = fma(r + r, i, ci), nr = cr + t;
        r = nr; i = ni; ++it;
        if (fma(i, i, r * r) > 4.0) { escaped = 1; break; }
        if (mode == 0) continue;
        if (it == start) { sr = tr = pr = p2r = r; si = ti = pi = p2i = i; c = 0; nh = 0; hr[nh] = r; hi_[nh++] = i; }
        if (it > start && (it - start) % chunk == 0) {
          ++c;
          const int eq_s = bits(r) == bits(sr) && bits(i) == bits(si);
          const int eq_p = bits(r) == bits(pr) && bits(i) == bits(pi);
          const int eq_p2 = bits(r) == bits(p2r) && bits(i) == bits(p2i);
          const int eq_t = bits(r) == bits(tr) && bits(i) == bits(ti);
          if (mode == 1 || mode == 4 || mode == 5) { if (eq_s) break; }
          if (mode == 2) { if (eq_s || eq_p) break; }
          if (mode == 3) { if (eq_s || eq_t) break; }
          if (mode == 6) { if (eq_s || eq_p2) break; }
          if (mode == 7) {
            int hit = 0;
            for (int h = 0; h < nh; ++h) if (bits(r) == bits(hr[h]) && bits(i) == bits(hi_[h])) { hit = 1; break; }
            if (hit) break;
            if (nh < 400) { hr[nh] = r; hi_[nh++] = i; }
          }
          const int nb = mode == 4 ? 3 : mode == 5 ? 1 : 2;
          if (sched(c, nb)) {
            if (mode == 3) { if (turn) { tr = r; ti = i; } else { sr = r; si = i; } turn ^= 1; }
            else { sr = r; si = i; }
          }
          p2r = pr; p2i = pi; pr = r; pi = i;
        }
      }
      if (escaped) break;
      if (mode == 0) ++never;
      tot[mode] += it;
    }
  }
  const char *names[M] = {"full", "Brent 2 bits", "Brent + previous chunk", "two Brent points in turn", "Brent 3 bits", "Brent 1 bit",
                          "Brent + two chunks ago", "ideal (any earlier chunk end)"};
  printf("samples %ld never-escaping %ld chunk %d\n", samples, never, chunk);
  for (int m = 0; m < M; ++m) printf("  %-32s %.2f iterations per sample\n", names[m], tot[m] / samples);
  return 0;
}
