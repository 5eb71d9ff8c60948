// head_map.c -- a map of parameter cells on which the OUTCOME OF THE FIRST FOUR ITERATIONS of the reference's sample loop
// (cudabrot.cu:390-399: draw, InMainCardioid / InOrder2Bulb, then IterateMandelbrot's first passes :326-337) is the same
// for every sample of the cell, PROVEN cell by cell; made on the CPU in seconds, embedded in the library, looked up by
// draw_wide_kernel's HEAD stage instead of computing those ~36 fp64 instructions for each of the 91 % of samples that
// end there.
//
// CELLS.  Level L: closed squares of side h = 2^-L of the c-plane, columns over re in [-2, 2], rows over |im| in [0, 2]
// (the Mandelbrot iteration commutes with conjugation bit for bit: i -> -i flips the sign of every odd intermediate and
// leaves every rounding alone; the Burning Ship's does not, its map has rows over im in [-2, 2]).  A sample is
// c = (v + 1 - 2^52) 2^-51 per coordinate with v = x1 | (x2 >> 11) << 32 (rocrand_uniform.h:102-109), so the top L + 2
// bits of x2 say which column it is in: column k holds exactly the samples with re in (-2 + k h, -2 + (k + 1) h], a
// subset of the closed cell -- the lookup is integer arithmetic on the generator's outputs, nothing is rounded.
//
// CLASSES (one byte per cell; the kernel's HEAD stage accumulates bits 0-3 and pushes a sample to Q0 iff bit 4 is set):
//   0x08        REJECTED    InMainCardioid(c) || InOrder2Bulb(c) is true for every sample of the cell (cudabrot.cu:398)
//   0x01..0x04  ESCAPED k   not rejected, and IterateMandelbrot's test |z|^2 > 4 (:336) fires first at its k-th pass,
//                           k = 1..4: the reference executes k iterations and drops the sample (k < min_iterations)
//   0x14        SURVIVOR    not rejected and the first four tests do not fire (the low bits: its 4 iterations)
//   0x30        UNDECIDED   none of the above could be proven: the kernel computes the sample exactly
//
// THE PROOF for a closed square Q of centre c0 and half side q, rc = q sqrt 2 (inflated): with z_k the orbit of c0 as
// this program computes it (the canonical sequence of device_math.h) and w_k the orbit the reference computes for any
// sample c of Q,
//   |w_k - z_k| <= rho_k,   rho_0 = rc,   rho_{k+1} = rho_k (2 |z_k| + rho_k) + rc + 2 eps
// (|w^2 - z^2| = |w - z| |w + z|; |c - c0| <= rc; eps = 2^-41 bounds the rounding of one pass on either side while
// |w|, |z| <= 8: at most 8 roundings of quantities below 80, 8 * 80 * 2^-53 < 2^-43 -- and the Burning Ship's |.| moves
// two points no further apart).  The reference's test value is fl(|w_k|^2) = |w_k|^2 (1 + d), |d| < 2^-51, so
//   the test at pass k fires for EVERY sample of Q   if  |z_k| - rho_k > 2 + mu,
//   and for NONE                                     if  |z_k| + rho_k < 2 - mu,       mu = 2^-30
// (passes are only chained while no test has fired, so every |w_j| met on the way is below 2 and the pass that fires has
// |w_k| <= 2^2 + 2.83: the bound on eps holds).  Cardioid and bulb: P(c) = q (q + x) - y^2 / 4 with x = re - 1/4,
// q = x^2 + y^2, and B(c) = (re + 1)^2 + y^2 - 1/16 are evaluated at c0; over Q they move by at most G rc with G a bound
// on |dP/dx| + |dP/dy| over Q (resp. on |grad B|), and the reference's roundings (quantities below 128, a dozen
// operations) move its compare by less than 2^-42 -- so "P < -G rc - mu" proves the cardioid test TRUE on all of Q and
// "P > G rc + mu" proves it FALSE; likewise B.  Every bound is a sum or product of positive doubles inflated by
// 1 + 2^-40.  A square that cannot be decided as a whole is decided if its four quarters (closed: they cover it) are
// decided alike, DEPTH levels down.
//
//   gcc -O2 -fopenmp -ffp-contract=off -o head_map tools/head_map.c -lm
//   ./head_map make LEVEL out.bin [DEPTH] [ship]   16 bytes of header ("CBHM", LEVEL, columns, rows: u32), then one byte
//                                                  per cell, COLUMN by column (index = column * rows + row)
//   ./head_map check LEVEL map.bin N [ship]        N uniform samples of [-2,2]^2, each computed with the reference's
//                                                  arithmetic and compared with its cell's class (must all agree)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
static int thread_index(void) { return omp_get_thread_num(); }
#else
static int thread_index(void) { return 0; }
#endif

typedef double bound_t;
static const bound_t kInflate = 1.0 + 0x1p-40;
static const bound_t kEps2 = 0x1p-40;  // 2 eps
static const bound_t kMu = 0x1p-30;

enum { REJECTED = 0x08, SURVIVOR = 0x14, UNDECIDED = 0x30 };

static int g_ship = 0;

static inline void step(double *r, double *i, double cr, double ci) {  // device_math.h: mandel_step / mandel_step_ship
  const double ii = *i * *i, t = fma(*r, *r, -ii);
  const double ni = g_ship ? fma(fabs(*r) + fabs(*r), fabs(*i), ci) : fma(*r + *r, *i, ci), nr = cr + t;
  *r = nr;
  *i = ni;
}
static inline bound_t modulus_up(double r, double i) { return sqrt((bound_t) r * r + (bound_t) i * i) * kInflate; }
static inline bound_t modulus_down(double r, double i) { return sqrt((bound_t) r * r + (bound_t) i * i) / kInflate; }

// the class of the closed square of centre (x0, y0) and half side q, as one disc
static int classify_disc(double x0, double y0, double q) {
  const bound_t rc = (bound_t) q * 1.4142135623730951 * kInflate;
  if (!g_ship) {
    // cudabrot.cu:284-298 at the centre, and how far the two expressions can move over the square
    const double x = x0 - 0.25, qq = x * x + y0 * y0;
    const double P = qq * (qq + x) - 0.25 * y0 * y0;
    const bound_t X = fabs(x) + q, Y = fabs(y0) + q, Q = (X * X + Y * Y) * kInflate;
    const bound_t G = ((2.0 * X * (Q + X) + Q * (2.0 * X + 1.0)) + (2.0 * Y * (2.0 * Q + X) + 0.5 * Y)) * kInflate;
    const bound_t dP = (G * rc + kMu) * kInflate;
    const double t = x0 + 1.0, B = t * t + y0 * y0 - 0.0625;
    const bound_t dB = ((2.0 * (fabs(t) + q) + 2.0 * Y) * rc + kMu) * kInflate;
    if (P < -dP || B < -dB) return REJECTED;        // one of the two tests is true on all of the square
    if (!(P > dP && B > dB)) return UNDECIDED;      // else both must be false on all of it
  }
  double r = x0, i = y0;
  bound_t rho = rc;
  for (int k = 1; k <= 4; ++k) {
    const bound_t mag = modulus_up(r, i);
    rho = (rho * (2.0 * mag + rho) + rc + kEps2) * kInflate;
    step(&r, &i, x0, y0);
    if (modulus_down(r, i) - rho > 2.0 + kMu) return k;       // fires for every sample
    if (!(modulus_up(r, i) + rho < 2.0 - kMu)) return UNDECIDED;  // else it must fire for none
  }
  return SURVIVOR;
}

static int classify_square(double x0, double y0, double q, int depth) {
  const int c = classify_disc(x0, y0, q);
  if (c != UNDECIDED || depth == 0) return c;
  const double h = 0.5 * q;  // exact: dyadic
  int first = -1;
  for (int k = 0; k < 4; ++k) {
    const int ck = classify_square(x0 + ((k & 1) ? h : -h), y0 + ((k & 2) ? h : -h), h, depth - 1);
    if (ck == UNDECIDED) return UNDECIDED;
    if (first < 0) first = ck;
    if (ck != first) return UNDECIDED;
  }
  return first;
}

// what the reference computes for ONE sample (cudabrot.cu:398, 326-337): the class it belongs to
static int class_of_sample(double cr, double ci) {
  if (!g_ship) {
    const double ii = ci * ci;
    double q = cr - 0.25;
    q = fma(q, q, ii);
    if (q * (q + (cr - 0.25)) < ii * 0.25) return REJECTED;
    const double t = cr + 1.0;
    if (fma(t, t, ii) < 0.0625) return REJECTED;
  }
  double r = cr, i = ci;
  for (int k = 1; k <= 4; ++k) {
    step(&r, &i, cr, ci);
    if (fma(i, i, r * r) > 4.0) return k;
  }
  return SURVIVOR;
}

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: head_map make LEVEL out.bin [DEPTH] [ship] | check LEVEL map.bin N [ship]\n"); return 2; }
  const int level = atoi(argv[2]);
  g_ship = (argc > 5 && !strcmp(argv[5], "ship")) ? 1 : 0;
  if (level < 3 || level > 11) { fprintf(stderr, "level 3..11\n"); return 2; }
  const long cols = 4l << level, rows = (g_ship ? 4l : 2l) << level, cells = cols * rows;
  const double h = ldexp(1.0, -level);
  uint8_t *map = malloc((size_t) cells);
  if (!strcmp(argv[1], "make")) {
    const int depth = argc > 4 ? atoi(argv[4]) : 6;
    long n[64] = {0};
#pragma omp parallel for schedule(dynamic, 8)
    for (long x = 0; x < cols; ++x) {
      for (long y = 0; y < rows; ++y) {
        const double x0 = -2.0 + (x + 0.5) * h, y0 = (g_ship ? -2.0 : 0.0) + (y + 0.5) * h;  // exact: dyadic
        map[x * rows + y] = (uint8_t) classify_square(x0, y0, 0.5 * h, depth);
      }
    }
    for (long k = 0; k < cells; ++k) n[map[k]]++;
    FILE *f = fopen(argv[3], "wb");
    const uint32_t header[4] = {g_ship ? 0x53484243u /* "CBHS" */ : 0x4d484243u /* "CBHM" */, (uint32_t) level, (uint32_t) cols, (uint32_t) rows};
    if (!f || fwrite(header, 1, 16, f) != 16 || fwrite(map, 1, (size_t) cells, f) != (size_t) cells) { perror(argv[3]); return 1; }
    fclose(f);
    printf("level %d%s: %ld x %ld cells, depth %d: rejected %ld, escaped at 1..4 %ld %ld %ld %ld, survivors %ld, undecided %ld (%.4f)\n",
           level, g_ship ? " (ship)" : "", cols, rows, depth, n[REJECTED], n[1], n[2], n[3], n[4], n[SURVIVOR], n[UNDECIDED],
           (double) n[UNDECIDED] / (double) cells);
    return 0;
  }
  if (!strcmp(argv[1], "check")) {
    FILE *f = fopen(argv[3], "rb");
    uint32_t header[4];
    if (!f || fread(header, 1, 16, f) != 16 || header[1] != (uint32_t) level || header[2] != (uint32_t) cols ||
        header[3] != (uint32_t) rows || fread(map, 1, (size_t) cells, f) != (size_t) cells) {
      fprintf(stderr, "%s: not a level-%d map\n", argv[3], level);
      return 1;
    }
    fclose(f);
    const long n = argc > 4 ? atol(argv[4]) : 100000000;
    long wrong = 0, undecided = 0;
#pragma omp parallel reduction(+ : wrong, undecided)
    {
      uint64_t st = 0x9e3779b97f4a7c15ull * (uint64_t) (1 + thread_index());
#pragma omp for schedule(static)
      for (long k = 0; k < n; ++k) {
        // the generator's outputs stand in for rocRAND's: x1, x2 per coordinate, mapped exactly as device_math.h does
        uint32_t o[4];
        for (int j = 0; j < 4; ++j) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; o[j] = (uint32_t) (st >> 32); }
        if ((k & 1023) == 0) { o[0] = 0xffffffffu; o[1] |= 0x1fffffu >> (k >> 10 & 7); }  // (cell edges now and then)
        const double cr = fma(fma((double) (o[1] >> 11), 4294967296.0, (double) o[0]), 0x1p-51, 0x1p-51 - 2.0);
        const double ci = fma(fma((double) (o[3] >> 11), 4294967296.0, (double) o[2]), 0x1p-51, 0x1p-51 - 2.0);
        const long col = o[1] >> (30 - level);
        long row = o[3] >> (30 - level);
        if (!g_ship) row = (o[3] >> 31) ? row - (2l << level) : (2l << level) - 1 - row;
        const int c = map[col * rows + row];
        if (c == UNDECIDED) { ++undecided; continue; }
        if (c != class_of_sample(cr, ci)) {
          if (++wrong <= 10) printf("WRONG: c = %.17g %.17g is of class %d, its cell says %d\n", cr, ci, class_of_sample(cr, ci), c);
        }
      }
    }
    printf("%ld samples: %ld in undecided cells (%.4f), %ld of the others against their cell's class\n", n, undecided,
           (double) undecided / (double) n, wrong);
    return wrong != 0;
  }
  return 2;
}
