// interior_map.c -- a map of parameter cells whose samples PROVABLY never escape under the reference's fp64 iteration
// (cudabrot.cu:331-336), made and checked on the CPU; and a model of what such a map saves the LONG stage.
//
// The never-escaping samples that pass the cardioid / period-2 test (0.83 % of all) are 16.8 of the 22.9 iterations the
// draw kernel executes per sample: each waits ~2000 iterations for a bit-exact repeat.  A sample whose
// cell is in this map needs none of them.
//
// THE CLAIM for a marked cell Q (a closed square of side s = 2^-LEVEL in the c-plane, |Im c| by symmetry): for every
// real c in Q enlarged by 2^-40, the sequence the reference computes from z_0 = c -- z_{n+1} = fl(z_n^2 + c) in fp64,
// whatever the order and fusing of its roundings -- satisfies |z_n| <= 1.99 for all n, so its escape test never fires
// and IterateMandelbrot returns max_iterations, for any max_iterations.
//
// THE PROOF, cell by cell (verify_cell).  Let c0 be the centre of Q, rc >= |c - c0| for all c in the enlarged cell,
// z_n the fp64 orbit of c0 as THIS program computes it, eps = 2^-45 a bound on |fl(z^2 + c) - (z^2 + c)| for |z| <= 2,
// |c| <= 2.6 in ANY evaluation order (at most 6 roundings of quantities below 16: 6 * 16 * 2^-53 < 2^-46).
//  (1) Shadowing.  If |w - z_k| <= rho and |z_k| + rho <= 2, then for any c in Q and w' = fl(w^2 + c):
//          |w' - z_{k+1}| <= |w^2 - z_k^2| + |c - c0| + 2 eps <= rho (2 |z_k| + rho) + rc + 2 eps =: next(rho, k),
//      and next is increasing in rho.  With rho_0 = rc (z_0 = c against c0) and rho_{k+1} = next(rho_k, k), every
//      c in Q has |w_k - z_k| <= rho_k as long as |z_j| + rho_j <= 1.99 held for all j < k.
//  (2) Trap.  Take a < b, P = b - a, d = |z_b - z_a|, R >= rho_a, rho'_0 = R, rho'_{k+1} = next(rho'_k, a + k).  If
//      |z_{a+k}| + rho'_k <= 1.99 for k <= P and rho'_P + d <= R, then any w within R of z_a is, P steps later, within
//      rho'_P of z_b, that is within R of z_a again, and in between within rho'_k of z_{a+k}: by induction over the
//      rounds every later point of the sequence lies within rho'_k of some z_{a+k}, hence below 1.99 in modulus.
//  (1) puts w_a(c) within rho_a <= R of z_a for every c in Q; (2) keeps it there.  The bounds are evaluated in
//  double -- sums, products and square roots of positive quantities only -- with every update inflated by 1 + 2^-40,
//  which dominates the 2^-53 roundings of its handful of operations (1.4142135623730951 > sqrt 2).
//
//   gcc -O2 -fopenmp -ffp-contract=off -o interior_map tools/interior_map.c -lm
//   ./interior_map make LEVEL out.bin [D]    the map: cells of side 2^-LEVEL over re in [-2, 0.5), |im| in [0, 1.25),
//                                            each proven as one disc or by its quarters, D (default 2) levels deep;
//                                            16 bytes of header ("CBIM", LEVEL, columns, rows: u32), then one bit per
//                                            cell, row by row (bit k & 7 of byte k >> 3, k = row * columns + column)
//   ./interior_map model LEVEL map.bin N     N uniform samples on [-2,2]^2: iterations per sample with the kernel's
//                                            periodicity check alone and with the map in front of it
//   ./interior_map check LEVEL map.bin N     N samples drawn INSIDE marked cells, iterated to max_iter 20000 with the
//                                            reference's arithmetic: how many escape (must be 0)
//   ./interior_map deepen LEVEL map.bin D depths.bin out_map.bin out_depths.bin SECONDS
//                                            try the unmarked cells that touch a marked one with D levels of quarters
//                                            (one more than the map was made with): an anytime run, see there
//   ./interior_map verify LEVEL map.bin D depths.bin [SAMPLE SKIP_BELOW SEED [PER_DEPTH]]
//                                            re-prove the marked cells of a kept map (all of them, noting the depth each
//                                            proof needed in depths.bin; or a stratified sample against those notes)
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RE_MIN (-2.0)
#define RE_SPAN 2.5
#define IM_SPAN 1.25
#define RING 128
typedef double bound_t;  // the bounds: positive quantities, sums, products and square roots only -- every update is
                         // inflated by 1 + 2^-40, which dominates the 2^-53 roundings of its handful of operations
static const bound_t kInflate = 1.0 + 0x1p-40;
static const bound_t kEps2 = 0x1p-44;   // 2 eps
static const bound_t kBound = 1.99;
static const bound_t kGiveUp = 0.02;    // a ball this large will not be trapped: the cell is too near a boundary

static inline void step(double *r, double *i, double cr, double ci) {  // device_math.h's canonical sequence
  const double ii = *i * *i, t = fma(*r, *r, -ii);
  const double ni = fma(*r + *r, *i, ci), nr = cr + t;
  *r = nr;
  *i = ni;
}
static inline int in_cardioid_or_bulb(double cr, double ci) {
  const double x = cr - 0.25, q = x * x + ci * ci;
  if (q * (q + x) < 0.25 * ci * ci) return 1;
  return (cr + 1.0) * (cr + 1.0) + ci * ci < 0.0625;
}
static inline bound_t next_rho(bound_t rho, bound_t mag, bound_t rc) {
  return (rho * (2.0 * mag + rho) + rc + kEps2) * kInflate;
}
static inline bound_t modulus_up(double r, double i) {
  return sqrt((bound_t) r * r + (bound_t) i * i) * kInflate;
}

// 1 if the cell of centre (c0r, c0i) and radius rc (covering the enlarged square) is proven never-escaping; -1 if its
// centre itself escapes (no part of a square around it can then be proven as a whole); 0 if the proof does not get there.
static int verify_cell(double c0r, double c0i, bound_t rc, int max_steps) {
  double zr[RING], zi[RING];
  bound_t rho_at[RING];
  double r = c0r, i = c0i;
  bound_t rho = rc * kInflate;
  int next_try = 64;
  for (int j = 0; j < max_steps; ++j) {
    const bound_t mag = modulus_up(r, i);
    if (mag + rho > kBound) {  // the ball is at the bound: a centre on its way out says so within a few steps
      for (int k = 0; k < 64; ++k) {
        if (!(fma(i, i, r * r) <= 4.0)) return -1;
        step(&r, &i, c0r, c0i);
      }
      return 0;
    }
    zr[j % RING] = r;
    zi[j % RING] = i;
    rho_at[j % RING] = rho;
    if (j == next_try) {
      next_try = j < 4096 ? 2 * j : j + 4096;
      for (int P = 1; P <= 96 && P <= j; ++P) {  // is the ball at a = j - P trapped by the round a .. j?
        const int a = j - P;
        const bound_t d = modulus_up(r - zr[a % RING], i - zi[a % RING]);  // (the difference of doubles: exact or rounded, inflated)
        const bound_t rho_a = rho_at[a % RING];
        if (d > rho_a) continue;  // not a return yet
        const bound_t R = 2.0 * rho_a + 8.0 * d;
        bound_t rp = R;
        int ok = 1;
        for (int k = 0; k < P; ++k) {
          const bound_t m = modulus_up(zr[(a + k) % RING], zi[(a + k) % RING]);
          if (m + rp > kBound) { ok = 0; break; }
          rp = next_rho(rp, m, rc);
          if (rp > 1.0) { ok = 0; break; }
        }
        if (ok && modulus_up(r, i) + rp <= kBound && (rp + d) * kInflate <= R) return 1;
      }
    }
    rho = next_rho(rho, mag, rc);
    if (rho > kGiveUp) return 0;
    step(&r, &i, c0r, c0i);
    if (!(fma(i, i, r * r) <= 4.0)) return -1;
  }
  return 0;
}

// A cell the proof does not reach as one disc may still be proven piece by piece: the square of centre (c0r, c0i) and
// half side h is marked if it is proven itself or if its four quarters are (to `depth` levels of quarters) -- the union
// of the quarters' enlarged squares covers the enlarged square.  A quarter has half the radius: the ball of (1) grows
// half as fast, and the unproven layer along a component's boundary shrinks with it.
static int verify_square(double c0r, double c0i, double h, int depth) {
  const bound_t rc = ((bound_t) h + 0x1p-40) * 1.4142135623730951 * kInflate;
  const int proven = verify_cell(c0r, c0i, rc, 20000);
  if (proven != 0 || depth == 0) return proven;  // (a centre that escapes is a point of every square around it)
  const double q = 0.5 * h;  // exact: dyadic
  for (int k = 0; k < 4; ++k) {
    if (verify_square(c0r + ((k & 1) ? q : -q), c0i + ((k & 2) ? q : -q), q, depth - 1) != 1) return 0;
  }
  return 1;
}

// verify_square, also reporting how many levels of quarters the proof went down (the smallest `depth` it succeeds with:
// the recursion subdivides only where the one-disc proof fails, so a larger budget never changes the tree above).
static int verify_square_depth(double c0r, double c0i, double h, int depth, int *used) {
  const bound_t rc = ((bound_t) h + 0x1p-40) * 1.4142135623730951 * kInflate;
  const int proven = verify_cell(c0r, c0i, rc, 20000);
  *used = 0;
  if (proven != 0 || depth == 0) return proven;
  const double q = 0.5 * h;
  int deepest = 0;
  for (int k = 0; k < 4; ++k) {
    int u = 0;
    if (verify_square_depth(c0r + ((k & 1) ? q : -q), c0i + ((k & 2) ? q : -q), q, depth - 1, &u) != 1) return 0;
    if (u > deepest) deepest = u;
  }
  *used = deepest + 1;
  return 1;
}

static long cols_of(int level) { return (long) ldexp(RE_SPAN, level); }
static long rows_of(int level) { return (long) ldexp(IM_SPAN, level); }
static inline long cell_of(double cr, double ci, int level, long cols, long rows) {  // -1: outside the map
  const double x = floor(ldexp(cr - RE_MIN, level)), y = floor(ldexp(fabs(ci), level));
  if (!(x >= 0 && x < cols && y >= 0 && y < rows)) return -1;
  return (long) y * cols + (long) x;
}

static uint64_t rng_state = 88172645463325252ull;
static inline double uniform01(void) {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (double) (rng_state >> 11) * (1.0 / 9007199254740992.0);
}
static int sched(unsigned c) {  // the kernel's schedule of saved points: chunk counts 1, 2, 3, 4, 6, 8, 12, ...
  int top = 31 - __builtin_clz(c);
  unsigned low = top >= 1 ? c & ((1u << (top - 1)) - 1u) : 0;
  return low == 0;
}
static inline uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: interior_map make|model|check|verify|deepen LEVEL file [N]\n"); return 2; }
  const int level = atoi(argv[2]);
  const long cols = cols_of(level), rows = rows_of(level), cells = cols * rows;
  const size_t bytes = (size_t) (cells + 7) / 8;
  uint8_t *map = calloc(bytes, 1);
  if (!strcmp(argv[1], "make")) {
    const double s = ldexp(1.0, -level);
    const int depth = argc > 4 ? atoi(argv[4]) : 2;  // levels of quarters a cell may be proven by
    long marked = 0, tried = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : marked, tried)
    for (long y = 0; y < rows; ++y) {
      for (long x = 0; x < cols; ++x) {
        const double c0r = RE_MIN + (x + 0.5) * s, c0i = (y + 0.5) * s;  // exact: dyadic
        if (c0r * c0r + c0i * c0i > 4.0 || in_cardioid_or_bulb(c0r, c0i)) continue;  // (never looked up / never reaches the lookup)
        ++tried;
        if (verify_square(c0r, c0i, 0.5 * s, depth) == 1) {
          const long k = y * cols + x;
#pragma omp atomic
          map[k >> 3] |= (uint8_t) (1u << (k & 7));
          ++marked;
        }
      }
    }
    map[0] &= (uint8_t) ~1u;  // cell 0 stands for "outside the map" in the kernel's lookup: never marked
    FILE *f = fopen(argv[3], "wb");
    const uint32_t header[4] = {0x4d494243u /* "CBIM" */, (uint32_t) level, (uint32_t) cols, (uint32_t) rows};
    if (!f || fwrite(header, 1, 16, f) != 16 || fwrite(map, 1, bytes, f) != bytes) { perror(argv[3]); return 1; }
    fclose(f);
    printf("level %d: %ld x %ld cells of side 2^-%d, %ld tried, %ld marked (%.4f of the plane's [-2,2]^2), %zu bytes\n", level, cols,
           rows, level, tried, marked, 2.0 * marked * s * s / 16.0, bytes);
    return 0;
  }
  FILE *f = fopen(argv[3], "rb");
  uint32_t header[4];
  if (!f || fread(header, 1, 16, f) != 16 || header[0] != 0x4d494243u || header[1] != (uint32_t) level ||
      fread(map, 1, bytes, f) != bytes) { fprintf(stderr, "%s: not a level-%d map\n", argv[3], level); return 1; }
  fclose(f);
  if (!strcmp(argv[1], "verify")) {
    // Re-prove the marked cells of a kept map, each with up to DEPTH levels of quarters, and note the depth every proof
    // needed in a sidecar: "CBID", LEVEL, marked cells, DEPTH (u32 each), then one byte per marked cell in the order of
    // their indices (0xff: not proven yet).  A sidecar that exists is continued (cells with a depth are skipped unless
    // SAMPLE is given); SAMPLE > 0: re-prove only the cells whose noted depth exceeds SKIP_BELOW plus SAMPLE others drawn
    // with SEED (the stratified check of tests/test_interior_map.py), changing nothing in the file.
    //   ./interior_map verify LEVEL map.bin DEPTH depths.bin [SAMPLE SKIP_BELOW SEED [PER_DEPTH]]
    // (PER_DEPTH: of the cells above SKIP_BELOW only so many per depth, drawn at random -- a test that fits minutes)
    if (argc < 6) { fprintf(stderr, "usage: interior_map verify LEVEL map.bin DEPTH depths.bin [SAMPLE SKIP_BELOW SEED [PER_DEPTH]]\n"); return 2; }
    const int depth = atoi(argv[4]);
    const long sample = argc > 6 ? atol(argv[6]) : 0;
    const int skip_below = argc > 7 ? atoi(argv[7]) : 3;
    if (argc > 8) rng_state ^= (uint64_t) atol(argv[8]) * 0x9e3779b97f4a7c15ull;
    const long per_depth = argc > 9 ? atol(argv[9]) : -1;  // at most so many of the cells of every depth above SKIP_BELOW (-1: all)
    const double s = ldexp(1.0, -level);
    long *marked = malloc(sizeof(long) * ((size_t) cells / 8 + 64));
    long nm = 0;
    for (long k = 0; k < cells; ++k) if ((map[k >> 3] >> (k & 7)) & 1) marked[nm++] = k;
    uint8_t *used = malloc((size_t) nm + 1);
    memset(used, 0xff, (size_t) nm + 1);
    FILE *g = fopen(argv[5], "rb");
    if (g) {
      uint32_t h4[4];
      if (fread(h4, 1, 16, g) != 16 || h4[0] != 0x44494243u || h4[1] != (uint32_t) level || h4[2] != (uint32_t) nm ||
          fread(used, 1, (size_t) nm, g) != (size_t) nm) { fprintf(stderr, "%s: not the depths of this map\n", argv[5]); return 1; }
      fclose(g);
    } else if (sample > 0) { fprintf(stderr, "%s: missing (a sample needs the noted depths)\n", argv[5]); return 1; }
    uint8_t *todo = calloc((size_t) nm + 1, 1);
    long n_todo = 0;
    if (sample > 0) {
      if (per_depth < 0) {
        for (long k = 0; k < nm; ++k) if (used[k] == 0xff || used[k] > skip_below) { todo[k] = 1; ++n_todo; }
      } else {  // a random stratum per depth: reservoir sampling, depth by depth
        for (int d = skip_below + 1; d <= 255; ++d) {
          long seen = 0, *pick = malloc(sizeof(long) * (size_t) (per_depth + 1));
          for (long k = 0; k < nm; ++k) {
            if (used[k] != (uint8_t) d) continue;
            if (seen < per_depth) pick[seen] = k;
            else { const long j = (long) (uniform01() * (double) (seen + 1)); if (j < per_depth) pick[j] = k; }
            ++seen;
          }
          for (long j = 0; j < (seen < per_depth ? seen : per_depth); ++j) if (!todo[pick[j]]) { todo[pick[j]] = 1; ++n_todo; }
          free(pick);
          if (d == 254) break;
        }
      }
      for (long k = 0; k < sample; ++k) {
        const long j = (long) (uniform01() * (double) nm);
        if (j < nm && !todo[j]) { todo[j] = 1; ++n_todo; }
      }
    } else {
      for (long k = 0; k < nm; ++k) if (used[k] == 0xff) { todo[k] = 1; ++n_todo; }
    }
    printf("%ld marked cells, %ld to prove at depth <= %d\n", nm, n_todo, depth);
    fflush(stdout);
    long failed = 0, done = 0, mismatched = 0;
    const uint32_t h4[4] = {0x44494243u /* "CBID" */, (uint32_t) level, (uint32_t) nm, (uint32_t) depth};
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : failed, mismatched)
    for (long k = 0; k < nm; ++k) {
      if (!todo[k]) continue;
      const long cell = marked[k], x = cell % cols, y = cell / cols;
      const double c0r = RE_MIN + (x + 0.5) * s, c0i = (y + 0.5) * s;
      int u = 0;
      const int ok = verify_square_depth(c0r, c0i, 0.5 * s, depth, &u) == 1;
      if (!ok) {
        if (++failed <= 20) printf("NOT PROVEN at depth %d: cell %ld (column %ld, row %ld)\n", depth, cell, x, y);
      } else if (sample > 0) {
        if (used[k] != 0xff && used[k] != (uint8_t) u) ++mismatched;
      } else {
        used[k] = (uint8_t) u;
      }
      long d;
#pragma omp atomic capture
      d = ++done;
      if (sample == 0 && d % 20000 == 0) {
#pragma omp critical
        {
          FILE *o = fopen(argv[5], "wb");  // (a cell being written right now is 0xff or its depth: either is true)
          if (o) { fwrite(h4, 1, 16, o); fwrite(used, 1, (size_t) nm, o); fclose(o); }
          printf("%ld of %ld\n", d, n_todo);
          fflush(stdout);
        }
      }
    }
    if (sample == 0) {
      FILE *o = fopen(argv[5], "wb");
      if (!o || fwrite(h4, 1, 16, o) != 16 || fwrite(used, 1, (size_t) nm, o) != (size_t) nm) { perror(argv[5]); return 1; }
      fclose(o);
    }
    long by_depth[17] = {0};
    for (long k = 0; k < nm; ++k) by_depth[used[k] == 0xff ? 16 : (used[k] > 15 ? 15 : used[k])]++;
    printf("proved %ld cells, %ld NOT proven, %ld with another depth than noted; cells by depth:", done - failed, failed, mismatched);
    for (int d = 0; d < 16; ++d) if (by_depth[d]) printf(" %d:%ld", d, by_depth[d]);
    if (by_depth[16]) printf(" unknown:%ld", by_depth[16]);
    printf("\n");
    return failed != 0 || mismatched != 0;
  }
  if (!strcmp(argv[1], "deepen")) {
    // One more level of quarters where it can pay: the UNMARKED cells that touch a marked one (the boundary layer the
    // kept budget left), each tried with DEPTH levels; a cell that is proven is marked, and its depth noted.  Every cell
    // stands on its own proof, so the run can be stopped at any time (SECONDS: no new cell is begun after that; the two
    // files are rewritten every ten minutes): what has been added by then is a valid, larger map.
    //   ./interior_map deepen LEVEL map.bin DEPTH depths.bin out_map.bin out_depths.bin SECONDS
    if (argc < 9) { fprintf(stderr, "usage: interior_map deepen LEVEL map.bin DEPTH depths.bin out_map.bin out_depths.bin SECONDS\n"); return 2; }
    const int depth = atoi(argv[4]);
    const double budget = atof(argv[8]);
    const double s = ldexp(1.0, -level);
    long nm = 0;
    for (long k = 0; k < cells; ++k) nm += (map[k >> 3] >> (k & 7)) & 1;
    uint8_t *cell_depth = malloc((size_t) cells);  // per CELL here (0xff: not marked)
    memset(cell_depth, 0xff, (size_t) cells);
    {
      FILE *g = fopen(argv[5], "rb");
      uint32_t h4[4];
      uint8_t *used = malloc((size_t) nm + 1);
      if (!g || fread(h4, 1, 16, g) != 16 || h4[0] != 0x44494243u || h4[1] != (uint32_t) level || h4[2] != (uint32_t) nm ||
          fread(used, 1, (size_t) nm, g) != (size_t) nm) { fprintf(stderr, "%s: not the depths of this map\n", argv[5]); return 1; }
      fclose(g);
      long j = 0;
      for (long k = 0; k < cells; ++k) if ((map[k >> 3] >> (k & 7)) & 1) cell_depth[k] = used[j++];
      free(used);
    }
    // candidates: unmarked, a marked cell among the eight around it, the centre inside |c| <= 2 and outside cardioid and bulb
    long *cand = malloc(sizeof(long) * (size_t) (cells / 16 + 64));
    long nc = 0;
    for (long y = 0; y < rows; ++y) {
      for (long x = 0; x < cols; ++x) {
        const long k = y * cols + x;
        if (k == 0 || ((map[k >> 3] >> (k & 7)) & 1)) continue;
        int touches = 0;
        for (int dy = -1; dy <= 1 && !touches; ++dy) {
          for (int dx = -1; dx <= 1; ++dx) {
            const long yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= rows || xx < 0 || xx >= cols) continue;
            const long kk = yy * cols + xx;
            if ((map[kk >> 3] >> (kk & 7)) & 1) { touches = 1; break; }
          }
        }
        if (!touches) continue;
        const double c0r = RE_MIN + (x + 0.5) * s, c0i = (y + 0.5) * s;
        if (c0r * c0r + c0i * c0i > 4.0 || in_cardioid_or_bulb(c0r, c0i)) continue;
        if (nc < cells / 16) cand[nc++] = k;
      }
    }
    for (long k = nc - 1; k > 0; --k) {  // a random order: a run that is stopped has tried a uniform sample
      const long j = (long) (uniform01() * (double) (k + 1));
      const long t = cand[k]; cand[k] = cand[j]; cand[j] = t;
    }
    printf("%ld marked cells, %ld candidates at depth <= %d, %.0f s\n", nm, nc, depth, budget);
    fflush(stdout);
    const double t_begin = omp_get_wtime();
    double t_saved = t_begin;
    long added = 0, tried = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (long j = 0; j < nc; ++j) {
      if (omp_get_wtime() - t_begin > budget) continue;
      const long k = cand[j], x = k % cols, y = k / cols;
      int u = 0;
      const int ok = verify_square_depth(RE_MIN + (x + 0.5) * s, (y + 0.5) * s, 0.5 * s, depth, &u) == 1;
#pragma omp critical
      {
        ++tried;
        if (ok) {
          map[k >> 3] |= (uint8_t) (1u << (k & 7));
          cell_depth[k] = (uint8_t) u;
          ++added;
        }
        const double now = omp_get_wtime();
        if (now - t_saved > 600.0 || tried == nc) {
          t_saved = now;
          const uint32_t header[4] = {0x4d494243u, (uint32_t) level, (uint32_t) cols, (uint32_t) rows};
          const uint32_t h4[4] = {0x44494243u, (uint32_t) level, (uint32_t) (nm + added), (uint32_t) depth};
          FILE *f = fopen(argv[6], "wb"), *g = fopen(argv[7], "wb");
          if (f && g) {
            fwrite(header, 1, 16, f);
            fwrite(map, 1, bytes, f);
            fwrite(h4, 1, 16, g);
            for (long c = 0; c < cells; ++c) if (cell_depth[c] != 0xff) fputc(cell_depth[c], g);
          }
          if (f) fclose(f);
          if (g) fclose(g);
          printf("%.0f s: %ld of %ld tried, %ld added\n", now - t_begin, tried, nc, added);
          fflush(stdout);
        }
      }
    }
    {
      const uint32_t header[4] = {0x4d494243u, (uint32_t) level, (uint32_t) cols, (uint32_t) rows};
      const uint32_t h4[4] = {0x44494243u, (uint32_t) level, (uint32_t) (nm + added), (uint32_t) depth};
      FILE *f = fopen(argv[6], "wb"), *g = fopen(argv[7], "wb");
      if (!f || !g) { perror("output"); return 1; }
      fwrite(header, 1, 16, f);
      fwrite(map, 1, bytes, f);
      fwrite(h4, 1, 16, g);
      for (long c = 0; c < cells; ++c) if (cell_depth[c] != 0xff) fputc(cell_depth[c], g);
      fclose(f);
      fclose(g);
    }
    printf("tried %ld of %ld candidates, added %ld cells: %ld marked\n", tried, nc, added, nm + added);
    return 0;
  }
  const long n = argc > 4 ? atol(argv[4]) : 1000000;
  const int max_iter = 20000, start = 20, chunk = 60;
  if (!strcmp(argv[1], "model")) {
    double with_check = 0, with_map = 0;
    long never = 0, covered = 0;
    for (long k = 0; k < n; ++k) {
      const double cr = uniform01() * 4.0 - 2.0, ci = uniform01() * 4.0 - 2.0;
      if (in_cardioid_or_bulb(cr, ci)) continue;
      double r = cr, i = ci, sr = 0, si = 0;
      int it = 0, escaped = 0;
      unsigned c = 0;
      while (it < max_iter) {
        step(&r, &i, cr, ci);
        ++it;
        if (fma(i, i, r * r) > 4.0) { escaped = 1; break; }
        if (it == start) { sr = r; si = i; c = 0; }
        if (it > start && (it - start) % chunk == 0) {
          ++c;
          if (bits(r) == bits(sr) && bits(i) == bits(si)) break;
          if (sched(c)) { sr = r; si = i; }
        }
      }
      with_check += it;
      const long cell = cell_of(cr, ci, level, cols, rows);
      const int hit = cell >= 0 && ((map[cell >> 3] >> (cell & 7)) & 1);
      if (hit && escaped) { printf("DISAGREEMENT: c = %.17g %.17g escapes at %d\n", cr, ci, it); return 1; }
      with_map += hit ? 4 : it;  // (the lookup sits behind HEAD's four iterations)
      if (!escaped) { ++never; covered += hit; }
    }
    printf("%ld samples: never-escaping beyond cardioid and disc %ld (%.4f), in marked cells %ld (%.3f of them)\n", n, never,
           (double) never / n, covered, never ? (double) covered / never : 0.0);
    printf("iterations per sample: periodicity check alone %.2f, with the map %.2f\n", with_check / n, with_map / n);
    return 0;
  }
  if (!strcmp(argv[1], "check")) {
    // samples inside marked cells (rejection from the cells' bounding box), full iteration with the reference's arithmetic
    long done = 0, escaped_total = 0;
    const double s = ldexp(1.0, -level);
    long *marked = malloc(sizeof(long) * (size_t) cells / 8 + 64);
    long nm = 0;
    for (long k = 0; k < cells; ++k) if ((map[k >> 3] >> (k & 7)) & 1) marked[nm++] = k;
    if (!nm) { printf("empty map\n"); return 1; }
#pragma omp parallel reduction(+ : done, escaped_total)
    {
      uint64_t st = 0x9e3779b97f4a7c15ull ^ (uint64_t) (1 + (unsigned) rand()) * 0xd1342543de82ef95ull;
      // sixteen samples side by side (the compiler's vector lanes), the escape test behind every 50 steps: for |c| <= 2
      // a point beyond 2 is never followed by one within (and an overflow ends as inf or nan, which the test rejects)
      enum { W = 16 };
#pragma omp for schedule(dynamic, 64)
      for (long k = 0; k < n / W; ++k) {
        double cr[W], ci[W], r[W], i[W];
        for (int l = 0; l < W; ++l) {
          st ^= st << 13; st ^= st >> 7; st ^= st << 17;
          const long cell = marked[(st >> 11) % (uint64_t) nm];
          st ^= st << 13; st ^= st >> 7; st ^= st << 17;
          const double u = (double) (st >> 11) * (1.0 / 9007199254740992.0);
          st ^= st << 13; st ^= st >> 7; st ^= st << 17;
          const double v = (double) (st >> 11) * (1.0 / 9007199254740992.0);
          cr[l] = RE_MIN + ((double) (cell % cols) + u) * s;
          ci[l] = ((double) (cell / cols) + v) * s;
          if (st & 1) ci[l] = -ci[l];
          r[l] = cr[l];
          i[l] = ci[l];
        }
        int esc = 0;
        for (int it = 0; it < max_iter && !esc; it += 50) {
          for (int q = 0; q < 50; ++q) {
#pragma omp simd
            for (int l = 0; l < W; ++l) {
              const double ii = i[l] * i[l], t = fma(r[l], r[l], -ii);
              const double ni = fma(r[l] + r[l], i[l], ci[l]), nr = cr[l] + t;
              r[l] = nr;
              i[l] = ni;
            }
          }
          for (int l = 0; l < W; ++l) {
            if (!(fma(i[l], i[l], r[l] * r[l]) <= 4.0)) {
              esc = 1;
              printf("ESCAPED: c = %.17g %.17g within %d\n", cr[l], ci[l], it + 50);
            }
          }
        }
        done += W;
        escaped_total += esc;
      }
    }
    printf("%ld samples inside %ld marked cells iterated to %d: %ld escaped\n", done, nm, max_iter, escaped_total);
    return escaped_total != 0;
  }
  return 2;
}
