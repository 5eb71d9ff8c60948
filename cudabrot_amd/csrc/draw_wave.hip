// draw_wave.hip -- draw_wave_kernel, the product path of DrawBuddhabrot (cudabrot.cu:379-414).
//
// Compiled with -ffp-contract=off (device_math.h).  wave = 64 lanes everywhere.
//
// Why this shape.  At max_iter = 20000 a sample needs 168 iterations on average, but 89 % of the
// samples need < 20 and 0.8 % need all 20000: one lane per reference thread in lock-step keeps ~2 % of
// the lanes busy (that is draw_simple_kernel and the reference itself).  Histogram
// increments commute, so ANY schedule of the fixed multiset of samples {(subsequence t, sample n)}
// gives the identical histogram, provided each subsequence is consumed in order with exactly four
// draws per sample.  Each wave therefore owns 64 subsequences (lane l of wave v is reference thread
// 64 v + l, as in the reference) and runs four stages over them, decoupled by three wave-private
// queues in LDS (no barriers: a wave's LDS operations are in order):
//
//   HEAD    all 64 lanes draw a starting point (4 XORWOW outputs, registers only), apply the
//           cardioid / bulb test and run the first head_steps (4) iterations under the EXEC mask.
//           Survivors (~20 %) are ballot-compacted into Q0 as c.
//   MID     64 survivors at a time run the next mid_steps (12..43) iterations in lock-step, dense:
//           this is where the many orbits that leave after a few dozen iterations leave.  mid_steps
//           is chosen so that min_iter - (head_steps + mid_steps) is a multiple of kChunk.
//           Survivors (~2 % of the samples) go to Q1 as (c, z).
//   LONG    each lane holds TWO deep orbits and iterates them side by side in chunks of kChunk (30)
//           steps of hand-written asm; finished slots refill from Q1 at chunk boundaries, so the lanes
//           stay on deep orbits.  An orbit is retired when it escapes, reaches max_iter, or is found
//           to be exactly periodic (then it can never escape).  Escaped orbits whose chunk lies at or
//           above min_iter go to Q2 as c.  No chunk straddles min_iter (the MID alignment), so
//           "which chunk" decides accept / too-fast exactly although the escape index inside the
//           chunk is not known; a last, shorter chunk before max_iter runs in a counted loop.
//   REPLAY  each lane pops one accepted starting point from Q2, re-iterates it from z0 = c with the
//           same step and records every visited point (asm burst into the pixel stream, or direct
//           device-scope u64 atomics without a workspace).  Lanes refill from Q2 as they finish; a
//           replay in flight is suspended (state stays in registers) while too few lanes are busy.
//
// Every coordinate in this kernel (c and z in registers, queues and the carry buffer) is DOUBLED:
// C = 2c, Z = 2z.  Scaling by two is exact, every rounded intermediate of the canonical sequence has
// an exact image, and the doubling r+r of the cross term disappears (device_math.h, mandel_step2):
// six fp64 instructions per step instead of seven, the same results.
//
// All step loops are hand-written gfx950 assembly: per iteration 6 fp64 VALU instructions + one
// compare and 2-3 scalar instructions for the exact lane-step count.  The compiler's own lowering of
// the same loops spent ~35 scalar instructions per step on mask bookkeeping, and the one scalar
// unit of a CU serves all four SIMDs.
#include <stdlib.h>

#include "draw_common.h"

// This file is compiled twice: as is, and with -DCB_BURNING_SHIP for the reference's
// RENDER_BURNING_SHIP variant (cudabrot.cu:15-17): real and imag are replaced by their magnitudes
// before each step (:327-330, :353-356), which only the cross term 2*real*imag notices -- a pair of
// |.| operand modifiers on one instruction of the step, free of charge -- and the cardioid / bulb
// shortcut is skipped (:397-399).  The second build exports launch_draw_wave_ship.
#ifdef CB_BURNING_SHIP
#define CB_AL "|"
#define CB_AR "|"
#define CB_LAUNCH_NAME launch_draw_wave_ship
#else
#define CB_AL ""
#define CB_AR ""
#define CB_LAUNCH_NAME launch_draw_wave
#endif

namespace cb {

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kOrbitsPerLane = 2;      // deep orbits a lane iterates side by side in the LONG stage
constexpr int kQ0Cap = 128;            // HEAD survivors: c            (2 KiB per wave)
constexpr int kQ1Cap = 96;             // MID survivors: (c, z)        (3 KiB per wave)
constexpr int kQ2Cap = 192;            // accepted starting points: c  (3 KiB per wave)
// Iterations of the HEAD stage (every lane, one sample each), 2..4.  After the shortcut test and k iterations
// 30.5 / 17.3 / 11.9 / 8.9 % of the samples are still iterating (k = 1..4): HEAD's later steps run with most
// lanes switched off, MID's re-derive them for the survivors only.
#ifndef CB_HEAD_STEPS
#define CB_HEAD_STEPS 4
#endif
#if CB_HEAD_STEPS == 4
#define CB_HEAD_MORE_STEPS CB_STEP_LIT CB_STEP_LIT CB_STEP_LIT
#define CB_MID_REDERIVE_MORE CB_STEP_NOTEST CB_STEP_NOTEST CB_STEP_NOTEST
#elif CB_HEAD_STEPS == 3
#define CB_HEAD_MORE_STEPS CB_STEP_LIT CB_STEP_LIT
#define CB_MID_REDERIVE_MORE CB_STEP_NOTEST CB_STEP_NOTEST
#elif CB_HEAD_STEPS == 2
#define CB_HEAD_MORE_STEPS CB_STEP_LIT
#define CB_MID_REDERIVE_MORE CB_STEP_NOTEST
#else
#error "CB_HEAD_STEPS: 2, 3 or 4"
#endif
constexpr int kHeadSteps = CB_HEAD_STEPS;
// -DCB_DBG_REPLAY: the timed kernel's wave dump counts REPLAY burst steps and the lanes active in them
// (tools/wave_dump_stats.py prints them as "chunks" and "slots") instead of LONG chunks and orbit slots.
#ifdef CB_DBG_REPLAY
constexpr bool kDbgReplay = true;
#else
constexpr bool kDbgReplay = false;
#endif
#ifndef CB_Q1_LOW
#define CB_Q1_LOW 32
#endif
#ifndef CB_Q1_EXIT
#define CB_Q1_EXIT 8
#endif
#ifndef CB_REPLAY_MIN
#define CB_REPLAY_MIN 56
#endif
#ifndef CB_REPLAY_BURST
#define CB_REPLAY_BURST 32
#endif
#ifndef CB_PRIO_CHUNKS
#define CB_PRIO_CHUNKS 32
#endif
constexpr int kQ1Low = CB_Q1_LOW;      // run MID while fewer deep orbits than this are queued
constexpr int kQ1Exit = CB_Q1_EXIT;    // LONG hands over to HEAD / MID below this many
constexpr int kReplayMin = CB_REPLAY_MIN;  // suspend REPLAY below this many busy lanes (unless draining)
constexpr uint32_t kReplayBurst = CB_REPLAY_BURST;  // replay steps per asm burst
#ifndef CB_BRENT_BITS
#define CB_BRENT_BITS 2
#endif
constexpr uint32_t kBrentBits = CB_BRENT_BITS;     // periodicity check: re-save when the chunk count has no bits below its top 2
constexpr uint32_t kPrioChunks = CB_PRIO_CHUNKS;  // LONG chunks per priority level in the rotation (power of two)

// The ring capacities are exact worst cases, not estimates: a stage runs only while its output ring can take
// everything the stage may push (CB_STATUS_QUEUE_OVERFLOW guards the reasoning, these guard the constants).
//   Q0: HEAD runs while q0_count < 64 and pushes at most 64                      -> 63 + 64
//   Q1: MID runs while q1_count < kQ1Low and pushes at most 64                   -> kQ1Low - 1 + 64
//   Q2: LONG runs while q2_count + replaying < 64 and one chunk can retire every orbit slot of the wave
//       (64 lanes x kOrbitsPerLane)                                              -> 63 + 64 * kOrbitsPerLane
static_assert(kQ0Cap >= 63 + 64, "Q0 must hold a full HEAD pass on top of 63 queued survivors");
static_assert(kQ1Cap >= kQ1Low - 1 + 64, "Q1 must hold a full MID pass on top of kQ1Low - 1 queued orbits");
static_assert(kQ2Cap >= 63 + 64 * kOrbitsPerLane, "Q2 must hold every orbit slot of a chunk on top of 63 queued points");

struct WaveQueues {
  double q0_cr[kQ0Cap], q0_ci[kQ0Cap];
  double q1_cr[kQ1Cap], q1_ci[kQ1Cap], q1_r[kQ1Cap], q1_i[kQ1Cap];
  double q2_cr[kQ2Cap], q2_ci[kQ2Cap];
  // iterations the orbit had left when the LONG chunk in which it escaped began (so its escape index lies in
  // [max_iter - q2_lrem, + kChunk)); 0: not known (accepted in another stage, or in its exact tail chunk)
  uint32_t q2_lrem[kQ2Cap];
};

struct Orbit {
  double cr, ci, r, i;
};

// This lane's bit of a wave-uniform mask, as a predicate: the mask itself becomes the condition
// register (s_and_saveexec), no vector instruction.
__device__ __forceinline__ bool lane_in(unsigned long long mask) {
  return __builtin_amdgcn_inverse_ballot_w64(mask);
}
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t) (v >> 32));
  return ((unsigned long long) hi << 32) | lo;
}

// ---- one orbit per lane under EXEC (HEAD, MID, the last short chunk of LONG) -----------------------
//
// One z <- z^2 + c step on the lanes in EXEC, on DOUBLED coordinates, in the order of
// device_math.h's mandel_step2:
//   a = i*i; a = fma(r,r,-a); i = fma(r,i,ci); r = fma(a,0.5,cr); a = r*r; a = fma(i,i,a)
// then EXEC &= !(16.0 < a) (v_cmpx: a lane leaves at its escape, cudabrot.cu:336), after adding the
// number of lanes that execute the step to the scalar counter.  k16 is 16.0 in a scalar pair.
#define CB_STEP                                       \
  "s_bcnt1_i32_b64 %[tmp], exec\n\t"                  \
  "v_mul_f64 %[a], %[i], %[i]\n\t"                    \
  "s_add_u32 %[cnt], %[cnt], %[tmp]\n\t"              \
  "v_fma_f64 %[a], %[r], %[r], -%[a]\n\t"             \
  "v_fma_f64 %[i], " CB_AL "%[r]" CB_AR ", " CB_AL "%[i]" CB_AR ", %[ci]\n\t"             \
  "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"              \
  "v_mul_f64 %[a], %[r], %[r]\n\t"                    \
  "v_fma_f64 %[a], %[i], %[i], %[a]\n\t"              \
  "v_cmpx_nlt_f64_e32 vcc, %[k16], %[a]\n\t"

// n steps (wave-uniform run-time count) on the lanes of `mask`, leaving early once every lane has
// escaped.  Returns the lanes that escaped; r, i of the others advance by n iterations; lane_steps
// receives the executed lane-steps (a lane that escapes at its j-th step counts j).
__device__ __forceinline__ unsigned long long iterate_steps(unsigned long long mask, uint32_t n,
                                                            Orbit &o, uint32_t &lane_steps) {
  unsigned long long save, escaped;
  uint32_t cnt, tmp, ctr;
  double a;
  const double k16 = 16.0;
  asm volatile(
      "s_mov_b64 %[save], exec\n\t"
      "s_mov_b32 %[cnt], 0\n\t"
      "s_mov_b64 exec, %[mask]\n\t"
      "s_mov_b32 %[ctr], %[n]\n\t"
      "s_cmp_eq_u32 %[n], 0\n\t"
      "s_cbranch_scc1 2f\n\t"
      "1:\n\t"
      CB_STEP
      "s_cbranch_execz 2f\n\t"
      "s_sub_u32 %[ctr], %[ctr], 1\n\t"
      "s_cmp_lg_u32 %[ctr], 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "2:\n\t"
      "s_andn2_b64 %[esc], %[mask], exec\n\t"
      "s_mov_b64 exec, %[save]\n\t"
      "s_nop 4\n\t"
      : [r] "+v"(o.r), [i] "+v"(o.i), [a] "=&v"(a), [save] "=&s"(save),
        [esc] "=&s"(escaped), [cnt] "=&s"(cnt), [tmp] "=&s"(tmp), [ctr] "=&s"(ctr)
      : [mask] "s"(mask), [n] "s"(n), [cr] "v"(o.cr), [ci] "v"(o.ci), [k16] "s"(k16)
      : "vcc", "scc");
  lane_steps = cnt;
  return escaped;
}

// Runs iterations [k0, k0 + n) of the orbits in `alive` (IterateMandelbrot, cudabrot.cu:326-337) and
// sorts the escapes by the accept filter of cudabrot.cu:407-408: an escape at index k < min_iter is
// too fast, one at k >= min_iter is accepted (k < max_iter holds by construction).  Removes the
// escaped lanes from `alive`; returns the accepted ones.
__device__ __forceinline__ unsigned long long iterate_window(unsigned long long &alive, int k0, int n,
                                                             int min_iter, Orbit &o,
                                                             unsigned long long &n_iterate,
                                                             unsigned long long &n_too_fast) {
  int n_fast = min_iter - k0;  // leading steps whose escapes are too fast
  n_fast = n_fast < 0 ? 0 : (n_fast > n ? n : n_fast);
  uint32_t steps = 0;
  if (alive != 0ull && n_fast != 0) {
    const unsigned long long esc = iterate_steps(alive, (uint32_t) n_fast, o, steps);
    n_iterate += steps;
    n_too_fast += (unsigned long long) __popcll(esc);
    alive &= ~esc;
  }
  unsigned long long accepted = 0ull;
  if (alive != 0ull && n - n_fast != 0) {
    accepted = iterate_steps(alive, (uint32_t) (n - n_fast), o, steps);
    n_iterate += steps;
    alive &= ~accepted;
  }
  return accepted;
}

// ---- HEAD in one piece (head_steps == kHeadSteps <= min_iter) -----------------------------------------------
//
// Every sample passes through HEAD, so its instruction count is a third of the kernel's.  One asm
// statement (head_loop) runs HEAD passes in a row -- until Q0 holds a MID pass, the input ends or the
// progress board is due -- so that between two passes nothing but the statement's own scalar
// bookkeeping is issued (as separate statements per pass the compiler's glue reloaded spilled scalars
// with nine v_readlane / v_writelane per pass).  A pass is
//   the draw    four XORWOW outputs (rocrand_xorwow.h:165-177) and the two starting coordinates
//               (device_math.h, sample_coordinate2), 8 + 5 VALU instructions per output pair half.
//               The generator's five words rotate by one place per output; instead of moving
//               registers the text is instantiated for the five rotations (ROT) and the kernel
//               keeps the current rotation in a scalar: logical word j lives in field (j + rot) % 5.
//   the test    cardioid / bulb test (in_main_cardioid2, in_order2_bulb2) and the first kHeadSteps
//               iterations under EXEC; the rounded I*I of the tests is the first product of step 1.
template <int K>
__device__ __forceinline__ uint32_t &xorwow_word(Xorwow &s) {
  static_assert(K >= 0 && K < 5, "five words");
  if constexpr (K == 0) return s.x0;
  if constexpr (K == 1) return s.x1;
  if constexpr (K == 2) return s.x2;
  if constexpr (K == 3) return s.x3;
  return s.x4;
}

// One output: xk is the oldest word (x0 of rocrand's step), xp the newest (x4); the new word
// replaces xk (which also serves as a temporary once t is formed); out = d + k * 362437 + new word
// (the multiple of the Weyl increment goes through one scratch scalar).
#define CB_XW_DRAW(xk, xp, out, ck)                      \
  "v_lshrrev_b32 %[t], 2, " xk "\n\t"                    \
  "v_lshlrev_b32 %[u], 4, " xp "\n\t"                    \
  "s_mov_b32 %[sc], " ck "\n\t"                          \
  "v_xor_b32 %[t], %[t], " xk "\n\t"                     \
  "v_lshlrev_b32 " xk ", 1, %[t]\n\t"                    \
  "v_bitop3_b32 %[u], %[u], " xp ", %[t] bitop3:0x96\n\t" \
  "v_xor_b32 " xk ", %[u], " xk "\n\t"                   \
  "v_add3_u32 " out ", " CB_V_D ", " xk ", %[sc]\n\t"
// C = fma(fma(hi, 2^32, lo), 2^-50, 2^-50 - 4) with lo = o1, hi = o2 >> 11 (0x41f00000: the high word
// of 2^32 as the literal of a VOP2 fmac)
#define CB_XW_COORD(c)                                   \
  "v_cvt_f64_u32 " c ", %[o1]\n\t"                       \
  "v_lshrrev_b32 %[o2], 11, %[o2]\n\t"                   \
  "v_cvt_f64_u32 %[f], %[o2]\n\t"                        \
  "v_fmac_f64_e32 " c ", 0x41f00000, %[f]\n\t"           \
  "v_fma_f64 " c ", " c ", %[k2m50], " CB_V_K "\n\t"
// Four outputs and both coordinates; X0..X4 = the registers of the logical words x0..x4.
#define CB_HEAD_DRAW(X0, X1, X2, X3, X4)                 \
  CB_XW_DRAW(X0, X4, "%[o1]", "0x587c5")                 \
  CB_XW_DRAW(X1, X0, "%[o2]", "0xb0f8a")                 \
  CB_XW_COORD("%[cr]")                                   \
  CB_XW_DRAW(X2, X1, "%[o1]", "0x10974f")                \
  CB_XW_DRAW(X3, X2, "%[o2]", "0x161f14")                \
  "v_add_u32 " CB_V_D ", %[sc], " CB_V_D "\n\t"        \
  CB_XW_COORD("%[ci]")
static_assert(362437u == 0x587c5u && 2u * 362437u == 0xb0f8au && 3u * 362437u == 0x10974fu &&
                  4u * 362437u == 0x161f14u,
              "multiples of the Weyl increment (rocrand_xorwow.h:174)");

// rot: logical word j lives in field (j + rot) % 5 of the generator (operands x0..x4; xd: the Weyl value;
// kk: the constant 2^-50 - 4 in a vector pair).  The six words are operands of head_loop, which runs several
// passes per statement, so the compiler knows they are live and any copies it makes are per statement, not per
// pass.  (They used to sit in v104..v109 behind a register budget -- amdgpu_num_vgpr(104) -- that the
// compiler turned out to be free to exceed: with a little more pressure it used those registers itself.)
// One statement holds the text for the five rotations behind scalar branches; it also steps rot.
#define CB_V_X0 "%[x0]"
#define CB_V_X1 "%[x1]"
#define CB_V_X2 "%[x2]"
#define CB_V_X3 "%[x3]"
#define CB_V_X4 "%[x4]"
#define CB_V_D "%[xd]"
#define CB_V_K "%[kk]"
// The draw of one pass: the text of the five rotations behind scalar branches; it also steps rot
// (four outputs later the words sit four places on: rot <- (rot + 4) % 5).
#define CB_HEAD_DRAW_ANY_ROT                                       \
  "s_cmp_lt_u32 %[rot], 2\n\t"                                     \
  "s_cbranch_scc1 11f\n\t"                                         \
  "s_cmp_eq_u32 %[rot], 2\n\t"                                     \
  "s_cbranch_scc1 12f\n\t"                                         \
  "s_cmp_eq_u32 %[rot], 3\n\t"                                     \
  "s_cbranch_scc1 13f\n\t"                                         \
  CB_HEAD_DRAW(CB_V_X4, CB_V_X0, CB_V_X1, CB_V_X2, CB_V_X3) /* rot 4 */ \
  "s_branch 19f\n\t"                                               \
  "13:\n\t"                                                        \
  CB_HEAD_DRAW(CB_V_X3, CB_V_X4, CB_V_X0, CB_V_X1, CB_V_X2)        \
  "s_branch 19f\n\t"                                               \
  "12:\n\t"                                                        \
  CB_HEAD_DRAW(CB_V_X2, CB_V_X3, CB_V_X4, CB_V_X0, CB_V_X1)        \
  "s_branch 19f\n\t"                                               \
  "11:\n\t"                                                        \
  "s_cmp_eq_u32 %[rot], 0\n\t"                                     \
  "s_cbranch_scc1 10f\n\t"                                         \
  CB_HEAD_DRAW(CB_V_X1, CB_V_X2, CB_V_X3, CB_V_X4, CB_V_X0)        \
  "s_branch 19f\n\t"                                               \
  "10:\n\t"                                                        \
  CB_HEAD_DRAW(CB_V_X0, CB_V_X1, CB_V_X2, CB_V_X3, CB_V_X4)        \
  "19:\n\t"                                                        \
  "s_add_u32 %[sc], %[rot], 4\n\t"                                 \
  "s_sub_u32 %[rot], %[rot], 1\n\t"                                \
  "s_cmp_ge_u32 %[sc], 5\n\t"                                      \
  "s_cselect_b32 %[rot], %[rot], %[sc]\n\t"

// The generator words in logical order again (rot back to 0), for store_rng and the generic HEAD.
template <int ROT>
__device__ __forceinline__ Xorwow xorwow_unrotated(Xorwow &s) {
  Xorwow r;
  r.x0 = xorwow_word<(0 + ROT) % 5>(s);
  r.x1 = xorwow_word<(1 + ROT) % 5>(s);
  r.x2 = xorwow_word<(2 + ROT) % 5>(s);
  r.x3 = xorwow_word<(3 + ROT) % 5>(s);
  r.x4 = xorwow_word<(4 + ROT) % 5>(s);
  r.d = s.d;
  return r;
}

// The test of one pass: cardioid / bulb test and HEAD's iterations of the lanes in `valid`, then the
// survivors' c goes to Q0 (slot (tail + rank) & 127 of the ring at LDS byte address q0_lds: q0_cr there,
// q0_ci 1024 bytes on).  alive0: lanes outside both regions (cudabrot.cu:398); alive4: lanes that have
// not escaped after HEAD's steps; cnt: the iterations the reference executes for these samples up to there.
// 0x3fd00000 / 0x40300000: the high words of 0.25 and 16.0 as VOPC literals.
#define CB_STEP_LIT                                   \
  "s_bcnt1_i32_b64 %[tmp], exec\n\t"                  \
  "v_mul_f64 %[a], %[i], %[i]\n\t"                    \
  "s_add_u32 %[cnt], %[cnt], %[tmp]\n\t"              \
  "v_fma_f64 %[a], %[r], %[r], -%[a]\n\t"             \
  "v_fma_f64 %[i], " CB_AL "%[r]" CB_AR ", " CB_AL "%[i]" CB_AR ", %[ci]\n\t"             \
  "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"              \
  "v_mul_f64 %[a], %[r], %[r]\n\t"                    \
  "v_fma_f64 %[a], %[i], %[i], %[a]\n\t"              \
  "v_cmpx_nlt_f64_e32 vcc, 0x40300000, %[a]\n\t"
#ifdef CB_BURNING_SHIP
#define CB_HEAD_SHORTCUT                                                                               \
  "v_mul_f64 %[a], %[ci], %[ci]\n\t" /* II; no shortcut in this variant (cudabrot.cu:397-399) */      \
  "s_mov_b64 %[alive0], exec\n\t"
#else
#define CB_HEAD_SHORTCUT                                                        \
  "v_mul_f64 %[a], %[ci], %[ci]\n\t"            /* II */                       \
  "v_add_f64 %[x], %[cr], -0.5\n\t"             /* X = 2 (re - 1/4) */         \
  "v_add_f64 %[r], %[cr], 2.0\n\t"              /* T = 2 (re + 1) */           \
  "v_fma_f64 %[q], %[x], %[x], %[a]\n\t"        /* Q */                        \
  "v_fma_f64 %[r], %[r], %[r], %[a]\n\t"        /* bulb: fma(T,T,II) */        \
  "v_fma_f64 %[x], %[x], 2.0, %[q]\n\t"         /* S */                        \
  "v_cmp_ngt_f64_e32 vcc, 0x3fd00000, %[r]\n\t" /* !(bulb < 1/4) */            \
  "v_mul_f64 %[q], %[q], %[x]\n\t"              /* Q * S */                    \
  "s_mov_b64 %[alive0], vcc\n\t"                                               \
  "v_cmp_nlt_f64_e64 %[alive4], %[q], %[a]\n\t" /* !(Q*S < II) */              \
  "s_and_b64 %[alive0], %[alive0], %[alive4]\n\t"
#endif
#define CB_HEAD_TEST                                                            \
  "s_mov_b64 exec, %[valid]\n\t"                                               \
  CB_HEAD_SHORTCUT                                                              \
  "s_mov_b64 exec, %[alive0]\n\t"                                              \
  "s_bcnt1_i32_b64 %[cnt], %[alive0]\n\t"                                      \
  /* step 1 from z = c; its first product I*I is II */                          \
  "v_fma_f64 %[a], %[cr], %[cr], -%[a]\n\t"                                    \
  "v_fma_f64 %[i], " CB_AL "%[cr]" CB_AR ", " CB_AL "%[ci]" CB_AR ", %[ci]\n\t" \
  "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"                                       \
  "v_mul_f64 %[a], %[r], %[r]\n\t"                                             \
  "v_fma_f64 %[a], %[i], %[i], %[a]\n\t"                                       \
  "v_cmpx_nlt_f64_e32 vcc, 0x40300000, %[a]\n\t"                               \
  CB_HEAD_MORE_STEPS                                                            \
  "s_mov_b64 %[alive4], exec\n\t"                                              \
  /* survivors (EXEC) -> Q0 */                                                  \
  "v_mbcnt_lo_u32_b32 %[slot], exec_lo, 0\n\t"                                 \
  "v_mbcnt_hi_u32_b32 %[slot], exec_hi, %[slot]\n\t"                           \
  "v_add_u32 %[slot], %[tail], %[slot]\n\t"                                    \
  "v_and_b32 %[slot], 0x7f, %[slot]\n\t"                                       \
  "v_lshl_add_u32 %[slot], %[slot], 3, %[lds]\n\t"                             \
  "ds_write2st64_b64 %[slot], %[cr], %[ci] offset1:2\n\t"                      \
  "s_mov_b64 exec, %[save]\n\t"

// HEAD passes in a row (at least one).  Each takes one sample per lane of `valid` (cudabrot.cu:392-393,
// 398, 326-337).  The statement stops behind the pass after which
//   the input has ended (samples_left == 0), or Q0 holds a MID pass (q0_count >= 64), or
//   the next pass is one the progress board is posted before (samples_left % 32 == 1).
// q0_tail = q0_head + q0_count on entry (only its low seven bits matter).  The three statistics are added to.
__device__ __forceinline__ void head_loop(Xorwow &g, uint32_t &samples_left, uint32_t &rot, unsigned long long valid,
                                          uint32_t q0_tail, uint32_t &q0_count, uint32_t q0_lds,
                                          uint32_t &n_rejected, uint32_t &n_too_fast, uint32_t &n_steps) {
  static_assert(kQ0Cap == 128, "ring mask and the 1024-byte distance of q0_ci in CB_HEAD_TEST");
  unsigned long long save, alive0, alive4, gone;
  uint32_t cnt, tmp, sc, slot, t, u, o1, o2;
  double a, r, i, x, q, f, cr, ci;
  // wave-uniform by construction; readfirstlane makes that provable where the compiler cannot see it
  samples_left = __builtin_amdgcn_readfirstlane(samples_left);
  rot = __builtin_amdgcn_readfirstlane(rot);
  q0_tail = __builtin_amdgcn_readfirstlane(q0_tail);
  q0_count = __builtin_amdgcn_readfirstlane(q0_count);
  n_rejected = __builtin_amdgcn_readfirstlane(n_rejected);
  n_too_fast = __builtin_amdgcn_readfirstlane(n_too_fast);
  n_steps = __builtin_amdgcn_readfirstlane(n_steps);
  asm volatile(
      "s_mov_b64 %[save], exec\n\t"
      "0:\n\t"
      "s_sub_u32 %[left], %[left], 1\n\t"
      CB_HEAD_DRAW_ANY_ROT
      CB_HEAD_TEST
      // the pass's statistics and the ring
      "s_andn2_b64 %[valid0], %[valid], %[alive0]\n\t"   // in the cardioid or the bulb
      "s_bcnt1_i32_b64 %[tmp], %[valid0]\n\t"
      "s_add_u32 %[rej], %[rej], %[tmp]\n\t"
      "s_andn2_b64 %[alive0], %[alive0], %[alive4]\n\t"  // escaped in HEAD: before min_iter
      "s_bcnt1_i32_b64 %[tmp], %[alive0]\n\t"
      "s_add_u32 %[fast], %[fast], %[tmp]\n\t"
      "s_add_u32 %[steps], %[steps], %[cnt]\n\t"
      "s_bcnt1_i32_b64 %[tmp], %[alive4]\n\t"
      "s_add_u32 %[q0c], %[q0c], %[tmp]\n\t"
      "s_add_u32 %[tail], %[tail], %[tmp]\n\t"
      // another pass?
      "s_cmp_eq_u32 %[left], 0\n\t"
      "s_cbranch_scc1 9f\n\t"
      "s_cmp_ge_u32 %[q0c], 64\n\t"
      "s_cbranch_scc1 9f\n\t"
      "s_and_b32 %[tmp], %[left], 31\n\t"
      "s_cmp_lg_u32 %[tmp], 1\n\t"
      "s_cbranch_scc1 0b\n\t"
      "9:\n\t"
      "s_nop 4\n\t"
      : [left] "+s"(samples_left), [rot] "+s"(rot), [tail] "+s"(q0_tail), [q0c] "+s"(q0_count), [rej] "+s"(n_rejected),
        [fast] "+s"(n_too_fast), [steps] "+s"(n_steps), [alive0] "=&s"(alive0), [alive4] "=&s"(alive4),
        [valid0] "=&s"(gone), [cnt] "=&s"(cnt), [save] "=&s"(save), [tmp] "=&s"(tmp), [sc] "=&s"(sc),
        [a] "=&v"(a), [r] "=&v"(r), [i] "=&v"(i), [x] "=&v"(x), [q] "=&v"(q), [slot] "=&v"(slot), [cr] "=&v"(cr),
        [ci] "=&v"(ci), [t] "=&v"(t), [u] "=&v"(u), [o1] "=&v"(o1), [o2] "=&v"(o2), [f] "=&v"(f),
        [x0] "+v"(g.x0), [x1] "+v"(g.x1), [x2] "+v"(g.x2), [x3] "+v"(g.x3), [x4] "+v"(g.x4), [xd] "+v"(g.d)
      : [valid] "s"(valid), [lds] "s"(q0_lds), [k2m50] "s"(0x1p-50), [kk] "v"(0x1p-50 - 4.0)
      : "vcc", "scc", "memory");
}

// ---- MID in one piece (every escape inside MID is too fast, survivors go on to LONG) ---------------
//
// The lanes of `take` pop c from Q0 (ring slot (q0_head + lane) & 127 at LDS byte address q0_lds),
// re-derive z after the HEAD iterations (Q0 keeps only c; no escape is possible there, so no
// compare), run n_steps more iterations under EXEC and push the survivors' (c, z) to Q1 (ring slot
// (q1_tail + rank) mod 96 at q1_lds; q1_ci, q1_r, q1_i follow at 768-byte distances).  lane_steps:
// the executed iterations of the n_steps window; alive: the survivors.
#define CB_STEP_NOTEST                                \
  "v_mul_f64 %[a], %[i], %[i]\n\t"                    \
  "v_fma_f64 %[a], %[r], %[r], -%[a]\n\t"             \
  "v_fma_f64 %[i], " CB_AL "%[r]" CB_AR ", " CB_AL "%[i]" CB_AR ", %[ci]\n\t"             \
  "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"
// map / shift / cols / rows: the interior map (DrawArgs::interior_map, 0: none; draw_wide.hip's mid_pass has the same
// lines): the cell of every popped c is looked up while the stage iterates, and a lane whose cell is marked -- proven
// never-escaping -- is not pushed (`hit`, among the lanes of `take`).
__device__ __forceinline__ void mid_pass(unsigned long long take, uint32_t lane_plus_head,
                                         uint32_t q0_lds, uint32_t n_steps, uint32_t q1_tail,
                                         uint32_t q1_lds, unsigned long long &alive,
                                         uint32_t &lane_steps, unsigned long long map, uint32_t shift,
                                         uint32_t cols, uint32_t rows, unsigned long long &hit) {
  static_assert(kQ0Cap == 128 && kQ1Cap == 96, "ring mask, ring length and plane distances below");
  unsigned long long save, t64;
  uint32_t cnt, tmp, ctr, slot, wr, mbit, hitb;
  double cr, ci, r, i, a;
  map = uniform_u64(map);
  shift = __builtin_amdgcn_readfirstlane(shift);
  cols = __builtin_amdgcn_readfirstlane(cols);
  rows = __builtin_amdgcn_readfirstlane(rows);
  asm volatile(
      "s_mov_b64 %[save], exec\n\t"
      "s_mov_b64 exec, %[take]\n\t"
      "v_and_b32 %[slot], 0x7f, %[lph]\n\t"
      "v_lshl_add_u32 %[slot], %[slot], 3, %[q0]\n\t"
      "ds_read_b64 %[cr], %[slot]\n\t"
      "ds_read_b64 %[ci], %[slot] offset:1024\n\t"
      "s_mov_b32 %[cnt], 0\n\t"
      "s_mov_b32 %[ctr], %[n]\n\t"
      "s_mov_b64 %[hit], 0\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_cmp_eq_u64 %[map], 0\n\t"
      "s_cbranch_scc1 10f\n\t"
      "v_add_f64 %[a], %[cr], 4.0\n\t"           // column floor((Cr + 4) 2^shift), row floor(|Ci| 2^shift): doubled coordinates
      "v_ldexp_f64 %[a], %[a], %[shift]\n\t"
      "v_cmp_le_f64_e32 vcc, 0, %[a]\n\t"
      "v_cvt_u32_f64 %[slot], %[a]\n\t"
      "v_ldexp_f64 %[a], |%[ci]|, %[shift]\n\t"
      "v_cvt_u32_f64 %[wr], %[a]\n\t"
      "v_cmp_gt_u32_e64 %[t64], %[cols], %[slot]\n\t"
      "s_and_b64 vcc, vcc, %[t64]\n\t"
      "v_cmp_gt_u32_e64 %[t64], %[rows], %[wr]\n\t"
      "s_and_b64 vcc, vcc, %[t64]\n\t"
      "v_mad_u32_u24 %[slot], %[wr], %[cols], %[slot]\n\t"
      "v_and_b32 %[mbit], 7, %[slot]\n\t"
      "v_lshrrev_b32 %[slot], 3, %[slot]\n\t"
      "v_mov_b32 %[hitb], 0\n\t"
      "s_mov_b64 exec, vcc\n\t"
      "global_load_ubyte %[hitb], %[slot], %[map]\n\t"
      "s_mov_b64 exec, %[take]\n\t"
      "10:\n\t"
      // HEAD's iterations again, from z = c (first product: I*I with I = ci)
      "v_mul_f64 %[a], %[ci], %[ci]\n\t"
      "v_fma_f64 %[a], %[cr], %[cr], -%[a]\n\t"
      "v_fma_f64 %[i], " CB_AL "%[cr]" CB_AR ", " CB_AL "%[ci]" CB_AR ", %[ci]\n\t"
      "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"
      CB_MID_REDERIVE_MORE
      "s_cmp_eq_u32 %[n], 0\n\t"
      "s_cbranch_scc1 2f\n\t"
      "1:\n\t"
      CB_STEP_LIT
      "s_cbranch_execz 2f\n\t"
      "s_sub_u32 %[ctr], %[ctr], 1\n\t"
      "s_cmp_lg_u32 %[ctr], 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "2:\n\t"
      "s_mov_b64 %[alive], exec\n\t"
      "s_cmp_eq_u64 %[map], 0\n\t"
      "s_cbranch_scc1 11f\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_mov_b64 exec, %[take]\n\t"
      "v_lshrrev_b32 %[hitb], %[mbit], %[hitb]\n\t"
      "v_and_b32 %[hitb], 1, %[hitb]\n\t"
      "v_cmp_ne_u32_e32 vcc, 0, %[hitb]\n\t"
      "s_mov_b64 %[hit], vcc\n\t"
      "s_andn2_b64 exec, %[alive], vcc\n\t"
      "11:\n\t"
      "v_mbcnt_lo_u32_b32 %[slot], exec_lo, 0\n\t"
      "v_mbcnt_hi_u32_b32 %[slot], exec_hi, %[slot]\n\t"
      "v_add_u32 %[slot], %[tail], %[slot]\n\t"          // < 96 + 64
      "v_subrev_u32 %[wr], 96, %[slot]\n\t"              // slot - 96: huge (unsigned) below 96
      "v_min_u32 %[slot], %[slot], %[wr]\n\t"
      "v_lshl_add_u32 %[slot], %[slot], 3, %[q1]\n\t"
      "ds_write_b64 %[slot], %[cr]\n\t"
      "ds_write_b64 %[slot], %[ci] offset:768\n\t"
      "ds_write_b64 %[slot], %[r] offset:1536\n\t"
      "ds_write_b64 %[slot], %[i] offset:2304\n\t"
      "s_mov_b64 exec, %[save]\n\t"
      "s_nop 4\n\t"
      : [alive] "=&s"(alive), [cnt] "=&s"(cnt), [save] "=&s"(save), [tmp] "=&s"(tmp), [ctr] "=&s"(ctr),
        [slot] "=&v"(slot), [wr] "=&v"(wr), [cr] "=&v"(cr), [ci] "=&v"(ci), [a] "=&v"(a), [r] "=&v"(r),
        [i] "=&v"(i), [hit] "=&s"(hit), [t64] "=&s"(t64), [mbit] "=&v"(mbit), [hitb] "=&v"(hitb)
      : [take] "s"(take), [lph] "v"(lane_plus_head), [q0] "s"(q0_lds), [n] "s"(n_steps),
        [tail] "v"(q1_tail), [q1] "s"(q1_lds), [map] "s"(map), [shift] "s"(shift), [cols] "s"(cols), [rows] "s"(rows)
      : "vcc", "scc", "memory");
  lane_steps = cnt;
}

// ---- two orbits per lane, EXEC untouched (LONG) ----------------------------------------------------
//
// The LONG stage keeps TWO independent orbits (A and B) per lane and interleaves them instruction by
// instruction: the step is a chain of ~6 dependent fp64 operations, and a SIMD whose waves drift
// apart must be able to fill the fp64 pipe from few waves.  Measured with this very chunk on MI355X
// (tools/microbench.hip): 84.5 % of fp64 issue peak at 4 waves per SIMD, 60 % from a single wave
// (one orbit per lane: 58 % / 32 %; measured with the seven-instruction form of the step).
//
// Two orbit sets cannot share one EXEC mask, so EXEC stays untouched: every lane computes every
// step (an idle or already escaped slot computes garbage that nothing reads) and the orbits still
// iterating are tracked in two scalar masks, la and lb: la &= !(4.0 < |zA|^2) after each step
// (cudabrot.cu:336).  The and for step n is issued inside step n+1, behind its first VALU
// instructions, so that the scalar unit never waits for the compare.  cnt += popcount(la) +
// popcount(lb) before each step keeps the exact count of iterations the reference would execute.
// Same instruction order per orbit as CB_STEP / mandel_step.
#define CB_STEP2                                          \
  "v_mul_f64 %[a0], %[ia], %[ia]\n\t"                     \
  "v_mul_f64 %[a1], %[ib], %[ib]\n\t"                     \
  "s_and_b64 %[la], %[la], %[c0]\n\t"                     \
  "v_fma_f64 %[a0], %[ra], %[ra], -%[a0]\n\t"             \
  "s_and_b64 %[lb], %[lb], %[c1]\n\t"                     \
  "v_fma_f64 %[a1], %[rb], %[rb], -%[a1]\n\t"             \
  "s_bcnt1_i32_b64 %[t0], %[la]\n\t"                      \
  "v_fma_f64 %[ia], " CB_AL "%[ra]" CB_AR ", " CB_AL "%[ia]" CB_AR ", %[cia]\n\t"             \
  "s_bcnt1_i32_b64 %[t1], %[lb]\n\t"                      \
  "v_fma_f64 %[ib], " CB_AL "%[rb]" CB_AR ", " CB_AL "%[ib]" CB_AR ", %[cib]\n\t"             \
  "s_add_u32 %[cnt], %[cnt], %[t0]\n\t"                   \
  "v_fma_f64 %[ra], %[a0], 0.5, %[cra]\n\t"               \
  "s_add_u32 %[cnt], %[cnt], %[t1]\n\t"                   \
  "v_fma_f64 %[rb], %[a1], 0.5, %[crb]\n\t"               \
  "v_mul_f64 %[a0], %[ra], %[ra]\n\t"                     \
  "v_mul_f64 %[a1], %[rb], %[rb]\n\t"                     \
  "v_fma_f64 %[a0], %[ia], %[ia], %[a0]\n\t"              \
  "v_fma_f64 %[a1], %[ib], %[ib], %[a1]\n\t"              \
  "v_cmp_nlt_f64_e64 %[c0], %[k16], %[a0]\n\t"            \
  "v_cmp_nlt_f64_e64 %[c1], %[k16], %[a1]\n\t"
#define CB_STEP2X4 CB_STEP2 CB_STEP2 CB_STEP2 CB_STEP2
#define CB_STEP2X24 CB_STEP2X4 CB_STEP2X4 CB_STEP2X4 CB_STEP2X4 CB_STEP2X4 CB_STEP2X4
#if CB_CHUNK == 24
#define CB_STEP2_CHUNK CB_STEP2X24
#elif CB_CHUNK == 30
#define CB_STEP2_CHUNK CB_STEP2X24 CB_STEP2X4 CB_STEP2 CB_STEP2
#elif CB_CHUNK == 32
#define CB_STEP2_CHUNK CB_STEP2X24 CB_STEP2X4 CB_STEP2X4
#elif CB_CHUNK == 36
#define CB_STEP2_CHUNK CB_STEP2X24 CB_STEP2X4 CB_STEP2X4 CB_STEP2X4
#elif CB_CHUNK == 60
#define CB_STEP2_CHUNK CB_STEP2X24 CB_STEP2X24 CB_STEP2X4 CB_STEP2X4 CB_STEP2X4
#elif CB_CHUNK == 90
#define CB_STEP2_CHUNK CB_STEP2X24 CB_STEP2X24 CB_STEP2X24 CB_STEP2X4 CB_STEP2X4 CB_STEP2X4 CB_STEP2X4 CB_STEP2 CB_STEP2
#elif CB_CHUNK == 120
#define CB_STEP2_CHUNK CB_STEP2X24 CB_STEP2X24 CB_STEP2X24 CB_STEP2X24 CB_STEP2X24
#else
#error "unroll CB_STEP2 for this chunk length"
#endif
#define CB_STR2(x) #x
#define CB_STR(x) CB_STR2(x)
#define CB_CHUNK_S CB_STR(CB_CHUNK)  // the chunk length as a constant of the asm blocks below (VOP2 / VOPC e32: may be a literal)

// kChunk steps on orbit A of the lanes in mask_a and on orbit B of the lanes in mask_b (both
// wave-uniform; called with EXEC = all 64 lanes).  esc_a / esc_b receive the lanes whose orbit
// escaped; the other orbits of the masks advance by kChunk iterations; lane_steps receives the
// executed lane-steps.  z of slots outside the masks is clobbered.
__device__ __forceinline__ void iterate_chunk2(unsigned long long mask_a, unsigned long long mask_b,
                                               Orbit &oa, Orbit &ob, unsigned long long &esc_a,
                                               unsigned long long &esc_b, uint32_t &lane_steps) {
  unsigned long long la = mask_a, lb = mask_b, c0, c1;
  uint32_t cnt, t0, t1;
  double a0, a1;
  const double k16 = 16.0;
  asm volatile(
      "s_mov_b32 %[cnt], 0\n\t"
      "s_mov_b64 %[c0], -1\n\t"
      "s_mov_b64 %[c1], -1\n\t"
      CB_STEP2_CHUNK
      "s_and_b64 %[la], %[la], %[c0]\n\t"
      "s_and_b64 %[lb], %[lb], %[c1]\n\t"
      : [ra] "+v"(oa.r), [ia] "+v"(oa.i), [rb] "+v"(ob.r), [ib] "+v"(ob.i), [la] "+s"(la),
        [lb] "+s"(lb), [a0] "=&v"(a0), [a1] "=&v"(a1),
        [c0] "=&s"(c0), [c1] "=&s"(c1), [cnt] "=&s"(cnt), [t0] "=&s"(t0), [t1] "=&s"(t1)
      : [cra] "v"(oa.cr), [cia] "v"(oa.ci), [crb] "v"(ob.cr), [cib] "v"(ob.ci), [k16] "s"(k16)
      : "scc");
  esc_a = mask_a & ~la;
  esc_b = mask_b & ~lb;
  lane_steps = cnt;
}

// ---- the same chunk with the escape test on every tenth step only -----------------------------------------
//
// Three of the seven instructions of a step exist for the escape test (|z|^2 and the compare).  The LONG
// stage does not need the step at which an orbit escapes, only WHETHER it escaped inside the chunk (the
// accept filter goes by chunk, cudabrot.cu:407-408; the REPLAY stage finds the exact index again), and escape
// is absorbing: with |c| <= 2, |z| > 2 implies |z^2 + c| >= |z|^2 - |c| > |z|.  In fp64 that holds up to
// rounding: if the reference's test fires at some step (4 |z|^2 = M > 16 on the doubled coordinates), then
// writing |Z_k| >= 4 - e_k for the steps after it, e' <= 4 e + 2^-39 (|C| <= 4 + 2^-40 for every sample that
// survives HEAD -- a larger |c| escapes at iteration 0 with margin; rounding of one step <= 2^-44 while |Z| <=
// 8, beyond that |Z'| >= 28), so nine steps later e <= 2^-21 and M >= 16 - 2^-17, or M is infinite / NaN.
// The chunk therefore computes |Z|^2 on every tenth step only and compares it with kSparseThreshold =
// 16 - 2^-10 (64 times the bound; `le` is false for NaN): a lane above it stops counting as alive.  At the
// end of the chunk such a lane has escaped for certain if its FINAL M is above 16 (or NaN): escaped at that
// very step if not before.  What is left -- a lane that was above the threshold at a test step and is at or
// below 16 at the end: an orbit grazing |z| = 2 without leaving, one test step in 3 x 10^8 -- is decided
// exactly by recomputing the orbit from z0 = c with the per-step test (verify_chunk_escape below).
// 8 instructions per step for the two orbits + 6 per test step: 258 per 30 steps instead of 420, and no scalar
// bookkeeping inside the chunk: the executed iterations are counted per chunk (kChunk per orbit that ran it)
// and the over-count of an orbit's last chunk is taken back when REPLAY knows its escape index
// (long_overcount).  Only used when every escape inside a full LONG chunk is accepted (min_iter <= the start
// of the LONG stage): an escape that is too fast is never replayed, so its index would stay unknown.
[[maybe_unused]] constexpr double kSparseThreshold = 16.0 - 0x1p-10;  // (the value capi.hip puts into DrawArgs::sparse_threshold)
#define CB_STEP2_NT                                       \
  "v_mul_f64 %[a0], %[ia], %[ia]\n\t"                     \
  "v_mul_f64 %[a1], %[ib], %[ib]\n\t"                     \
  "v_fma_f64 %[a0], %[ra], %[ra], -%[a0]\n\t"             \
  "v_fma_f64 %[a1], %[rb], %[rb], -%[a1]\n\t"             \
  "v_fma_f64 %[ia], " CB_AL "%[ra]" CB_AR ", " CB_AL "%[ia]" CB_AR ", %[cia]\n\t"             \
  "v_fma_f64 %[ib], " CB_AL "%[rb]" CB_AR ", " CB_AL "%[ib]" CB_AR ", %[cib]\n\t"             \
  "v_fma_f64 %[ra], %[a0], 0.5, %[cra]\n\t"               \
  "v_fma_f64 %[rb], %[a1], 0.5, %[crb]\n\t"
// the first step after a test step: the masks of that test are applied behind its first instructions
#define CB_STEP2_NT_AND                                   \
  "v_mul_f64 %[a0], %[ia], %[ia]\n\t"                     \
  "v_mul_f64 %[a1], %[ib], %[ib]\n\t"                     \
  "s_and_b64 %[la], %[la], %[c0]\n\t"                     \
  "v_fma_f64 %[a0], %[ra], %[ra], -%[a0]\n\t"             \
  "s_and_b64 %[lb], %[lb], %[c1]\n\t"                     \
  "v_fma_f64 %[a1], %[rb], %[rb], -%[a1]\n\t"             \
  "v_fma_f64 %[ia], " CB_AL "%[ra]" CB_AR ", " CB_AL "%[ia]" CB_AR ", %[cia]\n\t"             \
  "v_fma_f64 %[ib], " CB_AL "%[rb]" CB_AR ", " CB_AL "%[ib]" CB_AR ", %[cib]\n\t"             \
  "v_fma_f64 %[ra], %[a0], 0.5, %[cra]\n\t"               \
  "v_fma_f64 %[rb], %[a1], 0.5, %[crb]\n\t"
// |Z|^2 of both orbits and the test against the threshold
#define CB_STEP2_TEST                                     \
  "v_mul_f64 %[a0], %[ra], %[ra]\n\t"                     \
  "v_mul_f64 %[a1], %[rb], %[rb]\n\t"                     \
  "v_fma_f64 %[a0], %[ia], %[ia], %[a0]\n\t"              \
  "v_fma_f64 %[a1], %[ib], %[ib], %[a1]\n\t"              \
  "v_cmp_le_f64_e64 %[c0], %[a0], %[kt]\n\t"              \
  "v_cmp_le_f64_e64 %[c1], %[a1], %[kt]\n\t"
#define CB_STEP2_NTX8 CB_STEP2_NT CB_STEP2_NT CB_STEP2_NT CB_STEP2_NT CB_STEP2_NT CB_STEP2_NT CB_STEP2_NT CB_STEP2_NT
// ---- ... and on the LAST step only --------------------------------------------------------------------
//
// (-DCB_SPARSE_STRIDE=10 keeps the form above.)  The tenth-step tests exist because the bound above is weak where
// |C| may exceed 4 by a rounding: there an orbit just beyond |Z| = 4 need not move away from it.  For a sample
// with |C|^2 < 16 - 2^-10 (|C| < 4 - 2^-13) it must: |Z| > 4 implies |Z'| >= |Z|^2 / 2 - |C| > 4 + 2^-13,
// against a rounding of 2^-44 per step -- once the reference's test has fired, every later |Z|^2 is above 16
// (or infinite, or NaN: `nle` is true for NaN).  So for such a sample ONE test, behind the chunk's last step, is
// the exact answer to "did it escape inside the chunk": !(M_last <= 16), M_last being the very expression the
// reference tests.  The other samples -- |c| within 2^-14 of 2, and of those only what has survived HEAD and
// MID: about one in 10^8 -- are decided by verify_chunk_escape at every chunk they run.  The class of a sample
// is recomputed with the chunk (3 instructions per orbit set: cheaper than a flag that lives across chunks).
// 8 instructions per step for the two orbits + 12 per chunk: 492 per 60 steps instead of 516.
#ifndef CB_SPARSE_STRIDE
#define CB_SPARSE_STRIDE 0
#endif
#define CB_STEP2_LAST_TEST                                \
  "v_mul_f64 %[a0], %[ra], %[ra]\n\t"                     \
  "v_mul_f64 %[a1], %[rb], %[rb]\n\t"                     \
  "v_fma_f64 %[a0], %[ia], %[ia], %[a0]\n\t"              \
  "v_fma_f64 %[a1], %[ib], %[ib], %[a1]\n\t"              \
  "v_cmp_nle_f64_e64 %[d0], %[a0], %[k16]\n\t"            \
  "v_cmp_nle_f64_e64 %[d1], %[a1], %[k16]\n\t"            \
  "v_mul_f64 %[a0], %[cra], %[cra]\n\t"                   \
  "v_mul_f64 %[a1], %[crb], %[crb]\n\t"                   \
  "v_fma_f64 %[a0], %[cia], %[cia], %[a0]\n\t"            \
  "v_fma_f64 %[a1], %[cib], %[cib], %[a1]\n\t"            \
  "v_cmp_nlt_f64_e64 %[c0], %[a0], %[kt]\n\t"             \
  "v_cmp_nlt_f64_e64 %[c1], %[a1], %[kt]\n\t"
#define CB_STEP2_NTX10 CB_STEP2_NTX8 CB_STEP2_NT CB_STEP2_NT
#define CB_STEP2_NTX30 CB_STEP2_NTX10 CB_STEP2_NTX10 CB_STEP2_NTX10
constexpr int kSparseStride = 10;  // steps between tests; the bound above is for at most ten
#define CB_SPARSE_GROUP_FIRST CB_STEP2_NT CB_STEP2_NTX8 CB_STEP2_NT CB_STEP2_TEST
#define CB_SPARSE_GROUP_NEXT CB_STEP2_NT_AND CB_STEP2_NTX8 CB_STEP2_NT CB_STEP2_TEST
#if CB_CHUNK == 30
#define CB_SPARSE_CHUNK CB_SPARSE_GROUP_FIRST CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT
#elif CB_CHUNK == 60
#define CB_SPARSE_CHUNK CB_SPARSE_GROUP_FIRST CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT
#elif CB_CHUNK == 40
#define CB_SPARSE_CHUNK CB_SPARSE_GROUP_FIRST CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT
#elif CB_CHUNK == 90
#define CB_SPARSE_GROUP_NEXT3 CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT
#define CB_SPARSE_CHUNK CB_SPARSE_GROUP_FIRST CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT3 CB_SPARSE_GROUP_NEXT3
#elif CB_CHUNK == 120
#define CB_SPARSE_GROUP_NEXT3 CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT
#define CB_SPARSE_CHUNK CB_SPARSE_GROUP_FIRST CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT CB_SPARSE_GROUP_NEXT3 CB_SPARSE_GROUP_NEXT3 CB_SPARSE_GROUP_NEXT3
#else
#define CB_SPARSE_CHUNK
#endif
static_assert(kChunk % kSparseStride == 0 || true, "sparse chunks are groups of ten steps");

// The chunk of iterate_chunk2 with sparse tests.  esc_*: lanes of the masks that stopped counting as alive;
// sure_*: lanes whose final |Z|^2 is above 16 or NaN (for a lane of esc_*: it escaped inside the chunk for
// certain).  The caller decides esc & ~sure exactly (verify_chunk_escape).
__device__ __forceinline__ void iterate_chunk2_sparse(unsigned long long mask_a, unsigned long long mask_b,
                                                      Orbit &oa, Orbit &ob, unsigned long long &esc_a,
                                                      unsigned long long &esc_b, unsigned long long &sure_a,
                                                      unsigned long long &sure_b, double threshold) {
  unsigned long long la = mask_a, lb = mask_b, c0, c1, d0, d1;
  double a0, a1;
  const double k16 = 16.0, kt = threshold;  // kSparseThreshold, or a test's lower one (DrawArgs::sparse_threshold)
#if CB_SPARSE_STRIDE == 0
  // one test behind the last step: esc = escaped (exact for the lanes of sure); sure = |C|^2 below the threshold
  static_assert(kChunk == 30 || kChunk == 60 || kChunk == 90 || kChunk == 120, "unrolled in thirties");
  asm volatile(
      CB_STEP2_NTX30
#if CB_CHUNK >= 60
      CB_STEP2_NTX30
#endif
#if CB_CHUNK >= 90
      CB_STEP2_NTX30
#endif
#if CB_CHUNK >= 120
      CB_STEP2_NTX30
#endif
      CB_STEP2_LAST_TEST
      "s_nop 2\n\t"
      : [ra] "+v"(oa.r), [ia] "+v"(oa.i), [rb] "+v"(ob.r), [ib] "+v"(ob.i), [a0] "=&v"(a0), [a1] "=&v"(a1),
        [c0] "=&s"(c0), [c1] "=&s"(c1), [d0] "=&s"(d0), [d1] "=&s"(d1)
      : [cra] "v"(oa.cr), [cia] "v"(oa.ci), [crb] "v"(ob.cr), [cib] "v"(ob.ci), [k16] "s"(k16), [kt] "s"(kt)
      : "scc");
  (void) la;
  (void) lb;
  // the caller decides esc & ~sure exactly: make that "every lane that is not sure" (escaped or not at the end)
  esc_a = mask_a & (d0 | c0);
  esc_b = mask_b & (d1 | c1);
  sure_a = ~c0;
  sure_b = ~c1;
#else
  asm volatile(
      CB_SPARSE_CHUNK
      "v_cmp_nle_f64_e64 %[d0], %[a0], %[k16]\n\t"
      "v_cmp_nle_f64_e64 %[d1], %[a1], %[k16]\n\t"
      "s_and_b64 %[la], %[la], %[c0]\n\t"
      "s_and_b64 %[lb], %[lb], %[c1]\n\t"
      "s_nop 2\n\t"
      : [ra] "+v"(oa.r), [ia] "+v"(oa.i), [rb] "+v"(ob.r), [ib] "+v"(ob.i), [la] "+s"(la),
        [lb] "+s"(lb), [a0] "=&v"(a0), [a1] "=&v"(a1),
        [c0] "=&s"(c0), [c1] "=&s"(c1), [d0] "=&s"(d0), [d1] "=&s"(d1)
      : [cra] "v"(oa.cr), [cia] "v"(oa.ci), [crb] "v"(ob.cr), [cib] "v"(ob.ci), [k16] "s"(k16), [kt] "s"(kt)
      : "scc");
  esc_a = mask_a & ~la;
  esc_b = mask_b & ~lb;
  sure_a = d0;
  sure_b = d1;
#endif
}

// The exact decision for the lanes of `doubt`: did the orbit with starting point (cr, ci) escape during the
// kChunk iterations after its first `done` ones (lane-wise)?  Recomputed from z0 = c with the reference's
// test after every step (cudabrot.cu:326-337); the steps before the chunk passed that test when they were
// made.  Rare (see above), so plain C++ under EXEC: same arithmetic as the asm (device_math.h).
__device__ __forceinline__ unsigned long long verify_chunk_escape(unsigned long long doubt, const Orbit &o,
                                                                  int done) {
  bool escaped = false;
  if (lane_in(doubt)) {
    double r = o.cr, i = o.ci;
    for (int k = 0; k < done; ++k) {
#ifdef CB_BURNING_SHIP
      (void) mandel_step2_ship(o.cr, o.ci, r, i);
#else
      (void) mandel_step2(o.cr, o.ci, r, i);
#endif
    }
    for (int k = 0; k < kChunk && !escaped; ++k) {
#ifdef CB_BURNING_SHIP
      escaped = mandel_step2_ship(o.cr, o.ci, r, i) > 16.0;
#else
      escaped = mandel_step2(o.cr, o.ci, r, i) > 16.0;
#endif
    }
  }
  return __ballot(escaped);
}

// Iterations the sparse chunks counted beyond the escape of an orbit that took `steps` iterations in all
// (escape index + 1): an orbit that escaped inside a full LONG chunk was counted to the end of that chunk.
__device__ __forceinline__ uint32_t long_overcount(int steps, int long_start, int tail_start) {
  if (steps <= long_start || steps > tail_start) return 0u;  // escaped before the LONG stage / in its exact tail chunk
  const uint32_t rem = (uint32_t) (steps - long_start) % (uint32_t) kChunk;
  return rem ? (uint32_t) kChunk - rem : 0u;
}

// ---- LONG bookkeeping around a chunk, one orbit slot at a time ---------------------------------------
//
// long_refill: the idle lanes (l_rem == 0) of the slot take (c, z) from Q1 -- ring slot
// (q1_head + rank) mod 96 at LDS byte address q1_lds, planes 768 bytes apart -- with l_rem =
// long_steps and the periodicity check's saved point = the entry point.  `taken`: orbits popped;
// `full`: lanes with a whole chunk ahead; `tail`: lanes left with exactly the last, shorter chunk
// (l_rem == tail_steps; pass ~0 when there is none).
__device__ __forceinline__ void long_refill(Orbit &o, double &seen_r, double &seen_i, int &l_rem,
                                            uint32_t q1_head, uint32_t q1_count, uint32_t q1_lds,
                                            uint32_t long_steps, uint32_t tail_value, uint32_t &taken,
                                            unsigned long long &full, unsigned long long &tail) {
  static_assert(kQ1Cap == 96, "ring length and plane distances below");
  unsigned long long save;
  uint32_t n, rank, slot, t;
  asm volatile(
      "v_cmp_eq_u32_e32 vcc, 0, %[lrem]\n\t"            // idle lanes
      "s_bcnt1_i32_b64 %[n], vcc\n\t"
      "s_min_u32 %[n], %[n], %[qc]\n\t"
      "s_cmp_eq_u32 %[n], 0\n\t"
      "s_cbranch_scc1 1f\n\t"
      "v_mbcnt_lo_u32_b32 %[rank], vcc_lo, 0\n\t"
      "v_mbcnt_hi_u32_b32 %[rank], vcc_hi, %[rank]\n\t"
      "s_mov_b64 %[save], exec\n\t"
      "s_mov_b64 exec, vcc\n\t"
      "v_cmpx_gt_u32_e32 vcc, %[n], %[rank]\n\t"         // the first n idle lanes
      "v_add_u32 %[slot], %[head], %[rank]\n\t"          // < 96 + 64
      "v_subrev_u32 %[t], 96, %[slot]\n\t"
      "v_min_u32 %[slot], %[slot], %[t]\n\t"
      "v_lshl_add_u32 %[slot], %[slot], 3, %[q1]\n\t"
      "ds_read_b64 %[cr], %[slot]\n\t"
      "ds_read_b64 %[ci], %[slot] offset:768\n\t"
      "ds_read_b64 %[r], %[slot] offset:1536\n\t"
      "ds_read_b64 %[i], %[slot] offset:2304\n\t"
      "ds_read_b64 %[sr], %[slot] offset:1536\n\t"      // the saved point of the periodicity check = z
      "ds_read_b64 %[si], %[slot] offset:2304\n\t"
      "v_mov_b32 %[lrem], %[ls]\n\t"
      "s_mov_b64 exec, %[save]\n\t"
      "1:\n\t"
      "v_cmp_le_u32_e32 vcc, " CB_CHUNK_S ", %[lrem]\n\t"   // (e32: the chunk length may be a literal)
      "v_cmp_eq_u32_e64 %[tail], %[tv], %[lrem]\n\t"
      "s_mov_b64 %[full], vcc\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      : [cr] "+v"(o.cr), [ci] "+v"(o.ci), [r] "+v"(o.r), [i] "+v"(o.i), [sr] "+v"(seen_r), [si] "+v"(seen_i),
        [lrem] "+v"(l_rem), [n] "=&s"(n), [full] "=&s"(full), [tail] "=&s"(tail), [save] "=&s"(save),
        [rank] "=&v"(rank), [slot] "=&v"(slot), [t] "=&v"(t)
      : [qc] "s"(q1_count), [head] "s"(q1_head), [q1] "s"(q1_lds), [ls] "s"(long_steps), [tv] "s"(tail_value)
      : "vcc", "scc", "memory");
  taken = n;
}

#define CB_RETIRE_SURVIVORS                                \
      "v_subrev_u32 %[lrem], " CB_CHUNK_S ", %[lrem]\n\t" \
      "v_cmp_eq_u32_e64 %[ended], 0, %[lrem]\n\t"        \
      "s_cmp_eq_u32 %[chkf], 0\n\t"                      \
      "s_cbranch_scc1 2f\n\t"                            \
      "v_cmp_eq_u64_e32 vcc, %[r], %[sr]\n\t"            \
      "s_mov_b64 %[per], vcc\n\t"                        \
      "v_cmp_eq_u64_e32 vcc, %[i], %[si]\n\t"            \
      "s_and_b64 %[per], %[per], vcc\n\t"                \
      "s_andn2_b64 %[per], %[per], %[ended]\n\t"         \
      "s_cmp_eq_u64 %[per], 0\n\t"                       \
      "s_cbranch_scc1 2f\n\t"                            \
      "s_mov_b64 exec, %[per]\n\t"                       \
      "v_add_co_u32 %[klo], vcc, %[klo], %[lrem]\n\t"    \
      "v_addc_co_u32 %[khi], vcc, 0, %[khi], vcc\n\t"    \
      "v_mov_b32 %[lrem], 0\n\t"                         \
      "s_andn2_b64 exec, %[ran], %[esc]\n\t"             \
      "2:\n\t"                                           \
      "s_andn2_b64 exec, exec, %[per]\n\t"
#define CB_RETIRE_TAIL                                     \
      "v_sub_u32 %[t], %[ls], %[lrem]\n\t"               \
      "v_cvt_f32_u32 %[t], %[t]\n\t"                     \
      "v_mul_f32 %[t], %[invl], %[t]\n\t"                \
      "v_rndne_f32 %[t], %[t]\n\t"                       \
      "v_and_b32 %[t], %[kmask], %[t]\n\t"               \
      "v_cmpx_eq_u32_e32 vcc, 0, %[t]\n\t"               \
      "v_mov_b64 %[sr], %[r]\n\t"                        \
      "v_mov_b64 %[si], %[i]\n\t"                        \
      "3:\n\t"                                           \
      "s_mov_b64 exec, %[save]\n\t"                      \
      "s_nop 4\n\t"
#define CB_RETIRE_ESCAPED                                  \
      "s_mov_b64 %[save], exec\n\t"  \
      "s_mov_b64 %[push], 0\n\t"  \
      "s_mov_b64 %[ended], 0\n\t"  \
      "s_mov_b64 %[per], 0\n\t"  \
      "s_mov_b64 exec, %[esc]\n\t"  \
      "s_cbranch_execz 1f\n\t"  \
      "v_cmp_ge_i32_e32 vcc, %[thr], %[lrem]\n\t"  \
      "v_mov_b32 %[lr], %[lrem]\n\t"  \
      "v_mov_b32 %[lrem], 0\n\t"  \
      "s_mov_b64 %[push], vcc\n\t"  \
      "s_mov_b64 exec, vcc\n\t"  \
      "s_cbranch_execz 1f\n\t"  \
      "v_mbcnt_lo_u32_b32 %[slot], exec_lo, 0\n\t"  \
      "v_mbcnt_hi_u32_b32 %[slot], exec_hi, %[slot]\n\t"  \
      "v_add_u32 %[slot], %[tail2], %[slot]\n\t"  \
      "v_subrev_u32 %[t], 192, %[slot]\n\t"  \
      "v_min_u32 %[slot], %[slot], %[t]\n\t"  \
      "v_lshl_add_u32 %[t], %[slot], 2, %[q2]\n\t"  \
      "v_lshl_add_u32 %[slot], %[slot], 3, %[q2]\n\t"  \
      "ds_write_b32 %[t], %[lr] offset:3072\n\t"  /* q2_lrem: what the orbit had left at the start of this chunk */ \
      "ds_write_b64 %[slot], %[cr]\n\t"  \
      "ds_write_b64 %[slot], %[ci] offset:1536\n\t"  \
      "1:\n\t"  \
      "s_andn2_b64 exec, %[ran], %[esc]\n\t"  \
      "s_cbranch_execz 3f\n\t"

// long_retire: after a chunk on the lanes of `ran`, of which `esc` escaped.
//   escaped lanes    l_rem = 0; those whose chunk lies at or above min_iter (l_rem <= accept_rem, the
//                    chunk being on one side of min_iter, cudabrot.cu:407-408) push c to Q2 -- ring slot
//                    (q2_tail + rank) mod 192 at q2_lds, q2_ci 1536 bytes on: `push`
//   the others       l_rem -= kChunk; `ended`: reached max_iter (IterateMandelbrot returns max,
//                    cudabrot.cu:339); `periodic` (check_periodic != 0 only): z is bit for bit the saved
//                    point, so the orbit repeats for ever and can never escape -- retired with the
//                    remaining iterations added to skip (two 32-bit halves); else Brent's schedule,
//                    refined: re-save z when the number of chunks done has no set bit below its top
//                    kBrentBits = 2 (1, 1.5, 2, 3, 4, 6 ... chunks).  The saved point must be older than
//                    the cycle is long, yet young enough to lie on the cycle: measured at C3, this
//                    schedule executes 3 % fewer iterations than powers of two alone, and keeping the
//                    top three bits 8 % more.  (Two saved points replaced in turn cut another 4 %, but
//                    their 8 registers end the co-residency with the scatter's count kernels: slower.)
__device__ __forceinline__ void long_retire(const Orbit &o, double &seen_r, double &seen_i, int &l_rem,
                                            uint32_t &skip_lo, uint32_t &skip_hi, unsigned long long ran,
                                            unsigned long long esc, int accept_rem, uint32_t long_steps,
                                            uint32_t check_periodic, uint32_t q2_tail, uint32_t q2_lds,
                                            unsigned long long &push, unsigned long long &ended,
                                            unsigned long long &periodic) {
  static_assert(kQ2Cap == 192, "ring length and plane distances below (q2_ci 1536, q2_lrem 3072 bytes on)");
  static_assert(kBrentBits >= 1 && kBrentBits <= 8, "bits of the chunk count kept by the save schedule");
  unsigned long long save;
  uint32_t slot, t, lr;
  // chunks done = steps done / kChunk, exactly, as a float (a few thousand at most): the count has no set
  // bit below its top kBrentBits iff the float's mantissa is zero below its top kBrentBits - 1 bits
  const float inv_chunk = 1.0f / (float) kChunk;
  const uint32_t low_mantissa = (1u << (24 - kBrentBits)) - 1u;
#define CB_RETIRE_OPERANDS                                                                                   \
  : [sr] "+v"(seen_r), [si] "+v"(seen_i), [lrem] "+v"(l_rem), [klo] "+v"(skip_lo), [khi] "+v"(skip_hi),      \
    [push] "=&s"(push), [ended] "=&s"(ended), [per] "=&s"(periodic), [save] "=&s"(save), [slot] "=&v"(slot), \
    [t] "=&v"(t), [lr] "=&v"(lr)                                                                             \
  : [ran] "s"(ran), [esc] "s"(esc), [thr] "s"(accept_rem), [ls] "s"(long_steps), [tail2] "s"(q2_tail),       \
    [chkf] "s"(check_periodic), [invl] "s"(inv_chunk), [kmask] "s"(low_mantissa),                            \
    [q2] "s"(q2_lds), [cr] "v"(o.cr), [ci] "v"(o.ci), [r] "v"(o.r), [i] "v"(o.i)                             \
  : "vcc", "scc", "memory"
  asm volatile(CB_RETIRE_ESCAPED CB_RETIRE_SURVIVORS CB_RETIRE_TAIL CB_RETIRE_OPERANDS);
#undef CB_RETIRE_OPERANDS
}

// ---- REPLAY burst: IterateAndRecord (cudabrot.cu:347-365) into the pixel stream --------------------
//
// One step for the lanes of `act` (EXEC), in the order of the reference's loop body, on doubled
// coordinates (R = 2 re, I = 2 im):
//   z <- z^2 + c            (CB_STEP's six fp64 instructions, same order)
//   IncrementPixelCounter   if (re >= min_re && im >= min_im) { col = (int)((re-min_re)/d_re); row
//                           likewise; if (col <u w && row <u h) append row<<16|col to the stream }
//                           (cudabrot.cu:308-312; the unsigned compares also reject the saturated
//                           conversions, and col, row cannot be negative past the first test)
//   if (|z|^2 > 4) leave    (cudabrot.cu:363) -- after recording the escaped point, like the reference
// re - min_re = fma(R, 0.5, -min_re) exactly (halving is exact).  (re - min_re) / d_re:
//   CB_REPLAY_BIN_POW2  both deltas are powers of two: the quotient is (re - min_re) * (1/d) exactly
//                       and, scaling being exact, equals the single fma(R, 0.5/d, -min_re/d);
//   CB_REPLAY_BIN_DIV   else the correctly rounded IEEE quotient by the instruction sequence hipcc
//                       emits for a double division (v_div_scale / v_rcp / Newton steps /
//                       v_div_fmas / v_div_fixup) -- but only when it can matter.  Only trunc(q) of the
//                       quotient q = RN(a / d) is used.  The estimate q' = RN(a * RN(1/d)) is within
//                       q * 2^-51 of q, so below 2^22 the two lie less than 2^-29 apart, and when the
//                       fraction of q' is inside (2^-24, 1 - 2^-24) no integer lies between them:
//                       trunc(q') = trunc(q).  From 2^22 on both are beyond any canvas the stream can
//                       describe (sides <= 65536), whatever the conversion yields.  If any lane of the
//                       step fails the test (|fract(q') - 1/2| < 1/2 - 2^-24; a NaN or infinite estimate
//                       fails it too) the whole wave takes the exact division: about one step in 10^7.
// The four tests re >= min_re, col < w (and im, row) are two: the quotient q = (re - min_re) / d_re has the sign of
// re - min_re (-0.0 included: a negative difference never rounds to +0), and for q >= 0 trunc(q) < w iff q < w, so
// 0 <= q < w -- as ONE unsigned 64-bit compare of q's bits with those of (double) w: a set sign bit or a NaN is
// above any such bound, and so is every q that the reference's (int) conversion would saturate.  The hits of a
// step are compacted with v_mbcnt and stored side by side (one coalesced store).  The stream word is
// row << rsh | col | tag (one channel: rsh = 16, tag = 0; fused channels: the index of the pass's channel above
// row and col), and only the lanes of `emit` write (fused channels: not the lanes still measuring
// their escape index).
#define CB_REPLAY_BIN_POW2                                \
  "v_fma_f64 %[fx], %[r], %[sx], %[ox]\n\t"               \
  "v_fma_f64 %[fy], %[i], %[sy], %[oy]\n\t"
#define CB_DIV(q, num, den)                                         \
  "v_div_scale_f64 %[d0], %[scp], " den ", " den ", " num "\n\t"    \
  "v_rcp_f64 %[d2], %[d0]\n\t"                                      \
  "v_div_scale_f64 %[d1], vcc, " num ", " den ", " num "\n\t"       \
  "v_fma_f64 %[d3], -%[d0], %[d2], 1.0\n\t"                         \
  "v_fma_f64 %[d2], %[d2], %[d3], %[d2]\n\t"                        \
  "v_fma_f64 %[d3], -%[d0], %[d2], 1.0\n\t"                         \
  "v_fma_f64 %[d2], %[d2], %[d3], %[d2]\n\t"                        \
  "v_mul_f64 %[d3], %[d1], %[d2]\n\t"                               \
  "v_fma_f64 %[d0], -%[d0], %[d3], %[d1]\n\t"                       \
  "v_div_fmas_f64 %[d0], %[d0], %[d2], %[d3]\n\t"                   \
  "v_div_fixup_f64 " q ", %[d0], " den ", " num "\n\t"
#define CB_REPLAY_BIN_DIV                                 \
  "v_fma_f64 %[fx], %[r], 0.5, -%[ox]\n\t"                \
  "v_fma_f64 %[fy], %[i], 0.5, -%[oy]\n\t"                \
  "v_mul_f64 %[d0], %[fx], %[rx]\n\t"                     \
  "v_mul_f64 %[d1], %[fy], %[ry]\n\t"                     \
  "v_fract_f64 %[d2], %[d0]\n\t"                          \
  "v_fract_f64 %[d3], %[d1]\n\t"                          \
  "v_add_f64 %[d2], %[d2], -0.5\n\t"                      \
  "v_add_f64 %[d3], %[d3], -0.5\n\t"                      \
  "v_cmp_nlt_f64_e64 %[scp], |%[d2]|, %[kg]\n\t"          \
  "v_cmp_nlt_f64_e64 vcc, |%[d3]|, %[kg]\n\t"             \
  "s_or_b64 %[scp], %[scp], vcc\n\t"                      \
  "s_cbranch_scc0 4f\n\t"                                 \
  CB_DIV("%[fx]", "%[fx]", "%[sx]") CB_DIV("%[fy]", "%[fy]", "%[sy]") \
  "s_branch 5f\n\t"                                       \
  "4:\n\t"                                                \
  "v_mov_b64 %[fx], %[d0]\n\t"                            \
  "v_mov_b64 %[fy], %[d1]\n\t"                            \
  "5:\n\t"

#define CB_REPLAY_HEAD                                    \
  "s_mov_b64 %[save], exec\n\t"                           \
  "s_mov_b32 %[cs], 0\n\t"                                \
  "s_mov_b32 %[ch], 0\n\t"                                \
  "s_mov_b32 %[ctr], %[n]\n\t"                            \
  "1:\n\t"                                                \
  "s_mov_b64 exec, %[act]\n\t"                            \
  "s_bcnt1_i32_b64 %[t], %[act]\n\t"                      \
  "v_mul_f64 %[a], %[i], %[i]\n\t"                        \
  "s_add_u32 %[cs], %[cs], %[t]\n\t"                      \
  "v_fma_f64 %[a], %[r], %[r], -%[a]\n\t"                 \
  "v_fma_f64 %[i], " CB_AL "%[r]" CB_AR ", " CB_AL "%[i]" CB_AR ", %[ci]\n\t"                 \
  "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"                  \
  "v_add_u32 %[ps], 1, %[ps]\n\t"                         \
  "v_mul_f64 %[a], %[r], %[r]\n\t"                        \
  "v_fma_f64 %[a], %[i], %[i], %[a]\n\t"
// ..._TAGGED: fused channels (the word carries the channel's tag, lanes still measuring do not write);
// ..._PLAIN: one channel -- no tag, every lane of the step writes.
#define CB_REPLAY_TAIL_TAGGED                                  \
  "v_cvt_i32_f64 %[col], %[fx]\n\t"                       \
  "v_cvt_i32_f64 %[row], %[fy]\n\t"                       \
  "v_cmp_nlt_f64_e64 %[alive], %[k16], %[a]\n\t"          \
  "v_cmp_gt_u64_e64 %[hx], %[wb], %[fx]\n\t"              \
  "v_cmp_gt_u64_e64 vcc, %[hb], %[fy]\n\t"                \
  "v_lshl_or_b32 %[e], %[row], %[rsh], %[col]\n\t"        \
  "v_or_b32 %[e], %[e], %[tag]\n\t"                       \
  "s_and_b64 vcc, vcc, %[hx]\n\t"                         \
  "s_and_b64 vcc, vcc, %[emit]\n\t"                       \
  "v_mbcnt_lo_u32_b32 %[pidx], vcc_lo, 0\n\t"             \
  "v_mbcnt_hi_u32_b32 %[pidx], vcc_hi, %[pidx]\n\t"       \
  "s_bcnt1_i32_b64 %[t], vcc\n\t"                         \
  "v_add_lshl_u32 %[pidx], %[pidx], %[fill], 2\n\t"       \
  "s_mov_b64 exec, vcc\n\t"                               \
  "global_store_dword %[pidx], %[e], %[base]\n\t"
#define CB_REPLAY_TAIL_PLAIN                                   \
  "v_cvt_i32_f64 %[col], %[fx]\n\t"                       \
  "v_cvt_i32_f64 %[row], %[fy]\n\t"                       \
  "v_cmp_nlt_f64_e64 %[alive], %[k16], %[a]\n\t"          \
  "v_cmp_gt_u64_e64 %[hx], %[wb], %[fx]\n\t"              \
  "v_cmp_gt_u64_e64 vcc, %[hb], %[fy]\n\t"                \
  "v_lshl_or_b32 %[e], %[row], %[rsh], %[col]\n\t"        \
  "s_and_b64 vcc, vcc, %[hx]\n\t"                         \
  "v_mbcnt_lo_u32_b32 %[pidx], vcc_lo, 0\n\t"             \
  "v_mbcnt_hi_u32_b32 %[pidx], vcc_hi, %[pidx]\n\t"       \
  "s_bcnt1_i32_b64 %[t], vcc\n\t"                         \
  "v_add_lshl_u32 %[pidx], %[pidx], %[fill], 2\n\t"       \
  "s_mov_b64 exec, vcc\n\t"                               \
  "global_store_dword %[pidx], %[e], %[base]\n\t"
// kChunked (canvases beyond 1024 tiles, BinLayout::chunked): the word goes to the current chunk of its GROUP of
// 1024 tiles instead of the end of the wave's segment.  EXEC = the lanes with a word (vcc of the tail above).
// group = tile >> 10; the group's {next word, end of chunk} (indices into the wave's segment) sit in the wave's
// LDS at gcl + 8 * group: one returning add takes the place, lanes that find it behind the end of the chunk
// (`over`) keep their word in e -- the burst then ends with this step and the caller opens new chunks for them
// (replay_burst_chunked).  row and col are free once the word is formed.
#define CB_REPLAY_GROUP_TAGGED  /* fused channels: the planes are one canvas of n_planes * tiles_y tile rows */ \
  "s_bcnt1_i32_b64 %[t], vcc\n\t"                         \
  "s_mov_b64 exec, vcc\n\t"                               \
  "v_lshrrev_b32 %[pl], %[chs], %[tag]\n\t"               \
  "v_lshrrev_b32 %[row], 7, %[row]\n\t"                   \
  "v_lshrrev_b32 %[col], 7, %[col]\n\t"                   \
  "v_mad_u32_u24 %[row], %[pl], %[tly], %[row]\n\t"       \
  "v_mad_u32_u24 %[row], %[row], %[tlx], %[col]\n\t"      \
  "v_lshrrev_b32 %[grp], 10, %[row]\n\t"
#define CB_REPLAY_GROUP_PLAIN                             \
  "s_bcnt1_i32_b64 %[t], vcc\n\t"                         \
  "s_mov_b64 exec, vcc\n\t"                               \
  "v_lshrrev_b32 %[row], 7, %[row]\n\t"                   \
  "v_lshrrev_b32 %[col], 7, %[col]\n\t"                   \
  "v_mad_u32_u24 %[row], %[row], %[tlx], %[col]\n\t"      \
  "v_lshrrev_b32 %[grp], 10, %[row]\n\t"
#define CB_REPLAY_CHUNKED                                 \
  "v_lshl_add_u32 %[col], %[grp], 3, %[gcl]\n\t"          \
  "ds_add_rtn_u32 %[pos], %[col], %[one]\n\t"             \
  "ds_read_b32 %[lim], %[col] offset:4\n\t"               \
  "s_waitcnt lgkmcnt(0)\n\t"                              \
  "v_cmp_lt_u32_e32 vcc, %[pos], %[lim]\n\t"              \
  "s_andn2_b64 %[over], exec, vcc\n\t"                    \
  "s_mov_b64 exec, vcc\n\t"                               \
  "v_lshlrev_b32 %[pidx], 2, %[pos]\n\t"                  \
  "global_store_dword %[pidx], %[e], %[base]\n\t"
// the tail of the tagged form up to the hit mask (no compaction: the place comes from the group's cursor)
#define CB_REPLAY_TAIL_HITS                               \
  "v_cvt_i32_f64 %[col], %[fx]\n\t"                       \
  "v_cvt_i32_f64 %[row], %[fy]\n\t"                       \
  "v_cmp_nlt_f64_e64 %[alive], %[k16], %[a]\n\t"          \
  "v_cmp_gt_u64_e64 %[hx], %[wb], %[fx]\n\t"              \
  "v_cmp_gt_u64_e64 vcc, %[hb], %[fy]\n\t"                \
  "v_lshl_or_b32 %[e], %[row], %[rsh], %[col]\n\t"        \
  "v_or_b32 %[e], %[e], %[tag]\n\t"                       \
  "s_and_b64 vcc, vcc, %[hx]\n\t"                         \
  "s_and_b64 vcc, vcc, %[emit]\n\t"
#define CB_REPLAY_TAIL_HITS_PLAIN                         \
  "v_cvt_i32_f64 %[col], %[fx]\n\t"                       \
  "v_cvt_i32_f64 %[row], %[fy]\n\t"                       \
  "v_cmp_nlt_f64_e64 %[alive], %[k16], %[a]\n\t"          \
  "v_cmp_gt_u64_e64 %[hx], %[wb], %[fx]\n\t"              \
  "v_cmp_gt_u64_e64 vcc, %[hb], %[fy]\n\t"                \
  "v_lshl_or_b32 %[e], %[row], %[rsh], %[col]\n\t"        \
  "s_and_b64 vcc, vcc, %[hx]\n\t"
#define CB_REPLAY_TAIL2_CHUNKED                           \
  "s_add_u32 %[fill], %[fill], %[t]\n\t"                  \
  "s_add_u32 %[ch], %[ch], %[t]\n\t"                      \
  "s_and_b64 %[act], %[act], %[alive]\n\t"                \
  "s_cmp_lg_u64 %[over], 0\n\t"                           \
  "s_cbranch_scc1 2f\n\t"                                 \
  "s_cmp_eq_u64 %[act], 0\n\t"                            \
  "s_cbranch_scc1 2f\n\t"                                 \
  "s_sub_u32 %[ctr], %[ctr], 1\n\t"                       \
  "s_cmp_lg_u32 %[ctr], 0\n\t"                            \
  "s_cbranch_scc1 1b\n\t"                                 \
  "2:\n\t"                                                \
  "s_mov_b64 exec, %[save]\n\t"                           \
  "s_nop 4\n\t"
#define CB_REPLAY_TAIL2                                   \
  "s_add_u32 %[fill], %[fill], %[t]\n\t"                  \
  "s_add_u32 %[ch], %[ch], %[t]\n\t"                      \
  "s_and_b64 %[act], %[act], %[alive]\n\t"                \
  "s_cmp_eq_u64 %[act], 0\n\t"                            \
  "s_cbranch_scc1 2f\n\t"                                 \
  "s_sub_u32 %[ctr], %[ctr], 1\n\t"                       \
  "s_cmp_lg_u32 %[ctr], 0\n\t"                            \
  "s_cbranch_scc1 1b\n\t"                                 \
  "2:\n\t"                                                \
  "s_mov_b64 exec, %[save]\n\t"                           \
  "s_nop 4\n\t"

// Up to n_steps (>= 1) replay steps on the lanes of `act`; the stream region must have room for
// 64 * n_steps more entries.  On return `act` holds the lanes still replaying, `fill` the new fill,
// lane_steps / hits the executed lane-steps and the entries appended.  p holds DOUBLED coordinates.
// The kernel's arguments, read afresh: a scalar load from the argument segment at the point of use
// instead of a value held in (and spilled from) scalar registers since the kernel began.
typedef const DrawArgs __attribute__((address_space(4))) *KernelArgs;
__device__ __forceinline__ KernelArgs fresh_args() {
  KernelArgs p = (KernelArgs) __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

template <bool kPow2>
__device__ __forceinline__ void replay_burst(unsigned long long &act, uint32_t n_steps, Orbit &p,
                                             int &p_steps, const Canvas &cv, uint32_t *region,
                                             uint32_t &fill, uint32_t &lane_steps, uint32_t &hits,
                                             uint32_t row_shift, uint32_t tag,
                                             unsigned long long emit, bool tagged,
                                             uint32_t *burst_steps = nullptr) {
  unsigned long long save, alive, hx, hy, scp;
  uint32_t cs, ch, ctr, t;
  double a, fx, fy, d0, d1, d2, d3;
  uint32_t col, row, pidx, e;
  // All "s" operands are wave-uniform by construction; uniform_*/readfirstlane make that provable.
  (void) cv;
  const KernelArgs ka = fresh_args();
  const double wb = (double) ka->w, hb = (double) ka->h;  // the bounds of the quotients, compared as bit patterns
  region = reinterpret_cast<uint32_t *>(uniform_u64(reinterpret_cast<unsigned long long>(region)));
  act = uniform_u64(act);
  fill = __builtin_amdgcn_readfirstlane(fill);
  n_steps = __builtin_amdgcn_readfirstlane(n_steps);
  row_shift = __builtin_amdgcn_readfirstlane(row_shift);
  emit = uniform_u64(emit);
  const double k16 = 16.0;
  if (kPow2) {
    // fx = fma(R, 0.5/d, -min/d): scale in a scalar pair, offset in a (wave-constant) vector pair --
    // a VALU instruction reads one scalar operand
    const double sx = ka->replay_scale_real, sy = ka->replay_scale_imag;
    const double ox = ka->replay_offset_real, oy = ka->replay_offset_imag;
    if (tagged) {
    asm volatile(CB_REPLAY_HEAD CB_REPLAY_BIN_POW2 CB_REPLAY_TAIL_TAGGED CB_REPLAY_TAIL2
                 : [r] "+v"(p.r), [i] "+v"(p.i), [ps] "+v"(p_steps), [act] "+s"(act), [fill] "+s"(fill),
                   [save] "=&s"(save), [alive] "=&s"(alive), [hx] "=&s"(hx), [hy] "=&s"(hy),
                   [cs] "=&s"(cs), [ch] "=&s"(ch), [ctr] "=&s"(ctr), [t] "=&s"(t), [a] "=&v"(a),
                   [fx] "=&v"(fx), [fy] "=&v"(fy), [col] "=&v"(col), [row] "=&v"(row),
                   [pidx] "=&v"(pidx), [e] "=&v"(e)
                 : [n] "s"(n_steps), [cr] "v"(p.cr), [ci] "v"(p.ci), [sx] "s"(sx), [sy] "s"(sy), [ox] "v"(ox), [oy] "v"(oy),
                   [wb] "s"(wb), [hb] "s"(hb), [base] "s"(region), [k16] "s"(k16), [rsh] "s"(row_shift),
                   [tag] "v"(tag), [emit] "s"(emit)
                 : "vcc", "scc", "memory");
    } else {
    asm volatile(CB_REPLAY_HEAD CB_REPLAY_BIN_POW2 CB_REPLAY_TAIL_PLAIN CB_REPLAY_TAIL2
                 : [r] "+v"(p.r), [i] "+v"(p.i), [ps] "+v"(p_steps), [act] "+s"(act), [fill] "+s"(fill),
                   [save] "=&s"(save), [alive] "=&s"(alive), [hx] "=&s"(hx), [hy] "=&s"(hy),
                   [cs] "=&s"(cs), [ch] "=&s"(ch), [ctr] "=&s"(ctr), [t] "=&s"(t), [a] "=&v"(a),
                   [fx] "=&v"(fx), [fy] "=&v"(fy), [col] "=&v"(col), [row] "=&v"(row),
                   [pidx] "=&v"(pidx), [e] "=&v"(e)
                 : [n] "s"(n_steps), [cr] "v"(p.cr), [ci] "v"(p.ci), [sx] "s"(sx), [sy] "s"(sy), [ox] "v"(ox), [oy] "v"(oy),
                   [wb] "s"(wb), [hb] "s"(hb), [base] "s"(region), [k16] "s"(k16), [rsh] "s"(row_shift),
                   [tag] "v"(tag), [emit] "s"(emit)
                 : "vcc", "scc", "memory");
    }
  } else {
    const double sx = ka->replay_scale_real, sy = ka->replay_scale_imag;
    const double ox = ka->replay_offset_real, oy = ka->replay_offset_imag;
    const double rx = ka->rcp_delta_real, ry = ka->rcp_delta_imag;
    const double kg = 0.5 - 0x1p-24;
    if (tagged) {
    asm volatile(CB_REPLAY_HEAD CB_REPLAY_BIN_DIV CB_REPLAY_TAIL_TAGGED CB_REPLAY_TAIL2
                 : [r] "+v"(p.r), [i] "+v"(p.i), [ps] "+v"(p_steps), [act] "+s"(act), [fill] "+s"(fill),
                   [save] "=&s"(save), [alive] "=&s"(alive), [hx] "=&s"(hx), [hy] "=&s"(hy),
                   [scp] "=&s"(scp), [cs] "=&s"(cs), [ch] "=&s"(ch), [ctr] "=&s"(ctr), [t] "=&s"(t),
                   [a] "=&v"(a), [fx] "=&v"(fx), [fy] "=&v"(fy), [d0] "=&v"(d0),
                   [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [col] "=&v"(col), [row] "=&v"(row),
                   [pidx] "=&v"(pidx), [e] "=&v"(e)
                 : [n] "s"(n_steps), [cr] "v"(p.cr), [ci] "v"(p.ci), [sx] "s"(sx), [sy] "s"(sy), [ox] "s"(ox), [oy] "s"(oy),
                   [rx] "s"(rx), [ry] "s"(ry), [kg] "s"(kg),
                   [wb] "s"(wb), [hb] "s"(hb), [base] "s"(region), [k16] "s"(k16), [rsh] "s"(row_shift),
                   [tag] "v"(tag), [emit] "s"(emit)
                 : "vcc", "scc", "memory");
    } else {
    asm volatile(CB_REPLAY_HEAD CB_REPLAY_BIN_DIV CB_REPLAY_TAIL_PLAIN CB_REPLAY_TAIL2
                 : [r] "+v"(p.r), [i] "+v"(p.i), [ps] "+v"(p_steps), [act] "+s"(act), [fill] "+s"(fill),
                   [save] "=&s"(save), [alive] "=&s"(alive), [hx] "=&s"(hx), [hy] "=&s"(hy),
                   [scp] "=&s"(scp), [cs] "=&s"(cs), [ch] "=&s"(ch), [ctr] "=&s"(ctr), [t] "=&s"(t),
                   [a] "=&v"(a), [fx] "=&v"(fx), [fy] "=&v"(fy), [d0] "=&v"(d0),
                   [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [col] "=&v"(col), [row] "=&v"(row),
                   [pidx] "=&v"(pidx), [e] "=&v"(e)
                 : [n] "s"(n_steps), [cr] "v"(p.cr), [ci] "v"(p.ci), [sx] "s"(sx), [sy] "s"(sy), [ox] "s"(ox), [oy] "s"(oy),
                   [rx] "s"(rx), [ry] "s"(ry), [kg] "s"(kg),
                   [wb] "s"(wb), [hb] "s"(hb), [base] "s"(region), [k16] "s"(k16), [rsh] "s"(row_shift),
                   [tag] "v"(tag), [emit] "s"(emit)
                 : "vcc", "scc", "memory");
    }
  }
  lane_steps = cs;
  hits = ch;
  // (diagnostics) steps of this burst: the loop leaves before its counter steps when the last lane ends
  if (burst_steps) *burst_steps = n_steps - ctr + (act == 0ull ? 1u : 0u);
}


// The same burst on a CHUNKED stream (BinLayout::chunked, CB_REPLAY_CHUNKED): every word goes to the current chunk
// of its group.  cursors / cursors_lds: the wave's {next word, end of chunk} pairs in LDS and their LDS byte
// address; next_chunk: the
// first chunk of the wave's segment that is still free (the caller has made sure that a burst cannot run out:
// it opens at most one chunk per group and one per kChunkWords words); desc: the wave's row of chunk_desc.
template <bool kPow2>
__device__ __forceinline__ void replay_burst_chunked(unsigned long long &act, uint32_t n_steps, Orbit &p,
                                                     int &p_steps, uint32_t *region, uint32_t &fill,
                                                     uint32_t &lane_steps, uint32_t &hits, uint32_t row_shift,
                                                     uint32_t tag, unsigned long long emit, bool tagged,
                                                     uint32_t *cursors, uint32_t cursors_lds, uint32_t &next_chunk,
                                                     uint32_t *desc, uint32_t *burst_steps = nullptr) {
  unsigned long long save, alive, hx, hy, scp, over;
  uint32_t cs, ch, ctr, t;
  double a, fx, fy, d0, d1, d2, d3;
  uint32_t col, row, pidx, e, pl, grp, pos, lim;
  const KernelArgs ka = fresh_args();
  const double wb = (double) ka->w, hb = (double) ka->h;  // the bounds of the quotients, compared as bit patterns
  region = reinterpret_cast<uint32_t *>(uniform_u64(reinterpret_cast<unsigned long long>(region)));
  act = uniform_u64(act);
  fill = __builtin_amdgcn_readfirstlane(fill);
  n_steps = __builtin_amdgcn_readfirstlane(n_steps);
  row_shift = __builtin_amdgcn_readfirstlane(row_shift);
  emit = uniform_u64(emit);
  cursors_lds = __builtin_amdgcn_readfirstlane(cursors_lds);
  const double k16 = 16.0;
  const uint32_t tlx = ka->bin.tiles_x, tly = ka->bin.tiles_y, chs = ka->bin.e_chan_shift;
  const double sx = ka->replay_scale_real, sy = ka->replay_scale_imag;
  const double ox = ka->replay_offset_real, oy = ka->replay_offset_imag;
#define CB_CHUNKED_OUTPUTS                                                                                       \
  [r] "+v"(p.r), [i] "+v"(p.i), [ps] "+v"(p_steps), [act] "+s"(act), [fill] "+s"(fill), [save] "=&s"(save),      \
      [alive] "=&s"(alive), [hx] "=&s"(hx), [hy] "=&s"(hy), [cs] "=&s"(cs), [ch] "=&s"(ch), [ctr] "=&s"(ctr),    \
      [t] "=&s"(t), [over] "=&s"(over), [a] "=&v"(a), [fx] "=&v"(fx), [fy] "=&v"(fy), [col] "=&v"(col),          \
      [row] "=&v"(row), [pidx] "=&v"(pidx), [e] "=&v"(e), [pl] "=&v"(pl), [grp] "=&v"(grp), [pos] "=&v"(pos),    \
      [lim] "=&v"(lim)
#define CB_CHUNKED_INPUTS                                                                                        \
  [n] "s"(n_steps), [cr] "v"(p.cr), [ci] "v"(p.ci), [sx] "s"(sx),                                             \
      [sy] "s"(sy), [wb] "s"(wb), [hb] "s"(hb), [base] "s"(region), [k16] "s"(k16), [rsh] "s"(row_shift),            \
      [tag] "v"(tag), [emit] "s"(emit), [tlx] "s"(tlx), [tly] "s"(tly), [chs] "s"(chs), [gcl] "s"(cursors_lds),  \
      [one] "v"(1u)
  const double rx = kPow2 ? 0.0 : ka->rcp_delta_real, ry = kPow2 ? 0.0 : ka->rcp_delta_imag;
  const double kg = 0.5 - 0x1p-24;
  const uint32_t all_steps = n_steps;
  lane_steps = 0;
  hits = 0;
  for (;;) {  // the burst, resumed behind every step that had to open chunks
  act = uniform_u64(act);  // (wave-uniform by construction; provably so for the "s" operands below)
  fill = __builtin_amdgcn_readfirstlane(fill);
  if (kPow2 && tagged) {
    asm volatile(CB_REPLAY_HEAD CB_REPLAY_BIN_POW2 CB_REPLAY_TAIL_HITS CB_REPLAY_GROUP_TAGGED CB_REPLAY_CHUNKED CB_REPLAY_TAIL2_CHUNKED
                 : CB_CHUNKED_OUTPUTS
                 : CB_CHUNKED_INPUTS, [ox] "v"(ox), [oy] "v"(oy)
                 : "vcc", "scc", "memory");
  } else if (kPow2) {
    asm volatile(CB_REPLAY_HEAD CB_REPLAY_BIN_POW2 CB_REPLAY_TAIL_HITS_PLAIN CB_REPLAY_GROUP_PLAIN CB_REPLAY_CHUNKED CB_REPLAY_TAIL2_CHUNKED
                 : CB_CHUNKED_OUTPUTS
                 : CB_CHUNKED_INPUTS, [ox] "v"(ox), [oy] "v"(oy)
                 : "vcc", "scc", "memory");
  } else if (tagged) {
    asm volatile(CB_REPLAY_HEAD CB_REPLAY_BIN_DIV CB_REPLAY_TAIL_HITS CB_REPLAY_GROUP_TAGGED CB_REPLAY_CHUNKED CB_REPLAY_TAIL2_CHUNKED
                 : CB_CHUNKED_OUTPUTS, [scp] "=&s"(scp), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
                 : CB_CHUNKED_INPUTS, [ox] "s"(ox), [oy] "s"(oy), [rx] "s"(rx), [ry] "s"(ry), [kg] "s"(kg)
                 : "vcc", "scc", "memory");
  } else {
    asm volatile(CB_REPLAY_HEAD CB_REPLAY_BIN_DIV CB_REPLAY_TAIL_HITS_PLAIN CB_REPLAY_GROUP_PLAIN CB_REPLAY_CHUNKED CB_REPLAY_TAIL2_CHUNKED
                 : CB_CHUNKED_OUTPUTS, [scp] "=&s"(scp), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
                 : CB_CHUNKED_INPUTS, [ox] "s"(ox), [oy] "s"(oy), [rx] "s"(rx), [ry] "s"(ry), [kg] "s"(kg)
                 : "vcc", "scc", "memory");
  }
  lane_steps += cs;
  hits += ch;
  if (over == 0ull) {  // the burst ran to its end: ctr steps were left when the last lane stopped (0: none did)
    if (burst_steps) *burst_steps = all_steps - (ctr != 0u ? ctr - 1u : 0u);
    break;
  }
  // The lanes of `over` found their group's chunk full (or the group has none yet): their word is in e, the
  // place they took in pos -- lim, lim + 1, ... in some order, one run per group.  Group by group: open the next
  // free chunk, put the words at its start and the group's cursor behind them.  (Once per kChunkWords words of a
  // group.)  The step itself is complete: ctr - 1 are left.
  n_steps = __builtin_amdgcn_readfirstlane(ctr - 1u);
  while (over != 0ull) {
    const int lane0 = __ffsll((long long) over) - 1;
    const uint32_t g0 = __builtin_amdgcn_readlane(grp, lane0);
    const uint32_t lim0 = __builtin_amdgcn_readlane(lim, lane0);
    const bool mine = lane_in(over) && grp == g0;
    const unsigned long long same = __ballot(mine);
    const uint32_t first = next_chunk * kChunkWords;
    if (mine) region[first + (pos - lim0)] = e;
    if (lane_id() == lane0) {
      desc[next_chunk] = (g0 << 16) | kChunkWords;  // taken as full; the launch's end corrects the last one of each group
      cursors[2u * g0] = first + (uint32_t) __popcll(same);
      cursors[2u * g0 + 1u] = first + kChunkWords;
    }
    next_chunk++;
    over &= ~same;
  }
  if (n_steps == 0u || act == 0ull) {
    if (burst_steps) *burst_steps = all_steps - n_steps;
    break;
  }
  }  // resumed
#undef CB_CHUNKED_OUTPUTS
#undef CB_CHUNKED_INPUTS
}

// ring index helpers for the 96-entry Q1 and the 192-entry Q2
__device__ __forceinline__ int q1_wrap(int slot) { return slot >= kQ1Cap ? slot - kQ1Cap : slot; }
//
__device__ __forceinline__ int q2_wrap(int slot) { return slot >= kQ2Cap ? slot - kQ2Cap : slot; }

template <bool kTimed, bool kBinned, bool kFastHead, bool kChunked>
__global__ void __launch_bounds__(64 * kWavesPerBlock, 4)  // four waves per SIMD: at most 128 vector registers
draw_wave_kernel(DrawArgs a) {
  static_assert(64 * kWavesPerBlock == kDrawBlockThreads, "draw_wave_count() assumes this block");
  static_assert(!kChunked || kBinned, "chunks are chunks of the stream");
  __shared__ WaveQueues queues[kWavesPerBlock];
  WaveQueues &q = queues[threadIdx.x >> 6];
  // kChunked: {next word, end of chunk} of every group of 1024 tiles (kernels.h, kChunkWords), 512 B per wave
  __shared__ uint32_t group_cursors[kChunked ? kWavesPerBlock : 1][kChunked ? 2 * kChunkedGroupsMax : 1];
  uint32_t *const my_cursors = group_cursors[kChunked ? (threadIdx.x >> 6) : 0];
  if (kChunked) {
    for (uint32_t k = threadIdx.x & 63u; k < 2u * kChunkedGroupsMax; k += 64u) my_cursors[k] = 0u;  // no chunk yet
  }
  const uint32_t cursors_lds = kChunked ? __builtin_amdgcn_readfirstlane(
      (uint32_t) reinterpret_cast<uintptr_t>(static_cast<void *>(my_cursors))) : 0u;
  uint32_t next_chunk = 0;  // kChunked: chunks of this wave's segment in use

  // kBinned: this wave's region of the pixel stream (kernels.h, BinLayout)
  const uint32_t wave_id =
      __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  uint32_t *const region = kBinned ? a.bin.stream + (size_t) wave_id * a.bin.cap : nullptr;
  const uint32_t region_cap = kBinned ? a.bin.cap : 0u;
  uint32_t region_fill = 0;

  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = tid < a.n_threads;
  const unsigned long long valid_mask = __ballot(valid);
  const Canvas cv = make_canvas(a);
  const int max_iter = a.max_iter;
  const int min_iter = a.min_iter;
  const int head_steps = a.head_steps;                          // HEAD runs iterations [0, head_steps)
  const int mid_steps = a.mid_steps;                            // MID runs [head_steps, long_start)
  const int long_start = head_steps + mid_steps;                // <= max_iter
  const int long_steps = max_iter - long_start;                 // iterations left to the LONG stage
  const int tail_steps = long_steps % kChunk;                   // an orbit's last, shorter chunk

  Xorwow rng = {0, 0, 0, 0, 0, 0};
  if (valid) rng = load_rng(a.states, a.n_threads, tid);

  // wave-uniform scheduler state and statistics (scalar registers)
  uint32_t samples_left = a.samples_per_thread;
  int q0_head = 0, q0_count = 0;
  int q1_head = 0, q1_count = 0;
  int q2_head = 0, q2_count = 0;
  unsigned long long n_rejected = 0, n_never = 0, n_too_fast = 0, n_recorded = 0, n_iterate = 0,
                     n_replay = 0, n_incr = 0, status = 0;
  unsigned long long t_head = 0, t_long = 0, t_replay = 0, dbg_chunks = 0, dbg_slots = 0;
  const unsigned long long t_start = kTimed ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rt_start = kTimed ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // wave slot on its SIMD (HW_REG_HW_ID bits 3:0) and chunks done, for the priority rotation
  const uint32_t wave_slot = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11));
  uint32_t long_chunks = 0;
  // HEAD as one asm block (head_loop): the usual stage split, where every escape inside
  // HEAD is too fast and survivors always have iterations left
  // (launch_draw_wave picks the instance; a kernel holds one of the two HEAD forms so that the
  // generator words have a single user and stay in place)
  constexpr bool fast_head = kFastHead;
  uint32_t rot = 0;  // rotation of the generator words, see CB_HEAD_DRAW_ANY_ROT
  // LDS byte addresses of this wave's Q0 / Q1 rings (the low half of a flat LDS address is the LDS offset)
  const uint32_t q0_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t) reinterpret_cast<uintptr_t>(static_cast<void *>(&q.q0_cr[0])));
  const uint32_t q1_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t) reinterpret_cast<uintptr_t>(static_cast<void *>(&q.q1_cr[0])));
  const uint32_t q2_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t) reinterpret_cast<uintptr_t>(static_cast<void *>(&q.q2_cr[0])));
  // LONG lane state: two orbits per lane (see CB_STEP2)
  Orbit lo[kOrbitsPerLane] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  double seen_r[kOrbitsPerLane] = {0, 0}, seen_i[kOrbitsPerLane] = {0, 0};  // periodicity check
  int l_rem[kOrbitsPerLane] = {0, 0};  // iterations left before max_iter; 0 = idle
  uint32_t skip_lo = 0, skip_hi = 0;  // per lane, 64 bits: iterations the periodicity check made unnecessary
  unsigned long long skipped_s = 0;   // ... and the interior map (mid_pass)
  uint32_t over = 0;                  // per lane: iterations the sparse LONG chunks counted beyond an escape
  // REPLAY lane state
  Orbit po = {0, 0, 0, 0};
  bool p_act = false;
  int p_steps = 0;
  // Fused channels (DrawArgs::n_channels > 0): an orbit popped from Q2 is first replayed WITHOUT
  // recording (p_real = false) -- the stages before know its escape index only to a chunk -- and, once
  // the index k is known, once more for every channel whose window holds k, with that channel's index
  // as the tag of its words (above row and col; the scatter sorts the planes as one canvas).
  const bool multi = a.n_channels > 0;
  bool p_real = true;
  uint32_t p_tag = 0u;
  // p_direct: this orbit's channels were known from the chunk it escaped in (q2_lrem), so its first pass already
  // records; what the measuring pass would have settled (the over-count of its last LONG chunk) is settled at
  // the end of that pass
  bool p_direct = false;
  // lanes of `finished` have just ended a replay pass: the measuring one (then the orbit's channels are
  // known) or a recorded one (then the next of its channels follows, if any -- windows may overlap)
  auto channel_decision = [&](unsigned long long finished) {
    const bool fin = lane_in(finished);
    uint32_t set = 0u;
    if (fin) {
      const int k = p_steps - 1;  // index of the escaping iteration (cudabrot.cu:336)
      for (int j = 0; j < a.n_channels; ++j) {
        if (k >= a.chan_min[j] && k < a.chan_max[j]) set |= 1u << j;
      }
      if (p_real) set &= ~((2u << (p_tag >> a.bin.e_chan_shift)) - 1u);  // the channels still to come
    }
    const bool measured = fin && !p_real;
    if ((measured || (fin && p_direct)) && a.sparse_long) over += long_overcount(p_steps, a.long_start, a.tail_start);
    if (fin) p_direct = false;
    n_recorded += (unsigned long long) __popcll(__ballot(measured && set != 0u));
    n_too_fast += (unsigned long long) __popcll(__ballot(measured && set == 0u));  // in no window (cudabrot.cu:407-408 for every channel)
    if (fin && set != 0u) {  // another pass: the same orbit from z = c, recorded into its next channel
      po.r = po.cr;
      po.i = po.ci;
      p_steps = 0;
      p_real = true;
      p_tag = (uint32_t) (__ffs((int) set) - 1) << a.bin.e_chan_shift;
      p_act = true;
    }
  };

  // Carry-over: pick up the queues and orbit slots the previous launch left behind (DrawArgs::carry).
  static_assert(sizeof(WaveQueues) == kCarryQueueWords * 8, "carry layout follows WaveQueues");
  unsigned long long *const carry =
      a.carry ? a.carry + (size_t) wave_id * kCarryWordsPerWave : nullptr;
  // Progress board (kernels.h, kSchedWords; only with a carry buffer).  VALU issue goes by priority,
  // then by wave age, and a wave alone on its SIMD cannot fill the fp64 pipe: left to themselves the four
  // waves of a SIMD end 25 % of the launch apart (measured), and the launch lasts as long as the slowest.
  // So the waves that share a SIMD post how many samples they still have to draw in a row of the board
  // and take s_setprio from their rank: the one furthest behind issues first.  A heuristic on top of a
  // result that no schedule can change; stale or torn values only cost time.
  uint32_t *board_row = nullptr;
  if (a.carry) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID
    const uint32_t key = ((xcc & 0xfu) << 12) | ((hw >> 4) & 0xfffu);                  // SIMD, pipe, CU, SH, SE
    board_row = reinterpret_cast<uint32_t *>(a.carry + (size_t) gridDim.x * kWavesPerBlock * kCarryWordsPerWave) +
                (size_t) key * 16u;
  }
  auto post_progress_and_set_priority = [&](uint32_t still_to_draw) {
    const uint32_t mine = still_to_draw + 1u;  // 0 = no wave in this slot
    const uint32_t lane = (uint32_t) lane_id();
    if (lane == 0u) {
      __hip_atomic_store(board_row + (wave_slot & 15u), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t other = 0u;
    if (lane < 16u) other = __hip_atomic_load(board_row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // waves of this SIMD that are further behind (ties: the lower slot first)
    const bool before_me = (lane < 16u) && (lane != (wave_slot & 15u)) &&
                           (other > mine || (other == mine && lane < (wave_slot & 15u)));
    const int rank = __popcll(__ballot(before_me));
    switch (rank) {
      case 0: __builtin_amdgcn_s_setprio(3); break;
      case 1: __builtin_amdgcn_s_setprio(2); break;
      case 2: __builtin_amdgcn_s_setprio(1); break;
      default: __builtin_amdgcn_s_setprio(0); break;
    }
  };
  // Wave-uniform conditions of the hot loops as scalars (one s_cmp each): as plain bools the compiler keeps them
  // as 64-bit lane masks, which live in spill slots and cost a pair of v_readlane per HEAD pass.
  // (0 / 1 words combined with integer operations, so that a test is one s_cmp of a scalar register)
  const uint32_t no_board = __builtin_amdgcn_readfirstlane(board_row != nullptr ? 0u : 1u);
  const uint32_t keep_rest = __builtin_amdgcn_readfirstlane((carry != nullptr && !a.drain) ? 0u : 1u);  // 0: leave in-flight work to the next launch
  const bool has_board = no_board == 0u;
  if (has_board) post_progress_and_set_priority(a.samples_per_thread);
  // (a record of draw_wide_kernel, tag 2 in both of the headers it lies over: the two kernels do not share carried work)
  if (carry && carry[0] != 0ull && carry[0] != 1ull) status |= CB_STATUS_CARRY_FOREIGN;
  if (carry && carry[0] == 1ull) {  // wave-uniform: the header is one address
    q0_head = (int) __builtin_amdgcn_readfirstlane((uint32_t) carry[1]);
    q0_count = (int) __builtin_amdgcn_readfirstlane((uint32_t) (carry[1] >> 32));
    q1_head = (int) __builtin_amdgcn_readfirstlane((uint32_t) carry[2]);
    q1_count = (int) __builtin_amdgcn_readfirstlane((uint32_t) (carry[2] >> 32));
    q2_head = (int) __builtin_amdgcn_readfirstlane((uint32_t) carry[3]);
    q2_count = (int) __builtin_amdgcn_readfirstlane((uint32_t) (carry[3] >> 32));
    unsigned long long *lds_words = reinterpret_cast<unsigned long long *>(&q);
    const unsigned long long *img = carry + kCarryHeaderWords;
    for (uint32_t k = lane_id(); k < kCarryQueueWords; k += 64u) lds_words[k] = img[k];
    const unsigned long long *pl = img + kCarryQueueWords + lane_id();
#pragma unroll
    for (int o = 0; o < kOrbitsPerLane; ++o) {
      lo[o].cr = __longlong_as_double((long long) pl[(o * 4 + 0) * 64]);
      lo[o].ci = __longlong_as_double((long long) pl[(o * 4 + 1) * 64]);
      lo[o].r = __longlong_as_double((long long) pl[(o * 4 + 2) * 64]);
      lo[o].i = __longlong_as_double((long long) pl[(o * 4 + 3) * 64]);
      seen_r[o] = __longlong_as_double((long long) pl[(8 + o * 2 + 0) * 64]);
      seen_i[o] = __longlong_as_double((long long) pl[(8 + o * 2 + 1) * 64]);
    }
    po.cr = __longlong_as_double((long long) pl[12 * 64]);
    po.ci = __longlong_as_double((long long) pl[13 * 64]);
    po.r = __longlong_as_double((long long) pl[14 * 64]);
    po.i = __longlong_as_double((long long) pl[15 * 64]);
    l_rem[0] = (int) (uint32_t) pl[16 * 64];
    l_rem[1] = (int) (uint32_t) (pl[16 * 64] >> 32);
    p_steps = (int) (uint32_t) pl[17 * 64];
    p_act = (pl[17 * 64] >> 32) != 0ull;
    p_tag = (uint32_t) pl[18 * 64];
    p_real = ((pl[18 * 64] >> 32) & 1ull) != 0ull;
    p_direct = ((pl[18 * 64] >> 33) & 1ull) != 0ull;
  }

  for (;;) {
    const bool input_done = (samples_left == 0);
    // with a carry buffer the in-flight work is left for the next launch instead of being drained
    if ((samples_left | keep_rest) == 0u) break;  // input done and the rest is left to the next launch
    const bool l_any = __ballot(l_rem[0] > 0 || l_rem[1] > 0) != 0ull;
    const bool draining = input_done && (q0_count == 0) && (q1_count == 0) && !l_any;
    const int n_replaying = __popcll(__ballot(p_act));

    // ---------------------------------------------------------------- REPLAY
    if ((q2_count > 0 && q2_count + n_replaying >= 64) ||
        (draining && (q2_count > 0 || n_replaying > 0))) {
      const unsigned long long t0 = kTimed ? __builtin_amdgcn_s_memtime() : 0ull;
      for (;;) {
        {  // refill idle lanes from Q2
          const unsigned long long idle_mask = __ballot(!p_act);
          const int n_idle = __popcll(idle_mask);
          const int n = n_idle < q2_count ? n_idle : q2_count;
          if (n > 0) {
            const int rank = mask_prefix(idle_mask);
            bool took_direct = false;
            if (!p_act && rank < n) {
              const int slot = q2_wrap(q2_head + rank);
              po.cr = q.q2_cr[slot];
              po.ci = q.q2_ci[slot];
              po.r = po.cr;
              po.i = po.ci;
              p_steps = 0;
              p_act = true;
              p_real = !multi;  // fused channels: first pass measures the escape index, nothing is recorded
              p_tag = 0u;
              p_direct = false;
              if (multi) {
                // ... unless the chunk the orbit escaped in lies inside one set of windows: then its channels are
                // known and the first pass records into the lowest of them (half the replay steps of such an orbit)
                const uint32_t lr = q.q2_lrem[slot];
                const int k_lo = max_iter - (int) lr, k_hi = k_lo + kChunk - 1;
                uint32_t set = 0u;
                bool known = lr != 0u;
                for (int j = 0; j < a.n_channels; ++j) {
                  const int lo_edge = a.chan_min[j], hi_edge = a.chan_max[j];
                  if ((lo_edge > k_lo && lo_edge <= k_hi) || (hi_edge > k_lo && hi_edge <= k_hi)) known = false;
                  if (k_lo >= lo_edge && k_lo < hi_edge) set |= 1u << j;
                }
                if (known && set != 0u) {  // (in no window: still measured -- the iteration count wants its index)
                  p_real = true;
                  p_direct = true;
                  took_direct = true;
                  p_tag = (uint32_t) (__ffs((int) set) - 1) << a.bin.e_chan_shift;
                }
              }
            }
            q2_head = q2_wrap(q2_head + n);
            q2_count -= n;
            if (!multi) n_recorded += (unsigned long long) n;
            if (multi) n_recorded += (unsigned long long) __popcll(__ballot(took_direct));  // (a measured orbit is counted when its index is known)
          }
        }
        const int n_act = __popcll(__ballot(p_act));
        if (n_act == 0) break;
        if (!draining && q2_count == 0 && n_act < kReplayMin) break;  // suspend

        // kBinned: the visited pixels go to this wave's stream region (compacted, coalesced stores)
        // in a hand-written burst; a full region falls back to the direct-atomics loop below, so
        // the result never depends on the workspace size.
        // (kChunked: a burst opens at most one chunk per group and one per kChunkWords words)
        const bool room = kChunked ? (next_chunk + a.bin.n_groups + 64u * kReplayBurst / kChunkWords + 1u <= a.bin.chunks_per_wave)
                                   : (region_fill + 64u * kReplayBurst <= region_cap);
        if (kBinned && room) {
          unsigned long long act_mask = __ballot(p_act);
          const unsigned long long was_act = act_mask;
          const unsigned long long emit = multi ? __ballot(p_real) : ~0ull;
          uint32_t steps = 0, hits = 0, burst = 0;
          if (kChunked) {
            uint32_t *const desc = a.bin.chunk_desc + (size_t) wave_id * a.bin.chunks_per_wave;
            if (cv.pow2_real && cv.pow2_imag) {
              replay_burst_chunked<true>(act_mask, kReplayBurst, po, p_steps, region, region_fill, steps, hits,
                                         a.bin.e_row_shift, p_tag, emit, multi, my_cursors, cursors_lds, next_chunk, desc,
                                         kDbgReplay ? &burst : nullptr);
            } else {
              replay_burst_chunked<false>(act_mask, kReplayBurst, po, p_steps, region, region_fill, steps, hits,
                                          a.bin.e_row_shift, p_tag, emit, multi, my_cursors, cursors_lds, next_chunk, desc,
                                          kDbgReplay ? &burst : nullptr);
            }
          } else if (cv.pow2_real && cv.pow2_imag) {
            replay_burst<true>(act_mask, kReplayBurst, po, p_steps, cv, region, region_fill, steps, hits,
                               a.bin.e_row_shift, p_tag, emit, multi, kDbgReplay ? &burst : nullptr);
          } else {
            replay_burst<false>(act_mask, kReplayBurst, po, p_steps, cv, region, region_fill, steps, hits,
                                a.bin.e_row_shift, p_tag, emit, multi, kDbgReplay ? &burst : nullptr);
          }
          if (kTimed && kDbgReplay) {  // the wave dump then describes REPLAY bursts instead of LONG chunks
            dbg_chunks += burst;
            dbg_slots += steps;
          }
          n_replay += steps;
          n_incr += hits;
          p_act = lane_in(act_mask);
          if (multi) {
            channel_decision(was_act & ~act_mask);
          } else if ((was_act & ~act_mask) != 0ull) {  // replays that ended: their escape index is known now
            const KernelArgs ra = fresh_args();
            if (ra->sparse_long && lane_in(was_act & ~act_mask)) over += long_overcount(p_steps, ra->long_start, ra->tail_start);
          }
          if (__ballot(p_act && p_steps > max_iter) != 0ull) {
            // cannot happen: the orbit escaped within max_iter steps in an earlier stage
            status |= CB_STATUS_REPLAY_RUNAWAY;
            if (p_steps > max_iter) p_act = false;
          }
          continue;
        }
        for (uint32_t b = 0; b < kReplayBurst; ++b) {
          const unsigned long long act_mask = __ballot(p_act);
          if (act_mask == 0ull) break;
          n_replay += (unsigned long long) __popcll(act_mask);
          bool done = false, hit = false;
          int row = 0, col = 0;
          if (p_act) {
#ifdef CB_BURNING_SHIP
            const double m4 = mandel_step2_ship(po.cr, po.ci, po.r, po.i);  // cudabrot.cu:353-359
#else
            const double m4 = mandel_step2(po.cr, po.ci, po.r, po.i);  // cudabrot.cu:357-359
#endif
            hit = p_real && pixel_of(0.5 * po.r, 0.5 * po.i, cv, row, col);  // cudabrot.cu:308-311 (halving is exact)
            if (hit) {                                                 // cudabrot.cu:312
              if (!multi) {
                add_to_pixel(a.hist, cv, row, col, 1ull);
              } else {  // the plane of the channel this pass records into
                add_to_pixel(a.hist + (unsigned long long) (p_tag >> a.bin.e_chan_shift) * a.plane_pixels, cv, row,
                             col, 1ull);
              }
            }
            p_steps++;
            done = m4 > 16.0;                                          // cudabrot.cu:363
            if (!done && p_steps > max_iter) {
              status |= CB_STATUS_REPLAY_RUNAWAY;
              done = true;
            }
            if (done) {
              p_act = false;
              if (!multi && a.sparse_long) over += long_overcount(p_steps, a.long_start, a.tail_start);
            }
          }
          n_incr += (unsigned long long) __popcll(__ballot(hit));
          const unsigned long long done_mask = __ballot(done);
          if (multi && done_mask != 0ull) channel_decision(done_mask);
          if (done_mask != 0ull && q2_count > 0) break;
        }
      }
      if (kTimed) t_replay += __builtin_amdgcn_s_memtime() - t0;
      if (draining) break;
      continue;
    }
    if (draining) break;

    // HEAD and MID feed the queues in a loop of their own, in the same order of precedence as the
    // outer loop (REPLAY > HEAD > MID > LONG): they run far more often than the other stages, and
    // inside this loop only the generator and the queue counters are live-modified, so its back
    // edge is light (the outer loop's carries every orbit register).
    bool replay_ready = false;
    // the statistics of the asm HEAD / MID passes: 32 bits while the loop runs (one scalar each instead of a
    // 64-bit pair that lives in spill slots), folded into the launch's counters behind it
    uint32_t f_rejected = 0, f_too_fast = 0, f_steps = 0;
    for (;;) {
    if (q2_count > 0 && q2_count + n_replaying >= 64) {
      replay_ready = true;
      break;
    }
    const bool feed_input_done = (samples_left == 0);
    if ((samples_left | keep_rest) == 0u) break;  // the rest is left in the queues for the next launch
    // ---------------------------------------------------------------- HEAD
    if (!feed_input_done && q0_count < 64) {
      const unsigned long long t0 = kTimed ? __builtin_amdgcn_s_memtime() : 0ull;
      if constexpr (fast_head) {  // the usual split (plan_stages): HEAD passes in a row as one asm statement
        if ((((samples_left - 1u) & 31u) | no_board) == 0u) post_progress_and_set_priority(samples_left - 1u);
        uint32_t count = (uint32_t) q0_count;
        head_loop(rng, samples_left, rot, valid_mask, (uint32_t) (q0_head + q0_count), count, q0_lds, f_rejected,
                  f_too_fast, f_steps);  // survivors -> Q0
        q0_count = (int) count;
        if (q0_count > kQ0Cap) status |= CB_STATUS_QUEUE_OVERFLOW;
        if (kTimed) t_head += __builtin_amdgcn_s_memtime() - t0;
        continue;
      }
      samples_left--;
      if (((samples_left & 31u) | no_board) == 0u) post_progress_and_set_priority(samples_left);
      bool alive = false;
      Orbit o = {0, 0, 0, 0};
      if (valid) {
        o.cr = sample_coordinate2(rng);  // cudabrot.cu:392 (doubled, like everything below)
        o.ci = sample_coordinate2(rng);  // cudabrot.cu:393
#ifdef CB_BURNING_SHIP
        alive = true;  // cudabrot.cu:397-399: no shortcut in this variant
#else
        alive = !(in_main_cardioid2(o.cr, o.ci) || in_order2_bulb2(o.cr, o.ci));  // cudabrot.cu:398
#endif
      }
      o.r = o.cr;
      o.i = o.ci;
      unsigned long long alive_mask = __ballot(alive);
      n_rejected += (unsigned long long) __popcll(valid_mask & ~alive_mask);
      const unsigned long long accept_mask =
          iterate_window(alive_mask, 0, head_steps, min_iter, o, n_iterate, n_too_fast);
      if (accept_mask != 0ull) {
        if (lane_in(accept_mask)) {
          const int slot = q2_wrap(q2_wrap(q2_head + q2_count) + mask_prefix(accept_mask));
          q.q2_cr[slot] = o.cr;
          q.q2_ci[slot] = o.ci;
          q.q2_lrem[slot] = 0u;
        }
        q2_count += __popcll(accept_mask);
        if (q2_count > kQ2Cap) status |= CB_STATUS_QUEUE_OVERFLOW;
      }
      if (alive_mask != 0ull) {  // survivors of the head
        if (max_iter > head_steps) {
          if (lane_in(alive_mask)) {
            const int slot = (q0_head + q0_count + mask_prefix(alive_mask)) & (kQ0Cap - 1);
            q.q0_cr[slot] = o.cr;
            q.q0_ci[slot] = o.ci;
          }
          q0_count += __popcll(alive_mask);
          if (q0_count > kQ0Cap) status |= CB_STATUS_QUEUE_OVERFLOW;
        } else {
          n_never += (unsigned long long) __popcll(alive_mask);  // head_steps == max_iter
        }
      }
      if (kTimed) t_head += __builtin_amdgcn_s_memtime() - t0;
      continue;
    }

    // ---------------------------------------------------------------- MID
    if (q0_count > 0 && q1_count < kQ1Low && (q0_count >= 64 || feed_input_done)) {
      const unsigned long long t0 = kTimed ? __builtin_amdgcn_s_memtime() : 0ull;
      const int n = q0_count < 64 ? q0_count : 64;
      // MID as one asm block (mid_pass) under the usual split: HEAD did kHeadSteps iterations, every escape inside
      // MID is too fast (the stage ends at or before min_iter) and survivors have iterations left
      const KernelArgs ma = fresh_args();
      if (kFastHead && ma->fast_mid) {
        const unsigned long long take = (n == 64) ? ~0ull : ((1ull << n) - 1ull);
        unsigned long long alive, hit;
        uint32_t steps;
#ifdef CB_BURNING_SHIP
        const unsigned long long map = 0ull;  // (the map is the Mandelbrot set's)
#else
        const unsigned long long map = reinterpret_cast<unsigned long long>(ma->interior_map);
#endif
        mid_pass(take, (uint32_t) q0_head + (uint32_t) lane_id(), q0_lds, (uint32_t) ma->mid_steps,
                 (uint32_t) q1_wrap(q1_head + q1_count), q1_lds, alive, steps, map, ma->interior_shift,
                 ma->interior_cols, ma->interior_rows, hit);
        alive = uniform_u64(alive);
        hit = uniform_u64(hit);
        q0_head = (q0_head + n) & (kQ0Cap - 1);
        q0_count -= n;
        f_steps += steps;
        f_too_fast += (uint32_t) __popcll(take & ~alive);  // escaped before min_iter
        if (hit != 0ull) {  // samples of cells proven never-escaping: the reference iterates them to max_iter (cudabrot.cu:339)
          const uint32_t n_hit = (uint32_t) __popcll(hit & alive);
          n_never += n_hit;
          skipped_s += (unsigned long long) n_hit * (unsigned long long) (uint32_t) (ma->max_iter - ma->long_start);
          if ((hit & ~alive) != 0ull) status |= CB_STATUS_INTERIOR_MAP;
          alive &= ~hit;
        }
        q1_count += __popcll(alive);
        if (q1_count > kQ1Cap) status |= CB_STATUS_QUEUE_OVERFLOW;
        if (kTimed) t_head += __builtin_amdgcn_s_memtime() - t0;
        continue;
      }
      const bool mine = lane_id() < n;
      Orbit o = {0, 0, 0, 0};
      if (mine) {
        const int slot = (q0_head + lane_id()) & (kQ0Cap - 1);
        o.cr = q.q0_cr[slot];
        o.ci = q.q0_ci[slot];
      }
      o.r = o.cr;
      o.i = o.ci;
      q0_head = (q0_head + n) & (kQ0Cap - 1);
      q0_count -= n;
      unsigned long long alive_mask = __ballot(mine);
      {  // re-derive z after the head iterations (Q0 keeps only c); none of these can escape again
        uint32_t ignored = 0;
        (void) iterate_steps(alive_mask, (uint32_t) head_steps, o, ignored);
      }
      const unsigned long long accept_mask =
          iterate_window(alive_mask, head_steps, mid_steps, min_iter, o, n_iterate, n_too_fast);
      if (accept_mask != 0ull) {
        if (lane_in(accept_mask)) {
          const int slot = q2_wrap(q2_wrap(q2_head + q2_count) + mask_prefix(accept_mask));
          q.q2_cr[slot] = o.cr;
          q.q2_ci[slot] = o.ci;
          q.q2_lrem[slot] = 0u;
        }
        q2_count += __popcll(accept_mask);
        if (q2_count > kQ2Cap) status |= CB_STATUS_QUEUE_OVERFLOW;
      }
      if (alive_mask != 0ull) {  // survivors of the mid stage
        if (long_steps > 0) {
          if (lane_in(alive_mask)) {
            const int slot = q1_wrap(q1_wrap(q1_head + q1_count) + mask_prefix(alive_mask));
            q.q1_cr[slot] = o.cr;
            q.q1_ci[slot] = o.ci;
            q.q1_r[slot] = o.r;
            q.q1_i[slot] = o.i;
          }
          q1_count += __popcll(alive_mask);
          if (q1_count > kQ1Cap) status |= CB_STATUS_QUEUE_OVERFLOW;
        } else {
          n_never += (unsigned long long) __popcll(alive_mask);  // long_start == max_iter
        }
      }
      if (kTimed) t_head += __builtin_amdgcn_s_memtime() - t0;
      continue;
    }
    break;
    }  // feed loop
    n_rejected += f_rejected;
    n_too_fast += f_too_fast;
    n_iterate += f_steps;
    if (replay_ready) continue;
    if ((samples_left | keep_rest) == 0u) continue;  // leaves at the top: nothing drawn is lost

    // ---------------------------------------------------------------- LONG
    const unsigned long long t0 = kTimed ? __builtin_amdgcn_s_memtime() : 0ull;
    // the stage's constants: scalar loads from the argument segment on entry (fresh_args)
    const KernelArgs la = fresh_args();
    const uint32_t long_steps_u = la->long_steps, tail_value = la->tail_value;
    const uint32_t check_flag = (uint32_t) la->check_periodic;
    const int accept_rem = la->accept_rem;
    uint32_t l_orbit_chunks = 0, l_too_fast = 0, l_never = 0;  // as in the feed loop: 32 bits inside the stage
    for (;;) {
      // Even progress for the waves of a SIMD.  VALU issue goes by priority, then by wave age, so
      // with equal priorities the oldest wave races ahead and the youngest is left to finish alone,
      // where one wave cannot fill the fp64 pipe (measured: the 4 waves of a SIMD ended at 31 / 40 /
      // 53 / 65 ms of a 65 ms kernel).  Each wave therefore walks through the four priority levels
      // as it progresses, offset by its wave slot.
      if (((long_chunks & (kPrioChunks - 1u)) | no_board) == 0u) {
        post_progress_and_set_priority(samples_left);
      } else if (has_board) {
      } else if ((long_chunks & (kPrioChunks - 1u)) == 0u) {
        switch ((wave_slot + long_chunks / kPrioChunks) & 3u) {
          case 0: __builtin_amdgcn_s_setprio(0); break;
          case 1: __builtin_amdgcn_s_setprio(1); break;
          case 2: __builtin_amdgcn_s_setprio(2); break;
          default: __builtin_amdgcn_s_setprio(3); break;
        }
      }
      ++long_chunks;
      unsigned long long full_mask[kOrbitsPerLane], tail_mask[kOrbitsPerLane];
#pragma unroll
      for (int o = 0; o < kOrbitsPerLane; ++o) {  // refill idle orbit slots from Q1 (asm: long_refill)
        uint32_t taken = 0;
        long_refill(lo[o], seen_r[o], seen_i[o], l_rem[o], __builtin_amdgcn_readfirstlane((uint32_t) q1_head),
                    __builtin_amdgcn_readfirstlane((uint32_t) q1_count), q1_lds, long_steps_u, tail_value, taken,
                    full_mask[o], tail_mask[o]);
        q1_head = q1_wrap(q1_head + (int) taken);
        q1_count -= (int) taken;
      }
      if ((full_mask[0] | full_mask[1] | tail_mask[0] | tail_mask[1]) == 0ull) break;

      uint32_t steps = 0;
      if ((tail_mask[0] | tail_mask[1]) != 0ull) {
#pragma unroll
        for (int o = 0; o < kOrbitsPerLane; ++o) {
          if (tail_mask[o] != 0ull) {  // last, shorter chunk of these orbits: exactly tail_steps iterations
            const unsigned long long esc_t = iterate_steps(tail_mask[o], (uint32_t) tail_steps, lo[o], steps);
            n_iterate += steps;
            n_never += (unsigned long long) __popcll(tail_mask[o] & ~esc_t);  // reached max_iter (cudabrot.cu:339)
            // the tail chunk lies on one side of min_iter like every chunk (cudabrot.cu:407-408)
            const bool push = lane_in(esc_t) && (max_iter - l_rem[o] >= min_iter);
            const unsigned long long push_mask = __ballot(push);
            n_too_fast += (unsigned long long) __popcll(esc_t & ~push_mask);
            if (lane_in(tail_mask[o])) l_rem[o] = 0;
            if (push_mask != 0ull) {
              if (push) {
                const int slot = q2_wrap(q2_wrap(q2_head + q2_count) + mask_prefix(push_mask));
                q.q2_cr[slot] = lo[o].cr;
                q.q2_ci[slot] = lo[o].ci;
                q.q2_lrem[slot] = 0u;
              }
              q2_count += __popcll(push_mask);
              if (q2_count > kQ2Cap) status |= CB_STATUS_QUEUE_OVERFLOW;
            }
          }
        }
      }
      if ((full_mask[0] | full_mask[1]) != 0ull) {
        unsigned long long esc[kOrbitsPerLane];
        if (la->sparse_long) {  // the escape test on every tenth step (iterate_chunk2_sparse)
          unsigned long long sure[kOrbitsPerLane];
          iterate_chunk2_sparse(full_mask[0], full_mask[1], lo[0], lo[1], esc[0], esc[1], sure[0], sure[1],
                                la->sparse_threshold);
          l_orbit_chunks += (uint32_t) (__popcll(full_mask[0]) + __popcll(full_mask[1]));
          if (((esc[0] & ~sure[0]) | (esc[1] & ~sure[1])) != 0ull) {  // one chunk in ~10^7
#pragma unroll
            for (int o = 0; o < kOrbitsPerLane; ++o) {
              const unsigned long long doubt = esc[o] & ~sure[o];
              if (doubt != 0ull) {
                const unsigned long long really = verify_chunk_escape(doubt, lo[o], max_iter - l_rem[o]);
                esc[o] = (esc[o] & ~doubt) | (really & doubt);
              }
            }
          }
        } else {
          iterate_chunk2(full_mask[0], full_mask[1], lo[0], lo[1], esc[0], esc[1], steps);
          n_iterate += steps;
        }
        if (kTimed && !kDbgReplay) {
          dbg_chunks++;
          dbg_slots += (unsigned long long) (__popcll(full_mask[0]) + __popcll(full_mask[1]));
        }
        // Bookkeeping (asm: long_retire).  An orbit's chunk covered escape indices [k_lo, k_lo + 32)
        // with k_lo = max_iter - l_rem; min_iter - long_start is a multiple of kChunk, so the whole chunk
        // is on one side of min_iter (cudabrot.cu:407-408): escapes with l_rem <= max_iter - min_iter are
        // accepted.  Exact-periodicity early-out: if z is bit for bit a value this
        // orbit held at an earlier chunk boundary, the (deterministic) orbit repeats that stretch for
        // ever and every point of it passed the escape test: the sample can never escape, so
        // IterateMandelbrot would return max_iterations -- the same outcome, without executing the
        // remaining iterations.  Brent's scheme at chunk granularity: compare with one saved point,
        // re-save on a geometric schedule of chunk counts (long_retire); a cycle of period p is found
        // p / gcd(p, kChunk) chunks after a save that lies on it (hence kChunk = 30, kernels.h).
#pragma unroll
        for (int o = 0; o < kOrbitsPerLane; ++o) {
          unsigned long long push = 0ull, ended = 0ull, periodic = 0ull;
          const uint32_t q2_tail = __builtin_amdgcn_readfirstlane((uint32_t) q2_wrap(q2_head + q2_count));
          long_retire(lo[o], seen_r[o], seen_i[o], l_rem[o], skip_lo, skip_hi, uniform_u64(full_mask[o]),
                      uniform_u64(esc[o]), accept_rem, long_steps_u, check_flag, q2_tail, q2_lds, push, ended,
                      periodic);
          l_too_fast += (uint32_t) __popcll(esc[o] & ~push);
          q2_count += __popcll(push);
          if (q2_count > kQ2Cap) status |= CB_STATUS_QUEUE_OVERFLOW;
          l_never += (uint32_t) (__popcll(ended) + __popcll(periodic));
        }
      }
      // leave the stage when another one has work to do
      if (q2_count + __popcll(__ballot(p_act)) >= 64) break;                       // REPLAY can fill every lane
      if (q1_count < kQ1Exit && (samples_left != 0 || q0_count > 0)) break;       // HEAD / MID must top up
      if (l_orbit_chunks > (1u << 30)) break;                                      // (fold the 32-bit statistics; the stage is re-entered)
    }
    n_iterate += (unsigned long long) kChunk * l_orbit_chunks;
    n_too_fast += l_too_fast;
    n_never += l_never;
    if (kTimed) t_long += __builtin_amdgcn_s_memtime() - t0;
  }

  if (board_row && lane_id() == 0) {  // this wave no longer competes
    __hip_atomic_store(board_row + (wave_slot & 15u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  switch (rot) {  // back to the logical order of the generator words
    case 1: rng = xorwow_unrotated<1>(rng); break;
    case 2: rng = xorwow_unrotated<2>(rng); break;
    case 3: rng = xorwow_unrotated<3>(rng); break;
    case 4: rng = xorwow_unrotated<4>(rng); break;
    default: break;
  }
  if (valid) store_rng(a.states, a.n_threads, tid, rng);
  if (kBinned && !kChunked && lane_id() == 0) a.bin.wave_count[wave_id] = region_fill;
  if (kChunked) {
    // the last chunk of every group holds what its cursor says; every other chunk in use is full
    uint32_t *const desc = a.bin.chunk_desc + (size_t) wave_id * a.bin.chunks_per_wave;
    const uint32_t g = (uint32_t) lane_id();
    if (g < a.bin.n_groups) {
      const uint32_t pos = my_cursors[2u * g], lim = my_cursors[2u * g + 1u];
      if (lim != 0u) desc[lim / kChunkWords - 1u] = (g << 16) | (pos - (lim - kChunkWords));
    }
    if (lane_id() == 0) a.bin.wave_count[wave_id] = next_chunk;
  }
  if (carry) {  // leave queues and orbit slots for the next launch (empty after a drain)
    if (lane_id() == 0) {
      carry[0] = 1ull;
      carry[1] = (unsigned long long) (uint32_t) q0_head | ((unsigned long long) (uint32_t) q0_count << 32);
      carry[2] = (unsigned long long) (uint32_t) q1_head | ((unsigned long long) (uint32_t) q1_count << 32);
      carry[3] = (unsigned long long) (uint32_t) q2_head | ((unsigned long long) (uint32_t) q2_count << 32);
    }
    const unsigned long long *lds_words = reinterpret_cast<const unsigned long long *>(&q);
    unsigned long long *img = carry + kCarryHeaderWords;
    for (uint32_t k = lane_id(); k < kCarryQueueWords; k += 64u) img[k] = lds_words[k];
    unsigned long long *pl = img + kCarryQueueWords + lane_id();
#pragma unroll
    for (int o = 0; o < kOrbitsPerLane; ++o) {
      pl[(o * 4 + 0) * 64] = (unsigned long long) __double_as_longlong(lo[o].cr);
      pl[(o * 4 + 1) * 64] = (unsigned long long) __double_as_longlong(lo[o].ci);
      pl[(o * 4 + 2) * 64] = (unsigned long long) __double_as_longlong(lo[o].r);
      pl[(o * 4 + 3) * 64] = (unsigned long long) __double_as_longlong(lo[o].i);
      pl[(8 + o * 2 + 0) * 64] = (unsigned long long) __double_as_longlong(seen_r[o]);
      pl[(8 + o * 2 + 1) * 64] = (unsigned long long) __double_as_longlong(seen_i[o]);
    }
    pl[12 * 64] = (unsigned long long) __double_as_longlong(po.cr);
    pl[13 * 64] = (unsigned long long) __double_as_longlong(po.ci);
    pl[14 * 64] = (unsigned long long) __double_as_longlong(po.r);
    pl[15 * 64] = (unsigned long long) __double_as_longlong(po.i);
    pl[16 * 64] = (unsigned long long) (uint32_t) l_rem[0] | ((unsigned long long) (uint32_t) l_rem[1] << 32);
    pl[17 * 64] = (unsigned long long) (uint32_t) p_steps | ((unsigned long long) (p_act ? 1u : 0u) << 32);
    pl[18 * 64] = (unsigned long long) p_tag | ((unsigned long long) (p_real ? 1u : 0u) << 32) |
                  ((unsigned long long) (p_direct ? 1u : 0u) << 33);
  }
  const unsigned long long skipped_total = wave_sum(((unsigned long long) skip_hi << 32) | skip_lo) + skipped_s;
  const unsigned long long over_total = wave_sum((unsigned long long) over);
  if (a.counters && lane_id() == 0) {
    unsigned long long *c = reinterpret_cast<unsigned long long *>(a.counters);
    const unsigned long long n_samples =
        (unsigned long long) __popcll(valid_mask) * (unsigned long long) a.samples_per_thread;
    const unsigned long long v[9] = {n_samples, n_rejected, n_never,  n_too_fast, n_recorded,
                                     n_iterate + skipped_total - over_total, n_replay, n_incr, skipped_total};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (v[k]) __hip_atomic_fetch_add(c + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (status) __hip_atomic_fetch_or(c + 9, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (kTimed) {
      const unsigned long long t_all = __builtin_amdgcn_s_memtime() - t_start;
      __hip_atomic_fetch_add(c + 10, t_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(c + 11, t_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(c + 12, t_replay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(c + 13, t_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // wave residency on the 100 MHz constant clock: first start (kept as max of ~start), last end,
      // and the sum of wave lifetimes
      const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
      __hip_atomic_fetch_max(c + 14, ~rt_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_max(c + 15, rt_end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(c + 16, rt_end - rt_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.wave_dump) {
        unsigned long long *d = a.wave_dump + (size_t) wave_id * 8;
        d[0] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, all bits
        // XCC id (4 bits) | full LONG chunks run (28 bits) | orbit slots active in them (32 bits)
        d[1] = (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xfu) |
               ((dbg_chunks & 0xfffffffull) << 4) | (dbg_slots << 32);
        d[2] = rt_start;
        d[3] = rt_end;
        d[4] = t_head;
        d[5] = t_long;
        d[6] = t_replay;
        d[7] = t_all;
      }
    }
  }
}

}  // namespace

hipError_t CB_LAUNCH_NAME(const DrawArgs &a, bool timed, hipStream_t stream) {
  const bool drain_launch = a.carry != nullptr && a.drain != 0;
  if (a.n_threads == 0 || (a.samples_per_thread == 0 && !drain_launch)) return hipSuccess;
  const uint32_t threads = 64 * kWavesPerBlock;
  const uint32_t blocks = (a.n_threads + threads - 1u) / threads;
  const bool binned = a.bin.enabled != 0u;
  if (binned && a.bin.n_waves != blocks * kWavesPerBlock) return hipErrorInvalidValue;
  // HEAD as one asm block needs the usual stage split: every escape inside HEAD is too fast and
  // survivors always have iterations left
  const bool fast = (a.head_steps == kHeadSteps) && (a.min_iter >= kHeadSteps) && (a.max_iter > kHeadSteps);
  const dim3 grid(blocks), block(threads);
  // beyond 1024 tiles: the stream chunked by group of tiles as it is written (BinLayout::chunked)
  const bool chunked = binned && a.bin.chunked != 0u;
#define CB_LAUNCH(T, B, F, C) hipLaunchKernelGGL((draw_wave_kernel<T, B, F, C>), grid, block, 0, stream, a)
  if (timed) {
    if (binned) {
      if (chunked) {
        if (fast) CB_LAUNCH(true, true, true, true); else CB_LAUNCH(true, true, false, true);
      } else {
        if (fast) CB_LAUNCH(true, true, true, false); else CB_LAUNCH(true, true, false, false);
      }
    } else {
      if (fast) CB_LAUNCH(true, false, true, false); else CB_LAUNCH(true, false, false, false);
    }
  } else {
    if (binned) {
      if (chunked) {
        if (fast) CB_LAUNCH(false, true, true, true); else CB_LAUNCH(false, true, false, true);
      } else {
        if (fast) CB_LAUNCH(false, true, true, false); else CB_LAUNCH(false, true, false, false);
      }
    } else {
      if (fast) CB_LAUNCH(false, false, true, false); else CB_LAUNCH(false, false, false, false);
    }
  }
#undef CB_LAUNCH
  return hipGetLastError();
}

#ifndef CB_BURNING_SHIP
// Stage split.  HEAD does iterations [0, head), MID [head, head + mid), LONG the rest in chunks of
// kChunk.  mid is chosen so that min_iter - (head + mid) is a multiple of kChunk whenever min_iter
// lies beyond the MID stage: then no LONG chunk straddles min_iter.
void plan_stages(int max_iter, int min_iter, int *head_steps, int *mid_steps) {
  const int kHead = kHeadSteps, kMidMin = 12;
  int head = kHead;
  if (max_iter < head) head = max_iter < 0 ? 0 : max_iter;
  int rem = max_iter - head;
  int mid = 0;
  if (rem > 0) {
    if (min_iter <= head + kMidMin) {
      mid = 16;
    } else {
      mid = kMidMin + ((min_iter - head - kMidMin) % kChunk);
    }
    if (mid > rem) mid = rem;  // shallow runs: HEAD + MID do all of it
  }
  *head_steps = head;
  *mid_steps = mid;
}

#endif

}  // namespace cb
