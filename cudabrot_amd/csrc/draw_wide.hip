// draw_wide.hip -- draw_wide_kernel: the product path of DrawBuddhabrot (cudabrot.cu:379-414) for the usual
// configuration, shaped so that it fills the fp64 pipe from TWO waves per SIMD and leaves the other half of every
// CU (256 vector registers per SIMD lane, ~80 KiB of LDS) to the scatter kernels of the previous launch.
//
// Why.  The draw kernel is bound by vector issue, the scatter (scatter.hip) by memory; run one after the other a
// step costs their sum.  draw_wave_kernel (draw_wave.hip) needs four waves per SIMD of 128 registers each to hide
// the latency of its dependent fp64 chains, which fills the register file: nothing else can live on the CU.  This
// kernel keeps the same four stages over the same three wave-private queues (read draw_wave.hip's header first)
// but gets its latency hiding from instruction-level parallelism inside a wave instead of from more waves:
//
//   a wave owns 128 subsequences (two generators per lane, A and B), so the grid is 2048 waves = 2 per SIMD;
//   HEAD    software-pipelined: the draw of sample n+1 (integer chain) is interleaved, instruction by instruction,
//           with the cardioid / bulb test and the four iterations of sample n (fp64 chain); A and B alternate.
//           EXEC stays full (the generator must advance in every lane), the survivors are tracked in scalar masks;
//   MID     as in draw_wave.hip (one chain; 5 % of the instructions) -- and the place where a survivor of HEAD is
//           looked up in the interior map (DrawArgs::interior_map, tools/interior_map.c): a sample of a cell PROVEN
//           never-escaping is retired here instead of costing the LONG stage ~2000 iterations;
//   LONG    FOUR orbits per lane, interleaved instruction by instruction (escape test once per chunk);
//   REPLAY  software-pipelined: the pixel of point n is formed and stored while point n+1 is computed; a lane's
//           state between bursts is "z_n computed, not yet recorded".
//
// Launched for: a scatter workspace of one level (canvases of up to 1024 tiles) or a chunked one, one channel, the usual stage
// split with min_iter == the start of the LONG stage (every BASELINE config with the default -c 20), a carry
// buffer, n_threads a multiple of 512.  Everything else is draw_wave_kernel's (capi.hip picks; same results).
//
// Compiled with -ffp-contract=off; all coordinates DOUBLED (C = 2c, Z = 2z) as in draw_wave.hip; the arithmetic of
// every step is device_math.h's mandel_step2, instruction for instruction.
#include <stdlib.h>

#include "draw_common.h"

// Compiled twice like draw_wave.hip: as is, and with -DCB_BURNING_SHIP (cudabrot.cu:15-17).
#ifdef CB_BURNING_SHIP
#define CB_AL "|"
#define CB_AR "|"
#define CB_LAUNCH_NAME launch_draw_wide_ship
#else
#define CB_AL ""
#define CB_AR ""
#define CB_LAUNCH_NAME launch_draw_wide
#endif

namespace cb {

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kSlots = 4;              // deep orbits a lane iterates side by side in the LONG stage
constexpr int kQ0Cap = 128;            // HEAD survivors: c            (2 KiB per wave)
constexpr int kQ1Cap = 96;             // MID survivors: (c, z)        (3 KiB per wave)
constexpr int kQ2Cap = 320;            // accepted starting points: c  (5 KiB per wave)
constexpr int kHeadSteps = 4;
#ifndef CB_WQ1_LOW
#define CB_WQ1_LOW 32
#endif
#ifndef CB_WQ1_EXIT
#define CB_WQ1_EXIT 8
#endif
#ifndef CB_WREPLAY_MIN
#define CB_WREPLAY_MIN 56
#endif
#ifndef CB_WREPLAY_BURST
#define CB_WREPLAY_BURST 32
#endif
constexpr int kQ1Low = CB_WQ1_LOW;      // run MID while fewer deep orbits than this are queued
constexpr int kReplayMin = CB_WREPLAY_MIN;  // suspend REPLAY below this many busy lanes (unless draining)
constexpr uint32_t kReplayBurst = CB_WREPLAY_BURST;  // replay steps per asm burst
#ifndef CB_WIDE_PRIO_BEHIND
#define CB_WIDE_PRIO_BEHIND 2  // s_setprio of the wave of a SIMD that has more left to draw ...
#endif
#ifndef CB_WIDE_PRIO_AHEAD
#define CB_WIDE_PRIO_AHEAD 1   // ... and of the other
#endif
constexpr uint32_t kBrentBits = 2;      // periodicity check: re-save when the chunk count has no bits below its top 2
constexpr uint32_t kPrioChunks = 64;    // LONG chunks between two looks at the progress board (power of two)
constexpr uint32_t kPrioHalves = 256;   // HEAD half-passes between two looks (power of two; a look costs a round trip to L2)

// Ring capacities are exact worst cases (CB_STATUS_QUEUE_OVERFLOW guards the reasoning, these the constants):
//   Q0: a HEAD half-pass runs while q0_count < 64 and pushes at most 64             -> 63 + 64
//   Q1: MID runs while q1_count < kQ1Low and pushes at most 64                      -> kQ1Low - 1 + 64
//   Q2: LONG runs while q2_count + replaying < 64; one chunk can retire every slot  -> 63 + 64 * kSlots
static_assert(kQ0Cap >= 63 + 64, "Q0 must hold a full HEAD half-pass on top of 63 queued survivors");
static_assert(kQ1Cap >= kQ1Low - 1 + 64, "Q1 must hold a full MID pass on top of kQ1Low - 1 queued orbits");
static_assert(kQ2Cap >= 63 + 64 * kSlots, "Q2 must hold every orbit slot of a chunk on top of 63 queued points");

struct WideQueues {
  double q0_cr[kQ0Cap], q0_ci[kQ0Cap];
  double q1_cr[kQ1Cap], q1_ci[kQ1Cap], q1_r[kQ1Cap], q1_i[kQ1Cap];
  double q2_cr[kQ2Cap], q2_ci[kQ2Cap];
};
constexpr uint32_t kWideQueueWords = sizeof(WideQueues) / 8;
// carry record of a wave: header, the queues, the lanes' planes (u64 each):
//   LONG kSlots x {cr, ci, r, i, seen_r, seen_i}, l_rem[0..1], l_rem[2..3], REPLAY {cr, ci, r, i}, {p_start, p_act}
constexpr uint32_t kWideLanePlanes = kSlots * 6 + 2 + 4 + 1;
// The record lies over the two records of the waves this one replaces (draw_wave.hip's: kCarryWordsPerWave words each,
// tag 1 in word 0).  Both headers carry THIS kernel's tag, 2 -- the second record's header words are skipped by the lane
// planes -- so that either kernel, handed a carry buffer the other one wrote, sees a foreign tag in every record it
// reads and reports it (CB_STATUS_CARRY_FOREIGN) instead of taking the orbits in it for none.
constexpr uint32_t kWidePlanesFront = (kCarryWordsPerWave - kCarryHeaderWords - kWideQueueWords) / 64;
__host__ __device__ constexpr uint32_t wide_plane(uint32_t p) {  // word offset of lane plane p in the record
  return p < kWidePlanesFront ? kCarryHeaderWords + kWideQueueWords + 64u * p
                              : kCarryWordsPerWave + kCarryHeaderWords + 64u * (p - kWidePlanesFront);
}
static_assert(kCarryHeaderWords + kWideQueueWords <= kCarryWordsPerWave, "the queues end in front of the second header");
static_assert(wide_plane(kWideLanePlanes - 1) + 64u <= 2 * kCarryWordsPerWave,
              "a wide wave uses the carry records of the two waves it replaces");

struct Orbit {
  double cr, ci, r, i;
};

__device__ __forceinline__ bool lane_in(unsigned long long mask) {
  return __builtin_amdgcn_inverse_ballot_w64(mask);
}
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t) (v >> 32));
  return ((unsigned long long) hi << 32) | lo;
}

typedef const DrawArgs __attribute__((address_space(4))) *KernelArgs;
__device__ __forceinline__ KernelArgs fresh_args() {
  KernelArgs p = (KernelArgs) __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
// the same without the barrier: loads the compiler may hoist and keep in scalar registers (the stages of this kernel
// run at two waves per SIMD, where a wait for a scalar load at every stage entry is not hidden by other waves)
__device__ __forceinline__ KernelArgs kernel_args() { return (KernelArgs) __builtin_amdgcn_kernarg_segment_ptr(); }

// ---- one orbit per lane under EXEC (MID, the last short chunk of LONG, the first and last HEAD sample) --------
// (draw_wave.hip, CB_STEP: the six instructions of mandel_step2, the lane-step count, EXEC &= !(16 < |Z|^2))
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

// ---- HEAD, software-pipelined -----------------------------------------------------------------------------------
//
// A HALF-PASS takes one sample per lane from ONE of the lane's two generators (cudabrot.cu:392-393, 398, 326-337):
//   the draw    four XORWOW outputs (rocrand_xorwow.h:165-177) and the two starting coordinates (device_math.h,
//               sample_coordinate2): an integer chain of ~20 dependent operations;
//   the test    cardioid / bulb test (in_main_cardioid2, in_order2_bulb2) and the first four iterations: an fp64 chain
//               of ~25 dependent operations; the survivors' c goes to Q0.
// The two chains of ONE sample depend on each other, those of consecutive samples do not: a BODY of the loop below
// tests the sample that is PENDING (drawn by the body before it, c in cr / ci) while it draws the next one from the
// other generator, alternating the two instruction streams.  A body's text depends on which generator it draws from
// and on the rotation of that generator's words (draw_wave.hip, CB_HEAD_DRAW: logical word j lives in field
// (j + rot) % 5; four outputs move rot on by 4), so the loop is unrolled over the ten states
//   hs = 0..9:  generator hs & 1 (A, B, A, ...), rot = (5 - hs / 2) % 5  (0 0 4 4 3 3 2 2 1 1)
// and entered at the body of the current state.  EXEC is all ones throughout (every lane's generator advances with
// every draw); the lanes still iterating are scalar masks, and the iteration count of the reference is kept exact by
// popcounts of those masks, as in draw_wave.hip's CB_STEP2.
//
// Draw of one output (xk: the oldest word, x0 of rocrand's step; xp: the newest, x4): the new word replaces xk;
// out = d + k * 362437 + new word.
#define CBW_D1(xk) "v_lshrrev_b32 %[t], 2, " xk "\n\t"
#define CBW_D2(xp) "v_lshlrev_b32 %[u], 4, " xp "\n\t"
#define CBW_D3(xk) "v_xor_b32 %[t], %[t], " xk "\n\t"
#define CBW_D4(xk) "v_lshlrev_b32 " xk ", 1, %[t]\n\t"
#define CBW_D5(xp) "v_bitop3_b32 %[u], %[u], " xp ", %[t] bitop3:0x96\n\t"
#define CBW_D6(xk) "v_xor_b32 " xk ", %[u], " xk "\n\t"
#define CBW_D7(xk, xd, out, kw) "v_add3_u32 " out ", " xd ", " xk ", " kw "\n\t"  /* kw: k times the Weyl increment, in a scalar */
static_assert(362437u == 0x587c5u && 2u * 362437u == 0xb0f8au && 3u * 362437u == 0x10974fu &&
                  4u * 362437u == 0x161f14u,
              "multiples of the Weyl increment (rocrand_xorwow.h:174)");
// coordinate, up to the last operation: n = fma(hi, 2^32, lo) with lo = o1, hi = o2 >> 11 (0x41f00000: the high word
// of 2^32 as the literal of a VOP2 fmac); the last operation C = fma(n, 2^-50, 2^-50 - 4) writes cr / ci behind the
// test's last use of them
#define CBW_C1(n) "v_cvt_f64_u32 " n ", %[o1]\n\t"
#define CBW_C2 "v_lshrrev_b32 %[o2], 11, %[o2]\n\t"
#define CBW_C3 "v_cvt_f64_u32 %[f], %[o2]\n\t"
#define CBW_C4(n) "v_fmac_f64_e32 " n ", 0x41f00000, %[f]\n\t"

// The test's instructions, in order (T..): EXEC full; m0 = lanes outside both regions (cudabrot.cu:398), m1 = lanes
// still iterating; cnt = the iterations the reference executes for these samples in HEAD's four steps.
// 0x3fd00000 / 0x40300000: the high words of 0.25 and 16.0 as VOPC literals.
#ifdef CB_BURNING_SHIP
// no shortcut in this variant (cudabrot.cu:397-399): II only, every lane goes on
#define CBW_T2
#define CBW_T3
#define CBW_T4
#define CBW_T5
#define CBW_T6
#define CBW_T7
#define CBW_T7S "s_mov_b64 %[m0], -1\n\t"
#define CBW_T8
#define CBW_T9
#define CBW_T9S "s_bcnt1_i32_b64 %[cnt], %[m0]\n\t"
#else
#define CBW_T2 "v_add_f64 %[x], %[cr], -0.5\n\t"             /* X = 2 (re - 1/4) */
#define CBW_T3 "v_add_f64 %[r], %[cr], 2.0\n\t"              /* T = 2 (re + 1) */
#define CBW_T4 "v_fma_f64 %[q], %[x], %[x], %[a]\n\t"        /* Q */
#define CBW_T5 "v_fma_f64 %[r], %[r], %[r], %[a]\n\t"        /* bulb: fma(T,T,II) */
#define CBW_T6 "v_fma_f64 %[x], %[x], 2.0, %[q]\n\t"         /* S */
#define CBW_T7 "v_cmp_ngt_f64_e32 vcc, 0x3fd00000, %[r]\n\t" /* !(bulb < 1/4) */
#define CBW_T7S "s_mov_b64 %[m0], vcc\n\t"
#define CBW_T8 "v_mul_f64 %[q], %[q], %[x]\n\t"              /* Q * S */
#define CBW_T9 "v_cmp_nlt_f64_e64 %[m1], %[q], %[a]\n\t"     /* !(Q*S < II) */
#define CBW_T9S "s_and_b64 %[m0], %[m0], %[m1]\n\t"                              \
                "s_bcnt1_i32_b64 %[cnt], %[m0]\n\t"
#endif
#define CBW_T1 "v_mul_f64 %[a], %[ci], %[ci]\n\t"            /* II */
// step 1 from z = c; its first product I*I is II
#define CBW_S1A "v_fma_f64 %[a], %[cr], %[cr], -%[a]\n\t"
#define CBW_S1B "v_fma_f64 %[i], " CB_AL "%[cr]" CB_AR ", " CB_AL "%[ci]" CB_AR ", %[ci]\n\t"
#define CBW_SC "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"
#define CBW_SD "v_mul_f64 %[a], %[r], %[r]\n\t"
#define CBW_SE "v_fma_f64 %[a], %[i], %[i], %[a]\n\t"
#define CBW_SF "v_cmp_nlt_f64_e32 vcc, 0x40300000, %[a]\n\t"
#define CBW_S1G "s_and_b64 %[m1], %[m0], vcc\n\t"             /* still iterating after step 1 */
// steps 2..4
#define CBW_SN0 "s_bcnt1_i32_b64 %[tmp], %[m1]\n\t"
#define CBW_SA "v_mul_f64 %[a], %[i], %[i]\n\t"
#define CBW_SN1 "s_add_u32 %[cnt], %[cnt], %[tmp]\n\t"
#define CBW_SB "v_fma_f64 %[a], %[r], %[r], -%[a]\n\t"
#define CBW_SI "v_fma_f64 %[i], " CB_AL "%[r]" CB_AR ", " CB_AL "%[i]" CB_AR ", %[ci]\n\t"
#define CBW_SG "s_and_b64 %[m1], %[m1], vcc\n\t"
// survivors (m1) -> Q0: slot (tail + rank) & 127 of the ring at LDS byte address q0_lds (q0_ci 1024 bytes on)
#define CBW_P0 "s_mov_b64 vcc, %[m1]\n\t"
#define CBW_P1 "v_mbcnt_lo_u32_b32 %[slot], vcc_lo, 0\n\t"
#define CBW_P2 "v_mbcnt_hi_u32_b32 %[slot], vcc_hi, %[slot]\n\t"
#define CBW_P3 "v_add_u32 %[slot], %[tail], %[slot]\n\t"
#define CBW_P4 "v_and_b32 %[slot], 0x7f, %[slot]\n\t"
#define CBW_P5 "v_lshl_add_u32 %[slot], %[slot], 3, %[lds]\n\t"
#define CBW_P6                                                     \
  "s_mov_b64 exec, %[m1]\n\t"                                      \
  "ds_write2st64_b64 %[slot], %[cr], %[ci] offset1:2\n\t"          \
  "s_mov_b64 exec, -1\n\t"

// One body: the test of the pending sample (cr, ci) interleaved with the draw of the next one from the generator
// whose logical words x0..x4 are X0..X4 and whose Weyl value is XD.
// (A scalar instruction that reads what a compare wrote sits at least three instructions behind it: the wave has no
// other wave's instructions to fill the forwarding latency with.)
#define CBW_BODY(X0, X1, X2, X3, X4, XD)                                   \
  CBW_T1 CBW_D1(X0) CBW_T2 CBW_D2(X4)                                      \
  CBW_T3 CBW_D3(X0) CBW_T4 CBW_D4(X0) CBW_T5 CBW_D5(X4)                    \
  CBW_T6 CBW_D6(X0) CBW_T7 CBW_D7(X0, XD, "%[o1]", "%[k1]")                \
  CBW_T8 CBW_D1(X1) CBW_D2(X0) CBW_T7S CBW_T9 CBW_D3(X1)                   \
  CBW_S1A CBW_D4(X1) CBW_S1B CBW_T9S CBW_SC CBW_D5(X0)                     \
  CBW_SD CBW_D6(X1) CBW_SE CBW_D7(X1, XD, "%[o2]", "%[k2]")                \
  CBW_SF CBW_C1("%[nr]") CBW_C2 CBW_SA CBW_S1G CBW_C3                      \
  CBW_SB CBW_SN0 CBW_SI CBW_D1(X2) CBW_SN1                                 \
  CBW_SC CBW_C4("%[nr]") CBW_SD CBW_D2(X1)                                 \
  CBW_SE CBW_D3(X2) CBW_SF CBW_D4(X2) CBW_SA CBW_D5(X1) CBW_SG             \
  CBW_SB CBW_D6(X2) CBW_SN0 CBW_SI CBW_D7(X2, XD, "%[o1]", "%[k3]") CBW_SN1 \
  CBW_SC CBW_D1(X3) CBW_SD CBW_D2(X2)                                      \
  CBW_SE CBW_D3(X3) CBW_SF CBW_D4(X3) CBW_SA CBW_D5(X2) CBW_SG             \
  CBW_SB CBW_D6(X3) CBW_SN0 CBW_SI CBW_D7(X3, XD, "%[o2]", "%[k4]") CBW_SN1 \
  CBW_SC "v_add_u32 " XD ", %[k4], " XD "\n\t"                             \
  CBW_SD CBW_C1("%[ni]") CBW_SE CBW_C2 CBW_SF CBW_C3 CBW_C4("%[ni]")       \
  "s_bcnt1_i32_b64 %[tmp], %[m0]\n\t"                                      \
  "s_add_u32 %[acc0], %[acc0], %[tmp]\n\t"  /* lanes outside both regions, summed over the bodies */ \
  "s_add_u32 %[steps], %[steps], %[cnt]\n\t"                               \
  CBW_SG CBW_P0 CBW_P1 CBW_P2 CBW_P3 CBW_P4 CBW_P5 CBW_P6                  \
  "v_fma_f64 %[cr], %[nr], %[k2m50], %[kk]\n\t"                            \
  "v_fma_f64 %[ci], %[ni], %[k2m50], %[kk]\n\t"

// Behind a body: the ring, and whether the next body runs.  The statement ends behind the body after which Q0 holds a
// MID pass (nroom = q0_count - 64 has carried) or the caller's count of bodies is used up (ctr: the input has ended --
// the pending sample is then the launch's last, the caller tests it -- or the progress board is due): a scalar
// instruction and a branch not taken each (round 4: six scalar instructions instead of ten); the state (hs) is set on
// the way out only (the stubs CBW_EXIT).  rejected / escaped-in-HEAD are made from acc0 and the growth of q0_count
// behind the loop.
#define CBW_AFTER(exit_label, next_label)                            \
  "s_bcnt1_i32_b64 %[tmp], %[m1]\n\t"                                \
  "s_add_u32 %[tail], %[tail], %[tmp]\n\t"                           \
  "s_add_u32 %[nroom], %[nroom], %[tmp]\n\t"                         \
  "s_cbranch_scc1 7" exit_label "f\n\t"                              \
  "s_sub_u32 %[ctr], %[ctr], 1\n\t"                                  \
  "s_cbranch_scc1 " exit_label "f\n\t"                               \
  next_label
#define CBW_EXIT(label, next_hs)                                     \
  "7" label ":\n\t"                                                  \
  "s_sub_u32 %[ctr], %[ctr], 1\n\t"                                  \
  label ":\n\t"                                                      \
  "s_mov_b32 %[hs], " next_hs "\n\t"                                 \
  "s_branch 99f\n\t"

// On the way in: the PENDING sample -- the one the body before drew, whose test this call's first body makes -- is not
// kept in registers across the other stages (round 4: four persistent registers fewer): a generator's words x1..x4 ARE
// the xorshift words of its last four outputs, so until it draws again
//   out_j = d + x_j - (4 - j) * 362437        (rocrand_xorwow.h:174-176: out = d_j + x_new, d_j = d - (4 - j) * 362437)
// and the coordinates are made from them again (sample_coordinate2): 14 vector instructions per call of the statement.
// X1..X4, XD: the logical words and the Weyl value of the generator that drew it, as they are NOW (the mapping of the
// body that draws from it next).
#define CBW_PENDING(X1, X2, X3, X4, XD)                              \
  "v_add3_u32 %[o1], " XD ", " X1 ", %[km3]\n\t"                     \
  "v_add3_u32 %[o2], " XD ", " X2 ", %[km2]\n\t"                     \
  CBW_C1("%[nr]") CBW_C2 CBW_C3 CBW_C4("%[nr]")                      \
  "v_add3_u32 %[o1], " XD ", " X3 ", %[km1]\n\t"                     \
  "v_add_u32 %[o2], " XD ", " X4 "\n\t"                              \
  CBW_C1("%[ni]") CBW_C2 CBW_C3 CBW_C4("%[ni]")                      \
  "v_fma_f64 %[cr], %[nr], %[k2m50], %[kk]\n\t"                      \
  "v_fma_f64 %[ci], %[ni], %[k2m50], %[kk]\n\t"
#define CBW_PENDING_A(f1, f2, f3, f4) CBW_PENDING("%[a" #f1 "]", "%[a" #f2 "]", "%[a" #f3 "]", "%[a" #f4 "]", "%[ad]")
#define CBW_PENDING_B(f1, f2, f3, f4) CBW_PENDING("%[b" #f1 "]", "%[b" #f2 "]", "%[b" #f3 "]", "%[b" #f4 "]", "%[bd]")

// the ten bodies: generator A / B alternately, rot = 0 0 4 4 3 3 2 2 1 1 (logical word j in field (j + rot) % 5)
#define CBW_BODY_A(f0, f1, f2, f3, f4) CBW_BODY("%[a" #f0 "]", "%[a" #f1 "]", "%[a" #f2 "]", "%[a" #f3 "]", "%[a" #f4 "]", "%[ad]")
#define CBW_BODY_B(f0, f1, f2, f3, f4) CBW_BODY("%[b" #f0 "]", "%[b" #f1 "]", "%[b" #f2 "]", "%[b" #f3 "]", "%[b" #f4 "]", "%[bd]")

struct Xorwow2 {
  Xorwow a, b;
};
// sample_coordinate2 (device_math.h) of two outputs that have already been drawn
__device__ __forceinline__ double coordinate2_of(uint32_t v1, uint32_t v2) {
  const double v = __builtin_fma((double) (v2 >> 11), 4294967296.0, (double) v1);
  return __builtin_fma(v, 0x1p-50, 0x1p-50 - 4.0);
}

// Bodies in a row (at least one).  On entry a sample is pending in (cr, ci) and halves >= 1 draws are still to be
// made; hs is the state (which generator the next draw takes, and its rotation).  q0_tail = q0_head + q0_count (only
// its low seven bits matter).  The three statistics are added to.
// Like every statement of this file that writes lane registers which live across the scheduler's loop, it is EXECUTED
// in every iteration and skips its work when `enable` is 0: the loop body then is straight-line code for the
// compiler, and every such register has one definition per iteration (as conditional blocks the statements made it
// keep two copies of the whole lane state and move one into the other around every stage).
__device__ __forceinline__ void head_bodies(uint32_t enable, Xorwow2 &g, uint32_t &halves,
                                            uint32_t &hs, uint32_t q0_tail, uint32_t &q0_count, uint32_t q0_lds,
                                            uint32_t &n_rejected, uint32_t &n_too_fast, uint32_t &n_steps) {
  static_assert(kQ0Cap == 128, "ring mask and the 1024-byte distance of q0_ci in CBW_P4 / CBW_P6");
  unsigned long long save, m0, m1;
  uint32_t cnt, tmp, slot, t, u, o1, o2, acc0 = 0;
  double a, r, i, x, q, f, nr, ni, kk, cr, ci;
  enable = __builtin_amdgcn_readfirstlane(enable);
  halves = __builtin_amdgcn_readfirstlane(halves);
  hs = __builtin_amdgcn_readfirstlane(hs);
  q0_tail = __builtin_amdgcn_readfirstlane(q0_tail);
  q0_count = __builtin_amdgcn_readfirstlane(q0_count);
  n_rejected = __builtin_amdgcn_readfirstlane(n_rejected);
  n_too_fast = __builtin_amdgcn_readfirstlane(n_too_fast);
  n_steps = __builtin_amdgcn_readfirstlane(n_steps);
  // bodies until the input ends or the board is due (halves % 256 == 0), counted down from one less
  static_assert(kPrioHalves == 256, "a call ends where halves % 256 == 0");
  const uint32_t n_bodies = ((halves - 1u) & (kPrioHalves - 1u)) + 1u;
  uint32_t ctr = n_bodies - 1u, nroom = q0_count - 64u;
  const uint32_t q0_before = q0_count;
  asm volatile(
      "s_mov_b64 %[save], exec\n\t"
      "s_cmp_eq_u32 %[en], 0\n\t"
      "s_cbranch_scc1 99f\n\t"
      "s_mov_b64 exec, -1\n\t"
      "v_mov_b64 %[kk], %[k2m50]\n\t"
      "v_add_f64 %[kk], %[kk], -4.0\n\t"  // 2^-50 - 4, exactly
      // enter at the body of the current state, through the stub that makes its pending sample: drawn by the OTHER
      // generator, whose words lie as the body after this one expects them
      "s_cmp_ge_u32 %[hs], 5\n\t"
      "s_cbranch_scc1 60f\n\t"
      "s_cmp_eq_u32 %[hs], 0\n\t"
      "s_cbranch_scc1 30f\n\t"
      "s_cmp_eq_u32 %[hs], 1\n\t"
      "s_cbranch_scc1 31f\n\t"
      "s_cmp_eq_u32 %[hs], 2\n\t"
      "s_cbranch_scc1 32f\n\t"
      "s_cmp_eq_u32 %[hs], 3\n\t"
      "s_cbranch_scc1 33f\n\t"
      "s_branch 34f\n\t"
      "60:\n\t"
      "s_cmp_eq_u32 %[hs], 5\n\t"
      "s_cbranch_scc1 35f\n\t"
      "s_cmp_eq_u32 %[hs], 6\n\t"
      "s_cbranch_scc1 36f\n\t"
      "s_cmp_eq_u32 %[hs], 7\n\t"
      "s_cbranch_scc1 37f\n\t"
      "s_cmp_eq_u32 %[hs], 8\n\t"
      "s_cbranch_scc1 38f\n\t"
      "s_branch 39f\n\t"
      "30:\n\t" CBW_PENDING_B(1, 2, 3, 4) "s_branch 10f\n\t"   // (state 0 draws from A rot 0; B lies as in state 1: rot 0)
      "31:\n\t" CBW_PENDING_A(0, 1, 2, 3) "s_branch 11f\n\t"   // (A as in state 2: rot 4)
      "32:\n\t" CBW_PENDING_B(0, 1, 2, 3) "s_branch 12f\n\t"   // (B as in state 3: rot 4)
      "33:\n\t" CBW_PENDING_A(4, 0, 1, 2) "s_branch 13f\n\t"   // (A as in state 4: rot 3)
      "34:\n\t" CBW_PENDING_B(4, 0, 1, 2) "s_branch 14f\n\t"   // (B as in state 5: rot 3)
      "35:\n\t" CBW_PENDING_A(3, 4, 0, 1) "s_branch 15f\n\t"   // (A as in state 6: rot 2)
      "36:\n\t" CBW_PENDING_B(3, 4, 0, 1) "s_branch 16f\n\t"   // (B as in state 7: rot 2)
      "37:\n\t" CBW_PENDING_A(2, 3, 4, 0) "s_branch 17f\n\t"   // (A as in state 8: rot 1)
      "38:\n\t" CBW_PENDING_B(2, 3, 4, 0) "s_branch 18f\n\t"   // (B as in state 9: rot 1)
      "39:\n\t" CBW_PENDING_A(1, 2, 3, 4) "s_branch 19f\n\t"   // (A as in state 0: rot 0)
      "10:\n\t" CBW_BODY_A(0, 1, 2, 3, 4) CBW_AFTER("81", "11:\n\t")  // rot 0
      CBW_BODY_B(0, 1, 2, 3, 4) CBW_AFTER("82", "12:\n\t")
      CBW_BODY_A(4, 0, 1, 2, 3) CBW_AFTER("83", "13:\n\t")            // rot 4
      CBW_BODY_B(4, 0, 1, 2, 3) CBW_AFTER("84", "14:\n\t")
      CBW_BODY_A(3, 4, 0, 1, 2) CBW_AFTER("85", "15:\n\t")            // rot 3
      CBW_BODY_B(3, 4, 0, 1, 2) CBW_AFTER("86", "16:\n\t")
      CBW_BODY_A(2, 3, 4, 0, 1) CBW_AFTER("87", "17:\n\t")            // rot 2
      CBW_BODY_B(2, 3, 4, 0, 1) CBW_AFTER("88", "18:\n\t")
      CBW_BODY_A(1, 2, 3, 4, 0) CBW_AFTER("89", "19:\n\t")            // rot 1
      CBW_BODY_B(1, 2, 3, 4, 0) CBW_AFTER("80", "s_branch 10b\n\t")
      CBW_EXIT("81", "1") CBW_EXIT("82", "2") CBW_EXIT("83", "3") CBW_EXIT("84", "4") CBW_EXIT("85", "5")
      CBW_EXIT("86", "6") CBW_EXIT("87", "7") CBW_EXIT("88", "8") CBW_EXIT("89", "9") CBW_EXIT("80", "0")
      "99:\n\t"
      "s_mov_b64 exec, %[save]\n\t"
      "s_nop 4\n\t"
      : [ctr] "+s"(ctr), [hs] "+s"(hs), [tail] "+s"(q0_tail), [nroom] "+s"(nroom), [acc0] "+s"(acc0),
        [steps] "+s"(n_steps), [m0] "=&s"(m0), [m1] "=&s"(m1), [cnt] "=&s"(cnt),
        [save] "=&s"(save), [tmp] "=&s"(tmp), [a] "=&v"(a), [r] "=&v"(r), [i] "=&v"(i), [x] "=&v"(x),
        [q] "=&v"(q), [slot] "=&v"(slot), [t] "=&v"(t), [u] "=&v"(u), [o1] "=&v"(o1), [o2] "=&v"(o2), [f] "=&v"(f),
        [nr] "=&v"(nr), [ni] "=&v"(ni), [kk] "=&v"(kk), [cr] "=&v"(cr), [ci] "=&v"(ci),
        [a0] "+v"(g.a.x0), [a1] "+v"(g.a.x1), [a2] "+v"(g.a.x2), [a3] "+v"(g.a.x3), [a4] "+v"(g.a.x4), [ad] "+v"(g.a.d),
        [b0] "+v"(g.b.x0), [b1] "+v"(g.b.x1), [b2] "+v"(g.b.x2), [b3] "+v"(g.b.x3), [b4] "+v"(g.b.x4), [bd] "+v"(g.b.d)
      : [en] "s"(enable), [lds] "s"(q0_lds), [k2m50] "s"(0x1p-50), [k1] "s"(362437u), [k2] "s"(2u * 362437u),
        [k3] "s"(3u * 362437u), [k4] "s"(4u * 362437u), [km1] "s"(0u - 362437u), [km2] "s"(0u - 2u * 362437u),
        [km3] "s"(0u - 3u * 362437u)
      : "vcc", "scc", "memory");
  // rejected: the lanes inside a region; escaped in HEAD (too fast): outside both regions and not among the survivors
  // (ctr has gone down by one per body that ran, past zero if all of them did)
  const uint32_t bodies = enable ? n_bodies - 1u - __builtin_amdgcn_readfirstlane(ctr) : 0u;
  halves -= bodies;
  q0_count = __builtin_amdgcn_readfirstlane(nroom) + 64u;
  const uint32_t survivors = q0_count - q0_before;
  n_rejected += 64u * bodies - acc0;
  n_too_fast += acc0 - survivors;
}

// logical word j of a generator whose words are rotated by ROT (field (j + ROT) % 5), and back
template <int K>
__device__ __forceinline__ uint32_t &xorwow_word(Xorwow &s) {
  static_assert(K >= 0 && K < 5, "five words");
  if constexpr (K == 0) return s.x0;
  if constexpr (K == 1) return s.x1;
  if constexpr (K == 2) return s.x2;
  if constexpr (K == 3) return s.x3;
  return s.x4;
}
template <int ROT>
__device__ __forceinline__ Xorwow xorwow_unrotated(Xorwow &s) {  // rotated fields -> logical order
  Xorwow r;
  r.x0 = xorwow_word<(0 + ROT) % 5>(s);
  r.x1 = xorwow_word<(1 + ROT) % 5>(s);
  r.x2 = xorwow_word<(2 + ROT) % 5>(s);
  r.x3 = xorwow_word<(3 + ROT) % 5>(s);
  r.x4 = xorwow_word<(4 + ROT) % 5>(s);
  r.d = s.d;
  return r;
}
__device__ __forceinline__ Xorwow xorwow_unrotate(Xorwow s, uint32_t rot) {
  switch (rot) {
    case 1: return xorwow_unrotated<1>(s);
    case 2: return xorwow_unrotated<2>(s);
    case 3: return xorwow_unrotated<3>(s);
    case 4: return xorwow_unrotated<4>(s);
    default: return s;
  }
}
// logical order -> fields rotated by 4 (what one sample drawn in logical order leaves for the bodies: hs = 1)
__device__ __forceinline__ Xorwow xorwow_rotated4(const Xorwow &s) {
  Xorwow r;  // logical j -> field (j + 4) % 5
  r.x4 = s.x0;
  r.x0 = s.x1;
  r.x1 = s.x2;
  r.x2 = s.x3;
  r.x3 = s.x4;
  r.d = s.d;
  return r;
}

// ---- MID in one piece (draw_wave.hip, mid_pass): 64 lanes pop c from Q0, re-derive z after HEAD's four
// iterations, run n_steps more under EXEC, push the survivors' (c, z) to Q1 ------------------------------------
// a tested step with the lane count taken behind its first instruction (which waits for the EXEC of the compare
// before it anyway)
#define CBW_MID_STEP                                  \
  "v_mul_f64 %[a], %[i], %[i]\n\t"                    \
  "s_bcnt1_i32_b64 %[tmp], exec\n\t"                  \
  "v_fma_f64 %[a], %[r], %[r], -%[a]\n\t"             \
  "s_add_u32 %[cnt], %[cnt], %[tmp]\n\t"              \
  "v_fma_f64 %[i], " CB_AL "%[r]" CB_AR ", " CB_AL "%[i]" CB_AR ", %[ci]\n\t"             \
  "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"              \
  "v_mul_f64 %[a], %[r], %[r]\n\t"                    \
  "v_fma_f64 %[a], %[i], %[i], %[a]\n\t"              \
  "v_cmpx_nlt_f64_e32 vcc, 0x40300000, %[a]\n\t"
#define CB_STEP_NOTEST                                \
  "v_mul_f64 %[a], %[i], %[i]\n\t"                    \
  "v_fma_f64 %[a], %[r], %[r], -%[a]\n\t"             \
  "v_fma_f64 %[i], " CB_AL "%[r]" CB_AR ", " CB_AL "%[i]" CB_AR ", %[ci]\n\t"             \
  "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"
// map / shift / cols / rows: the interior map (DrawArgs::interior_map; map = 0: none).  The cell of every popped c is
// looked up while the stage iterates -- column floor((Cr + 4) 2^shift), row floor(|Ci| 2^shift) on doubled coordinates,
// one byte load per lane behind the LDS reads, waited for in front of the push -- and a lane whose cell is marked is not
// pushed: `hit` (among the lanes of `take`; the caller counts those of `alive` as never-escaping and takes one outside
// `alive`, a sample of a proven cell that escaped, for the broken invariant it would be).
__device__ __forceinline__ void mid_pass(unsigned long long take, uint32_t lane_plus_head,
                                         uint32_t q0_lds, uint32_t n_steps, uint32_t q1_tail,
                                         uint32_t q1_lds, unsigned long long &alive,
                                         uint32_t &lane_steps, unsigned long long map, uint32_t shift,
                                         uint32_t cols, uint32_t rows, unsigned long long &hit) {
  static_assert(kQ0Cap == 128 && kQ1Cap == 96 && kHeadSteps == 4, "ring mask, ring length and plane distances below");
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
      "v_add_f64 %[a], %[cr], 4.0\n\t"
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
      CB_STEP_NOTEST CB_STEP_NOTEST CB_STEP_NOTEST
      // the steps in groups of four (one look at EXEC and one taken branch per group: at two waves per SIMD nothing
      // hides the scalar unit's wait for the compare), then what is left one by one
      "s_cmp_lt_u32 %[ctr], 4\n\t"
      "s_cbranch_scc1 3f\n\t"
      "4:\n\t"
      CBW_MID_STEP CBW_MID_STEP CBW_MID_STEP CBW_MID_STEP
      "s_sub_u32 %[ctr], %[ctr], 4\n\t"
      "s_cbranch_execz 2f\n\t"
      "s_cmp_ge_u32 %[ctr], 4\n\t"
      "s_cbranch_scc1 4b\n\t"
      "3:\n\t"
      "s_cmp_eq_u32 %[ctr], 0\n\t"
      "s_cbranch_scc1 2f\n\t"
      "1:\n\t"
      CBW_MID_STEP
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

// ---- LONG: four orbits per lane, EXEC untouched, one escape test per chunk ------------------------------------
//
// draw_wave.hip's iterate_chunk2_sparse with four chains instead of two: every lane computes every step of its four
// slots (an idle slot computes garbage that nothing reads), 16 fp64 instructions per step, and ONE test behind the
// chunk's last step -- !(|Z|^2 <= 16), exact for every sample with |C|^2 below the threshold (escape is absorbing
// there: DESIGN.md 4.1a), while the others (|c| within 2^-14 of 2: one sample in 10^8) are decided by recomputing the
// orbit (verify_chunk_escape).
#define CBW_L4A                                           \
  "v_mul_f64 %[t0], %[i0], %[i0]\n\t"                     \
  "v_mul_f64 %[t1], %[i1], %[i1]\n\t"                     \
  "v_mul_f64 %[t2], %[i2], %[i2]\n\t"                     \
  "v_mul_f64 %[t3], %[i3], %[i3]\n\t"                     \
  "v_fma_f64 %[t0], %[r0], %[r0], -%[t0]\n\t"             \
  "v_fma_f64 %[t1], %[r1], %[r1], -%[t1]\n\t"             \
  "v_fma_f64 %[t2], %[r2], %[r2], -%[t2]\n\t"             \
  "v_fma_f64 %[t3], %[r3], %[r3], -%[t3]\n\t"             \
  "v_fma_f64 %[i0], " CB_AL "%[r0]" CB_AR ", " CB_AL "%[i0]" CB_AR ", %[ci0]\n\t" \
  "v_fma_f64 %[i1], " CB_AL "%[r1]" CB_AR ", " CB_AL "%[i1]" CB_AR ", %[ci1]\n\t" \
  "v_fma_f64 %[i2], " CB_AL "%[r2]" CB_AR ", " CB_AL "%[i2]" CB_AR ", %[ci2]\n\t" \
  "v_fma_f64 %[i3], " CB_AL "%[r3]" CB_AR ", " CB_AL "%[i3]" CB_AR ", %[ci3]\n\t" \
  "v_fma_f64 %[r0], %[t0], 0.5, %[cr0]\n\t"               \
  "v_fma_f64 %[r1], %[t1], 0.5, %[cr1]\n\t"               \
  "v_fma_f64 %[r2], %[t2], 0.5, %[cr2]\n\t"               \
  "v_fma_f64 %[r3], %[t3], 0.5, %[cr3]\n\t"
#define CBW_L4X5 CBW_L4A CBW_L4A CBW_L4A CBW_L4A CBW_L4A
#define CBW_L4X30 CBW_L4X5 CBW_L4X5 CBW_L4X5 CBW_L4X5 CBW_L4X5 CBW_L4X5
// behind the last step: d_k = escaped (!(|Z|^2 <= 16)), c_k = the sample is not of the sure class (!(|C|^2 < kt))
#define CBW_L4TEST                                        \
  "v_mul_f64 %[t0], %[r0], %[r0]\n\t"                     \
  "v_mul_f64 %[t1], %[r1], %[r1]\n\t"                     \
  "v_mul_f64 %[t2], %[r2], %[r2]\n\t"                     \
  "v_mul_f64 %[t3], %[r3], %[r3]\n\t"                     \
  "v_fma_f64 %[t0], %[i0], %[i0], %[t0]\n\t"              \
  "v_fma_f64 %[t1], %[i1], %[i1], %[t1]\n\t"              \
  "v_fma_f64 %[t2], %[i2], %[i2], %[t2]\n\t"              \
  "v_fma_f64 %[t3], %[i3], %[i3], %[t3]\n\t"              \
  "v_cmp_nle_f64_e64 %[d0], %[t0], %[k16]\n\t"            \
  "v_cmp_nle_f64_e64 %[d1], %[t1], %[k16]\n\t"            \
  "v_cmp_nle_f64_e64 %[d2], %[t2], %[k16]\n\t"            \
  "v_cmp_nle_f64_e64 %[d3], %[t3], %[k16]\n\t"            \
  "v_mul_f64 %[t0], %[cr0], %[cr0]\n\t"                   \
  "v_mul_f64 %[t1], %[cr1], %[cr1]\n\t"                   \
  "v_mul_f64 %[t2], %[cr2], %[cr2]\n\t"                   \
  "v_mul_f64 %[t3], %[cr3], %[cr3]\n\t"                   \
  "v_fma_f64 %[t0], %[ci0], %[ci0], %[t0]\n\t"            \
  "v_fma_f64 %[t1], %[ci1], %[ci1], %[t1]\n\t"            \
  "v_fma_f64 %[t2], %[ci2], %[ci2], %[t2]\n\t"            \
  "v_fma_f64 %[t3], %[ci3], %[ci3], %[t3]\n\t"            \
  "v_cmp_nlt_f64_e64 %[c0], %[t0], %[kt]\n\t"             \
  "v_cmp_nlt_f64_e64 %[c1], %[t1], %[kt]\n\t"             \
  "v_cmp_nlt_f64_e64 %[c2], %[t2], %[kt]\n\t"             \
  "v_cmp_nlt_f64_e64 %[c3], %[t3], %[kt]\n\t"

// kChunk steps on the four slots (called with EXEC = all 64 lanes).  esc[k]: lanes of mask[k] whose orbit escaped or
// whose sample is not of the sure class; doubt: lanes (of any mask) that hold such a sample in some slot -- for
// them the caller decides exactly (rare).  z of slots outside the masks is clobbered.
__device__ __forceinline__ void iterate_chunk4(uint32_t enable, const unsigned long long (&mask)[kSlots],
                                               Orbit (&o)[kSlots], unsigned long long (&esc)[kSlots],
                                               unsigned long long &doubt, double threshold) {
  static_assert(kChunk == 60 && kSlots == 4, "unrolled: 2 x 30 steps of four chains");
  unsigned long long c0, c1, c2, c3, d0, d1, d2, d3;
  double t0, t1, t2, t3;
  const double k16 = 16.0, kt = threshold;
  enable = __builtin_amdgcn_readfirstlane(enable);
  asm volatile(
      "s_mov_b64 %[c0], 0\n\t"
      "s_mov_b64 %[c1], 0\n\t"
      "s_mov_b64 %[c2], 0\n\t"
      "s_mov_b64 %[c3], 0\n\t"
      "s_mov_b64 %[d0], 0\n\t"
      "s_mov_b64 %[d1], 0\n\t"
      "s_mov_b64 %[d2], 0\n\t"
      "s_mov_b64 %[d3], 0\n\t"
      "s_cmp_eq_u32 %[en], 0\n\t"
      "s_cbranch_scc1 9f\n\t"
      CBW_L4X30 CBW_L4X30 CBW_L4TEST
      "9:\n\t"
      "s_nop 2\n\t"
      : [r0] "+v"(o[0].r), [i0] "+v"(o[0].i), [r1] "+v"(o[1].r), [i1] "+v"(o[1].i), [r2] "+v"(o[2].r),
        [i2] "+v"(o[2].i), [r3] "+v"(o[3].r), [i3] "+v"(o[3].i), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
        [t3] "=&v"(t3), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3), [d0] "=&s"(d0),
        [d1] "=&s"(d1), [d2] "=&s"(d2), [d3] "=&s"(d3)
      : [cr0] "v"(o[0].cr), [ci0] "v"(o[0].ci), [cr1] "v"(o[1].cr), [ci1] "v"(o[1].ci), [cr2] "v"(o[2].cr),
        [ci2] "v"(o[2].ci), [cr3] "v"(o[3].cr), [ci3] "v"(o[3].ci), [k16] "s"(k16), [kt] "s"(kt), [en] "s"(enable)
      : "scc");
  esc[0] = mask[0] & (d0 | c0);
  esc[1] = mask[1] & (d1 | c1);
  esc[2] = mask[2] & (d2 | c2);
  esc[3] = mask[3] & (d3 | c3);
  doubt = (mask[0] & c0) | (mask[1] & c1) | (mask[2] & c2) | (mask[3] & c3);
}

// The exact decision for the lanes of `doubt` (draw_wave.hip, verify_chunk_escape): did the orbit with starting
// point (cr, ci) escape during the kChunk iterations after its first `done` ones?
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

#define CB_STR2(x) #x
#define CB_STR(x) CB_STR2(x)
#define CB_CHUNK_S CB_STR(CB_CHUNK)

// long_refill4 (draw_wave.hip, long_refill, for the four slots in one statement): the idle lanes (l_rem == 0) of
// every slot in turn take (c, z) from Q1 -- ring slot (q1_head + rank) mod 96 at LDS byte address q1_lds, planes 768
// bytes apart -- with l_rem = long_steps and the periodicity check's saved point = the entry point; ONE wait for the
// LDS behind all of them (at two waves per SIMD a wait per slot is not hidden by other waves).  full[k]: lanes of
// slot k with a whole chunk ahead; tail[k]: lanes left with exactly the last, shorter chunk (l_rem == tail_value;
// ~0 when there is none).  q1_head / q1_count are moved on.
#define CBW_REFILL_SLOT(k)                                                \
      "v_cmp_eq_u32_e32 vcc, 0, %[lrem" #k "]\n\t"     /* idle lanes */  \
      "s_bcnt1_i32_b64 %[n], vcc\n\t"                                     \
      "s_min_u32 %[n], %[n], %[qc]\n\t"                                   \
      "s_cmp_eq_u32 %[n], 0\n\t"                                          \
      "s_cbranch_scc1 2" #k "f\n\t"                                       \
      "v_mbcnt_lo_u32_b32 %[rank], vcc_lo, 0\n\t"                         \
      "v_mbcnt_hi_u32_b32 %[rank], vcc_hi, %[rank]\n\t"                   \
      "s_mov_b64 exec, vcc\n\t"                                           \
      "v_cmpx_gt_u32_e32 vcc, %[n], %[rank]\n\t"       /* the first n idle lanes */ \
      "v_add_u32 %[slot], %[head], %[rank]\n\t"        /* < 96 + 64 */   \
      "v_subrev_u32 %[t], 96, %[slot]\n\t"                                \
      "v_min_u32 %[slot], %[slot], %[t]\n\t"                              \
      "v_lshl_add_u32 %[slot], %[slot], 3, %[q1]\n\t"                     \
      "ds_read_b64 %[cr" #k "], %[slot]\n\t"                              \
      "ds_read_b64 %[ci" #k "], %[slot] offset:768\n\t"                   \
      "ds_read_b64 %[r" #k "], %[slot] offset:1536\n\t"                   \
      "ds_read_b64 %[i" #k "], %[slot] offset:2304\n\t"                   \
      "ds_read_b64 %[sr" #k "], %[slot] offset:1536\n\t" /* the saved point of the periodicity check = z */ \
      "ds_read_b64 %[si" #k "], %[slot] offset:2304\n\t"                  \
      "v_mov_b32 %[lrem" #k "], %[ls]\n\t"                                \
      "s_mov_b64 exec, %[save]\n\t"                                       \
      "s_add_u32 %[head], %[head], %[n]\n\t"                              \
      "s_sub_u32 %[qc], %[qc], %[n]\n\t"                                  \
      "s_sub_u32 %[n], %[head], 96\n\t"                                   \
      "s_cmp_ge_u32 %[head], 96\n\t"                                      \
      "s_cselect_b32 %[head], %[n], %[head]\n\t"                          \
      "2" #k ":\n\t"                                                      \
      "v_cmp_le_u32_e32 vcc, " CB_CHUNK_S ", %[lrem" #k "]\n\t"           \
      "v_cmp_eq_u32_e64 %[tail" #k "], %[tv], %[lrem" #k "]\n\t"          \
      "s_mov_b64 %[full" #k "], vcc\n\t"
__device__ __forceinline__ void long_refill4(Orbit (&o)[kSlots], double (&seen_r)[kSlots], double (&seen_i)[kSlots],
                                             int (&l_rem)[kSlots], uint32_t &q1_head, uint32_t &q1_count, uint32_t q1_lds,
                                             uint32_t long_steps, uint32_t tail_value,
                                             unsigned long long (&full)[kSlots], unsigned long long (&tail)[kSlots]) {
  static_assert(kQ1Cap == 96 && kSlots == 4, "ring length, plane distances and the four slot blocks below");
  unsigned long long save;
  uint32_t n, rank, slot, t;
  q1_head = __builtin_amdgcn_readfirstlane(q1_head);
  q1_count = __builtin_amdgcn_readfirstlane(q1_count);
  asm volatile(
      "s_mov_b64 %[save], exec\n\t"
      CBW_REFILL_SLOT(0) CBW_REFILL_SLOT(1) CBW_REFILL_SLOT(2) CBW_REFILL_SLOT(3)
      "s_waitcnt lgkmcnt(0)\n\t"
      : [cr0] "+v"(o[0].cr), [ci0] "+v"(o[0].ci), [r0] "+v"(o[0].r), [i0] "+v"(o[0].i), [sr0] "+v"(seen_r[0]),
        [si0] "+v"(seen_i[0]), [lrem0] "+v"(l_rem[0]),
        [cr1] "+v"(o[1].cr), [ci1] "+v"(o[1].ci), [r1] "+v"(o[1].r), [i1] "+v"(o[1].i), [sr1] "+v"(seen_r[1]),
        [si1] "+v"(seen_i[1]), [lrem1] "+v"(l_rem[1]),
        [cr2] "+v"(o[2].cr), [ci2] "+v"(o[2].ci), [r2] "+v"(o[2].r), [i2] "+v"(o[2].i), [sr2] "+v"(seen_r[2]),
        [si2] "+v"(seen_i[2]), [lrem2] "+v"(l_rem[2]),
        [cr3] "+v"(o[3].cr), [ci3] "+v"(o[3].ci), [r3] "+v"(o[3].r), [i3] "+v"(o[3].i), [sr3] "+v"(seen_r[3]),
        [si3] "+v"(seen_i[3]), [lrem3] "+v"(l_rem[3]),
        [head] "+s"(q1_head), [qc] "+s"(q1_count), [n] "=&s"(n), [save] "=&s"(save),
        [full0] "=&s"(full[0]), [full1] "=&s"(full[1]), [full2] "=&s"(full[2]), [full3] "=&s"(full[3]),
        [tail0] "=&s"(tail[0]), [tail1] "=&s"(tail[1]), [tail2] "=&s"(tail[2]), [tail3] "=&s"(tail[3]),
        [rank] "=&v"(rank), [slot] "=&v"(slot), [t] "=&v"(t)
      : [q1] "s"(q1_lds), [ls] "s"(long_steps), [tv] "s"(tail_value)
      : "vcc", "scc", "memory");
}

// One slot's part of long_retire4 (K: the slot).  ran / esc: the lanes of the slot that ran the chunk / whose orbit
// escaped in it.
//   escaped lanes    l_rem = 0; those whose chunk lies at or above min_iter (l_rem <= accept_rem: all of them in this
//                    kernel's launches, `bad` says otherwise) push c to Q2 and add cntk - l_rem to `counted`
//   the others       l_rem -= kChunk; ended: reached max_iter (cudabrot.cu:339); periodic (chkf != 0 only): z is bit
//                    for bit the saved point -- retired, the remaining iterations added to skip; else Brent's
//                    schedule, refined (draw_wave.hip, long_retire): re-save z when the number of chunks done has no
//                    set bit below its top two
#define CBW_RETIRE_SLOT(K)                                                 \
      "s_cmp_eq_u64 %[ran" #K "], 0\n\t"  /* nothing ran in this slot */ \
      "s_cbranch_scc1 3" #K "f\n\t"                                        \
      "s_mov_b64 exec, %[esc" #K "]\n\t"                                   \
      "s_cbranch_execz 1" #K "f\n\t"                                       \
      "v_cmp_ge_i32_e32 vcc, %[thr], %[lrem" #K "]\n\t"                    \
      "v_sub_u32 %[t], %[cntk], %[lrem" #K "]\n\t"                         \
      "v_mov_b32 %[lrem" #K "], 0\n\t"                                     \
      "s_andn2_b64 %[m], exec, vcc\n\t"        /* escaped but not accepted: cannot happen */ \
      "s_or_b64 %[bad], %[bad], %[m]\n\t"                                  \
      "s_mov_b64 exec, vcc\n\t"                                            \
      "s_cbranch_execz 1" #K "f\n\t"                                       \
      "v_add_u32 %[acc], %[acc], %[t]\n\t"                                 \
      "v_mbcnt_lo_u32_b32 %[slot], exec_lo, 0\n\t"                         \
      "v_mbcnt_hi_u32_b32 %[slot], exec_hi, %[slot]\n\t"                   \
      "s_bcnt1_i32_b64 %[n], exec\n\t"                                     \
      "v_add_u32 %[slot], %[tail2], %[slot]\n\t"                           \
      "s_add_u32 %[pushed], %[pushed], %[n]\n\t"                           \
      "v_subrev_u32 %[t], 320, %[slot]\n\t"                                \
      "s_add_u32 %[tail2], %[tail2], %[n]\n\t"                             \
      "v_min_u32 %[slot], %[slot], %[t]\n\t"                               \
      "s_sub_u32 %[n], %[tail2], 320\n\t"                                  \
      "v_lshl_add_u32 %[slot], %[slot], 3, %[q2]\n\t"                      \
      "s_cmp_ge_u32 %[tail2], 320\n\t"                                     \
      "ds_write_b64 %[slot], %[cr" #K "]\n\t"                              \
      "s_cselect_b32 %[tail2], %[n], %[tail2]\n\t"                         \
      "ds_write_b64 %[slot], %[ci" #K "] offset:2560\n\t"                  \
      "1" #K ":\n\t"                                                       \
      "s_andn2_b64 exec, %[ran" #K "], %[esc" #K "]\n\t"                   \
      "s_cbranch_execz 3" #K "f\n\t"                                       \
      "v_subrev_u32 %[lrem" #K "], " CB_CHUNK_S ", %[lrem" #K "]\n\t"      \
      "v_sub_u32 %[t], %[ls], %[lrem" #K "]\n\t"      /* steps done, for the save schedule below */ \
      "v_cmp_eq_u32_e64 %[m], 0, %[lrem" #K "]\n\t"   /* ended */         \
      "v_cvt_f32_u32 %[t], %[t]\n\t"                                       \
      "s_cmp_eq_u32 %[chkf], 0\n\t"                                        \
      "s_cbranch_scc1 2" #K "f\n\t"                                        \
      "v_cmp_eq_u64_e32 vcc, %[r" #K "], %[sr" #K "]\n\t"                  \
      "v_cmp_eq_u64_e64 %[per], %[i" #K "], %[si" #K "]\n\t"               \
      "v_mul_f32 %[t], %[invl], %[t]\n\t"                                  \
      "s_bcnt1_i32_b64 %[n], %[m]\n\t"                                     \
      "s_add_u32 %[never], %[never], %[n]\n\t"                             \
      "s_and_b64 %[per], %[per], vcc\n\t"                                  \
      "s_andn2_b64 %[per], %[per], %[m]\n\t"                               \
      "s_cmp_eq_u64 %[per], 0\n\t"                                         \
      "s_cbranch_scc1 4" #K "f\n\t"                                        \
      "s_bcnt1_i32_b64 %[n], %[per]\n\t"                                   \
      "s_add_u32 %[never], %[never], %[n]\n\t"                             \
      "s_mov_b64 %[m], exec\n\t"                                           \
      "s_mov_b64 exec, %[per]\n\t"                                         \
      "v_add_co_u32 %[klo], vcc, %[klo], %[lrem" #K "]\n\t"                \
      "v_mov_b32 %[lrem" #K "], 0\n\t"                                     \
      "v_addc_co_u32 %[khi], vcc, 0, %[khi], vcc\n\t"                      \
      "s_andn2_b64 exec, %[m], %[per]\n\t"                                 \
      "s_branch 4" #K "f\n\t"                                              \
      "2" #K ":\n\t"                                                       \
      "v_mul_f32 %[t], %[invl], %[t]\n\t"                                  \
      "s_bcnt1_i32_b64 %[n], %[m]\n\t"                                     \
      "s_add_u32 %[never], %[never], %[n]\n\t"                             \
      "4" #K ":\n\t"                                                       \
      "v_rndne_f32 %[t], %[t]\n\t"                                         \
      "v_and_b32 %[t], %[kmask], %[t]\n\t"                                 \
      "v_cmpx_eq_u32_e32 vcc, 0, %[t]\n\t"                                 \
      "v_mov_b64 %[sr" #K "], %[r" #K "]\n\t"                              \
      "v_mov_b64 %[si" #K "], %[i" #K "]\n\t"                              \
      "3" #K ":\n\t"

// long_retire4: after a chunk, the four slots in one statement (draw_wave.hip's long_retire per slot, with this
// kernel's Q2 ring -- 320 entries, q2_ci 2560 bytes on -- and its iteration bookkeeping).  The lanes that push add
// what the stages have COUNTED for their orbit so far -- cntk - l_rem with cntk = max_iter + kChunk: HEAD + MID +
// every LONG chunk incl. this one -- to `counted`.  The reference executes k + 1 iterations for an orbit that escapes
// at index k, and exactly that many replay steps record it later (cudabrot.cu:336,363); so over all pushed orbits
// counted - replay steps  is what the sparse chunks counted beyond the escapes, and the kernel reports
// iterate_steps - (counted - replay steps): exact without any lane ever knowing the escape index of an orbit.
// pushed / never: orbits pushed to Q2 / retired as never escaping (ended + periodic); bad: lanes that escaped in a
// chunk below min_iter (none in the launches this kernel takes).
__device__ __forceinline__ void long_retire4(const Orbit (&o)[kSlots], double (&seen_r)[kSlots], double (&seen_i)[kSlots],
                                             int (&l_rem)[kSlots], uint32_t &skip_lo, uint32_t &skip_hi, uint32_t &counted,
                                             const unsigned long long (&ran)[kSlots], const unsigned long long (&esc)[kSlots],
                                             int accept_rem, uint32_t long_steps, uint32_t counted_base,
                                             uint32_t check_periodic, uint32_t q2_tail, uint32_t q2_lds, uint32_t &pushed,
                                             uint32_t &never, unsigned long long &bad) {
  static_assert(kQ2Cap == 320 && kSlots == 4, "ring length, plane distance and the four slot blocks");
  unsigned long long save, m, per;
  uint32_t slot, t, n;
  const float inv_chunk = 1.0f / (float) kChunk;
  const uint32_t low_mantissa = (1u << (24 - kBrentBits)) - 1u;
  q2_tail = __builtin_amdgcn_readfirstlane(q2_tail);
  asm volatile(
      "s_mov_b64 %[save], exec\n\t"
      "s_mov_b32 %[pushed], 0\n\t"
      "s_mov_b32 %[never], 0\n\t"
      "s_mov_b64 %[bad], 0\n\t"
      CBW_RETIRE_SLOT(0) CBW_RETIRE_SLOT(1) CBW_RETIRE_SLOT(2) CBW_RETIRE_SLOT(3)
      "s_mov_b64 exec, %[save]\n\t"
      "s_nop 4\n\t"
      : [sr0] "+v"(seen_r[0]), [si0] "+v"(seen_i[0]), [lrem0] "+v"(l_rem[0]), [sr1] "+v"(seen_r[1]), [si1] "+v"(seen_i[1]),
        [lrem1] "+v"(l_rem[1]), [sr2] "+v"(seen_r[2]), [si2] "+v"(seen_i[2]), [lrem2] "+v"(l_rem[2]), [sr3] "+v"(seen_r[3]),
        [si3] "+v"(seen_i[3]), [lrem3] "+v"(l_rem[3]), [klo] "+v"(skip_lo), [khi] "+v"(skip_hi), [acc] "+v"(counted),
        [tail2] "+s"(q2_tail), [pushed] "=&s"(pushed), [never] "=&s"(never), [bad] "=&s"(bad), [save] "=&s"(save),
        [m] "=&s"(m), [per] "=&s"(per), [n] "=&s"(n), [slot] "=&v"(slot), [t] "=&v"(t)
      : [ran0] "s"(ran[0]), [esc0] "s"(esc[0]), [ran1] "s"(ran[1]), [esc1] "s"(esc[1]), [ran2] "s"(ran[2]),
        [esc2] "s"(esc[2]), [ran3] "s"(ran[3]), [esc3] "s"(esc[3]), [thr] "s"(accept_rem), [ls] "s"(long_steps),
        [cntk] "s"(counted_base), [chkf] "s"(check_periodic), [invl] "s"(inv_chunk), [kmask] "s"(low_mantissa),
        [q2] "s"(q2_lds), [cr0] "v"(o[0].cr), [ci0] "v"(o[0].ci), [r0] "v"(o[0].r), [i0] "v"(o[0].i), [cr1] "v"(o[1].cr),
        [ci1] "v"(o[1].ci), [r1] "v"(o[1].r), [i1] "v"(o[1].i), [cr2] "v"(o[2].cr), [ci2] "v"(o[2].ci), [r2] "v"(o[2].r),
        [i2] "v"(o[2].i), [cr3] "v"(o[3].cr), [ci3] "v"(o[3].ci), [r3] "v"(o[3].r), [i3] "v"(o[3].i)
      : "vcc", "scc", "memory");
}

// ---- REPLAY burst, software-pipelined: IterateAndRecord (cudabrot.cu:347-365) into the pixel stream -------------
//
// A replaying lane holds z_n COMPUTED BUT NOT YET RECORDED (a fresh orbit: z_1, made when it is popped).  One
// iteration of the loop, for the lanes of `act` (EXEC):
//   record z_n     IncrementPixelCounter (cudabrot.cu:308-312): pixel of (R/2, I/2), the four bound tests as two
//                  unsigned 64-bit compares of the quotients' bit patterns, the hits compacted with v_mbcnt and
//                  stored side by side as row << 16 | col (draw_wave.hip, CB_REPLAY_*)
//   leave if |z_n|^2 > 4   (cudabrot.cu:363) -- after recording the escaped point, like the reference
//   z_{n+1} <- z_n^2 + c   (mandel_step2's six instructions, same order) and its |.|^2
// and the instructions of "record z_n" alternate with those of "z_{n+1}": the two depend on z_n only.  A lane that
// leaves has computed one point too many, which nothing reads.  17 vector instructions per step.
#define CBW_REPLAY_BIN_POW2                               \
  "v_fma_f64 %[fx], %[r], %[sx], %[ox]\n\t"               \
  "v_fma_f64 %[fy], %[i], %[sy], %[oy]\n\t"
#define CBW_DIV(q, num, den)                                        \
  "v_div_scale_f64 %[d0], %[scp], " den ", " den ", " num "\n\t"    \
  "v_rcp_f64 %[d2], %[d0]\n\t"                                      \
  "v_div_scale_f64 %[d1], vcc, " num ", " den ", " num "\n\t"       \
  "v_fma_f64 %[d3], -%[d0], %[d2], 1.0\n\t"                         \
  "v_fma_f64 %[d2], %[d2], %[d3], %[d2]\n\t"                        \
  "v_fma_f64 %[d3], -%[d0], %[d2], 1.0\n\t"                         \
  "v_fma_f64 %[d2], %[d2], %[d3], %[d2]\n\t"                        \
  "v_mul_f64 %[d3], %[d1], %[d2]\n\t"                               \
  "v_fma_f64 %[d0], -%[d0], %[d3], %[d1]\n\t"                       \
  "s_nop 1\n\t"                                                     \
  "v_div_fmas_f64 %[d0], %[d0], %[d2], %[d3]\n\t"                   \
  "v_div_fixup_f64 " q ", %[d0], " den ", " num "\n\t"
// (draw_wave.hip, CB_REPLAY_BIN_DIV: the estimate RN(a * RN(1/d)) decides unless its fraction is within 2^-24 of
// an integer; then the whole wave takes the IEEE division)
#define CBW_REPLAY_BIN_DIV CBW_REPLAY_BIN_DIV_L("4", "5")
#define CBW_REPLAY_BIN_DIV2 CBW_REPLAY_BIN_DIV_L("14", "15")
#define CBW_REPLAY_BIN_DIV_L(LA, LB)                      \
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
  CBW_DIV("%[fx]", "%[fx]", "%[sx]") CBW_DIV("%[fy]", "%[fy]", "%[sy]") \
  "s_branch 5f\n\t"                                       \
  "4:\n\t"                                                \
  "v_mov_b64 %[fx], %[d0]\n\t"                            \
  "v_mov_b64 %[fy], %[d1]\n\t"                            \
  "5:\n\t"
// The statement: refill, then the burst.
//   refill   the idle lanes (not in `pact`) pop c from Q2 -- ring slot (q2_head + rank) mod 320 at LDS byte address
//            q2_lds, q2_ci 2560 bytes on -- and make z_1 = c^2 + c (the first point the reference records,
//            cudabrot.cu:357-360); their start time is the wave's replay clock
//   burst    unless no lane replays, or fewer than kReplayMin do while Q2 is empty and the wave is not draining
//            (then the replay waits, its state stays in registers)
#define CBW_REPLAY_REFILL                                 \
  "s_mov_b64 %[save], exec\n\t"                           \
  "s_mov_b32 %[cs], 0\n\t"                                \
  "s_mov_b32 %[ch], 0\n\t"                                \
  "s_mov_b32 %[nn], 0\n\t"                                \
  "s_cmp_eq_u32 %[en], 0\n\t"                             \
  "s_cbranch_scc1 2f\n\t"                                 \
  "s_not_b64 vcc, %[act]\n\t"                             \
  "s_bcnt1_i32_b64 %[nn], vcc\n\t"                        \
  "s_min_u32 %[nn], %[nn], %[q2c]\n\t"                    \
  "s_cmp_eq_u32 %[nn], 0\n\t"                             \
  "s_cbranch_scc1 6f\n\t"                                 \
  "v_mbcnt_lo_u32_b32 %[pidx], vcc_lo, 0\n\t"             \
  "v_mbcnt_hi_u32_b32 %[pidx], vcc_hi, %[pidx]\n\t"       \
  "s_mov_b64 exec, vcc\n\t"                               \
  "v_cmpx_gt_u32_e32 vcc, %[nn], %[pidx]\n\t"   /* the first nn idle lanes */ \
  "v_add_u32 %[pidx], %[q2h], %[pidx]\n\t"      /* < 320 + 64 */ \
  "v_subrev_u32 %[e], 320, %[pidx]\n\t"                   \
  "v_min_u32 %[pidx], %[pidx], %[e]\n\t"                  \
  "v_lshl_add_u32 %[pidx], %[pidx], 3, %[q2]\n\t"         \
  "ds_read_b64 %[cr], %[pidx]\n\t"                        \
  "ds_read_b64 %[ci], %[pidx] offset:2560\n\t"            \
  "s_or_b64 %[act], %[act], exec\n\t"                     \
  "v_mov_b32 %[pst], %[clk]\n\t"                          \
  "s_waitcnt lgkmcnt(0)\n\t"                              \
  "v_mul_f64 %[a], %[ci], %[ci]\n\t"            /* z_1: mandel_step2 from z = c */ \
  "v_fma_f64 %[a], %[cr], %[cr], -%[a]\n\t"               \
  "v_fma_f64 %[i], " CB_AL "%[cr]" CB_AR ", " CB_AL "%[ci]" CB_AR ", %[ci]\n\t" \
  "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"                  \
  "6:\n\t"                                                \
  "s_bcnt1_i32_b64 %[t], %[act]\n\t"                      \
  "s_cmp_eq_u32 %[t], 0\n\t"                              \
  "s_cbranch_scc1 2f\n\t"                                 \
  "s_sub_u32 %[ctr], %[q2c], %[nn]\n\t"         /* fewer than minact lanes and Q2 empty: the replay waits */ \
  "s_cmp_lg_u32 %[ctr], 0\n\t"                          \
  "s_cselect_b32 %[ctr], 1, %[minact]\n\t"              \
  "s_cmp_ge_u32 %[t], %[ctr]\n\t"                       \
  "s_cbranch_scc0 2f\n\t"
#define CBW_REPLAY_HEAD                                   \
  "s_sub_u32 %[ctr], %[n], 1\n\t"       /* the loop below runs ctr + 1 = n steps */ \
  "s_mov_b32 %[ch], %[fill]\n\t"        /* (the hits of the burst: fill afterwards - fill before) */ \
  "s_mov_b64 exec, %[act]\n\t"                            \
  "v_mul_f64 %[a], %[r], %[r]\n\t"      /* |Z_n|^2 of the pending point, as the step that made it computed it */ \
  "v_fma_f64 %[a], %[i], %[i], %[a]\n\t"                  \
  "s_cmp_lg_u32 %[direct], 0\n\t"                         \
  "s_cbranch_scc1 7f\n\t"                                 \
  "1:\n\t"                                                \
  "s_bcnt1_i32_b64 %[t], %[act]\n\t"
// The loop's tail.  As few scalar instructions and taken branches as it can do with: at two waves per SIMD nothing
// hides them (one taken branch per step, the back edge).
#define CBW_REPLAY_STEP_COMMON                            \
  "s_add_u32 %[cs], %[cs], %[t]\n\t"                      \
  "v_cmp_nlt_f64_e64 %[alive], %[k16], %[a]\n\t"          \
  "v_mul_f64 %[a], %[i], %[i]\n\t"                        \
  "v_cmp_gt_u64_e64 %[hx], %[wb], %[fx]\n\t"              \
  "v_fma_f64 %[a], %[r], %[r], -%[a]\n\t"                 \
  "v_cmp_gt_u64_e64 vcc, %[hb], %[fy]\n\t"                \
  "v_fma_f64 %[i], " CB_AL "%[r]" CB_AR ", " CB_AL "%[i]" CB_AR ", %[ci]\n\t" \
  "v_cvt_i32_f64 %[col], %[fx]\n\t"                       \
  "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"                  \
  "v_cvt_i32_f64 %[row], %[fy]\n\t"
// Cache policy of the stream's stores (gfx942+ bits: sc0, sc1, nt), as text behind the instruction.  The stream is
// written once and read once, a launch later.
#ifndef CB_STREAM_STORE_POLICY_ID  // (a number, so that a sweep can pass it through make: tools/gpu_define_sweep.sh)
#define CB_STREAM_STORE_POLICY_ID 0
#endif
#if CB_STREAM_STORE_POLICY_ID == 0
#define CB_STREAM_STORE_POLICY ""
#elif CB_STREAM_STORE_POLICY_ID == 1
#define CB_STREAM_STORE_POLICY " nt"
#elif CB_STREAM_STORE_POLICY_ID == 2
#define CB_STREAM_STORE_POLICY " sc1"
#elif CB_STREAM_STORE_POLICY_ID == 3
#define CB_STREAM_STORE_POLICY " sc0 sc1"
#elif CB_STREAM_STORE_POLICY_ID == 4
#define CB_STREAM_STORE_POLICY " sc0 sc1 nt"
#elif CB_STREAM_STORE_POLICY_ID == 5
#define CB_STREAM_STORE_POLICY " sc0"
#elif CB_STREAM_STORE_POLICY_ID == 6
#define CB_STREAM_STORE_POLICY " sc1 nt"
#endif
#ifdef CB_EXPERIMENT_PACKED24  /* timing only: 3-byte entries, two stores per step (the sort does not read them) */
#define CBW_REPLAY_WORD "v_lshl_or_b32 %[e], %[row], 12, %[col]\n\t"
#define CBW_REPLAY_PLACE "v_add_u32 %[pidx], %[pidx], %[fill]\n\t" "v_mul_u32_u24 %[pidx], 3, %[pidx]\n\t"
#define CBW_REPLAY_STORE "global_store_short %[pidx], %[e], %[base]\n\t" "global_store_byte_d16_hi %[pidx], %[e], %[base] offset:2\n\t"
#elif defined(CB_EXPERIMENT_NO_STREAM_STORE)  /* timing only: the replay without its store (one-level streams) */
#define CBW_REPLAY_WORD "v_lshl_or_b32 %[e], %[row], 16, %[col]\n\t"
#define CBW_REPLAY_PLACE "v_add_lshl_u32 %[pidx], %[pidx], %[fill], 2\n\t"
#define CBW_REPLAY_STORE ""
#else
#define CBW_REPLAY_WORD "v_lshl_or_b32 %[e], %[row], 16, %[col]\n\t"
#define CBW_REPLAY_PLACE "v_add_lshl_u32 %[pidx], %[pidx], %[fill], 2\n\t"
#define CBW_REPLAY_STORE "global_store_dword %[pidx], %[e], %[base]" CB_STREAM_STORE_POLICY "\n\t"
#endif
#define CBW_REPLAY_LOOP                                   \
  CBW_REPLAY_STEP_COMMON                                  \
  "s_and_b64 vcc, vcc, %[hx]\n\t"                         \
  "v_mul_f64 %[a], %[r], %[r]\n\t"                        \
  CBW_REPLAY_WORD                                         \
  "v_mbcnt_lo_u32_b32 %[pidx], vcc_lo, 0\n\t"             \
  "v_fma_f64 %[a], %[i], %[i], %[a]\n\t"                  \
  "v_mbcnt_hi_u32_b32 %[pidx], vcc_hi, %[pidx]\n\t"       \
  "s_bcnt1_i32_b64 %[t], vcc\n\t"                         \
  CBW_REPLAY_PLACE                                        \
  "s_and_b64 %[act], %[act], %[alive]\n\t"                \
  "s_mov_b64 exec, vcc\n\t"                               \
  CBW_REPLAY_STORE                                        \
  "s_add_u32 %[fill], %[fill], %[t]\n\t"                  \
  "s_mov_b64 exec, %[act]\n\t"                            \
  "s_cbranch_execz 8f\n\t"              /* (not taken but once) */ \
  "s_sub_u32 %[ctr], %[ctr], 1\n\t"     /* scc = borrow: the count had reached 0 */ \
  "s_cbranch_scc0 1b\n\t"                                 \
  "s_add_u32 %[clk], %[clk], %[n]\n\t"  /* the clock moves on by the steps made: all n ... */ \
  "s_branch 18f\n\t"                                      \
  "8:\n\t"                                                \
  "s_sub_u32 %[t], %[n], %[ctr]\n\t"    /* ... or, the last lane gone in step j = n - ctr, that many */ \
  "s_add_u32 %[clk], %[clk], %[t]\n\t"                    \
  "18:\n\t"                                               \
  "s_sub_u32 %[ch], %[fill], %[ch]\n\t"
// The loop on a CHUNKED stream (BinLayout::chunked: canvases beyond 1024 tiles; draw_wave.hip, CB_REPLAY_CHUNKED): the word
// goes to the current chunk of its GROUP of 1024 tiles -- group = tile >> 10; the group's {next word, end of chunk} sit
// in the wave's LDS at gcl + 8 * group; a returning add takes the place.  The step is ordered so that the LDS round
// trip has the six fp64 instructions of z_{n+1} to hide behind (two waves per SIMD: nothing else would): record first
// -- pixel, bounds, word, group, the two LDS instructions -- then z_{n+1}, then the store.  Lanes that find their
// chunk full (`over`) keep word, group, place and limit in e / pidx / pos / lim; the burst ends behind that step and
// replay_stage's caller opens new chunks for them (wide_open_chunks).  `fill` only counts the hits here.  23 vector
// instructions per step.
#define CBW_REPLAY_LOOP_CHUNKED                           \
  "s_add_u32 %[cs], %[cs], %[t]\n\t"                      \
  "v_cmp_nlt_f64_e64 %[alive], %[k16], %[a]\n\t"          \
  "v_cmp_gt_u64_e64 %[hx], %[wb], %[fx]\n\t"              \
  "v_cmp_gt_u64_e64 vcc, %[hb], %[fy]\n\t"                \
  "v_cvt_i32_f64 %[col], %[fx]\n\t"                       \
  "v_cvt_i32_f64 %[row], %[fy]\n\t"                       \
  "v_mul_f64 %[a], %[i], %[i]\n\t"                        \
  "v_mov_b32 %[lim], 1\n\t"                               \
  "s_and_b64 vcc, vcc, %[hx]\n\t"                         \
  "v_lshl_or_b32 %[e], %[row], 16, %[col]\n\t"            \
  "s_mov_b64 %[hit], vcc\n\t"                             \
  "s_mov_b64 exec, vcc\n\t"                               \
  "v_lshrrev_b32 %[row], 7, %[row]\n\t"                   \
  "v_lshrrev_b32 %[col], 7, %[col]\n\t"                   \
  "v_mad_u32_u24 %[row], %[row], %[tlx], %[col]\n\t"      \
  "v_lshrrev_b32 %[pidx], 10, %[row]\n\t"                 \
  "v_lshl_add_u32 %[col], %[pidx], 3, %[gcl]\n\t"         \
  "ds_add_rtn_u32 %[pos], %[col], %[lim]\n\t"             \
  "ds_read_b32 %[lim], %[col] offset:4\n\t"               \
  "s_bcnt1_i32_b64 %[t], vcc\n\t"                         \
  "s_mov_b64 exec, %[act]\n\t"                            \
  "v_fma_f64 %[a], %[r], %[r], -%[a]\n\t"                 \
  "v_fma_f64 %[i], " CB_AL "%[r]" CB_AR ", " CB_AL "%[i]" CB_AR ", %[ci]\n\t" \
  "v_fma_f64 %[r], %[a], 0.5, %[cr]\n\t"                  \
  "v_mul_f64 %[a], %[r], %[r]\n\t"                        \
  "v_fma_f64 %[a], %[i], %[i], %[a]\n\t"                  \
  "s_and_b64 %[act], %[act], %[alive]\n\t"                \
  "s_add_u32 %[fill], %[fill], %[t]\n\t"                  \
  "s_mov_b64 exec, %[hit]\n\t"                            \
  "s_waitcnt lgkmcnt(0)\n\t"                              \
  "v_cmp_lt_u32_e32 vcc, %[pos], %[lim]\n\t"              \
  "s_andn2_b64 %[over], exec, vcc\n\t"                    \
  "s_mov_b64 exec, vcc\n\t"                               \
  "v_lshlrev_b32 %[pos], 2, %[pos]\n\t"                   \
  "global_store_dword %[pos], %[e], %[base]" CB_STREAM_STORE_POLICY "\n\t" \
  "s_cmp_lg_u64 %[over], 0\n\t"                           \
  "s_cbranch_scc1 8f\n\t"               /* (the step is complete: it counts) */ \
  "s_mov_b64 exec, %[act]\n\t"                            \
  "s_cbranch_execz 8f\n\t"                                \
  "s_sub_u32 %[ctr], %[ctr], 1\n\t"                       \
  "s_cbranch_scc0 1b\n\t"                                 \
  "s_add_u32 %[clk], %[clk], %[n]\n\t"                    \
  "s_branch 18f\n\t"                                      \
  "8:\n\t"                                                \
  "s_sub_u32 %[t], %[n], %[ctr]\n\t"                      \
  "s_add_u32 %[clk], %[clk], %[t]\n\t"                    \
  "18:\n\t"                                               \
  "s_sub_u32 %[ch], %[fill], %[ch]\n\t"
// `direct` != 0 (the wave's stream region is full; rare): the same loop with the hits added to the histogram by
// device-scope atomics (a one-level canvas has at most 2^24 pixels: the byte offset of a pixel fits 32 bits).  BIN:
// the text that forms fx, fy (a second copy of it).
#define CBW_REPLAY_DIRECT(BIN)                            \
  "s_branch 9f\n\t"                                       \
  "7:\n\t"                                                \
  "s_mov_b32 %[ch], 0\n\t"                                \
  "3:\n\t"                                                \
  "s_bcnt1_i32_b64 %[t], %[act]\n\t"                      \
  BIN                                                     \
  CBW_REPLAY_STEP_COMMON                                  \
  "v_mad_u32_u24 %[pidx], %[row], %[wi], %[col]\n\t"      \
  "s_and_b64 vcc, vcc, %[hx]\n\t"                         \
  "v_mul_f64 %[a], %[r], %[r]\n\t"                        \
  "v_lshlrev_b32 %[pidx], 3, %[pidx]\n\t"                 \
  "v_fma_f64 %[a], %[i], %[i], %[a]\n\t"                  \
  "v_mov_b64 %[fx], 1\n\t"             /* the 64-bit one (fx is free: its compares are done) */ \
  "s_bcnt1_i32_b64 %[t], vcc\n\t"                         \
  "s_and_b64 %[act], %[act], %[alive]\n\t"                \
  "s_mov_b64 exec, vcc\n\t"                               \
  "global_atomic_add_x2 %[pidx], %[fx], %[hist]\n\t"      \
  "s_add_u32 %[ch], %[ch], %[t]\n\t"                      \
  "s_mov_b64 exec, %[act]\n\t"                            \
  "s_cbranch_execz 19f\n\t"                               \
  "s_sub_u32 %[ctr], %[ctr], 1\n\t"                       \
  "s_cbranch_scc0 3b\n\t"                                 \
  "s_add_u32 %[clk], %[clk], %[n]\n\t"                    \
  "s_branch 9f\n\t"                                       \
  "19:\n\t"                                               \
  "s_sub_u32 %[t], %[n], %[ctr]\n\t"                      \
  "s_add_u32 %[clk], %[clk], %[t]\n\t"                    \
  "9:\n\t"
// behind either loop: the way out that the skipped forms share
#define CBW_REPLAY_END                                    \
  "2:\n\t"                                                \
  "s_mov_b64 exec, %[save]\n\t"                           \
  "s_nop 4\n\t"

// The REPLAY stage's statement (see above; executed in every iteration of the scheduler, `enable` = 0 skips it).
// act: the lanes with an orbit in flight; q2_head / q2_count: the ring; min_active: fewest lanes a burst is run
// for; n_steps: steps of a burst (>= 1); the stream region must have room for 64 * n_steps more entries unless
// `direct`.  Returns the orbits popped; lane_steps / hits: the executed lane-steps and the points recorded.
// kChunked: the words that found their chunk full (lanes of `over`: word, group, the place they took, the chunk's end)
struct ChunkOverflow {
  unsigned long long over;
  uint32_t word, group, pos, lim;
};
template <bool kPow2, bool kChunked>
__device__ __forceinline__ uint32_t replay_stage(uint32_t enable, unsigned long long &act, uint32_t q2_head,
                                                 uint32_t q2_count, uint32_t q2_lds, uint32_t min_active,
                                                 uint32_t n_steps, uint32_t direct, Orbit &p, uint32_t &p_start,
                                                 uint32_t *region, uint32_t &fill, uint32_t &clock,
                                                 uint32_t &lane_steps, uint32_t &hits, uint32_t cursors_lds,
                                                 ChunkOverflow &ovf) {
  static_assert(kQ2Cap == 320, "ring length and plane distance in CBW_REPLAY_REFILL");
  unsigned long long save, alive, hx, scp;
  uint32_t cs, ch, ctr, t, nn;
  double a, fx, fy, d0, d1, d2, d3, ox, oy;
  uint32_t pidx, e, col, row;
  unsigned long long over = 0ull, hit;
  uint32_t pos = 0u, lim = 0u;
  const uint32_t tlx = kChunked ? kernel_args()->bin.tiles_x : 0u;
  cursors_lds = __builtin_amdgcn_readfirstlane(cursors_lds);
  const KernelArgs ka = kernel_args();
  const double wb = ka->replay_bound_w, hb = ka->replay_bound_h;  // (double) w, h: the bounds of the quotients, compared as bit patterns
  const uint32_t wi = (uint32_t) ka->w;
  const unsigned long long hist = reinterpret_cast<unsigned long long>(ka->hist);
  region = reinterpret_cast<uint32_t *>(uniform_u64(reinterpret_cast<unsigned long long>(region)));
  enable = __builtin_amdgcn_readfirstlane(enable);
  act = uniform_u64(act);
  fill = __builtin_amdgcn_readfirstlane(fill);
  clock = __builtin_amdgcn_readfirstlane(clock);
  n_steps = __builtin_amdgcn_readfirstlane(n_steps);
  q2_head = __builtin_amdgcn_readfirstlane(q2_head);
  q2_count = __builtin_amdgcn_readfirstlane(q2_count);
  min_active = __builtin_amdgcn_readfirstlane(min_active);
  direct = __builtin_amdgcn_readfirstlane(direct);
  const double k16 = 16.0;
  const double sx = ka->replay_scale_real, sy = ka->replay_scale_imag;
  const double oxs = ka->replay_offset_real, oys = ka->replay_offset_imag;
#define CBW_REPLAY_OUT                                                                                             \
  [r] "+v"(p.r), [i] "+v"(p.i), [cr] "+v"(p.cr), [ci] "+v"(p.ci), [pst] "+v"(p_start), [act] "+s"(act),            \
      [fill] "+s"(fill), [clk] "+s"(clock), [save] "=&s"(save), [alive] "=&s"(alive), [hx] "=&s"(hx),              \
      [cs] "=&s"(cs), [ch] "=&s"(ch), [ctr] "=&s"(ctr), [t] "=&s"(t), [nn] "=&s"(nn), [a] "=&v"(a), [fx] "=&v"(fx), \
      [fy] "=&v"(fy), [e] "=&v"(e), [col] "=&v"(col), [row] "=&v"(row), [pidx] "=&v"(pidx)
#define CBW_REPLAY_IN                                                                                              \
  [en] "s"(enable), [n] "s"(n_steps), [q2h] "s"(q2_head), [q2c] "s"(q2_count), [q2] "s"(q2_lds),                   \
      [minact] "s"(min_active), [direct] "s"(direct), [sx] "s"(sx), [sy] "s"(sy), [wb] "s"(wb), [hb] "s"(hb),      \
      [wi] "s"(wi), [hist] "s"(hist), [base] "s"(region), [k16] "s"(k16)
  (void) ox;
  (void) oy;
#define CBW_CHUNKED_OUT [over] "+s"(over), [hit] "=&s"(hit), [pos] "+v"(pos), [lim] "+v"(lim)
#define CBW_CHUNKED_IN [tlx] "s"(tlx), [gcl] "s"(cursors_lds)
  if (kPow2 && kChunked) {
    asm volatile("v_mov_b64 %[ox], %[oxs]\n\t"
                 "v_mov_b64 %[oy], %[oys]\n\t"
                 CBW_REPLAY_REFILL CBW_REPLAY_HEAD CBW_REPLAY_BIN_POW2 CBW_REPLAY_LOOP_CHUNKED
                 CBW_REPLAY_DIRECT(CBW_REPLAY_BIN_POW2) CBW_REPLAY_END
                 : CBW_REPLAY_OUT, [ox] "=&v"(ox), [oy] "=&v"(oy), CBW_CHUNKED_OUT
                 : CBW_REPLAY_IN, [oxs] "s"(oxs), [oys] "s"(oys), CBW_CHUNKED_IN
                 : "vcc", "scc", "memory");
  } else if (kPow2) {
    asm volatile("v_mov_b64 %[ox], %[oxs]\n\t"  // (in every lane: EXEC is all ones here, not behind the refill)
                 "v_mov_b64 %[oy], %[oys]\n\t"
                 CBW_REPLAY_REFILL CBW_REPLAY_HEAD CBW_REPLAY_BIN_POW2 CBW_REPLAY_LOOP
                 CBW_REPLAY_DIRECT(CBW_REPLAY_BIN_POW2) CBW_REPLAY_END
                 : CBW_REPLAY_OUT, [ox] "=&v"(ox), [oy] "=&v"(oy)
                 : CBW_REPLAY_IN, [oxs] "s"(oxs), [oys] "s"(oys)
                 : "vcc", "scc", "memory");
  } else if (kChunked) {
    const double rx = ka->rcp_delta_real, ry = ka->rcp_delta_imag;
    const double kg = 0.5 - 0x1p-24;
    asm volatile(CBW_REPLAY_REFILL CBW_REPLAY_HEAD CBW_REPLAY_BIN_DIV CBW_REPLAY_LOOP_CHUNKED
                 CBW_REPLAY_DIRECT(CBW_REPLAY_BIN_DIV2) CBW_REPLAY_END
                 : CBW_REPLAY_OUT, [scp] "=&s"(scp), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3),
                   CBW_CHUNKED_OUT
                 : CBW_REPLAY_IN, [ox] "s"(oxs), [oy] "s"(oys), [rx] "s"(rx), [ry] "s"(ry), [kg] "s"(kg), CBW_CHUNKED_IN
                 : "vcc", "scc", "memory");
  } else {
    const double rx = ka->rcp_delta_real, ry = ka->rcp_delta_imag;
    const double kg = 0.5 - 0x1p-24;
    asm volatile(CBW_REPLAY_REFILL CBW_REPLAY_HEAD CBW_REPLAY_BIN_DIV CBW_REPLAY_LOOP
                 CBW_REPLAY_DIRECT(CBW_REPLAY_BIN_DIV2) CBW_REPLAY_END
                 : CBW_REPLAY_OUT, [scp] "=&s"(scp), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
                 : CBW_REPLAY_IN, [ox] "s"(oxs), [oy] "s"(oys), [rx] "s"(rx), [ry] "s"(ry), [kg] "s"(kg)
                 : "vcc", "scc", "memory");
  }
#undef CBW_CHUNKED_OUT
#undef CBW_CHUNKED_IN
  if (kChunked) {
    ovf.over = uniform_u64(over);
    ovf.word = e;
    ovf.group = pidx;
    ovf.pos = pos;
    ovf.lim = lim;
  }
#undef CBW_REPLAY_OUT
#undef CBW_REPLAY_IN
  lane_steps = cs;
  hits = ch;
  return nn;
}

// The last, shorter chunk of the orbits of one slot (the lanes of `mask`; none: the statement does nothing): exactly
// n_steps iterations with the reference's test after each (cudabrot.cu:326-337).  Every escape of the LONG stage is
// accepted, so the escaped lanes push c to Q2 (ring slot (q2_tail + rank) mod 320 at q2_lds); all lanes of `mask`
// end idle (l_rem = 0).  Returns the escaped lanes; lane_steps: the executed lane-steps.
__device__ __forceinline__ unsigned long long long_tail(unsigned long long mask, uint32_t n_steps, Orbit &o, int &l_rem,
                                                        uint32_t q2_tail, uint32_t q2_lds, uint32_t &lane_steps) {
  unsigned long long save, escaped;
  uint32_t cnt, tmp, ctr, slot, t;
  double a;
  const double k16 = 16.0;
  asm volatile(
      "s_mov_b64 %[save], exec\n\t"
      "s_mov_b32 %[cnt], 0\n\t"
      "s_mov_b64 %[esc], 0\n\t"
      "s_cmp_eq_u64 %[mask], 0\n\t"
      "s_cbranch_scc1 9f\n\t"
      "s_mov_b64 exec, %[mask]\n\t"
      "s_mov_b32 %[ctr], %[n]\n\t"
      "1:\n\t"
      CB_STEP
      "s_cbranch_execz 2f\n\t"
      "s_sub_u32 %[ctr], %[ctr], 1\n\t"
      "s_cmp_lg_u32 %[ctr], 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "2:\n\t"
      "s_andn2_b64 %[esc], %[mask], exec\n\t"
      "s_mov_b64 exec, %[esc]\n\t"
      "s_cbranch_execz 3f\n\t"
      "v_mbcnt_lo_u32_b32 %[slot], exec_lo, 0\n\t"
      "v_mbcnt_hi_u32_b32 %[slot], exec_hi, %[slot]\n\t"
      "v_add_u32 %[slot], %[tail2], %[slot]\n\t"
      "v_subrev_u32 %[t], 320, %[slot]\n\t"
      "v_min_u32 %[slot], %[slot], %[t]\n\t"
      "v_lshl_add_u32 %[slot], %[slot], 3, %[q2]\n\t"
      "ds_write_b64 %[slot], %[cr]\n\t"
      "ds_write_b64 %[slot], %[ci] offset:2560\n\t"
      "3:\n\t"
      "s_mov_b64 exec, %[mask]\n\t"
      "v_mov_b32 %[lrem], 0\n\t"
      "9:\n\t"
      "s_mov_b64 exec, %[save]\n\t"
      "s_nop 4\n\t"
      : [r] "+v"(o.r), [i] "+v"(o.i), [lrem] "+v"(l_rem), [a] "=&v"(a), [save] "=&s"(save), [esc] "=&s"(escaped),
        [cnt] "=&s"(cnt), [tmp] "=&s"(tmp), [ctr] "=&s"(ctr), [slot] "=&v"(slot), [t] "=&v"(t)
      : [mask] "s"(mask), [n] "s"(n_steps), [cr] "v"(o.cr), [ci] "v"(o.ci), [k16] "s"(k16), [tail2] "s"(q2_tail),
        [q2] "s"(q2_lds)
      : "vcc", "scc", "memory");
  lane_steps = cnt;
  return escaped;
}

// kChunked, behind a burst that ended with words left over (draw_wave.hip, replay_burst_chunked): group by group, open
// the next free chunk of the wave's segment, put the words at its start -- the places they took run on from the old
// chunk's end, lim -- and the group's cursor behind them.  Once per kChunkWords words of a group.
__device__ __forceinline__ void wide_open_chunks(const ChunkOverflow &v, uint32_t *region, uint32_t *cursors,
                                                 uint32_t &next_chunk, uint32_t *desc) {
  unsigned long long over = v.over;
  asm volatile("" : "+s"(region), "+s"(desc));
  while (over != 0ull) {
    const int lane0 = __ffsll((long long) over) - 1;
    const uint32_t g0 = (uint32_t) __builtin_amdgcn_readlane((int) v.group, lane0);
    const uint32_t lim0 = (uint32_t) __builtin_amdgcn_readlane((int) v.lim, lane0);
    const bool mine = lane_in(over) && v.group == g0;
    const unsigned long long same = uniform_u64(__ballot(mine));
    const uint32_t first = next_chunk * kChunkWords;
    if (mine) region[first + (v.pos - lim0)] = v.word;
    if (lane_id() == lane0) {
      desc[next_chunk] = (g0 << 16) | kChunkWords;  // taken as full; the launch's end corrects the last one of each group
      cursors[2u * g0] = first + (uint32_t) __popcll(same);
      cursors[2u * g0 + 1u] = first + kChunkWords;
    }
    next_chunk++;
    over &= ~same;
  }
}

__device__ __forceinline__ int q1_wrap(int slot) { return slot >= kQ1Cap ? slot - kQ1Cap : slot; }
__device__ __forceinline__ int q2_wrap(int slot) { return slot >= kQ2Cap ? slot - kQ2Cap : slot; }

// kPow2: both pixel sides are powers of two (the pixel of a point is one exact fma); else the guarded reciprocal with
// the IEEE division behind it (CBW_REPLAY_BIN_DIV) -- an instance of its own, so that the division's temporaries do
// not count against the other's registers.
// kTimed: stage clocks (s_memtime) into cb_counters, like draw_wave_kernel's timed variant (--kernel timed).
// kChunked: the stream chunked by group of 1024 tiles as it is written (canvases beyond 1024 tiles).
template <bool kPow2, bool kTimed, bool kChunked>
__global__ void __launch_bounds__(64 * kWavesPerBlock, 4)  // at most 128 vector registers: two of these waves and the
draw_wide_kernel(DrawArgs a) {                            // scatter's four (64 each) share a SIMD's 512
  __shared__ WideQueues queues[kWavesPerBlock];
  WideQueues &q = queues[threadIdx.x >> 6];
  // kChunked: the groups' cursors, 512 bytes per wave of DYNAMIC LDS (launch_draw_wide): as static LDS they would
  // take the workgroup beyond a quarter of the CU's 160 KiB, and the compiler, seeing room for three waves per SIMD
  // only, would help itself to more than 128 registers
  extern __shared__ uint32_t group_cursors[];
  uint32_t *const my_cursors = group_cursors + (kChunked ? (threadIdx.x >> 6) * 2u * kChunkedGroupsMax : 0u);
  if (kChunked) {
    for (uint32_t k = threadIdx.x & 63u; k < 2u * kChunkedGroupsMax; k += 64u) my_cursors[k] = 0u;  // no chunk yet
  }
  const uint32_t cursors_lds = kChunked ? __builtin_amdgcn_readfirstlane(
      (uint32_t) reinterpret_cast<uintptr_t>(static_cast<void *>(my_cursors))) : 0u;
  uint32_t next_chunk = 0;  // kChunked: the first chunk of the wave's (double) segment that is still free

  // this wave: subsequences [128 wave_id, 128 wave_id + 128): generator A of lane l is subsequence 128 wave_id + l,
  // generator B 128 wave_id + 64 + l; stream segments 2 wave_id and 2 wave_id + 1 of the workspace (contiguous)
  const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  uint32_t *const region = a.bin.stream + (size_t) (2u * wave_id) * a.bin.cap;
  const uint32_t region_cap = 2u * a.bin.cap;
  uint32_t region_fill = 0;
  const uint32_t tid_a = wave_id * 128u + (threadIdx.x & 63u), tid_b = tid_a + 64u;

  const int max_iter = a.max_iter;
  const int long_start = a.head_steps + a.mid_steps;
  const int tail_steps = (max_iter - long_start) % kChunk;

  Xorwow2 rng;
  rng.a = load_rng(a.states, a.n_threads, tid_a);
  rng.b = load_rng(a.states, a.n_threads, tid_b);

  // wave-uniform scheduler state and statistics (scalar registers)
  uint32_t halves_left = 2u * a.samples_per_thread;  // draws still to make (A and B alternately)
  uint32_t hs = 0;                                   // head_bodies' state
  uint32_t pending = 0;                              // 1: a drawn sample waits for its test (in its generator's words: CBW_PENDING)
  int q0_head = 0, q0_count = 0;
  int q1_head = 0, q1_count = 0;
  int q2_head = 0, q2_count = 0;
  unsigned long long n_rejected = 0, n_never = 0, n_too_fast = 0, n_recorded = 0, n_iterate = 0,
                     n_replay = 0, n_incr = 0, status = 0;
  const uint32_t wave_slot = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11));  // HW_REG_HW_ID bits 3:0
  uint32_t long_chunks = 0;
  unsigned long long t_head = 0, t_mid = 0, t_long = 0, t_replay = 0;
#ifdef CB_WIDE_PROBE  // (tools/wide_stage_probe.py: units of work per stage beside the stage clocks)
  unsigned long long probe_chunks = 0, probe_lane_chunks = 0, probe_replay_steps = 0, probe_bursts = 0;
  unsigned long long probe_lanes_at_start = 0, probe_lanes_at_end = 0;
#endif
  const unsigned long long t_start = kTimed ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rt_start = kTimed ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const uint32_t q0_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t) reinterpret_cast<uintptr_t>(static_cast<void *>(&q.q0_cr[0])));
  const uint32_t q1_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t) reinterpret_cast<uintptr_t>(static_cast<void *>(&q.q1_cr[0])));
  const uint32_t q2_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t) reinterpret_cast<uintptr_t>(static_cast<void *>(&q.q2_cr[0])));
  // LONG lane state: four orbits per lane
  Orbit lo[kSlots] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double seen_r[kSlots] = {0, 0, 0, 0}, seen_i[kSlots] = {0, 0, 0, 0};  // periodicity check
  int l_rem[kSlots] = {0, 0, 0, 0};   // iterations left before max_iter; 0 = idle
  uint32_t skip_lo = 0, skip_hi = 0;  // per lane, 64 bits: iterations the periodicity check made unnecessary
  uint32_t counted = 0;               // per lane: iterations counted for the orbits this lane pushed (long_retire)
  // REPLAY lane state: z_n computed, not yet recorded
  Orbit po = {0, 0, 0, 0};
  unsigned long long pact = 0ull;     // lanes with a replay in flight (wave-uniform mask)
  uint32_t p_start = 0;               // the wave's replay clock when this lane's orbit was popped
  uint32_t replay_clock = 0;
  unsigned long long counted_s = 0;   // iterations counted for the orbits pushed by long_tail (see long_retire)
  unsigned long long skipped_s = 0;   // iterations not made for the samples the interior map retired (mid_pass)

  // Carry-over: pick up the queues and orbit slots the previous launch left behind (DrawArgs::carry).
  unsigned long long *const carry = a.carry + (size_t) wave_id * (2u * kCarryWordsPerWave);
  // Progress board (kernels.h, kSchedWords; draw_wave.hip): the waves that share a SIMD post how many draws they
  // still have to make and take s_setprio from their rank -- the one furthest behind issues first.
  uint32_t *board_row;
  {
    const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID
    const uint32_t key = ((xcc & 0xfu) << 12) | ((hw >> 4) & 0xfffu);                  // SIMD, pipe, CU, SH, SE
    board_row = reinterpret_cast<uint32_t *>(a.carry + (size_t) gridDim.x * kWavesPerBlock * (2u * kCarryWordsPerWave)) +
                (size_t) key * 16u;
  }
  auto post_progress_and_set_priority = [&](uint32_t still_to_draw) {
    const uint32_t mine = still_to_draw + 1u;  // 0 = no wave in this slot
    const uint32_t lane = (uint32_t) lane_id();
    uint32_t *row = board_row;
    asm volatile("" : "+s"(row));  // (the lanes' addresses are formed here, not kept in registers across the launch)
    if (lane == 0u) {
      __hip_atomic_store(row + (wave_slot & 15u), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t other = 0u;
    if (lane < 16u) other = __hip_atomic_load(row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool before_me = (lane < 16u) && (lane != (wave_slot & 15u)) &&
                           (other > mine || (other == mine && lane < (wave_slot & 15u)));
    const int rank = __popcll(__ballot(before_me));
    // (the scatter's waves beside these: scatter.hip, CB_SCATTER_PRIO)
    if (rank == 0) {
      __builtin_amdgcn_s_setprio(CB_WIDE_PRIO_BEHIND);
    } else {
      __builtin_amdgcn_s_setprio(CB_WIDE_PRIO_AHEAD);
    }
  };
  const uint32_t keep_rest = __builtin_amdgcn_readfirstlane(a.drain ? 1u : 0u);  // 0: leave in-flight work to the next launch
  post_progress_and_set_priority(halves_left);
  if (carry[0] != 0ull && carry[0] != 2ull) status |= CB_STATUS_CARRY_FOREIGN;  // draw_wave_kernel's: not ours to resume
  if (carry[0] == 2ull) {  // wave-uniform: the header is one address (2: a record of this kernel)
    q0_head = (int) __builtin_amdgcn_readfirstlane((uint32_t) carry[1]);
    q0_count = (int) __builtin_amdgcn_readfirstlane((uint32_t) (carry[1] >> 32));
    q1_head = (int) __builtin_amdgcn_readfirstlane((uint32_t) carry[2]);
    q1_count = (int) __builtin_amdgcn_readfirstlane((uint32_t) (carry[2] >> 32));
    q2_head = (int) __builtin_amdgcn_readfirstlane((uint32_t) carry[3]);
    q2_count = (int) __builtin_amdgcn_readfirstlane((uint32_t) (carry[3] >> 32));
    replay_clock = __builtin_amdgcn_readfirstlane((uint32_t) carry[4]);
    unsigned long long *lds_words = reinterpret_cast<unsigned long long *>(&q);
    const unsigned long long *img = carry + kCarryHeaderWords;
    for (uint32_t k = lane_id(); k < kWideQueueWords; k += 64u) lds_words[k] = img[k];
    const unsigned long long *pl = carry + lane_id();
#pragma unroll
    for (int o = 0; o < kSlots; ++o) {
      lo[o].cr = __longlong_as_double((long long) pl[wide_plane(o * 6 + 0)]);
      lo[o].ci = __longlong_as_double((long long) pl[wide_plane(o * 6 + 1)]);
      lo[o].r = __longlong_as_double((long long) pl[wide_plane(o * 6 + 2)]);
      lo[o].i = __longlong_as_double((long long) pl[wide_plane(o * 6 + 3)]);
      seen_r[o] = __longlong_as_double((long long) pl[wide_plane(o * 6 + 4)]);
      seen_i[o] = __longlong_as_double((long long) pl[wide_plane(o * 6 + 5)]);
    }
    l_rem[0] = (int) (uint32_t) pl[wide_plane(24)];
    l_rem[1] = (int) (uint32_t) (pl[wide_plane(24)] >> 32);
    l_rem[2] = (int) (uint32_t) pl[wide_plane(25)];
    l_rem[3] = (int) (uint32_t) (pl[wide_plane(25)] >> 32);
    po.cr = __longlong_as_double((long long) pl[wide_plane(26)]);
    po.ci = __longlong_as_double((long long) pl[wide_plane(27)]);
    po.r = __longlong_as_double((long long) pl[wide_plane(28)]);
    po.i = __longlong_as_double((long long) pl[wide_plane(29)]);
    p_start = (uint32_t) pl[wide_plane(30)];
    pact = __ballot((pl[wide_plane(30)] >> 32) != 0ull);
  }

  // The launch's first sample: generator A, drawn in logical order, its words then rotated the way the bodies expect
  // them after one sample (outside the loop: the generator's registers are touched by head_bodies alone in there).
  if (halves_left != 0u) {
    (void) xorwow_next(rng.a);  // cudabrot.cu:392-393: four outputs; they stay in the generator's words (CBW_PENDING)
    (void) xorwow_next(rng.a);
    (void) xorwow_next(rng.a);
    (void) xorwow_next(rng.a);
    rng.a = xorwow_rotated4(rng.a);
    halves_left--;
    hs = 1;
    pending = 1u;
  }

  // Everything loaded so far (generators, carried state) has arrived before the loop begins: a wait for it inside the
  // loop would also wait, every time round, for the pixel stream's stores of the bursts before (vmcnt counts both).
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)

  // ONE stage action per iteration, in the order of precedence REPLAY > HEAD > MID > LONG (scalar branches: the
  // scheduler's state is made provably wave-uniform at the top of every iteration).
  for (;;) {
    // the scheduler's state is wave-uniform by construction; readfirstlane makes that provable (scalar registers,
    // scalar branches)
    q0_head = (int) __builtin_amdgcn_readfirstlane((uint32_t) q0_head);
    q0_count = (int) __builtin_amdgcn_readfirstlane((uint32_t) q0_count);
    q1_head = (int) __builtin_amdgcn_readfirstlane((uint32_t) q1_head);
    q1_count = (int) __builtin_amdgcn_readfirstlane((uint32_t) q1_count);
    q2_head = (int) __builtin_amdgcn_readfirstlane((uint32_t) q2_head);
    q2_count = (int) __builtin_amdgcn_readfirstlane((uint32_t) q2_count);
    halves_left = __builtin_amdgcn_readfirstlane(halves_left);
    pending = __builtin_amdgcn_readfirstlane(pending);
    pact = uniform_u64(pact);
    const uint32_t input_left = halves_left | pending;
    if ((input_left | keep_rest) == 0u) break;  // input done and the rest is left to the next launch
    const int n_replaying = __popcll(pact);
    bool draining = false;
    if (input_left == 0u && q0_count == 0 && q1_count == 0) {
      draining = __ballot(l_rem[0] > 0 || l_rem[1] > 0 || l_rem[2] > 0 || l_rem[3] > 0) == 0ull;
    }
    // (wave-uniform by construction; readfirstlane makes that provable, so that the stages' branches are scalar)
    const auto uniform = [](bool c) { return __builtin_amdgcn_readfirstlane(c ? 1u : 0u) != 0u; };
    const bool do_replay = uniform((q2_count > 0 && q2_count + n_replaying >= 64) || n_replaying >= kReplayMin ||
                                   (draining && (q2_count > 0 || n_replaying > 0)));
    if (uniform(draining) && !do_replay) break;  // nothing is left at all
    const bool do_head = !do_replay && uniform(input_left != 0u && q0_count < 64);
    const bool do_mid = !do_replay && !do_head &&
                        uniform(q0_count > 0 && q1_count < kQ1Low && (q0_count >= 64 || input_left == 0u));
    const bool do_long = !do_replay && !do_head && !do_mid;

    // ---------------------------------------------------------------- REPLAY
    const unsigned long long t0 = kTimed ? __builtin_amdgcn_s_memtime() : 0ull;
    if (do_replay) {
      // The visited pixels go to this wave's stream region (compacted, coalesced stores); a full region makes the
      // burst add to the histogram directly, so the result never depends on the workspace size.
      // (kChunked: a burst opens at most one chunk per group and one per kChunkWords words)
      const KernelArgs ra = kernel_args();
      const uint32_t direct =
          kChunked ? ((next_chunk + ra->bin.n_groups + 64u * kReplayBurst / kChunkWords + 1u <= 2u * ra->bin.chunks_per_wave) ? 0u : 1u)
                   : ((region_fill + 64u * kReplayBurst <= region_cap) ? 0u : 1u);
      ChunkOverflow ovf;
      uint32_t steps = 0, hits = 0, popped;
#ifdef CB_WIDE_PROBE
      const uint32_t clock_before = replay_clock;
#endif
      popped = replay_stage<kPow2, kChunked>(do_replay ? 1u : 0u, pact, (uint32_t) q2_head, (uint32_t) q2_count, q2_lds,
                                             draining ? 1u : (uint32_t) kReplayMin, kReplayBurst, direct, po, p_start,
                                             region, region_fill, replay_clock, steps, hits, cursors_lds, ovf);
      if (kChunked && ovf.over != 0ull) {
        wide_open_chunks(ovf, region, my_cursors, next_chunk,
                         ra->bin.chunk_desc + (size_t) (2u * wave_id) * ra->bin.chunks_per_wave);
        next_chunk = __builtin_amdgcn_readfirstlane(next_chunk);
      }
      q2_head = q2_wrap(q2_head + (int) popped);
      q2_count -= (int) popped;
      n_recorded += popped;
      n_replay += steps;
      n_incr += hits;
#ifdef CB_WIDE_PROBE
      if (replay_clock != clock_before) {
        probe_replay_steps += replay_clock - clock_before;
        ++probe_bursts;
        probe_lanes_at_start += (uint32_t) n_replaying + popped;
        probe_lanes_at_end += (uint32_t) __popcll(pact);
      }
#endif
      if (do_replay && __ballot(lane_in(pact) && (replay_clock - p_start) > (uint32_t) max_iter) != 0ull) {
        // cannot happen: the orbit escaped within max_iter steps in an earlier stage
        status |= CB_STATUS_REPLAY_RUNAWAY;
        pact &= ~__ballot((replay_clock - p_start) > (uint32_t) max_iter);
      }
      if (kTimed) t_replay += __builtin_amdgcn_s_memtime() - t0;
    }

    // ---------------------------------------------------------------- HEAD
    if (do_head) {
      uint32_t f_rejected = 0, f_too_fast = 0, f_steps = 0;
      const bool bodies = do_head && __builtin_amdgcn_readfirstlane(halves_left) != 0u;
      static_assert(kPrioHalves == 256, "CBW_AFTER ends the statement where halves % 256 == 0");
      if (bodies && (halves_left & (kPrioHalves - 1u)) == 0u) post_progress_and_set_priority(halves_left);
      uint32_t count = (uint32_t) q0_count;
      if (bodies) head_bodies(1u, rng, halves_left, hs, (uint32_t) (q0_head + q0_count), count,
                  q0_lds, f_rejected, f_too_fast, f_steps);  // survivors -> Q0
      q0_count = (int) count;
      if (do_head && !bodies) {  // the launch's last sample: its test alone (cudabrot.cu:398, 326-337)
        // (its outputs, out of its generator's words: hs = 2 p + g -- A has drawn p + g samples, B p; the last draw was
        // A's if g == 1)
        const uint32_t from_b = (hs & 1u) ^ 1u, drawn = (hs >> 1) + (from_b ? 0u : 1u);
        const Xorwow lg = xorwow_unrotate(from_b ? rng.b : rng.a, (5u - drawn % 5u) % 5u);
        const double p_cr = coordinate2_of(lg.d + lg.x1 - 3u * 362437u, lg.d + lg.x2 - 2u * 362437u);
        const double p_ci = coordinate2_of(lg.d + lg.x3 - 362437u, lg.d + lg.x4);
        Orbit o = {p_cr, p_ci, p_cr, p_ci};
#ifdef CB_BURNING_SHIP
        const bool alive = true;  // cudabrot.cu:397-399: no shortcut in this variant
#else
        const bool alive = !(in_main_cardioid2(o.cr, o.ci) || in_order2_bulb2(o.cr, o.ci));  // cudabrot.cu:398
#endif
        unsigned long long alive_mask = __ballot(alive);
        f_rejected = 64u - (uint32_t) __popcll(alive_mask);
        if (alive_mask != 0ull) {
          const unsigned long long esc = iterate_steps(alive_mask, (uint32_t) kHeadSteps, o, f_steps);
          f_too_fast = (uint32_t) __popcll(esc);  // before min_iter (min_iter >= the start of the LONG stage)
          alive_mask &= ~esc;
        }
        if (alive_mask != 0ull) {
          if (lane_in(alive_mask)) {
            const int slot = (q0_head + q0_count + mask_prefix(alive_mask)) & (kQ0Cap - 1);
            q.q0_cr[slot] = o.cr;
            q.q0_ci[slot] = o.ci;
          }
          q0_count += __popcll(alive_mask);
        }
        pending = 0u;
      }
      if (q0_count > kQ0Cap) status |= CB_STATUS_QUEUE_OVERFLOW;
      n_rejected += f_rejected;
      n_too_fast += f_too_fast;
      n_iterate += f_steps;
      if (kTimed) t_head += __builtin_amdgcn_s_memtime() - t0;
    }

    // ---------------------------------------------------------------- MID (touches no lane register that lives on)
    if (do_mid) {
      const int n = (int) __builtin_amdgcn_readfirstlane((uint32_t) (q0_count < 64 ? q0_count : 64));
      const KernelArgs ma = kernel_args();
      const unsigned long long take = uniform_u64((n == 64) ? ~0ull : ((1ull << n) - 1ull));
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
      n_iterate += steps;
      n_too_fast += (unsigned long long) __popcll(take & ~alive);  // escaped before min_iter
      if (hit != 0ull) {  // samples of cells proven never-escaping: the reference iterates them to max_iter (cudabrot.cu:339)
        const uint32_t n_hit = (uint32_t) __popcll(hit & alive);
        n_never += n_hit;
        skipped_s += (unsigned long long) n_hit * (unsigned long long) (uint32_t) (max_iter - long_start);
        if ((hit & ~alive) != 0ull) status |= CB_STATUS_INTERIOR_MAP;
        alive &= ~hit;
      }
      q1_count += __popcll(alive);
      if (q1_count > kQ1Cap) status |= CB_STATUS_QUEUE_OVERFLOW;
      if (kTimed) t_mid += __builtin_amdgcn_s_memtime() - t0;
    }

    // ---------------------------------------------------------------- LONG
    if (do_long) {
      const KernelArgs la = kernel_args();
      const uint32_t long_steps_u = la->long_steps, tail_value = la->tail_value;
      if (do_long) {
        if ((long_chunks & (kPrioChunks - 1u)) == 0u) post_progress_and_set_priority(halves_left);
        ++long_chunks;
      }
      unsigned long long full_mask[kSlots], tail_mask[kSlots];
      {  // refill idle orbit slots from Q1
        uint32_t head = (uint32_t) q1_head, count = (uint32_t) q1_count;
        long_refill4(lo, seen_r, seen_i, l_rem, head, count, q1_lds, long_steps_u, tail_value, full_mask, tail_mask);
        q1_head = (int) head;
        q1_count = (int) count;
      }
      const unsigned long long any_full = full_mask[0] | full_mask[1] | full_mask[2] | full_mask[3];
#pragma unroll
      for (int o = 0; o < kSlots; ++o) {  // the last, shorter chunk of an orbit (rare; the statement skips an empty mask)
        uint32_t steps = 0;
        const uint32_t q2_tail = __builtin_amdgcn_readfirstlane((uint32_t) q2_wrap(q2_head + q2_count));
        const unsigned long long esc_t = long_tail(tail_mask[o], (uint32_t) tail_steps, lo[o], l_rem[o], q2_tail, q2_lds, steps);
        if (tail_mask[o] != 0ull) {
          const uint32_t n_esc = (uint32_t) __popcll(esc_t), n_end = (uint32_t) __popcll(tail_mask[o] & ~esc_t);
          n_iterate += steps;
          n_never += n_end;  // reached max_iter (cudabrot.cu:339)
          // what was counted for the pushed orbits: everything before this chunk and their own steps in it (the lanes
          // that ended made tail_steps each)
          counted_s += (unsigned long long) n_esc * (unsigned long long) (uint32_t) (max_iter - tail_steps) +
                       (unsigned long long) (steps - n_end * (uint32_t) tail_steps);
          q2_count += (int) n_esc;
          if (q2_count > kQ2Cap) status |= CB_STATUS_QUEUE_OVERFLOW;
        }
      }
      unsigned long long esc[kSlots], doubt;
      iterate_chunk4(any_full != 0ull ? 1u : 0u, full_mask, lo, esc, doubt, la->sparse_threshold);
      n_iterate += (unsigned long long) kChunk * (unsigned long long) (__popcll(full_mask[0]) + __popcll(full_mask[1]) +
                                                                       __popcll(full_mask[2]) + __popcll(full_mask[3]));
#ifdef CB_WIDE_PROBE
      probe_chunks += any_full != 0ull ? 1u : 0u;
      probe_lane_chunks += (unsigned long long) (__popcll(full_mask[0]) + __popcll(full_mask[1]) + __popcll(full_mask[2]) +
                                                 __popcll(full_mask[3]));
#endif
      if (doubt != 0ull) {  // a sample with |c| next to 2 somewhere in the wave: decided exactly, slot by slot
        const double kt = la->sparse_threshold;
#pragma unroll
        for (int o = 0; o < kSlots; ++o) {
          const bool unsure = !(__builtin_fma(lo[o].ci, lo[o].ci, lo[o].cr * lo[o].cr) < kt);
          const unsigned long long d = uniform_u64(__ballot(unsure)) & full_mask[o];
          if (d != 0ull) {
            const unsigned long long really = uniform_u64(verify_chunk_escape(d, lo[o], max_iter - l_rem[o]));
            esc[o] = (esc[o] & ~d) | (really & d);
          }
          asm volatile("" ::: "memory");  // one slot after the other (a rare path: keep its registers few)
        }
      }
      const uint32_t check_flag = (uint32_t) la->check_periodic;
      const int accept_rem = la->accept_rem;
      const uint32_t counted_base = (uint32_t) la->max_iter + (uint32_t) kChunk;
      {
        unsigned long long ran_u[kSlots], esc_u[kSlots], bad;
#pragma unroll
        for (int o = 0; o < kSlots; ++o) {
          ran_u[o] = uniform_u64(full_mask[o]);
          esc_u[o] = uniform_u64(esc[o]);
        }
        uint32_t pushed, never;
        long_retire4(lo, seen_r, seen_i, l_rem, skip_lo, skip_hi, counted, ran_u, esc_u, accept_rem, long_steps_u,
                     counted_base, check_flag, (uint32_t) q2_wrap(q2_head + q2_count), q2_lds, pushed, never, bad);
        q2_count += (int) pushed;
        if (q2_count > kQ2Cap || bad != 0ull) status |= CB_STATUS_QUEUE_OVERFLOW;  // (bad: cannot happen, every LONG escape is accepted)
        n_never += never;
      }
      if (kTimed) t_long += __builtin_amdgcn_s_memtime() - t0;
    }
  }

  if (lane_id() == 0) {  // this wave no longer competes
    __hip_atomic_store(board_row + (wave_slot & 15u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // back to the logical order of the generator words: hs = 2 p + g -- A has drawn p + g samples (mod 5), B p
  {
    const uint32_t p = hs >> 1, g = hs & 1u;
    rng.a = xorwow_unrotate(rng.a, (5u - (p + g) % 5u) % 5u);
    rng.b = xorwow_unrotate(rng.b, (5u - p % 5u) % 5u);
  }
  // (the arguments read afresh and the pointers made opaque: addresses formed here, not kept in registers since the
  // kernel began)
  const KernelArgs ea = fresh_args();
  {
    uint32_t *const st = ea->states;
    const uint32_t nt = ea->n_threads;
    const uint32_t ta = __builtin_amdgcn_readfirstlane(wave_id) * 128u + (uint32_t) lane_id();
    store_rng(st, nt, ta, rng.a);
    store_rng(st, nt, ta + 64u, rng.b);
  }
  if (kChunked) {
    // the last chunk of every group holds what its cursor says; every other chunk in use is full.  The wave's chunks
    // as those of the two segments the scatter knows (their rows of chunk_desc lie side by side).
    const uint32_t cpw = ea->bin.chunks_per_wave;
    uint32_t *desc = ea->bin.chunk_desc + (size_t) (2u * wave_id) * cpw;
    asm volatile("" : "+s"(desc));
    const uint32_t g = (uint32_t) lane_id();
    if (g < ea->bin.n_groups) {
      const uint32_t pos = my_cursors[2u * g], lim = my_cursors[2u * g + 1u];
      if (lim != 0u) desc[lim / kChunkWords - 1u] = (g << 16) | (pos - (lim - kChunkWords));
    }
    if (lane_id() == 0) {
      uint32_t *const wc = ea->bin.wave_count;
      wc[2u * wave_id] = next_chunk < cpw ? next_chunk : cpw;
      wc[2u * wave_id + 1u] = next_chunk < cpw ? 0u : next_chunk - cpw;
    }
  } else if (lane_id() == 0) {  // the wave's segment as the two segments the scatter knows
    uint32_t *const wc = ea->bin.wave_count;
    const uint32_t cap = ea->bin.cap;
    wc[2u * wave_id] = region_fill < cap ? region_fill : cap;
    wc[2u * wave_id + 1u] = region_fill < cap ? 0u : region_fill - cap;
  }
  {  // leave queues and orbit slots for the next launch (empty after a drain)
    unsigned long long *carry = ea->carry + (size_t) wave_id * (2u * kCarryWordsPerWave);
    asm volatile("" : "+s"(carry));
    if (lane_id() == 0) {
      carry[0] = 2ull;
      carry[kCarryWordsPerWave] = 2ull;  // (where draw_wave_kernel's odd waves look for their tag)
      carry[1] = (unsigned long long) (uint32_t) q0_head | ((unsigned long long) (uint32_t) q0_count << 32);
      carry[2] = (unsigned long long) (uint32_t) q1_head | ((unsigned long long) (uint32_t) q1_count << 32);
      carry[3] = (unsigned long long) (uint32_t) q2_head | ((unsigned long long) (uint32_t) q2_count << 32);
      carry[4] = (unsigned long long) replay_clock;
    }
    const unsigned long long *lds_words = reinterpret_cast<const unsigned long long *>(&q);
    unsigned long long *img = carry + kCarryHeaderWords;
    for (uint32_t k = lane_id(); k < kWideQueueWords; k += 64u) img[k] = lds_words[k];
    unsigned long long *pl = carry + lane_id();
#pragma unroll
    for (int o = 0; o < kSlots; ++o) {
      pl[wide_plane(o * 6 + 0)] = (unsigned long long) __double_as_longlong(lo[o].cr);
      pl[wide_plane(o * 6 + 1)] = (unsigned long long) __double_as_longlong(lo[o].ci);
      pl[wide_plane(o * 6 + 2)] = (unsigned long long) __double_as_longlong(lo[o].r);
      pl[wide_plane(o * 6 + 3)] = (unsigned long long) __double_as_longlong(lo[o].i);
      pl[wide_plane(o * 6 + 4)] = (unsigned long long) __double_as_longlong(seen_r[o]);
      pl[wide_plane(o * 6 + 5)] = (unsigned long long) __double_as_longlong(seen_i[o]);
    }
    pl[wide_plane(24)] = (unsigned long long) (uint32_t) l_rem[0] | ((unsigned long long) (uint32_t) l_rem[1] << 32);
    pl[wide_plane(25)] = (unsigned long long) (uint32_t) l_rem[2] | ((unsigned long long) (uint32_t) l_rem[3] << 32);
    pl[wide_plane(26)] = (unsigned long long) __double_as_longlong(po.cr);
    pl[wide_plane(27)] = (unsigned long long) __double_as_longlong(po.ci);
    pl[wide_plane(28)] = (unsigned long long) __double_as_longlong(po.r);
    pl[wide_plane(29)] = (unsigned long long) __double_as_longlong(po.i);
    pl[wide_plane(30)] = (unsigned long long) p_start | ((unsigned long long) (lane_in(pact) ? 1u : 0u) << 32);
  }
  const unsigned long long skipped_total = wave_sum(((unsigned long long) skip_hi << 32) | skip_lo) + skipped_s;
  const unsigned long long counted_total = wave_sum((unsigned long long) counted) + counted_s;
  if (ea->counters && lane_id() == 0) {
    unsigned long long *c = reinterpret_cast<unsigned long long *>(ea->counters);
    const unsigned long long n_samples = 128ull * (unsigned long long) ea->samples_per_thread;
    // iterate_steps: what the stages counted, minus (counted for the pushed orbits - their replay steps): see
    // long_retire.  Modulo 2^64 per launch (an orbit may be pushed in one launch and replayed in the next); exact
    // once the carried work is complete, which is when anything reads the counters.
    const unsigned long long v[9] = {n_samples, n_rejected, n_never,  n_too_fast, n_recorded,
                                     n_iterate + skipped_total - counted_total + n_replay, n_replay, n_incr,
                                     skipped_total};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (v[k]) __hip_atomic_fetch_add(c + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (status) __hip_atomic_fetch_or(c + 9, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (kTimed) {  // cycles_head = HEAD + MID, as draw_wave_kernel reports them
      const unsigned long long t_all = __builtin_amdgcn_s_memtime() - t_start;
      __hip_atomic_fetch_add(c + 10, t_head + t_mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(c + 11, t_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(c + 12, t_replay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(c + 13, t_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef CB_WIDE_PROBE
      (void) rt_start;
      __hip_atomic_fetch_add(c + 14, probe_chunks | (probe_lane_chunks << 28), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(c + 15, probe_replay_steps | (probe_bursts << 36), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
      const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
      __hip_atomic_fetch_max(c + 14, ~rt_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_max(c + 15, rt_end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      // (the MID stage's share of cycles_head, in the slot draw_wave_kernel uses for the waves' lifetimes)
#ifdef CB_WIDE_PROBE
      __hip_atomic_fetch_add(c + 16, probe_lanes_at_start | (probe_lanes_at_end << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
      __hip_atomic_fetch_add(c + 16, t_mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
  }
}

}  // namespace

#ifndef CB_BURNING_SHIP
// Can this kernel take the launch?  (Everything else is draw_wave_kernel's.)
bool draw_wide_takes(const DrawArgs &a) {
  // a one-level stream, or a chunked one on a canvas whose pixels' byte offsets fit 32 bits (the direct form of the burst)
  const bool binned_one_level =
      a.bin.enabled != 0u &&
      ((a.bin.two_level == 0u && a.bin.chunked == 0u) ||
       (a.bin.chunked != 0u && a.bin.n_planes == 1u && a.bin.e_row_shift == 16u &&
        (unsigned long long) a.w * (unsigned long long) a.h < (1ull << 29)));
  const bool usual_split = a.head_steps == kHeadSteps && a.fast_mid != 0 && a.sparse_long != 0 &&
                           a.min_iter == a.long_start && a.max_iter > a.long_start;
  return binned_one_level && usual_split && a.n_channels == 0 && a.carry != nullptr && a.n_threads != 0u &&
         a.n_threads % (128u * kWavesPerBlock) == 0u && a.bin.n_waves == a.n_threads / 64u &&
         a.bin.cap >= 64u * kReplayBurst;
}
#endif

hipError_t CB_LAUNCH_NAME(const DrawArgs &a, bool timed, hipStream_t stream) {
  const bool drain_launch = a.drain != 0;
  if (a.samples_per_thread == 0 && !drain_launch) return hipSuccess;
  const uint32_t threads = 64 * kWavesPerBlock;
  const uint32_t blocks = a.n_threads / (128u * kWavesPerBlock);
  const bool pow2 = a.pow2_real && a.pow2_imag;
  const bool chunked = a.bin.chunked != 0u;
#define CBW_LAUNCH(P, T, C)                                                                                        \
  hipLaunchKernelGGL((draw_wide_kernel<P, T, C>), dim3(blocks), dim3(threads),                                     \
                     (C) ? kWavesPerBlock * 2u * kChunkedGroupsMax * sizeof(uint32_t) : 0u, stream, a)
  if (chunked) {
    if (timed) {
      if (pow2) CBW_LAUNCH(true, true, true); else CBW_LAUNCH(false, true, true);
    } else {
      if (pow2) CBW_LAUNCH(true, false, true); else CBW_LAUNCH(false, false, true);
    }
  } else if (timed) {
    if (pow2) CBW_LAUNCH(true, true, false); else CBW_LAUNCH(false, true, false);
  } else {
    if (pow2) CBW_LAUNCH(true, false, false); else CBW_LAUNCH(false, false, false);
  }
#undef CBW_LAUNCH
  return hipGetLastError();
}

}  // namespace cb
