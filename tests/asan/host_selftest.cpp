// host_selftest.cpp -- TEST INFRASTRUCTURE: drives the product's host code (state_files.cpp, host_output.cpp,
// xorwow_host.cpp) under ASan + UBSan, with the oracle (buddha_oracle.c) as the checker where a result can be
// compared.  Prints one "ok <name>" line per check and exits nonzero at the first failure; the sanitizers
// abort on their own findings.  Usage: host_selftest <scratch directory>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "../../include/cudabrot_amd.h"
#include "../../oracle/buddha_oracle.h"
#include "kernels.h"
#include "state_files.h"

namespace {

int g_checks = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) {                                                       \
      printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);           \
      exit(1);                                                           \
    }                                                                    \
    ++g_checks;                                                          \
  } while (0)

uint64_t g_lcg = 0x9e3779b97f4a7c15ull;
uint64_t rnd() {
  g_lcg = g_lcg * 6364136223846793005ull + 1442695040888963407ull;
  return g_lcg >> 11;
}

std::vector<unsigned char> slurp(const std::string &path) {
  std::vector<unsigned char> out;
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return out;
  unsigned char buf[4096];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + n);
  fclose(f);
  return out;
}

void spit(const std::string &path, const void *data, size_t n) {
  FILE *f = fopen(path.c_str(), "wb");
  EXPECT(f != nullptr);
  if (n) EXPECT(fwrite(data, n, 1, f) == 1);
  fclose(f);
}

void test_state_files(const std::string &dir) {
  using cb::FileResult;
  const uint32_t w = 37, h = 11;
  std::vector<cb_pixel> a((size_t) w * h), b((size_t) w * h, 7);
  for (auto &v : a) v = rnd() << (rnd() % 30);  // counts beyond 2^32 as well
  const std::string p = dir + "/state.bin";
  unlink(p.c_str());
  EXPECT(cb::load_state_file(p.c_str(), w, h, 1, b.data()) == FileResult::kAbsent);  // cudabrot.cu:227-231
  EXPECT(b[0] == 7);
  EXPECT(cb::save_state_file(p.c_str(), w, h, 1, a.data()) == FileResult::kOk);
  EXPECT(slurp(p).size() == sizeof(cb::StateHeader) + a.size() * 8);
  EXPECT(cb::load_state_file(p.c_str(), w, h, 1, b.data()) == FileResult::kOk);
  EXPECT(a == b);
  // another canvas, other plane count: refused, destination untouched
  std::fill(b.begin(), b.end(), 7);
  EXPECT(cb::load_state_file(p.c_str(), w, h + 1, 1, b.data()) == FileResult::kError);
  EXPECT(cb::load_state_file(p.c_str(), h, w, 1, b.data()) == FileResult::kError);
  EXPECT(b[0] == 7 && b.back() == 7);
  // the advisor's case: a native buffer of HALF the pixel count has w*h*4 bytes of payload for the full canvas
  {
    std::vector<cb_pixel> big((size_t) w * 2 * h, 9);
    EXPECT(cb::load_state_file(p.c_str(), w * 2, h, 1, big.data()) == FileResult::kError);
    EXPECT(big[0] == 9);
  }
  // truncated native file (header intact): refused by the size check; a header alone too
  std::vector<unsigned char> bytes = slurp(p);
  spit(p, bytes.data(), bytes.size() - 5);
  EXPECT(cb::load_state_file(p.c_str(), w, h, 1, b.data()) == FileResult::kError);
  spit(p, bytes.data(), sizeof(cb::StateHeader));
  EXPECT(cb::load_state_file(p.c_str(), w, h, 1, b.data()) == FileResult::kError);
  spit(p, bytes.data(), 3);  // shorter than a header, not a reference size either
  EXPECT(cb::load_state_file(p.c_str(), w, h, 1, b.data()) == FileResult::kError);
  spit(p, bytes.data(), 0);
  EXPECT(cb::load_state_file(p.c_str(), w, h, 1, b.data()) == FileResult::kError);
  // a header that lies about its counter width
  {
    std::vector<unsigned char> lie = bytes;
    lie[20] = 4;  // counter_bytes
    spit(p, lie.data(), lie.size());
    EXPECT(cb::load_state_file(p.c_str(), w, h, 1, b.data()) == FileResult::kError);
  }
  // the reference's format: bare uint32[h][w] (cudabrot.cu:262-280), widened; only for one plane
  std::vector<uint32_t> narrow((size_t) w * h);
  for (auto &v : narrow) v = (uint32_t) rnd();
  spit(p, narrow.data(), narrow.size() * 4);
  EXPECT(cb::load_state_file(p.c_str(), w, h, 1, b.data()) == FileResult::kOk);
  for (size_t i = 0; i < narrow.size(); ++i) EXPECT(b[i] == narrow[i]);
  {
    std::vector<cb_pixel> two((size_t) w * h * 2, 5);
    EXPECT(cb::load_state_file(p.c_str(), w, h, 2, two.data()) == FileResult::kError);
  }
  // several planes round trip
  {
    std::vector<cb_pixel> planes((size_t) w * h * 3), back((size_t) w * h * 3);
    for (auto &v : planes) v = rnd();
    EXPECT(cb::save_state_file(p.c_str(), w, h, 3, planes.data()) == FileResult::kOk);
    EXPECT(cb::load_state_file(p.c_str(), w, h, 3, back.data()) == FileResult::kOk);
    EXPECT(planes == back);
    EXPECT(cb::load_state_file(p.c_str(), w, h, 1, back.data()) == FileResult::kError);
  }
  // --state-format raw: the reference's bare buffer (cudabrot.cu:262-280).  uint32 when every count fits ...
  {
    using cb::StateFormat;
    std::vector<cb_pixel> small((size_t) w * h), back((size_t) w * h, 3);
    for (auto &v : small) v = (uint32_t) rnd();
    small[5] = 0xffffffffull;  // the largest count that still fits
    EXPECT(cb::save_state_file(p.c_str(), w, h, 1, small.data(), StateFormat::kRaw) == FileResult::kOk);
    std::vector<unsigned char> raw = slurp(p);
    EXPECT(raw.size() == small.size() * 4);
    for (size_t i = 0; i < small.size(); ++i) {
      uint32_t v;
      memcpy(&v, raw.data() + 4 * i, 4);
      EXPECT(v == (uint32_t) small[i]);
    }
    EXPECT(cb::load_state_file(p.c_str(), w, h, 1, back.data(), StateFormat::kRaw) == FileResult::kOk);
    EXPECT(small == back);
    EXPECT(cb::load_state_file(p.c_str(), w, h, 1, back.data()) == FileResult::kOk);  // also without the flag (announced)
    // ... uint64 once a count does not fit, or with several planes: only --state-format raw reads that back
    small[7] = 0x100000000ull;
    EXPECT(cb::save_state_file(p.c_str(), w, h, 1, small.data(), StateFormat::kRaw) == FileResult::kOk);
    EXPECT(slurp(p).size() == small.size() * 8);
    std::fill(back.begin(), back.end(), 3);
    EXPECT(cb::load_state_file(p.c_str(), w, h, 1, back.data()) == FileResult::kError);  // headerless 64-bit: never guessed
    EXPECT(back[0] == 3);
    EXPECT(cb::load_state_file(p.c_str(), w, h, 1, back.data(), StateFormat::kRaw) == FileResult::kOk);
    EXPECT(small == back);
    EXPECT(cb::load_state_file(p.c_str(), w + 1, h, 1, back.data(), StateFormat::kRaw) == FileResult::kError);
    std::vector<cb_pixel> planes((size_t) w * h * 2), pback((size_t) w * h * 2);
    for (auto &v : planes) v = (uint32_t) rnd();
    EXPECT(cb::save_state_file(p.c_str(), w, h, 2, planes.data(), StateFormat::kRaw) == FileResult::kOk);
    EXPECT(slurp(p).size() == planes.size() * 8);
    EXPECT(cb::load_state_file(p.c_str(), w, h, 2, pback.data(), StateFormat::kRaw) == FileResult::kOk);
    EXPECT(planes == pback);
    // a native file is still recognised by its magic under the flag
    EXPECT(cb::save_state_file(p.c_str(), w, h, 1, a.data()) == FileResult::kOk);
    EXPECT(cb::load_state_file(p.c_str(), w, h, 1, back.data(), StateFormat::kRaw) == FileResult::kOk);
    EXPECT(a == back);
  }
  // unwritable / unreadable paths
  EXPECT(cb::save_state_file((dir + "/no/such/dir/x").c_str(), w, h, 1, a.data()) == FileResult::kError);
  EXPECT(cb::load_state_file(dir.c_str(), w, h, 1, b.data()) == FileResult::kError);  // a directory
  printf("ok state_files\n");
}

void test_rng_sidecar(const std::string &dir) {
  using cb::FileResult;
  const std::string p = dir + "/state.rng";
  unlink(p.c_str());
  const uint32_t threads = 64;
  const size_t blob = cb_rng_state_bytes(threads);
  EXPECT(blob == 64 * 24);
  std::vector<std::vector<unsigned char>> in(3, std::vector<unsigned char>(blob)), out;
  for (auto &v : in) {
    for (auto &c : v) c = (unsigned char) rnd();
  }
  uint64_t passes = 0;
  EXPECT(cb::load_rng_sidecar(p.c_str(), 1337, threads, 3, blob, &out, &passes) == FileResult::kAbsent);
  EXPECT(cb::save_rng_sidecar(p.c_str(), 1337, threads, 12, in) == FileResult::kOk);
  EXPECT(cb::load_rng_sidecar(p.c_str(), 1337, threads, 3, blob, &out, &passes) == FileResult::kOk);
  EXPECT(passes == 12 && out == in);
  EXPECT(cb::load_rng_sidecar(p.c_str(), 99, threads, 3, blob, &out, &passes) == FileResult::kError);        // other seed
  EXPECT(cb::load_rng_sidecar(p.c_str(), 1337, threads, 2, blob, &out, &passes) == FileResult::kError);      // other rank count
  EXPECT(cb::load_rng_sidecar(p.c_str(), 1337, threads * 2, 3, cb_rng_state_bytes(threads * 2), &out, &passes) ==
         FileResult::kError);
  std::vector<unsigned char> bytes = slurp(p);
  spit(p, bytes.data(), bytes.size() - 1);  // short read inside the last generator
  EXPECT(cb::load_rng_sidecar(p.c_str(), 1337, threads, 3, blob, &out, &passes) == FileResult::kError);
  bytes.push_back(0);  // trailing garbage
  spit(p, bytes.data(), bytes.size());
  EXPECT(cb::load_rng_sidecar(p.c_str(), 1337, threads, 3, blob, &out, &passes) == FileResult::kError);
  bytes.pop_back();
  bytes[sizeof(cb::RngStateHeader)] ^= 1;  // rank 0 claims another first subsequence
  spit(p, bytes.data(), bytes.size());
  EXPECT(cb::load_rng_sidecar(p.c_str(), 1337, threads, 3, blob, &out, &passes) == FileResult::kError);
  spit(p, "CBRNGST1", 8);  // a round-1 sidecar / a truncated header
  EXPECT(cb::load_rng_sidecar(p.c_str(), 1337, threads, 1, blob, &out, &passes) == FileResult::kError);
  printf("ok rng_sidecar\n");
}

void test_output_stage(const std::string &dir) {
  const int w = 53, h = 17;
  std::vector<cb_pixel> hist((size_t) w * h);
  for (const double gamma : {1.0, 2.2, 0.5, 0.0, -1.0}) {
    for (auto &v : hist) v = (rnd() % 5 == 0) ? 0 : rnd() % 100000;
    hist[5] = (1ull << 33) + 17;  // a count beyond 32 bits
    std::vector<uint16_t> mine((size_t) w * h), theirs((size_t) w * h);
    uint64_t mx = 0;
    double scale = 0, oscale = 0;
    cb_set_grayscale_pixels(hist.data(), w, h, gamma, mine.data(), &mx, &scale);
    const uint64_t omx = orc_set_grayscale_pixels(hist.data(), w, h, gamma, theirs.data(), &oscale);
    EXPECT(mx == omx && scale == oscale && mine == theirs);
    for (int i = 0; i < w * h; i += 7) EXPECT(cb_tone_value(hist[(size_t) i], mx, gamma) == mine[(size_t) i]);
    // PGM bytes: the oracle's encoder vs both writers
    std::vector<uint8_t> want(64 + 2 * (size_t) w * h);
    want.resize(orc_encode_pgm(theirs.data(), w, h, want.data()));
    const std::string p = dir + "/img.pgm";
    std::vector<uint16_t> swapped = mine;
    EXPECT(cb_save_image(p.c_str(), swapped.data(), w, h) == 0);  // swaps in place
    EXPECT(slurp(p) == want);
    EXPECT(cb_save_image_be(p.c_str(), swapped.data(), w, h) == 0);
    EXPECT(slurp(p) == want);
  }
  // an empty histogram: max = 0, scale = inf, every pixel 0 (pinned)
  std::fill(hist.begin(), hist.end(), 0);
  std::vector<uint16_t> gray((size_t) w * h, 1);
  uint64_t mx = 1;
  double scale = 0;
  cb_set_grayscale_pixels(hist.data(), w, h, 1.0, gray.data(), &mx, &scale);
  EXPECT(mx == 0);
  for (uint16_t v : gray) EXPECT(v == 0);
  uint16_t one = 0;
  EXPECT(cb_save_image((dir + "/no/such/dir/x.pgm").c_str(), &one, 1, 1) == 1);
  printf("ok output_stage\n");
}

void test_xorwow_tables() {
  std::vector<uint32_t> mine((size_t) cb::kSeqJumpMatrices * cb::kMatrixWords);
  cb::build_sequence_jump_matrices(mine.data());
  // the oracle has rocRAND's layout: matrix i = A^(2^(67 + 2 i)); the product keeps one per binary digit
  for (int i = 0; i < 32; ++i) {
    uint32_t theirs[800];
    orc_xorwow_sequence_jump_matrix(i, theirs);
    EXPECT(memcmp(theirs, mine.data() + (size_t) (2 * i) * cb::kMatrixWords, sizeof(theirs)) == 0);
  }
  for (const uint64_t seed : {1337ull, 0ull, 99ull, 123456789012345ull, ~0ull}) {
    uint32_t x[5], d;
    cb::seed_state(seed, x, &d);
    orc_xorwow st;
    orc_xorwow_init(seed, 0, 0, &st);
    EXPECT(d == st.d && memcmp(x, st.x, sizeof(x)) == 0);
  }
  printf("ok xorwow_tables\n");
}

void test_canvas_validation() {
  cb_fractal_dimensions d = {1000, 1000, -2.0, -2.0, 2.0, 2.0, 0.0, 0.0};
  const char *msg = nullptr;
  EXPECT(cb_recompute_pixel_deltas(&d, &msg) == 1 && d.delta_real == 0.004 && d.delta_imag == 0.004);
  orc_dims o = {1000, 1000, -2.0, -2.0, 2.0, 2.0, 0.0, 0.0};
  EXPECT(orc_recompute_pixel_deltas(&o) == 1 && o.delta_real == d.delta_real && o.delta_imag == d.delta_imag);
  d.w = 0;
  EXPECT(cb_recompute_pixel_deltas(&d, &msg) == 0 && strcmp(msg, "Output width must be positive.") == 0);
  d.w = 5;
  d.h = -1;
  EXPECT(cb_recompute_pixel_deltas(&d, nullptr) == 0);
  d.h = 5;
  d.max_real = -2.0;
  EXPECT(cb_recompute_pixel_deltas(&d, &msg) == 0);
  d.max_real = 2.0;
  d.max_imag = -3.0;
  EXPECT(cb_recompute_pixel_deltas(&d, &msg) == 0);
  printf("ok canvas_validation\n");
}

}  // namespace

int main(int argc, char **argv) {
  if (argc != 2) {
    printf("usage: host_selftest <scratch directory>\n");
    return 2;
  }
  const std::string dir = argv[1];
  test_state_files(dir);
  test_rng_sidecar(dir);
  test_output_stage(dir);
  test_xorwow_tables();
  test_canvas_validation();
  printf("host_selftest: %d checks passed\n", g_checks);
  return 0;
}
