// cli_main.cpp -- `cudabrot`: command-line drop-in for the reference binary.
//
// The process boundary IS the reference's public interface: argv in, stdout text,
// exit code, a 16-bit PGM and the raw -s buffer out.  This file reproduces that contract
// (cudabrot.cu:579-791: flags :662-754, messages, usage -> exit 0, errors on stdout -> exit 1) on top of
// the C ABI in include/cudabrot_amd.h.  Rendering is done by the hand-written gfx950 kernels only: there
// is no CPU fallback, without a usable GPU the program prints the reference's error line and exits 1.
//
// Observable differences, all deliberate:
//  * the -s buffer holds 64-bit counters behind a 32-byte header that names its own shape (magic, w, h, planes,
//    counter width), so a file of another canvas can never be mistaken for this one's; a reference-format
//    file (exactly w*h*4 bytes of uint32, no header) is accepted on load, announced, and widened;
//  * reference passes (512*512 threads x 50 samples) are fused into launches of about 0.2 s, so -t and
//    Ctrl+C act at launch granularity; the printed pass count still counts reference-sized passes;
//  * extension flags, which the reference answers with its usage text: --passes N, --kernel NAME,
//    --stats, --tonemap FORM, --seed N, --rng-state FILE, --burning-ship, --channel MAX:MIN:FILE, --gpus N,
//    --state-format native|raw (raw: the -s file as the reference's bare buffer, uint32 when every count fits).
#include <errno.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cudabrot_amd.h"
#include "state_files.h"

namespace {

volatile sig_atomic_t g_quit_requested = 0;

struct Settings {
  int device = 0;                                   // -d
  const char *output_image = "output.pgm";          // -o   (cudabrot.cu:26,764)
  const char *inprogress_file = nullptr;            // -s
  double seconds_to_run = 10.0;                     // -t   (cudabrot.cu:769)
  double gamma_correction = 1.0;                    // -g   (cudabrot.cu:770)
  cb_iteration_control iterations = {100, 20};      // -m -c (cudabrot.cu:765-766)
  cb_fractal_dimensions canvas = {1000, 1000, -2.0, -2.0, 2.0, 2.0, 0.0, 0.0};  // cudabrot.cu:533-538
  long fixed_passes = -1;                           // --passes (extension; <0: run by the clock)
  int kernel_variant = CB_KERNEL_DEFAULT;           // --kernel (extension)
  bool print_stats = false;                         // --stats  (extension)
  bool burning_ship = false;                        // --burning-ship (extension; cudabrot.cu:15-17)
  int gpus = 1;                                     // --gpus N (extension): devices -d .. -d + N - 1
  // --channel MAX:MIN:FILE (extension, repeatable): fused multi-channel render, one image per window
  int n_channels = 0;
  cb_iteration_control channel_window[CB_MAX_CHANNELS] = {};
  std::string channel_file[CB_MAX_CHANNELS];
  bool bad_channel = false;
  uint64_t seed = CB_DEFAULT_RNG_SEED;              // --seed (extension; cudabrot.cu:37)
  const char *rng_state_file = nullptr;             // --rng-state (extension): true-resume sidecar
  bool raw_state = false;                           // --state-format raw (extension): -s as the reference's bare buffer
  bool bad_state_format = false;
  int tone_mode = CB_TONE_AUTO;                     // --tonemap (extension): device table / thresholds
  bool host_tonemap = false;                        //   ... or the reference's host loop
};

// The usage text is the command's documented interface (cudabrot.cu:579-620) and is printed as is.
const char kUsageBody[] =
    "Options may be one or more of the following:\n"
    "  --help: Prints these instructions.\n"
    "  -d <device number>: Sets which GPU to use. Defaults to GPU 0.\n"
    "  -o <output file name>: If provided, the rendered image will be saved\n"
    "     to a .pgm file with the given name. Otherwise, saves the image\n"
    "     to output.pgm.\n"
    "  -m <max escape iterations>: The maximum number of iterations to use\n"
    "     before giving up on seeing whether a point escapes.\n"
    "  -c <min escape iterations>: If a point escapes before this number of\n"
    "     iterations, it will be ignored.\n"
    "  -g <gamma correction>: A gamma-correction value to use on the\n"
    "     resulting image. If negative, no gamma correction will occur.\n"
    "  -t <seconds to run>: A number of seconds to run the calculation for.\n"
    "     Defaults to 10.0. If negative, the program will run continuously\n"
    "     and will terminate (saving the image) when it receives a SIGINT.\n"
    "  -w <width>: The width of the output image, in pixels. Defaults to\n"
    "     1000.\n"
    "  -h <height>: The height of the output image, in pixels. Defaults to\n"
    "     1000.\n"
    "  -s <save/load file>: If provided, this gives a file name into which\n"
    "     the rendering buffer will be saved, for future continuation.\n"
    "     If the program is loaded and the file exists, the buffer will be\n"
    "     filled with the contents of the file, but the dimensions must\n"
    "     match. Note that this file may be huge for high-resolution images.\n"
    "\n"
    "The following settings control the location of the output image on the\n"
    "complex plane, but samples are always drawn from the entire Mandelbrot-\n"
    "set domain (-2-2i to 2+2i). So these settings can be used to save\n"
    "memory or \"crop\" the output, but won't otherwise speed up rendering:\n"
    "  --min-real <min real>: The minimum value along the real axis to\n"
    "             include in the output image. Defaults to -2.0.\n"
    "  --max-real <max real>: The maximum value along the real axis to\n"
    "             include in the output image. Defaults to 2.0.\n"
    "  --min-imag <min imag>: The minimum value along the imaginary axis to\n"
    "             include in the output image. Defaults to -2.0.\n"
    "  --max-imag <max imag>: The maximum value along the imaginary axis to\n"
    "             include in the output image. Defaults to 2.0.\n";

// Usage always ends the process with status 0, also after a bad argument (cudabrot.cu:619).
[[noreturn]] void usage_and_exit(const char *program) {
  printf("Usage: %s [options]\n\n", program);
  fputs(kUsageBody, stdout);
  exit(0);
}

// ---- argument table ----------------------------------------------------------------------------

enum class Value { kNone, kInt, kLong, kDouble, kText };  // kInt: truncated to int like the reference's flags

struct Flag {
  const char *name;
  Value value;
  const char *missing_value_message;  // nullptr: "Argument %s needs a value."
  bool revalidates_canvas;            // -w -h --min/max-*: canvas re-checked at once (:704-749)
  std::function<void(Settings &, long, double, const char *)> store;
};

const std::vector<Flag> &flag_table() {
  static const std::vector<Flag> table = {
      {"-d", Value::kInt, nullptr, false,
       [](Settings &s, long i, double, const char *) { s.device = (int) i; }},
      {"-o", Value::kText, "Missing output file name.", false,
       [](Settings &s, long, double, const char *t) { s.output_image = t; }},
      {"-s", Value::kText, "Missing in-progress buffer file name.", false,
       [](Settings &s, long, double, const char *t) { s.inprogress_file = t; }},
      {"-m", Value::kInt, nullptr, false,
       [](Settings &s, long i, double, const char *) {
         s.iterations.max_escape_iterations = (int) i;
         if (s.iterations.max_escape_iterations > 60000) {  // cudabrot.cu:692-695
           printf("Warning: Using a high number of iterations may cause the "
                  "program respond slowly to Ctrl+C or time running out.\n");
         }
       }},
      {"-c", Value::kInt, nullptr, false,
       [](Settings &s, long i, double, const char *) {
         s.iterations.min_escape_iterations = (int) i;
       }},
      {"-w", Value::kInt, nullptr, true,
       [](Settings &s, long i, double, const char *) { s.canvas.w = (int) i; }},
      {"-h", Value::kInt, nullptr, true,
       [](Settings &s, long i, double, const char *) { s.canvas.h = (int) i; }},
      {"-g", Value::kDouble, nullptr, false,
       [](Settings &s, long, double d, const char *) { s.gamma_correction = d; }},
      {"-t", Value::kDouble, nullptr, false,
       [](Settings &s, long, double d, const char *) { s.seconds_to_run = d; }},
      {"--min-real", Value::kDouble, nullptr, true,
       [](Settings &s, long, double d, const char *) { s.canvas.min_real = d; }},
      {"--max-real", Value::kDouble, nullptr, true,
       [](Settings &s, long, double d, const char *) { s.canvas.max_real = d; }},
      {"--min-imag", Value::kDouble, nullptr, true,
       [](Settings &s, long, double d, const char *) { s.canvas.min_imag = d; }},
      {"--max-imag", Value::kDouble, nullptr, true,
       [](Settings &s, long, double d, const char *) { s.canvas.max_imag = d; }},
      // extensions
      {"--passes", Value::kInt, nullptr, false,
       [](Settings &s, long i, double, const char *) { s.fixed_passes = i < 0 ? 0 : i; }},
      {"--kernel", Value::kText, nullptr, false,
       [](Settings &s, long, double, const char *t) {
         s.kernel_variant = (strcmp(t, "simple") == 0)  ? CB_KERNEL_SIMPLE
                            : (strcmp(t, "timed") == 0) ? CB_KERNEL_TIMED
                            : (strcmp(t, "full") == 0)  ? CB_KERNEL_FULL_ITERATE
                                                        : CB_KERNEL_DEFAULT;
       }},
      {"--stats", Value::kNone, nullptr, false,
       [](Settings &s, long, double, const char *) { s.print_stats = true; }},
      {"--channel", Value::kText, nullptr, false,
       [](Settings &s, long, double, const char *t) {
         int mx = 0, mn = 0, used = 0;
         if (s.n_channels >= CB_MAX_CHANNELS || sscanf(t, "%d:%d:%n", &mx, &mn, &used) < 2 || used == 0 ||
             t[used] == 0) {
           s.bad_channel = true;
           return;
         }
         s.channel_window[s.n_channels] = {mx, mn};
         s.channel_file[s.n_channels] = t + used;
         s.n_channels++;
       }},
      {"--gpus", Value::kInt, nullptr, false,
       [](Settings &s, long i, double, const char *) { s.gpus = (i < 1) ? 1 : (i > 64 ? 64 : (int) i); }},
      {"--burning-ship", Value::kNone, nullptr, false,
       [](Settings &s, long, double, const char *) { s.burning_ship = true; }},
      {"--seed", Value::kLong, nullptr, false,  // the generator's seed is 64 bits wide (rocrand_init)
       [](Settings &s, long i, double, const char *) { s.seed = (uint64_t) i; }},
      {"--rng-state", Value::kText, nullptr, false,
       [](Settings &s, long, double, const char *t) { s.rng_state_file = t; }},
      {"--state-format", Value::kText, nullptr, false,
       [](Settings &s, long, double, const char *t) {
         s.raw_state = strcmp(t, "raw") == 0;
         s.bad_state_format = !s.raw_state && strcmp(t, "native") != 0;
       }},
      {"--tonemap", Value::kText, nullptr, false,
       [](Settings &s, long, double, const char *t) {
         s.host_tonemap = strcmp(t, "host") == 0;
         s.tone_mode = (strcmp(t, "lut") == 0)          ? CB_TONE_LUT
                       : (strcmp(t, "thresholds") == 0) ? CB_TONE_THRESHOLDS
                                                        : CB_TONE_AUTO;
       }},
  };
  return table;
}

// Canvas validation with the reference's messages (cudabrot.cu:505-527).
bool canvas_ok(Settings &s) {
  const char *why = nullptr;
  if (cb_recompute_pixel_deltas(&s.canvas, &why)) return true;
  printf("%s\n", why);
  return false;
}

Settings parse_arguments(int argc, char **argv) {
  Settings s;
  if (!canvas_ok(s)) {  // cudabrot.cu:539-542
    printf("Internal error setting default canvas boundaries!\n");
    exit(1);
  }
  for (int i = 1; i < argc; i++) {
    const char *arg = argv[i];
    if (strcmp(arg, "--help") == 0) usage_and_exit(argv[0]);
    const Flag *flag = nullptr;
    for (const Flag &f : flag_table()) {
      if (strcmp(arg, f.name) == 0) {
        flag = &f;
        break;
      }
    }
    if (!flag) {
      printf("Invalid argument: %s\n", arg);  // cudabrot.cu:751
      usage_and_exit(argv[0]);
    }
    long as_int = 0;
    double as_double = 0.0;
    const char *text = nullptr;
    if (flag->value != Value::kNone) {
      if (i + 1 >= argc) {
        if (flag->missing_value_message) {
          printf("%s\n", flag->missing_value_message);
        } else {
          printf("Argument %s needs a value.\n", arg);  // cudabrot.cu:629,648
        }
        usage_and_exit(argv[0]);
      }
      text = argv[++i];
      if (flag->value != Value::kText) {
        // whole-string numbers only; an empty string is not a number (cudabrot.cu:632-639,651-656)
        char *end = nullptr;
        if (flag->value == Value::kInt) {
          as_int = (int) strtol(text, &end, 10);  // truncated to int like the reference
        } else if (flag->value == Value::kLong) {
          as_int = (long) strtoull(text, &end, 10);
        } else {
          as_double = strtod(text, &end);
        }
        if (*end != 0 || text[0] == 0) {
          printf("Invalid number given to argument %s: %s\n", arg, text);
          usage_and_exit(argv[0]);
        }
      }
    }
    flag->store(s, as_int, as_double, text);
    if (s.bad_channel) {
      printf("Invalid channel (want MAX:MIN:FILE, at most %d of them): %s\n", CB_MAX_CHANNELS, text);
      usage_and_exit(argv[0]);
    }
    if (s.bad_state_format) {
      printf("Invalid state format (want native or raw): %s\n", text);
      usage_and_exit(argv[0]);
    }
    if (flag->revalidates_canvas && !canvas_ok(s)) usage_and_exit(argv[0]);
  }
  return s;
}

// ---- the run -------------------------------------------------------------------------------------

double wall_seconds() {  // cudabrot.cu:122-129
  struct timespec ts;
  if (clock_gettime(CLOCK_REALTIME, &ts) != 0) {
    printf("Error getting time.\n");
    exit(1);
  }
  return (double) ts.tv_sec + (double) ts.tv_nsec / 1e9;
}

class Run {
 public:
  explicit Run(const Settings &s) : cfg_(s) {}
  ~Run() { release(); }

  int execute() {
    int max_iterations = cfg_.iterations.max_escape_iterations;
    for (int j = 0; j < cfg_.n_channels; ++j) {
      if (j == 0 || cfg_.channel_window[j].max_escape_iterations > max_iterations) {
        max_iterations = cfg_.channel_window[j].max_escape_iterations;
      }
    }
    printf("Creating %dx%d image, %d max iterations.\n", cfg_.canvas.w, cfg_.canvas.h,
           max_iterations);  // cudabrot.cu:779-780
    printf("Calculating image...\n");
    setup();
    load_inprogress();
    load_rng_state();
    render();
    save_inprogress();
    save_rng_state();
    if (cfg_.n_channels > 0) {
      save_channels();
    } else {
      printf("Saving image.\n");
      save_image(cfg_.output_image);
      printf("Done! Output image saved: %s\n", cfg_.output_image);
    }
    release();
    return 0;
  }

 private:
  Settings cfg_;
  cb_renderer *renderer_ = nullptr;      // rank 0: loads, receives the reduced histogram, tone-maps, saves
  // --gpus N: ranks 1..N-1, one per further device, subsequences [r T, (r+1) T) of the
  // same seed; their histograms are summed onto rank 0 once, after the pass loop (cb_renderers_reduce)
  std::vector<cb_renderer *> peers_;
  cb_pixel *counts_ = nullptr;   // host mirror of the histogram (only with -s or --tonemap host)
  uint16_t *gray_ = nullptr;
  bool gray_is_big_endian_ = false;

  bool need_host_counts() const { return cfg_.inprogress_file != nullptr || cfg_.host_tonemap; }

  uint64_t pixel_count() const { return (uint64_t) cfg_.canvas.w * (uint64_t) cfg_.canvas.h; }
  uint64_t planes() const { return cfg_.n_channels > 0 ? (uint64_t) cfg_.n_channels : 1u; }
  uint64_t buffer_bytes() const { return planes() * pixel_count() * sizeof(cb_pixel); }

  void release() {  // cudabrot.cu:112-119
    for (cb_renderer *p : peers_) cb_renderer_destroy(p);
    peers_.clear();
    cb_renderer_destroy(renderer_);
    renderer_ = nullptr;
    free(gray_);
    gray_ = nullptr;
    free(counts_);
    counts_ = nullptr;
  }

  [[noreturn]] void die() {
    release();
    exit(1);
  }

  // The reference's device-error line (cudabrot.cu:134-141), wording kept: scripts may match on it.
  void check(int rc, const char *what, int line) {
    if (rc == 0) return;
    printf("CUDA error %d (%s) in %s, line %d (%s)\n", rc, cb_error_string(rc), __FILE__, line,
           what);
    die();
  }
#define CB_CHECK(call) check((call), #call, __LINE__)

  void setup() {  // cudabrot.cu:153-189
    float gpu_mib = (float) (buffer_bytes() + cb_rng_state_bytes(CB_DEFAULT_THREADS));
    gpu_mib /= (1024.0 * 1024.0);
    float cpu_mib = (float) (buffer_bytes() + pixel_count() * sizeof(uint16_t));
    cpu_mib /= (1024.0 * 1024.0);
    printf("Approximate memory needed: %.03f MiB GPU, %.03f MiB CPU\n", gpu_mib, cpu_mib);
    // CUDABROT_AMD_FAKE_GPUS=1: every rank on device -d (rehearsal of --gpus on a one-GPU box)
    const bool fake = cb_debug_knob("CUDABROT_AMD_FAKE_GPUS") != nullptr;
    for (int r = 0; r < cfg_.gpus; ++r) {
      cb_renderer *one = nullptr;
      const int device = cfg_.device + (fake ? 0 : r);
      const uint64_t first = (uint64_t) r * CB_DEFAULT_THREADS;
      if (cfg_.n_channels > 0) {
        CB_CHECK(cb_renderer_create_channels(&one, device, &cfg_.canvas, cfg_.channel_window, cfg_.n_channels,
                                             cfg_.seed, first, CB_DEFAULT_THREADS));
      } else {
        CB_CHECK(cb_renderer_create(&one, device, &cfg_.canvas, &cfg_.iterations, cfg_.seed, first,
                                    CB_DEFAULT_THREADS));
      }
      if (r == 0) {
        renderer_ = one;
      } else {
        peers_.push_back(one);
      }
    }
    if (need_host_counts()) {
      counts_ = (cb_pixel *) calloc(1, buffer_bytes());
      if (!counts_) die();
    }
    gray_ = (uint16_t *) calloc(pixel_count(), sizeof(uint16_t));
    if (!gray_) {
      printf("Failed allocating grayscale image.\n");
      die();
    }
  }

  void load_inprogress() {  // cudabrot.cu:215-258; the file work is state_files.cpp's
    if (!cfg_.inprogress_file) return;
    const cb::FileResult res = cb::load_state_file(cfg_.inprogress_file, (uint32_t) cfg_.canvas.w, (uint32_t) cfg_.canvas.h,
                                                   (uint32_t) planes(), counts_,
                                                   cfg_.raw_state ? cb::StateFormat::kRaw : cb::StateFormat::kNative);
    if (res == cb::FileResult::kError) die();
    if (res == cb::FileResult::kOk) CB_CHECK(cb_renderer_write_histogram(renderer_, counts_));
  }

  void save_inprogress() {  // cudabrot.cu:262-280
    if (!cfg_.inprogress_file) return;
    if (cb::save_state_file(cfg_.inprogress_file, (uint32_t) cfg_.canvas.w, (uint32_t) cfg_.canvas.h, (uint32_t) planes(),
                            counts_, cfg_.raw_state ? cb::StateFormat::kRaw : cb::StateFormat::kNative) ==
        cb::FileResult::kError) {
      die();
    }
  }

  // True-resume sidecar (extension, off unless --rng-state is given).  The -s buffer
  // is the histogram only, so the reference -- and this program by default -- replays seed 1337 from
  // the start when it resumes (cudabrot.cu:179,215-258).  The sidecar keeps the generator states of every
  // rank, so that buffer + sidecar continue the sample stream: P1 passes, save, resume, P2 passes gives the
  // histogram of one run of P1 + P2 passes -- with --gpus N as well (N generators, the same N to resume).
  uint64_t passes_before_ = 0, passes_this_run_ = 0;

  cb_renderer *rank_renderer(int r) { return r == 0 ? renderer_ : peers_[(size_t) r - 1]; }

  void load_rng_state() {
    if (!cfg_.rng_state_file) return;
    std::vector<std::vector<unsigned char>> blobs;
    const cb::FileResult res =
        cb::load_rng_sidecar(cfg_.rng_state_file, cfg_.seed, CB_DEFAULT_THREADS, (uint32_t) cfg_.gpus,
                             cb_rng_state_bytes(CB_DEFAULT_THREADS), &blobs, &passes_before_);
    if (res == cb::FileResult::kError) die();
    if (res != cb::FileResult::kOk) return;
    for (int r = 0; r < cfg_.gpus; ++r) {
      CB_CHECK(cb_renderer_write_rng_states(rank_renderer(r), blobs[(size_t) r].data()));
    }
  }

  void save_rng_state() {
    if (!cfg_.rng_state_file) return;
    std::vector<std::vector<unsigned char>> blobs((size_t) cfg_.gpus,
                                                  std::vector<unsigned char>(cb_rng_state_bytes(CB_DEFAULT_THREADS)));
    for (int r = 0; r < cfg_.gpus; ++r) {
      CB_CHECK(cb_renderer_read_rng_states(rank_renderer(r), blobs[(size_t) r].data()));
    }
    if (cb::save_rng_sidecar(cfg_.rng_state_file, cfg_.seed, CB_DEFAULT_THREADS, passes_before_ + passes_this_run_,
                             blobs) == cb::FileResult::kError) {
      die();
    }
  }

  // The pass loop (cudabrot.cu:471-501).  Launch length follows the measured pass time so that the
  // clock and the quit flag are looked at about every 0.2 s.
  void render() {
    printf("Calculating Buddhabrot.\n");
    const bool by_clock = cfg_.fixed_passes < 0;
    if (by_clock) {
      if (cfg_.seconds_to_run < 0) {
        printf("Press ctrl+C to finish.\n");
      } else {
        printf("Running for %.03f seconds.\n", cfg_.seconds_to_run);
      }
    }
    fflush(stdout);
    const int variant = cfg_.kernel_variant | (cfg_.burning_ship ? CB_KERNEL_FLAG_BURNING_SHIP : 0);
    // what the reference allocates in SetupCUDA, before its clock starts (cudabrot.cu:153-189,476)
    CB_CHECK(cb_renderer_prepare(renderer_, variant));
    for (cb_renderer *p : peers_) CB_CHECK(cb_renderer_prepare(p, variant));
    const double t0 = wall_seconds();
    // The pass loop.  One GPU: batches of `next` reference passes, sized so that the clock and the quit flag are
    // looked at every ~0.2 s.  --gpus N: render_sharded (persistent rank threads and a shared pass budget).
    const double launch_seconds = 0.2;
    long done = 0, next = 1;  // passes per rank
    if (cfg_.gpus > 1) {
      done = render_sharded(variant, by_clock, t0);
    } else {
    while (!g_quit_requested) {
      if (!by_clock) {
        if (done >= cfg_.fixed_passes) break;
        next = cfg_.fixed_passes - done;
        if (next > 256) next = 256;
      }
      CB_CHECK(cb_renderer_render_passes(renderer_, (uint32_t) next, variant));
      done += next;
      if (!by_clock) continue;
      const double elapsed = wall_seconds() - t0;
      if (cfg_.seconds_to_run >= 0 && elapsed > cfg_.seconds_to_run) break;
      double budget = launch_seconds;
      if (cfg_.seconds_to_run >= 0 && cfg_.seconds_to_run - elapsed < budget) {
        budget = cfg_.seconds_to_run - elapsed;
      }
      const double per_pass = elapsed / (double) done;
      next = per_pass > 0 ? (long) (budget / per_pass) : next * 2;
      if (next < 1) next = 1;
      if (next > 4096) next = 4096;
      if (next > 128) next -= next % 128;  // whole launches of 128 passes (cb_renderer's): a short launch drains badly
    }
    }
    passes_this_run_ = (uint64_t) done;  // per rank: what the generators have consumed
    done *= cfg_.gpus;                   // reference-sized passes over all ranks
    if (!peers_.empty()) {  // the one exchange of the path: sum the shards onto rank 0
      std::vector<cb_renderer *> all(1, renderer_);
      all.insert(all.end(), peers_.begin(), peers_.end());
      CB_CHECK(cb_renderers_reduce(all.data(), (int) all.size()));
    }
    if (need_host_counts()) {
      CB_CHECK(cb_renderer_read_histogram(renderer_, counts_));  // cudabrot.cu:496-497
    } else {
      CB_CHECK(cb_renderer_finish(renderer_));
    }
    printf("%ld Buddhabrot passes took %f seconds.\n", done, wall_seconds() - t0);
    if (cfg_.print_stats) print_stats();
    if (cfg_.n_channels == 0) tone_map(0);
  }

  // The pass loop of --gpus N.  One persistent host thread per rank; the ranks do not meet at batch
  // ends (boxes differ by 3-4 %, and every rendezvous would cost the faster ranks that much plus a restart of their
  // pipelines).  What they share is a pass BUDGET: `granted`, raised by the main thread alone -- which alone reads
  // the clock and the quit flag -- about 0.4 s ahead of the slowest rank; a rank renders whatever the budget allows
  // beyond its own count, in whole launches.  When the run ends (--passes reached, -t over, Ctrl+C) the budget is
  // frozen at what the furthest rank has been given and every rank completes it: an N-GPU run is "N T threads for
  // P passes" however it ends, which is what makes it reproducible and resumable (--rng-state).  A device error on
  // any rank stops the budget; it is reported from the main thread after every worker has been joined.
  long render_sharded(int variant, bool by_clock, double t0) {
    const int n = cfg_.gpus;
    struct Shared {
      std::mutex m;
      std::condition_variable cv;
      long granted = 0;
      bool closed = false;            // the budget is final
      std::vector<long> done, taken;  // per rank: passes rendered / rendered + in flight
      std::vector<int> rc;
    } sh;
    sh.done.assign((size_t) n, 0);
    sh.taken.assign((size_t) n, 0);
    sh.rc.assign((size_t) n, 0);
    std::vector<std::thread> workers;
    for (int k = 0; k < n; ++k) {
      workers.emplace_back([&, k] {
        cb_renderer *mine = rank_renderer(k);
        for (;;) {
          long piece = 0;
          {
            std::unique_lock<std::mutex> lock(sh.m);
            sh.cv.wait(lock, [&] { return sh.granted > sh.done[(size_t) k] || sh.closed; });
            piece = sh.granted - sh.done[(size_t) k];
            if (piece <= 0) return;       // closed and complete
            if (piece > 1024) piece = 1024;  // (a call returns when its launches are done: look at the budget again)
            sh.taken[(size_t) k] = sh.done[(size_t) k] + piece;
          }
          const int rc = cb_renderer_render_passes(mine, (uint32_t) piece, variant);
          std::lock_guard<std::mutex> lock(sh.m);
          if (rc != 0) {
            sh.rc[(size_t) k] = rc;
            sh.taken[(size_t) k] = sh.done[(size_t) k];
            sh.closed = true;             // no further grants; the others finish what they have taken
            sh.granted = 0;
            for (int j = 0; j < n; ++j) sh.granted = sh.taken[(size_t) j] > sh.granted ? sh.taken[(size_t) j] : sh.granted;
            sh.cv.notify_all();
            return;
          }
          sh.done[(size_t) k] += piece;
          sh.cv.notify_all();
        }
      });
    }
    {
      std::unique_lock<std::mutex> lock(sh.m);
      const long lead_min = 128;  // whole launches of cb_renderer (a short launch drains badly)
      if (!by_clock) {
        sh.granted = cfg_.fixed_passes > 0 ? cfg_.fixed_passes : 0;
        if (sh.granted == 0) sh.closed = true;
      } else {
        sh.granted = 1;  // the first pass calibrates the pace
      }
      sh.cv.notify_all();
      for (;;) {
        bool failed = false;
        long slowest = sh.granted, furthest = 0;
        for (int k = 0; k < n; ++k) {
          failed = failed || sh.rc[(size_t) k] != 0;
          slowest = sh.done[(size_t) k] < slowest ? sh.done[(size_t) k] : slowest;
          furthest = sh.taken[(size_t) k] > furthest ? sh.taken[(size_t) k] : furthest;
        }
        if (failed || sh.closed) break;
        const double elapsed = wall_seconds() - t0;
        const bool time_up = by_clock && cfg_.seconds_to_run >= 0 && elapsed > cfg_.seconds_to_run && slowest >= 1;
        const bool all_done = !by_clock && slowest >= sh.granted;
        if (g_quit_requested || time_up || all_done) {
          // final budget: what the furthest rank has been given (at least one pass, like the reference's loop)
          sh.granted = furthest > 1 ? furthest : 1;
          if (!by_clock && !g_quit_requested) sh.granted = cfg_.fixed_passes;
          sh.closed = true;
          sh.cv.notify_all();
          break;
        }
        if (by_clock && slowest >= 1) {
          const double per_pass = elapsed / (double) slowest;  // the slowest rank's pace
          // whole launches of cb_renderer, to the end: the clock acts at the granularity of a batch, as it does on one GPU
          // (a budget that shrank with the time left made the ranks issue 1-pass launches, which drain badly)
          const double ahead = 2.0 * launch_seconds_;
          long lead = per_pass > 0 ? (long) (ahead / per_pass) : lead_min;
          if (lead > 4096) lead = 4096;
          lead -= lead % lead_min;
          if (lead < lead_min) lead = lead_min;
          if (slowest + lead > sh.granted) {
            sh.granted = slowest + lead;
            sh.cv.notify_all();
          }
        }
        sh.cv.wait_for(lock, std::chrono::milliseconds(10));
      }
      // every rank completes the final budget
      sh.cv.wait(lock, [&] {
        for (int k = 0; k < n; ++k) {
          if (sh.rc[(size_t) k] == 0 && sh.done[(size_t) k] < sh.granted) return false;
        }
        return true;
      });
    }
    for (std::thread &w : workers) w.join();
    for (int r = 0; r < n; ++r) {
      if (sh.rc[(size_t) r] != 0) printf("GPU %d of %d:\n", r, n);
      CB_CHECK(sh.rc[(size_t) r]);
    }
    return sh.granted;
  }
  static constexpr double launch_seconds_ = 0.2;

  void tone_map(int plane) {
    uint64_t max = 0;
    double scale = 0.0;
    if (cfg_.host_tonemap) {
      cb_set_grayscale_pixels(counts_ + (uint64_t) plane * pixel_count(), cfg_.canvas.w, cfg_.canvas.h,
                              cfg_.gamma_correction, gray_, &max, &scale);
      gray_is_big_endian_ = false;
    } else {  // tone map on the device: only the 16-bit image crosses to the host
      CB_CHECK(cb_renderer_grayscale_plane(renderer_, plane, cfg_.gamma_correction, cfg_.tone_mode, gray_,
                                           &max, &scale));
      gray_is_big_endian_ = true;
    }
    printf("Max value: %lu, scale: %f\n", (unsigned long) max, scale);  // cudabrot.cu:437
  }

  // Fused multi-channel render: one image per window.
  void save_channels() {
    for (int j = 0; j < cfg_.n_channels; ++j) {
      printf("Channel %d: %d max iterations, %d min iterations.\n", j,
             cfg_.channel_window[j].max_escape_iterations, cfg_.channel_window[j].min_escape_iterations);
      tone_map(j);
      printf("Saving image.\n");
      save_image(cfg_.channel_file[j].c_str());
      printf("Done! Output image saved: %s\n", cfg_.channel_file[j].c_str());
    }
  }

  void print_stats() {
    cb_counters c;
    CB_CHECK(cb_renderer_read_counters(renderer_, &c));
    for (cb_renderer *p : peers_) {  // workload counters add up over the ranks (the clocks are rank 0's)
      cb_counters o;
      CB_CHECK(cb_renderer_read_counters(p, &o));
      c.samples += o.samples;
      c.rejected += o.rejected;
      c.never_escaped += o.never_escaped;
      c.too_fast += o.too_fast;
      c.recorded += o.recorded;
      c.iterate_steps += o.iterate_steps;
      c.replay_steps += o.replay_steps;
      c.increments += o.increments;
      c.skipped_steps += o.skipped_steps;
      c.status |= o.status;
    }
    // the level of the interior map every rank's last launch used (0: none), rank by rank
    std::string levels = std::to_string(cb_renderer_interior_map_level(renderer_));
    for (cb_renderer *p : peers_) levels += ", " + std::to_string(cb_renderer_interior_map_level(p));
    fprintf(stderr, "{\"interior_map_levels\": [%s], ", levels.c_str());
    fprintf(stderr,
            "\"samples\": %llu, \"rejected\": %llu, \"never_escaped\": %llu, \"too_fast\": %llu, "
            "\"recorded\": %llu, \"iterate_steps\": %llu, \"replay_steps\": %llu, "
            "\"increments\": %llu, \"skipped_steps\": %llu, \"status\": %llu, \"cycles_head\": %llu, "
            "\"cycles_long\": %llu, "
            "\"cycles_replay\": %llu, \"cycles_total\": %llu, \"rt_span\": %llu, \"rt_wave_life_sum\": %llu}\n",
            (unsigned long long) c.samples, (unsigned long long) c.rejected,
            (unsigned long long) c.never_escaped, (unsigned long long) c.too_fast,
            (unsigned long long) c.recorded, (unsigned long long) c.iterate_steps,
            (unsigned long long) c.replay_steps, (unsigned long long) c.increments,
            (unsigned long long) c.skipped_steps,
            (unsigned long long) c.status, (unsigned long long) c.cycles_head,
            (unsigned long long) c.cycles_long, (unsigned long long) c.cycles_replay,
            (unsigned long long) c.cycles_total,
            (unsigned long long) (c.rt_last_end ? c.rt_last_end - ~c.rt_not_first_start : 0),
            (unsigned long long) c.rt_wave_life_sum);
  }

  void save_image(const char *path) {  // cudabrot.cu:548-577: failures are reported and the run still ends with 0
    static const char *const kWhy[] = {nullptr, "Failed opening output image.",
                                       "Failed writing pgm header.", "Failed writing pixel data."};
    const int rc = gray_is_big_endian_ ? cb_save_image_be(path, gray_, cfg_.canvas.w, cfg_.canvas.h)
                                       : cb_save_image(path, gray_, cfg_.canvas.w, cfg_.canvas.h);
    if (rc >= 1 && rc <= 3) printf("%s\n", kWhy[rc]);
  }
#undef CB_CHECK
};

}  // namespace

// cudabrot.cu:756-760
extern "C" void on_sigint(int signal_number) {
  g_quit_requested = 1;
  printf("Signal %d received, waiting for current pass to finish...\n", signal_number);
}

int main(int argc, char **argv) {
  const Settings settings = parse_arguments(argc, argv);
  if (signal(SIGINT, on_sigint) == SIG_ERR) {  // cudabrot.cu:774-778
    printf("Failed setting signal handler.\n");
    return 1;
  }
  Run run(settings);
  return run.execute();
}
