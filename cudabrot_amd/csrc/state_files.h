// state_files.h -- the binary's two checkpoint files: the -s histogram buffer (cudabrot.cu:215-280) and the
// --rng-state sidecar.  Host code without a device in sight, so that the size checks, the
// short-read paths and the format decisions can be driven by a sanitized CPU build (tests/asan).
// Every function prints the line the binary shows for the outcome (all messages go to stdout like the
// reference's, cudabrot.cu:134-141) and returns what happened; the caller decides about exiting.
#pragma once

#include <stdint.h>

#include <vector>

#include "../../include/cudabrot_amd.h"

namespace cb {

enum class FileResult {
  kOk,      // loaded / written
  kAbsent,  // load only: the file does not exist yet (the reference carries on, cudabrot.cu:227-231)
  kError,   // message printed; the reference exits 1 here
};

// The -s file: this header, then planes * h * w counters of counter_bytes each, row 0 = min_imag.  The
// reference's own file is the bare counters as uint32[h][w] (cudabrot.cu:262-280): such a file -- no magic,
// exactly w*h*4 bytes, one plane -- is accepted on load, announced and widened.
struct StateHeader {
  char magic[8];  // "CBHIST64"
  uint32_t w, h, planes, counter_bytes;
  uint64_t reserved;
};
static_assert(sizeof(StateHeader) == 32, "the -s header is 32 bytes");

// --state-format raw (default: native, the header above): the file is the bare counters, as the reference writes it
// (cudabrot.cu:262-280) -- uint32[h][w] when there is one plane and every count fits 32 bits, so that the reference
// (or a script that np.fromfile()s the buffer) reads it back; else uint64, announced.  On load, raw accepts a
// headerless file of exactly the 32-bit or the 64-bit size (with an explicit flag the size is not a guess).
enum class StateFormat { kNative, kRaw };

FileResult load_state_file(const char *path, uint32_t w, uint32_t h, uint32_t planes, cb_pixel *counts,
                           StateFormat format = StateFormat::kNative);
FileResult save_state_file(const char *path, uint32_t w, uint32_t h, uint32_t planes, const cb_pixel *counts,
                           StateFormat format = StateFormat::kNative);

// The sidecar: this header, then for every rank {uint64 first_subsequence = rank * n_threads,
// cb_rng_state_bytes(n_threads) bytes}.  With --gpus N it holds N generators and resumes only an N-GPU run.
struct RngStateHeader {
  char magic[8];  // "CBRNGST2"
  uint64_t seed, passes_done;
  uint32_t n_threads, n_ranks;
};
static_assert(sizeof(RngStateHeader) == 32, "the sidecar header is 32 bytes");

// blobs: n_ranks vectors of blob_bytes each (resized here on load).
FileResult load_rng_sidecar(const char *path, uint64_t seed, uint32_t n_threads, uint32_t n_ranks, size_t blob_bytes,
                            std::vector<std::vector<unsigned char>> *blobs, uint64_t *passes_done);
FileResult save_rng_sidecar(const char *path, uint64_t seed, uint32_t n_threads, uint64_t passes_done,
                            const std::vector<std::vector<unsigned char>> &blobs);

}  // namespace cb
