// state_files.cpp -- see state_files.h.
#include "state_files.h"

#include <errno.h>
#include <stdio.h>
#include <string.h>

namespace cb {

namespace {

const char kStateMagic[8] = {'C', 'B', 'H', 'I', 'S', 'T', '6', '4'};
const char kRngMagic[8] = {'C', 'B', 'R', 'N', 'G', 'S', 'T', '2'};

struct File {  // closes on every path out
  FILE *f;
  explicit File(FILE *f_) : f(f_) {}
  ~File() {
    if (f) fclose(f);
  }
  File(const File &) = delete;
  File &operator=(const File &) = delete;
};

}  // namespace

FileResult load_state_file(const char *path, uint32_t w, uint32_t h, uint32_t planes, cb_pixel *counts,
                           StateFormat format) {
  const uint64_t pixels = (uint64_t) w * (uint64_t) h;
  const uint64_t native_bytes = (uint64_t) planes * pixels * sizeof(cb_pixel);
  File in(fopen(path, "rb"));
  printf("Loading previous image state from %s.\n", path);  // cudabrot.cu:224
  if (!in.f) {
    if (errno == ENOENT) {
      printf("File %s doesn't exist yet. Not loading.\n", path);  // cudabrot.cu:228
      return FileResult::kAbsent;
    }
    printf("Failed opening %s: %s\n", path, strerror(errno));
    return FileResult::kError;
  }
  // size check, cudabrot.cu:192-211,236-245
  long size = -1;
  if (fseek(in.f, 0, SEEK_END) != 0) {
    printf("Failed seeking file end: %s\n", strerror(errno));
    return FileResult::kError;
  }
  if ((size = ftell(in.f)) < 0) {
    printf("Failed reading file size: %s\n", strerror(errno));
    return FileResult::kError;
  }
  if (fseek(in.f, 0, SEEK_SET) != 0) {
    printf("Failed seeking file start: %s\n", strerror(errno));
    return FileResult::kError;
  }
  // The format is decided by the magic, never by the size alone: a native buffer of half the pixel count has
  // exactly the reference size of this canvas.
  StateHeader hd;
  memset(&hd, 0, sizeof(hd));
  const bool has_header = (uint64_t) size >= sizeof(hd) && fread(&hd, sizeof(hd), 1, in.f) == 1 &&
                          memcmp(hd.magic, kStateMagic, 8) == 0;
  const uint64_t narrow_bytes = pixels * sizeof(uint32_t);
  bool ok;
  if (has_header) {
    if (hd.w != w || hd.h != h || hd.planes != planes || hd.counter_bytes != sizeof(cb_pixel) ||
        (uint64_t) size != sizeof(hd) + native_bytes) {
      printf("%s holds a %ux%u buffer of %u plane(s), %u-byte counters, %ld bytes.\n", path, hd.w, hd.h, hd.planes,
             hd.counter_bytes, size);
      printf("The size of %s doesn't match the expected size of %lu bytes.\n", path,
             (unsigned long) (sizeof(hd) + native_bytes));  // cudabrot.cu:240-242
      return FileResult::kError;
    }
    ok = native_bytes == 0 || fread(counts, native_bytes, 1, in.f) == 1;
  } else if ((uint64_t) size == narrow_bytes && planes == 1) {
    printf("%s has no header and the size of 32-bit counters: read as the reference's format.\n", path);
    ok = fseek(in.f, 0, SEEK_SET) == 0;
    std::vector<uint32_t> row(w);
    for (uint64_t r = 0; ok && r < h; ++r) {  // a row at a time: no second buffer of the canvas' size
      ok = fread(row.data(), sizeof(uint32_t), w, in.f) == w;
      for (uint32_t c = 0; ok && c < w; ++c) counts[r * w + c] = row[c];
    }
  } else if (format == StateFormat::kRaw && (uint64_t) size == native_bytes) {
    printf("%s has no header and the size of 64-bit counters: read as raw 64-bit counters.\n", path);
    ok = fseek(in.f, 0, SEEK_SET) == 0 && (native_bytes == 0 || fread(counts, native_bytes, 1, in.f) == 1);
  } else {
    printf("The size of %s doesn't match the expected size of %lu bytes.\n", path,
           (unsigned long) (format == StateFormat::kRaw ? native_bytes : sizeof(hd) + native_bytes));
    return FileResult::kError;
  }
  if (!ok) {
    printf("Failed reading %s: %s\n", path, strerror(errno));  // cudabrot.cu:250-252
    return FileResult::kError;
  }
  return FileResult::kOk;
}

FileResult save_state_file(const char *path, uint32_t w, uint32_t h, uint32_t planes, const cb_pixel *counts,
                           StateFormat format) {
  const uint64_t native_bytes = (uint64_t) planes * (uint64_t) w * (uint64_t) h * sizeof(cb_pixel);
  printf("Saving in-progress buffer to %s.\n", path);  // cudabrot.cu:265
  File out(fopen(path, "wb"));
  if (!out.f) {
    printf("Failed opening %s: %s\n", path, strerror(errno));
    return FileResult::kError;
  }
  if (format == StateFormat::kRaw) {  // the reference's file: bare counters (cudabrot.cu:262-280)
    const uint64_t n = (uint64_t) planes * (uint64_t) w * (uint64_t) h;
    bool narrow = planes == 1;
    for (uint64_t k = 0; narrow && k < n; ++k) narrow = counts[k] <= 0xffffffffull;
    bool ok = true;
    if (narrow) {  // uint32[h][w], what the reference's LoadInProgressBuffer expects (cudabrot.cu:236-245)
      std::vector<uint32_t> row(w);
      for (uint64_t r = 0; ok && r < h; ++r) {
        for (uint32_t c = 0; c < w; ++c) row[c] = (uint32_t) counts[r * w + c];
        ok = w == 0 || fwrite(row.data(), sizeof(uint32_t), w, out.f) == w;
      }
    } else {
      printf("%s: raw 64-bit counters (%s): not a file the reference can load.\n", path,
             planes == 1 ? "a count exceeds 32 bits" : "more than one plane");
      ok = native_bytes == 0 || fwrite(counts, native_bytes, 1, out.f) == 1;
    }
    if (!ok) {
      printf("Failed writing data to %s: %s\n", path, strerror(errno));  // cudabrot.cu:275-277
      return FileResult::kError;
    }
    return FileResult::kOk;
  }
  StateHeader hd;
  memset(&hd, 0, sizeof(hd));
  memcpy(hd.magic, kStateMagic, 8);
  hd.w = w;
  hd.h = h;
  hd.planes = planes;
  hd.counter_bytes = (uint32_t) sizeof(cb_pixel);
  if (fwrite(&hd, sizeof(hd), 1, out.f) != 1 || (native_bytes != 0 && fwrite(counts, native_bytes, 1, out.f) != 1)) {
    printf("Failed writing data to %s: %s\n", path, strerror(errno));  // cudabrot.cu:275-277
    return FileResult::kError;
  }
  return FileResult::kOk;
}

FileResult load_rng_sidecar(const char *path, uint64_t seed, uint32_t n_threads, uint32_t n_ranks, size_t blob_bytes,
                            std::vector<std::vector<unsigned char>> *blobs, uint64_t *passes_done) {
  File in(fopen(path, "rb"));
  printf("Loading generator state from %s.\n", path);
  if (!in.f) {
    if (errno == ENOENT) {
      printf("File %s doesn't exist yet. Not loading.\n", path);
      return FileResult::kAbsent;
    }
    printf("Failed opening %s: %s\n", path, strerror(errno));
    return FileResult::kError;
  }
  RngStateHeader hd;
  bool ok = fread(&hd, sizeof(hd), 1, in.f) == 1 && memcmp(hd.magic, kRngMagic, 8) == 0 && hd.seed == seed &&
            hd.n_threads == n_threads && hd.n_ranks == n_ranks;
  blobs->assign(n_ranks, std::vector<unsigned char>(blob_bytes));
  for (uint32_t r = 0; ok && r < n_ranks; ++r) {
    uint64_t first = ~0ull;
    ok = fread(&first, sizeof(first), 1, in.f) == 1 && first == (uint64_t) r * n_threads &&
         fread((*blobs)[r].data(), 1, blob_bytes, in.f) == blob_bytes;
  }
  if (ok) ok = fgetc(in.f) == EOF;  // nothing behind the last generator
  if (!ok) {
    printf("%s is not a generator state for seed %lu, %u threads and %u GPU(s).\n", path, (unsigned long) seed,
           (unsigned) n_threads, (unsigned) n_ranks);
    return FileResult::kError;
  }
  *passes_done = hd.passes_done;
  printf("Continuing the sample stream after %lu passes.\n", (unsigned long) hd.passes_done);
  return FileResult::kOk;
}

FileResult save_rng_sidecar(const char *path, uint64_t seed, uint32_t n_threads, uint64_t passes_done,
                            const std::vector<std::vector<unsigned char>> &blobs) {
  printf("Saving generator state to %s.\n", path);
  RngStateHeader hd;
  memset(&hd, 0, sizeof(hd));
  memcpy(hd.magic, kRngMagic, 8);
  hd.seed = seed;
  hd.passes_done = passes_done;
  hd.n_threads = n_threads;
  hd.n_ranks = (uint32_t) blobs.size();
  File out(fopen(path, "wb"));
  if (!out.f) {
    printf("Failed opening %s: %s\n", path, strerror(errno));
    return FileResult::kError;
  }
  bool ok = fwrite(&hd, sizeof(hd), 1, out.f) == 1;
  for (size_t r = 0; ok && r < blobs.size(); ++r) {
    const uint64_t first = (uint64_t) r * n_threads;
    ok = fwrite(&first, sizeof(first), 1, out.f) == 1 &&
         (blobs[r].empty() || fwrite(blobs[r].data(), blobs[r].size(), 1, out.f) == 1);
  }
  if (!ok) {
    printf("Failed writing data to %s: %s\n", path, strerror(errno));
    return FileResult::kError;
  }
  return FileResult::kOk;
}

}  // namespace cb
