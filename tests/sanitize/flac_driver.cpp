// CPU sanitizer target ("compile host C++ with -fsanitize=address,undefined in tests"): the FLAC stream decoder
// of csrc/flac.hip - the one piece of the library that parses untrusted bytes on the host - built host-only with
// AddressSanitizer + UndefinedBehaviorSanitizer and driven over the streams tests/test_sanitizer_cpu.py writes: good
// files of every coding variant, and for each one truncations and single-byte corruptions, which must come back as
// error codes with no out-of-bounds access, no signed overflow, no misaligned or uninitialised read on the way.
//   argv[1..]: .flac files.  Prints one line per file: "<file> rc <code> decoded <n> sum <checksum>".
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/nspeech_hip.h"

static char g_err[512];
void ns_set_error(const char* fmt, ...) {       // core.hip's, restated: that file holds kernels and cannot be built host-only
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int run(const std::vector<uint8_t>& data, const char* tag) {
  int sr = 0, ch = 0, bps = 0;
  int64_t total = 0;
  uint8_t md5[16];
  int rc = ns_flac_info(data.data(), data.size(), &sr, &ch, &bps, &total, md5);
  int64_t decoded = 0;
  unsigned long long sum = 0;
  if (rc == 0 && ch > 0 && ch <= 8) {
    // exact capacity when the total is known (one more sample would be an overflow the sanitizer must not see), a
    // bounded guess otherwise
    const int64_t cap = total > 0 ? total : (int64_t)data.size() * 4 + 65536;
    std::vector<int32_t> out((size_t)cap * ch);
    rc = ns_flac_decode(data.data(), data.size(), out.data(), cap, &decoded);
    if (rc == 0)
      for (int64_t i = 0; i < decoded * ch; ++i) sum = sum * 1315423911ull + (unsigned)out[(size_t)i];
  }
  printf("%s rc %d decoded %lld sum %llu\n", tag, rc, (long long)decoded, sum);
  return rc;
}

int main(int argc, char** argv) {
  int good = 0;
  for (int a = 1; a < argc; ++a) {
    FILE* f = fopen(argv[a], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(f);
    if (run(data, argv[a]) == 0) ++good;
    // truncations: every prefix length on a coarse grid + the last 40 bytes one by one
    for (size_t cut = 0; cut < data.size(); cut += (cut + 40 < data.size() ? 97 : 1)) {
      std::vector<uint8_t> t(data.begin(), data.begin() + cut);
      char tag[64];
      snprintf(tag, sizeof(tag), "  cut@%zu", cut);
      run(t, tag);
    }
    // single-byte corruptions on a grid (headers, residuals, CRCs)
    for (size_t pos = 0; pos < data.size(); pos += 53) {
      std::vector<uint8_t> t(data);
      t[pos] ^= 0x5a;
      char tag[64];
      snprintf(tag, sizeof(tag), "  flip@%zu", pos);
      run(t, tag);
    }
  }
  return good == argc - 1 ? 0 : 1;
}
