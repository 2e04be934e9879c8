// FLAC stream decoder (host code only: no kernel in this file).  The reference reads LibriSpeech's .flac files through
// librosa.core.load -> audioread / libsndfile (datasets/corpus/ljspeech.py:17 `load_libre_2`, utils/audio.py:13-14);
// neither is in this image, so the container format is decoded here from its published specification
// (xiph.org/flac/format.html): STREAMINFO, frame headers with CRC-8, CONSTANT / VERBATIM / FIXED / LPC subframes,
// partitioned Rice residuals (4- and 5-bit parameters, escape partitions), wasted bits, left/side, side/right and
// mid/side decorrelation, frame CRC-16.  Output: interleaved int32 samples; the caller checks STREAMINFO's MD5 of the
// decoded PCM (utils/audio.py: _load_flac), which makes every real file its own known answer.
#include "common.h"
#include <stdlib.h>

namespace {
struct BitReader {
  const uint8_t* p; size_t n; size_t pos;      // pos in bits
  bool fail;
  inline uint32_t bit() {
    if ((pos >> 3) >= n) { fail = true; return 0; }
    const uint32_t b = (p[pos >> 3] >> (7 - (pos & 7))) & 1u;
    ++pos;
    return b;
  }
  inline uint64_t bits(int k) {                // k <= 57
    uint64_t v = 0;
    while (k > 0) {
      if ((pos >> 3) >= n) { fail = true; return 0; }
      const int avail = 8 - (int)(pos & 7), take = k < avail ? k : avail;
      const uint32_t byte = p[pos >> 3];
      v = (v << take) | ((byte >> (avail - take)) & ((1u << take) - 1u));
      pos += take; k -= take;
    }
    return v;
  }
  inline int64_t sbits(int k) {
    if (k == 0) return 0;
    const uint64_t v = bits(k);
    return (int64_t)(v << (64 - k)) >> (64 - k);
  }
  inline uint32_t unary() {                    // number of 0 bits in front of the next 1
    uint32_t q = 0;
    for (;;) {
      if ((pos >> 3) >= n) { fail = true; return 0; }
      const int avail = 8 - (int)(pos & 7);
      const uint32_t rest = p[pos >> 3] & ((1u << avail) - 1u);
      if (rest == 0) { q += avail; pos += avail; continue; }
      const int lead = __builtin_clz(rest) - (32 - avail);
      q += lead; pos += lead + 1;
      return q;
    }
  }
  inline void align() { pos = (pos + 7) & ~(size_t)7; }
};

uint8_t crc8(const uint8_t* d, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= d[i];
    for (int b = 0; b < 8; ++b) c = (c & 0x80) ? (uint8_t)((c << 1) ^ 0x07) : (uint8_t)(c << 1);
  }
  return c;
}
uint16_t crc16(const uint8_t* d, size_t n) {
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= (uint16_t)d[i] << 8;
    for (int b = 0; b < 8; ++b) c = (c & 0x8000) ? (uint16_t)((c << 1) ^ 0x8005) : (uint16_t)(c << 1);
  }
  return c;
}

struct StreamInfo { int min_block, max_block, rate, channels, bps; int64_t total; uint8_t md5[16]; size_t first_frame; };

int parse_header(const uint8_t* d, size_t n, StreamInfo* si) {
  if (n < 4 + 4 + 34 || memcmp(d, "fLaC", 4) != 0) { ns_set_error("flac: no fLaC marker"); return NS_ERR_BAD_ARG; }
  size_t off = 4;
  bool have = false;
  for (;;) {
    if (off + 4 > n) { ns_set_error("flac: truncated metadata"); return NS_ERR_BAD_ARG; }
    const bool last = d[off] & 0x80;
    const int type = d[off] & 0x7f;
    const size_t len = ((size_t)d[off + 1] << 16) | ((size_t)d[off + 2] << 8) | d[off + 3];
    off += 4;
    if (off + len > n) { ns_set_error("flac: truncated metadata block"); return NS_ERR_BAD_ARG; }
    if (type == 0) {
      if (len < 34) { ns_set_error("flac: short STREAMINFO"); return NS_ERR_BAD_ARG; }
      BitReader br{d + off, len, 0, false};
      si->min_block = (int)br.bits(16); si->max_block = (int)br.bits(16);
      br.bits(24); br.bits(24);
      si->rate = (int)br.bits(20); si->channels = (int)br.bits(3) + 1; si->bps = (int)br.bits(5) + 1;
      si->total = (int64_t)br.bits(36);
      memcpy(si->md5, d + off + 18, 16);
      have = true;
    }
    off += len;
    if (last) break;
  }
  if (!have) { ns_set_error("flac: no STREAMINFO block"); return NS_ERR_BAD_ARG; }
  si->first_frame = off;
  return NS_OK;
}

int decode_residual(BitReader& br, int32_t* res, int blocksize, int order) {
  const int method = (int)br.bits(2);
  if (method > 1) { ns_set_error("flac: reserved residual coding method"); return NS_ERR_BAD_ARG; }
  const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
  const int porder = (int)br.bits(4);
  const int parts = 1 << porder;
  if ((blocksize >> porder) << porder != blocksize && porder > 0) { ns_set_error("flac: partition order does not divide the block"); return NS_ERR_BAD_ARG; }
  int i = order;
  for (int pt = 0; pt < parts; ++pt) {
    int count = (blocksize >> porder) - (pt == 0 ? order : 0);
    if (count < 0) { ns_set_error("flac: predictor order exceeds the first partition"); return NS_ERR_BAD_ARG; }
    const int k = (int)br.bits(pbits);
    if (k == esc) {
      const int raw = (int)br.bits(5);
      for (int j = 0; j < count; ++j) res[i++] = (int32_t)br.sbits(raw);
    } else {
      for (int j = 0; j < count; ++j) {
        const uint32_t q = br.unary();
        const uint32_t u = (q << k) | (uint32_t)br.bits(k);
        res[i++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
      }
    }
    if (br.fail) { ns_set_error("flac: truncated residual"); return NS_ERR_BAD_ARG; }
  }
  return NS_OK;
}

int decode_subframe(BitReader& br, int64_t* out, int32_t* res, int blocksize, int bps) {
  if (br.bit()) { ns_set_error("flac: subframe padding bit set"); return NS_ERR_BAD_ARG; }
  const int type = (int)br.bits(6);
  int wasted = 0;
  if (br.bit()) wasted = (int)br.unary() + 1;
  bps -= wasted;
  if (bps <= 0) { ns_set_error("flac: wasted bits exceed the sample size"); return NS_ERR_BAD_ARG; }
  // A well-formed stream never predicts a sample outside its size; a damaged one can make the predictor run away until
  // the 64-bit sums overflow (found by the UBSan run of tests/test_sanitizer_cpu.py).  One bit of slack, then refuse.
  const int64_t lim = (int64_t)1 << (bps + 1);
  auto in_range = [&](int64_t v) { return v > -lim && v < lim; };
  if (type == 0) {
    const int64_t v = br.sbits(bps);
    for (int i = 0; i < blocksize; ++i) out[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < blocksize; ++i) out[i] = br.sbits(bps);
  } else if (type >= 8 && type <= 12) {
    const int order = type - 8;
    if (order > blocksize) { ns_set_error("flac: fixed order exceeds the block"); return NS_ERR_BAD_ARG; }
    for (int i = 0; i < order; ++i) out[i] = br.sbits(bps);
    const int rc = decode_residual(br, res, blocksize, order);
    if (rc) return rc;
    for (int i = order; i < blocksize; ++i) {
      int64_t pred = 0;
      switch (order) {
        case 1: pred = out[i - 1]; break;
        case 2: pred = 2 * out[i - 1] - out[i - 2]; break;
        case 3: pred = 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
        case 4: pred = 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
        default: break;
      }
      out[i] = pred + res[i];
      if (!in_range(out[i])) { ns_set_error("flac: predicted sample outside the sample size (damaged stream)"); return NS_ERR_BAD_ARG; }
    }
  } else if (type >= 32) {
    const int order = (type & 31) + 1;
    if (order > blocksize) { ns_set_error("flac: LPC order exceeds the block"); return NS_ERR_BAD_ARG; }
    for (int i = 0; i < order; ++i) out[i] = br.sbits(bps);
    const int prec = (int)br.bits(4) + 1;
    if (prec == 16) { ns_set_error("flac: reserved LPC precision"); return NS_ERR_BAD_ARG; }
    const int shift = (int)br.sbits(5);
    if (shift < 0) { ns_set_error("flac: negative LPC shift"); return NS_ERR_BAD_ARG; }
    int64_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = br.sbits(prec);
    const int rc = decode_residual(br, res, blocksize, order);
    if (rc) return rc;
    for (int i = order; i < blocksize; ++i) {
      int64_t s = 0;
      for (int j = 0; j < order; ++j) s += coef[j] * out[i - 1 - j];
      out[i] = (s >> shift) + res[i];
      if (!in_range(out[i])) { ns_set_error("flac: predicted sample outside the sample size (damaged stream)"); return NS_ERR_BAD_ARG; }
    }
  } else {
    ns_set_error("flac: reserved subframe type %d", type);
    return NS_ERR_BAD_ARG;
  }
  if (wasted)
    for (int i = 0; i < blocksize; ++i) out[i] = out[i] * ((int64_t)1 << wasted);
  if (br.fail) { ns_set_error("flac: truncated subframe"); return NS_ERR_BAD_ARG; }
  return NS_OK;
}
}  // namespace

/* Stream parameters of a FLAC file held in memory (total_samples = samples per channel; 0 if the encoder left it out). */
extern "C" int ns_flac_info(const uint8_t* data, size_t n, int* sample_rate, int* channels, int* bits_per_sample,
                            int64_t* total_samples, uint8_t* md5_16) {
  NS_CHECK_ARG(data && sample_rate && channels && bits_per_sample && total_samples, "ns_flac_info: null");
  StreamInfo si;
  const int rc = parse_header(data, n, &si);
  if (rc) return rc;
  *sample_rate = si.rate; *channels = si.channels; *bits_per_sample = si.bps; *total_samples = si.total;
  if (md5_16) memcpy(md5_16, si.md5, 16);
  return NS_OK;
}

/* Decodes every frame into out[sample * channels + channel] (int32, sign-extended to the stream's sample size);
 * capacity and *decoded count samples per channel.  Frames are checked against their CRC-8 / CRC-16. */
extern "C" int ns_flac_decode(const uint8_t* data, size_t n, int32_t* out, int64_t capacity, int64_t* decoded) {
  NS_CHECK_ARG(data && out && decoded, "ns_flac_decode: null");
  StreamInfo si;
  int rc = parse_header(data, n, &si);
  if (rc) return rc;
  const int C = si.channels;
  const int maxb = si.max_block > 0 ? si.max_block : 65535;
  int64_t* ch[8];
  int64_t* buf = (int64_t*)malloc(sizeof(int64_t) * (size_t)maxb * C);
  int32_t* res = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxb);
  if (!buf || !res) { free(buf); free(res); ns_set_error("flac: out of memory"); return NS_ERR_BAD_ARG; }
  for (int c = 0; c < C; ++c) ch[c] = buf + (size_t)c * maxb;
  int64_t done = 0;
  size_t off = si.first_frame;
  static const int kBlock[16] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
  static const int kBps[8] = {0, 8, 12, 0, 16, 20, 24, 0};
  rc = NS_OK;
  while (off + 2 <= n && rc == NS_OK) {
    if (data[off] != 0xFF || (data[off + 1] & 0xFC) != 0xF8) {      // trailing padding / ID3 tags end the audio
      if (si.total && done >= si.total) break;
      // no total in STREAMINFO: whatever follows the last whole frame (a tag, padding) ends the audio; the caller still
      // has STREAMINFO's MD5 of the PCM to tell a cut stream from a complete one
      if (!si.total && done > 0) break;
      ns_set_error("flac: lost frame sync at byte %zu", off); rc = NS_ERR_BAD_ARG; break;
    }
    BitReader br{data + off, n - off, 0, false};
    br.bits(14); br.bit(); br.bit();
    const int bcode = (int)br.bits(4), rcode = (int)br.bits(4), chan = (int)br.bits(4), scode = (int)br.bits(3);
    if (br.bit()) { ns_set_error("flac: reserved header bit"); rc = NS_ERR_BAD_ARG; break; }
    {  // UTF-8 style frame / sample number
      const uint32_t first = (uint32_t)br.bits(8);
      int extra = 0;
      if (first >= 0xFE) extra = 6; else if (first >= 0xFC) extra = 5; else if (first >= 0xF8) extra = 4;
      else if (first >= 0xF0) extra = 3; else if (first >= 0xE0) extra = 2; else if (first >= 0xC0) extra = 1;
      else if (first >= 0x80) { ns_set_error("flac: bad frame number"); rc = NS_ERR_BAD_ARG; break; }
      for (int i = 0; i < extra; ++i) br.bits(8);
    }
    int blocksize = kBlock[bcode];
    if (bcode == 6) blocksize = (int)br.bits(8) + 1;
    else if (bcode == 7) blocksize = (int)br.bits(16) + 1;
    if (rcode == 12) br.bits(8); else if (rcode == 13 || rcode == 14) br.bits(16);
    else if (rcode == 15) { ns_set_error("flac: invalid sample rate code"); rc = NS_ERR_BAD_ARG; break; }
    const size_t hdr_bytes = br.pos >> 3;
    const uint8_t want8 = (uint8_t)br.bits(8);
    if (br.fail || blocksize <= 0 || blocksize > maxb) { ns_set_error("flac: bad frame header"); rc = NS_ERR_BAD_ARG; break; }
    if (crc8(data + off, hdr_bytes) != want8) { ns_set_error("flac: frame header CRC-8 mismatch at byte %zu", off); rc = NS_ERR_BAD_ARG; break; }
    const int bps = scode == 0 ? si.bps : kBps[scode];
    if (bps == 0) { ns_set_error("flac: reserved sample size"); rc = NS_ERR_BAD_ARG; break; }
    int nch = chan < 8 ? chan + 1 : 2;
    if (chan > 10 || nch != C) { ns_set_error("flac: channel assignment %d does not match the stream", chan); rc = NS_ERR_BAD_ARG; break; }
    for (int c = 0; c < nch && rc == NS_OK; ++c) {
      const bool side = (chan == 8 && c == 1) || (chan == 9 && c == 0) || (chan == 10 && c == 1);
      rc = decode_subframe(br, ch[c], res, blocksize, bps + (side ? 1 : 0));
    }
    if (rc) break;
    if (chan == 8) { for (int i = 0; i < blocksize; ++i) ch[1][i] = ch[0][i] - ch[1][i]; }
    else if (chan == 9) { for (int i = 0; i < blocksize; ++i) ch[0][i] = ch[0][i] + ch[1][i]; }
    else if (chan == 10) {
      for (int i = 0; i < blocksize; ++i) {
        const int64_t sd = ch[1][i], mid = ch[0][i] * 2 + (sd & 1);      // not "<< 1": a negative mid is a sample, not a bit pattern
        ch[0][i] = (mid + sd) >> 1;
        ch[1][i] = (mid - sd) >> 1;
      }
    }
    br.align();
    const size_t body = br.pos >> 3;
    const uint16_t want16 = (uint16_t)br.bits(16);
    if (br.fail) { ns_set_error("flac: truncated frame"); rc = NS_ERR_BAD_ARG; break; }
    if (crc16(data + off, body) != want16) { ns_set_error("flac: frame CRC-16 mismatch at byte %zu", off); rc = NS_ERR_BAD_ARG; break; }
    if (done + blocksize > capacity) { ns_set_error("flac: output buffer too small"); rc = NS_ERR_SHORT_BUFFER; break; }
    for (int i = 0; i < blocksize; ++i)
      for (int c = 0; c < C; ++c) out[(done + i) * C + c] = (int32_t)ch[c][i];
    done += blocksize;
    off += br.pos >> 3;
  }
  free(buf); free(res);
  *decoded = done;
  return rc;
}
