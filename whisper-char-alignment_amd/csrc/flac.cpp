// Host-side FLAC decoder (no GPU code): the reference reads LibriSpeech's .flac files through torchaudio.load
// (dataset.py:31,104); neither torchaudio nor libFLAC is part of this engine's dependencies, so the (small, public)
// FLAC subset the corpora use is decoded here: STREAMINFO, fixed- and variable-blocksize frames, CONSTANT / VERBATIM /
// FIXED (order 0-4) / LPC (order 1-32) subframes, Rice and Rice2 coded residuals incl. escaped partitions, wasted bits,
// independent / left-side / right-side / mid-side stereo, 4-32 bit samples (up to 8 channels), CRC-8 / CRC-16 checked.
// Called from Python reader threads through ctypes (the GIL is released for the whole decode).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/wca.h"

// Fuzzing builds only (tools/flac_fuzz.cpp, tests/test_flac_fuzz.py: g++ -fsanitize=address,undefined on the CPU): with
// -DWCA_FLAC_FUZZ_SKIP_CRC the frame CRCs are computed but not enforced, so that mutated streams reach the subframe / residual
// decoders instead of being rejected at the checksum. The product build (build.py) never defines it.
#ifdef WCA_FLAC_FUZZ_SKIP_CRC
#define WCA_FLAC_CRC_MISMATCH(a, b) (((a) != (b)) && false)
#else
#define WCA_FLAC_CRC_MISMATCH(a, b) ((a) != (b))
#endif

namespace {

struct BitReader {
  const uint8_t* p;
  size_t n, pos = 0;  // byte position of the next byte to load
  uint64_t acc = 0;
  int bits = 0;  // valid bits in acc (MSB-aligned at bit `bits - 1`)
  bool bad = false;
  BitReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
  inline void fill() {
    while (bits <= 56 && pos < n) {
      acc = (acc << 8) | p[pos++];
      bits += 8;
    }
  }
  inline uint32_t read(int k) {  // k in [0, 32]
    if (k == 0) return 0;
    if (bits < k) fill();
    if (bits < k) {
      bad = true;
      return 0;
    }
    const uint32_t v = (uint32_t)((acc >> (bits - k)) & ((k == 32) ? 0xffffffffull : ((1ull << k) - 1)));
    bits -= k;
    return v;
  }
  inline int32_t read_signed(int k) {
    if (k == 0) return 0;
    const uint32_t v = read(k);
    if (k == 32) return (int32_t)v;
    const uint32_t m = 1u << (k - 1);
    return (int32_t)((v ^ m) - m);
  }
  inline uint32_t read_unary() {  // number of 0 bits before the next 1 bit
    uint32_t q = 0;
    for (;;) {
      if (bits == 0) fill();
      if (bits == 0) {
        bad = true;
        return 0;
      }
      const uint64_t window = acc & ((bits == 64) ? ~0ull : ((1ull << bits) - 1));
      if (window == 0) {
        q += bits;
        bits = 0;
        continue;
      }
      const int lead = __builtin_clzll(window) - (64 - bits);
      q += lead;
      bits -= lead + 1;
      return q;
    }
  }
  inline void align_byte() { bits -= bits & 7; }
  inline size_t byte_pos() const { return pos - (size_t)(bits >> 3); }  // valid only when byte aligned
};

uint8_t crc8(const uint8_t* d, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= d[i];
    for (int b = 0; b < 8; ++b) c = (c & 0x80) ? (uint8_t)((c << 1) ^ 0x07) : (uint8_t)(c << 1);
  }
  return c;
}

// built once, thread-safe (C++11 magic static): the decoder is called from several reader threads at once
struct Crc16Table {
  uint16_t t[256];
  Crc16Table() {
    for (int i = 0; i < 256; ++i) {
      uint16_t c = (uint16_t)(i << 8);
      for (int b = 0; b < 8; ++b) c = (c & 0x8000) ? (uint16_t)((c << 1) ^ 0x8005) : (uint16_t)(c << 1);
      t[i] = c;
    }
  }
};

uint16_t crc16(const uint8_t* d, size_t n) {
  static const Crc16Table table;
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ table.t[((c >> 8) ^ d[i]) & 0xff]);
  return c;
}

struct StreamInfo {
  int sample_rate = 0, channels = 0, bps = 0;
  int64_t total = 0;
  size_t audio_off = 0;
};

int parse_header(const uint8_t* buf, size_t n, StreamInfo* si) {
  if (n < 42 || memcmp(buf, "fLaC", 4) != 0) return -1;
  size_t pos = 4;
  bool have = false;
  for (;;) {
    if (pos + 4 > n) return -1;
    const bool last = buf[pos] & 0x80;
    const int type = buf[pos] & 0x7f;
    const size_t len = ((size_t)buf[pos + 1] << 16) | ((size_t)buf[pos + 2] << 8) | buf[pos + 3];
    pos += 4;
    if (pos + len > n) return -1;
    if (type == 0) {
      if (len < 34) return -1;
      const uint8_t* s = buf + pos;
      si->sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      si->channels = ((s[12] >> 1) & 7) + 1;
      si->bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      si->total = ((int64_t)(s[13] & 0xf) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
      have = true;
    }
    pos += len;
    if (last) break;
  }
  if (!have) return -1;
  si->audio_off = pos;
  return 0;
}

// residual of one subframe into out[order .. blocksize)
bool read_residual(BitReader& br, int32_t* out, int blocksize, int order) {
  const int method = br.read(2);
  if (method > 1) return false;
  const int pbits = method == 0 ? 4 : 5;
  const uint32_t esc = method == 0 ? 15 : 31;
  const int porder = br.read(4);
  const int parts = 1 << porder;
  if ((blocksize >> porder) << porder != blocksize && porder > 0) return false;
  int idx = order;
  for (int pi = 0; pi < parts; ++pi) {
    int cnt = (blocksize >> porder) - (pi == 0 ? order : 0);
    if (porder == 0) cnt = blocksize - order;
    if (cnt < 0) return false;
    const uint32_t k = br.read(pbits);
    if (k == esc) {
      const int nb = br.read(5);
      for (int i = 0; i < cnt; ++i) out[idx++] = br.read_signed(nb);
    } else {
      for (int i = 0; i < cnt; ++i) {
        const uint32_t q = br.read_unary();
        const uint32_t u = (q << k) | br.read((int)k);
        out[idx++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
      }
    }
    if (br.bad) return false;
  }
  return idx == blocksize;
}

bool read_subframe(BitReader& br, int32_t* out, int blocksize, int bps) {
  if (br.read(1) != 0) return false;
  const int type = br.read(6);
  int wasted = 0;
  if (br.read(1)) wasted = (int)br.read_unary() + 1;
  bps -= wasted;
  if (bps < 1 || bps > 33) return false;
  auto rd = [&](int bits) -> int32_t {  // side channels of 32-bit streams need 33 bits: not supported (never in speech corpora)
    return br.read_signed(bits > 32 ? 32 : bits);
  };
  if (type == 0) {
    const int32_t v = rd(bps);
    for (int i = 0; i < blocksize; ++i) out[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < blocksize; ++i) out[i] = rd(bps);
  } else if (type >= 8 && type <= 12) {
    const int order = type - 8;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) out[i] = rd(bps);
    if (!read_residual(br, out, blocksize, order)) return false;
    switch (order) {  // int64 arithmetic: 24-bit material can overflow 32 bits at order 4
      case 0: break;
      case 1: for (int i = 1; i < blocksize; ++i) out[i] = (int32_t)((int64_t)out[i] + out[i - 1]); break;
      case 2: for (int i = 2; i < blocksize; ++i) out[i] = (int32_t)((int64_t)out[i] + 2 * (int64_t)out[i - 1] - out[i - 2]); break;
      case 3: for (int i = 3; i < blocksize; ++i) out[i] = (int32_t)((int64_t)out[i] + 3 * (int64_t)out[i - 1] - 3 * (int64_t)out[i - 2] + out[i - 3]); break;
      case 4: for (int i = 4; i < blocksize; ++i) out[i] = (int32_t)((int64_t)out[i] + 4 * (int64_t)out[i - 1] - 6 * (int64_t)out[i - 2] + 4 * (int64_t)out[i - 3] - out[i - 4]); break;
    }
  } else if (type >= 32) {
    const int order = (type & 31) + 1;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) out[i] = rd(bps);
    const int prec = (int)br.read(4) + 1;
    if (prec == 16) return false;
    const int shift = br.read_signed(5);
    if (shift < 0) return false;
    int32_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = br.read_signed(prec);
    if (!read_residual(br, out, blocksize, order)) return false;
    for (int i = order; i < blocksize; ++i) {
      int64_t acc = 0;
      for (int j = 0; j < order; ++j) acc += (int64_t)coef[j] * out[i - 1 - j];
      out[i] = (int32_t)((int64_t)out[i] + (acc >> shift));
    }
  } else {
    return false;  // reserved subframe type
  }
  if (wasted)
    for (int i = 0; i < blocksize; ++i) out[i] = (int32_t)((uint32_t)out[i] << wasted);
  return !br.bad;
}

}  // namespace

extern "C" {

int wca_flac_info(const uint8_t* buf, int64_t nbytes, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample,
                  int64_t* total_samples) {
  StreamInfo si;
  if (!buf || nbytes < 0 || parse_header(buf, (size_t)nbytes, &si) != 0) return WCA_ERR_INVALID;
  if (sample_rate) *sample_rate = si.sample_rate;
  if (channels) *channels = si.channels;
  if (bits_per_sample) *bits_per_sample = si.bps;
  if (total_samples) *total_samples = si.total;
  return WCA_OK;
}

int wca_flac_decode(const uint8_t* buf, int64_t nbytes, float* out, int64_t capacity_per_channel, int64_t* n_decoded) {
  StreamInfo si;
  if (!buf || !n_decoded || nbytes < 0 || parse_header(buf, (size_t)nbytes, &si) != 0) return WCA_ERR_INVALID;
  const size_t n = (size_t)nbytes;
  size_t pos = si.audio_off;
  int64_t done = 0;
  std::vector<int32_t> ch[8];
  const float scale = 1.0f / (float)(1ull << (si.bps - 1));
  while (pos + 6 <= n) {
    if (!(buf[pos] == 0xff && (buf[pos + 1] & 0xfe) == 0xf8)) return WCA_ERR_INVALID;  // frame sync lost
    BitReader br(buf + pos, n - pos);
    br.read(15);
    br.read(1);  // blocking strategy (the coded number is a frame or a sample index; neither is needed to decode in order)
    const int bs_code = br.read(4), sr_code = br.read(4), ch_code = br.read(4), ss_code = br.read(3);
    if (br.read(1) != 0) return WCA_ERR_INVALID;
    {  // UTF-8 style coded frame / sample number
      const uint32_t b0 = br.read(8);
      int extra = 0;
      if (b0 >= 0xfe) extra = 6;
      else if (b0 >= 0xfc) extra = 5;
      else if (b0 >= 0xf8) extra = 4;
      else if (b0 >= 0xf0) extra = 3;
      else if (b0 >= 0xe0) extra = 2;
      else if (b0 >= 0xc0) extra = 1;
      else if (b0 >= 0x80) return WCA_ERR_INVALID;
      for (int i = 0; i < extra; ++i)
        if ((br.read(8) & 0xc0) != 0x80) return WCA_ERR_INVALID;
    }
    int blocksize;
    if (bs_code == 0) return WCA_ERR_INVALID;
    else if (bs_code == 1) blocksize = 192;
    else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
    else if (bs_code == 6) blocksize = (int)br.read(8) + 1;
    else if (bs_code == 7) blocksize = (int)br.read(16) + 1;
    else blocksize = 256 << (bs_code - 8);
    if (sr_code == 12) br.read(8);
    else if (sr_code == 13 || sr_code == 14) br.read(16);
    else if (sr_code == 15) return WCA_ERR_INVALID;
    static const int ss_table[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    int bps = ss_table[ss_code];
    if (bps < 0) return WCA_ERR_INVALID;
    if (bps == 0) bps = si.bps;
    if (bps != si.bps) return WCA_ERR_INVALID;  // a mid-stream change of sample size is not supported
    const size_t hdr_len = br.byte_pos();
    const uint8_t want8 = (uint8_t)br.read(8);
    if (br.bad || WCA_FLAC_CRC_MISMATCH(crc8(buf + pos, hdr_len), want8)) return WCA_ERR_INVALID;
    int nch;
    if (ch_code < 8) nch = ch_code + 1;
    else if (ch_code <= 10) nch = 2;
    else return WCA_ERR_INVALID;
    if (nch != si.channels) return WCA_ERR_INVALID;
    for (int c = 0; c < nch; ++c) {
      ch[c].resize((size_t)blocksize);
      int b = bps;
      if ((ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1)) b += 1;  // the side channel
      if (!read_subframe(br, ch[c].data(), blocksize, b)) return WCA_ERR_INVALID;
    }
    br.align_byte();
    const size_t body_len = br.byte_pos();
    const uint16_t want16 = (uint16_t)br.read(16);
    if (br.bad || WCA_FLAC_CRC_MISMATCH(crc16(buf + pos, body_len), want16)) return WCA_ERR_INVALID;
    // inter-channel decorrelation in wrapping / 64-bit arithmetic: a crafted (CRC-valid) stream can carry side values that overflow int32
    // (found by the sanitizer fuzz build, tests/test_flac_fuzz.py; valid streams never get there)
    if (ch_code == 8) {
      for (int i = 0; i < blocksize; ++i) ch[1][i] = (int32_t)((uint32_t)ch[0][i] - (uint32_t)ch[1][i]);
    } else if (ch_code == 9) {
      for (int i = 0; i < blocksize; ++i) ch[0][i] = (int32_t)((uint32_t)ch[0][i] + (uint32_t)ch[1][i]);
    } else if (ch_code == 10) {
      for (int i = 0; i < blocksize; ++i) {
        const int64_t side = ch[1][i];
        const int64_t mid = (int64_t)(((uint64_t)(int64_t)ch[0][i] << 1) | ((uint64_t)side & 1));
        ch[0][i] = (int32_t)((mid + side) >> 1);
        ch[1][i] = (int32_t)((mid - side) >> 1);
      }
    }
    int64_t take = blocksize;
    if (si.total > 0 && done + take > si.total) take = si.total - done;
    if (out && done + take > capacity_per_channel) return WCA_ERR_NOMEM;
    for (int c = 0; out && c < nch; ++c) {
      float* o = out + (size_t)c * (size_t)capacity_per_channel + (size_t)done;
      const int32_t* s = ch[c].data();
      for (int64_t i = 0; i < take; ++i) o[i] = (float)s[i] * scale;
    }
    done += take;
    pos += body_len + 2;
    if (si.total > 0 && done >= si.total) break;
  }
  *n_decoded = done;
  if (si.total > 0 && done != si.total) return WCA_ERR_INVALID;
  return WCA_OK;
}

}  // extern "C"
