// pt/image_io.hpp — the image files of the reference's host, without its stb dependency: decode PNG / baseline JPEG / binary PPM to
// RGB8 (what image_texture_factory needs: stbi_load(file, &w, &h, &n, 3), texture.hpp:97-117) and write an RGB8 PNG (save_image_png,
// main.cpp:33-59: stbi_write_png).  Header-only C++20, no library: inflate (RFC 1951), the PNG filters (RFC 2083) and a baseline
// JPEG decoder (ITU T.81: Huffman, sequential DCT, restart intervals, 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0) are written out here.
//
// Which decoder's PIXELS?  A JPEG does not define its decoded pixels bit for bit: the inverse DCT, the chroma up-sampling and the
// colour conversion are the decoder's choice.  This one follows the choices of the IJG library's default path — the accurate integer
// inverse DCT (13-bit constants, two passes with 2 guard bits), triangle-filter ("fancy") up-sampling of sub-sampled chroma, 16-bit
// fixed-point YCbCr -> RGB — because that is what the Python host's loader (PIL, built on libjpeg-turbo) produces, so both hosts of
// this repository put the SAME texels into the atlas: tests/test_image_io_cpu.py compares every decoded pixel with PIL's on the
// reference's own images/Xilinx.jpg and images/SYCL.png and on generated files (sub-sampled chroma, restart intervals, odd sizes,
// grey, palette, alpha, every PNG filter type).  The reference's stb_image makes other (equally legal) choices in those three
// places; its texels differ from libjpeg's by at most a few grey levels on a JPEG and not at all on a PNG.
//
// Not decoded (failure reason returned, the caller falls back like the reference does on a load failure): progressive / arithmetic /
// lossless / 12-bit JPEG, CMYK, interlaced PNG.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <new>
#include <stdexcept>
#include <vector>

namespace pt::image_io {

struct Image {
  std::vector<uint8_t> rgb; // height * width * 3, rows top-down
  std::size_t width = 0, height = 0;
};

namespace detail {

inline bool read_file(const char* path, std::vector<uint8_t>& out) {
  std::FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  uint8_t buf[1 << 16];
  std::size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
  std::fclose(f);
  return true;
}

// ---- inflate (RFC 1951) -----------------------------------------------------------------------------------------------------
struct BitsLSB {
  const uint8_t* p; std::size_t n, at = 0; uint32_t acc = 0; int cnt = 0; bool bad = false;
  uint32_t get(int k) {
    while (cnt < k) { if (at >= n) { bad = true; return 0; } acc |= (uint32_t)p[at++] << cnt; cnt += 8; }
    const uint32_t v = acc & ((k == 32) ? 0xffffffffu : ((1u << k) - 1u));
    acc = (k == 32) ? 0 : acc >> k; cnt -= k;
    return v;
  }
};
struct Canon { // canonical Huffman code: symbols ordered by (length, symbol)
  uint16_t count[16] = {0}, symbol[288] = {0};
  bool build(const uint8_t* len, int n) {
    for (int i = 0; i < 16; i++) count[i] = 0;
    for (int i = 0; i < n; i++) count[len[i]]++;
    int left = 1;
    for (int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; } // over-subscribed
    uint16_t offs[16]; offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    return true;
  }
  int decode(BitsLSB& b) const {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
      code |= (int)b.get(1);
      if (b.bad) return -1;
      const int c = count[l];
      if (code - c < first) return symbol[index + (code - first)];
      index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
  }
};
// max_out: the largest output the caller can use (a PNG's (stride + 1) * height): a stream that expands past it is rejected as soon as it
// does, so a small hostile file cannot make the decoder allocate without bound
inline const char* inflate(const uint8_t* src, std::size_t n, std::vector<uint8_t>& out, std::size_t max_out = (std::size_t)1 << 32) {
  static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  BitsLSB b{src, n};
  for (;;) {
    const int last = (int)b.get(1), type = (int)b.get(2);
    if (b.bad) return "truncated deflate stream";
    if (type == 0) {
      b.acc = 0; b.cnt = 0; // to the byte boundary
      if (b.at + 4 > n) return "truncated stored block";
      const unsigned len = src[b.at] | (src[b.at + 1] << 8), nlen = src[b.at + 2] | (src[b.at + 3] << 8);
      if ((len ^ 0xffffu) != nlen || b.at + 4 + len > n) return "corrupt stored block";
      if (len > max_out - out.size()) return "deflate stream larger than the image it belongs to";
      out.insert(out.end(), src + b.at + 4, src + b.at + 4 + len);
      b.at += 4 + len;
    } else if (type == 1 || type == 2) {
      Canon lit, dist;
      uint8_t len[320];
      if (type == 1) {
        for (int i = 0; i < 144; i++) len[i] = 8;
        for (int i = 144; i < 256; i++) len[i] = 9;
        for (int i = 256; i < 280; i++) len[i] = 7;
        for (int i = 280; i < 288; i++) len[i] = 8;
        lit.build(len, 288);
        for (int i = 0; i < 30; i++) len[i] = 5;
        dist.build(len, 30);
      } else {
        const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
        if (b.bad || nlen > 286 || ndist > 30) return "bad dynamic block header";
        uint8_t cl[19] = {0};
        for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)b.get(3);
        Canon cc;
        if (!cc.build(cl, 19)) return "bad code-length code";
        int i = 0;
        while (i < nlen + ndist) {
          const int sym = cc.decode(b);
          if (sym < 0) return "bad code lengths";
          if (sym < 16) len[i++] = (uint8_t)sym;
          else {
            int prev = 0, rep;
            if (sym == 16) { if (i == 0) return "repeat without a length"; prev = len[i - 1]; rep = 3 + (int)b.get(2); }
            else if (sym == 17) rep = 3 + (int)b.get(3);
            else rep = 11 + (int)b.get(7);
            if (i + rep > nlen + ndist) return "too many code lengths";
            while (rep--) len[i++] = (uint8_t)prev;
          }
        }
        if (b.bad || len[256] == 0) return "bad dynamic block";
        if (!lit.build(len, nlen) || !dist.build(len + nlen, ndist)) return "over-subscribed code";
      }
      for (;;) {
        int sym = lit.decode(b);
        if (sym < 0) return "bad literal/length code";
        if (sym < 256) { if (out.size() >= max_out) return "deflate stream larger than the image it belongs to"; out.push_back((uint8_t)sym); }
        else if (sym == 256) break;
        else {
          sym -= 257;
          if (sym >= 29) return "bad length symbol";
          const int l = lbase[sym] + (int)b.get(lext[sym]);
          const int ds = dist.decode(b);
          if (ds < 0 || ds >= 30) return "bad distance code";
          const std::size_t d = dbase[ds] + b.get(dext[ds]);
          if (b.bad || d > out.size()) return "distance too far back";
          if ((std::size_t)l > max_out - out.size()) return "deflate stream larger than the image it belongs to";
          const std::size_t from = out.size() - d;
          for (int k = 0; k < l; k++) out.push_back(out[from + (std::size_t)k]);
        }
      }
    } else return "bad block type";
    if (last) return nullptr;
  }
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t crc32(const uint8_t* p, std::size_t n, uint32_t crc = 0) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1; table[i] = c; } ready = true; }
  crc = ~crc;
  for (std::size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 255] ^ (crc >> 8);
  return ~crc;
}
inline uint32_t adler32(const uint8_t* p, std::size_t n) {
  uint32_t a = 1, b = 0;
  for (std::size_t i = 0; i < n; i++) { a = (a + p[i]) % 65521u; b = (b + a) % 65521u; }
  return (b << 16) | a;
}

// ---- PNG (RFC 2083): 8/16-bit grey, RGB, palette, with or without alpha; non-interlaced ---------------------------------------------
inline const char* decode_png(const std::vector<uint8_t>& f, Image& im) {
  if (f.size() < 8 + 25) return "truncated PNG";
  std::size_t at = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, palette;
  bool seen_ihdr = false, seen_end = false;
  while (at + 12 <= f.size()) {
    const uint32_t len = be32(&f[at]);
    if (len > f.size() - at - 12) return "truncated PNG chunk";
    const uint8_t* tag = &f[at + 4];
    const uint8_t* data = &f[at + 8];
    if (crc32(tag, 4 + (std::size_t)len) != be32(data + len)) return "bad PNG chunk CRC";
    if (!std::memcmp(tag, "IHDR", 4)) {
      if (len != 13) return "bad IHDR";
      w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
      if (data[10] != 0 || data[11] != 0) return "unknown PNG compression / filter method";
      seen_ihdr = true;
    } else if (!std::memcmp(tag, "PLTE", 4)) palette.assign(data, data + len);
    else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!std::memcmp(tag, "IEND", 4)) { seen_end = true; break; }
    at += 12 + (std::size_t)len;
  }
  if (!seen_ihdr || !seen_end || idat.size() < 6) return "incomplete PNG";
  if (w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24) || (uint64_t)w * h > (1ull << 28)) return "PNG too large";
  if (interlace) return "interlaced PNG not supported";
  const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!channels) return "bad PNG colour type";
  const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                        (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                        ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
  if (!depth_ok) return "bad PNG bit depth";
  if (ctype == 3 && palette.size() < 3) return "palette PNG without a palette";
  if ((idat[0] & 15) != 8 || ((idat[0] << 8) | idat[1]) % 31 != 0 || (idat[1] & 32)) return "bad zlib header";
  const std::size_t stride = ((std::size_t)w * channels * depth + 7) / 8, bpp = (std::size_t)(channels * depth + 7) / 8;
  std::vector<uint8_t> raw;
  if (const char* e = inflate(idat.data() + 2, idat.size() - 6, raw, (stride + 1) * (std::size_t)h)) return e;
  if (raw.size() < (stride + 1) * h) return "PNG pixel data too short";
  if (adler32(raw.data(), raw.size()) != be32(&idat[idat.size() - 4])) return "bad zlib checksum";
  std::vector<uint8_t> prev(stride, 0);
  im.width = w; im.height = h;
  im.rgb.resize((std::size_t)w * h * 3);
  for (uint32_t y = 0; y < h; y++) {
    uint8_t* row = &raw[(stride + 1) * y + 1];
    const int ft = row[-1];
    if (ft > 4) return "bad PNG filter type";
    for (std::size_t i = 0; i < stride; i++) {
      const int a = i >= bpp ? row[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int pred = 0;
      if (ft == 1) pred = a;
      else if (ft == 2) pred = b;
      else if (ft == 3) pred = (a + b) >> 1;
      else if (ft == 4) { const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p; pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
      row[i] = (uint8_t)(row[i] + pred);
    }
    std::memcpy(prev.data(), row, stride);
    uint8_t* out = &im.rgb[(std::size_t)y * w * 3];
    for (uint32_t x = 0; x < w; x++) {
      auto sample = [&](std::size_t k) -> int { // k-th sample of the row, reduced to 8 bits
        if (depth == 8) return row[k];
        if (depth == 16) return row[2 * k]; // the high byte
        const int per = 8 / depth, v = (row[k / per] >> ((per - 1 - (int)(k % per)) * depth)) & ((1 << depth) - 1);
        return ctype == 3 ? v : v * (255 / ((1 << depth) - 1));
      };
      if (ctype == 3) {
        const std::size_t idx = (std::size_t)sample(x) * 3;
        if (idx + 3 > palette.size()) { out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = 0; } // an index past the palette: black
        else { out[3 * x] = palette[idx]; out[3 * x + 1] = palette[idx + 1]; out[3 * x + 2] = palette[idx + 2]; }
      } else if (channels <= 2) { const uint8_t g = (uint8_t)sample((std::size_t)x * channels); out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = g; }
      else for (int k = 0; k < 3; k++) out[3 * x + k] = (uint8_t)sample((std::size_t)x * channels + k);
    }
  }
  return nullptr;
}

// ---- baseline JPEG (ITU T.81) ----------------------------------------------------------------------------------------------------
struct JHuff { // code lengths 1..16 -> (min code, max code, first value index); decoded bit by bit
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  bool present = false;
};
struct JComp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; int bw = 0, bh = 0; /* blocks across / down, padded to the MCU */ std::vector<uint8_t> plane; int pw = 0, ph = 0; };
struct JBits {
  const uint8_t* p; std::size_t n, at; uint32_t acc = 0; int cnt = 0; int marker = 0;
  int bit() {
    if (cnt == 0) {
      int c = 0;
      if (marker == 0 && at < n) {
        c = p[at++];
        if (c == 0xff) {
          int c2 = at < n ? p[at] : 0xd9;
          while (c2 == 0xff && at + 1 < n) c2 = p[++at]; // fill bytes
          if (c2 == 0) at++;                              // stuffed zero
          else { marker = c2; at++; c = 0; }             // a marker ends the entropy-coded segment: feed zeros
        }
      }
      acc = (uint32_t)c; cnt = 8;
    }
    cnt--;
    return (int)(acc >> cnt) & 1;
  }
  int receive(int s) { int v = 0; while (s--) v = (v << 1) | bit(); return v; }
  void reset() { cnt = 0; marker = 0; }
};
inline int jdecode(JBits& b, const JHuff& t) {
  int code = b.bit();
  for (int l = 1; l <= 16; l++) {
    if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) return t.vals[t.valptr[l] + code - t.mincode[l]];
    code = (code << 1) | b.bit();
  }
  return -1;
}
inline int jextend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
// a dequantised coefficient, in 64 bits, limited to +-2^20 (a valid 8-bit stream stays below 2^16; 16-bit quantisation tables times 15-bit
// values of a crafted one would overflow int — and the inverse DCT's 32-bit workspace — otherwise)
inline int jclampc(long long v) { return (int)(v < -(1ll << 20) ? -(1ll << 20) : v > (1ll << 20) ? (1ll << 20) : v); }
inline uint8_t jrange(int v) { // the post-IDCT range limit of the IJG code: a table indexed with the low 10 bits of (value), centred on 128
  const int i = v & 1023;
  return (uint8_t)(i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896);
}
// the accurate integer inverse DCT (13-bit constants, PASS1 keeps 2 extra bits): coefficients in natural order -> 8x8 samples
inline void jidct(const int* in, uint8_t* out, int stride) {
  constexpr int CB = 13, P1 = 2;
  constexpr int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  auto descale = [](long x, int n) { return (int)((x + (1L << (n - 1))) >> n); };
  int ws[64];
  for (int c = 0; c < 8; c++) {
    const int* p = in + c;
    int* w = ws + c;
    if (!(p[8] | p[16] | p[24] | p[32] | p[40] | p[48] | p[56])) { const int dc = p[0] * (1 << P1); for (int r = 0; r < 8; r++) w[8 * r] = dc; continue; }
    long z2 = p[16], z3 = p[48];
    long z1 = (z2 + z3) * F0541;
    long tmp2 = z1 + z3 * -F1847, tmp3 = z1 + z2 * F0765;
    z2 = p[0]; z3 = p[32];
    long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = p[56]; tmp1 = p[40]; tmp2 = p[24]; tmp3 = p[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    w[0] = descale(tmp10 + tmp3, CB - P1); w[56] = descale(tmp10 - tmp3, CB - P1);
    w[8] = descale(tmp11 + tmp2, CB - P1); w[48] = descale(tmp11 - tmp2, CB - P1);
    w[16] = descale(tmp12 + tmp1, CB - P1); w[40] = descale(tmp12 - tmp1, CB - P1);
    w[24] = descale(tmp13 + tmp0, CB - P1); w[32] = descale(tmp13 - tmp0, CB - P1);
  }
  for (int r = 0; r < 8; r++) {
    const int* w = ws + 8 * r;
    uint8_t* o = out + (std::size_t)r * stride;
    if (!(w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7])) { const uint8_t dc = jrange(descale(w[0], P1 + 3)); for (int c = 0; c < 8; c++) o[c] = dc; continue; }
    long z2 = w[2], z3 = w[6];
    long z1 = (z2 + z3) * F0541;
    long tmp2 = z1 + z3 * -F1847, tmp3 = z1 + z2 * F0765;
    long tmp0 = ((long)w[0] + w[4]) * (1L << CB), tmp1 = ((long)w[0] - w[4]) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    constexpr int S = CB + P1 + 3;
    o[0] = jrange(descale(tmp10 + tmp3, S)); o[7] = jrange(descale(tmp10 - tmp3, S));
    o[1] = jrange(descale(tmp11 + tmp2, S)); o[6] = jrange(descale(tmp11 - tmp2, S));
    o[2] = jrange(descale(tmp12 + tmp1, S)); o[5] = jrange(descale(tmp12 - tmp1, S));
    o[3] = jrange(descale(tmp13 + tmp0, S)); o[4] = jrange(descale(tmp13 - tmp0, S));
  }
}

inline const char* decode_jpeg(const std::vector<uint8_t>& f, Image& im) {
  static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  int qt[4][64]; bool qt_ok[4] = {false, false, false, false};
  JHuff dc[4], ac[4];
  std::vector<JComp> comp;
  int W = 0, H = 0, hmax = 1, vmax = 1, restart = 0, adobe_transform = -1;
  bool sof = false, jfif = false;
  std::size_t at = 2;
  if (f.size() < 4 || f[0] != 0xff || f[1] != 0xd8) return "not a JPEG";
  for (;;) {
    while (at < f.size() && f[at] != 0xff) at++;           // (garbage between segments is skipped, as the IJG code does with a warning)
    while (at < f.size() && f[at] == 0xff) at++;
    if (at >= f.size()) return "JPEG ends before its image data is complete";
    const int m = f[at++];
    if (m == 0xd8 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
    if (m == 0xd9) break;
    if (at + 2 > f.size()) return "truncated JPEG";
    const std::size_t len = ((std::size_t)f[at] << 8) | f[at + 1];
    if (len < 2 || at + len > f.size()) return "truncated JPEG segment";
    const uint8_t* s = &f[at + 2];
    const std::size_t n = len - 2;
    if (m == 0xc0 || m == 0xc1) {
      if (n < 6 || s[0] != 8) return "only 8-bit JPEG is decoded";
      H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4];
      const int nc = s[5];
      if (W == 0 || H == 0 || (nc != 1 && nc != 3) || n < 6 + 3 * (std::size_t)nc) return nc == 4 ? "CMYK JPEG not supported" : "bad JPEG frame header";
      if ((uint64_t)W * (uint64_t)H > (1ull << 28)) return "JPEG too large"; // (the PNG path's limit; checked BEFORE anything is sized by the header)
      if (sof) return "JPEG with more than one frame header";
      comp.resize((std::size_t)nc);
      for (int i = 0; i < nc; i++) {
        comp[(std::size_t)i].id = s[6 + 3 * i]; comp[(std::size_t)i].h = s[7 + 3 * i] >> 4; comp[(std::size_t)i].v = s[7 + 3 * i] & 15; comp[(std::size_t)i].tq = s[8 + 3 * i] & 3;
        if (comp[(std::size_t)i].h < 1 || comp[(std::size_t)i].h > 2 || comp[(std::size_t)i].v < 1 || comp[(std::size_t)i].v > 2) return "JPEG sampling factors other than 1 and 2 are not supported";
        hmax = comp[(std::size_t)i].h > hmax ? comp[(std::size_t)i].h : hmax; vmax = comp[(std::size_t)i].v > vmax ? comp[(std::size_t)i].v : vmax;
      }
      if (nc == 3 && (comp[0].h != hmax || comp[0].v != vmax || comp[1].h != comp[2].h || comp[1].v != comp[2].v)) return "unusual JPEG sampling layout not supported";
      if (nc == 1) { comp[0].h = comp[0].v = hmax = vmax = 1; } // a single-component scan is never interleaved: its sampling factors mean nothing
      const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
      for (auto& c : comp) { c.bw = mcux * c.h; c.bh = mcuy * c.v; c.pw = c.bw * 8; c.ph = c.bh * 8; c.plane.assign((std::size_t)c.pw * c.ph, 0); }
      sof = true;
    } else if (m == 0xc2) return "progressive JPEG not supported";
    else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) return "this JPEG process (lossless / hierarchical / arithmetic) is not supported";
    else if (m == 0xcc) return "arithmetic-coded JPEG not supported";
    else if (m == 0xdb) {
      std::size_t i = 0;
      while (i < n) {
        const int pq = s[i] >> 4, tq = s[i] & 15;
        if (tq > 3 || i + 1 + (pq ? 128 : 64) > n) return "bad quantisation table";
        for (int k = 0; k < 64; k++) qt[tq][zigzag[k]] = pq ? ((s[i + 1 + 2 * k] << 8) | s[i + 2 + 2 * k]) : s[i + 1 + k];
        qt_ok[tq] = true;
        i += 1 + (pq ? 128 : 64);
      }
    } else if (m == 0xc4) {
      std::size_t i = 0;
      while (i + 17 <= n) {
        const int tc = s[i] >> 4, th = s[i] & 15;
        if (tc > 1 || th > 3) return "bad Huffman table id";
        JHuff& t = tc ? ac[th] : dc[th];
        int total = 0;
        for (int l = 1; l <= 16; l++) total += s[i + (std::size_t)l];
        if (total > 256 || i + 17 + (std::size_t)total > n) return "bad Huffman table";
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
          const int cnt = s[i + (std::size_t)l];
          t.valptr[l] = k; t.mincode[l] = code;
          t.maxcode[l] = cnt ? code + cnt - 1 : -1;
          code = (code + cnt) << 1; k += cnt;
        }
        std::memcpy(t.vals, s + i + 17, (std::size_t)total);
        t.present = true;
        i += 17 + (std::size_t)total;
      }
    } else if (m == 0xdd) { if (n >= 2) restart = (s[0] << 8) | s[1]; }
    else if (m == 0xe0) { if (n >= 5 && !std::memcmp(s, "JFIF", 5)) jfif = true; }
    else if (m == 0xee) { if (n >= 12 && !std::memcmp(s, "Adobe", 5)) adobe_transform = s[11]; }
    else if (m == 0xda) {
      if (!sof || n < 1) return "scan before the frame header";
      const int ns = s[0];
      if (ns < 1 || ns > (int)comp.size() || n < 1 + 2 * (std::size_t)ns + 3) return "bad scan header";
      std::vector<JComp*> sc;
      for (int i = 0; i < ns; i++) {
        JComp* c = nullptr;
        for (auto& k : comp) if (k.id == s[1 + 2 * i]) c = &k;
        if (!c) return "scan names an unknown component";
        c->td = s[2 + 2 * i] >> 4; c->ta = s[2 + 2 * i] & 15;
        if (c->td > 3 || c->ta > 3 || !dc[c->td].present || !ac[c->ta].present || !qt_ok[c->tq]) return "scan uses a table that was not defined";
        sc.push_back(c);
      }
      if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63) return "not a sequential scan";
      JBits b{f.data(), f.size(), at + len};
      for (auto* c : sc) c->pred = 0;
      const bool inter = ns > 1;
      // interleaved: MCUs of h x v blocks per component over the padded frame; a single component: its own blocks, unpadded
      const int mx = inter ? comp[0].bw / comp[0].h : (((W * sc[0]->h + hmax - 1) / hmax) + 7) / 8;
      const int my = inter ? comp[0].bh / comp[0].v : (((H * sc[0]->v + vmax - 1) / vmax) + 7) / 8;
      int todo = restart;
      int coef[64];
      for (int yy = 0; yy < my; yy++)
        for (int xx = 0; xx < mx; xx++) {
          if (restart && todo == 0) { // RSTn expected here
            b.cnt = 0;
            if (b.marker == 0) { // the marker has not been met yet: find it
              while (b.at + 1 < b.n && !(b.p[b.at] == 0xff && b.p[b.at + 1] >= 0xd0 && b.p[b.at + 1] <= 0xd7)) b.at++;
              if (b.at + 1 < b.n) b.at += 2;
            }
            b.reset();
            for (auto* c : sc) c->pred = 0;
            todo = restart;
          }
          for (auto* c : sc)
            for (int by = 0; by < (inter ? c->v : 1); by++)
              for (int bx = 0; bx < (inter ? c->h : 1); bx++) {
                std::memset(coef, 0, sizeof coef);
                const int t = jdecode(b, dc[c->td]);
                if (t < 0 || t > 15) return "bad DC code";
                c->pred += jextend(b.receive(t), t); // |difference| < 2^15 and |pred| <= 2^15 before: no overflow
                if (c->pred < -32768 || c->pred > 32767) return "DC coefficient out of range";
                coef[0] = jclampc((long long)c->pred * qt[c->tq][0]);
                for (int k = 1; k < 64;) {
                  const int rs = jdecode(b, ac[c->ta]);
                  if (rs < 0) return "bad AC code";
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                  k += r;
                  if (k > 63) return "AC run past the block";
                  coef[zigzag[k]] = jclampc((long long)jextend(b.receive(sz), sz) * qt[c->tq][zigzag[k]]);
                  k++;
                }
                const int px = (inter ? xx * c->h + bx : xx) * 8, py = (inter ? yy * c->v + by : yy) * 8;
                if (px + 8 <= c->pw && py + 8 <= c->ph) jidct(coef, &c->plane[(std::size_t)py * c->pw + px], c->pw);
              }
          if (restart) todo--;
        }
      // continue at the marker that ended the scan (or, if the decoder stopped short of it, at the next real marker: 0xff followed by
      // anything but a stuffed zero, a fill byte or a restart marker)
      at = b.marker ? b.at - 2 : b.at;
      while (at + 1 < f.size() && !(f[at] == 0xff && f[at + 1] != 0 && f[at + 1] != 0xff && !(f[at + 1] >= 0xd0 && f[at + 1] <= 0xd7))) at++;
      continue;
    }
    at += len;
  }
  if (!sof) return "JPEG without a frame";
  im.width = (std::size_t)W; im.height = (std::size_t)H;
  im.rgb.resize((std::size_t)W * H * 3);
  // chroma to full resolution: the triangle filter ("fancy up-sampling") of the IJG code for 2:1 horizontally and / or vertically
  auto upsample = [&](const JComp& c, std::vector<uint8_t>& full) {
    const int fw = c.pw * (hmax / c.h), fh = c.ph * (vmax / c.v);
    full.assign((std::size_t)fw * fh, 0);
    const int dw = (W * c.h + hmax - 1) / hmax, dh = (H * c.v + vmax - 1) / vmax; // the component's own size (samples that carry image)
    const bool h2 = hmax / c.h == 2, v2 = vmax / c.v == 2;
    auto in = [&](int x, int y) { return (int)c.plane[(std::size_t)y * c.pw + x]; };
    if (!h2 && !v2) { for (int y = 0; y < c.ph; y++) std::memcpy(&full[(std::size_t)y * fw], &c.plane[(std::size_t)y * c.pw], (std::size_t)c.pw); return fw; }
    for (int oy = 0; oy < (v2 ? 2 * dh : dh); oy++) {
      const int y0 = v2 ? oy / 2 : oy;
      int y1 = v2 ? ((oy & 1) ? y0 + 1 : y0 - 1) : y0; // the farther of the two rows that blend into this one
      y1 = y1 < 0 ? 0 : (y1 >= dh ? dh - 1 : y1);      // edge rows blend with themselves
      uint8_t* o = &full[(std::size_t)oy * fw];
      if (h2 && !v2) {
        if (dw == 1) { o[0] = o[1] = (uint8_t)in(0, y0); continue; }
        o[0] = (uint8_t)in(0, y0); o[1] = (uint8_t)((in(0, y0) * 3 + in(1, y0) + 2) >> 2);
        for (int x = 1; x < dw - 1; x++) { const int v = in(x, y0) * 3; o[2 * x] = (uint8_t)((v + in(x - 1, y0) + 1) >> 2); o[2 * x + 1] = (uint8_t)((v + in(x + 1, y0) + 2) >> 2); }
        o[2 * dw - 2] = (uint8_t)((in(dw - 1, y0) * 3 + in(dw - 2, y0) + 1) >> 2); o[2 * dw - 1] = (uint8_t)in(dw - 1, y0);
      } else if (h2 && v2) {
        auto colsum = [&](int x) { return in(x, y0) * 3 + in(x, y1); };
        if (dw == 1) { const int s = colsum(0); o[0] = (uint8_t)((s * 4 + 8) >> 4); o[1] = (uint8_t)((s * 4 + 7) >> 4); continue; }
        int last, cur = colsum(0), next = colsum(1);
        o[0] = (uint8_t)((cur * 4 + 8) >> 4); o[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
        last = cur; cur = next;
        for (int x = 1; x < dw - 1; x++) { next = colsum(x + 1); o[2 * x] = (uint8_t)((cur * 3 + last + 8) >> 4); o[2 * x + 1] = (uint8_t)((cur * 3 + next + 7) >> 4); last = cur; cur = next; }
        o[2 * dw - 2] = (uint8_t)((cur * 3 + last + 8) >> 4); o[2 * dw - 1] = (uint8_t)((cur * 4 + 7) >> 4);
      } else { // v2 only: the vertical triangle filter, (3 near + far + 1 or 2) >> 2
        const int bias = (oy & 1) ? 2 : 1;
        for (int x = 0; x < dw; x++) o[x] = (uint8_t)((in(x, y0) * 3 + in(x, y1) + bias) >> 2);
      }
    }
    return fw;
  };
  if (comp.size() == 1) {
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) { const uint8_t g = comp[0].plane[(std::size_t)y * comp[0].pw + x]; uint8_t* o = &im.rgb[((std::size_t)y * W + x) * 3]; o[0] = o[1] = o[2] = g; }
    return nullptr;
  }
  std::vector<uint8_t> cbf, crf;
  const int cw = upsample(comp[1], cbf);
  upsample(comp[2], crf);
  // YCbCr unless an Adobe marker says the three components are RGB (transform 0); without JFIF / Adobe markers, ids 'R','G','B' mean RGB
  const bool ycc = adobe_transform >= 0 ? adobe_transform != 0 : (jfif || !(comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B'));
  auto clamp8 = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
  auto fix = [](double x) { return (long)(x * 65536.0 + 0.5); };
  const long f1402 = fix(1.40200), f1772 = fix(1.77200), f0714 = fix(0.71414), f0344 = fix(0.34414);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const int Y = comp[0].plane[(std::size_t)y * comp[0].pw + x], cb = cbf[(std::size_t)y * cw + x], cr = crf[(std::size_t)y * cw + x];
      uint8_t* o = &im.rgb[((std::size_t)y * W + x) * 3];
      if (!ycc) { o[0] = (uint8_t)Y; o[1] = (uint8_t)cb; o[2] = (uint8_t)cr; continue; }
      const long b = cb - 128, r = cr - 128; // 16-bit fixed point, rounded by adding a half before the (arithmetic) shift
      o[0] = clamp8(Y + (int)((f1402 * r + 32768) >> 16));
      o[1] = clamp8(Y + (int)((-f0344 * b + 32768 - f0714 * r) >> 16));
      o[2] = clamp8(Y + (int)((f1772 * b + 32768) >> 16));
    }
  return nullptr;
}

inline const char* decode_ppm(const std::vector<uint8_t>& f, Image& im) {
  std::size_t at = 2;
  auto token = [&](unsigned long& out) { // header integers, '#' comments allowed between them
    for (;;) {
      while (at < f.size() && (f[at] == ' ' || f[at] == '\t' || f[at] == '\n' || f[at] == '\r')) at++;
      if (at >= f.size() || f[at] != '#') break;
      while (at < f.size() && f[at] != '\n') at++;
    }
    if (at >= f.size() || f[at] < '0' || f[at] > '9') return false;
    out = 0;
    while (at < f.size() && f[at] >= '0' && f[at] <= '9') { out = out * 10 + (unsigned long)(f[at++] - '0'); if (out > (1ul << 30)) return false; }
    if (at >= f.size() || !(f[at] == ' ' || f[at] == '\t' || f[at] == '\n' || f[at] == '\r')) return false;
    at++; // exactly one whitespace byte ends a header field
    return true;
  };
  unsigned long uw = 0, uh = 0, maxval = 0;
  if (!token(uw) || !token(uh) || !token(maxval)) return "bad PPM header";
  if (uw == 0 || uh == 0 || maxval != 255) return "unsupported PPM (need width, height > 0 and maxval 255)";
  if (uw > (1ul << 30) / uh / 3) return "too large";
  if (f.size() - at < (std::size_t)uw * uh * 3) return "truncated PPM";
  im.width = uw; im.height = uh;
  im.rgb.assign(f.begin() + (std::ptrdiff_t)at, f.begin() + (std::ptrdiff_t)(at + (std::size_t)uw * uh * 3));
  return nullptr;
}

} // namespace detail

// Decode an image file to RGB8 (rows top-down).  Returns nullptr on success, else the failure reason — the role of
// stbi_failure_reason() (texture.hpp:108).
inline const char* load_rgb8(const char* path, Image& im) {
  if (!path) return "no file name";
  std::vector<uint8_t> f;
  if (!detail::read_file(path, f)) return "can't fopen";
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  try { // the documented contract is "a failure reason, the caller falls back" (texture.hpp:106-111): an allocation failure is one
    if (f.size() >= 8 && !std::memcmp(f.data(), png_sig, 8)) return detail::decode_png(f, im);
    if (f.size() >= 3 && f[0] == 0xff && f[1] == 0xd8 && f[2] == 0xff) return detail::decode_jpeg(f, im);
    if (f.size() >= 2 && f[0] == 'P' && f[1] == '6') return detail::decode_ppm(f, im);
  } catch (const std::bad_alloc&) { return "out of memory"; }
    catch (const std::length_error&) { return "out of memory"; }
  return "unknown image type (this host decodes PNG, baseline JPEG and binary PPM 'P6')";
}

// RGB8 -> PNG file (colour type 2, 8 bits, filter 0, stored deflate blocks: no compressor needed; any PNG reader decodes the exact
// bytes).  The output stage of main.cpp:33-59 ends in stbi_write_png.  Returns false if the file cannot be written.
inline bool write_png(const char* path, const uint8_t* rgb, std::size_t width, std::size_t height) {
  if (!path || !rgb || !width || !height) return false;
  std::vector<uint8_t> raw;
  raw.reserve(height * (width * 3 + 1));
  for (std::size_t y = 0; y < height; y++) { raw.push_back(0); raw.insert(raw.end(), rgb + y * width * 3, rgb + (y + 1) * width * 3); }
  std::vector<uint8_t> z = {0x78, 0x01};
  for (std::size_t at = 0; at < raw.size() || at == 0;) {
    const std::size_t n = raw.size() - at < 65535 ? raw.size() - at : 65535;
    z.push_back(at + n >= raw.size() ? 1 : 0);
    z.push_back((uint8_t)(n & 255)); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)(~n & 255)); z.push_back((uint8_t)((~n >> 8) & 255));
    z.insert(z.end(), raw.begin() + (std::ptrdiff_t)at, raw.begin() + (std::ptrdiff_t)(at + n));
    at += n;
    if (n == 0) break;
  }
  const uint32_t ad = detail::adler32(raw.data(), raw.size());
  for (int k = 3; k >= 0; k--) z.push_back((uint8_t)(ad >> (8 * k)));
  std::FILE* f = std::fopen(path, "wb");
  if (!f) return false;
  auto put32 = [&](uint32_t v) { const uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v}; std::fwrite(b, 1, 4, f); };
  auto chunk = [&](const char* tag, const std::vector<uint8_t>& data) {
    std::vector<uint8_t> td(tag, tag + 4);
    td.insert(td.end(), data.begin(), data.end());
    put32((uint32_t)data.size());
    std::fwrite(td.data(), 1, td.size(), f);
    put32(detail::crc32(td.data(), td.size()));
  };
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  std::fwrite(sig, 1, 8, f);
  std::vector<uint8_t> ihdr(13, 0);
  for (int k = 0; k < 4; k++) { ihdr[(std::size_t)k] = (uint8_t)(width >> (8 * (3 - k))); ihdr[(std::size_t)(4 + k)] = (uint8_t)(height >> (8 * (3 - k))); }
  ihdr[8] = 8; ihdr[9] = 2;
  chunk("IHDR", ihdr);
  chunk("IDAT", z);
  chunk("IEND", {});
  return std::fclose(f) == 0;
}

} // namespace pt::image_io
