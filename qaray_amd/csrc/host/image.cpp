// image.cpp — see image.h.
#include "image.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace qaray_hip {

// ---------------------------------------------------------------------------------------------
// CRC-32 (PNG chunks) and Adler-32 (zlib trailer)
// ---------------------------------------------------------------------------------------------
uint32_t Crc32(const unsigned char *p, size_t n, uint32_t crc)
{
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
  return ~crc;
}

static uint32_t Adler32(const unsigned char *p, size_t n)
{
  uint32_t a = 1, b = 0;
  for (size_t i = 0; i < n; ++i) { a = (a + p[i]) % 65521u; b = (b + a) % 65521u; }
  return (b << 16) | a;
}

// ---------------------------------------------------------------------------------------------
// Inflate (RFC 1951) behind a zlib header (RFC 1950)
// ---------------------------------------------------------------------------------------------
namespace {

struct BitReader {
  const unsigned char *p;
  size_t n, pos = 0;
  uint32_t acc = 0;
  int cnt = 0;
  bool bad = false;
  uint32_t Bits(int k)
  {
    while (cnt < k) {
      if (pos >= n) { bad = true; return 0; }
      acc |= (uint32_t) p[pos++] << cnt;
      cnt += 8;
    }
    const uint32_t v = acc & ((k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1));
    acc >>= k;
    cnt -= k;
    return v;
  }
  void AlignByte() { acc = 0; cnt = 0; }
};

// Canonical Huffman decoder (count/symbol tables, bit-serial decode)
struct Huffman {
  uint16_t count[16];
  uint16_t symbol[320];
  bool Build(const unsigned char *lengths, int n)
  {
    memset(count, 0, sizeof(count));
    for (int i = 0; i < n; ++i) count[lengths[i]]++;
    int left = 1;
    for (int len = 1; len < 16; ++len) {
      left <<= 1;
      left -= count[len];
      if (left < 0) return false;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = offs[len] + count[len];
    for (int i = 0; i < n; ++i) if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t) i;
    return true;
  }
  int Decode(BitReader &br) const
  {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; ++len) {
      code |= (int) br.Bits(1);
      if (br.bad) return -1;
      const int c = count[len];
      if (code - c < first) return symbol[index + (code - first)];
      index += c;
      first += c;
      first <<= 1;
      code <<= 1;
    }
    return -1;
  }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint16_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint16_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

bool InflateBlock(BitReader &br, const Huffman &lit, const Huffman &dist, std::vector<unsigned char> &out)
{
  while (true) {
    const int sym = lit.Decode(br);
    if (sym < 0) return false;
    if (sym < 256) { out.push_back((unsigned char) sym); continue; }
    if (sym == 256) return true;
    const int li = sym - 257;
    if (li >= 29) return false;
    const int len = kLenBase[li] + (int) br.Bits(kLenExtra[li]);
    const int ds = dist.Decode(br);
    if (ds < 0 || ds >= 30) return false;
    const size_t d = kDistBase[ds] + br.Bits(kDistExtra[ds]);
    if (br.bad || d > out.size()) return false;
    const size_t from = out.size() - d;
    for (int i = 0; i < len; ++i) out.push_back(out[from + i]);
  }
}

}  // namespace

bool Inflate(const unsigned char *src, size_t n, std::vector<unsigned char> &out)
{
  out.clear();
  if (n < 6) return false;
  if ((src[0] & 0x0F) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return false;
  BitReader br{src + 2, n - 2};
  bool last = false;
  while (!last) {
    last = br.Bits(1) != 0;
    const uint32_t type = br.Bits(2);
    if (br.bad) return false;
    if (type == 0) {
      br.AlignByte();
      if (br.pos + 4 > br.n) return false;
      const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8);
      const uint32_t nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
      br.pos += 4;
      if ((len ^ 0xFFFF) != nlen || br.pos + len > br.n) return false;
      out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
      br.pos += len;
    } else if (type == 1) {
      unsigned char l[288];
      for (int i = 0; i < 144; ++i) l[i] = 8;
      for (int i = 144; i < 256; ++i) l[i] = 9;
      for (int i = 256; i < 280; ++i) l[i] = 7;
      for (int i = 280; i < 288; ++i) l[i] = 8;
      unsigned char d[30];
      for (int i = 0; i < 30; ++i) d[i] = 5;
      Huffman lit, dist;
      lit.Build(l, 288);
      dist.Build(d, 30);
      if (!InflateBlock(br, lit, dist, out)) return false;
    } else if (type == 2) {
      const int nlen = (int) br.Bits(5) + 257, ndist = (int) br.Bits(5) + 1, ncode = (int) br.Bits(4) + 4;
      if (br.bad || nlen > 286 || ndist > 30) return false;
      static const unsigned char order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      unsigned char cl[19] = {0};
      for (int i = 0; i < ncode; ++i) cl[order[i]] = (unsigned char) br.Bits(3);
      Huffman clh;
      if (!clh.Build(cl, 19)) return false;
      unsigned char lens[320] = {0};
      int i = 0;
      while (i < nlen + ndist) {
        const int s = clh.Decode(br);
        if (s < 0) return false;
        if (s < 16) lens[i++] = (unsigned char) s;
        else {
          int rep, val = 0;
          if (s == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int) br.Bits(2); }
          else if (s == 17) rep = 3 + (int) br.Bits(3);
          else rep = 11 + (int) br.Bits(7);
          if (i + rep > nlen + ndist) return false;
          while (rep--) lens[i++] = (unsigned char) val;
        }
      }
      Huffman lit, dist;
      if (!lit.Build(lens, nlen)) return false;
      dist.Build(lens + nlen, ndist);  // incomplete distance codes are legal
      if (!InflateBlock(br, lit, dist, out)) return false;
    } else return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// PNG
// ---------------------------------------------------------------------------------------------
static uint32_t Be32(const unsigned char *p) { return ((uint32_t) p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

static int Paeth(int a, int b, int c)
{
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

bool LoadPNG(const char *filename, int &width, int &height, std::vector<unsigned char> &rgb, std::string *err)
{
  auto fail = [&](const char *why) { if (err) *err = why; return false; };
  FILE *f = fopen(filename, "rb");
  if (!f) return fail("cannot open file");
  std::vector<unsigned char> buf;
  unsigned char tmp[65536];
  size_t n;
  while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  fclose(f);
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (buf.size() < 8 || memcmp(buf.data(), sig, 8) != 0) return fail("not a PNG");
  size_t pos = 8;
  int depth = 0, ctype = 0, interlace = 0;
  bool haveHdr = false;
  std::vector<unsigned char> idat, palette;
  while (pos + 12 <= buf.size()) {
    const uint32_t len = Be32(&buf[pos]);
    const unsigned char *type = &buf[pos + 4];
    const unsigned char *data = &buf[pos + 8];
    if (pos + 12 + len > buf.size()) return fail("truncated chunk");
    if (memcmp(type, "IHDR", 4) == 0 && len >= 13) {
      width = (int) Be32(data);
      height = (int) Be32(data + 4);
      depth = data[8];
      ctype = data[9];
      interlace = data[12];
      haveHdr = true;
    } else if (memcmp(type, "PLTE", 4) == 0) palette.assign(data, data + len);
    else if (memcmp(type, "IDAT", 4) == 0) idat.insert(idat.end(), data, data + len);
    else if (memcmp(type, "IEND", 4) == 0) break;
    pos += 12 + len;
  }
  if (!haveHdr || width <= 0 || height <= 0) return fail("missing IHDR");
  if ((unsigned long long) width * (unsigned long long) height > (1ull << 28)) return fail("image too large");
  if (interlace) return fail("interlaced PNG not supported");
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return fail("bad colour type");
  }
  if (depth != 8 && depth != 16 && !(depth < 8 && (ctype == 0 || ctype == 3))) return fail("unsupported bit depth");
  std::vector<unsigned char> raw;
  if (!Inflate(idat.data(), idat.size(), raw)) return fail("inflate failed");
  const size_t bpp = (size_t) (channels * depth + 7) / 8;             // bytes per complete pixel
  const size_t stride = ((size_t) width * channels * depth + 7) / 8;  // bytes per scanline
  if (raw.size() < (stride + 1) * (size_t) height) return fail("short image data");
  std::vector<unsigned char> img(stride * height);
  for (int y = 0; y < height; ++y) {
    const unsigned char *in = &raw[(stride + 1) * y];
    const int ft = in[0];
    unsigned char *cur = &img[stride * y];
    const unsigned char *prev = y ? &img[stride * (y - 1)] : nullptr;
    for (size_t x = 0; x < stride; ++x) {
      const int a = x >= bpp ? cur[x - bpp] : 0;
      const int b = prev ? prev[x] : 0;
      const int c = (prev && x >= bpp) ? prev[x - bpp] : 0;
      int v = in[1 + x];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += Paeth(a, b, c); break;
        default: return fail("bad filter type");
      }
      cur[x] = (unsigned char) v;
    }
  }
  rgb.resize((size_t) width * height * 3);
  for (int y = 0; y < height; ++y) {
    const unsigned char *row = &img[stride * y];
    for (int x = 0; x < width; ++x) {
      unsigned char *o = &rgb[3 * ((size_t) y * width + x)];
      auto sample = [&](int ch) -> unsigned {
        if (depth == 8) return row[(size_t) x * channels + ch];
        if (depth == 16) return row[2 * ((size_t) x * channels + ch)];  // high byte
        const size_t bit = (size_t) x * depth;
        const unsigned v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
        return v;
      };
      if (ctype == 3) {
        const unsigned idx = sample(0);
        if (3 * idx + 2 < palette.size()) { o[0] = palette[3 * idx]; o[1] = palette[3 * idx + 1]; o[2] = palette[3 * idx + 2]; }
        else o[0] = o[1] = o[2] = 0;
      } else if (ctype == 0 || ctype == 4) {
        unsigned g = sample(0);
        if (depth < 8) g = g * 255u / ((1u << depth) - 1);
        o[0] = o[1] = o[2] = (unsigned char) g;
      } else {
        o[0] = (unsigned char) sample(0); o[1] = (unsigned char) sample(1); o[2] = (unsigned char) sample(2);
      }
    }
  }
  return true;
}

static void PutChunk(FILE *f, const char *type, const std::vector<unsigned char> &data)
{
  unsigned char len[4] = {(unsigned char) (data.size() >> 24), (unsigned char) (data.size() >> 16),
                          (unsigned char) (data.size() >> 8), (unsigned char) data.size()};
  fwrite(len, 1, 4, f);
  std::vector<unsigned char> td(type, type + 4);
  td.insert(td.end(), data.begin(), data.end());
  fwrite(td.data(), 1, td.size(), f);
  const uint32_t crc = Crc32(td.data(), td.size());
  unsigned char c[4] = {(unsigned char) (crc >> 24), (unsigned char) (crc >> 16), (unsigned char) (crc >> 8), (unsigned char) crc};
  fwrite(c, 1, 4, f);
}

bool SavePNG(const char *filename, const unsigned char *data, int width, int height, int comps)
{
  if (comps != 1 && comps != 3) return false;
  FILE *f = fopen(filename, "wb");
  if (!f) return false;
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  fwrite(sig, 1, 8, f);
  std::vector<unsigned char> ihdr(13);
  ihdr[0] = width >> 24; ihdr[1] = width >> 16; ihdr[2] = width >> 8; ihdr[3] = width;
  ihdr[4] = height >> 24; ihdr[5] = height >> 16; ihdr[6] = height >> 8; ihdr[7] = height;
  ihdr[8] = 8; ihdr[9] = comps == 1 ? 0 : 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  PutChunk(f, "IHDR", ihdr);
  // filter type 0 scanlines in stored (uncompressed) deflate blocks
  const size_t stride = (size_t) width * comps;
  std::vector<unsigned char> raw;
  raw.reserve((stride + 1) * height);
  for (int y = 0; y < height; ++y) {
    raw.push_back(0);
    raw.insert(raw.end(), data + stride * y, data + stride * (y + 1));
  }
  std::vector<unsigned char> z;
  z.push_back(0x78); z.push_back(0x01);
  size_t off = 0;
  do {
    const size_t n = std::min<size_t>(65535, raw.size() - off);
    z.push_back(off + n >= raw.size() ? 1 : 0);
    z.push_back(n & 0xFF); z.push_back(n >> 8);
    z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
    z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
    off += n;
  } while (off < raw.size());
  const uint32_t ad = Adler32(raw.data(), raw.size());
  z.push_back(ad >> 24); z.push_back(ad >> 16); z.push_back(ad >> 8); z.push_back(ad);
  PutChunk(f, "IDAT", z);
  PutChunk(f, "IEND", {});
  fclose(f);
  return true;
}

// src/textures/texture.cpp:36-56 (P6, comments allowed after the magic and the size line)
bool LoadPPM(const char *filename, int &width, int &height, std::vector<unsigned char> &rgb)
{
  FILE *fp = fopen(filename, "rb");
  if (!fp) return false;
  char line[1024];
  auto readLine = [&]() { return fgets(line, sizeof(line), fp) != nullptr; };
  bool ok = readLine() && line[0] == 'P' && line[1] == '6';
  if (ok) {
    ok = readLine();
    while (ok && line[0] == '#') ok = readLine();
    ok = ok && sscanf(line, "%d %d", &width, &height) == 2;
  }
  if (ok) {
    ok = readLine();
    while (ok && line[0] == '#') ok = readLine();
  }
  if (ok && width > 0 && height > 0) {
    // never allocate more than the file can hold
    const long here = ftell(fp);
    fseek(fp, 0, SEEK_END);
    const long left = ftell(fp) - here;
    fseek(fp, here, SEEK_SET);
    if ((unsigned long long) width * (unsigned long long) height * 3ull > (unsigned long long) (left < 0 ? 0 : left)) {
      fclose(fp);
      return false;
    }
    rgb.assign((size_t) width * height * 3, 0);
    const size_t got = fread(rgb.data(), 3, (size_t) width * height, fp);
    (void) got;
  } else ok = false;
  fclose(fp);
  return ok;
}

}  // namespace qaray_hip
