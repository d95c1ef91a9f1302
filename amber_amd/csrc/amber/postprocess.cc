// postprocess.cc -- Filmic / Gamma tone mapping and the PNG / EXR writers (see postprocess.h).
#include "postprocess.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace amber {
namespace postprocess {

namespace {
// filmic.cc:30-37
constexpr float kA = 0.22f, kB = 0.30f, kC = 0.10f, kD = 0.20f, kE = 0.01f, kF = 0.30f, kW = 0.70f, kExposure = 16.0f;
}  // namespace

HDR Filmic::Map(const HDR& hdr) noexcept {   // filmic.cc:59-66, component-wise with splatted constants
  return (hdr * (hdr * HDR(kA) + HDR(kB * kC)) + HDR(kD * kE)) / (hdr * (hdr * HDR(kA) + HDR(kB)) + HDR(kD * kF)) - HDR(kE / kF);
}

HDRImage Filmic::operator()(HDRImage input) const {   // filmic.cc:46-57
  for (prelude::pixel_size_type j = 0; j < input.Height(); j++)
    for (prelude::pixel_size_type i = 0; i < input.Width(); i++) {
      HDR& p = input[prelude::Pixel(i, j)];
      p = Map(p * HDR(kExposure)) / Map(HDR(kW));
    }
  return input;
}

LDRImage Gamma::operator()(const HDRImage& input) const {   // gamma.cc:36-52
  LDRImage out;
  out.width = input.Width(); out.height = input.Height();
  out.rgb.resize(static_cast<std::size_t>(out.width) * out.height * 3);
  auto quantise = [&](float x) -> std::uint8_t {
    const float v = 255 * std::min<float>(1, std::pow(x, 1 / gamma));
    if (!(v >= 0)) return 0;                 // NaN / negative: the reference's float -> uint8 conversion is undefined there
    return static_cast<std::uint8_t>(v);     // truncation, as the implicit conversion to uint_fast8_t does
  };
  for (prelude::pixel_size_type j = 0; j < input.Height(); j++)
    for (prelude::pixel_size_type i = 0; i < input.Width(); i++) {
      const HDR& h = input[prelude::Pixel(i, j)];
      std::uint8_t* p = &out.rgb[(static_cast<std::size_t>(j) * out.width + i) * 3];
      p[0] = quantise(h.x); p[1] = quantise(h.y); p[2] = quantise(h.z);
    }
  return out;
}

}  // namespace postprocess

namespace cli {

namespace {

struct File {
  std::FILE* f;
  explicit File(const std::string& name) : f(std::fopen(name.c_str(), "wb")) {
    if (!f) throw std::runtime_error("cannot open " + name + " for writing");
  }
  ~File() { if (f) std::fclose(f); }
  void write(const void* p, std::size_t n) { if (n && std::fwrite(p, 1, n, f) != n) throw std::runtime_error("short write"); }
};

std::uint32_t Crc32(const std::uint8_t* p, std::size_t n, std::uint32_t crc = 0) {
  static std::uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (std::uint32_t i = 0; i < 256; i++) {
      std::uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  crc = ~crc;
  for (std::size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return ~crc;
}
void Be32(std::vector<std::uint8_t>& v, std::uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back(static_cast<std::uint8_t>(x >> s)); }
void Chunk(File& f, const char type[4], const std::vector<std::uint8_t>& data) {
  std::vector<std::uint8_t> head; Be32(head, static_cast<std::uint32_t>(data.size()));
  f.write(head.data(), 4);
  std::vector<std::uint8_t> body(type, type + 4);
  body.insert(body.end(), data.begin(), data.end());
  f.write(body.data(), body.size());
  std::vector<std::uint8_t> crc; Be32(crc, Crc32(body.data(), body.size()));
  f.write(crc.data(), 4);
}

}  // namespace

void ExportPNG(const postprocess::LDRImage& image, const std::string& filename) {
  const std::size_t W = image.width, H = image.height;
  // raw scanlines: filter byte 0 + RGB, x mirrored (cli/image.cc:61-69)
  std::vector<std::uint8_t> raw;
  raw.reserve(H * (1 + 3 * W));
  for (std::size_t j = 0; j < H; j++) {
    raw.push_back(0);
    for (std::size_t col = 0; col < W; col++) {
      const std::uint8_t* p = &image.rgb[(j * W + (W - 1 - col)) * 3];
      raw.push_back(p[0]); raw.push_back(p[1]); raw.push_back(p[2]);
    }
  }
  // zlib stream of stored (uncompressed) deflate blocks
  std::vector<std::uint8_t> z = {0x78, 0x01};
  std::uint32_t a = 1, b = 0;
  for (std::uint8_t c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
  std::size_t pos = 0;
  do {
    const std::size_t n = std::min<std::size_t>(65535, raw.size() - pos);
    z.push_back(pos + n == raw.size() ? 1 : 0);
    z.push_back(static_cast<std::uint8_t>(n & 0xff)); z.push_back(static_cast<std::uint8_t>(n >> 8));
    z.push_back(static_cast<std::uint8_t>(~n & 0xff)); z.push_back(static_cast<std::uint8_t>((~n >> 8) & 0xff));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    pos += n;
  } while (pos < raw.size());
  Be32(z, (b << 16) | a);
  File f(filename);
  const std::uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  f.write(sig, 8);
  std::vector<std::uint8_t> ihdr;
  Be32(ihdr, static_cast<std::uint32_t>(W)); Be32(ihdr, static_cast<std::uint32_t>(H));
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit, truecolour
  Chunk(f, "IHDR", ihdr);
  Chunk(f, "IDAT", z);
  Chunk(f, "IEND", {});
}

void ExportEXR(const postprocess::HDRImage& image, const std::string& filename) {
  const std::uint32_t W = static_cast<std::uint32_t>(image.Width()), H = static_cast<std::uint32_t>(image.Height());
  std::vector<std::uint8_t> hd;
  auto u32 = [&](std::uint32_t x) { for (int s = 0; s < 32; s += 8) hd.push_back(static_cast<std::uint8_t>(x >> s)); };
  auto f32 = [&](float x) { std::uint32_t u; std::memcpy(&u, &x, 4); u32(u); };
  auto str = [&](const char* s) { while (*s) hd.push_back(static_cast<std::uint8_t>(*s++)); hd.push_back(0); };
  auto attr = [&](const char* name, const char* type, std::uint32_t size) { str(name); str(type); u32(size); };
  u32(20000630u); u32(2u);                                   // magic, version 2 (single-part scanline)
  attr("channels", "chlist", 3 * 18 + 1);
  for (const char* ch : {"B", "G", "R"}) { str(ch); u32(2u); hd.push_back(0); hd.push_back(0); hd.push_back(0); hd.push_back(0); u32(1u); u32(1u); }
  hd.push_back(0);
  attr("compression", "compression", 1); hd.push_back(0);
  attr("dataWindow", "box2i", 16); u32(0); u32(0); u32(W - 1); u32(H - 1);
  attr("displayWindow", "box2i", 16); u32(0); u32(0); u32(W - 1); u32(H - 1);
  attr("lineOrder", "lineOrder", 1); hd.push_back(0);
  attr("pixelAspectRatio", "float", 4); f32(1.0f);
  attr("screenWindowCenter", "v2f", 8); f32(0.0f); f32(0.0f);
  attr("screenWindowWidth", "float", 4); f32(1.0f);
  hd.push_back(0);                                           // end of header
  const std::uint64_t line_bytes = 8ull + 12ull * W;
  const std::uint64_t table_at = hd.size(), data_at = table_at + 8ull * H;
  for (std::uint32_t y = 0; y < H; y++) {
    const std::uint64_t off = data_at + y * line_bytes;
    for (int s = 0; s < 64; s += 8) hd.push_back(static_cast<std::uint8_t>(off >> s));
  }
  File f(filename);
  f.write(hd.data(), hd.size());
  std::vector<float> line(3 * static_cast<std::size_t>(W));
  for (std::uint32_t y = 0; y < H; y++) {
    for (std::uint32_t col = 0; col < W; col++) {             // x mirrored (cli/image.cc:50), channels B, G, R
      const postprocess::HDR& p = image[prelude::Pixel(W - 1 - col, y)];
      line[col] = p.z; line[W + col] = p.y; line[2 * static_cast<std::size_t>(W) + col] = p.x;
    }
    const std::uint32_t head[2] = {y, 12u * W};
    f.write(head, 8);
    f.write(line.data(), line.size() * 4);
  }
}

}  // namespace cli
}  // namespace amber
