// amber/postprocess.h -- the output stage that follows Algorithm::Render in the reference's CLI
// (application.cc:98-115): Filmic tone map -> Gamma 2.2 -> 8-bit PNG, and the raw image as OpenEXR,
// both mirrored in x (cli/image.cc:45-71).  Host code; binary32 arithmetic in the reference's order.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "prelude.h"

#pragma GCC visibility push(default)   // the C++ interface of the host object model is exported (bin/amber links against it)
namespace amber {
namespace postprocess {

using HDR = prelude::Vector3;                       // postprocess/forward.h:39
using HDRImage = prelude::Image<HDR>;
struct LDRImage {                                    // Image<Vector3<uint_fast8_t>>, row-major x + y*W, RGB
  prelude::pixel_size_type width = 0, height = 0;
  std::vector<std::uint8_t> rgb;
};

/** Filmic (Hable) operator, src/amber/postprocess/filmic.cc:30-67: p = Map(p * 16) / Map(0.70). */
struct Filmic { HDRImage operator()(HDRImage input) const; static HDR Map(const HDR& hdr) noexcept; };
/** Gamma, src/amber/postprocess/gamma.cc:27-52: 255 * min(1, pow(x, 1/gamma)) truncated to 8 bits. */
struct Gamma {
  float gamma = 2.2f;
  LDRImage operator()(const HDRImage& input) const;
};

}  // namespace postprocess

namespace cli {
/** cli/image.cc:45-71.  The reference goes through OpenCV; these are self-contained writers of the same pixels:
 *  column i of the image lands in file column W-1-i.  PNG: 8-bit RGB, zlib "stored" blocks.  EXR: 32-bit float
 *  B,G,R channels, uncompressed scanlines. */
void ExportPNG(const postprocess::LDRImage& image, const std::string& filename);
void ExportEXR(const postprocess::HDRImage& image, const std::string& filename);
}  // namespace cli
}  // namespace amber
#pragma GCC visibility pop
