// Test driver of pt/image_io.hpp (tests/test_image_io_cpu.py):
//   image_io_main decode IN OUT.rgb      -> "W H\n" + raw RGB8, or exit code 3 and the failure reason on stderr
//   image_io_main encode W H IN.rgb OUT.png
//   image_io_main many FILE...           -> one line per file: "ok W H" or "err <reason>" (corrupt-file sweeps under ASan / UBSan)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pt/image_io.hpp"

int main(int argc, char** argv) {
  if (argc == 4 && !std::strcmp(argv[1], "decode")) {
    pt::image_io::Image im;
    if (const char* e = pt::image_io::load_rgb8(argv[2], im)) { std::fprintf(stderr, "%s\n", e); return 3; }
    std::FILE* f = std::fopen(argv[3], "wb");
    if (!f) return 2;
    std::fprintf(f, "%zu %zu\n", im.width, im.height);
    std::fwrite(im.rgb.data(), 1, im.rgb.size(), f);
    return std::fclose(f) ? 2 : 0;
  }
  if (argc == 6 && !std::strcmp(argv[1], "encode")) {
    const std::size_t w = std::strtoul(argv[2], nullptr, 10), h = std::strtoul(argv[3], nullptr, 10);
    std::vector<uint8_t> rgb(w * h * 3);
    std::FILE* f = std::fopen(argv[4], "rb");
    if (!f || std::fread(rgb.data(), 1, rgb.size(), f) != rgb.size()) return 2;
    std::fclose(f);
    return pt::image_io::write_png(argv[5], rgb.data(), w, h) ? 0 : 2;
  }
  if (argc >= 3 && !std::strcmp(argv[1], "many")) { // hostile-input sweep: decode every file named, one line each; a crash / sanitizer report is the failure
    for (int i = 2; i < argc; i++) {
      pt::image_io::Image im;
      const char* e = pt::image_io::load_rgb8(argv[i], im);
      if (e) std::printf("err %s\n", e);
      else if (im.rgb.size() != im.width * im.height * 3) { std::printf("BAD size\n"); return 4; }
      else std::printf("ok %zu %zu\n", im.width, im.height);
    }
    return 0;
  }
  return 1;
}
