// ref_xorshift_kat.cpp — known-answer generator built from the REFERENCE's own
// header.  xorshift.hpp is self-contained (standard headers only), so it compiles
// from where it lies under /root/reference/include with plain g++ — no stand-in
// headers, nothing copied.  Output (text) pins oracle/pt_oracle.c:xs32 and, through
// tests/golden/xorshift32_kat.json, the HIP kernel's generator.
//
// Build: see oracle/Makefile (target _ref/xorshift_kat).  TEST INFRASTRUCTURE ONLY.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string_view>

#include "xorshift.hpp" // from -I/root/reference/include

int main(int argc, char** argv) {
  // usage: xorshift_kat <count> <seed> [<seed> ...]   ("default" = xorshift<>::initial_state)
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s count seed...\n", argv[0]);
    return 2;
  }
  const int count = std::atoi(argv[1]);
  for (int a = 2; a < argc; a++) {
    xorshift<32> g;
    if (std::string_view(argv[a]) != "default")
      g = xorshift<32>(static_cast<std::uint32_t>(std::strtoul(argv[a], nullptr, 10)));
    std::printf("%u:", static_cast<unsigned>(g.state));
    for (int i = 0; i < count; i++) std::printf(" %u", static_cast<unsigned>(g()));
    std::printf("\n");
  }
  return 0;
}
