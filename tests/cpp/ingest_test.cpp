// Host-only test program: the C++ ingest (include/hunter_ingest.hpp) on the reference's own files.
//   ingest_test <task.info> <hunter.urdf> <reference.info> <gait.info> <out.bin>   -> writes the HB02 image and prints a summary
//   ingest_test --roundtrip <in.bin> <out.bin>                                      -> load + write (format round trip)
#include <cstdio>
#include <cstring>
#include <hunter_ingest.hpp>
int main(int argc, char** argv) {
  try {
    if (argc == 4 && std::strcmp(argv[1], "--roundtrip") == 0) {
      hunter_hip::writeParametersBlob(hunter_hip::loadParametersBlob(argv[2]), argv[3]);
      return 0;
    }
    if (argc != 6) { std::fprintf(stderr, "usage\n"); return 2; }
    const hunter_hip::Parameters p = hunter_hip::loadParameters(argv[1], argv[2], argv[3], argv[4]);
    hunter_hip::writeParametersBlob(p, argv[5]);
    double mass = 0;
    for (int b = 0; b < HB_NBODY; ++b) mass += p.model.mass[b];
    std::printf("total_mass %.17g horizon %.17g gaits %zu trot_modes %zu\n", mass, p.timeHorizon, p.gaits.size(),
                p.gaits.count("trot") ? p.gaits.at("trot").modes.size() : size_t(0));
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ingest_test: %s\n", e.what());
    return 1;
  }
}
