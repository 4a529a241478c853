// descriptor_harness.cpp -- TEST PROGRAM: runs vsg_test_descriptor_passes (tests/host/descriptor_model.inc,
// compiled into region_segmentation.cpp with -DVSG_TEST_MODELS).  descriptor_harness <cases> <seed>
#include <cstdio>
#include <cstdlib>
extern "C" int vsg_test_descriptor_passes(int cases, unsigned seed);
int main(int argc, char** argv) {
  const int cases = argc > 1 ? std::atoi(argv[1]) : 200;
  const unsigned seed = argc > 2 ? (unsigned)std::atoi(argv[2]) : 1u;
  const int rc = vsg_test_descriptor_passes(cases, seed);
  if (rc == 0) std::printf("%d cases identical\n", cases);
  return rc;
}
