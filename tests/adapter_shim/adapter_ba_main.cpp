// tests/adapter_shim/adapter_ba_main.cpp — TEST INFRASTRUCTURE.
// Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*) AS DEFINED BY include/planar_adapters.hpp (PLANAR_ADAPTERS_WITH_LOCAL_BA, over libplanar_hip.so) on the
// stand-in map objects of oracle/shim/opt_standins.hpp (-DSTANDINS_NO_REFERENCE: no reference headers needed).  The harness (oracle/ref_opt_harness.hpp: run_ba)
// and the file formats are those of oracle/_ref/ref_opt's `ba` mode, where the same call resolves to the REAL src/Optimizer.cc; expected output =
// tests/golden/opt_ref.npz.
//   adapter_ba ba <in.bin> <out.bin>
#include <string>

namespace Planar_SLAM {
class Optimizer {
public:
    void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap);   // include/Optimizer.h:35
};
}  // namespace Planar_SLAM

#define PLANAR_ADAPTERS_WITH_LOCAL_BA
#include "planar_adapters.hpp"

float Planar_SLAM::Frame::fx, Planar_SLAM::Frame::fy, Planar_SLAM::Frame::cx, Planar_SLAM::Frame::cy;
std::mutex Planar_SLAM::MapPoint::mGlobalMutex, Planar_SLAM::MapLine::mGlobalMutex, Planar_SLAM::MapPlane::mGlobalMutex;

#include "ref_opt_harness.hpp"

int main(int argc, char** argv) {
    if (argc != 4 || std::string(argv[1]) != "ba") { std::fprintf(stderr, "usage: adapter_ba ba <in> <out>\n"); return 2; }
    return run_ba(argv[2], argv[3]);
}
