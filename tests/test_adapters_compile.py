"""CPU test: include/planar_adapters.hpp (reference-signature C++ classes over the C ABI) compiles and links against
libplanar_hip.so.  OpenCV is not in this image, so the OpenCV stand-in from oracle/shim is used for the headers —
test infrastructure only; no GPU call is made."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#define PLANAR_ADAPTERS_WITH_LINES
#include "planar_adapters.hpp"
int main(int argc, char**) {
    if (argc > 100) {   // never executed here: there is no GPU in the test container; this only has to compile and link
        Planar_SLAM::ORBextractor ex(1000, 1.2f, 8, 20, 7);
        cv::Mat img(480, 640, CV_8UC1), desc;
        std::vector<cv::KeyPoint> kps;
        ex(img, cv::Mat(), kps, desc);
        (void)ex.GetLevels(); (void)ex.GetScaleFactors(); (void)ex.mvImagePyramid.size();
        PlaneDetection pd;
        cv::Mat depth(480, 640, CV_16U), K(3, 3, CV_32F);
        pd.readDepthImage(depth, K, 1.0f / 5000);
        pd.runPlaneDetection(480, 640);
        Planar_SLAM::LineSegment ls;
        std::vector<cv::line_descriptor::KeyLine> kl; cv::Mat ldesc; std::vector<Eigen::Vector3d> eqs;
        ls.ExtractLineSegment(img, kl, ldesc, eqs);
        return pd.plane_num_ + (int)pd.plane_vertices_.size() + (int)pd.plane_filter.extractedPlanes.size() + (int)kl.size();
    }
    return 0;
}
'''


def test_adapters_compile_and_link():
    lib = os.path.join(ROOT, "planarslam_amd", "libplanar_hip.so")
    assert os.path.exists(lib), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-std=c++14", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle", "shim"),
                               src, os.path.join(ROOT, "oracle", "cvprim.cpp"), lib, "-Wl,-rpath," + os.path.dirname(lib),
                               "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
        subprocess.check_call([exe])
