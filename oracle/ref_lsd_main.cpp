// oracle/ref_lsd_main.cpp — TEST INFRASTRUCTURE.  Driver for the REAL reference line-extraction wrapper, src/LSDextractor.cpp,
// compiled where it lies into oracle/_ref/ref_lsd.  LSDDetector / BinaryDescriptor (opencv_contrib, un-vendored) are stand-ins that
// forward to oracle/lsd_oracle.cpp, so what this pins is LineSegment::ExtractLineSegment itself: the call sequence, the std::sort by
// response, the cut to 40, the re-numbering and the homogeneous line equations.
//   ref_lsd <gray.raw> <W> <H> <tie_order> <out.bin>      out: int32 n; n x KeyLine(68 B); n x 32 B; n x 3 double
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "LSDextractor.h"

namespace orc { int g_tie_order = 1; }

int main(int argc, char** argv) {
    if (argc != 6) { std::fprintf(stderr, "usage: ref_lsd <gray.raw> <W> <H> <tie_order> <out.bin>\n"); return 2; }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]);
    orc::g_tie_order = std::atoi(argv[4]);
    std::vector<unsigned char> buf((size_t)W * H);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::fclose(f);
    cv::Mat img(H, W, CV_8UC1, buf.data());
    std::vector<cv::line_descriptor::KeyLine> keylines;
    cv::Mat ldesc;
    std::vector<Eigen::Vector3d> eqs;
    Planar_SLAM::LineSegment ls;
    ls.ExtractLineSegment(img, keylines, ldesc, eqs);
    FILE* o = std::fopen(argv[5], "wb");
    if (!o) return 2;
    const int n = (int)keylines.size();
    std::fwrite(&n, 4, 1, o);
    if (n) {
        std::fwrite(keylines.data(), sizeof(cv::line_descriptor::KeyLine), n, o);
        for (int i = 0; i < n; i++) std::fwrite(ldesc.ptr(i), 1, 32, o);
        for (int i = 0; i < n; i++) std::fwrite(eqs[i].data(), sizeof(double), 3, o);
    }
    std::fclose(o);
    return 0;
}
