// oracle/ref_line3d_main.cpp — TEST INFRASTRUCTURE.  Driver for the REAL reference 3-D line code: Frame::isLineGood (src/Frame.cc:189-267) and compPt3dCov,
// extract3dline_mahdist, verify3dLine, mah_dist3d_pt_line, computeLine3d_svd (src/LineExtractor.cpp), random_unique (include/LSDextractor.h), extracted by
// line range at build time (oracle/Makefile -> oracle/_ref/gen/line3d_extract.cpp, never committed) and compiled against the OpenCV / Eigen stand-ins
// into oracle/_ref/ref_line3d.  The reference draws from the process-global rand(); the driver calls srand(seed + i) before line i and runs the reference's
// loop on a Frame holding that one key line, which is the seeding the product defines (include/planar_abi.h, planar_is_line_good).
//   ref_line3d <in.bin> <out.bin>
// in : int32 n, W, H; uint32 seed; float fx, fy, cx, cy, factor; n x 68-byte KeyLine; W*H u16 depth
// out: per line: float depth_line; double lines3d[6]; uint8 good; double direction[3]; int32 n_inliers (= rndpts3d.size() of the pushed FrameLine, 0 otherwise)
#include <cstdio>
#include <cstring>

#include "_ref/gen/line3d_extract.cpp"

float Planar_SLAM::Frame::cx, Planar_SLAM::Frame::cy, Planar_SLAM::Frame::invfx, Planar_SLAM::Frame::invfy;

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    FILE* fi = std::fopen(argv[1], "rb");
    if (!fi) return 2;
    int32_t hdr[3]; uint32_t seed; float cam[5];
    if (std::fread(hdr, 4, 3, fi) != 3 || std::fread(&seed, 4, 1, fi) != 1 || std::fread(cam, 4, 5, fi) != 5) return 2;
    const int n = hdr[0], W = hdr[1], H = hdr[2];
    std::vector<KeyLine> kls(n);
    if (n && std::fread(kls.data(), sizeof(KeyLine), n, fi) != (size_t)n) return 2;
    std::vector<uint16_t> d16((size_t)W * H);
    if (std::fread(d16.data(), 2, d16.size(), fi) != d16.size()) return 2;
    std::fclose(fi);
    cv::Mat depth(H, W, CV_32F);                       // imDepth.convertTo(depth, CV_32F, depthMapFactor): float multiply (src/Frame.cc:81-83)
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) depth.at<float>(y, x) = (float)d16[(size_t)y * W + x] * cam[4];
    using Planar_SLAM::Frame;
    Frame::cx = cam[2]; Frame::cy = cam[3]; Frame::invfx = 1.0f / cam[0]; Frame::invfy = 1.0f / cam[1];
    cv::Mat tmpK = (cv::Mat_<double>(3, 3) << cam[0], 0, cam[2], 0, cam[1], cam[3], 0, 0, 1);   // src/Frame.cc:84-86
    cv::Mat gray;
    FILE* fo = std::fopen(argv[2], "wb");
    if (!fo) return 2;
    for (int i = 0; i < n; i++) {
        Frame F;
        F.mvKeylinesUn.assign(1, kls[i]);
        srand(seed + (unsigned)i);
        F.isLineGood(gray, depth, tmpK);
        const float dl = F.mvDepthLine[0];
        double l3[6], dir[3] = {0, 0, 0};
        for (int k = 0; k < 6; k++) l3[k] = F.mvLines3D[0][k];
        const uint8_t good = F.mVF3DLines.empty() ? 0 : 1;
        int32_t ninl = 0;
        if (good) { dir[0] = F.mVF3DLines[0].direction.x; dir[1] = F.mVF3DLines[0].direction.y; dir[2] = F.mVF3DLines[0].direction.z; ninl = (int32_t)F.mVF3DLines[0].rndpts3d.size(); }
        std::fwrite(&dl, 4, 1, fo); std::fwrite(l3, 8, 6, fo); std::fwrite(&good, 1, 1, fo); std::fwrite(dir, 8, 3, fo); std::fwrite(&ninl, 4, 1, fo);
    }
    std::fclose(fo);
    return 0;
}
