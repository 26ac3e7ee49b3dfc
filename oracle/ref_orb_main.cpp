// oracle/ref_orb_main.cpp — TEST INFRASTRUCTURE, not product code.
//
// Driver for the REAL reference ORB extractor: links /root/reference/src/ORBextractor.cc
// (compiled where it lies, against oracle/shim) and dumps its outputs, so the CPU
// restatement in orb_oracle.cpp can be pinned against the reference's own code for the
// in-tree arithmetic (cell grid, threshold fallback, octree, IC_Angle, rBRIEF, level
// concatenation).  Built only into oracle/_ref/ (git-ignored).
//
// usage: ref_orb in.raw W H nfeatures scaleFactor nlevels iniTh minTh out.bin
// out.bin: int32 n | n*28 B cv::KeyPoint | n*32 B descriptors | int32 nlevels | per level: int32 w,h | w*h px
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include "ORBextractor.h"

// Monotonic (bump) allocator: the reference breaks octree ties by comparing heap addresses
// (src/ORBextractor.cc:684 sorts pair<int,ExtractorNode*>); with addresses increasing in
// allocation order the tie-break becomes "creation order", which is what the oracle restates.
namespace {
char* g_cur = nullptr; char* g_end = nullptr;
void* bump(std::size_t n) {
    n = (n + 15) & ~std::size_t(15);
    if (!g_cur || g_cur + n > g_end) {
        std::size_t chunk = n > (std::size_t(1) << 30) ? n : (std::size_t(1) << 30);
        // chunks are requested in increasing address order only once in practice (1 GiB);
        // abort rather than silently lose monotonicity.
        if (g_cur) { std::fprintf(stderr, "ref_orb: bump arena exhausted\n"); std::abort(); }
        g_cur = (char*)std::malloc(chunk); g_end = g_cur + chunk;
        if (!g_cur) std::abort();
    }
    void* p = g_cur; g_cur += n; return p;
}
}  // namespace
void* operator new(std::size_t n) { return bump(n); }
void* operator new[](std::size_t n) { return bump(n); }
void operator delete(void*) noexcept {}
void operator delete[](void*) noexcept {}
void operator delete(void*, std::size_t) noexcept {}
void operator delete[](void*, std::size_t) noexcept {}

int main(int argc, char** argv) {
    if (argc != 10) { std::fprintf(stderr, "usage: %s in.raw W H nfeatures scale nlevels ini min out.bin\n", argv[0]); return 2; }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]);
    cv::Mat img(H, W, CV_8UC1);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(img.data, 1, (size_t)W * H, f) != (size_t)W * H) { std::fprintf(stderr, "read failed\n"); return 1; }
    std::fclose(f);
    Planar_SLAM::ORBextractor ex(std::atoi(argv[4]), (float)std::atof(argv[5]), std::atoi(argv[6]), std::atoi(argv[7]), std::atoi(argv[8]));
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    ex(img, cv::Mat(), kps, desc);
    FILE* o = std::fopen(argv[9], "wb");
    int n = (int)kps.size();
    std::fwrite(&n, 4, 1, o);
    if (n) std::fwrite(kps.data(), sizeof(cv::KeyPoint), n, o);
    for (int i = 0; i < n; i++) std::fwrite(desc.ptr(i), 1, 32, o);
    int nl = ex.GetLevels();
    std::fwrite(&nl, 4, 1, o);
    for (int l = 0; l < nl; l++) {
        const cv::Mat& m = ex.mvImagePyramid[l];
        std::fwrite(&m.cols, 4, 1, o); std::fwrite(&m.rows, 4, 1, o);
        for (int y = 0; y < m.rows; y++) std::fwrite(m.ptr(y), 1, m.cols, o);
    }
    std::fclose(o);
    return 0;
}
