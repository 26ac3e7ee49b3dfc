// oracle/ref_peac_main.cpp — TEST INFRASTRUCTURE, not product code.
//
// Driver for the REAL reference plane extractor: compiles /root/reference/src/PlaneExtractor.cpp and
// include/peac/*.hpp (where they lie) against oracle/shim and dumps the outputs Frame::ComputePlanes
// reads (src/Frame.cc:652-672).  Un-vendored pieces replaced by restatements: Eigen's 3x3 eigen-solver
// (oracle/eigprim.cpp).  Built only into oracle/_ref/ (git-ignored).
//
// usage: ref_peac depth.raw(u16) W H fx fy cx cy depthMapFactor out.bin
// out.bin: int32 nplanes | per plane: int32 N_stats, double normal[3], center[3], mse, int32 nverts |
//          int32 labels[H*W] (-1 = none, from plane_vertices_)
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include "PlaneExtractor.h"

// Monotonic allocator: PlaneSeg::NbSet is std::set<PlaneSeg*> (AHCPlaneSeg.hpp:188), iterated in address
// order by ahCluster (:1023-1047); with addresses increasing in allocation order that is creation order,
// which is what the oracle restates.
namespace {
char* g_cur = nullptr; char* g_end = nullptr;
void* bump(std::size_t n) {
    n = (n + 15) & ~std::size_t(15);
    if (!g_cur || g_cur + n > g_end) {
        if (g_cur) { std::fprintf(stderr, "ref_peac: bump arena exhausted\n"); std::abort(); }
        const std::size_t chunk = std::size_t(3) << 30;
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
    if (argc != 10) { std::fprintf(stderr, "usage: %s depth.raw W H fx fy cx cy factor out.bin\n", argv[0]); return 2; }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]);
    cv::Mat depth(H, W, CV_16U);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(depth.data, 2, (size_t)W * H, f) != (size_t)W * H) { std::fprintf(stderr, "read failed\n"); return 1; }
    std::fclose(f);
    cv::Mat K(3, 3, CV_32F);
    for (int i = 0; i < 9; i++) K.at<float>(i) = 0;
    K.at<float>(0, 0) = (float)std::atof(argv[4]); K.at<float>(1, 1) = (float)std::atof(argv[5]);
    K.at<float>(0, 2) = (float)std::atof(argv[6]); K.at<float>(1, 2) = (float)std::atof(argv[7]); K.at<float>(2, 2) = 1;
    const float factor = (float)std::atof(argv[8]);
    PlaneDetection pd;
    if (!pd.readDepthImage(depth, K, factor)) return 1;
    pd.runPlaneDetection(H, W);
    FILE* o = std::fopen(argv[9], "wb");
    int n = pd.plane_num_;
    std::fwrite(&n, 4, 1, o);
    std::vector<int> labels((size_t)W * H, -1);
    for (int i = 0; i < n; i++) {
        const auto& pl = *pd.plane_filter.extractedPlanes[i];
        std::fwrite(&pl.N, 4, 1, o);
        std::fwrite(pl.normal, 8, 3, o); std::fwrite(pl.center, 8, 3, o); std::fwrite(&pl.mse, 8, 1, o);
        int nv = (int)pd.plane_vertices_[i].size();
        std::fwrite(&nv, 4, 1, o);
        for (int idx : pd.plane_vertices_[i]) labels[idx] = i;
    }
    std::fwrite(labels.data(), 4, labels.size(), o);
    std::fclose(o);
    return 0;
}
