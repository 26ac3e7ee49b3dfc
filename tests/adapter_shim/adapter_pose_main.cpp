// tests/adapter_shim/adapter_pose_main.cpp — TEST INFRASTRUCTURE.
// Drives Planar_SLAM::Optimizer::PoseOptimization / TranslationOptimization AS DEFINED BY include/planar_adapters.hpp (the C++ drop-in over
// libplanar_hip.so) on stand-in Frames.  File formats are those of oracle/ref_opt_main.cpp's `pose` mode (tests/oracle_lib.run_ref_pose), so
// the same inputs run through the real reference (oracle/_ref/ref_opt -> tests/golden/opt_ref.npz) and through this binary.
//   adapter_pose pose <in.bin> <out.bin>
#include <cstdio>
#include <cstring>
#include <vector>

#include "Optimizer.h"

using namespace Planar_SLAM;

float Frame::fx, Frame::fy, Frame::cx, Frame::cy, Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY, Frame::mfGridElementWidthInv,
    Frame::mfGridElementHeightInv;

struct Reader {
    std::vector<unsigned char> buf;
    size_t off = 0;
    explicit Reader(const char* path) {
        FILE* f = std::fopen(path, "rb");
        if (!f) { std::perror(path); std::exit(2); }
        std::fseek(f, 0, SEEK_END); buf.resize((size_t)std::ftell(f)); std::fseek(f, 0, SEEK_SET);
        if (std::fread(buf.data(), 1, buf.size(), f) != buf.size()) std::exit(2);
        std::fclose(f);
    }
    template <typename T> T get() { T v; std::memcpy(&v, &buf[off], sizeof(T)); off += sizeof(T); return v; }
    template <typename T> const T* arr(size_t n) { const T* p = (const T*)&buf[off]; off += n * sizeof(T); return p; }
};

static cv::Mat mat_f32(int r, int c, const float* src) { cv::Mat m(r, c, CV_32F); std::memcpy(m.data, src, sizeof(float) * r * c); return m; }

int main(int argc, char** argv) {
    if (argc != 4 || std::string(argv[1]) != "pose") { std::fprintf(stderr, "usage: adapter_pose pose <in> <out>\n"); return 2; }
    Reader r(argv[2]);
    const int B = r.get<int>(), MP = r.get<int>(), ML = r.get<int>(), MM = r.get<int>(), mode = r.get<int>();
    const float* cam = r.arr<float>(5);
    const double* cfg = r.arr<double>(6);
    const char* names[6] = {"Plane.AngleInfo", "Plane.DistanceInfo", "Plane.ParallelInfo", "Plane.VerticalInfo", "Plane.Chi", "Plane.VPChi"};
    for (int i = 0; i < 6; i++) Config::table()[names[i]] = cfg[i];
    const int* n_points = r.arr<int>(B); const int* n_lines = r.arr<int>(B); const int* n_planes = r.arr<int>(B);
    const unsigned char* pt_valid = r.arr<unsigned char>((size_t)B * MP);
    const float* pt_xw = r.arr<float>((size_t)B * MP * 3);
    const float* pt_obs = r.arr<float>((size_t)B * MP * 3);
    const float* pt_is2 = r.arr<float>((size_t)B * MP);
    const unsigned char* ln_valid = r.arr<unsigned char>((size_t)B * ML);
    const double* ln_obs = r.arr<double>((size_t)B * ML * 3);
    const double* ln_xw = r.arr<double>((size_t)B * ML * 6);
    const float* pl_meas = r.arr<float>((size_t)B * MM * 4);
    const unsigned char* pl_valid = r.arr<unsigned char>((size_t)B * MM * 3);
    const float* pl_world = r.arr<float>((size_t)B * MM * 12);
    const float* Tcw = r.arr<float>((size_t)B * 16);
    FILE* fo = std::fopen(argv[3], "wb");
    if (!fo) return 2;
    for (int b = 0; b < B; b++) {
        Frame F;
        Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3]; F.mbf = cam[4];
        F.mTcw = mat_f32(4, 4, Tcw + (size_t)b * 16);
        const int N = n_points[b], NL = n_lines[b], M = n_planes[b];
        std::vector<MapPoint> mps(N);
        std::vector<MapLine> mls(NL);
        std::vector<MapPlane> mpl((size_t)M * 3);
        F.N = N; F.mvKeysUn.resize(N); F.mvuRight.resize(N); F.mvpMapPoints.assign(N, nullptr); F.mvbOutlier.assign(N, false); F.mvInvLevelSigma2.resize(N);
        for (int i = 0; i < N; i++) {
            const size_t k = (size_t)b * MP + i;
            F.mvKeysUn[i].pt.x = pt_obs[k * 3]; F.mvKeysUn[i].pt.y = pt_obs[k * 3 + 1]; F.mvKeysUn[i].octave = i;   // one "level" per point
            F.mvuRight[i] = pt_obs[k * 3 + 2]; F.mvInvLevelSigma2[i] = pt_is2[k];
            if (pt_valid[k]) { mps[i].pos = mat_f32(3, 1, pt_xw + k * 3); F.mvpMapPoints[i] = &mps[i]; }
        }
        F.NL = NL; F.mvKeyLineFunctions.resize(NL); F.mvpMapLines.assign(NL, nullptr); F.mvbLineOutlier.assign(NL, false);
        for (int i = 0; i < NL; i++) {
            const size_t k = (size_t)b * ML + i;
            F.mvKeyLineFunctions[i] = Eigen::Vector3d(ln_obs[k * 3], ln_obs[k * 3 + 1], ln_obs[k * 3 + 2]);
            if (ln_valid[k]) { for (int j = 0; j < 6; j++) mls[i].mWorldPos[j] = ln_xw[k * 6 + j]; F.mvpMapLines[i] = &mls[i]; }
        }
        F.mnPlaneNum = M; F.mvPlaneCoefficients.resize(M);
        F.mvpMapPlanes.assign(M, nullptr); F.mvpParallelPlanes.assign(M, nullptr); F.mvpVerticalPlanes.assign(M, nullptr);
        F.mvbPlaneOutlier.assign(M, false); F.mvbParPlaneOutlier.assign(M, false); F.mvbVerPlaneOutlier.assign(M, false);
        for (int i = 0; i < M; i++) {
            const size_t k = (size_t)b * MM + i;
            F.mvPlaneCoefficients[i] = mat_f32(4, 1, pl_meas + k * 4);
            for (int j = 0; j < 3; j++)
                if (pl_valid[k * 3 + j]) {
                    MapPlane* p = &mpl[(size_t)i * 3 + j];
                    p->pos = mat_f32(4, 1, pl_world + (k * 3 + j) * 4);
                    (j == 0 ? F.mvpMapPlanes : j == 1 ? F.mvpParallelPlanes : F.mvpVerticalPlanes)[i] = p;
                }
        }
        const int ret = mode == 0 ? Optimizer::PoseOptimization(&F) : Optimizer::TranslationOptimization(&F);
        std::fwrite(&ret, 4, 1, fo);
        std::fwrite(F.mTcw.data, 4, 16, fo);
        std::vector<unsigned char> fl;
        fl.assign((size_t)MP, 0); for (int i = 0; i < N; i++) fl[i] = F.mvbOutlier[i]; std::fwrite(fl.data(), 1, fl.size(), fo);
        fl.assign((size_t)ML, 0); for (int i = 0; i < NL; i++) fl[i] = F.mvbLineOutlier[i]; std::fwrite(fl.data(), 1, fl.size(), fo);
        fl.assign((size_t)MM * 3, 0);
        for (int i = 0; i < M; i++) { fl[i * 3] = F.mvbPlaneOutlier[i]; fl[i * 3 + 1] = F.mvbParPlaneOutlier[i]; fl[i * 3 + 2] = F.mvbVerPlaneOutlier[i]; }
        std::fwrite(fl.data(), 1, fl.size(), fo);
    }
    std::fclose(fo);
    return 0;
}
