// oracle/ref_frame_main.cpp — TEST INFRASTRUCTURE: harness around line ranges of the REAL src/Frame.cc, src/MapPoint.cc, src/MapLine.cpp and
// src/Tracking.cc (extracted at build time into oracle/_ref/gen/frame_extract.cpp by oracle/Makefile; declarations: shim/frame_standins.hpp).
// Output: oracle/_ref/ref_frame (git-ignored).  Modes (flat little-endian arrays in, flat arrays out):
//   manhattan        Tracking::TrackManhattanFrame (+ ProjectSN2Conic, ProjectSN2MF, MeanShift)
//   frustum_points   Frame::isInFrustum(MapPoint*, limit)  (+ MapPoint::PredictScale, Get{Min,Max}DistanceInvariance)
//   frustum_lines    Frame::isInFrustum(MapLine*, limit)   (+ MapLine::PredictScale, ...)
//   area             Frame::AssignFeaturesToGrid + GetFeaturesInArea queries, GetLinesInArea queries
//   plane_world      Frame::ComputePlaneWorldCoeff
//   stereo           Frame::ComputeStereoFromRGBD + UnprojectStereo
//   distinctive      MapPoint::ComputeDistinctiveDescriptors (+ ORBmatcher::DescriptorDistance)
//   normal_depth     MapPoint::UpdateNormalAndDepth (+ KeyFrame::SetPose, GetCameraCenter)
// This file only moves data; every computed number comes out of the reference's own function bodies.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "frame_standins.hpp"

using namespace Planar_SLAM;

namespace {
struct Reader {
    std::vector<unsigned char> buf;
    size_t off = 0;
    explicit Reader(const char* path) {
        FILE* f = fopen(path, "rb");
        if (!f) { perror(path); exit(2); }
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        buf.resize((size_t)n);
        if (fread(buf.data(), 1, (size_t)n, f) != (size_t)n) exit(2);
        fclose(f);
    }
    template <class T> T get() { T v; memcpy(&v, buf.data() + off, sizeof(T)); off += sizeof(T); return v; }
    template <class T> const T* arr(size_t n) { const T* p = (const T*)(buf.data() + off); off += n * sizeof(T); if (off > buf.size()) { fprintf(stderr, "short input\n"); exit(2); } return p; }
};
struct Writer {
    FILE* f;
    explicit Writer(const char* path) { f = fopen(path, "wb"); if (!f) { perror(path); exit(2); } }
    ~Writer() { fclose(f); }
    template <class T> void put(const T& v) { fwrite(&v, sizeof(T), 1, f); }
    template <class T> void arr(const T* p, size_t n) { if (n) fwrite(p, sizeof(T), n, f); }
};
cv::Mat mat_f32(int r, int c, const float* p) { cv::Mat m(r, c, CV_32F); memcpy(m.data, p, sizeof(float) * r * c); return m; }

// camera block shared by the frame modes: fx fy cx cy bf minX maxX minY maxY logScaleFactor (10 floats), nLevels (int)
void set_camera(Frame& F, Reader& r) {
    const float* c = r.arr<float>(10);
    Frame::fx = c[0]; Frame::fy = c[1]; Frame::cx = c[2]; Frame::cy = c[3]; F.mbf = c[4];
    Frame::mnMinX = c[5]; Frame::mnMaxX = c[6]; Frame::mnMinY = c[7]; Frame::mnMaxY = c[8];
    F.mfLogScaleFactor = c[9];
    F.mnScaleLevels = r.get<int>();
    Frame::mfGridElementWidthInv = (float)FRAME_GRID_COLS / (Frame::mnMaxX - Frame::mnMinX);     // src/Frame.cc:113-114
    Frame::mfGridElementHeightInv = (float)FRAME_GRID_ROWS / (Frame::mnMaxY - Frame::mnMinY);
}

int run_manhattan(Reader& r, Writer& w) {
    const int B = r.get<int>();
    for (int b = 0; b < B; b++) {
        const float* R = r.arr<float>(9);
        const int n = r.get<int>(), nl = r.get<int>();
        const float* normals = r.arr<float>((size_t)n * 3);
        const double* lines = r.arr<double>((size_t)nl * 3);
        std::vector<SurfaceNormal> sn(n);
        for (int i = 0; i < n; i++) { sn[i].normal = cv::Point3f(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]); sn[i].FramePosition = cv::Point2i(i, 0); }
        std::vector<FrameLine> fl(nl);
        for (int i = 0; i < nl; i++) { fl[i].direction = cv::Point3d(lines[3 * i], lines[3 * i + 1], lines[3 * i + 2]); fl[i].p = cv::Point2d(i, 0); fl[i].q = cv::Point2d(i, 1); }
        Tracking T;
        cv::Mat last = mat_f32(3, 3, R);
        cv::Mat out = T.TrackManhattanFrame(last, sn, fl);
        w.arr((const float*)out.data, 9);
        // which elements reached MeanShift of axis a: the frame positions / end points ProjectSN2MF pushed (index carried in .x)
        std::vector<unsigned char> member((size_t)n + nl, 0);
        const Frame& F = T.mCurrentFrame;
        for (const cv::Point2i& p : F.vSurfaceNormalx) member[p.x] |= 1;
        for (const cv::Point2i& p : F.vSurfaceNormaly) member[p.x] |= 2;
        for (const cv::Point2i& p : F.vSurfaceNormalz) member[p.x] |= 4;
        for (const auto& v : F.vVanishingLinex) member[n + (int)v[2].x] |= 1;   // pointPair(2) then two push_backs: [2] = end point p
        for (const auto& v : F.vVanishingLiney) member[n + (int)v[2].x] |= 2;
        for (const auto& v : F.vVanishingLinez) member[n + (int)v[2].x] |= 4;
        w.arr(member.data(), member.size());
    }
    return 0;
}

// src/Tracking.cc:251-253 + :1778: mRotation_wc = (Rotation_cm * MF_can^T)^T copied into mTcw's rotation block
int run_manhattan_pose(Reader& r, Writer& w) {
    const int B = r.get<int>();
    for (int b = 0; b < B; b++) {
        const float* R0 = r.arr<float>(9); const float* Rn = r.arr<float>(9); const float* T = r.arr<float>(16);
        Tracking K;
        K.Rotation_cm = mat_f32(3, 3, R0); K.MF_can = mat_f32(3, 3, Rn); K.mCurrentFrame.mTcw = mat_f32(4, 4, T);
        K.ManhattanPoseStatements();
        w.arr((const float*)K.mCurrentFrame.mTcw.data, 16);
    }
    return 0;
}

int run_frustum_points(Reader& r, Writer& w) {
    Frame F;
    set_camera(F, r);
    const float* Tcw = r.arr<float>(16);
    const float limit = r.get<float>();
    const int n = r.get<int>();
    const float* xw = r.arr<float>((size_t)n * 3); const float* nrm = r.arr<float>((size_t)n * 3);
    const float* mind = r.arr<float>(n); const float* maxd = r.arr<float>(n);
    F.SetPose(mat_f32(4, 4, Tcw));
    for (int j = 0; j < n; j++) {
        MapPoint mp;
        mp.mWorldPos = mat_f32(3, 1, xw + 3 * j); mp.mNormalVector = mat_f32(3, 1, nrm + 3 * j);
        mp.mfMinDistance = mind[j]; mp.mfMaxDistance = maxd[j];
        const bool in = F.isInFrustum(&mp, limit);
        const unsigned char v = in ? 1 : 0;
        w.put(v); w.put(mp.mbTrackInView ? (unsigned char)1 : (unsigned char)0);
        w.put(mp.mTrackProjX); w.put(mp.mTrackProjY); w.put(mp.mTrackProjXR); w.put(mp.mnTrackScaleLevel); w.put(mp.mTrackViewCos);
    }
    return 0;
}

int run_frustum_lines(Reader& r, Writer& w) {
    Frame F;
    set_camera(F, r);
    const float* Tcw = r.arr<float>(16);
    const float limit = r.get<float>();
    const int n = r.get<int>();
    const double* xw6 = r.arr<double>((size_t)n * 6); const double* nrm = r.arr<double>((size_t)n * 3);
    const float* mind = r.arr<float>(n); const float* maxd = r.arr<float>(n);
    F.SetPose(mat_f32(4, 4, Tcw));
    for (int j = 0; j < n; j++) {
        MapLine ml;
        for (int k = 0; k < 6; k++) ml.mWorldPos[k] = xw6[6 * j + k];
        ml.mNormalVector = Eigen::Vector3d(nrm[3 * j], nrm[3 * j + 1], nrm[3 * j + 2]);
        ml.mfMinDistance = mind[j]; ml.mfMaxDistance = maxd[j];
        const bool in = F.isInFrustum(&ml, limit);
        w.put(in ? (unsigned char)1 : (unsigned char)0);
        w.put(ml.mTrackProjX1); w.put(ml.mTrackProjY1); w.put(ml.mTrackProjX2); w.put(ml.mTrackProjY2); w.put(ml.mnTrackScaleLevel); w.put(ml.mTrackViewCos);
    }
    return 0;
}

int run_area(Reader& r, Writer& w) {
    Frame F;
    set_camera(F, r);
    const int N = r.get<int>();
    const float* kp = r.arr<float>((size_t)N * 2); const int* oct = r.arr<int>(N);
    F.N = N; F.mvKeysUn.resize(N);
    for (int i = 0; i < N; i++) { F.mvKeysUn[i].pt.x = kp[2 * i]; F.mvKeysUn[i].pt.y = kp[2 * i + 1]; F.mvKeysUn[i].octave = oct[i]; }
    F.AssignFeaturesToGrid();
    const int Q = r.get<int>();
    for (int q = 0; q < Q; q++) {
        const float* a = r.arr<float>(3); const int* lv = r.arr<int>(2);
        std::vector<size_t> v = F.GetFeaturesInArea(a[0], a[1], a[2], lv[0], lv[1]);
        w.put<int>((int)v.size());
        for (size_t x : v) w.put<int>((int)x);
    }
    const int NL = r.get<int>();
    F.mvKeylinesUn.resize(NL);
    for (int i = 0; i < NL; i++) { const float* k = r.arr<float>(3); F.mvKeylinesUn[i].pt.x = k[0]; F.mvKeylinesUn[i].pt.y = k[1]; F.mvKeylinesUn[i].angle = k[2]; F.mvKeylinesUn[i].octave = r.get<int>(); }
    const int QL = r.get<int>();
    for (int q = 0; q < QL; q++) {
        const float* a = r.arr<float>(5); const int* lv = r.arr<int>(2);
        std::vector<size_t> v = F.GetLinesInArea(a[0], a[1], a[2], a[3], a[4], lv[0], lv[1]);
        w.put<int>((int)v.size());
        for (size_t x : v) w.put<int>((int)x);
    }
    return 0;
}

// Frame::ComputeStereoFromRGBD + UnprojectStereo.  The depth image arrives as the float image Tracking::GrabImageRGBD hands to the Frame
// (imDepth.convertTo(CV_32F, mDepthMapFactor), src/Tracking.cc:174-175); that conversion is OpenCV's and done by the caller of this harness.
int run_stereo(Reader& r, Writer& w) {
    Frame F;
    set_camera(F, r);
    Frame::invfx = 1.0f / Frame::fx; Frame::invfy = 1.0f / Frame::fy;       // src/Frame.cc:127-128
    const float* Tcw = r.arr<float>(16);
    const int W = r.get<int>(), H = r.get<int>(), N = r.get<int>();
    const float* depth = r.arr<float>((size_t)W * H);
    const float* kp = r.arr<float>((size_t)N * 2);
    F.SetPose(mat_f32(4, 4, Tcw));
    F.N = N; F.mvKeys.resize(N); F.mvKeysUn.resize(N);
    for (int i = 0; i < N; i++) { F.mvKeys[i].pt.x = kp[2 * i]; F.mvKeys[i].pt.y = kp[2 * i + 1]; F.mvKeysUn[i] = F.mvKeys[i]; }
    cv::Mat im(H, W, CV_32F);
    memcpy(im.data, depth, sizeof(float) * W * H);
    F.ComputeStereoFromRGBD(im);
    for (int i = 0; i < N; i++) {
        w.put(F.mvuRight[i]); w.put(F.mvDepth[i]);
        cv::Mat x = F.UnprojectStereo(i);
        float o[3] = {0, 0, 0};
        if (!x.empty()) for (int k = 0; k < 3; k++) o[k] = x.at<float>(k);
        w.arr(o, 3);
    }
    return 0;
}

int run_plane_world(Reader& r, Writer& w) {
    Frame F;
    const float* Tcw = r.arr<float>(16);
    const int n = r.get<int>();
    F.SetPose(mat_f32(4, 4, Tcw));
    for (int i = 0; i < n; i++) F.mvPlaneCoefficients.push_back(mat_f32(4, 1, r.arr<float>(4)));
    for (int i = 0; i < n; i++) { cv::Mat c = F.ComputePlaneWorldCoeff(i); w.arr((const float*)c.data, 4); }
    return 0;
}
// in: int32 npoints; per point int32 nobs, uint8 bad[nobs], uint8 desc[nobs][32].  out: per point uint8 mDescriptor[32] (zeros if it stays empty).
// The observing key frames live in one array, so the std::map<KeyFrame*, size_t> iterates them in index order.
int run_distinctive(Reader& r, Writer& w) {
    const int np = r.get<int>();
    for (int p = 0; p < np; p++) {
        const int n = r.get<int>();
        const unsigned char* bad = r.arr<unsigned char>(n);
        const unsigned char* d = r.arr<unsigned char>((size_t)n * 32);
        std::vector<KeyFrame> kfs(n);
        MapPoint mp;
        for (int i = 0; i < n; i++) {
            kfs[i].mDescriptors = cv::Mat(1, 32, CV_8U);
            memcpy(kfs[i].mDescriptors.data, d + (size_t)i * 32, 32);
            kfs[i].mbBad = bad[i] != 0;
            mp.mObservations[&kfs[i]] = 0;
        }
        mp.ComputeDistinctiveDescriptors();
        unsigned char out[32] = {0};
        if (!mp.mDescriptor.empty()) memcpy(out, mp.mDescriptor.data, 32);
        w.arr(out, 32);
    }
    return 0;
}
// MapLine::ComputeDistinctiveDescriptors (src/MapLine.cpp:241-312), same format: per line int32 nobs, uint8 bad[nobs], uint8 desc[nobs][32] -> mLDescriptor[32]
int run_distinctive_lines(Reader& r, Writer& w) {
    const int np = r.get<int>();
    for (int p = 0; p < np; p++) {
        const int n = r.get<int>();
        const unsigned char* bad = r.arr<unsigned char>(n);
        const unsigned char* d = r.arr<unsigned char>((size_t)n * 32);
        std::vector<KeyFrame> kfs(n);
        MapLine ml;
        for (int i = 0; i < n; i++) {
            kfs[i].mLineDescriptors = cv::Mat(1, 32, CV_8U);
            memcpy(kfs[i].mLineDescriptors.data, d + (size_t)i * 32, 32);
            kfs[i].mbBad = bad[i] != 0;
            ml.mObservations[&kfs[i]] = 0;
        }
        ml.ComputeDistinctiveDescriptors();
        unsigned char out[32] = {0};
        if (!ml.mLDescriptor.empty()) memcpy(out, ml.mLDescriptor.data, 32);
        w.arr(out, 32);
    }
    return 0;
}
// in: int32 nkf, nlev; float sf[nlev]; per key frame float Tcw[16]; int32 npoints; per point float pos[3], int32 ref (index of mpRefKF), level (octave of its
// keypoint there), nobs, int32 obs[nobs] (observing key frames).  out: per point float normal[3], min, max.
int run_normal_depth(Reader& r, Writer& w) {
    const int nkf = r.get<int>(), nlev = r.get<int>();
    const float* sf = r.arr<float>(nlev);
    std::vector<KeyFrame> kfs(nkf);
    for (int k = 0; k < nkf; k++) {
        kfs[k].SetPose(mat_f32(4, 4, r.arr<float>(16)));
        kfs[k].mvScaleFactors.assign(sf, sf + nlev);
        kfs[k].mnScaleLevels = nlev;
        kfs[k].mvKeysUn.resize(1);
    }
    const int np = r.get<int>();
    for (int p = 0; p < np; p++) {
        MapPoint mp;
        mp.mWorldPos = mat_f32(3, 1, r.arr<float>(3));
        const int ref = r.get<int>(), level = r.get<int>(), nobs = r.get<int>();
        const int* obs = r.arr<int>(nobs);
        for (int i = 0; i < nobs; i++) mp.mObservations[&kfs[obs[i]]] = 0;
        mp.mpRefKF = &kfs[ref];
        kfs[ref].mvKeysUn[0].octave = level;
        mp.mNormalVector = cv::Mat::zeros(3, 1, CV_32F);
        mp.UpdateNormalAndDepth();
        w.arr((const float*)mp.mNormalVector.data, 3);
        w.put(mp.mfMinDistance); w.put(mp.mfMaxDistance);
    }
    return 0;
}
}  // namespace

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: ref_frame manhattan|frustum_points|frustum_lines|area|plane_world <in> <out>\n"); return 2; }
    const std::string m = argv[1];
    Reader r(argv[2]);
    Writer w(argv[3]);
    if (m == "manhattan") return run_manhattan(r, w);
    if (m == "manhattan_pose") return run_manhattan_pose(r, w);
    if (m == "frustum_points") return run_frustum_points(r, w);
    if (m == "frustum_lines") return run_frustum_lines(r, w);
    if (m == "area") return run_area(r, w);
    if (m == "plane_world") return run_plane_world(r, w);
    if (m == "stereo") return run_stereo(r, w);
    if (m == "distinctive") return run_distinctive(r, w);
    if (m == "distinctive_lines") return run_distinctive_lines(r, w);
    if (m == "normal_depth") return run_normal_depth(r, w);
    return 2;
}
