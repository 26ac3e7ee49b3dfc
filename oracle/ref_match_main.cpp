// oracle/ref_match_main.cpp — TEST INFRASTRUCTURE.  Driver for the REAL reference matchers, compiled from
// /root/reference/src/ORBmatcher.cc, src/LSDmatcher.cpp and src/PlaneMatcher.cpp where they lie (never copied) against the stand-ins in
// oracle/shim (cvshim.hpp + cvalgebra.hpp + match_standins.hpp) into oracle/_ref/ref_match.
//   ref_match <mode> <in.bin> <out.bin>     mode = proj_frame | proj_map | bow | match_orb | plane | lsd_proj | lsd_desc | fuse | lsd_fuse
// in/out files are sequences of blocks {int64 nbytes; bytes}; tests/oracle_lib.py (run_ref_match) writes and reads them.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "LSDmatcher.h"
#include "ORBmatcher.h"
#include "PlaneMatcher.h"

using namespace Planar_SLAM;

float Frame::fx, Frame::fy, Frame::cx, Frame::cy, Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY, Frame::mfGridElementWidthInv,
    Frame::mfGridElementHeightInv;

struct Blocks {
    std::vector<std::vector<uint8_t>> b;
    size_t next = 0;
    bool load(const char* path) {
        FILE* f = std::fopen(path, "rb");
        if (!f) return false;
        int64_t n;
        while (std::fread(&n, 8, 1, f) == 1) { b.emplace_back((size_t)n); if (n && std::fread(b.back().data(), 1, (size_t)n, f) != (size_t)n) return false; }
        std::fclose(f);
        return true;
    }
    template <typename T> const T* get(size_t* count = nullptr) { auto& v = b.at(next++); if (count) *count = v.size() / sizeof(T); return (const T*)v.data(); }
};
struct Out {
    FILE* f;
    template <typename T> void put(const T* p, size_t n) { int64_t nb = (int64_t)(n * sizeof(T)); std::fwrite(&nb, 8, 1, f); if (nb) std::fwrite(p, 1, (size_t)nb, f); }
};

struct KP7 { float x, y, size, angle, response; int32_t octave, class_id; };

static cv::Mat mat_f32(int r, int c, const float* src) { cv::Mat m(r, c, CV_32F); std::memcpy(m.data, src, sizeof(float) * r * c); return m; }
static cv::Mat desc_mat(int n, const uint8_t* src) { cv::Mat m(n, 32, CV_8UC1); if (n) std::memcpy(m.data, src, (size_t)n * 32); return m; }

// Frame fields shared by proj_frame / proj_map: keys [N], u_right, desc, blocked, intr[12] = {min_x,max_x,min_y,max_y,gw,gh,fx,fy,cx,cy,bf,b}, scale factors
static void fill_frame(Frame& F, Blocks& in, std::vector<MapPoint>& dummies) {
    size_t n;
    const KP7* k = in.get<KP7>(&n);
    const int N = (int)n;
    const float* ur = in.get<float>();
    const uint8_t* desc = in.get<uint8_t>();
    const uint8_t* blocked = in.get<uint8_t>();
    const float* intr = in.get<float>();
    size_t nl;
    const float* sf = in.get<float>(&nl);
    F.N = N;
    F.mvKeysUn.resize(N); F.mvKeys.resize(N); F.mvuRight.assign(ur, ur + N);
    for (int i = 0; i < N; i++) { cv::KeyPoint kp(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id); F.mvKeysUn[i] = kp; F.mvKeys[i] = kp; }
    F.mDescriptors = desc_mat(N, desc);
    Frame::mnMinX = intr[0]; Frame::mnMaxX = intr[1]; Frame::mnMinY = intr[2]; Frame::mnMaxY = intr[3];
    Frame::mfGridElementWidthInv = intr[4]; Frame::mfGridElementHeightInv = intr[5];
    Frame::fx = intr[6]; Frame::fy = intr[7]; Frame::cx = intr[8]; Frame::cy = intr[9];
    F.mbf = intr[10]; F.mb = intr[11];
    F.mvScaleFactors.assign(sf, sf + nl);
    F.mvpMapPoints.assign(N, nullptr);
    dummies.resize(N);
    for (int i = 0; i < N; i++) if (blocked[i]) { dummies[i].nobs = 1; dummies[i].index = -1; F.mvpMapPoints[i] = &dummies[i]; }
    F.AssignFeaturesToGrid();
}

int main(int argc, char** argv) {
    if (argc != 4) { std::fprintf(stderr, "usage: ref_match <mode> <in.bin> <out.bin>\n"); return 2; }
    const std::string mode = argv[1];
    Blocks in;
    if (!in.load(argv[2])) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
    Out out{std::fopen(argv[3], "wb")};
    if (!out.f) return 2;
    const float* prm = in.get<float>();   // mode-specific scalars
    if (mode == "proj_frame") {
        // prm = {th, mono, check_ori}
        Frame Cur, Last;
        std::vector<MapPoint> dummies;
        fill_frame(Cur, in, dummies);
        Cur.mTcw = mat_f32(4, 4, in.get<float>());
        Last.mTcw = mat_f32(4, 4, in.get<float>());
        size_t nl;
        const uint8_t* usable = in.get<uint8_t>(&nl);
        const float* xw = in.get<float>();
        const int32_t* oct = in.get<int32_t>();
        const float* ang = in.get<float>();
        const uint8_t* mpd = in.get<uint8_t>();
        const uint8_t* obs = in.get<uint8_t>();
        const int NL = (int)nl;
        std::vector<MapPoint> mps(NL);
        Last.N = NL; Last.mvpMapPoints.assign(NL, nullptr); Last.mvbOutlier.assign(NL, false); Last.mvKeys.resize(NL); Last.mvKeysUn.resize(NL);
        for (int i = 0; i < NL; i++) {
            Last.mvKeys[i].octave = oct[i]; Last.mvKeysUn[i].angle = ang[i];
            if (usable[i]) { mps[i].pos = mat_f32(3, 1, xw + 3 * i); mps[i].desc = desc_mat(1, mpd + 32 * (size_t)i); mps[i].nobs = obs[i] ? 1 : 0; mps[i].index = i; Last.mvpMapPoints[i] = &mps[i]; }
        }
        ORBmatcher matcher(0.9f, prm[2] != 0);
        const int nm = matcher.SearchByProjection(Cur, Last, prm[0], prm[1] != 0);
        std::vector<int32_t> match(Cur.N, -1);
        for (int i = 0; i < Cur.N; i++) if (Cur.mvpMapPoints[i]) match[i] = Cur.mvpMapPoints[i]->index;
        out.put(match.data(), match.size()); out.put(&nm, 1);
    } else if (mode == "proj_map") {
        // prm = {th, nn_ratio}
        Frame F;
        std::vector<MapPoint> dummies;
        fill_frame(F, in, dummies);
        size_t np;
        const uint8_t* inview = in.get<uint8_t>(&np);
        const float *px = in.get<float>(), *py = in.get<float>(), *pxr = in.get<float>();
        const int32_t* lvl = in.get<int32_t>();
        const float* vc = in.get<float>();
        const uint8_t* d = in.get<uint8_t>();
        const uint8_t* obs = in.get<uint8_t>();
        std::vector<MapPoint> mps(np);
        std::vector<MapPoint*> vp(np);
        for (size_t i = 0; i < np; i++) {
            mps[i].mbTrackInView = inview[i] != 0; mps[i].mTrackProjX = px[i]; mps[i].mTrackProjY = py[i]; mps[i].mTrackProjXR = pxr[i];
            mps[i].mnTrackScaleLevel = lvl[i]; mps[i].mTrackViewCos = vc[i]; mps[i].desc = desc_mat(1, d + 32 * i); mps[i].nobs = obs[i] ? 1 : 0; mps[i].index = (int)i;
            vp[i] = &mps[i];
        }
        ORBmatcher matcher(prm[1], true);
        const int nm = matcher.SearchByProjection(F, vp, prm[0]);
        std::vector<int32_t> match(F.N, -1);
        for (int i = 0; i < F.N; i++) if (F.mvpMapPoints[i]) match[i] = F.mvpMapPoints[i]->index;
        out.put(match.data(), match.size()); out.put(&nm, 1);
    } else if (mode == "bow") {
        // prm = {nn_ratio, check_ori}
        size_t nk, nf;
        const int32_t* knode = in.get<int32_t>(&nk);
        const uint8_t* kusable = in.get<uint8_t>();
        const float* kang = in.get<float>();
        const uint8_t* kdesc = in.get<uint8_t>();
        const int32_t* fnode = in.get<int32_t>(&nf);
        const float* fang = in.get<float>();
        const uint8_t* fdesc = in.get<uint8_t>();
        KeyFrame KF;
        Frame F;
        std::vector<MapPoint> mps(nk);
        KF.N = (int)nk; KF.mps.assign(nk, nullptr); KF.mvKeysUn.resize(nk); KF.mDescriptors = desc_mat((int)nk, kdesc);
        for (size_t i = 0; i < nk; i++) {
            KF.mvKeysUn[i].angle = kang[i];
            if (kusable[i]) { mps[i].index = (int)i; KF.mps[i] = &mps[i]; }
            if (knode[i] >= 0) KF.mFeatVec.addFeature((DBoW2::NodeId)knode[i], (unsigned)i);   // DBoW2 transform(): features in index order
        }
        F.N = (int)nf; F.mvKeys.resize(nf); F.mDescriptors = desc_mat((int)nf, fdesc);
        for (size_t i = 0; i < nf; i++) { F.mvKeys[i].angle = fang[i]; if (fnode[i] >= 0) F.mFeatVec.addFeature((DBoW2::NodeId)fnode[i], (unsigned)i); }
        ORBmatcher matcher(prm[0], prm[1] != 0);
        std::vector<MapPoint*> res;
        const int nm = matcher.SearchByBoW(&KF, F, res);
        std::vector<int32_t> match(nf, -1);
        for (size_t i = 0; i < nf; i++) if (res[i]) match[i] = res[i]->index;
        out.put(match.data(), match.size()); out.put(&nm, 1);
    } else if (mode == "match_orb") {
        size_t nc, nl;
        const uint8_t* cd = in.get<uint8_t>(&nc); nc /= 32;
        const uint8_t* ld = in.get<uint8_t>(&nl); nl /= 32;
        const uint8_t* has = in.get<uint8_t>();
        const uint8_t* outl = in.get<uint8_t>();
        Frame Cur, Last;
        std::vector<MapPoint> mps(nl);
        Cur.N = (int)nc; Cur.mDescriptors = desc_mat((int)nc, cd); Cur.mvpMapPoints.assign(nc, nullptr);
        Last.N = (int)nl; Last.mDescriptors = desc_mat((int)nl, ld); Last.mvpMapPoints.assign(nl, nullptr); Last.mvbOutlier.assign(nl, false);
        for (size_t j = 0; j < nl; j++) { if (has[j]) { mps[j].index = (int)j; Last.mvpMapPoints[j] = &mps[j]; } Last.mvbOutlier[j] = outl[j] != 0; }
        ORBmatcher matcher(0.9f, true);
        const int nm = matcher.MatchORBPoints(Cur, Last);
        std::vector<int32_t> match(nc, -1);
        for (size_t i = 0; i < nc; i++) if (Cur.mvpMapPoints[i]) match[i] = Cur.mvpMapPoints[i]->index;
        out.put(match.data(), match.size()); out.put(&nm, 1);
    } else if (mode == "plane") {
        // prm = {dTh, aTh, verTh, parTh}
        size_t np4, nm4;
        const float* coef = in.get<float>(&np4);
        const float* T = in.get<float>();
        const uint8_t* valid = in.get<uint8_t>();
        const float* mcoef = in.get<float>(&nm4);
        const int32_t* npts = in.get<int32_t>();
        size_t npf;
        const float* pts = in.get<float>(&npf);
        const int NP = (int)(np4 / 4), NM = (int)(nm4 / 4), PS = NM ? (int)(npf / 3 / NM) : 0;
        Frame F;
        F.mnPlaneNum = NP; F.mTcw = mat_f32(4, 4, T);
        F.mvpMapPlanes.assign(NP, nullptr); F.mvpParallelPlanes.assign(NP, nullptr); F.mvpVerticalPlanes.assign(NP, nullptr);
        for (int i = 0; i < NP; i++) F.mvPlaneCoefficients.push_back(mat_f32(4, 1, coef + 4 * i));
        std::vector<MapPlane> planes(NM);
        std::vector<MapPlane*> vp(NM);
        for (int j = 0; j < NM; j++) {
            planes[j].pos = mat_f32(4, 1, mcoef + 4 * j); planes[j].bad = !valid[j]; planes[j].index = j;
            planes[j].mvPlanePoints.reset(new MapPlane::PointCloud());
            for (int k = 0; k < npts[j]; k++) { const float* p = pts + ((size_t)j * PS + k) * 3; planes[j].mvPlanePoints->points.push_back(pcl::PointXYZRGB{p[0], p[1], p[2], 0}); }
            vp[j] = &planes[j];
        }
        PlaneMatcher pm(prm[0], prm[1], prm[2], prm[3]);
        const int nm = pm.SearchMapByCoefficients(F, vp);
        std::vector<int32_t> a(NP, -1), v(NP, -1), p(NP, -1);
        for (int i = 0; i < NP; i++) {
            if (F.mvpMapPlanes[i]) a[i] = F.mvpMapPlanes[i]->index;
            if (F.mvpVerticalPlanes[i]) v[i] = F.mvpVerticalPlanes[i]->index;
            if (F.mvpParallelPlanes[i]) p[i] = F.mvpParallelPlanes[i]->index;
        }
        out.put(a.data(), a.size()); out.put(v.data(), v.size()); out.put(p.data(), p.size()); out.put(&nm, 1);
    } else if (mode == "lsd_proj") {
        // prm = {th, nn_ratio}; LSDmatcher::SearchByProjection(Frame&, const vector<MapLine*>&, th)
        size_t nl, nm;
        const cv::line_descriptor::KeyLine* kl = in.get<cv::line_descriptor::KeyLine>(&nl);
        const uint8_t* ldesc = in.get<uint8_t>();
        const uint8_t* blocked = in.get<uint8_t>();
        const uint8_t* inview = in.get<uint8_t>(&nm);
        const float* proj = in.get<float>();
        const int32_t* lvl = in.get<int32_t>();
        const float* vc = in.get<float>();
        const uint8_t* mdesc = in.get<uint8_t>();
        const uint8_t* obs = in.get<uint8_t>();
        size_t nsf;
        const float* sf = in.get<float>(&nsf);
        Frame F;
        F.NL = (int)nl; F.mvKeylinesUn.assign(kl, kl + nl); F.mLdesc = desc_mat((int)nl, ldesc); F.mvScaleFactors.assign(sf, sf + nsf);
        std::vector<MapLine> dummies(nl), mls(nm);
        F.mvpMapLines.assign(nl, nullptr);
        for (size_t i = 0; i < nl; i++) if (blocked[i]) { dummies[i].nobs = 1; F.mvpMapLines[i] = &dummies[i]; }
        std::vector<MapLine*> vp(nm);
        for (size_t j = 0; j < nm; j++) {
            mls[j].mbTrackInView = inview[j] != 0; mls[j].mTrackProjX1 = proj[4 * j]; mls[j].mTrackProjY1 = proj[4 * j + 1]; mls[j].mTrackProjX2 = proj[4 * j + 2];
            mls[j].mTrackProjY2 = proj[4 * j + 3]; mls[j].mnTrackScaleLevel = lvl[j]; mls[j].mTrackViewCos = vc[j]; mls[j].mLDescriptor = desc_mat(1, mdesc + 32 * j);
            mls[j].nobs = obs[j] ? 1 : 0; mls[j].index = (int)j; vp[j] = &mls[j];
        }
        LSDmatcher matcher(prm[1], true);
        const int n = matcher.SearchByProjection(F, vp, prm[0]);
        std::vector<int32_t> match(nl, -1);
        for (size_t i = 0; i < nl; i++) if (F.mvpMapLines[i]) match[i] = F.mvpMapLines[i]->index;
        out.put(match.data(), match.size()); out.put(&n, 1);
    } else if (mode == "lsd_desc") {
        // LSDmatcher::SearchByDescriptor(KeyFrame*, Frame&, vector<MapLine*>&)
        size_t nk, nc;
        const uint8_t* kd = in.get<uint8_t>(&nk); nk /= 32;
        const uint8_t* cd = in.get<uint8_t>(&nc); nc /= 32;
        const uint8_t* has = in.get<uint8_t>();
        KeyFrame KF;
        Frame F;
        std::vector<MapLine> mls(nk);
        KF.mLineDescriptors = desc_mat((int)nk, kd); KF.mls.assign(nk, nullptr);
        for (size_t i = 0; i < nk; i++) if (has[i]) { mls[i].index = (int)i; KF.mls[i] = &mls[i]; }
        F.NL = (int)nc; F.mLdesc = desc_mat((int)nc, cd);
        LSDmatcher matcher(0.6f, true);
        std::vector<MapLine*> res;
        const int n = matcher.SearchByDescriptor(&KF, F, res);
        std::vector<int32_t> match(nc, -1);
        for (size_t i = 0; i < nc; i++) if (res[i]) match[i] = res[i]->index;
        out.put(match.data(), match.size()); out.put(&n, 1);
    } else if (mode == "fuse") {
        // ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:829-979).  prm = {th, log_scale_factor, n_levels}
        Frame F;
        std::vector<MapPoint> dummies;
        fill_frame(F, in, dummies);
        const float* inv_sigma2 = in.get<float>();
        const float* Tcw = in.get<float>();
        size_t np;
        const uint8_t* usable = in.get<uint8_t>(&np);
        const float *xw = in.get<float>(), *nrm = in.get<float>(), *mind = in.get<float>(), *maxd = in.get<float>();
        const uint8_t* d = in.get<uint8_t>();
        const uint8_t* kf_state = in.get<uint8_t>();     // per keypoint of the key frame: 0 no map point, 1 a map point, 2 a bad map point
        const int32_t* kf_obs = in.get<int32_t>();       // its Observations()
        const int32_t* mp_obs = in.get<int32_t>();
        const int n_levels = (int)prm[2];
        KeyFrame kf;
        kf.N = F.N; kf.mvKeysUn = F.mvKeysUn; kf.mvKeys = F.mvKeys; kf.mvuRight = F.mvuRight; kf.mDescriptors = F.mDescriptors;
        kf.fx = Frame::fx; kf.fy = Frame::fy; kf.cx = Frame::cx; kf.cy = Frame::cy; kf.mbf = F.mbf; kf.mb = F.mb;
        kf.mnMinX = Frame::mnMinX; kf.mnMaxX = Frame::mnMaxX; kf.mnMinY = Frame::mnMinY; kf.mnMaxY = Frame::mnMaxY;
        kf.mfGridElementWidthInv = Frame::mfGridElementWidthInv; kf.mfGridElementHeightInv = Frame::mfGridElementHeightInv;
        kf.mvScaleFactors = F.mvScaleFactors; kf.mvInvLevelSigma2.assign(inv_sigma2, inv_sigma2 + n_levels);
        kf.mfLogScaleFactor = prm[1]; kf.mnScaleLevels = n_levels;
#ifdef STANDINS_REAL_FRAME_FUNCS
        kf.mGrid.resize(kf.mnGridCols);                  // as the KeyFrame constructor does (src/KeyFrame.cc:56-63)
        for (int i = 0; i < kf.mnGridCols; i++) { kf.mGrid[i].resize(kf.mnGridRows); for (int j = 0; j < kf.mnGridRows; j++) kf.mGrid[i][j] = F.mGrid[i][j]; }
        kf.SetPose(mat_f32(4, 4, Tcw));
#else
        kf.Tcw = mat_f32(4, 4, Tcw);                     // the adapter reads GetPose(); the grid is built on the device
#endif
        std::vector<MapPoint> kfmps(F.N);
        kf.mps.assign(F.N, nullptr);
        for (int i = 0; i < F.N; i++) if (kf_state[i]) { kfmps[i].bad = kf_state[i] == 2; kfmps[i].nobs = kf_obs[i]; kfmps[i].index = -2 - i; kfmps[i].kf_slot = i; kf.mps[i] = &kfmps[i]; }
        std::vector<MapPoint> mps(np);
        std::vector<MapPoint*> vp(np, nullptr);
        for (size_t i = 0; i < np; i++) {
            mps[i].pos = mat_f32(3, 1, xw + 3 * i); mps[i].normal = mat_f32(3, 1, nrm + 3 * i); mps[i].desc = desc_mat(1, d + 32 * i);
            mps[i].mfMinDistance = mind[i]; mps[i].mfMaxDistance = maxd[i]; mps[i].nobs = mp_obs[i]; mps[i].index = (int)i;
            // not usable: a NULL entry, a bad point or one the key frame already observes, in turn
            if (usable[i]) vp[i] = &mps[i];
            else if (i % 3 == 1) { mps[i].bad = true; vp[i] = &mps[i]; }
            else if (i % 3 == 2) { mps[i].in_kf = true; vp[i] = &mps[i]; }
        }
        fuse_log().clear();
        ORBmatcher matcher(0.6f, true);
        const int nFused = matcher.Fuse(&kf, vp, prm[0]);
        // which slot each point was paired with: from the edits (AddObservation / Replace); a pairing with a BAD map point of the key frame makes no edit and is
        // read off the GetMapPoint call instead (the reference asks for the descriptor of the point being searched right before)
        std::vector<int32_t> idx(np, -1);
        for (size_t i = 0; i < np; i++) idx[i] = mps[i].fuse_idx;
        for (auto& e : fuse_log()) if (e.first >= 0 && idx[e.first] < 0 && kf.mps[e.second] && kf.mps[e.second]->bad) idx[e.first] = e.second;
        out.put(idx.data(), idx.size()); out.put(&nFused, 1);
    } else if (mode == "lsd_fuse") {
        // LSDmatcher::Fuse(KeyFrame*, const vector<MapLine*>&, th) (src/LSDmatcher.cpp:884-1015).  prm = {th, log_scale_factor}
        size_t nl, nm, nsf;
        const cv::line_descriptor::KeyLine* kl = in.get<cv::line_descriptor::KeyLine>(&nl);
        const uint8_t* ldesc = in.get<uint8_t>();
        const float* intr = in.get<float>();             // min_x, max_x, min_y, max_y, fx, fy, cx, cy, bf
        const float* sf = in.get<float>(&nsf);
        const float* Tcw = in.get<float>();
        const uint8_t* usable = in.get<uint8_t>(&nm);
        const double *xw6 = in.get<double>(), *nrm = in.get<double>();
        const float *mind = in.get<float>(), *maxd = in.get<float>();
        const uint8_t* d = in.get<uint8_t>();
        const uint8_t* kf_state = in.get<uint8_t>();
        const int32_t* kf_obs = in.get<int32_t>();
        const int32_t* ml_obs = in.get<int32_t>();
        KeyFrame kf;
        kf.mvKeyLines.assign(kl, kl + nl); kf.mLineDescriptors = desc_mat((int)nl, ldesc);
        kf.mnMinX = intr[0]; kf.mnMaxX = intr[1]; kf.mnMinY = intr[2]; kf.mnMaxY = intr[3]; kf.fx = intr[4]; kf.fy = intr[5]; kf.cx = intr[6]; kf.cy = intr[7]; kf.mbf = intr[8];
        kf.mvScaleFactors.assign(sf, sf + nsf); kf.mfLogScaleFactor = prm[1]; kf.mnScaleLevels = (int)nsf;
#ifdef STANDINS_REAL_FRAME_FUNCS
        kf.SetPose(mat_f32(4, 4, Tcw));
#else
        kf.Tcw = mat_f32(4, 4, Tcw);
#endif
        std::vector<MapLine> kfmls(nl), mls(nm);
        kf.mls.assign(nl, nullptr);
        for (size_t i = 0; i < nl; i++) if (kf_state[i]) { kfmls[i].bad = kf_state[i] == 2; kfmls[i].nobs = kf_obs[i]; kfmls[i].index = -2 - (int)i; kfmls[i].kf_slot = (int)i; kf.mls[i] = &kfmls[i]; }
        std::vector<MapLine*> vp(nm, nullptr);
        for (size_t i = 0; i < nm; i++) {
            for (int k = 0; k < 6; k++) mls[i].mWorldPos(k) = xw6[6 * i + k];
            for (int k = 0; k < 3; k++) mls[i].normal(k) = nrm[3 * i + k];
            mls[i].mLDescriptor = desc_mat(1, d + 32 * i); mls[i].mfMinDistance = mind[i]; mls[i].mfMaxDistance = maxd[i]; mls[i].nobs = ml_obs[i]; mls[i].index = (int)i;
            if (usable[i]) vp[i] = &mls[i];
            else if (i % 2 == 1) { mls[i].bad = true; vp[i] = &mls[i]; }      // not usable: a NULL entry or a bad line, in turn
        }
        fuse_log().clear();
        LSDmatcher matcher;
        const int nFused = matcher.Fuse(&kf, vp, prm[0]);
        std::vector<int32_t> idx(nm, -1);
        for (size_t i = 0; i < nm; i++) idx[i] = mls[i].fuse_idx;
        for (auto& e : fuse_log()) if (e.first >= 0 && idx[e.first] < 0 && kf.mls[e.second] && kf.mls[e.second]->bad) idx[e.first] = e.second;
        out.put(idx.data(), idx.size()); out.put(&nFused, 1);
    } else { std::fprintf(stderr, "unknown mode\n"); return 2; }
    std::fclose(out.f);
    return 0;
}
