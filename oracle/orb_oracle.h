// oracle/orb_oracle.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
// CPU restatement of PlanarSLAM's ORB extractor (reference src/ORBextractor.cc).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace orc {

// Same 28-byte layout as cv::KeyPoint.
struct KeyPoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

struct Candidate { int x, y, score; };   // FAST survivor, coords relative to (minBorderX,minBorderY)

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px;   // row-major, step == w
    const uint8_t* row(int y) const { return px.data() + (size_t)y * w; }
};

class OrbOracle {
public:
    // reference src/ORBextractor.cc:410-470
    OrbOracle(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);

    // reference src/ORBextractor.cc:1043-1105 (operator()); returns number of keypoints.
    int extract(const uint8_t* gray, int W, int H, int pitch,
                std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc);

    // stage outputs of the last extract() (for stage-by-stage parity tests)
    std::vector<Image> pyramid;                        // borderless levels (a2)
    std::vector<Image> blurred;                        // a7, only levels with keypoints
    std::vector<std::vector<Candidate>> candidates;    // a3/a4, reference emission order
    std::vector<std::vector<KeyPoint>> level_kps;      // after octree + orientation (level coords)

    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> features_per_level, umax;
    int nfeatures, nlevels, ini_th, min_th;
    double scale_factor;

    // pieces, exposed for unit tests
    void compute_pyramid(const uint8_t* gray, int W, int H, int pitch);
    void detect_level(int level, std::vector<Candidate>& cand) const;
    // reference :539-763; deterministic tie-break documented in orb_oracle.cpp
    std::vector<Candidate> distribute_octree(const std::vector<Candidate>& keys, int minX, int maxX,
                                             int minY, int maxY, int N) const;
    float ic_angle(const Image& img, int x, int y) const;
    void brief(const Image& blurred, const KeyPoint& kp, uint8_t* desc32) const;
};

extern const int8_t kBriefPattern[1024];

}  // namespace orc
