// oracle/ref_bow_main.cpp — TEST INFRASTRUCTURE.  Driver for the REAL DBoW2 vocabulary: Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h, FORB.cpp, BowVector.cpp,
// FeatureVector.cpp, ScoringObject.cpp and DUtils/Random.cpp, Timestamp.cpp compiled where they lie (never copied) against the OpenCV stand-in into
// oracle/_ref/ref_bow.
//   ref_bow <vocabulary.txt> <in.bin> <out.bin>
// vocabulary.txt: the text format TemplatedVocabulary::loadFromTextFile reads (ORBvoc.txt's format; here a synthetic tree written by tests/bow_cases.py).
// in.bin : int32 levelsup, int32 n, n x 32 descriptor bytes.      (Frame::ComputeBoW: transform(vCurrentDesc, mBowVec, mFeatVec, 4))
// out.bin: int32 nwords, nwords x {uint32 word id, double value}  (mBowVec in map order), then n x int32 node id of each feature (-1: in no node),
//          n x uint32 word id and n x double weight from the per-feature transform(feature, id, weight, &nid, levelsup).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "Thirdparty/DBoW2/DBoW2/FORB.h"
#include "Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h"

typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> ORBVocabularyBase;   // include/ORBVocabulary.h:31
struct ORBVocabulary : ORBVocabularyBase { using ORBVocabularyBase::transform; };                // the per-feature overload is protected

int main(int argc, char** argv) {
    if (argc != 4) { std::fprintf(stderr, "usage: ref_bow <vocabulary.txt> <in.bin> <out.bin>\n"); return 2; }
    ORBVocabulary voc;
    if (!voc.loadFromTextFile(argv[1])) return 3;
    FILE* fi = std::fopen(argv[2], "rb");
    if (!fi) return 2;
    int32_t hdr[2];
    if (std::fread(hdr, 4, 2, fi) != 2) return 2;
    const int levelsup = hdr[0], n = hdr[1];
    std::vector<uint8_t> raw((size_t)n * 32);
    if (n && std::fread(raw.data(), 32, n, fi) != (size_t)n) return 2;
    std::fclose(fi);
    std::vector<cv::Mat> feats(n);
    for (int i = 0; i < n; i++) { feats[i] = cv::Mat(1, 32, CV_8U); std::memcpy(feats[i].data, &raw[(size_t)i * 32], 32); }
    DBoW2::BowVector v;
    DBoW2::FeatureVector fv;
    voc.transform(feats, v, fv, levelsup);
    FILE* fo = std::fopen(argv[3], "wb");
    if (!fo) return 2;
    const int32_t nw = (int32_t)v.size();
    std::fwrite(&nw, 4, 1, fo);
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it) { const uint32_t w = it->first; const double x = it->second; std::fwrite(&w, 4, 1, fo); std::fwrite(&x, 8, 1, fo); }
    std::vector<int32_t> node(n, -1);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it)
        for (unsigned int f : it->second) node[f] = (int32_t)it->first;
    std::fwrite(node.data(), 4, n, fo);
    std::vector<uint32_t> wid(n); std::vector<double> wt(n);
    for (int i = 0; i < n; i++) { DBoW2::WordId id; DBoW2::WordValue w; DBoW2::NodeId nid; voc.transform(feats[i], id, w, &nid, levelsup); wid[i] = id; wt[i] = w; }
    std::fwrite(wid.data(), 4, n, fo); std::fwrite(wt.data(), 8, n, fo);
    std::fclose(fo);
    return 0;
}
