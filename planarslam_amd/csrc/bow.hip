// planarslam_amd/csrc/bow.hip — DBoW2 vocabulary transform for MI355X (gfx950).
//
// Replaces ORBVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) as Frame::ComputeBoW / KeyFrame::ComputeBoW call it
// (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1180 and :1203-1250, FORB::distance FORB.cpp:81-100, BowVector.cpp:29-47, :62-84;
// TF_IDF weighting + L1 norm, the configuration of ORBvoc.txt "10 6 0 0").  The FeatureVector comes out as ONE node id per feature, which is the
// form planar_search_by_bow takes.
//
//   bow_descend   thread = feature: walk the tree, at every level the child with the smallest Hamming distance (first one on ties, strict <),
//                 node table = 8 x u32 per node (35 MB for k = 10, L = 6: L2 / HBM gathers, 60 node reads per feature)
//   bow_vector    workgroup = frame: sort the (word id) keys of the non-stopped features (bitonic, LDS), one thread per distinct word adds its idf
//                 weight tf times in sequence (BowVector::addWeight), one lane sums |v| in ascending word order (BowVector::normalize walks a
//                 std::map), everyone divides.  FP64 in the reference's order: values are bit-identical.
#include <algorithm>
#include <numeric>

#include "common.h"

namespace planar {
namespace bow {

struct VocabDev {
    int L, n_nodes, n_words;
    const int* child_start;       // [n_nodes + 1] CSR over child_ids
    const int* child_ids;
    const uint32_t* desc;         // [n_nodes][8]
    const double* weight;         // [n_nodes]
    const int* word_id;           // [n_nodes], -1 for inner nodes
    const double* word_weight;    // [n_words]
};

__global__ __launch_bounds__(256) void bow_descend(VocabDev V, const uint8_t* __restrict__ desc, const int32_t* __restrict__ n, int stride, int levelsup,
                                                   int32_t* __restrict__ word, double* __restrict__ weight, int32_t* __restrict__ node) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= stride) return;
    const size_t o = (size_t)b * stride + i;
    if (i >= n[b]) { word[o] = -1; weight[o] = 0.0; node[o] = -1; return; }
    const uint4* dp = (const uint4*)(desc + o * 32);
    const uint4 d0 = dp[0], d1 = dp[1];
    const int nid_level = V.L - levelsup;
    int nid = nid_level <= 0 ? 0 : -1, cur = 0, level = 0;
    do {
        ++level;
        const int c0 = V.child_start[cur], c1 = V.child_start[cur + 1];
        int best = -1, best_d = 0x7fffffff;
        for (int c = c0; c < c1; c++) {
            const int id = V.child_ids[c];
            const uint4* q = (const uint4*)(V.desc + (size_t)id * 8);
            const uint4 a = q[0], e = q[1];
            const int d = __popc(a.x ^ d0.x) + __popc(a.y ^ d0.y) + __popc(a.z ^ d0.z) + __popc(a.w ^ d0.w) + __popc(e.x ^ d1.x) + __popc(e.y ^ d1.y) +
                          __popc(e.z ^ d1.z) + __popc(e.w ^ d1.w);
            if (d < best_d) { best_d = d; best = id; }
        }
        cur = best;
        if (level == nid_level) nid = cur;
    } while (V.child_start[cur + 1] > V.child_start[cur]);
    const double w = V.weight[cur];
    word[o] = V.word_id[cur]; weight[o] = w;
    node[o] = w > 0 ? nid : -1;                                // "if (w > 0) // not stopped": only then the feature enters the FeatureVector
}

__global__ __launch_bounds__(256) void bow_vector(VocabDev V, const int32_t* __restrict__ n, int stride, int P, const int32_t* __restrict__ word,
                                                  const double* __restrict__ weight, int32_t* __restrict__ bow_word, double* __restrict__ bow_value,
                                                  int32_t* __restrict__ bow_n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    double* s_val = (double*)smem;                 // [P]
    uint32_t* s_key = (uint32_t*)(s_val + P);      // [P]
    uint16_t* s_pos = (uint16_t*)(s_key + P);      // [P + 1]
    __shared__ int s_cnt[4], s_nvalid;
    __shared__ double s_norm;
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x, nb = n[b];
    for (int i = tid; i < P; i += NT) {
        uint32_t k = 0xffffffffu;
        if (i < stride && i < nb) { const size_t o = (size_t)b * stride + i; if (weight[o] > 0) k = (uint32_t)word[o]; }
        s_key[i] = k;
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int str = size >> 1; str > 0; str >>= 1) {
            for (int t = tid; t < P / 2; t += NT) {
                const int lo = (t / str) * (2 * str) + (t % str), hi = lo + str;
                const bool up = ((lo & size) == 0);
                const uint32_t a = s_key[lo], c = s_key[hi];
                if ((a > c) == up) { s_key[lo] = c; s_key[hi] = a; }
            }
            __syncthreads();
        }
    // heads of runs of equal word ids, compacted in order (block-wide exclusive scan of the head flags, 64-lane ballots + per-wave counts)
    if (tid == 0) s_nvalid = 0;
    __syncthreads();
    int base = 0;
    for (int i0 = 0; i0 < P; i0 += NT) {
        const int i = i0 + tid;
        const uint32_t k = i < P ? s_key[i] : 0xffffffffu;
        const bool valid = k != 0xffffffffu;
        const bool head = valid && (i == 0 || s_key[i - 1] != k);
        const unsigned long long m = __ballot(head);
        const int lane = tid & 63, wave = tid >> 6;
        if (lane == 0) s_cnt[wave] = __popcll(m);
        const unsigned long long mv = __ballot(valid);
        if (lane == 0 && mv) atomicAdd(&s_nvalid, __popcll(mv));
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < (NT >> 6); w++) { if (w < wave) before += s_cnt[w]; total += s_cnt[w]; }
        if (head) s_pos[base + before + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
        base += total;
        __syncthreads();
    }
    const int nu = base, nvalid = s_nvalid;
    if (tid == 0) s_pos[nu] = (uint16_t)nvalid;
    __syncthreads();
    for (int u = tid; u < nu; u += NT) {
        const int c = (int)s_pos[u + 1] - (int)s_pos[u];
        const uint32_t k = s_key[s_pos[u]];
        const double w = V.word_weight[k];
        double v = w;                                           // addWeight: insert (id, w), then += w for every further feature of the word
        for (int j = 1; j < c; j++) v += w;
        s_val[u] = v;
        bow_word[(size_t)b * stride + u] = (int32_t)k;
    }
    __syncthreads();
    if (tid == 0) {
        double norm = 0.0;
        for (int u = 0; u < nu; u++) norm += fabs(s_val[u]);     // std::map iteration = ascending word id
        s_norm = norm;
        bow_n[b] = nu;
    }
    __syncthreads();
    const double norm = s_norm;
    for (int u = tid; u < nu; u += NT) bow_value[(size_t)b * stride + u] = norm > 0.0 ? s_val[u] / norm : s_val[u];
}

}  // namespace bow
}  // namespace planar

struct planar_vocab {
    planar_ctx* ctx = nullptr;
    planar::bow::VocabDev V{};
    int k = 0;
    planar::DevBuf d_child_start, d_child_ids, d_desc, d_weight, d_word_id, d_word_weight;
};

using namespace planar;

extern "C" {

int planar_vocab_create(planar_ctx* ctx, int k, int L, int n, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc, const double* weight,
                        planar_vocab** out) {
    PLANAR_REQUIRE(ctx && parent && is_leaf && desc && weight && out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(k >= 1 && L >= 1 && L <= 10 && n >= 1, PLANAR_EINVAL, "bad vocabulary shape");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    const int N = n + 1;
    std::vector<int> cnt(N + 1, 0), start(N + 1, 0), ids(n), word_id(N, -1);
    for (int i = 1; i <= n; i++) { PLANAR_REQUIRE(parent[i - 1] >= 0 && parent[i - 1] < i, PLANAR_EINVAL, "a node's parent must precede it (file order)"); cnt[parent[i - 1]]++; }
    for (int i = 0; i < N; i++) start[i + 1] = start[i] + cnt[i];
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int i = 1; i <= n; i++) ids[fill[parent[i - 1]]++] = i;                      // children in file order, as loadFromTextFile pushes them
    int n_words = 0;
    std::vector<double> w(N, 0.0), ww;
    std::vector<uint32_t> dd((size_t)N * 8, 0);
    for (int i = 1; i <= n; i++) {
        const bool leaf = is_leaf[i - 1] != 0;
        PLANAR_REQUIRE(leaf == (cnt[i] == 0), PLANAR_EINVAL, "isLeaf flag and children disagree");
        w[i] = weight[i - 1];
        std::memcpy(&dd[(size_t)i * 8], desc + (size_t)(i - 1) * 32, 32);
        if (leaf) { word_id[i] = n_words++; ww.push_back(weight[i - 1]); }
    }
    PLANAR_REQUIRE(cnt[0] > 0, PLANAR_EINVAL, "the root has no children");
    planar_vocab* v = new planar_vocab;
    v->ctx = ctx; v->k = k;
    int rc;
    if ((rc = v->d_child_start.alloc(start.size() * 4)) || (rc = v->d_child_ids.alloc(ids.size() * 4)) || (rc = v->d_desc.alloc(dd.size() * 4)) ||
        (rc = v->d_weight.alloc(w.size() * 8)) || (rc = v->d_word_id.alloc(word_id.size() * 4)) || (rc = v->d_word_weight.alloc(std::max<size_t>(ww.size(), 1) * 8))) { delete v; return rc; }
    hipError_t e = hipMemcpy(v->d_child_start.p, start.data(), start.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_child_ids.p, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_desc.p, dd.data(), dd.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_weight.p, w.data(), w.size() * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_word_id.p, word_id.data(), word_id.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && !ww.empty()) e = hipMemcpy(v->d_word_weight.p, ww.data(), ww.size() * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error("vocabulary upload failed: %s", hipGetErrorString(e)); delete v; return PLANAR_EDEVICE; }
    v->V = bow::VocabDev{L, N, n_words, v->d_child_start.as<int>(), v->d_child_ids.as<int>(), v->d_desc.as<uint32_t>(), v->d_weight.as<double>(), v->d_word_id.as<int>(),
                         v->d_word_weight.as<double>()};
    *out = v;
    return PLANAR_OK;
}

void planar_vocab_destroy(planar_vocab* v) { delete v; }
int planar_vocab_words(const planar_vocab* v) { return v ? v->V.n_words : PLANAR_EINVAL; }

int planar_bow_transform_dev(planar_vocab* v, const uint8_t* d_desc, const int32_t* d_n, int B, int stride, int levelsup, int32_t* d_word, double* d_weight,
                             int32_t* d_node, int32_t* d_bow_word, double* d_bow_value, int32_t* d_bow_n) {
    PLANAR_REQUIRE(v && d_desc && d_n && d_word && d_weight && d_node, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && stride >= 1 && stride <= PLANAR_MAX_FRAME_KEYS, PLANAR_EINVAL, "bad size");
    PLANAR_REQUIRE(!d_bow_word == !d_bow_value && !d_bow_word == !d_bow_n, PLANAR_EINVAL, "the three BowVector outputs come together");
    hipStream_t st = v->ctx->stream;
    hipLaunchKernelGGL(bow::bow_descend, dim3((stride + 255) / 256, B), dim3(256), 0, st, v->V, d_desc, d_n, stride, levelsup, d_word, d_weight, d_node);
    if (d_bow_word) {
        int P = 2;
        while (P < stride) P <<= 1;
        const size_t smem = (size_t)P * 12 + (size_t)(P + 1) * 2 + 16;
        if (smem > 48 * 1024) PLANAR_HIP_CHECK(hipFuncSetAttribute((const void*)bow::bow_vector, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(bow::bow_vector, dim3(B), dim3(256), smem, st, v->V, d_n, stride, P, d_word, d_weight, d_bow_word, d_bow_value, d_bow_n);
    }
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_bow_transform(planar_vocab* v, const uint8_t* desc, const int32_t* n, int B, int stride, int levelsup, int32_t* word, double* weight, int32_t* node,
                         int32_t* bow_word, double* bow_value, int32_t* bow_n) {
    PLANAR_REQUIRE(v && desc && n && word && weight && node && bow_word && bow_value && bow_n, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && stride >= 1 && stride <= PLANAR_MAX_FRAME_KEYS, PLANAR_EINVAL, "bad size");
    PLANAR_HIP_CHECK(hipSetDevice(v->ctx->device));
    hipStream_t st = v->ctx->stream;
    const size_t bs = (size_t)B * stride;
    Stager S;
    const int i_desc = S.in(desc, bs * 32), i_n = S.in(n, (size_t)B * 4), i_word = S.out(word, bs * 4), i_w = S.out(weight, bs * 8), i_node = S.out(node, bs * 4),
              i_bw = S.out(bow_word, bs * 4), i_bv = S.out(bow_value, bs * 8), i_bn = S.out(bow_n, (size_t)B * 4);
    int rc = S.upload(st);
    if (rc) return rc;
    if ((rc = planar_bow_transform_dev(v, S.dev<uint8_t>(i_desc), S.dev<int32_t>(i_n), B, stride, levelsup, S.dev<int32_t>(i_word), S.dev<double>(i_w), S.dev<int32_t>(i_node),
                                       S.dev<int32_t>(i_bw), S.dev<double>(i_bv), S.dev<int32_t>(i_bn))))
        return rc;
    return S.download(st);
}

}  // extern "C"
