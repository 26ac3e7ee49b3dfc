// tests/host_shim/isort_host.cpp — TEST INFRASTRUCTURE: the product's std::sort-arrangement engine (planarslam_amd/csrc/isort.h, the same source
// hipcc compiles for gfx950) compiled for the host and run on the wave64 emulator of wave_emul.h, next to the REAL std::sort of this libstdc++ with a
// comparator that looks at the key only (what pcl::VoxelGrid and OpenCV's LSD call).  tests/test_isort_emul.py compares the two word for word.
#include "wave_emul.h"

#include "../../planarslam_amd/csrc/isort.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace planar::isort;

namespace {
struct GArgs { uint32_t* arr; const Range* init; int n_init, n_stage, nr_cap; Range* ranges; Block* blocks; int max_blocks; int* counts; int rows_cap; int* status; int shift; HeapSink HS; uint32_t skip_key;
               int rows_long; uint32_t* gpos; int gpos_half; };
template <int SHIFT, int T>
void g_entry(void* p) {
    auto* A = (GArgs*)p;
    PLANAR_DYN_SMEM(lds);
    global_tier<SHIFT, T>(A->arr, A->init, A->n_init, A->n_stage, A->nr_cap, A->ranges, A->blocks, A->max_blocks, A->counts, lds, A->rows_cap, A->HS, A->status, A->skip_key,
                          A->rows_long, A->gpos, A->gpos_half);
}
struct LArgs { uint32_t* arr; const Range* ranges; int nr, f, l; int* status; HeapSink HS; uint32_t skip_key; };
struct HArgs { uint32_t* arr; const HeapJob* jobs; int njobs, cap; };
template <int SHIFT>
void h_entry(void* p) {
    auto* A = (HArgs*)p;
    PLANAR_DYN_SMEM(lds);
    heap_jobs<SHIFT>(A->arr, A->jobs, A->njobs, 0, 1, (uint32_t*)lds, A->cap, 0, 1 << 30);
}
template <int SHIFT, int T, int E>
void l_entry(void* p) {
    auto* A = (LArgs*)p;
    PLANAR_DYN_SMEM(lds);
    lds_tier<SHIFT, T, E>(A->arr, A->ranges, A->nr, A->f, A->l, lds, A->HS, A->status, A->skip_key);
}
template <int SHIFT, int TG, int T, int E>
int run(uint32_t* arr, const int* bounds, int n_ranges, int n_stage, int* status, long* stats, int heap_cap = 36864, uint32_t skip_key = 0xffffffffu, int force_rows_cap = 0) {
    std::vector<Range> init(n_ranges);
    int longest = 0;
    for (int i = 0; i < n_ranges; i++) {
        const int n = bounds[i + 1] - bounds[i];
        int lg = 0; for (int t = n; t > 1; t >>= 1) lg++;
        init[i] = Range{bounds[i], bounds[i + 1], 2 * lg};
        longest = std::max(longest, n);
    }
    if (n_stage <= 0 || n_stage > T * E) n_stage = T * E;
    // the product's sizing (GlobalLayout::plan over the whole array), or - force_rows_cap - bitmaps for short ranges only, so that every longer range takes wg_partition_long
    const int total = bounds[n_ranges] - bounds[0];
    int rows_cap = 0, rows_long = 0;
    if (force_rows_cap > 0) {
        rows_cap = force_rows_cap; rows_long = GlobalLayout<TG>::rows_for(total);
        if (GlobalLayout<TG>::off_q(rows_cap) < GlobalLayout<TG>::long_bytes(rows_long)) throw std::runtime_error("forced rows_cap too small for the long path's prefixes");
    } else if (!GlobalLayout<TG>::plan(longest, rows_cap, rows_long)) throw std::runtime_error("array too large for the global tier");
    std::vector<uint32_t> gpos((size_t)2 * (total / 2 + 1) + 2);
    std::vector<Range> ranges(G_FMAX);
    std::vector<Block> blocks(G_FMAX);
    int counts[2] = {0, 0};
    std::vector<HeapJob> jobs(4096);
    int njobs = 0;
    const HeapSink HS{jobs.data(), &njobs, (int)jobs.size()};
    GArgs ga{arr, init.data(), n_ranges, n_stage, std::min(T, 64), ranges.data(), blocks.data(), G_FMAX, counts, rows_cap, status, SHIFT, HS, skip_key, rows_long, gpos.data(), total / 2 + 1};
    wave_emul::Dim3 bi, bd; bd.x = TG; bd.y = 1; bd.z = 1;
    wave_emul::launch_block(g_entry<SHIFT, TG>, &ga, TG, bi, bd, (size_t)GlobalLayout<TG>::bytes(rows_cap), 256 * 1024);
    if (stats) { stats[0] = counts[0]; stats[1] = counts[1]; stats[2] = wave_emul::S().n_sync; }
    bd.x = T;
    for (int b = 0; b < counts[1]; b++) {
        LArgs la{arr, ranges.data() + blocks[b].r0, blocks[b].nr, blocks[b].f, blocks[b].l, status, HS, skip_key};
        wave_emul::launch_block(l_entry<SHIFT, T, E>, &la, T, bi, bd, (size_t)LdsLayout<T, E>::bytes, 256 * 1024);
    }
    long heap_elems = 0;
    for (int j = 0; j < std::min(njobs, (int)jobs.size()); j++) heap_elems += jobs[j].l - jobs[j].f;
    if (njobs) {   // the fallback jobs: one wavefront; heap_cap words of the range in "LDS", the rest addressed in the array
        HArgs ha{arr, jobs.data(), std::min(njobs, (int)jobs.size()), heap_cap};
        bd.x = 64;
        wave_emul::launch_block(h_entry<SHIFT>, &ha, 64, bi, bd, (size_t)heap_cap * 4, 256 * 1024);
    }
    if (stats) { stats[3] = wave_emul::S().n_sync; stats[4] = g_levels; stats[5] = g_segs; stats[6] = heap_elems; stats[7] = njobs; g_levels = 0; g_segs = 0; }
    if (getenv("ISORT_STATS")) fprintf(stderr, "isort: wavefront-scope passes %ld, elements per lane and pass %.2f; workgroup-scope passes %ld, elements per lane and pass %.2f; all-equal segments in closed form %ld (%ld elements)\n", g_wlev, g_wlev ? (double)g_wlanes / g_wlev : 0.0, g_glev, g_glev ? (double)g_glanes / g_glev : 0.0, g_eqseg, g_eqelem);
    g_wlev = g_wlanes = g_glev = g_glanes = g_eqseg = g_eqelem = 0;
    return 0;
}
}  // namespace

extern "C" {
// arr [total]: in/out.  bounds [n_ranges + 1]: every [bounds[i], bounds[i+1]) is sorted on its own, as std::sort(first, last, key <) would.
// config 0: the product's shapes (global tier 1024 threads; LDS tier 256 threads x 23 elements: planepost.hip PS_T / PS_LT / PS_E, lsd.hip SORT_*); 4: 512 x 23; 5: the product's shapes with stop bitmaps for ranges of <= 16 384 words only (every longer range goes through wg_partition_long, the path of frames beyond ~390 000 words); 1: small shapes that force many levels and the
// global tier on short arrays (256 threads; 256 x 5).  shift: 19 or 20.  n_stage: LDS-tier capacity override (0 = the configuration's).
// Returns 0, or -1 with a message in err.  stats [8] (last two: elements that went through the heap-sort fallback, its jobs): ranges, blocks, rendezvous count after the global tier, after everything, LDS-tier levels, segments partitioned there.
int isort_emul(uint32_t* arr, const int* bounds, int n_ranges, int shift, int config, int n_stage, int* status, long* stats, char* err, int errlen) {
    try {
        *status = 0;
        if (shift == 19 && config == 0) return run<19, 1024, 256, 23>(arr, bounds, n_ranges, n_stage, status, stats);
        if (shift == 20 && config == 0) return run<20, 1024, 256, 23>(arr, bounds, n_ranges, n_stage, status, stats);
        if (shift == 19 && config == 1) return run<19, 256, 256, 5>(arr, bounds, n_ranges, n_stage, status, stats);
        if (shift == 20 && config == 1) return run<20, 256, 256, 5>(arr, bounds, n_ranges, n_stage, status, stats);
        if (shift == 19 && config == 2) return run<19, 256, 128, 32>(arr, bounds, n_ranges, n_stage, status, stats);
        if (shift == 19 && config == 4) return run<19, 1024, 512, 23>(arr, bounds, n_ranges, n_stage, status, stats);                  // eight wavefronts per LDS block
        if (shift == 19 && config == 5) return run<19, 1024, 256, 23>(arr, bounds, n_ranges, n_stage, status, stats, 36864, 0xffffffffu, 256);  // bitmaps for <= 16 384 words only: longer ranges through wg_partition_long
        if (shift == 20 && config == 5) return run<20, 1024, 256, 23>(arr, bounds, n_ranges, n_stage, status, stats, 36864, 0xffffffffu, 256);
        if (shift == 19 && config == 3) return run<19, 1024, 256, 23>(arr, bounds, n_ranges, n_stage, status, stats, 1000);   // fallback jobs with only 1000 words in LDS
        throw std::runtime_error("isort_emul: unknown configuration");
    } catch (const std::exception& e) { snprintf(err, errlen, "%s", e.what()); return -1; }
}

// the same with a skip key (isort.h, global_tier): only the elements whose key is <= skip_key have to end where std::sort puts them.  config 0 or 1 as above.
int isort_emul_skip(uint32_t* arr, const int* bounds, int n_ranges, int shift, int config, unsigned skip_key, int* status, long* stats, char* err, int errlen) {
    try {
        *status = 0;
        if (shift == 20 && config == 0) return run<20, 1024, 256, 23>(arr, bounds, n_ranges, 0, status, stats, 36864, skip_key);
        if (shift == 20 && config == 1) return run<20, 256, 256, 5>(arr, bounds, n_ranges, 0, status, stats, 36864, skip_key);
        throw std::runtime_error("isort_emul_skip: unknown configuration");
    } catch (const std::exception& e) { snprintf(err, errlen, "%s", e.what()); return -1; }
}

// the real thing: std::sort with a comparator on the key only
void isort_std_sort(uint32_t* arr, const int* bounds, int n_ranges, int shift) {
    for (int i = 0; i < n_ranges; i++)
        std::sort(arr + bounds[i], arr + bounds[i + 1], [shift](uint32_t a, uint32_t b) { return (a >> shift) < (b >> shift); });
}
}
