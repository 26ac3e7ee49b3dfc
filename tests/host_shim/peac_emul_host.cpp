// tests/host_shim/peac_emul_host.cpp — TEST INFRASTRUCTURE: compiles the product's PEAC block kernel and clustering kernel (planarslam_amd/csrc/
// peac_common.h, peac_ahc2.h - the same source hipcc compiles for gfx950) for the host and runs them on the wave64 emulator of wave_emul.h.
// tests/test_peac_emul.py compares the clustering state they leave in the frame workspace with the oracle's (orc_peac_cluster_state).
#include "wave_emul.h"

#include "../../planarslam_amd/csrc/peac_ahc2.h"

#include <cstdio>

using namespace planar::peac;

namespace {
struct BlocksArgs { Layout L; Intr K; const uint16_t* depth; int pitch; int64_t stride; uint8_t* ws; };
void blocks_entry(void* a) { auto* A = (BlocksArgs*)a; peac_blocks(A->L, A->K, A->depth, A->pitch, A->stride, A->ws); }
struct AhcArgs { Layout L; Consts C; uint8_t* ws; int32_t* status; long long* timing; int* next; int retry_inline; };
void ahc_exact_entry(void* a) { auto* A = (AhcArgs*)a; peac_ahc2(A->L, A->C, A->ws, A->status, A->timing, A->next, nullptr, 0); }
void ahc_fast_entry(void* a) { auto* A = (AhcArgs*)a; peac_ahc3(A->L, A->C, A->ws, A->status, A->timing, A->next, nullptr, A->retry_inline); }
}  // namespace

extern "C" {
// Runs peac_blocks + peac_ahc2 for one frame on the emulator.  Outputs (sizes from peac_emul_dims): nodes [NB2][18] = N, rid, mse, center[3], normal[3],
// stats[9]; hand [4 + 128] = n_ext, err, n_nodes, -, extracted ids; dsp / dss [NB] as the kernel leaves them; nouse [(NB2 + 31) / 32].
// Returns 0, or -1 with a message in err.
int peac_emul_dims(int W, int H, int* NB, int* NB2) { const Layout L = make_layout(W, H); *NB = L.NB; *NB2 = L.NB2; return 0; }
// mode bit 2 (| 4): every pruned evaluation of a pooled bag is checked against the in-order evaluation of all its candidates (kernel status 9).
// mode 0: what the library does (the fast kernel, then the exact kernel if it left ST_RETRY); 1: exact kernel only; 2: fast kernel only.
// stats_out: [0] status, [1] rendezvous count, [2] phases << 40 | nodes evaluated << 20 | valid-record pops, [3] pops of pool bags, [4] 1 if the fast kernel gave up, [5] eigen-solves spent on pool bags
int peac_emul_cluster(const uint16_t* depth, int W, int H, float fx, float fy, float cx, float cy, float factor, int mode, double* nodes, int32_t* hand,
                      uint16_t* dsp, uint16_t* dss, uint32_t* nouse, int64_t* stats_out, char* err, int errlen) {
    try {
        planar::peac::g_peac_check_prune = (mode & 4) ? 1 : 0; mode &= 3;
        const Layout L = make_layout(W, H);
        const Consts C = make_consts();
        if (L.NB2 > 65535 || ahc2_smem_bytes(L) > 160 * 1024 - 2048) throw std::runtime_error("image too large for the clustering kernel");
        std::vector<uint8_t> ws(L.frame_bytes, 0xEE);   // poisoned: the kernels must initialise what they read
        const Intr K{fx, fy, cx, cy, factor};
        BlocksArgs ba{L, K, depth, W, (int64_t)W * H, ws.data()};
        for (int bx = 0; bx < (L.NB + 63) / 64; bx++) {
            wave_emul::Dim3 bi; bi.x = bx; bi.y = 0;
            wave_emul::Dim3 bd; bd.x = 64; bd.y = 1; bd.z = 1;
            wave_emul::launch_block(blocks_entry, &ba, 64, bi, bd, 0);
        }
        int32_t status = -1; int next = 0;
        long long timing[TSLOTS] = {0};
        AhcArgs aa{L, C, ws.data(), &status, timing, &next, mode == 0 ? 1 : 0};
        wave_emul::Dim3 bi, bd; bd.x = 64; bd.y = 1; bd.z = 1;
        bool retried = false;
        if (mode != 1) {   // mode 0: the library's launch (a frame the fast attempt gives up on is redone by the same workgroup); mode 2: the fast attempt alone
            wave_emul::launch_block(ahc_fast_entry, &aa, 64, bi, bd, (size_t)ahc2_smem_bytes(L));
            retried = status == ST_RETRY || timing[12] != 0;
        } else { next = 0; wave_emul::launch_block(ahc_exact_entry, &aa, 64, bi, bd, (size_t)ahc2_smem_bytes(L)); }
        const uint8_t* F = ws.data();
        const double* st = (const double*)(F + L.off_stats); const double* ge = (const double*)(F + L.off_geo);
        const int* N = (const int*)(F + L.off_N); const uint16_t* rid = (const uint16_t*)(F + L.off_h_rid);
        const int* g_hand = (const int*)(F + L.off_h_hand);
        const int n_nodes = g_hand[2];
        for (int i = 0; i < L.NB2; i++) {
            double* o = nodes + (size_t)i * 18;
            if (i >= n_nodes) { for (int k = 0; k < 18; k++) o[k] = 0; continue; }
            o[0] = N[i]; o[1] = rid[i]; o[2] = ge[i * 7 + 6];
            for (int k = 0; k < 6; k++) o[3 + k] = ge[i * 7 + k];
            for (int k = 0; k < 9; k++) o[9 + k] = st[i * 9 + k];
        }
        memcpy(hand, g_hand, (4 + MAX_PLANES) * 4);
        memcpy(dsp, F + L.off_h_dsp, (size_t)L.NB * 2); memcpy(dss, F + L.off_h_dss, (size_t)L.NB * 2);
        memcpy(nouse, F + L.off_h_nouse, (size_t)((L.NB2 + 31) / 32) * 4);
        if (stats_out) { stats_out[0] = status; stats_out[1] = wave_emul::S().n_sync; stats_out[2] = timing[7]; stats_out[3] = timing[10]; stats_out[4] = retried; stats_out[5] = timing[11]; }
        return 0;
    } catch (const std::exception& e) {
        snprintf(err, errlen, "%s", e.what());
        return -1;
    }
}
}
