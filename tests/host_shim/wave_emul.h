// tests/host_shim/wave_emul.h — TEST INFRASTRUCTURE: a host-side emulator for single-workgroup HIP kernels written in the wave64 style of
// planarslam_amd/csrc (ballots, shuffles, LDS, workgroup barriers), so that the SAME kernel source can be compiled with g++ and run where there is
// no GPU.  Include this header BEFORE the kernel header.
//
// Model: one fiber (ucontext) per thread of the workgroup, run cooperatively on one OS thread.  A fiber runs alone until it reaches a cross-lane
// operation (__ballot, __shfl*, readlane), a wavefront fence (__builtin_amdgcn_wave_barrier, __threadfence_block) or a workgroup barrier, where it
// waits for the other live threads of its wavefront / workgroup.  That is the GPU's lockstep at the granularity the kernels rely on: LDS / global
// accesses of one wavefront between two such points are seen by the other lanes after the next point.  Threads that have returned from the kernel no
// longer take part (as exited lanes on the GPU).  A thread that waits while every other thread of its group has exited or waits at a DIFFERENT kind
// of point is reported as a divergence error (cross-lane operations must be reached by the whole wavefront in this code base).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <ucontext.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define PLANAR_DYN_SMEM(name) uint8_t* name = ::wave_emul::S().dyn_smem.data()
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() ::wave_emul::sync_wave(__LINE__)
#define __threadfence_block() ::wave_emul::sync_wave(__LINE__)
#define __syncthreads() ::wave_emul::sync_block(__LINE__)
#define __builtin_amdgcn_readlane(v, k) ::wave_emul::shfl(__LINE__, (v), (k))
// cross-lane operations carry their source line: every thread of a rendezvous must have come from the same one
#define __ballot(p) ::wave_emul::ballot(__LINE__, (p))
#define __shfl(v, s) ::wave_emul::shfl(__LINE__, (v), (s))
#define __shfl_up(v, d) ::wave_emul::shfl(__LINE__, (v), ((::wave_emul::S().cur & 63) - (d) >= 0 ? (::wave_emul::S().cur & 63) - (d) : (::wave_emul::S().cur & 63)))
#define __shfl_down(v, d) ::wave_emul::shfl(__LINE__, (v), ((::wave_emul::S().cur & 63) + (d) < 64 ? (::wave_emul::S().cur & 63) + (d) : (::wave_emul::S().cur & 63)))
#define __shfl_xor(v, m) ::wave_emul::shfl(__LINE__, (v), (::wave_emul::S().cur & 63) ^ (m))
#define threadIdx (::wave_emul::S().tid3())
#define blockIdx (::wave_emul::S().block_idx)
#define blockDim (::wave_emul::S().block_dim)

namespace wave_emul {

struct Dim3 { int x = 0, y = 0, z = 0; };

struct State {
    static constexpr int MAXT = 1024;
    ucontext_t main_ctx;
    std::vector<ucontext_t> ctx;
    std::vector<std::vector<char>> stacks;
    std::vector<uint8_t> dyn_smem;
    void (*entry)(void*) = nullptr;
    void* entry_arg = nullptr;
    int nthreads = 0, cur = 0;
    std::vector<char> finished;
    // per-wavefront and per-workgroup rendezvous
    struct Group { int arrived = 0, live = 0, tag = 0; long generation = 0; };
    std::vector<Group> wave;
    Group block;
    uint64_t slot[MAXT];
    Dim3 block_idx, block_dim;
    std::string error;
    long n_sync = 0;
    Dim3 tid3() const { Dim3 d; d.x = cur; return d; }
};
inline State& S() { static State s; return s; }

inline void switch_to_next() {   // round robin over the unfinished fibers; back to the launcher when none is left
    State& s = S();
    const int me = s.cur;
    for (int k = 1; k <= s.nthreads; k++) {
        const int nx = (me + k) % s.nthreads;
        if (!s.finished[nx]) {
            if (nx == me) return;
            s.cur = nx;
            swapcontext(&s.ctx[me], &s.ctx[nx]);
            return;
        }
    }
    swapcontext(&s.ctx[me], &s.main_ctx);
}

inline void fail(const std::string& msg) {   // abandon the launch: back to launch_block, which throws
    State& s = S();
    if (s.error.empty()) s.error = msg;
    swapcontext(&s.ctx[s.cur], &s.main_ctx);
}
inline void rendezvous(State::Group& g, int tag, const char* what) {
    State& s = S();
    s.n_sync++;
    const long gen = g.generation;
    if (g.arrived == 0) g.tag = tag;
    else if (g.tag != tag)
        fail(std::string("wave_emul: divergent ") + what + ": thread " + std::to_string(s.cur) + " is at source line " + std::to_string(tag >> 2) + ", another thread of the group at line " + std::to_string(g.tag >> 2));
    if (++g.arrived >= g.live) { g.arrived = 0; g.generation++; return; }
    long spins = 0;
    while (g.generation == gen) {
        switch_to_next();
        if (++spins > 200000L) fail(std::string("wave_emul: thread ") + std::to_string(s.cur) + " stuck at " + what + ", source line " + std::to_string(tag >> 2) + " (the other threads never arrive)");
    }
}
inline void sync_wave(int line, int phase = 0) { rendezvous(S().wave[S().cur >> 6], line * 4 + phase, "wavefront operation"); }
inline void sync_block(int line) { rendezvous(S().block, line * 4 + 3, "__syncthreads"); }

template <typename T>
inline T shfl(int line, T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shfl of a type wider than 8 bytes");
    State& s = S();
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    s.slot[s.cur] = raw;
    sync_wave(line, 1);
    const int base = s.cur & ~63;
    int src = base + (src_lane & 63);
    if (src >= s.nthreads) src = s.cur;
    const uint64_t got = s.slot[src];
    sync_wave(line, 2);
    T r;
    memcpy(&r, &got, sizeof(T));
    return r;
}
inline unsigned long long ballot(int line, bool pred) {
    State& s = S();
    s.slot[s.cur] = pred ? 1 : 0;
    sync_wave(line, 1);
    const int base = s.cur & ~63;
    unsigned long long m = 0;
    for (int l = 0; l < 64 && base + l < s.nthreads; l++)
        if (!s.finished[base + l] && s.slot[base + l]) m |= 1ull << l;
    sync_wave(line, 2);
    return m;
}

inline void fiber_main() {
    State& s = S();
    s.entry(s.entry_arg);
    // this thread leaves the kernel: it no longer takes part in any rendezvous
    const int me = s.cur;
    s.finished[me] = 1;
    State::Group& w = s.wave[me >> 6];
    w.live--; s.block.live--;
    if (w.live > 0 && w.arrived >= w.live) { w.arrived = 0; w.generation++; }
    if (s.block.live > 0 && s.block.arrived >= s.block.live) { s.block.arrived = 0; s.block.generation++; }
    switch_to_next();
}

// Runs one workgroup of `nthreads` threads of kernel body `fn` (which reads threadIdx / blockIdx / the dynamic LDS block through the macros above).
inline void launch_block(void (*fn)(void*), void* arg, int nthreads, Dim3 bidx, Dim3 bdim, size_t dyn_smem_bytes, size_t stack_bytes = 512 * 1024) {
    State& s = S();
    if (nthreads > State::MAXT) throw std::runtime_error("wave_emul: too many threads");
    s.entry = fn; s.entry_arg = arg; s.nthreads = nthreads; s.block_idx = bidx; s.block_dim = bdim; s.error.clear();
    s.dyn_smem.assign(dyn_smem_bytes + 64, 0xCD);
    s.ctx.resize(nthreads); s.stacks.resize(nthreads); s.finished.assign(nthreads, 0);
    s.wave.assign((nthreads + 63) / 64, State::Group());
    for (int t = 0; t < nthreads; t++) s.wave[t >> 6].live++;
    s.block = State::Group(); s.block.live = nthreads;
    for (int t = 0; t < nthreads; t++) {
        s.stacks[t].resize(stack_bytes);
        getcontext(&s.ctx[t]);
        s.ctx[t].uc_stack.ss_sp = s.stacks[t].data();
        s.ctx[t].uc_stack.ss_size = stack_bytes;
        s.ctx[t].uc_link = &s.main_ctx;
        makecontext(&s.ctx[t], (void (*)())fiber_main, 0);
    }
    s.cur = 0;
    swapcontext(&s.main_ctx, &s.ctx[0]);
    if (!s.error.empty()) throw std::runtime_error(s.error);
    for (int t = 0; t < nthreads; t++) if (!s.finished[t]) throw std::runtime_error("wave_emul: kernel ended with unfinished threads");
}

}  // namespace wave_emul

// ---- planarslam_amd/csrc/wave_ops.h for the emulator ----
#define PLANAR_WAVE_EMUL 1
#define PLANAR_DPP_F64(v, ctrl, rmask, idv) ::planar::dpp_f64(__LINE__, (v), (ctrl), (idv))
namespace planar {
inline int wave_uni(int v) { return ::wave_emul::shfl(900001, v, 0); }
inline unsigned wave_uni(unsigned v) { return ::wave_emul::shfl(900002, v, 0); }
inline int wave_lane(int v, int l) { return ::wave_emul::shfl(900003, v, l); }
inline unsigned wave_lane(unsigned v, int l) { return ::wave_emul::shfl(900004, v, l); }
inline double wave_lane(double v, int l) { return ::wave_emul::shfl(900005, v, l); }
inline int wave_scan_add(int v) {
    const int l = ::wave_emul::S().cur & 63;
    for (int o = 1; o < 64; o <<= 1) { const int t = ::wave_emul::shfl(900010 + o, v, l - o >= 0 ? l - o : l); if (l >= o) v += t; }
    return v;
}
inline unsigned wave_min_u32(unsigned v) {
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = ::wave_emul::shfl(900100 + o, v, (::wave_emul::S().cur & 63) ^ o); v = t < v ? t : v; }
    return v;
}
// DPP on an FP64 value (wave_ops.h PLANAR_DPP_F64): only wave_shr:1 (0x138) is emulated - lane l receives lane l - 1's value, lane 0 keeps `idv`
inline double dpp_f64(int line, double v, int ctrl, double idv) {
    if (ctrl != 0x138) ::wave_emul::fail("wave_emul: DPP control not emulated");
    const int l = ::wave_emul::S().cur & 63;
    const double got = ::wave_emul::shfl(line, v, l > 0 ? l - 1 : 0);
    return l > 0 ? got : idv;
}
inline double wave_min_f64(double v) {
    for (int o = 32; o > 0; o >>= 1) { const double t = ::wave_emul::shfl(900200 + o, v, (::wave_emul::S().cur & 63) ^ o); v = t < v ? t : v; }
    return v;
}
}  // namespace planar

// ---- the HIP device functions the kernels use ----
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
inline double __hiloint2double(int hi, int lo) { const uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double d; memcpy(&d, &u, 8); return d; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
struct uint4 { unsigned x, y, z, w; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
inline long long wall_clock64() { return 0; }
inline long long clock64() { return 0; }
template <typename T> inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { const T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicAnd(T* p, T v) { const T o = *p; *p = o & v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { const T o = *p; *p = std::min(o, v); return o; }
using std::max;
using std::min;
