// planarslam_amd/csrc/wave_ops.h — wavefront-level helpers for the single-wavefront kernels (gfx950, wave64).
//
// A lone wavefront issues roughly one instruction every five cycles, so these kernels are bound by their instruction count; and hipcc treats the
// result of __shfl (ds_bpermute) as DIVERGENT, which drags every value computed from it - loop counters, error flags - into VGPRs and wraps the
// control flow around them in exec-mask bookkeeping.  The helpers keep wave-uniform values in SGPRs (v_readlane / v_readfirstlane instead of shuffles)
// and run prefix sums on DPP row operations (6 VALU instructions) instead of six LDS-pipe shuffles.
// tests/host_shim/wave_emul.h provides the same functions for the host emulator (PLANAR_WAVE_EMUL).
#pragma once

#ifndef PLANAR_WAVE_EMUL
namespace planar {

__device__ __forceinline__ int wave_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned wave_uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
// value of lane `l` (l wave-uniform): an SGPR
__device__ __forceinline__ int wave_lane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ unsigned wave_lane(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ double wave_lane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// inclusive prefix sum over the 64 lanes: Kogge-Stone inside the 16-lane rows (row_shr 1, 2, 4, 8), then the row totals through row_bcast15 / row_bcast31
// (GFX9 DPP; the sequence LLVM's atomic optimizer emits for a wave64 scan)
__device__ __forceinline__ int wave_scan_add(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// minimum over the 64 lanes, as a wave-uniform value: the same DPP ladder with v_min_u32 (lanes without a source see ~0)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    const int top = -1;
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(top, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(top, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(top, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(top, (int)v, 0x118, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(top, (int)v, 0x142, 0xa, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(top, (int)v, 0x143, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// reductions over the 64 lanes with the result in every lane (wave-uniform): the DPP ladder leaves the total in lane 63.  A butterfly of __shfl_xor costs
// six LDS-pipe permutes per 32-bit word (~120 cycles each for a lone wavefront); the ladder is six VALU steps.
__device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(wave_scan_add(v), 63); }
#define PLANAR_DPP_F64(v, ctrl, rmask, idv)                                                                                   \
    __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(idv), __double2hiint(v), ctrl, rmask, 0xf, false),            \
                     __builtin_amdgcn_update_dpp(__double2loint(idv), __double2loint(v), ctrl, rmask, 0xf, false))
__device__ __forceinline__ double wave_max_f64(double v) {
    const double id = -__builtin_inf();
    v = fmax(v, PLANAR_DPP_F64(v, 0x111, 0xf, id)); v = fmax(v, PLANAR_DPP_F64(v, 0x112, 0xf, id));
    v = fmax(v, PLANAR_DPP_F64(v, 0x114, 0xf, id)); v = fmax(v, PLANAR_DPP_F64(v, 0x118, 0xf, id));
    v = fmax(v, PLANAR_DPP_F64(v, 0x142, 0xa, id)); v = fmax(v, PLANAR_DPP_F64(v, 0x143, 0xc, id));
    return wave_lane(v, 63);
}
__device__ __forceinline__ double wave_min_f64(double v) {
    const double id = __builtin_inf();
    v = fmin(v, PLANAR_DPP_F64(v, 0x111, 0xf, id)); v = fmin(v, PLANAR_DPP_F64(v, 0x112, 0xf, id));
    v = fmin(v, PLANAR_DPP_F64(v, 0x114, 0xf, id)); v = fmin(v, PLANAR_DPP_F64(v, 0x118, 0xf, id));
    v = fmin(v, PLANAR_DPP_F64(v, 0x142, 0xa, id)); v = fmin(v, PLANAR_DPP_F64(v, 0x143, 0xc, id));
    return wave_lane(v, 63);
}

}  // namespace planar
#endif
