// planarslam_amd/csrc/lsd_nfa.h — the binomial tail of LSD's NFA (rect_nfa -> nfa(), the loop of the OpenCV LSD the reference calls through
// src/LSDextractor.cpp; restated with citations in oracle/lsd_oracle.cpp:265-277) for a wavefront.  Included by lsd.hip; also compiled for the host by
// tests/host_shim/lsd_nfa_host.cpp on the wave64 emulator (tests/test_lsd_nfa_emul.py), so it only depends on wave_ops.h.
#pragma once
#include "wave_ops.h"

namespace planar {
namespace lsd {

// term = the binomial term of k, n, p (exp(log1term) in nfa()); returns -log10(bin_tail) - LOG_NT where bin_tail is what the reference's loop
//     for (i = k + 1; i <= n; ++i) { bin_term = (n - i + 1) / i; mult_term = bin_term * p_term; term *= mult_term; bin_tail += term;
//                                    if (bin_term < 1) { err = term * ((1 - pow(mult_term, n - i + 1)) / (1 - mult_term) - 1);
//                                                        if (err < tolerance * fabs(-log10(bin_tail) - LOG_NT) * bin_tail) break; } }
// leaves.  All 64 lanes call it together with the same arguments (lane = the caller's lane id).
__device__ inline double nfa_tail(double term, int n, int k, double p_term, double LOG_NT, int lane) {
    double bin_tail = term;
    const double tolerance = 0.1;
    // The tail is a sequential recurrence (term *= mult; bin_tail += term), but the stopping rule of iteration i only reads iteration i's
    // values.  Per block of 64 iterations: lane j computes the multiplier of "its" iteration (the division is the expensive part and is
    // independent), the chain runs through the lanes (below), every lane ends up with the values of its own step and evaluates the expensive
    // rule (pow, log10) for it - only in blocks that contain a step with bin_term < 1; the first lane whose rule fires is where the reference breaks.
    for (int i = k + 1; i <= n;) {
        const int cnt = min(64, n - i + 1);
        const double my_bin = lane < cnt ? double(n - (i + lane) + 1) / double(i + lane) : 2.0;
        const double my_mult = my_bin * p_term;
        const bool rule = lane < cnt && my_bin < 1;
        // The chain travels through the lanes: every step, each lane takes the pair (term, tail) of the lane below it (DPP wave_shr:1; lane 0 takes the
        // block's input), multiplies / adds ITS iteration's multiplier.  Lane 0 is right from step 0 on, lane s from step s on (its input no longer changes),
        // so after cnt steps lane s < cnt holds exactly the reference's values after iteration i + s - the same operations in the same order, six VALU
        // instructions per step and no cross-lane reads (a v_readlane pair per step cost 4x as much).
        double t = term, bt = bin_tail;
        for (int step = 0; step < cnt; step++) {                 // lane cnt - 1 is right after cnt steps
            const double tin = PLANAR_DPP_F64(t, 0x138, 0xf, term), bin = PLANAR_DPP_F64(bt, 0x138, 0xf, bin_tail);
            t = tin * my_mult;
            bt = bin + t;
        }
        if (__ballot(rule) != 0) {
            bool brk = false;
            if (rule) {
                const double err = t * ((1 - pow(my_mult, double(n - (i + lane) + 1))) / (1 - my_mult) - 1);
                brk = err < tolerance * fabs(-log10(bt) - LOG_NT) * bt;
            }
            const unsigned long long m = __ballot(brk);
            if (m) { bin_tail = planar::wave_lane(bt, __ffsll((long long)m) - 1); return -log10(bin_tail) - LOG_NT; }
        }
        term = planar::wave_lane(t, cnt - 1); bin_tail = planar::wave_lane(bt, cnt - 1); i += cnt;
    }
    return -log10(bin_tail) - LOG_NT;
}

}  // namespace lsd
}  // namespace planar
