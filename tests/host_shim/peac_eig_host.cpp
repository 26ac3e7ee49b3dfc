// tests/host_shim/peac_eig_host.cpp — TEST INFRASTRUCTURE: compiles the product's wavefront eigen-solver (planarslam_amd/csrc/peac_eig.h)
// for the host so that tests/test_peac_eig.py can compare it with the oracle's restatement of Eigen (oracle/eigprim.cpp) bit for bit.
#include <cstdint>
#include <cstring>

#include "../../planarslam_amd/csrc/peac_eig.h"
#include "../../oracle/eigprim.h"

extern "C" {
// mats: [n][6] lower triangles a00,a10,a11,a20,a21,a22.  Returns the number of matrices whose (ev[3], v0[3]) differ in any bit.
long peac_eig_compare(const double* mats, long n, long* first_bad) {
    long bad = 0;
    for (long i = 0; i < n; i++) {
        const double* a = mats + i * 6;
        double ev[3], v0[3];
        planar::peac::eig33u(a[0], a[1], a[2], a[3], a[4], a[5], ev, v0);
        const double A[3][3] = {{a[0], a[1], a[3]}, {a[1], a[2], a[4]}, {a[3], a[4], a[5]}};
        double oe[3], Q[3][3];
        orc::eig33_selfadjoint(A, oe, Q);
        const double ov[3] = {Q[0][0], Q[1][0], Q[2][0]};
        if (std::memcmp(ev, oe, sizeof ev) != 0 || std::memcmp(v0, ov, sizeof ov) != 0) { if (!bad && first_bad) *first_bad = i; bad++; }
    }
    return bad;
}

// stats: [n][9] moments (sx sy sz sxx syy szz sxy syz sxz), N: [n].  lb[i] = merged_mse_lower_bound, mse[i] = what PlaneSeg::Stats::compute's
// solver returns for the same moments (the expressions of stats_compute_u in peac_ahc2.h).  Returns the number of i with lb[i] > mse[i].
long peac_lb_check(const double* stats, const int* N, long n, double* lb, double* mse) {
    long bad = 0;
    for (long i = 0; i < n; i++) {
        const double* s = stats + i * 9;
        const double sc = 1.0 / N[i];
        const double k00 = s[3] - s[0] * s[0] * sc, k01 = s[6] - s[0] * s[1] * sc, k02 = s[8] - s[0] * s[2] * sc;
        const double k11 = s[4] - s[1] * s[1] * sc, k12 = s[7] - s[1] * s[2] * sc, k22 = s[5] - s[2] * s[2] * sc;
        double ev[3], v[3];
        planar::peac::eig33u(k00, k01, k11, k02, k12, k22, ev, v);
        mse[i] = ev[0] * sc;
        lb[i] = planar::peac::merged_mse_lower_bound(s, N[i]);
        if (lb[i] > mse[i]) bad++;
    }
    return bad;
}
}
