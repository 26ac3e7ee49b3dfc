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
}
