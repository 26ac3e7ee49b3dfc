// tests/host_shim/lsd_nfa_host.cpp — TEST INFRASTRUCTURE: the wavefront version of LSD's NFA tail (planarslam_amd/csrc/lsd_nfa.h, the chain that
// travels through the lanes on DPP) compiled for the host on the wave64 emulator, against the reference's sequential loop.
#include "wave_emul.h"

#include "../../planarslam_amd/csrc/lsd_nfa.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace {
struct Args { double term; int n, k; double p_term, log_nt; double out[64]; };
void entry(void* a) {
    Args* A = (Args*)a;
    const int lane = threadIdx.x;
    A->out[lane] = planar::lsd::nfa_tail(A->term, A->n, A->k, A->p_term, A->log_nt, lane);
}
// the loop as the library runs it (restated with citations in oracle/lsd_oracle.cpp:265-277)
double scalar_tail(double term, int n, int k, double p_term, double LOG_NT) {
    double bin_tail = term;
    const double tolerance = 0.1;
    for (int i = k + 1; i <= n; ++i) {
        const double bin_term = double(n - i + 1) / double(i);
        const double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1) {
            const double err = term * ((1 - std::pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
            if (err < tolerance * std::fabs(-std::log10(bin_tail) - LOG_NT) * bin_tail) break;
        }
    }
    return -std::log10(bin_tail) - LOG_NT;
}
}  // namespace

extern "C" {
// cases [count][4] = term, n, k, p_term.  got / want [count]: the wavefront's value (lane 0) and the sequential loop's.  Returns the number of cases in which
// any lane differs from the sequential loop in any bit, or -1 with a message.
long lsd_nfa_tail_compare(const double* cases, long count, double log_nt, double* got, double* want, char* err, int errlen) {
    try {
        long bad = 0;
        for (long c = 0; c < count; c++) {
            Args A{cases[4 * c], (int)cases[4 * c + 1], (int)cases[4 * c + 2], cases[4 * c + 3], log_nt, {}};
            wave_emul::Dim3 bi, bd; bd.x = 64; bd.y = 1; bd.z = 1;
            wave_emul::launch_block(entry, &A, 64, bi, bd, 0);
            want[c] = scalar_tail(A.term, A.n, A.k, A.p_term, log_nt);
            got[c] = A.out[0];
            bool same = true;
            for (int l = 0; l < 64; l++) same = same && std::memcmp(&A.out[l], &want[c], 8) == 0;
            if (!same) bad++;
        }
        return bad;
    } catch (const std::exception& e) { snprintf(err, errlen, "%s", e.what()); return -1; }
}
}
