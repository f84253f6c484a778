// xorwow_rocrand_check.cpp -- an INDEPENDENT implementation of the XORWOW step and of the 2^67-stride subsequence jump,
// for tests/test_oracle_pins.py: rocRAND's own engine (/opt/rocm/include/rocrand/rocrand_xorwow.h, part of the ROCm image, not
// of the reference and not of this repository), driven on the HOST.
//
// The reference draws from cuRAND's XORWOW (curand_init(seed, subsequence, 0), render.cuh:72); the oracle and the HIP library
// restate that generator from the published algorithm.  rocRAND implements the same generator -- same xorshift step, same
// Weyl increment 362437, same definition of a subsequence (2^67 draws, applied with precomputed GF(2) matrices) -- and differs
// from cuRAND only in the constants that scramble the seed into the initial state.  So: start rocRAND's engine from an
// arbitrary 160 + 32-bit state (the oracle's cuRAND-scrambled state of the seed), let IT skip to subsequence k and draw,
// and compare with what the oracle's curand_init(seed, k) + draws give.  What this cannot check is the scramble itself
// (five constants recalled from the public cuRAND header: SURVEY Appendix A.6).
//
// usage: xorwow_rocrand_check d v0 v1 v2 v3 v4 n_draws k1 k2 ...   ->   one line per k: "k d v0 v1 v2 v3 v4 raw1 .. rawN"
#include <cstdio>
#include <cstdlib>

#include <rocrand/rocrand_xorwow.h>

struct Engine : rocrand_device::xorwow_engine {
    void set(unsigned d, const unsigned v[5]) {
        m_state.d = d;
        for (int i = 0; i < 5; i++) m_state.x[i] = v[i];
    }
    void print(unsigned long long k, int draws) {
        printf("%llu %u %u %u %u %u %u", k, m_state.d, m_state.x[0], m_state.x[1], m_state.x[2], m_state.x[3], m_state.x[4]);
        for (int i = 0; i < draws; i++) printf(" %u", next());
        printf("\n");
    }
};

int main(int argc, char **argv) {
    if (argc < 9) return 2;
    const unsigned d = (unsigned)strtoul(argv[1], nullptr, 10);
    unsigned v[5];
    for (int i = 0; i < 5; i++) v[i] = (unsigned)strtoul(argv[2 + i], nullptr, 10);
    const int draws = atoi(argv[7]);
    for (int a = 8; a < argc; a++) {
        const unsigned long long k = strtoull(argv[a], nullptr, 10);
        Engine e;
        e.set(d, v);
        e.discard_subsequence(k);
        e.print(k, draws);
    }
    return 0;
}
