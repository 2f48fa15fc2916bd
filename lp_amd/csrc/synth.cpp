// synth.cpp -- reproducible synthetic dense LPs.  Host code.
//
// Equality-form planted LP with a strictly complementary optimum, so that the slack-form matrix the
// hot path sees is exactly the dense m x n A (n_slack = 0, linear_program.rs:161) and x* is a
// known answer:  A_ij ~ N(0,1); basis B = m distinct columns; x*_B ~ U(1,2), x*_N = 0;
// y* ~ N(0,1); z*_N ~ U(1,2), z*_B = 0;  b = A x*,  c = A^T y* + z*.
// RNG: splitmix64(seed) -> xoshiro256**; U(0,1) from the top 53 bits; Box-Muller normals
// (both outputs used); Fisher-Yates for B.  Draw order: A (row-major), B, x*_B (in B order),
// y*, z*_N (ascending column).  lp_amd/synth.py mirrors this in numpy for cross-checking.
#include <cmath>
#include <cstdint>
#include <vector>
#include "../../include/lpipm.h"

namespace {
struct Rng {
    uint64_t s[4];
    bool have_spare = false;
    double spare = 0.0;
    explicit Rng(uint64_t seed) {
        uint64_t z = seed;
        for (int i = 0; i < 4; ++i) {  // splitmix64
            z += 0x9E3779B97F4A7C15ull;
            uint64_t r = z;
            r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull;
            r = (r ^ (r >> 27)) * 0x94D049BB133111EBull;
            s[i] = r ^ (r >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {  // xoshiro256**
        const uint64_t result = rotl(s[1] * 5, 7) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    double u01() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double normal() {
        if (have_spare) { have_spare = false; return spare; }
        const double u1 = 1.0 - u01();  // (0, 1]
        const double u2 = u01();
        const double r = std::sqrt(-2.0 * std::log(u1));
        const double th = 6.283185307179586476925286766559 * u2;
        spare = r * std::sin(th);
        have_spare = true;
        return r * std::cos(th);
    }
};
}  // namespace

extern "C" int lpipm_synth_planted_lp(uint64_t seed, uint64_t m, uint64_t n, double* A, double* b, double* c,
                                      double* xstar_out) {
    if (!A || !b || !c || m == 0 || n < m) return LPIPM_ERR_BAD_ARGUMENT;
    Rng g(seed);
    for (uint64_t i = 0; i < m * n; ++i) A[i] = g.normal();
    g.have_spare = false;
    std::vector<uint64_t> perm(n);
    for (uint64_t j = 0; j < n; ++j) perm[j] = j;
    for (uint64_t i = 0; i < m; ++i) {  // partial Fisher-Yates: perm[0..m) is the basis
        const uint64_t j = i + g.next() % (n - i);
        const uint64_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    std::vector<double> xs(n, 0.0), zs(n, 0.0), ys(m);
    std::vector<char> inB(n, 0);
    for (uint64_t i = 0; i < m; ++i) { xs[perm[i]] = 1.0 + g.u01(); inB[perm[i]] = 1; }
    for (uint64_t i = 0; i < m; ++i) ys[i] = g.normal();
    for (uint64_t j = 0; j < n; ++j) if (!inB[j]) zs[j] = 1.0 + g.u01();
    for (uint64_t i = 0; i < m; ++i) {
        const double* Ai = A + i * n;
        double s = 0.0;
        for (uint64_t j = 0; j < n; ++j) s += Ai[j] * xs[j];
        b[i] = s;
    }
    for (uint64_t j = 0; j < n; ++j) c[j] = zs[j];
    for (uint64_t i = 0; i < m; ++i) {
        const double* Ai = A + i * n;
        const double yi = ys[i];
        for (uint64_t j = 0; j < n; ++j) c[j] += Ai[j] * yi;
    }
    if (xstar_out) for (uint64_t j = 0; j < n; ++j) xstar_out[j] = xs[j];
    return LPIPM_OK;
}
