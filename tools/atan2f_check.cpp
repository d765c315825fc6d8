// Exhaustive host check of the fdlibm ports in f_renderer_amd/csrc/frr_exact.h against this
// machine's libm (glibc): all 2^32 atanf inputs, 2*10^9 atan2f pairs (half raw bit patterns, half
// screen-space-like magnitudes), all 2^32 y with x == 1.0 and a grid of special values -- for both
// the branchy port (fd_*) and the branch-free device form (fd_*_bf).
//   g++ -O2 -ffp-contract=off -pthread -Itools/hostshim tools/atan2f_check.cpp -o /tmp/atan2f_check
#include "../f_renderer_amd/csrc/frr_exact.h"
#include <atomic>
#include <cmath>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

using frr::f2u;
using frr::u2f;
static const int NT = 8;
static std::atomic<uint64_t> bad{0};

static inline bool same(float a, float b) { return f2u(a) == f2u(b) || (a != a && b != b); }

static inline void check2(float y, float x, uint64_t &b)
{
    const float a = atan2f(y, x), c = frr::fd_atan2f(y, x), d = frr::fd_atan2f_bf(y, x);
    if (!same(a, c) || !same(a, d)) {
        if (b < 3) printf("atan2f mismatch y=%a x=%a libm=%a port=%a bf=%a\n", y, x, a, c, d);
        ++b;
    }
}

template <class F> static void par(F f)
{
    std::vector<std::thread> th;
    for (int t = 0; t < NT; ++t) th.emplace_back([=] { uint64_t b = 0; f(t, b); bad += b; });
    for (auto &t : th) t.join();
}

int main()
{
    par([](int t, uint64_t &b) {
        for (uint64_t u = t; u < (1ull << 32); u += NT) {
            const float x = u2f((uint32_t)u), a = atanf(x), c = frr::fd_atanf(x), d = frr::fd_atanf_bf(x);
            if (!same(a, c) || !same(a, d)) {
                if (b < 3) printf("atanf mismatch x=%a libm=%a port=%a bf=%a\n", x, a, c, d);
                ++b;
            }
        }
    });
    printf("atanf exhaustive mismatches: %llu\n", (unsigned long long)bad.load());
    uint64_t total = bad.exchange(0);

    par([](int t, uint64_t &b) {
        for (uint64_t u = t; u < (1ull << 32); u += NT) check2(u2f((uint32_t)u), 1.0f, b);
    });
    printf("atan2f(y, 1.0) exhaustive mismatches: %llu\n", (unsigned long long)bad.load());
    total += bad.exchange(0);

    {
        const uint32_t sp[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x007fffffu, 0x00800000u, 0x3f800000u,
                               0xbf800000u, 0x7f7fffffu, 0xff7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0xffc00000u,
                               0x7f800001u, 0x5f800000u, 0x1f800000u, 0x4c000000u, 0x4bffffffu, 0x31000000u, 0x30ffffffu};
        uint64_t b = 0;
        for (uint32_t yy : sp)
            for (uint32_t xx : sp) check2(u2f(yy), u2f(xx), b);
        // exponent-difference sweep around the k = +-60 thresholds
        for (int ey = 1; ey < 255; ++ey)
            for (int ex = 1; ex < 255; ++ex)
                for (int s = 0; s < 4; ++s)
                    check2(u2f(((uint32_t)ey << 23) | ((s & 1) << 31) | 0x2aaaaau), u2f(((uint32_t)ex << 23) | ((uint32_t)(s >> 1) << 31) | 0x155555u), b);
        bad += b;
    }
    printf("atan2f special grid mismatches: %llu\n", (unsigned long long)bad.load());
    total += bad.exchange(0);

    par([](int t, uint64_t &b) {
        std::mt19937_64 g(1234 + t);
        for (uint64_t i = 0; i < 250000000ull; ++i) {
            const uint64_t r = g();
            float y, x;
            if (i & 1) {
                y = u2f((uint32_t)r);
                x = u2f((uint32_t)(r >> 32));
            } else {
                y = (float)((int32_t)(r & 0xffffff) - 0x800000) * (1.0f / 4096.f);
                x = (float)((int32_t)((r >> 24) & 0xffffff) - 0x800000) * (1.0f / 4096.f);
                if (i & 2) y *= 1e-3f;
            }
            check2(y, x, b);
        }
    });
    printf("atan2f 2e9 random mismatches: %llu\n", (unsigned long long)bad.load());
    total += bad.exchange(0);
    printf("TOTAL mismatches: %llu\n", (unsigned long long)total);
    return total ? 1 : 0;
}
