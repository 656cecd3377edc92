// icet_amd/csrc/icet_shuffle.h -- the down-sample indices of the map maker (src/simpleMapMaker.cpp:147-158): the reference fills a vector with 0 .. n-1,
// std::shuffle()s it with the node's std::mt19937 and takes the first `map_downsample` (2000) entries.  The n - 1 draws are the cost that cannot go (the generator has to
// leave the call in the state std::shuffle leaves it in: the next frame continues the stream); the n-entry vector can.  libstdc++'s shuffle walks i = 1 .. n-1 and swaps
// entry i with entry j <= i; entry i still holds i when its turn comes (earlier steps only touched entries below it), so a step with i >= m changes the first m entries only
// when j < m -- entry j becomes i -- and whatever leaves to position i never comes back below m.  Pure C++ (tests/cpp/test_shuffle.cpp compares it with std::shuffle on the host;
// the node runs that comparison once when it is created and keeps the full shuffle if this C++ library's std::shuffle should draw differently).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <numeric>
#include <random>
#include <type_traits>
#include <utility>
#include <vector>

namespace icet_shuffle {

// head = the first min(m, n) entries of v after { std::iota(v.begin(), v.end(), 0); std::shuffle(v.begin(), v.end(), g); } for a v of n std::size_t, with g advanced as
// std::shuffle advances it.  The two branches are libstdc++'s (bits/stl_algo.h: pairs of swap positions from one draw while n * n fits the generator's range, one draw per
// step otherwise), called with the library's own distribution objects in the library's order.
template <class Gen>
inline void head_of_shuffled_iota(std::size_t n, std::size_t m, Gen& g, std::vector<std::size_t>& head) {
    using ud = typename std::make_unsigned<std::ptrdiff_t>::type;
    using D = std::uniform_int_distribution<ud>;
    using P = typename D::param_type;
    using uc = typename std::common_type<typename std::remove_reference<Gen>::type::result_type, ud>::type;
    if (m > n) m = n;
    head.resize(m);
    std::iota(head.begin(), head.end(), (std::size_t)0);
    if (n == 0) return;
    auto step = [&](std::size_t i, std::size_t j) { if (i < m) std::swap(head[i], head[j]); else if (j < m) head[j] = i; };
    const uc urngrange = g.max() - g.min();
    const uc urange = uc(n);
    if (urngrange / urange >= urange) {
        std::size_t i = 1;
        if ((urange % 2) == 0) { D d{0, 1}; const std::size_t j = (std::size_t)d(g); step(i, j); i++; }
        while (i != n) {
            const uc b0 = uc(i) + 1, b1 = b0 + 1;
            const uc x = std::uniform_int_distribution<uc>{0, (b0 * b1) - 1}(g);
            step(i, (std::size_t)(x / b1)); i++;
            step(i, (std::size_t)(x % b1)); i++;
        }
        return;
    }
    D d;
    for (std::size_t i = 1; i != n; ++i) step(i, (std::size_t)d(g, P(0, i)));
}

// The same with the generator and the bounded draw written out: std::mt19937 is specified bit for bit by the standard (MT19937, default seed 5489), and libstdc++ maps a
// 32-bit generator onto [0, range) by Lemire's multiply-and-reject (bits/uniform_int_dist.h, _S_nd) -- that part is the LIBRARY's choice, so a node uses this form only after
// matches_std_shuffle() below has seen it reproduce std::shuffle on the running library, both branches.  About a third faster than the library objects on the hosts measured
// (the state is refilled 624 words at a time in a loop the compiler vectorises; no per-draw parameter objects).
struct FastMt {
    std::uint32_t s[624]; int at = 624;
    explicit FastMt(std::uint32_t seed = 5489u) { s[0] = seed; for (int i = 1; i < 624; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (std::uint32_t)i; }
    void refill() {
        auto tw = [](std::uint32_t u, std::uint32_t v, std::uint32_t w) { const std::uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu); return w ^ (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); };
        for (int i = 0; i < 227; i++) s[i] = tw(s[i], s[i + 1], s[i + 397]);
        for (int i = 227; i < 623; i++) s[i] = tw(s[i], s[i + 1], s[i - 227]);
        s[623] = tw(s[623], s[0], s[396]);
        at = 0;
    }
    std::uint32_t operator()() {
        if (at >= 624) refill();
        std::uint32_t y = s[at++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    std::uint32_t below(std::uint32_t range) {                   // uniform in [0, range), range >= 1: libstdc++'s _S_nd<uint64_t> on a 32-bit generator
        std::uint64_t product = (std::uint64_t)(*this)() * range;
        std::uint32_t low = (std::uint32_t)product;
        if (low < range) {
            const std::uint32_t threshold = (0u - range) % range;
            while (low < threshold) { product = (std::uint64_t)(*this)() * range; low = (std::uint32_t)product; }
        }
        return (std::uint32_t)(product >> 32);
    }
};
inline void head_of_shuffled_iota(std::size_t n, std::size_t m, FastMt& g, std::vector<std::size_t>& head) {
    if (m > n) m = n;
    head.resize(m);
    std::iota(head.begin(), head.end(), (std::size_t)0);
    if (n == 0) return;
    auto step = [&](std::size_t i, std::size_t j) { if (i < m) std::swap(head[i], head[j]); else if (j < m) head[j] = i; };
    if (0xffffffffull / n >= n) {                                 // two swap positions from one draw
        std::size_t i = 1;
        if ((n % 2) == 0) { step(i, g.below(2u)); i++; }
        while (i != n) {
            const std::uint64_t b0 = i + 1, b1 = b0 + 1;
            const std::uint64_t x = g.below((std::uint32_t)(b0 * b1));
            step(i, (std::size_t)(x / b1)); i++;
            step(i, (std::size_t)(x % b1)); i++;
        }
        return;
    }
    if (n > 0xffffffffull) return;                                // (never: a scan has fewer than 2^30 rows)
    for (std::size_t i = 1; i != n; ++i) step(i, g.below((std::uint32_t)(i + 1)));
}

// Does the function above draw what THIS C++ library's std::shuffle draws?  (both branches, odd and even sizes, m below and above n; the generators must end in one state)
inline bool matches_std_shuffle() {
    const std::size_t sizes[] = {0, 1, 2, 3, 7, 1000, 1001, 65535, 65536, 70001};
    for (std::size_t n : sizes)
        for (std::size_t m : {(std::size_t)0, (std::size_t)5, (std::size_t)2000}) {
            std::mt19937 a(12345u + (unsigned)n), b(12345u + (unsigned)n);
            std::vector<std::size_t> v(n); std::iota(v.begin(), v.end(), (std::size_t)0);
            std::shuffle(v.begin(), v.end(), a);
            std::vector<std::size_t> head;
            head_of_shuffled_iota(n, m, b, head);
            if (head.size() != std::min(m, n)) return false;
            for (std::size_t k = 0; k < head.size(); k++) if (head[k] != v[k]) return false;
            if (a() != b()) return false;
        }
    return true;
}
// ... and does the written-out form (FastMt) draw the same?  Chained calls on one generator, both branches.
inline bool fast_matches_std_shuffle() {
    std::mt19937 a; FastMt b;
    const std::size_t sizes[] = {1, 2, 5, 1000, 64001, 65535, 65536, 65537, 99991, 3, 131072, 0, 4096};
    for (std::size_t n : sizes) {
        std::vector<std::size_t> v(n); std::iota(v.begin(), v.end(), (std::size_t)0);
        std::shuffle(v.begin(), v.end(), a);
        std::vector<std::size_t> head;
        head_of_shuffled_iota(n, (std::size_t)2000, b, head);
        if (head.size() != std::min((std::size_t)2000, n)) return false;
        for (std::size_t k = 0; k < head.size(); k++) if (head[k] != v[k]) return false;
    }
    return a() == b();
}

}  // namespace icet_shuffle
