#include "../../bulletproofs-r1cs-gadgets_amd/host/scalar_host.hpp"
#include <chrono>
#include <cstdio>
#include <random>
using namespace bpr1cs;
int main() {
    std::mt19937_64 rng(12345);
    auto rnd = [&]() { uint8_t b[64]; for (int i = 0; i < 64; i += 8) { uint64_t x = rng(); memcpy(b + i, &x, 8); } return sc_mont_from_wide(b); };
    int bad = 0;
    sc edge[6] = {sc_zero(), sc_one_mont(), sc_neg(sc_one_mont()), sc_mont_from_u64(2), sc_neg(sc_mont_from_u64(2)), sc_const(SC_R2)};
    for (int i = 0; i < 200000; i++) {
        sc a = i < 6 ? edge[i] : rnd(), b = rnd();
        if (i > 6 && i < 200) { memset(a.v, 0, 32); a.v[(i % 8)] = (uint32_t)rng() | 1u; a = sc_mul(a, sc_const(SC_R2)); }   // small / sparse values
        sc i1 = sc_invert(a), i2 = hostsc::invert(a);
        if (memcmp(i1.v, i2.v, 32)) { bad++; if (bad < 5) printf("invert mismatch at %d\n", i); }
        sc m1 = sc_mul(a, b), m2 = hostsc::mul(a, b), s1 = sc_add(a, b), s2 = hostsc::add(a, b), d1 = sc_sub(a, b), d2 = hostsc::sub(a, b);
        if (memcmp(m1.v, m2.v, 32) || memcmp(s1.v, s2.v, 32) || memcmp(d1.v, d2.v, 32)) { bad++; if (bad < 5) printf("arith mismatch at %d\n", i); }
        sc f1 = sc_from_mont(a), f2 = hostsc::from_mont(a);   // out of Montgomery form: the reduction alone
        if (memcmp(f1.v, f2.v, 32)) { bad++; if (bad < 5) printf("from_mont mismatch at %d\n", i); }
    }
    sc x = rnd(), acc = sc_one_mont();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200000; i++) { x = hostsc::add(x, acc); acc = hostsc::mul(acc, hostsc::invert(x)); }
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000000; i++) { x = hostsc::add(hostsc::mul(x, acc), acc); }
    auto t2 = std::chrono::steady_clock::now();
    printf("%s: %d mismatches; invert+mul+add %.2f us, mul+add %.1f ns (%08x)\n", bad ? "FAIL" : "OK", bad, std::chrono::duration<double, std::micro>(t1 - t0).count() / 2e5,
           std::chrono::duration<double, std::nano>(t2 - t1).count() / 2e6, x.v[0]);
    return bad != 0;
}
