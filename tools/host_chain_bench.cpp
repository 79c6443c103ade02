// Rate of the host-side TranscriptRng chain (csrc/host_chain.hpp) on this machine: microseconds per Keccak-f[1600] permutation of
// a chain of `draws` draws, on 1 .. 2 x budget threads at once, and what host_cpu_budget() reports.
//   g++ -O3 -std=c++17 -DBPR1CS_HOST_ONLY -Ibulletproofs-r1cs-gadgets_amd/csrc tools/host_chain_bench.cpp -o /tmp/hcb -pthread && /tmp/hcb
#include <chrono>
#include <stdio.h>
#include "host_chain.hpp"

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 18656, m = 100;
    printf("host_cpu_budget = %u, hardware_concurrency = %u, bmi path = %d\n", host_cpu_budget(), std::thread::hardware_concurrency(),
           (int)(keccak_host_select() != keccak_f1600_host_generic));
    strobe init;
    merlin_new(init, (const uint8_t*)"bench", 5);
    const size_t draws = 2 * (size_t)n + 8;
    for (unsigned T : {1u, 2u, 4u, 8u, 16u, 32u, 64u}) {
        const uint32_t B = T;
        std::vector<uint8_t> V((size_t)B * m * 32, 7), bl((size_t)B * m * 32, 1), seeds((size_t)B * 32, 3);
        std::vector<uint64_t> raw(draws * B * 8);
        std::vector<strobe> tr(B);
        for (int rep = 0; rep < 2; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < T; t++)
                pool.emplace_back([&, t]() { host_front_chain(init, V.data() + (size_t)t * m * 32, bl.data() + (size_t)t * m * 32, seeds.data() + 32 * t, m, n, &tr[t], raw.data() + (size_t)t * 8 * draws); });
            for (auto& th : pool) th.join();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("%2u threads x 1 chain of %zu draws: %.2f ms wall, %.3f us per permutation per chain (%016llx)\n", T, draws, us / 1e3, us / draws, (unsigned long long)raw[8 * 5]);
        }
    }
    return 0;
}
