// Stress test of april_asr_amd/csrc/host_pool.h (built by tests/test_abi.py::test_host_pool, with ThreadSanitizer when
// the toolchain has it): many jobs of random size, every index visited exactly once, results visible to the caller.
#include "host_pool.h"
#include <cstdio>
#include <cstdlib>
#include <numeric>

int main()
{
    for (int helpers : {0, 1, 3, 7}) {
        aprilx::HostPool pool(helpers);
        unsigned seed = 12345u + (unsigned)helpers;
        for (int job = 0; job < 3000; ++job) {
            seed = seed * 1664525u + 1013904223u;
            const size_t n = (seed >> 8) % 3000;
            const size_t grain = 1 + (seed >> 20) % 97;
            std::vector<int> hits(n, 0);
            std::vector<long> out(n, 0);
            pool.run(n, grain, [&](size_t i) { hits[i] += 1; out[i] = (long)i * 3 + job; });
            for (size_t i = 0; i < n; ++i)
                if (hits[i] != 1 || out[i] != (long)i * 3 + job) { printf("FAIL helpers=%d job=%d i=%zu hits=%d\n", helpers, job, i, hits[i]); return 1; }
        }
    }
    printf("ok\n");
    return 0;
}
