// What waking the host pool costs the work it is woken for, and what waking it AHEAD of the work (host_pool_prewake) saves.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pool_wake_bench.hip -o /tmp/pool_wake_bench -Lzklaim_amd -lzkg -Wl,-rpath,$PWD/zklaim_amd && /tmp/pool_wake_bench
// Each sample: the workers asleep for 1.5 ms (as between two steps), then 16 tasks of 18 us (the windows' chunk sums of a 2^20-point job).
#include "../zklaim_amd/csrc/common.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
using namespace zk;
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void busy(double us) { const double t0 = now_us(); while (now_us() - t0 < us) {} }
int main() {
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            std::vector<double> v;
            for (int k = 0; k < 300; ++k) {
                std::this_thread::sleep_for(std::chrono::microseconds(1500));
                if (mode) host_pool_prewake(400);                                   // (run with ZKG_POOL_PREWAKE=1)
                busy(250.0);                                                  // the caller waits for the GPU (here: spins)
                const double t0 = now_us();
                host_parallel_for(16, [&](int) { busy(18.0); });
                v.push_back(now_us() - t0);
            }
            std::sort(v.begin(), v.end());
            printf("%s: 16 tasks of 18 us took median %.1f us  p90 %.1f  min %.1f\n", mode ? "woken 250 us ahead" : "woken with the work ", v[150], v[270], v[0]);
        }
    return 0;
}
