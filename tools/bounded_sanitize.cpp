// bounded_sanitize.cpp — csrc/hip/bounded.h (pure host code) under ThreadSanitizer / AddressSanitizer: calls that return in time, calls that
// outlive their bound (the helper is abandoned, the next call gets a fresh one), concurrent callers, phases beginning and ending while the
// watcher runs.   tools/sanitize_bounded.sh builds and runs it both ways.
#include <cassert>
#include <cstdio>
#include <thread>
#include <vector>

#include "bounded.h"

using namespace bounded_ns;

int main() {
    setenv("GPUART_HIP_PHASE_LOG", "0", 1);
    // in time
    for (int k = 0; k < 50; k++) {
        Outcome o = bounded("quick", 1000, [k](std::string &d) { d = "n" + std::to_string(k); return k; });
        assert(!o.timed_out && o.rc == k && o.detail == "n" + std::to_string(k));
    }
    // outlives its bound: the caller gets a timeout, the layer is marked stuck, a later call is served by a fresh helper
    Outcome t = bounded("slow", 50, [](std::string &) { std::this_thread::sleep_for(std::chrono::milliseconds(400)); return 0; });
    assert(t.timed_out && stuck().load() && stuck_in() == "slow");
    Outcome again = bounded("after", 1000, [](std::string &d) { d = "ok"; return 7; });
    assert(!again.timed_out && again.rc == 7);
    // several callers at once (each on its own helper; none may lose its result)
    std::vector<std::thread> th;
    std::atomic<int> sum{0};
    for (int k = 0; k < 8; k++)
        th.emplace_back([k, &sum] {
            for (int i = 0; i < 20; i++) {
                Outcome o = bounded("mt", 2000, [k, i](std::string &) { return 100 * k + i; });
                assert(!o.timed_out && o.rc == 100 * k + i);
                sum += 1;
                log_error(-2, "message " + std::to_string(k));
            }
        });
    // phases while that runs
    for (int k = 0; k < 200; k++) { phase_begin("p", 5000); phase_end(); }
    for (auto &x : th) x.join();
    assert(sum == 160);
    // calls that wait for EACH OTHER (every rank of a one-thread-per-rank process inside ncclCommInitRank): all must be inside at once
    {
        std::vector<std::thread> ranks;
        auto arrived = std::make_shared<std::atomic<int>>(0);
        std::atomic<int> ok{0};
        for (int k = 0; k < 6; k++)
            ranks.emplace_back([arrived, &ok] {
                Outcome o = bounded("rendezvous", 5000, [arrived](std::string &) {
                    arrived->fetch_add(1);
                    while (arrived->load() < 6) std::this_thread::sleep_for(std::chrono::milliseconds(1));
                    return 0;
                });
                if (!o.timed_out) ok++;
            });
        for (auto &x : ranks) x.join();
        assert(ok == 6);
    }
    assert(!recent_errors().empty());
    std::this_thread::sleep_for(std::chrono::milliseconds(450));  // let the abandoned helper finish its sleep before exit
    printf("bounded.h: ok\n");
    return 0;
}
