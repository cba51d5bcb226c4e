// handover_tsan.cpp - the helper / waiter hand-over of csrc/ttm_handover.h (what bounds ncclCommInitRank in csrc/ttm_comm.cpp)
// under ThreadSanitizer, with a stand-in for the blocking call: the producer sleeps a random time around the waiter's limit, so
// that all three outcomes occur many times - result taken, wait abandoned (the helper must drop what it makes), and the two
// compare-exchanges racing.  Checks: every result is either taken or dropped, exactly once.
//     g++ -O1 -g -std=c++17 -fsanitize=thread -pthread tools/sanitize/handover_tsan.cpp -o /tmp/handover_tsan && /tmp/handover_tsan
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../triangular_transport_toolbox_amd/csrc/ttm_handover.h"

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    std::atomic<int> made{0}, dropped{0};
    int taken = 0, abandoned = 0;
    std::mt19937 rng(7);
    for (int i = 0; i < rounds; ++i) {
        const int us = 500 + (int)(rng() % 3000);                 // producer: 0.5 .. 3.5 ms; waiter's limit: 2 ms
        int out = -1;
        const int how = ttm_handover::run_with_timeout<int>(
            [us, i, &made] { std::this_thread::sleep_for(std::chrono::microseconds(us)); made.fetch_add(1); return i; },
            [&dropped](const int&) { dropped.fetch_add(1); }, 0.002, out, 0);
        if (how == 0) { if (out != i) { printf("wrong result %d != %d\n", out, i); return 1; } ++taken; }
        else if (how == 1) ++abandoned;
        else { printf("no thread\n"); return 1; }
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(50));  // (the abandoned helpers finish)
    printf("rounds %d: taken %d, abandoned %d, made %d, dropped %d\n", rounds, taken, abandoned, made.load(), dropped.load());
    if (made.load() != rounds || taken + dropped.load() != rounds || dropped.load() != abandoned) { printf("FAILED: a result was lost or handled twice\n"); return 1; }
    if (taken == 0 || abandoned == 0) { printf("FAILED: one of the outcomes never occurred\n"); return 1; }
    printf("ok\n");
    return 0;
}
