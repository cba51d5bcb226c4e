// ttm_handover.h - a blocking call run on a helper thread with a time limit on the caller's side.
//
// ncclCommInitRank blocks until every rank has called it and cannot be cancelled from outside: csrc/ttm_comm.cpp runs it on a
// helper thread and waits a bounded time.  Hand-over between helper and waiter is ONE tri-state word: PENDING -> DONE
// (helper: the result is published) or PENDING -> ABANDONED (waiter: timed out).  Whoever loses the compare-exchange knows
// what the other did - a helper that finds ABANDONED owns the result it has just made and drops it (a communicator nobody
// will ever take is destroyed, never handed out and destroyed, never leaked); a waiter whose exchange fails finds DONE and
// takes the result.  The reference count only decides who frees the shared block.
// (A header of its own so that the protocol can be driven under ThreadSanitizer with a stand-in for the blocking call:
// tools/sanitize/handover_tsan.cpp.)
#pragma once

#include <atomic>
#include <chrono>
#include <thread>

namespace ttm_handover {

enum : int { PENDING = 0, DONE = 1, ABANDONED = 2 };

template <class R>
struct Pending {
    std::atomic<int> state{PENDING};
    std::atomic<int> owners{2};
    R result{};
};

template <class R>
inline void release(Pending<R>* p) {
    if (p->owners.fetch_sub(1, std::memory_order_acq_rel) == 1) delete p;
}

// Runs `produce()` (-> R) on a detached helper thread and waits up to limit_s seconds for it.
// Returns 0 and sets `out` when the result arrived in time; 1 when the wait was abandoned - the helper will then call
// `drop(result)` on whatever it produces; -1 when no thread could be started.  poll_ms: the waiter's polling interval.
template <class R, class Produce, class Drop>
int run_with_timeout(Produce produce, Drop drop, double limit_s, R& out, int poll_ms = 1) {
    Pending<R>* pend = new Pending<R>;                 // (shared with the helper: freed by whoever finishes last)
    try {
        std::thread([pend, produce, drop] {
            pend->result = produce();
            int expect = PENDING;
            if (!pend->state.compare_exchange_strong(expect, DONE, std::memory_order_acq_rel)) drop(pend->result);   // ABANDONED
            release(pend);
        }).detach();
    } catch (...) {
        delete pend;
        return -1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (pend->state.load(std::memory_order_acquire) != DONE) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
            int expect = PENDING;
            if (pend->state.compare_exchange_strong(expect, ABANDONED, std::memory_order_acq_rel)) {
                release(pend);
                return 1;
            }
            break;                                     // the helper finished in the same instant: state is DONE
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(poll_ms));
    }
    out = pend->result;
    release(pend);
    return 0;
}

}  // namespace ttm_handover
