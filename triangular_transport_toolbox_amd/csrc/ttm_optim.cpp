// ttm_optim.cpp - the optimiser loops of optimize() as host C++ behind the C ABI (include/ttm.h "optimisers").
//
// The reference hands every map component to scipy.optimize.minimize (TM:3108-3114 L-BFGS-B for separable maps,
// TM:3252-3257 BFGS for integrated-rectifier maps), one Python call per objective evaluation.  Here the same
// algorithm (csrc/ttm_lbfgsb.h) runs as a host loop that launches the device reduction, waits for the stream and
// reads the 1 + m sums from pinned memory: ~10 us per evaluation instead of ~80 us; csrc/ttm_bfgs.h is the same for
// SciPy's BFGS over the integrated-rectifier objective (ttm_optimize_integrated).  With a communicator the sums of
// all ranks are combined by ONE all-reduce of the fused [objective | gradient] buffer per evaluation (RCCL,
// ttm_allreduce_f64) before they are read - the sample-sharded optimisation of SURVEY.md section 8e.

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#ifndef TTM_HOST_ONLY          // (tests/hostemu compiles this file for the host: no streams, nothing to wait for)
#include <hip/hip_runtime.h>
#include <immintrin.h>
#endif

#include "../../include/ttm.h"
#include "ttm_bfgs.h"
#include "ttm_lbfgsb.h"

namespace {

#ifndef TTM_HOST_ONLY
// Poll *flag (pinned host memory, written by the device behind the results it announces) until it holds `mark`.
// Every status of the stream other than "not ready" ends the wait: an idle stream without the mark means the launch
// behind it failed, an error status (sticky launch failure, lost or reset device) that it never will arrive - results that
// were not written are never read.  The loads of the results that follow are ordered behind the load that saw the mark.
int poll_mark(const double* flag_, double mark, void* stream) {
    const std::atomic<double>* flag = reinterpret_cast<const std::atomic<double>*>(flag_);
    static_assert(sizeof(std::atomic<double>) == sizeof(double), "lock-free fp64 atomics expected");
    for (long spins = 0; flag->load(std::memory_order_acquire) != mark; ++spins) {
        if ((spins & 0xfffff) != 0xfffff) continue;
        const hipError_t st = hipStreamQuery((hipStream_t)stream);
        if (st == hipErrorNotReady) continue;
        if (st == hipSuccess) (void)hipStreamSynchronize((hipStream_t)stream);   // (idle: anything queued has been executed)
        if (flag->load(std::memory_order_acquire) != mark) return TTM_E_HIP;
        break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return TTM_OK;
}
#endif

// The self-validating results of ttm_objective_sep_cached_sent: the host fills the n slots with a bit pattern no arithmetic
// produces (arm_values), the finishing workgroup overwrites each slot with one 8-byte store, and the slots are polled until
// none holds the pattern - no completion mark, no drain on the device in front of it.  TTM_E_HIP: the stream went idle or
// failed without results, or the device gave up waiting for its own workgroups (the FAIL pattern).
const uint64_t kSentBits = 0x7FF4DEADBEEF0001ull, kSentFail = 0x7FF4DEADBEEF0002ull;
void arm_values(double* v_, int n) {
    std::atomic<uint64_t>* v = reinterpret_cast<std::atomic<uint64_t>*>(v_);
    for (int i = 0; i < n; ++i) v[i].store(kSentBits, std::memory_order_relaxed);
    std::atomic_thread_fence(std::memory_order_release);
}
int poll_values(const double* v_, int n, void* stream, bool* unanswered = nullptr) {
    if (unanswered) *unanswered = false;
    const std::atomic<uint64_t>* v = reinterpret_cast<const std::atomic<uint64_t>*>(v_);
    static_assert(sizeof(std::atomic<uint64_t>) == sizeof(double), "lock-free 64-bit atomics expected");
    auto pending = [&]() {
        for (int i = 0; i < n; ++i)
            if (v[i].load(std::memory_order_acquire) == kSentBits) return true;
        return false;
    };
    for (long spins = 0; pending(); ++spins) {
#ifndef TTM_HOST_ONLY
        if ((spins & 0xfffff) != 0xfffff) continue;
        const hipError_t st = hipStreamQuery((hipStream_t)stream);
        if (st == hipErrorNotReady) continue;
        if (st == hipSuccess) (void)hipStreamSynchronize((hipStream_t)stream);
        if (pending()) {
            if (unanswered) *unanswered = st == hipSuccess;      // (an idle stream and no results: nobody is going to answer)
            return TTM_E_HIP;
        }
#else
        (void)stream;
        return TTM_E_HIP;
#endif
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int i = 0; i < n; ++i)
        if (v[i].load(std::memory_order_relaxed) == kSentFail) return TTM_E_HIP;
    return TTM_OK;
}

// Completion of the work queued on `stream` so far: a mark written behind it into pinned host memory (*flag), polled
// here - a hipStreamSynchronize per evaluation costs ~12 us of host / driver latency on top of the ~13 us of device work.
int wait_for_mark(double* flag_, long& seq, void* stream) {
#ifndef TTM_HOST_ONLY
    const double mark = (double)(++seq);
    const int rc = ttm_signal(flag_, mark, stream);
    if (rc) return rc;
    return poll_mark(flag_, mark, stream);
#else
    (void)flag_; (void)seq; (void)stream;
    return TTM_OK;
#endif
}

// Independent component problems side by side (the reference's process pool over components, TM:2789-2845): worker
// threads draw tasks from a shared counter; each worker owns one HIP stream, so the reductions of different components
// overlap on the device and their completion polls overlap on the host.  run(t, stream) -> rc of task t.
template <class Run>
int run_batch(int ntasks, int nthreads, void* stream, Run run) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > ntasks) nthreads = ntasks;
    if (nthreads > 64) nthreads = 64;
    std::vector<int> rcs(ntasks, TTM_OK);
    // the error text of a failing task lives in the worker thread's own buffer (thread-local): the first failure's text is
    // copied here under a lock and republished on the calling thread before run_batch returns
    std::mutex err_lock;
    std::string err_text;
    auto note_failure = [&](int rc) {
        if (!rc) return;
        std::lock_guard<std::mutex> g(err_lock);
        if (err_text.empty()) err_text = ttm_last_error_string();
    };
    if (nthreads == 1) {
        for (int t = 0; t < ntasks; ++t) rcs[t] = run(t, stream);
    } else {
#ifndef TTM_HOST_ONLY
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return TTM_E_HIP;
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return TTM_E_HIP;    // the producers of the inputs are done
        static std::vector<hipStream_t> pool[64];                                         // per device; kept for the process
        static std::atomic_flag pool_lock = ATOMIC_FLAG_INIT;
        while (pool_lock.test_and_set(std::memory_order_acquire)) {}
        std::vector<hipStream_t>& streams = pool[dev & 63];
        bool ok = true;
        while ((int)streams.size() < nthreads && ok) {
            hipStream_t s2;
            ok = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess;
            if (ok) streams.push_back(s2);
        }
        std::vector<hipStream_t> mine(streams.begin(), streams.begin() + (ok ? nthreads : 0));
        pool_lock.clear(std::memory_order_release);
        if (!ok) return TTM_E_HIP;
#endif
        std::atomic<int> next(0);
        auto worker = [&](int w) {
#ifndef TTM_HOST_ONLY
            const bool on_device = hipSetDevice(dev) == hipSuccess;
            void* st = (void*)mine[w];
#else
            const bool on_device = true;
            void* st = stream;
            (void)w;
#endif
            for (int t; (t = next.fetch_add(1)) < ntasks;) {
                rcs[t] = on_device ? run(t, st) : (int)TTM_E_HIP;
                note_failure(rcs[t]);
            }
        };
        std::vector<std::thread> threads;
        for (int w = 1; w < nthreads; ++w) {
            try {
                threads.emplace_back(worker, w);
            } catch (...) {                                              // no more threads to be had: the ones running share the tasks
                break;
            }
        }
        worker(0);
        for (auto& th : threads) th.join();
    }
    for (int t = 0; t < ntasks; ++t)
        if (rcs[t]) {
            if (!err_text.empty()) ttm_set_error_string(err_text.c_str());
            return rcs[t];
        }
    return TTM_OK;
}

// launch(flag, mark) enqueues a reduction that writes *flag = mark behind its results; wait until it has
template <class Launch>
int objective_and_wait(double* flag_, long& seq, void* stream, Launch launch) {
    const double mark = (double)(++seq);
    const int rc = launch(flag_, mark);
    if (rc) return rc;
#ifndef TTM_HOST_ONLY
    return poll_mark(flag_, mark, stream);
#else
    (void)stream;
    return TTM_OK;
#endif
}

}  // namespace

extern "C" {

int ttm_lbfgsb_minimize(int32_t n, double* x, const double* lb, const double* ub, ttm_objective_cb fun, void* user,
                        int32_t maxiter, double* result) {
    if (n < 1 || !x || !fun) return TTM_E_ARG;
    std::vector<double> l(n), u(n);
    std::vector<int> nbd(n);
    for (int i = 0; i < n; ++i) {
        const bool hl = lb && lb[i] > -INFINITY, hu = ub && ub[i] < INFINITY;
        l[i] = hl ? lb[i] : 0.0;
        u[i] = hu ? ub[i] : 0.0;
        nbd[i] = hl ? (hu ? 2 : 1) : (hu ? 3 : 0);
    }
    ttm_opt::LbfgsbOptions opt;
    if (maxiter > 0) opt.maxiter = maxiter;
    const ttm_opt::LbfgsbResult r = ttm_opt::lbfgsb_minimize(
        n, x, l.data(), u.data(), nbd.data(), [&](const double* xx, double* f, double* g) { return fun(n, xx, f, g, user); }, opt);
    if (result) {
        result[0] = r.f; result[1] = r.pgnorm; result[2] = r.nit; result[3] = r.nfev; result[4] = r.status;
    }
    return r.status < 0 ? TTM_E_HIP : TTM_OK;
}

}  // extern "C"

namespace {

// the reduced separable problem of one component (TM:2978-3018) minimised by the L-BFGS-B loop; launch(cc, out, flag,
// mark, stream) enqueues the reduction of the sums for coefficients cc into out (flag != NULL: with the completion mark)
//
// ONE monotone term (m = 1: the filtering and smoothing maps of Examples C, most components of Markov-type maps): the sample
// sums are known in closed form once they have been taken at one point.  dS_n = dPsi_n c + delta dPsi_n = dPsi_n (c + delta)
// (TM:2990-2993), so sum_n log dS_n = N log(c + delta) + sum_n log dPsi_n and sum_n dPsi_n / dS_n = N / (c + delta): the
// first evaluation of the loop goes to the device, every later one is two host operations - the same function to
// rounding (1e-16 relative, like the order of a reduction), no launch, no round trip.  `delta` < 0 switches it off.
typedef int (*SentLaunch)(const double* cc, double* out_host, void* stream, void* user);

// what the evaluation server of a loop needs (ttm_objective_sep_server_start; cached derivative basis only)
struct ServerArgs { const double* dPsi; int64_t ldp, N; int32_t m; double delta; double* work; };
std::atomic<uint32_t> g_server_gen{0};

template <class Launch>
int optimize_separable_with(Launch launch, int32_t m, const double* A, const double* b, double Ntotal, const double* lb,
                            const double* ub, double* x, double* sums_dev, double* sums_host, ttm_comm* comm, void* stream,
                            int32_t maxiter, double* result, double delta = -1.0, SentLaunch sent = nullptr, void* sent_user = nullptr,
                            const double* pre_x = nullptr, const ServerArgs* server = nullptr) {
    struct Ctx {
        Launch& launch;
        const double *A, *b;
        double invN;
        double *sums_dev, *sums_host;
        ttm_comm* comm;
        void* stream;
        int rc;
        long seq;
        double delta, Nw, KN;                                // closed form of m = 1: weights N, sum_n log dPsi_n
        bool closed, have0;
        SentLaunch sent;                                     // the evaluation with self-validating results (no ticket, no mark)
        void* sent_user;
        const double* pre_x;                                 // an evaluation at this point is in flight already (its results
        const ServerArgs* server;                            // arrive in sums_host, armed); ONE launch answers every evaluation of the loop
        unsigned char* box;                                  //   its mailbox (fine-grained device memory, host-written), once it runs
        uint32_t gen, round;
    } c{launch, A, b, 1.0 / Ntotal, sums_dev, sums_host, comm, stream, 0, 0, delta, 0.0, 0.0,
        m == 1 && delta >= 0.0 && lb && lb[0] >= 0.0, false, comm ? nullptr : sent, sent_user, (sent && !comm) ? pre_x : nullptr,
        (sent && !comm && m > 1) ? server : nullptr, nullptr, 0, 0};
    if (!c.pre_x) sums_host[1 + m] = 0.0;                    // the completion mark (sums_host: >= 2 + m doubles)
    auto fun = [](int32_t n, const double* cc, double* f, double* g, void* user) -> int32_t {
        Ctx& c = *(Ctx*)user;
        // sums[0] = sum_n log dS_n, sums[1 + i] = sum_n dPsi_{n,i} / dS_n  (TM:2990-3006), over the local samples
        double* out = c.comm ? c.sums_dev : c.sums_host;
        bool have = false;
        if (c.pre_x) {                                       // the evaluation that was launched ahead: wait for it whatever it is good for
            have = memcmp(cc, c.pre_x, (size_t)n * 8) == 0;
            c.pre_x = nullptr;
            c.rc = poll_values(c.sums_host, 1 + n, c.stream);
            if (c.rc) return c.rc;
        }
#ifndef TTM_HOST_ONLY
        if (!have && c.server && !c.box) {                   // the first evaluation of the loop: start its server
            c.box = (unsigned char*)ttm_mailbox_acquire();
            if (c.box) {
                c.gen = ++g_server_gen;
                c.round = 0;
                const ServerArgs& a = *c.server;
                const int rc = ttm_objective_sep_server_start(a.dPsi, a.ldp, a.N, a.m, a.delta, a.work, c.sums_host, c.box, c.gen, c.stream);
                if (rc != TTM_OK) { ttm_mailbox_release(c.box); c.box = nullptr; }
            }
            if (!c.box) c.server = nullptr;                  // (no mailbox / a larger grid / switched off: a launch per evaluation)
        }
#endif
        if (have) {
        } else if (c.closed && c.have0 && cc[0] + c.delta > 0.0) {
            c.sums_host[0] = c.Nw * log(cc[0] + c.delta) + c.KN;
            c.sums_host[1] = c.Nw / (cc[0] + c.delta);
#ifndef TTM_HOST_ONLY
        } else if (c.box) {
            // request: the coefficients, then - behind a store fence: the mailbox is a write-combining mapping - the word that
            // announces them; the results arrive where a launch per evaluation puts them
            // (tests: TTM_SRV_TEST_STALL = k makes the host miss the server's 0.2 s in front of request k of every loop)
            static const int stall_at = [] { const char* e = getenv("TTM_SRV_TEST_STALL"); return e ? atoi(e) : 0; }();
            if (stall_at > 0 && (int)c.round + 1 == stall_at) std::this_thread::sleep_for(std::chrono::milliseconds(300));
            for (int attempt = 0;; ++attempt) {
                arm_values(c.sums_host, 1 + n);
                volatile double* bc = (volatile double*)(c.box + 8);
                for (int i = 0; i < n; ++i) bc[i] = cc[i];
                _mm_sfence();
                *(volatile uint64_t*)c.box = ((uint64_t)c.gen << 32) | (uint64_t)(++c.round);
                _mm_sfence();
                bool unanswered = false;
                c.rc = poll_values(c.sums_host, 1 + n, c.stream, &unanswered);
                if (!c.rc || !unanswered || attempt >= 2) break;
                // no answer and the stream idle: the server waited 0.2 s for this request (a host thread that was not scheduled) and
                // left - both regions of the rows armed, as it leaves them between requests.  A new one takes over.
                const ServerArgs& a = *c.server;
                c.gen = ++g_server_gen;
                c.round = 0;
                if (ttm_objective_sep_server_start(a.dPsi, a.ldp, a.N, a.m, a.delta, a.work, c.sums_host, c.box, c.gen, c.stream) != TTM_OK) break;
            }
            if (c.rc) return c.rc;
#endif
        } else if (c.sent) {
            arm_values(c.sums_host, 1 + n);
            c.rc = c.sent(cc, c.sums_host, c.stream, c.sent_user);
            if (c.rc) return c.rc;
            c.rc = poll_values(c.sums_host, 1 + n, c.stream);
            if (c.rc) return c.rc;
        } else if (!c.comm) {                                // results and completion mark from the reduction itself
            c.rc = objective_and_wait(c.sums_host + 1 + n, c.seq, c.stream,
                                      [&](double* flag, double mark) { return c.launch(cc, out, flag, mark, c.stream); });
            if (c.rc) return c.rc;
        } else {
            c.rc = c.launch(cc, out, (double*)nullptr, 0.0, c.stream);
            if (c.rc) return c.rc;
            c.rc = ttm_allreduce_f64(c.comm, c.sums_dev, 1 + n, TTM_OP_SUM, c.stream);
            if (c.rc) return c.rc;
#ifdef TTM_HOST_ONLY
            memcpy(c.sums_host, c.sums_dev, (size_t)(1 + n) * 8);
#else
            if (hipMemcpyAsync(c.sums_host, c.sums_dev, (size_t)(1 + n) * 8, hipMemcpyDeviceToHost, (hipStream_t)c.stream) != hipSuccess)
                return c.rc = TTM_E_HIP;
#endif
            c.rc = wait_for_mark(c.sums_host + 1 + n, c.seq, c.stream);
            if (c.rc) return c.rc;
        }
        if (c.closed && !c.have0 && cc[0] + c.delta > 0.0) {  // the point the closed form is anchored at
            const double nw = c.sums_host[1] * (cc[0] + c.delta), kn = c.sums_host[0] - nw * log(cc[0] + c.delta);
            if (nw > 0.0 && nw < 1.0e300 && kn == kn && kn > -1.0e300 && kn < 1.0e300) { c.Nw = nw; c.KN = kn; c.have0 = true; }
            else c.closed = false;                           // (a vanishing or negative derivative somewhere: the sums stay on the device)
        }
        // J = c'Ac/2 - sum log dS / N + c.b,  grad = Ac - sums/N + b   (TM:3008-3018)
        double quad = 0.0, lin = 0.0;
        for (int i = 0; i < n; ++i) {
            double ax = 0.0;
            for (int j = 0; j < n; ++j) ax += c.A[i * n + j] * cc[j];
            quad += cc[i] * ax;
            lin += cc[i] * c.b[i];
            g[i] = ax - c.sums_host[1 + i] * c.invN + c.b[i];
        }
        *f = quad / 2.0 - c.sums_host[0] * c.invN + lin;
        return 0;
    };
    const int rc = ttm_lbfgsb_minimize(m, x, lb, ub, fun, &c, maxiter, result);
#ifndef TTM_HOST_ONLY
    if (c.box) {                                             // the loop is over: the server leaves (and would by itself after 0.2 s)
        *(volatile uint64_t*)c.box = ((uint64_t)c.gen << 32) | 0xffffffffull;
        _mm_sfence();
        ttm_mailbox_release(c.box);
    }
#endif
    return c.rc ? c.rc : rc;
}



bool closed_form_enabled() {
    static const bool closed = [] { const char* e = getenv("TTM_SEP_CLOSED_FORM"); return !e || atoi(e) != 0; }();
    return closed;
}

// ttm_optimize_separable; pre_x != NULL: the rows of `work` are armed and an evaluation at pre_x is in flight (results: sums_host)
int optimize_separable_cached(const double* dPsi, int64_t ldp, int64_t N, int32_t m, const double* A, const double* b, double Ntotal,
                              double delta, const double* lb, const double* ub, double* x, double* work, uint32_t* counter,
                              double* sums_dev, double* sums_host, ttm_comm* comm, void* stream, int32_t maxiter, double* result,
                              const double* pre_x, int32_t* armed = nullptr) {
    auto launch = [&](const double* cc, double* out, double* flag, double mark, void* st) {
        return ttm_objective_sep_cached_marked(dPsi, ldp, N, m, cc, delta, work, counter, out, flag, mark, st);
    };
    // up to 128 workgroups and no communicator: evaluations with self-validating partial sums and results (ttm_sentinel_fill
    // arms the rows once; every evaluation leaves them armed)
    struct SentArgs { const double* dPsi; int64_t ldp, N; int32_t m; double delta; double* work; } sa{dPsi, ldp, N, m, delta, work};
    SentLaunch sent = nullptr;
    // (armed: the caller vouches that a previous loop on this `work` left the rows armed for the same m and N - no fill launch)
    if (!comm && (pre_x || (armed && *armed) || ttm_sentinel_fill(work, m, N, stream) == TTM_OK))
        sent = [](const double* cc, double* out_host, void* st, void* user) -> int {
            const SentArgs& a = *(const SentArgs*)user;
            return ttm_objective_sep_cached_sent(a.dPsi, a.ldp, a.N, a.m, cc, a.delta, a.work, out_host, st);
        };
    const ServerArgs srv{dPsi, ldp, N, m, delta, work};
    const int rc = optimize_separable_with(launch, m, A, b, Ntotal, lb, ub, x, sums_dev, sums_host, comm, stream, maxiter, result,
                                           closed_form_enabled() ? delta : -1.0, sent, &sa, pre_x, sent ? &srv : nullptr);
    if (armed) *armed = (sent && rc == TTM_OK) ? 1 : 0;
    return rc;
}

// one task of ttm_optimize_separable_batch on stream st
int run_task(ttm_sep_task& q, int64_t N, double Ntotal, double delta, void* st, int32_t maxiter) {
    if (q.dPsi)
        return q.rc = (!q.A || !q.b || !q.x || !q.work || !q.counter || !q.sums_host || q.m < 1 || !(Ntotal > 0.0))
                          ? (int)TTM_E_ARG
                          : optimize_separable_cached(q.dPsi, q.ldp, N, q.m, q.A, q.b, Ntotal, delta, q.lb, q.ub, q.x, q.work, q.counter, nullptr,
                                                      q.sums_host, nullptr, st, maxiter, q.result, nullptr, &q.armed);
    // derivative basis recomputed from the x_k column per evaluation
    if (!q.xk || !q.kinds || !q.pars || !q.A || !q.b || !q.x || !q.work || !q.counter || !q.sums_host || q.m < 1 || !(Ntotal > 0.0))
        return q.rc = TTM_E_ARG;
    auto launch = [&](const double* cc, double* out, double* flag, double mark, void* s2) {
        return ttm_objective_sep_direct_marked(q.xk, N, q.m, q.kinds, q.pars, cc, delta, q.work, q.counter, out, flag, mark, s2);
    };
    // (self-validating sums as in ttm_optimize_separable: the same finish, hence the same bits, as the cached basis)
    struct SentArgs { const ttm_sep_task* q; int64_t N; double delta; } sa{&q, N, delta};
    SentLaunch sent = nullptr;
    if (q.armed || ttm_sentinel_fill(q.work, q.m, N, st) == TTM_OK)
        sent = [](const double* cc, double* out_host, void* s2, void* user) -> int {
            const SentArgs& a = *(const SentArgs*)user;
            return ttm_objective_sep_direct_sent(a.q->xk, a.N, a.q->m, a.q->kinds, a.q->pars, cc, a.delta, a.q->work, out_host, s2);
        };
    q.rc = optimize_separable_with(launch, q.m, q.A, q.b, Ntotal, q.lb, q.ub, q.x, nullptr, q.sums_host, nullptr, st, maxiter,
                                   q.result, -1.0, sent, &sa);
    q.armed = (sent && q.rc == TTM_OK) ? 1 : 0;
    return q.rc;
}

}  // namespace

extern "C" {

int ttm_optimize_separable(const double* dPsi, int64_t ldp, int64_t N, int32_t m, const double* A, const double* b, double Ntotal,
                           double delta, const double* lb, const double* ub, double* x, double* work, uint32_t* counter,
                           double* sums_dev, double* sums_host, ttm_comm* comm, void* stream, int32_t maxiter, double* result) {
    if (!dPsi || !A || !b || !x || !work || !counter || !sums_host || m < 1 || N < 1 || !(Ntotal > 0.0)) return TTM_E_ARG;
    if (comm && !sums_dev) return TTM_E_ARG;
    return optimize_separable_cached(dPsi, ldp, N, m, A, b, Ntotal, delta, lb, ub, x, work, counter, sums_dev, sums_host, comm, stream, maxiter,
                                     result, nullptr);
}

int ttm_optimize_separable_batch(ttm_sep_task* tasks, int32_t ntasks, int64_t N, double Ntotal, double delta, int32_t nthreads,
                                 void* stream, int32_t maxiter) {
    if (!tasks || ntasks < 1 || N < 1) return TTM_E_ARG;
#ifndef TTM_HOST_ONLY
    // Components with ONE monotone term need one device evaluation (the closed form of optimize_separable_with takes over behind it).
    // When at most one other component is in the batch - the filter's maps: two such components and one with special terms - no
    // threads: the first evaluations of the one-term components are launched ahead on `stream`, the other component's loop runs
    // behind them, and the one-term loops find their sums in place (a thread per component cost ~40 us each to start and fought
    // the long loop for the runtime's launch lock).  The same evaluations at the same points: the same bits as the threaded batch.
    {
        std::vector<int> ahead, rest;
        for (int t = 0; t < ntasks; ++t) {
            const ttm_sep_task& q = tasks[t];
            const bool one = q.dPsi && q.m == 1 && closed_form_enabled() && delta >= 0.0 && q.lb && q.lb[0] >= 0.0 && q.A && q.b && q.x &&
                             q.work && q.counter && q.sums_host;
            (one ? ahead : rest).push_back(t);
        }
        if (!ahead.empty() && rest.size() <= 1 && Ntotal > 0.0) {
            std::vector<double> x0(ntasks, 0.0);
            std::vector<char> flying(ntasks, 0);
            int first_rc = TTM_OK;
            for (int t : ahead) {
                ttm_sep_task& q = tasks[t];
                double v = q.x[0];                                       // the start as lbfgsb_minimize projects it
                if (v <= q.lb[0]) v = q.lb[0];
                else if (q.ub && q.ub[0] < INFINITY && v >= q.ub[0]) v = q.ub[0];
                x0[t] = v;
                if (!q.armed && ttm_sentinel_fill(q.work, 1, N, stream) != TTM_OK) continue;
                arm_values(q.sums_host, 2);
                if (ttm_objective_sep_cached_sent(q.dPsi, q.ldp, N, 1, &x0[t], delta, q.work, q.sums_host, stream) == TTM_OK) flying[t] = 1;
            }
            auto run_one = [&](int t) {
                ttm_sep_task& q = tasks[t];
                if (!q.dPsi) return -1;
                q.rc = optimize_separable_cached(q.dPsi, q.ldp, N, q.m, q.A, q.b, Ntotal, delta, q.lb, q.ub, q.x, q.work, q.counter, nullptr,
                                                 q.sums_host, nullptr, stream, maxiter, q.result, flying[t] ? &x0[t] : nullptr, &q.armed);
                if (q.rc && !first_rc) first_rc = q.rc;
                return 0;
            };
            bool direct_rest = false;
            for (int t : rest) if (run_one(t) < 0) direct_rest = true;
            for (int t : ahead) run_one(t);
            if (!direct_rest) return first_rc;
            // (the other component recomputes its basis: it takes the general path below, alone)
            if (first_rc) return first_rc;
            const int t = rest[0];
            return run_batch(1, 1, stream, [&](int, void* st) { return run_task(tasks[t], N, Ntotal, delta, st, maxiter); });
        }
    }
#endif
    return run_batch(ntasks, nthreads, stream, [&](int t, void* st) { return run_task(tasks[t], N, Ntotal, delta, st, maxiter); });
}


// (Gnn + ridge I)^-1 Gnm by Cholesky on the diagonally equilibrated matrix with one step of iterative refinement - the host
// class's _normal_solve, ridge > 0.  G: the (n + m) x (n + m) Gram matrix, row-major; y: n x m.  false: not positive
// definite / not finite (the caller decides what then).
static bool sep_normal_solve(const double* G, int n, int m, double ridge, std::vector<double>& y) {
    const int ld = n + m;
    std::vector<double> d(n), Ms((size_t)n * n), U((size_t)n * n, 0.0), rhs((size_t)n * m), r((size_t)n * m);
    for (int i = 0; i < n; ++i) {
        d[i] = sqrt(G[i * ld + i] + ridge);
        if (!(d[i] > 0.0) || !(d[i] < INFINITY)) return false;
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            Ms[i * n + j] = ((G[i * ld + j] + (i == j ? ridge : 0.0)) / d[i]) / d[j];
            if (!(fabs(Ms[i * n + j]) < INFINITY)) return false;
        }
    for (int j = 0; j < n; ++j) {                       // Ms = U'U, U upper triangular
        double s = Ms[j * n + j];
        for (int k = 0; k < j; ++k) s -= U[k * n + j] * U[k * n + j];
        if (!(s > 0.0)) return false;
        const double ujj = sqrt(s);
        U[j * n + j] = ujj;
        for (int i = j + 1; i < n; ++i) {
            double t = Ms[j * n + i];
            for (int k = 0; k < j; ++k) t -= U[k * n + j] * U[k * n + i];
            U[j * n + i] = t / ujj;
        }
    }
    auto solve = [&](std::vector<double>& b) {          // U'U x = b, column by column, in place
        for (int c = 0; c < m; ++c) {
            for (int i = 0; i < n; ++i) {
                double t = b[i * m + c];
                for (int k = 0; k < i; ++k) t -= U[k * n + i] * b[k * m + c];
                b[i * m + c] = t / U[i * n + i];
            }
            for (int i = n - 1; i >= 0; --i) {
                double t = b[i * m + c];
                for (int k = i + 1; k < n; ++k) t -= U[i * n + k] * b[k * m + c];
                b[i * m + c] = t / U[i * n + i];
            }
        }
    };
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < m; ++c) {
            rhs[i * m + c] = G[i * ld + n + c] / d[i];
            if (!(fabs(rhs[i * m + c]) < INFINITY)) return false;
        }
    y = rhs;
    solve(y);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < m; ++c) {
            double t = rhs[i * m + c];
            for (int k = 0; k < n; ++k) t -= Ms[i * n + k] * y[k * m + c];
            r[i * m + c] = t;
        }
    solve(r);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < m; ++c) y[i * m + c] = (y[i * m + c] + r[i * m + c]) / d[i];
    return true;
}

// The reduced separable problem with L2 regularisation (TM:3021-3050, 3148-3169) from the Gram matrix of
// [Psi_nonmon | Psi_mon]: A (m x m) of the monotone coefficients' objective and the map c_mon -> c_nonmon = -sol c_mon
// (sol: n x m).  What transport_map.separable_setup computes with NumPy, for the optimiser batches of the filter and smoother
// updates (six small solves per update: 0.2 ms of Python).  TTM_E_UNSUPPORTED: a matrix is not positive definite - the
// caller's own dense solve takes over.
int ttm_separable_reduce_l2(const double* G, int32_t n, int32_t m, double lam, double* A, double* sol) {
    if (!G || !A || !sol || n < 1 || m < 1 || n > 256 || m > 64 || !(lam > 0.0)) return TTM_E_ARG;
    const int ld = n + m;
    std::vector<double> Gm, S2;
    if (!sep_normal_solve(G, n, m, lam, Gm) || !sep_normal_solve(G, n, m, 2.0 * lam, S2)) return TTM_E_UNSUPPORTED;
    // dd = Gmm - Gnm'Gm - Gm'Gnm + (Gm'Gnn) Gm;  A = dd / 2 + lam (Gm'Gm + I), symmetrised
    std::vector<double> W((size_t)m * n), dd((size_t)m * m);
    for (int a = 0; a < m; ++a)
        for (int j = 0; j < n; ++j) {
            double t = 0.0;
            for (int i = 0; i < n; ++i) t += Gm[i * m + a] * G[i * ld + j];
            W[a * n + j] = t;                           // (Gm'Gnn)[a][j]
        }
    for (int a = 0; a < m; ++a)
        for (int b = 0; b < m; ++b) {
            double t1 = 0.0, t2 = 0.0, t3 = 0.0, gg = 0.0;
            for (int i = 0; i < n; ++i) {
                t1 += G[i * ld + n + a] * Gm[i * m + b];
                t2 += Gm[i * m + a] * G[i * ld + n + b];
                t3 += W[a * n + i] * Gm[i * m + b];
                gg += Gm[i * m + a] * Gm[i * m + b];
            }
            dd[a * m + b] = (((G[(n + a) * ld + n + b] - t1) - t2) + t3) / 2.0 + lam * (gg + (a == b ? 1.0 : 0.0));
        }
    for (int a = 0; a < m; ++a)
        for (int b = 0; b < m; ++b) A[a * m + b] = (dd[a * m + b] + dd[b * m + a]) / 2.0;
    memcpy(sol, S2.data(), (size_t)n * m * 8);
    return TTM_OK;
}

int ttm_optimize_integrated_batch(const ttm_program* p, ttm_int_task* tasks, int32_t ntasks, const double* Xsoa, int64_t ldx, int64_t N,
                                  double Ntotal, int32_t nthreads, void* stream, int32_t maxiter) {
    if (!p || !tasks || ntasks < 1 || N < 1) return TTM_E_ARG;
    return run_batch(ntasks, nthreads, stream, [&](int t, void* st) {
        ttm_int_task& q = tasks[t];
        return q.rc = ttm_optimize_integrated(p, q.k, q.m, Xsoa, ldx, N, Ntotal, q.regularization, q.lambda, q.x, q.work, q.counter,
                                              nullptr, q.sums_host, nullptr, st, maxiter, q.result);
    });
}

int ttm_bfgs_minimize(int32_t n, double* x, ttm_objective_cb fun, void* user, int32_t maxiter, double* result) {
    if (n < 1 || !x || !fun) return TTM_E_ARG;
    ttm_opt::BfgsOptions opt;
    if (maxiter > 0) opt.maxiter = maxiter;
    const ttm_opt::BfgsResult r =
        ttm_opt::bfgs_minimize(n, x, [&](const double* xx, double* f, double* g) { return fun(n, xx, f, g, user); }, opt);
    if (result) {
        result[0] = r.f; result[1] = r.gnorm; result[2] = r.nit; result[3] = r.nfev; result[4] = r.status;
    }
    return r.status < 0 ? TTM_E_HIP : TTM_OK;
}

int ttm_optimize_integrated(const ttm_program* p, int32_t k, int32_t m, const double* Xsoa, int64_t ldx, int64_t N, double Ntotal,
                            int32_t regularization, const double* lambda, double* x, double* work, uint32_t* counter,
                            double* sums_dev, double* sums_host, ttm_comm* comm, void* stream, int32_t maxiter, double* result) {
    if (!p || !Xsoa || !x || !work || !counter || !sums_host || m < 1 || m > 128 || N < 1 || !(Ntotal > 0.0) || regularization < 0 ||
        regularization > 2 || (regularization && !lambda))
        return TTM_E_ARG;
    if (comm && !sums_dev) return TTM_E_ARG;
    struct Ctx {
        const ttm_program* p;
        int k;
        const double* Xsoa;
        int64_t ldx, N;
        double invN;
        int reg;
        const double* lam;
        double *work, *sums_dev, *sums_host;
        uint32_t* counter;
        ttm_comm* comm;
        void* stream;
        int rc;
        long seq;
    } c{p, (int)k, Xsoa, ldx, N, 1.0 / Ntotal, (int)regularization, lambda, work, sums_dev, sums_host, counter, comm, stream, 0, 0};
    sums_host[1 + m] = 0.0;                                  // the completion mark (sums_host: >= 2 + m doubles)
    auto fun = [](int32_t n, const double* cc, double* f, double* g, void* user) -> int32_t {
        Ctx& c = *(Ctx*)user;
        // sums[0] = sum_n of the objective's sample terms, sums[1 + i] = sum_n of their derivatives (TM:3300-3380, 3435-3573)
        double* out = c.comm ? c.sums_dev : c.sums_host;
        if (!c.comm) {
            c.rc = objective_and_wait(c.sums_host + 1 + n, c.seq, c.stream, [&](double* flag, double mark) {
                return ttm_objective_host_marked(c.p, c.k, cc, c.Xsoa, c.ldx, c.N, c.work, c.counter, out, flag, mark, c.stream);
            });
            if (c.rc) return c.rc;
        } else {
            c.rc = ttm_objective_host(c.p, c.k, cc, c.Xsoa, c.ldx, c.N, c.work, c.counter, out, c.stream);
            if (c.rc) return c.rc;
        }
        if (c.comm) {
            c.rc = ttm_allreduce_f64(c.comm, c.sums_dev, 1 + n, TTM_OP_SUM, c.stream);
            if (c.rc) return c.rc;
#ifdef TTM_HOST_ONLY
            memcpy(c.sums_host, c.sums_dev, (size_t)(1 + n) * 8);
#else
            if (hipMemcpyAsync(c.sums_host, c.sums_dev, (size_t)(1 + n) * 8, hipMemcpyDeviceToHost, (hipStream_t)c.stream) != hipSuccess)
                return c.rc = TTM_E_HIP;
#endif
            c.rc = wait_for_mark(c.sums_host + 1 + n, c.seq, c.stream);
            if (c.rc) return c.rc;
        }
        // mean over the ensemble + the penalty on the coefficients (TM:3382-3431, 3575-3633)
        double pen = 0.0;
        for (int i = 0; i < n; ++i) {
            double gi = c.sums_host[1 + i] * c.invN;
            if (c.reg == 1) {
                pen += c.lam[i] * fabs(cc[i]);
                gi += c.lam[i] * (cc[i] > 0.0 ? 1.0 : cc[i] < 0.0 ? -1.0 : 0.0);
            } else if (c.reg == 2) {
                pen += c.lam[i] * cc[i] * cc[i];
                gi += c.lam[i] * 2 * cc[i];
            }
            g[i] = gi;
        }
        *f = c.sums_host[0] * c.invN + pen;
        return 0;
    };
    const int rc = ttm_bfgs_minimize(m, x, fun, &c, maxiter, result);
    return c.rc ? c.rc : rc;
}

}  // extern "C"
