// bounded.h — host-side waits of libgpuart_hip.so that may depend on somebody else arriving, made finite (pure host code).
//
// The multi-GPU read-out (gather.h; the step the reference performs with its ptracingNormalize draw, src/renderer.cpp:601-616,
// inside the render loop of src/main.cpp:549-599) has calls that return only when every rank — or RCCL's own bootstrap — plays
// along: ncclCommInitAll / ncclCommInitRank, ncclGroupEnd (RCCL connects peers lazily, inside the first group that names them),
// ncclCommDestroy. None of them takes a timeout. Two tools:
//
//   * bounded(): runs such a call on a helper thread and waits for it with a bound. When the bound runs out the caller gets
//     GPUART_HIP_ERR_TIMEOUT and the name of the call; the helper thread stays parked in the call (it cannot be cancelled), the
//     communicator layer is marked stuck — every later RCCL-facing entry point fails at once instead of queueing behind it — and
//     nothing the parked call may still touch is freed. What the process does next is the caller's business (gpuart_cli and
//     bench.py report and _exit; nothing is re-executed in a process that has touched the GPU).
//   * the phase watchdog (gpuart_hip_phase_begin / _end of include/gpuart_hip.h): the caller names what it is about to do and
//     how long that may take; a watcher thread that sees a phase outlive its bound prints the phase, how long it has been
//     running and the library's most recent error messages of ALL threads, then _exit()s with GPUART_HIP_WATCHDOG_EXIT. For
//     whatever bounded() does not wrap (a stuck hipMalloc, a driver call, torch.distributed's own rendezvous in bench.py).
#pragma once
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

namespace bounded_ns {

// ---- the library's recent errors, of every thread (gpuart_hip_last_error is per thread) -------------------------------------
struct ErrorLog {
    std::mutex m;
    std::string msg[8];
    double at[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long n = 0;
};
inline ErrorLog &error_log() { static ErrorLog *l = new ErrorLog(); return *l; }  // (never destroyed: the watcher may run during exit)
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline void log_error(int code, const std::string &msg) {
    ErrorLog &l = error_log();
    std::lock_guard<std::mutex> g(l.m);
    l.msg[l.n % 8] = "[" + std::to_string(code) + "] " + msg;
    l.at[l.n % 8] = now_s();
    l.n++;
}
inline std::string recent_errors() {
    ErrorLog &l = error_log();
    std::lock_guard<std::mutex> g(l.m);
    if (!l.n) return "  (none)\n";
    std::string s;
    const double t = now_s();
    for (unsigned long long k = l.n > 8 ? l.n - 8 : 0; k < l.n; k++) {
        char age[48];
        snprintf(age, sizeof age, "  %.1f s ago: ", t - l.at[k % 8]);
        s += age + l.msg[k % 8] + "\n";
    }
    return s;
}

/// cv.wait_for(lk, timeout, pred) on the STEADY clock, in slices of at most 100 ms of pthread_cond_timedwait (system clock): a wall-clock
/// jump costs one slice, not the bound — and the sanitizers of this toolchain intercept pthread_cond_timedwait but not the
/// pthread_cond_clockwait that std::condition_variable::wait_for(steady) compiles to (tools/sanitize_bounded.sh would drown in false reports).
template <class Pred>
inline bool wait_steady(std::condition_variable &cv, std::unique_lock<std::mutex> &lk, double seconds, Pred pred) {
    const double end = now_s() + seconds;
    while (!pred()) {
        const double left = end - now_s();
        if (left <= 0) return false;
        cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::system_clock::duration>(std::chrono::duration<double>(left < 0.1 ? left : 0.1)));
    }
    return true;
}

// ---- calls that cannot be given a timeout, run beside a bounded wait ---------------------------------------------------------
inline std::atomic<bool> &stuck() { static std::atomic<bool> s{false}; return s; }
inline std::string &stuck_in() { static std::string *s = new std::string(); return *s; }  // written once, before stuck() is set

struct Outcome {
    bool timed_out = false;
    int rc = 0;            ///< what the call returned (its own convention)
    std::string detail;    ///< what the call wants the caller to know (an error string of its own)
};

/// bound on the RCCL calls that wait for peers / the bootstrap: GPUART_HIP_COMM_TIMEOUT_MS, default 120 s, 0 = no bound (call inline)
inline uint32_t comm_timeout_ms() {
    const char *v = getenv("GPUART_HIP_COMM_TIMEOUT_MS");
    if (!v) return 120000u;
    const long x = strtol(v, nullptr, 10);
    return x <= 0 ? 0u : (uint32_t)(x > 3600000 ? 3600000 : x);
}

/// Runs `fn` (which returns its rc and may fill a detail string) and waits at most timeout_ms for it. fn must own everything
/// it touches through captured VALUES or objects that are never freed once the layer is stuck: it may outlive the caller's frame.
/// One helper thread per process serves the calls (a gather makes two of them inside bench.py's timed region: starting a thread
/// for each would cost a rank of an 8-GPU run a few per cent of its 3 ms); a helper whose call never returned is abandoned with
/// it and the next call — the test hook's only: the real entry points refuse once the layer is stuck — gets a fresh one.
struct Helper {
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::function<int(std::string &)> job;
    unsigned long long posted = 0, done = 0;
    int rc = 0;
    std::string detail;
};
inline void helper_loop(std::shared_ptr<Helper> h) {
    std::unique_lock<std::mutex> lk(h->m);
    for (unsigned long long next = 1;; next++) {
        h->cv_job.wait(lk, [&] { return h->posted >= next; });
        std::function<int(std::string &)> fn = std::move(h->job);
        lk.unlock();
        std::string d;
        const int rc = fn(d);
        fn = nullptr;
        lk.lock();
        h->rc = rc; h->detail = std::move(d); h->done = next;
        h->cv_done.notify_all();
    }
}
inline Outcome bounded(const char *what, uint32_t timeout_ms, std::function<int(std::string &)> fn) {
    Outcome o;
    if (!timeout_ms) { o.rc = fn(o.detail); return o; }
    static std::mutex callers;                    // one bounded call at a time (they are rare and their order matters to RCCL anyway)
    static std::shared_ptr<Helper> *helper = new std::shared_ptr<Helper>();
    std::lock_guard<std::mutex> one(callers);
    if (!*helper) {
        *helper = std::make_shared<Helper>();
        std::thread(helper_loop, *helper).detach();
    }
    std::shared_ptr<Helper> h = *helper;
    std::unique_lock<std::mutex> lk(h->m);
    h->job = std::move(fn);
    const unsigned long long mine = ++h->posted;
    h->cv_job.notify_all();
    if (wait_steady(h->cv_done, lk, timeout_ms / 1000.0, [&] { return h->done >= mine; })) {
        o.rc = h->rc; o.detail = h->detail;
        return o;
    }
    helper->reset();  // that helper stays parked in the call (it keeps its own reference); it is never given another job
    o.timed_out = true;
    o.detail = std::string(what) + " has not returned after " + std::to_string(timeout_ms) + " ms (GPUART_HIP_COMM_TIMEOUT_MS); its thread stays "
               "parked in the call, the communicator layer of this process is out of service";
    if (!stuck().load()) { stuck_in() = what; stuck().store(true); }
    return o;
}

// ---- phase watchdog ---------------------------------------------------------------------------------------------------------
#define GPUART_HIP_WATCHDOG_EXIT_CODE 86
struct Watchdog {
    std::mutex m;
    std::condition_variable cv;
    bool running = false;
    bool armed = false;
    std::string phase;
    double begun = 0, bound_s = 0;
    bool log = false;
};
inline Watchdog &watchdog() { static Watchdog *w = new Watchdog(); return *w; }

inline void watch_loop() {
    Watchdog &w = watchdog();
    std::unique_lock<std::mutex> lk(w.m);
    for (;;) {
        if (!w.armed) { w.cv.wait(lk); continue; }
        const double left = w.begun + w.bound_s - now_s();
        if (left > 0) {  // (woken early by every begin / end; a slice of at most 250 ms otherwise)
            w.cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::system_clock::duration>(std::chrono::duration<double>(left < 0.25 ? left : 0.25)));
            continue;
        }
        // the phase has outlived its bound: say so and end the process (no unwinding: whatever is stuck would be waited for again)
        char head[512];
        snprintf(head, sizeof head, "gpuart watchdog (pid %d): phase '%s' has been running for %.1f s (bound %.1f s) — giving up.\n"
                                    "most recent errors of libgpuart_hip.so, all threads:\n",
                 (int)getpid(), w.phase.c_str(), now_s() - w.begun, w.bound_s);
        std::string out = head + recent_errors();
        if (stuck().load()) out += "the communicator layer is stuck in: " + stuck_in() + "\n";
        (void)!write(2, out.data(), out.size());
        _exit(GPUART_HIP_WATCHDOG_EXIT_CODE);
    }
}

inline int phase_begin(const char *name, uint32_t timeout_ms) {
    Watchdog &w = watchdog();
    std::lock_guard<std::mutex> g(w.m);
    if (!w.running) {
        w.running = true;
        const char *v = getenv("GPUART_HIP_PHASE_LOG");
        w.log = !v || atoi(v) != 0;  // default on: a phase line costs nothing and is what a post-mortem needs
        std::thread(watch_loop).detach();
    }
    w.phase = name ? name : "?";
    w.begun = now_s();
    w.bound_s = timeout_ms / 1000.0;
    w.armed = timeout_ms != 0;
    if (w.log) fprintf(stderr, "gpuart phase begin: %s (pid %d, bound %.1f s)\n", w.phase.c_str(), (int)getpid(), w.bound_s);
    w.cv.notify_all();
    return 0;
}

inline int phase_end() {
    Watchdog &w = watchdog();
    std::lock_guard<std::mutex> g(w.m);
    if (w.log && !w.phase.empty()) fprintf(stderr, "gpuart phase end:   %s (%.3f s)\n", w.phase.c_str(), now_s() - w.begun);
    w.armed = false;
    w.phase.clear();
    w.cv.notify_all();
    return 0;
}

}  // namespace bounded_ns
