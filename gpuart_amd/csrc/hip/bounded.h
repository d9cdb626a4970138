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
#include <pthread.h>
#include <time.h>
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

/// A condition variable whose timed wait runs on CLOCK_MONOTONIC: a wall clock stepped in EITHER direction (NTP, an administrator)
/// neither shortens nor stretches a bound. pthread_cond_t with pthread_condattr_setclock rather than std::condition_variable:
/// libstdc++'s wait_until(system_clock) is an absolute CLOCK_REALTIME deadline — a clock stepped backwards blocks for the size of the
/// step (round 5's version sliced such waits, which bounded a forward step only) — and its wait_for(steady) compiles to
/// pthread_cond_clockwait, which the sanitizers of this toolchain do not intercept (tools/sanitize_bounded.sh would drown in false
/// reports); pthread_cond_timedwait they do.
struct MonoCond {
    pthread_cond_t c;
    MonoCond() {
        pthread_condattr_t a;
        pthread_condattr_init(&a);
        pthread_condattr_setclock(&a, CLOCK_MONOTONIC);
        pthread_cond_init(&c, &a);
        pthread_condattr_destroy(&a);
    }
    ~MonoCond() { pthread_cond_destroy(&c); }
    MonoCond(const MonoCond &) = delete;
    MonoCond &operator=(const MonoCond &) = delete;
    void notify_all() { pthread_cond_broadcast(&c); }
    void wait(std::unique_lock<std::mutex> &lk) { pthread_cond_wait(&c, lk.mutex()->native_handle()); }
    template <class Pred> void wait(std::unique_lock<std::mutex> &lk, Pred pred) { while (!pred()) wait(lk); }
    /// woken or `seconds` of CLOCK_MONOTONIC later, whichever comes first (spurious wake-ups as usual: callers re-check)
    void wait_for(std::unique_lock<std::mutex> &lk, double seconds) {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        const double whole = seconds < 0 ? 0 : (seconds > 1e8 ? 1e8 : seconds);
        ts.tv_sec += (time_t)whole;
        ts.tv_nsec += (long)((whole - (double)(time_t)whole) * 1e9);
        if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
        pthread_cond_timedwait(&c, lk.mutex()->native_handle(), &ts);
    }
};

/// Waits until pred() holds or `seconds` of the monotonic clock have passed (false).
template <class Pred>
inline bool wait_steady(MonoCond &cv, std::unique_lock<std::mutex> &lk, double seconds, Pred pred) {
    const double end = now_s() + seconds;
    while (!pred()) {
        const double left = end - now_s();
        if (left <= 0) return false;
        cv.wait_for(lk, left);
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
/// One helper thread per CALLING thread serves its calls (a gather makes two of them inside bench.py's timed region: starting a thread
/// for each would cost a rank of an 8-GPU run a few per cent of its 3 ms). Per calling thread, not per process: a process that drives
/// its ranks from one thread each (RCCL's other usage model besides one process per GPU) has every rank inside ncclCommInitRank or the
/// share exchange at the same time, each waiting for the others — round 5's single helper behind one lock let the first rank in and
/// kept the peers it was waiting for outside (tests/test_gather_inprocess.py, the first time that form ran). RCCL's group state is per
/// thread, and a calling thread's calls keep their order on its helper. A helper whose call never returned is abandoned with it — the
/// next call (the test hook's only: the real entry points refuse once the layer is stuck) gets a fresh one —; a helper whose calling
/// thread ends is told to end.
struct Helper {
    std::mutex m;
    std::condition_variable cv_job;
    MonoCond cv_done;
    std::function<int(std::string &)> job;
    unsigned long long posted = 0, done = 0;
    bool quit = false;
    int rc = 0;
    std::string detail;
};
inline void helper_loop(std::shared_ptr<Helper> h) {
    std::unique_lock<std::mutex> lk(h->m);
    for (unsigned long long next = 1;; next++) {
        h->cv_job.wait(lk, [&] { return h->posted >= next || h->quit; });
        if (h->posted < next) return;
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
struct HelperOwner {
    std::shared_ptr<Helper> h;
    ~HelperOwner() {
        if (!h) return;
        std::lock_guard<std::mutex> g(h->m);
        h->quit = true;
        h->cv_job.notify_all();
    }
};
inline Outcome bounded(const char *what, uint32_t timeout_ms, std::function<int(std::string &)> fn) {
    Outcome o;
    if (!timeout_ms) { o.rc = fn(o.detail); return o; }
    static thread_local HelperOwner own;
    if (!own.h) {
        own.h = std::make_shared<Helper>();
        std::thread(helper_loop, own.h).detach();
    }
    std::shared_ptr<Helper> h = own.h;
    std::unique_lock<std::mutex> lk(h->m);
    h->job = std::move(fn);
    const unsigned long long mine = ++h->posted;
    h->cv_job.notify_all();
    if (wait_steady(h->cv_done, lk, timeout_ms / 1000.0, [&] { return h->done >= mine; })) {
        o.rc = h->rc; o.detail = h->detail;
        return o;
    }
    lk.unlock();
    own.h.reset();  // that helper stays parked in the call (it keeps its own reference); it is never given another job, nor told to end
    o.timed_out = true;
    o.detail = std::string(what) + " has not returned after " + std::to_string(timeout_ms) + " ms (GPUART_HIP_COMM_TIMEOUT_MS); its thread stays "
               "parked in the call, the communicator layer of this process is out of service";
    static std::mutex first;  // (several ranks' calls may run out at once: one of them names the layer's state)
    std::lock_guard<std::mutex> g(first);
    if (!stuck().load()) { stuck_in() = what; stuck().store(true); }
    return o;
}

// ---- phase watchdog ---------------------------------------------------------------------------------------------------------
#define GPUART_HIP_WATCHDOG_EXIT_CODE 86
struct Watchdog {
    std::mutex m;
    MonoCond cv;
    bool running = false;
    bool armed = false;
    std::string phase;
    double begun = 0, bound_s = 0;
    bool log = false, log_known = false;
};
inline Watchdog &watchdog() { static Watchdog *w = new Watchdog(); return *w; }

inline void watch_loop() {
    Watchdog &w = watchdog();
    std::unique_lock<std::mutex> lk(w.m);
    for (;;) {
        if (!w.armed) { w.cv.wait(lk); continue; }
        const double left = w.begun + w.bound_s - now_s();
        if (left > 0) {  // (woken early by every begin / end)
            w.cv.wait_for(lk, left);
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
        if (!w.log_known) {
            const char *v = getenv("GPUART_HIP_PHASE_LOG");
            w.log = !v || atoi(v) != 0;  // default on: a phase line is what a post-mortem needs
            w.log_known = true;
        }
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

/// Phase lines on (1) / off (0) from here on, whatever GPUART_HIP_PHASE_LOG said; returns what it was. For phases inside a timed
/// region (bench.py's gathers: two fprintf under a lock per gather, where a rank's whole job is ~3 ms): the watchdog stays armed.
inline int phase_log(int on) {
    Watchdog &w = watchdog();
    std::lock_guard<std::mutex> g(w.m);
    if (!w.log_known) {
        const char *v = getenv("GPUART_HIP_PHASE_LOG");
        w.log = !v || atoi(v) != 0;
        w.log_known = true;
    }
    const int was = w.log ? 1 : 0;
    w.log = on != 0;
    return was;
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
