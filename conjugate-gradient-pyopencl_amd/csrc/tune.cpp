// Process-wide tuning configuration (cgamd_tune) and its per-handle / per-thread snapshots.
#include <functional>
#include <mutex>

#include "cgamd_internal.h"

namespace cgamd {

Tuning g_tune;
static std::mutex g_tune_mutex;
static thread_local const Tuning *t_tune = nullptr;
static thread_local Tuning t_tune_fallback;
Tuning tune_snapshot() {
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    return g_tune;
}
void tune_set(const std::function<void(Tuning &)> &edit) {
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    edit(g_tune);
}
const Tuning &tune() {
    if (t_tune) return *t_tune;
    t_tune_fallback = tune_snapshot();      // handle-less entry (stand-alone ops): the global configuration as of now
    return t_tune_fallback;
}
// HIP decides per CALLING thread whether an API call (hipMalloc, a synchronous copy, ...) made while some stream is being
// captured is an error that invalidates that capture: the default interaction mode of a thread is "global".  The
// reference's threading model has one thread per device working independently, each capturing its own iteration graphs,
// so every thread that enters the library switches its own mode to "relaxed" once (a worker's hipMalloc must not kill a
// sibling's capture: "operation failed due to a previous error during capture").
void thread_hip_setup() {
    static thread_local bool done = false;
    if (done) return;
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    done = true;
}
TuneScope::TuneScope(const Tuning *t) : prev(t_tune) { t_tune = t; thread_hip_setup(); }
TuneScope::~TuneScope() { t_tune = prev; }

}  // namespace cgamd
