// Runtime glue of libpmhip: error text, HIP graph capture/replay, HIP events.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "pm_common.h"

thread_local char pm_err_text[256] = "";
bool pm_ktag_on = false;
static thread_local char pm_ktag_text[160] = "";

void pm_ktagf(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(pm_ktag_text, sizeof(pm_ktag_text), fmt, ap);
    va_end(ap);
}
static thread_local char pm_kvar_text[64] = "";
void pm_kvarf(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(pm_kvar_text, sizeof(pm_kvar_text), fmt, ap);
    va_end(ap);
}
extern "C" void pm_kernel_names_enable(int on) { pm_ktag_on = on != 0; pm_ktag_text[0] = 0; pm_kvar_text[0] = 0; }
extern "C" const char* pm_last_kernel_name(void) { return pm_ktag_text; }
extern "C" const char* pm_last_kernel_variant(void) { return pm_kvar_text; }
extern "C" void pm_clear_kernel_name(void) { pm_ktag_text[0] = 0; pm_kvar_text[0] = 0; }

int pm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return PM_OK;
    snprintf(pm_err_text, sizeof(pm_err_text), "%s: %s", what, hipGetErrorString(e));
    return PM_ELAUNCH;
}

static int check(hipError_t e, const char* what) {
    if (e == hipSuccess) return PM_OK;
    snprintf(pm_err_text, sizeof(pm_err_text), "%s: %s", what, hipGetErrorString(e));
    return PM_ELAUNCH;
}

extern "C" const char* pm_strerror(int code) {
    switch (code) {
        case PM_OK: return "ok";
        case PM_EINVAL: return "invalid argument or unsupported shape";
        case PM_ELAUNCH: return "HIP launch/runtime error (see pm_last_error)";
        default: return "unknown error";
    }
}
extern "C" const char* pm_last_error(void) { return pm_err_text; }
extern "C" int pm_version(void) { return 1; }

extern "C" int pm_graph_begin(pm_stream_t stream) {
    return check(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
}
extern "C" int pm_graph_end(pm_stream_t stream, void** graph_exec) {
    if (!graph_exec) return PM_EINVAL;
    hipGraph_t graph = nullptr;
    int rc = check(hipStreamEndCapture((hipStream_t)stream, &graph), "hipStreamEndCapture");
    if (rc) return rc;
    hipGraphExec_t exec = nullptr;
    rc = check(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0), "hipGraphInstantiate");
    (void)hipGraphDestroy(graph);
    if (rc) return rc;
    *graph_exec = (void*)exec;
    return PM_OK;
}
extern "C" int pm_graph_launch(void* graph_exec, pm_stream_t stream) {
    if (!graph_exec) return PM_EINVAL;
    return check(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream), "hipGraphLaunch");
}
extern "C" int pm_graph_destroy(void* graph_exec) {
    if (!graph_exec) return PM_EINVAL;
    return check(hipGraphExecDestroy((hipGraphExec_t)graph_exec), "hipGraphExecDestroy");
}

extern "C" int pm_event_create(void** ev) {
    if (!ev) return PM_EINVAL;
    hipEvent_t e;
    int rc = check(hipEventCreate(&e), "hipEventCreate");
    if (!rc) *ev = (void*)e;
    return rc;
}
extern "C" int pm_event_record(void* ev, pm_stream_t stream) {
    if (!ev) return PM_EINVAL;
    return check(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream), "hipEventRecord");
}
extern "C" int pm_event_synchronize(void* ev) {
    if (!ev) return PM_EINVAL;
    return check(hipEventSynchronize((hipEvent_t)ev), "hipEventSynchronize");
}
extern "C" int pm_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
    if (!ev_start || !ev_stop || !ms) return PM_EINVAL;
    return check(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop), "hipEventElapsedTime");
}
extern "C" int pm_event_destroy(void* ev) {
    if (!ev) return PM_EINVAL;
    return check(hipEventDestroy((hipEvent_t)ev), "hipEventDestroy");
}
