// trace.h — optional tracing of the C-ABI entries ("Side effects": the reference prints a clock() pair around
// PartRecog, R/Alignment/Alignment.cpp:46-52, and writes debug files; the library does neither — a caller that wants timing
// registers a callback, mvs_set_trace, and / or has every entry marked as a roctx range for rocprofv3, mvs_set_trace_roctx).
// MVS_TRACE() is the first statement of every compute entry; with tracing off it costs one load of a global.
#ifndef MVS_TRACE_H_
#define MVS_TRACE_H_
#include <chrono>

bool mvs_trace_on();
void mvs_trace_enter(const char* entry);
void mvs_trace_leave(const char* entry, double host_ms);

struct MvsTraceScope {
    const char* name;
    std::chrono::steady_clock::time_point t0;
    bool on;
    explicit MvsTraceScope(const char* n) : name(n), on(mvs_trace_on()) {
        if (on) { t0 = std::chrono::steady_clock::now(); mvs_trace_enter(name); }
    }
    ~MvsTraceScope() {
        if (on) mvs_trace_leave(name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};
#define MVS_TRACE() MvsTraceScope mvs_trace_scope_(__func__)

#endif
