// knobs.h — experiment knobs and the debug trace of libmvs_hip.
//
// Product build: every knob IS its default, a compile-time constant — the library reads no tuning variable from the
// environment.  Diagnostics build (make EXPERIMENTS=1 -> -DMVS_EXPERIMENTS, used by scripts/*.py only): a knob is read from
// the environment ONCE, at its first use, and clamped to [lo, hi]; a malformed or out-of-range value falls back to the
// default.  Nothing is read per pass.
#ifndef MVS_KNOBS_H_
#define MVS_KNOBS_H_
#include <cstdlib>

#ifdef MVS_EXPERIMENTS
inline double mvs_knob_env(const char* name, double dflt, double lo, double hi) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    char* end = nullptr;
    const double v = strtod(e, &end);
    if (end == e || !(v >= lo) || !(v <= hi)) return dflt;
    return v;
}
#define MVS_KNOB(name, dflt, lo, hi) ([]() -> double { static const double v_ = mvs_knob_env(name, dflt, lo, hi); return v_; }())
#else
#define MVS_KNOB(name, dflt, lo, hi) (static_cast<double>(dflt))
#endif

// trace level of the host side: MVS_DEBUG_CG=1 (plans, verdicts, set-up laps) or 2 (+ residual histories) in the
// environment when the library is loaded; read once (api_deform.cpp), 0 otherwise.  It changes no result.
int mvs_debug_level();

#endif
