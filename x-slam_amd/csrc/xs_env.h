// xs_env.h — tuning and A/B switches read from the environment.  They exist in an -DXS_EXPERIMENTS build only (what the scripts under
// profiles/tools/ build: `make EXTRAFLAGS=-DXS_EXPERIMENTS`); the product library compiles every one of them to its default, so nothing
// a caller's environment holds changes what libxslam_hip.so does.  The switches the product library does read are listed in
// INTEGRATION.md ("Environment") and go through xs::product_env_*.
#pragma once
#include <stdlib.h>
#include <string.h>

namespace xs {
#ifdef XS_EXPERIMENTS
inline int exp_env_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
inline float exp_env_float(const char *name, float dflt) { const char *v = getenv(name); return v ? (float)atof(v) : dflt; }
inline bool exp_env_set(const char *name) { return getenv(name) != nullptr; }
inline const char *exp_env_str(const char *name) { return getenv(name); }
#else
inline int exp_env_int(const char *, int dflt) { return dflt; }
inline float exp_env_float(const char *, float dflt) { return dflt; }
inline bool exp_env_set(const char *) { return false; }
inline const char *exp_env_str(const char *) { return nullptr; }
#endif
// documented switches of the product library
inline int product_env_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
inline bool product_env_set(const char *name) { return getenv(name) != nullptr; }
}  // namespace xs
