// lab_env.hpp — the lab's A/B and tuning switches.  A default build reads none of them (the call is a constant nullptr and the
// branches behind it fold away); builds with -DFSGPU_EXPERIMENTS (FSGPU_BUILD_DEFS, frankensearch_amd/build.py) read them
// from the environment.  The switches a default build does read are listed in include/fsgpu.h.
#pragma once

#include <cstdlib>

namespace fsgpu {

#ifdef FSGPU_EXPERIMENTS
inline const char* lab_env(const char* name) { return std::getenv(name); }
#else
inline const char* lab_env(const char*) { return nullptr; }
#endif

}  // namespace fsgpu
