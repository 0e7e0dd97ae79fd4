// mm_env.h — the MM_* environment switches, read once per process.
//
// A run used to call getenv() for every switch on every launch (eight or more scans of the environment: a few
// microseconds of a 36 us small-batch call).  mm_env("NAME") reads the variable at its first use and keeps the
// answer; MM_ENV_DYNAMIC=1 (set by tests/conftest.py and by the tuning scripts under tools/, which flip switches
// between runs of one process) restores the read-every-time behaviour.
#pragma once
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>

namespace mm {

inline const char *mm_env(const char *name) {
    static const bool dynamic = getenv("MM_ENV_DYNAMIC") != nullptr;
    if (dynamic) return getenv(name);
    struct Entry {
        bool set;
        std::string value;
    };
    static std::mutex mu;
    static std::unordered_map<std::string, Entry> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(name);
    if (it == cache.end()) {
        const char *v = getenv(name);
        it = cache.emplace(name, Entry{v != nullptr, v ? std::string(v) : std::string()}).first;
    }
    return it->second.set ? it->second.value.c_str() : nullptr;
}

// Switches that change RESULTS (timing experiments: MM_DEBUG, MM_JIT_DEFS, MM_FASTA_DEBUG, ...) or only serve the
// tuning scripts exist in the experiments build alone (-DMM_EXPERIMENTS -> libsimd_minimizers_amd_exp.so, loaded by
// tools/ and two tests through MM_LIB_PATH).  The product library never reads them: a variable leaked into a
// caller's environment cannot change what it computes.
inline const char *mm_exp_env(const char *name) {
#ifdef MM_EXPERIMENTS
    return mm_env(name);
#else
    (void)name;
    return nullptr;
#endif
}

}  // namespace mm
