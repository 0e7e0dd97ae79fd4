// mm_jit.hip — run-time specialisation of the fused kernel for window sizes without a prebuilt
// instance.  The kernel source (mm_common.h + mm_fused_impl.h, embedded at build time) is compiled
// with hiprtc for the requested <W, CANON, HASH_RC, MODE, SK, READS>, loaded as a HIP module and
// cached per process and on disk, so every w runs the same single-pass kernel as the prebuilt
// window sizes (the reference is generic over w at full speed; so is this).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "mm_env.h"
#include "mm_launch.h"

namespace mm {
namespace {

const char kCommonSrc[] =
#include "mm_common.inc"
    ;
const char kImplSrc[] =
#include "mm_fused_impl.inc"
    ;

// hiprtc brings its own HIP built-ins but no libc headers
const char kPrelude[] =
    "typedef unsigned char uint8_t; typedef unsigned short uint16_t; typedef unsigned int uint32_t;\n"
    "typedef int int32_t; typedef unsigned long long uint64_t; typedef unsigned long uintptr_t;\n";

std::string strip_includes(std::string s) {
    size_t p;
    while ((p = s.find("#include")) != std::string::npos) s.erase(p, s.find('\n', p) - p);
    while ((p = s.find("#pragma once")) != std::string::npos) s.erase(p, 12);
    return s;
}

const std::string &kernel_source() {
    static const std::string src = std::string(kPrelude) + strip_includes(kCommonSrc) + strip_includes(kImplSrc);
    return src;
}

uint64_t fnv1a(const std::string &s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

std::string cache_dir() {
    if (const char *d = getenv("MM_JIT_CACHE_DIR")) return *d ? std::string(d) : std::string();
    std::string base;
    if (const char *x = getenv("XDG_CACHE_HOME")) base = x;
    else if (const char *h = getenv("HOME")) base = std::string(h) + "/.cache";
    if (base.empty()) return std::string();
    return base + "/simd_minimizers_amd";
}

// The cache holds code this process will load and run.  An entry is read only from a directory that is a
// real directory (no symlink) owned by this user and writable by nobody else, and only if the entry itself
// is a regular file owned by this user that nobody else can write; the checks run on the descriptor the
// bytes are then read from (O_NOFOLLOW + fstat), so nothing can be swapped between check and read.
bool trusted_dir(const std::string &dir) {
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0) return false;
    return S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}

bool read_trusted_file(const std::string &path, std::vector<char> &out) {
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() &&
              (st.st_mode & (S_IWGRP | S_IWOTH)) == 0 && st.st_size > 0 && st.st_size < (off_t)(1ll << 30);
    if (ok) {
        out.resize((size_t)st.st_size);
        size_t got = 0;
        while (got < out.size()) {
            const ssize_t n = read(fd, out.data() + got, out.size() - got);
            if (n <= 0) break;
            got += (size_t)n;
        }
        ok = got == out.size();
    }
    close(fd);
    return ok;
}

void write_file_atomic(const std::string &dir, const std::string &path, const std::vector<char> &data) {
    mkdir(dir.substr(0, dir.find_last_of('/')).c_str(), 0755);
    mkdir(dir.c_str(), 0700);
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid()) return;  // not ours: no cache
    if (st.st_mode & (S_IWGRP | S_IWOTH)) chmod(dir.c_str(), 0700);
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
}

std::mutex g_mu;
std::map<std::string, hipFunction_t> g_functions;  // "<device>:<kernel name>"
std::map<std::string, std::string> g_failed;       // kernel name -> why (do not retry)

// "gfx950" of the current device's gcnArchName ("gfx950:sramecc+:xnack-")
std::string device_arch() {
    int device = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) return "gfx950";
    std::string a = prop.gcnArchName;
    const size_t colon = a.find(':');
    if (colon != std::string::npos) a.resize(colon);
    return a.empty() ? std::string("gfx950") : a;
}

bool compile(const std::string &name, std::vector<char> &code, std::string &lowered, std::string *err) {
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, kernel_source().c_str(), "mm_fused_jit.hip", 0, nullptr, nullptr) !=
        HIPRTC_SUCCESS) {
        *err = "hiprtcCreateProgram failed";
        return false;
    }
    hiprtcAddNameExpression(prog, name.c_str());
    const std::string threads = "-DMM_FUSED_THREADS=" + std::to_string(kFusedThreads);
    std::vector<std::string> extra;  // MM_JIT_DEFS: extra -D options (tuning experiments; experiments build only)
#ifdef MM_EXPERIMENTS
    extra.push_back("-DMM_EXPERIMENTS");  // (kernels that read FusedParams::debug)
#endif
    if (const char *d = mm_exp_env("MM_JIT_DEFS")) {
        std::string cur;
        for (const char *c = d;; ++c) {
            if (*c == ' ' || *c == '\0') {
                if (!cur.empty()) extra.push_back(cur);
                cur.clear();
                if (!*c) break;
            } else {
                cur.push_back(*c);
            }
        }
    }
    const std::string arch = "--offload-arch=" + device_arch();
    std::vector<const char *> opts = {arch.c_str(), "-O3", "-std=c++17", threads.c_str()};
    for (const std::string &e : extra) opts.push_back(e.c_str());
    const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        *err = std::string("hiprtc: ") + hiprtcGetErrorString(r) + "\n" + log.substr(0, 2000);
        hiprtcDestroyProgram(&prog);
        return false;
    }
    const char *low = nullptr;
    size_t cs = 0;
    if (hiprtcGetLoweredName(prog, name.c_str(), &low) != HIPRTC_SUCCESS || !low ||
        hiprtcGetCodeSize(prog, &cs) != HIPRTC_SUCCESS || cs == 0) {
        *err = "hiprtc: no code object";
        hiprtcDestroyProgram(&prog);
        return false;
    }
    lowered = low;
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return true;
}

}  // namespace

bool jit_enabled() {
    const char *e = mm_env("MM_JIT");
    return !(e && e[0] == '0');
}

std::string fused_kernel_name(uint32_t w, bool canon, bool hash_rc, int mode, bool sk, bool reads, bool walk) {
    auto b = [](bool x) { return x ? "true" : "false"; };
    return std::string(walk ? "mm::walk_kernel<" : "mm::fused_kernel<") + std::to_string(w) + ", " + b(canon) + ", " + b(hash_rc) + ", " +
           std::to_string(mode) + ", " + b(sk) + ", " + b(reads) + ">";
}

// Compiled-and-loaded kernel for the current device, or nullptr (then *err says why).
hipFunction_t jit_fused_kernel(uint32_t w, bool canon, bool hash_rc, int mode, bool sk, bool reads,
                               std::string *err, bool walk) {
    std::string local_err;
    if (!err) err = &local_err;
    if (!jit_enabled() || w == 0 || w > kJitMaxW) {
        *err = "run-time specialisation disabled or w out of its range";
        return nullptr;
    }
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) {
        *err = "hipGetDevice failed";
        return nullptr;
    }
    const std::string name = fused_kernel_name(w, canon, hash_rc, mode, sk, reads, walk);
    const char *user_defs = mm_exp_env("MM_JIT_DEFS");
#ifdef MM_EXPERIMENTS
    const std::string defs_s = std::string("-DMM_EXPERIMENTS ") + (user_defs ? user_defs : "");
    const char *defs = defs_s.c_str();
#else
    const char *defs = user_defs;  // (always null: the product compiles the source as it is)
#endif
    const std::string key = std::to_string(device) + ":" + name + "|" + (defs ? defs : "");
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_functions.find(key);
    if (it != g_functions.end()) return it->second;
    auto bad = g_failed.find(key);
    if (bad != g_failed.end()) {
        *err = bad->second;
        return nullptr;
    }

    // disk cache: <hash of source, name, compiler>.hsaco + .name (the lowered symbol)
    int rtc_major = 0, rtc_minor = 0;
    hiprtcVersion(&rtc_major, &rtc_minor);
    const uint64_t h = fnv1a(name + "|" + (defs ? defs : "") + "|" + device_arch() + "|" + std::to_string(rtc_major) + "." + std::to_string(rtc_minor) +
                             "|" + std::to_string(kFusedThreads), fnv1a(kernel_source()));
    char hex[32];
    snprintf(hex, sizeof hex, "%016llx", (unsigned long long)h);
    const std::string dir = cache_dir();
    const std::string path = dir.empty() ? std::string() : dir + "/" + hex + ".hsaco";
    std::vector<char> code, lowered_buf;
    std::string lowered;
    bool from_disk = false;
    if (!path.empty() && trusted_dir(dir) && read_trusted_file(path, code) &&
        read_trusted_file(path + ".name", lowered_buf)) {
        lowered.assign(lowered_buf.begin(), lowered_buf.end());
        from_disk = true;
    }
    if (!from_disk) {
        if (!compile(name, code, lowered, err)) {
            g_failed[key] = *err;
            return nullptr;
        }
        if (!path.empty()) {
            write_file_atomic(dir, path, code);
            write_file_atomic(dir, path + ".name", std::vector<char>(lowered.begin(), lowered.end()));
        }
    }
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    hipError_t e = hipModuleLoadData(&mod, code.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&fn, mod, lowered.c_str());
    if (e != hipSuccess && from_disk) {
        // a stale or damaged cache entry: compile afresh once
        code.clear();
        if (compile(name, code, lowered, err)) {
            write_file_atomic(dir, path, code);
            write_file_atomic(dir, path + ".name", std::vector<char>(lowered.begin(), lowered.end()));
            e = hipModuleLoadData(&mod, code.data());
            if (e == hipSuccess) e = hipModuleGetFunction(&fn, mod, lowered.c_str());
        }
    }
    if (e != hipSuccess || !fn) {
        *err = std::string("loading the specialised kernel failed: ") + hipGetErrorString(e);
        g_failed[key] = *err;
        return nullptr;
    }
    g_functions[key] = fn;
    return fn;
}

}  // namespace mm
