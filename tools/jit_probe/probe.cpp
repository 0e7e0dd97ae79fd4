// hiprtc compile-time probe: can the fused kernel be compiled at run time for an arbitrary W?
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
static std::string slurp(const char *p) { std::ifstream f(p); std::stringstream s; s << f.rdbuf(); return s.str(); }
int main(int argc, char **argv) {
    std::string common = slurp("../../simd-minimizers_amd/csrc/mm_common.h");
    std::string impl = slurp("../../simd-minimizers_amd/csrc/mm_fused_impl.h");
    // drop the includes (hiprtc has its own built-ins) and the nested include of mm_common.h
    auto strip = [](std::string s) {
        size_t p;
        while ((p = s.find("#include")) != std::string::npos) s.erase(p, s.find('\n', p) - p);
        while ((p = s.find("#pragma once")) != std::string::npos) s.erase(p, 12);
        return s;
    };
    std::string src = "typedef unsigned char uint8_t; typedef unsigned short uint16_t; typedef unsigned int uint32_t;\n"
                      "typedef int int32_t; typedef unsigned long long uint64_t; typedef unsigned long uintptr_t;\n" +
                      strip(common) + strip(impl);
    const char *name = argc > 1 ? argv[1] : "mm::fused_kernel<18, true, true, 0, false, false>";
    hiprtcProgram prog;
    hiprtcCreateProgram(&prog, src.c_str(), "mm_fused_jit.hip", 0, nullptr, nullptr);
    hiprtcAddNameExpression(prog, name);
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    auto t0 = std::chrono::steady_clock::now();
    hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, 0); if (ls) hiprtcGetProgramLog(prog, &log[0]);
    printf("compile: %s in %.2f s\n%s\n", hiprtcGetErrorString(r), dt, log.substr(0, 3000).c_str());
    if (r != HIPRTC_SUCCESS) return 1;
    const char *lowered = nullptr; hiprtcGetLoweredName(prog, name, &lowered);
    size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
    printf("lowered: %s\ncode size: %zu\n", lowered, cs);
    { std::vector<char> code(cs); hiprtcGetCode(prog, code.data()); FILE *f = fopen("/tmp/jit.co", "wb"); fwrite(code.data(), 1, cs, f); fclose(f); }
    return 0;
}
