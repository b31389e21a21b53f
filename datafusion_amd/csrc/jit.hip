// jit.hip — runtime specialisation of fused operator nodes.
//
// The fused FilterExec + ProjectionExec + AggregateExec node interprets its expression forest per row
// (rowprog.hpp).  Measured on MI355X (profiles/r1_q1_pmc.md) the interpreter is bound by per-instruction
// latency — operand fetch, scalar loads of the instruction words, dispatch branches — at about 12 % of HBM
// peak for TPC-H Q1.  For large inputs the node is therefore specialised at plan time: the hand-written HIP
// kernel skeleton (aggregate.hip, `agg_node_source`) gets the forest spliced in as straight-line typed code
// (RowProgramCompiler::finish emits it), is compiled once with hiprtc for gfx950, and is cached per process by
// its source text.  Small inputs (and any forest hiprtc rejects) keep using the interpreter.
#include <hip/hiprtc.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>

#include "internal.hpp"

namespace dfgpu {

struct JitEntry {
  hipModule_t module = nullptr;
  std::unordered_map<std::string, hipFunction_t> fns;
};
static std::mutex g_jit_mu;
static std::unordered_map<std::string, JitEntry> g_jit_cache;
static double g_jit_compile_ms = 0.0;
static int64_t g_jit_compiles = 0;

#define DFGPU_RTC(expr)                                                                                  \
  do {                                                                                                   \
    hiprtcResult _r = (expr);                                                                            \
    if (_r != HIPRTC_SUCCESS) throw ::dfgpu::Error(std::string("hiprtc error ") + hiprtcGetErrorString(_r) + " at " #expr); \
  } while (0)

hipFunction_t jit_get(const std::string& source, const char* kernel_name) {
  std::lock_guard<std::mutex> g(g_jit_mu);
  auto function_of = [&](JitEntry& e) {
    auto f = e.fns.find(kernel_name);
    if (f != e.fns.end()) return f->second;
    hipFunction_t fn = nullptr;
    DFGPU_HIP(hipModuleGetFunction(&fn, e.module, kernel_name));
    e.fns.emplace(kernel_name, fn);
    return fn;
  };
  auto it = g_jit_cache.find(source);
  if (it != g_jit_cache.end()) return function_of(it->second);
  auto t0 = std::chrono::steady_clock::now();
  if (const char* dump = std::getenv("DFGPU_JIT_DUMP")) {  // debugging aid: the generated sources, one file per node
    const std::string path = std::string(dump) + "/node_" + std::to_string(g_jit_compiles) + "_" + kernel_name + ".hip";
    if (FILE* f = fopen(path.c_str(), "w")) {
      fwrite(source.data(), 1, source.size(), f);
      fclose(f);
    }
  }
  hiprtcProgram prog;
  DFGPU_RTC(hiprtcCreateProgram(&prog, source.c_str(), "dfgpu_node.hip", 0, nullptr, nullptr));
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  hiprtcResult rc = hiprtcCompileProgram(prog, 3, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, log.data());
    hiprtcDestroyProgram(&prog);
    throw Error("hiprtc compilation failed: " + log);
  }
  size_t sz = 0;
  DFGPU_RTC(hiprtcGetCodeSize(prog, &sz));
  std::vector<char> code(sz);
  DFGPU_RTC(hiprtcGetCode(prog, code.data()));
  DFGPU_RTC(hiprtcDestroyProgram(&prog));
  JitEntry e;
  DFGPU_HIP(hipModuleLoadData(&e.module, code.data()));
  auto ins = g_jit_cache.emplace(source, e);
  g_jit_compile_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  g_jit_compiles++;
  return function_of(ins.first->second);
}

void jit_launch(hipFunction_t fn, int grid, int block, size_t lds_bytes, void* args, size_t args_bytes) {
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &args_bytes, HIP_LAUNCH_PARAM_END};
  DFGPU_HIP(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, (unsigned)lds_bytes, rt().stream, nullptr, config));
}

void jit_stats(int64_t* compiles, double* compile_ms) {
  std::lock_guard<std::mutex> g(g_jit_mu);
  *compiles = g_jit_compiles;
  *compile_ms = g_jit_compile_ms;
}

}  // namespace dfgpu

extern "C" int dfgpu_jit_stats(int64_t* compiles, double* compile_ms) {
  return dfgpu::guarded([&] { dfgpu::jit_stats(compiles, compile_ms); });
}
