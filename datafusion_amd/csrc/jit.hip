// jit.hip — runtime specialisation of fused operator nodes.
//
// The fused FilterExec + ProjectionExec + AggregateExec node interprets its expression forest per row
// (rowprog.hpp).  Measured on MI355X (profiles/r1_q1_pmc.md) the interpreter is bound by per-instruction
// latency — operand fetch, scalar loads of the instruction words, dispatch branches — at about 12 % of HBM
// peak for TPC-H Q1.  For large inputs the node is therefore specialised at plan time: the hand-written HIP
// kernel skeleton (aggregate.hip, `agg_node_source`) gets the forest spliced in as straight-line typed code
// (RowProgramCompiler::finish emits it), is compiled once with hiprtc for gfx950, cached on disk as a code object and
// per process by its source text, and loaded as a module once per device.  Small inputs (and any forest hiprtc
// rejects) keep using the interpreter.
#include <hip/hiprtc.h>

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <unordered_map>

#include "internal.hpp"

namespace dfgpu {

struct JitEntry {
  hipModule_t module = nullptr;
  std::unordered_map<std::string, hipFunction_t> fns;
};
static std::mutex g_jit_mu;
// code objects are per process (compiled or read from disk once per distinct source); MODULES are per device: a hipModule_t loaded on
// device 0 cannot be launched on device 1, and one process may drive several GPUs (dfgpu_init(ids, n > 1))
static std::unordered_map<std::string, std::shared_ptr<std::vector<char>>> g_jit_code;
static std::map<std::pair<int, std::string>, JitEntry> g_jit_modules;
static double g_jit_compile_ms = 0.0;
static int64_t g_jit_compiles = 0, g_jit_disk_hits = 0, g_jit_disk_writes = 0, g_jit_modules_loaded = 0;

#define DFGPU_RTC(expr)                                                                                  \
  do {                                                                                                   \
    hiprtcResult _r = (expr);                                                                            \
    if (_r != HIPRTC_SUCCESS) throw ::dfgpu::Error(std::string("hiprtc error ") + hiprtcGetErrorString(_r) + " at " #expr); \
  } while (0)

// ---- on-disk code-object cache: the 150-160 ms hiprtc takes per distinct forest is paid once per MACHINE, not once per process
// (plans repeat across the processes of a job: one per GPU, restarts, benchmark runs).  Directory: $DFGPU_JIT_CACHE_DIR, else
// $XDG_CACHE_HOME/dfgpu/jit, else ~/.cache/dfgpu/jit; DFGPU_JIT_CACHE=0 turns it off.  A file is named by a 128-bit hash of
// (target, hiprtc version, source) and carries the source length and a third hash, so a collision or a torn file is a miss, never
// a wrong kernel; files are written to a temporary name and renamed.
static uint64_t fnv1a(const std::string& s, uint64_t h) {
  for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ull; }
  return h;
}
static std::string jit_cache_dir() {
  const std::string off_s = option_str("jit.cache", "");
  const char* off = off_s.empty() ? nullptr : off_s.c_str();
  if (off && std::string(off) == "0") return "";
  if (const std::string d = option_str("jit.cache_dir", ""); !d.empty()) return d;
  if (const char* x = std::getenv("XDG_CACHE_HOME")) return std::string(x) + "/dfgpu/jit";
  if (const char* h = std::getenv("HOME")) return std::string(h) + "/.cache/dfgpu/jit";
  return "";
}
static void mkdirs(const std::string& dir) {
  for (size_t i = 1; i <= dir.size(); i++)
    if (i == dir.size() || dir[i] == '/') (void)::mkdir(dir.substr(0, i).c_str(), 0755);
}
struct DiskHeader {
  uint64_t magic, source_bytes, check, code_bytes;
};
static const uint64_t DISK_MAGIC = 0x4a49544746583935ull;  // "JITGFX95"
static std::string jit_keyed(const std::string& source) {
  int maj = 0, min = 0;
  (void)hiprtcVersion(&maj, &min);
  return "gfx950|hiprtc " + std::to_string(maj) + "." + std::to_string(min) + "|" + source;
}
static std::string disk_path(const std::string& dir, const std::string& keyed) {
  char name[64];
  snprintf(name, sizeof name, "%016llx%016llx.hsaco", (unsigned long long)fnv1a(keyed, 0xcbf29ce484222325ull), (unsigned long long)fnv1a(keyed, 0x9e3779b97f4a7c15ull));
  return dir + "/" + name;
}
static std::shared_ptr<std::vector<char>> disk_load(const std::string& dir, const std::string& keyed) {
  if (dir.empty()) return nullptr;
  FILE* f = fopen(disk_path(dir, keyed).c_str(), "rb");
  if (!f) return nullptr;
  DiskHeader h{};
  std::shared_ptr<std::vector<char>> code;
  if (fread(&h, sizeof h, 1, f) == 1 && h.magic == DISK_MAGIC && h.source_bytes == keyed.size() && h.check == fnv1a(keyed, 0x2545f4914f6cdd1dull) &&
      h.code_bytes > 0 && h.code_bytes < (64u << 20)) {
    code = std::make_shared<std::vector<char>>((size_t)h.code_bytes);
    if (fread(code->data(), 1, code->size(), f) != code->size()) code = nullptr;
  }
  fclose(f);
  return code;
}
static void disk_store(const std::string& dir, const std::string& keyed, const std::vector<char>& code) {
  if (dir.empty()) return;
  mkdirs(dir);
  const std::string path = disk_path(dir, keyed), tmp = path + ".tmp" + std::to_string((long)::getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return;  // a read-only home: the cache is an optimisation
  const DiskHeader h{DISK_MAGIC, keyed.size(), fnv1a(keyed, 0x2545f4914f6cdd1dull), code.size()};
  const bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(code.data(), 1, code.size(), f) == code.size();
  if (fclose(f) == 0 && ok && ::rename(tmp.c_str(), path.c_str()) == 0) g_jit_disk_writes++;
  else (void)::remove(tmp.c_str());
}

static std::shared_ptr<std::vector<char>> jit_compile(const std::string& source, const char* kernel_name) {
  auto t0 = std::chrono::steady_clock::now();
  const std::string dump_s = option_str("jit.dump_dir", "");
  if (const char* dump = dump_s.empty() ? nullptr : dump_s.c_str()) {  // debugging aid: the generated sources, one file per node
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
  auto code = std::make_shared<std::vector<char>>(sz);
  DFGPU_RTC(hiprtcGetCode(prog, code->data()));
  DFGPU_RTC(hiprtcDestroyProgram(&prog));
  g_jit_compile_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  g_jit_compiles++;
  return code;
}

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
  const int device = rt().device;  // the calling thread's current device (hipSetDevice is kept in step by rt())
  auto mod = g_jit_modules.find({device, source});
  if (mod != g_jit_modules.end()) return function_of(mod->second);
  std::shared_ptr<std::vector<char>> code;
  auto have = g_jit_code.find(source);
  if (have != g_jit_code.end()) {
    code = have->second;
  } else {
    const std::string dir = jit_cache_dir(), keyed = jit_keyed(source);
    code = disk_load(dir, keyed);
    if (code) {
      g_jit_disk_hits++;
    } else {
      code = jit_compile(source, kernel_name);
      disk_store(dir, keyed, *code);
    }
    g_jit_code.emplace(source, code);
  }
  JitEntry e;
  DFGPU_HIP(hipModuleLoadData(&e.module, code->data()));
  g_jit_modules_loaded++;
  auto ins = g_jit_modules.emplace(std::make_pair(device, source), e);
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
void jit_cache_stats(int64_t* disk_hits, int64_t* disk_writes, int64_t* modules_loaded) {
  std::lock_guard<std::mutex> g(g_jit_mu);
  if (disk_hits) *disk_hits = g_jit_disk_hits;
  if (disk_writes) *disk_writes = g_jit_disk_writes;
  if (modules_loaded) *modules_loaded = g_jit_modules_loaded;
}

}  // namespace dfgpu

extern "C" int dfgpu_jit_stats(int64_t* compiles, double* compile_ms) {
  return dfgpu::guarded([&] { dfgpu::jit_stats(compiles, compile_ms); });
}
extern "C" int dfgpu_jit_cache_stats(int64_t* disk_hits, int64_t* disk_writes, int64_t* modules_loaded) {
  return dfgpu::guarded([&] { dfgpu::jit_cache_stats(disk_hits, disk_writes, modules_loaded); });
}
