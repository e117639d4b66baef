// Runtime specialisation of the row front-end with hiprtc.
//
// For large inputs the interpreter's per-row dispatch cost matters: the same typed DAG the
// bytecode comes from is emitted as a straight-line C++ function (expr_compile.cpp::jit_source) and
// compiled together with the UNCHANGED sink kernel source (embedded in this library at build time) into
// one gfx950 code object.  hiprtc is loaded with dlopen, so the library still loads where it is absent;
// the interpreter kernels are always there.
#include "jit_runtime.h"
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <thread>
#include <stdexcept>
#include <vector>

extern "C" {
extern const char* const gpuq_embedded_names[];
extern const char* const gpuq_embedded_sources[];
extern const int gpuq_embedded_count;
}

namespace gpuq {
namespace {
typedef struct _hiprtcProgram* hiprtcProgram;
struct Rtc {
  void* h = nullptr;
  int (*Create)(hiprtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*Compile)(hiprtcProgram, int, const char* const*) = nullptr;
  int (*LogSize)(hiprtcProgram, size_t*) = nullptr;
  int (*Log)(hiprtcProgram, char*) = nullptr;
  int (*CodeSize)(hiprtcProgram, size_t*) = nullptr;
  int (*Code)(hiprtcProgram, char*) = nullptr;
  int (*Destroy)(hiprtcProgram*) = nullptr;
  bool ok = false;
};
Rtc& rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, []() {
    for (const char* n : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h) return;
    r.Create = (decltype(r.Create))dlsym(r.h, "hiprtcCreateProgram");
    r.Compile = (decltype(r.Compile))dlsym(r.h, "hiprtcCompileProgram");
    r.LogSize = (decltype(r.LogSize))dlsym(r.h, "hiprtcGetProgramLogSize");
    r.Log = (decltype(r.Log))dlsym(r.h, "hiprtcGetProgramLog");
    r.CodeSize = (decltype(r.CodeSize))dlsym(r.h, "hiprtcGetCodeSize");
    r.Code = (decltype(r.Code))dlsym(r.h, "hiprtcGetCode");
    r.Destroy = (decltype(r.Destroy))dlsym(r.h, "hiprtcDestroyProgram");
    r.ok = r.Create && r.Compile && r.LogSize && r.Log && r.CodeSize && r.Code && r.Destroy;
  });
  return r;
}
const char* file_of(int kernel_id) {
  if (kernel_id >= 1 && kernel_id <= 3) return "kernels_scan.hip";
  if ((kernel_id >= 4 && kernel_id <= 7) || (kernel_id >= 11 && kernel_id <= 15)) return "kernels_hash.hip";
  if (kernel_id >= 8 && kernel_id <= 10) return "kernels_sort.hip";
  throw std::runtime_error("jit: bad kernel id");
}
std::mutex g_mu;
std::map<std::string, JitFn> g_cache;
}  // namespace

bool jit_available() { return rtc().ok; }

// GPUQ_JIT_DEFINES="A=1;B=2" (environment, tuning experiments): extra #defines in front of every JIT translation unit
static std::string env_defines() {
  const char* e = std::getenv("GPUQ_JIT_DEFINES");
  std::string r, tok;
  if (!e) return r;
  for (const char* p = e;; ++p) {
    if (*p == ';' || *p == 0) {
      if (!tok.empty()) { const size_t eq = tok.find('='); r += "#define " + (eq == std::string::npos ? tok + " 1" : tok.substr(0, eq) + " " + tok.substr(eq + 1)) + "\n"; }
      tok.clear(); if (!*p) break;
    } else tok += *p;
  }
  return r;
}

// the entry point carries the sink's name so that profiles tell the specialised kernels apart
static const char* jit_entry_name(int kernel_id) {
  static const char* n[] = {"gpuq_jit_entry", "gpuq_jit_filter_bitmap", "gpuq_jit_project", "gpuq_jit_agg_tiny", "gpuq_jit_agg_hash", "gpuq_jit_join_build",
                            "gpuq_jit_join_probe", "gpuq_jit_join_probe_unique", "gpuq_jit_sort_minmax", "gpuq_jit_sort_pack", "gpuq_jit_part_pid",
                            "gpuq_jit_agg_bucket_id", "gpuq_jit_agg_bucket", "gpuq_jit_agg_lds", "gpuq_jit_join_keyrange", "gpuq_jit_rj_pack"};
  return (kernel_id >= 1 && kernel_id <= 15) ? n[kernel_id] : n[0];
}

// "//@label xyz" anywhere in the front-end source (an operator's descriptor "label": a plan node id) is appended to the entry point's name:
// two call sites of one sink kernel -- the two probes of q3 -- then show up as two kernels in a profile
static std::string entry_name(const std::string& eval_src, int kernel_id) {
  std::string n = jit_entry_name(kernel_id);
  const size_t p = eval_src.find("//@label ");
  if (p != std::string::npos) {
    std::string l;
    for (size_t i = p + 9; i < eval_src.size() && (std::isalnum((unsigned char)eval_src[i]) || eval_src[i] == '_') && l.size() < 32; ++i) l += eval_src[i];
    if (!l.empty()) n += "_" + l;
  }
  return n;
}
std::string jit_full_source(const std::string& eval_src, int kernel_id) {
  return env_defines() + "#define gpuq_jit_entry " + entry_name(eval_src, kernel_id) + "\n#define GPUQ_JIT 1\n#define GPUQ_JIT_KERNEL " + std::to_string(kernel_id) +
         "\n#include \"gpuq_kernels.h\"\nnamespace gpuq {\n" + eval_src + "}\n#include \"" + file_of(kernel_id) + "\"\n";
}

// ---- on-disk code-object cache.  A one-shot task must not pay 0.3-1.9 s of hiprtc for a pipeline some earlier process on this
// host already compiled: code objects are kept under $GPUQ_JIT_CACHE_DIR (default $XDG_CACHE_HOME/gpuq-jit or ~/.cache/gpuq-jit;
// "off" disables), one file per (full translation unit, compiler options, hiprtc version), named by a 128-bit hash of that text.
// Files are written to a temporary name and renamed, so concurrent executors on one node share the directory safely.
static std::string cache_dir() {
  static const std::string dir = []() -> std::string {
    const char* e = std::getenv("GPUQ_JIT_CACHE_DIR");
    std::string d;
    if (e && *e) { if (std::string(e) == "off") return std::string(); d = e; }
    else if (const char* x = std::getenv("XDG_CACHE_HOME")) d = std::string(x) + "/gpuq-jit";
    else if (const char* h = std::getenv("HOME")) d = std::string(h) + "/.cache/gpuq-jit";
    else d = "/tmp/gpuq-jit-" + std::to_string((long)getuid());
    for (size_t i = 1; i <= d.size(); ++i)
      if (i == d.size() || d[i] == '/') { const std::string p = d.substr(0, i); if (::mkdir(p.c_str(), i == d.size() ? 0700 : 0777) != 0 && errno != EEXIST) return std::string(); }
    // Code objects loaded from this directory run on the device unauthenticated: it is only trusted when it is a directory that
    // belongs to this user and that nobody else can write to (a pre-created /tmp/gpuq-jit-<uid> of another owner is refused).
    struct stat st;
    if (::lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != ::getuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return std::string();
    return d;
  }();
  return dir;
}
static std::string cache_name(const std::string& text) {
  unsigned long long a = 0xcbf29ce484222325ull, b = 0x84222325cbf29ce4ull;
  for (unsigned char c : text) { a = (a ^ c) * 0x100000001b3ull; b = (b ^ (c + 0x9e)) * 0x100000001b3ull; b ^= b >> 29; }
  char buf[48]; std::snprintf(buf, sizeof(buf), "%016llx%016llx.co", a, b);
  return buf;
}
static bool cache_read(const std::string& path, std::vector<char>& code) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  bool ok = n > 0;
  if (ok) { code.resize((size_t)n); ok = std::fread(code.data(), 1, (size_t)n, f) == (size_t)n; }
  std::fclose(f);
  return ok;
}
static void cache_write(const std::string& path, const std::vector<char>& code) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return;
  const bool ok = std::fwrite(code.data(), 1, code.size(), f) == code.size();
  std::fclose(f);
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());
}
static int g_disk_hits = 0, g_compiles = 0;
void jit_cache_stats(int* disk_hits, int* compiles) { std::lock_guard<std::mutex> lk(g_mu); if (disk_hits) *disk_hits = g_disk_hits; if (compiles) *compiles = g_compiles; }

// hiprtc compile (or disk-cache hit) + module load of one (front-end source, sink) pair on the CURRENT device; no lock held
static JitFn compile_and_load(const std::string& eval_src, int kernel_id) {
  Rtc& r = rtc();
  if (!r.ok) throw std::runtime_error("jit: hiprtc is not available on this host");
  const std::string src = jit_full_source(eval_src, kernel_id);
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  std::vector<char> code;
  std::string cpath;
  if (!cache_dir().empty()) {
    int ver = 0; (void)hipRuntimeGetVersion(&ver);
    std::string ident = src + "|" + std::to_string(ver);
    for (const char* o : opts) { ident += "|"; ident += o; }
    for (int i = 0; i < gpuq_embedded_count; ++i) { ident += "|"; ident += gpuq_embedded_sources[i]; }      // the #included sink sources are part of the unit
    cpath = cache_dir() + "/" + cache_name(ident);
  }
  bool from_disk = !cpath.empty() && cache_read(cpath, code);
  if (!from_disk) {
    hiprtcProgram prog = nullptr;
    if (r.Create(&prog, src.c_str(), "gpuq_jit.hip", gpuq_embedded_count, gpuq_embedded_sources, gpuq_embedded_names) != 0)
      throw std::runtime_error("jit: hiprtcCreateProgram failed");
    const int rc = r.Compile(prog, 3, opts);
    if (rc != 0) {
      size_t ls = 0; r.LogSize(prog, &ls);
      std::string log(ls + 1, 0); if (ls) r.Log(prog, &log[0]);
      r.Destroy(&prog);
      throw std::runtime_error("jit: compile failed:\n" + log.substr(0, 4000));
    }
    size_t cs = 0; r.CodeSize(prog, &cs);
    code.resize(cs); r.Code(prog, code.data());
    r.Destroy(&prog);
  }
  JitFn f{};
  hipModule_t mod = nullptr; hipFunction_t fn = nullptr;
  hipError_t e = hipModuleLoadData(&mod, code.data());
  if (e != hipSuccess && from_disk) {      // a damaged cache file: compile as if it were not there
    (void)hipGetLastError(); (void)std::remove(cpath.c_str());
    const std::string saved = cpath;
    return compile_and_load(eval_src, kernel_id);
  }
  if (e != hipSuccess) throw std::runtime_error(std::string("jit: hipModuleLoadData: ") + hipGetErrorString(e));
  e = hipModuleGetFunction(&fn, mod, entry_name(eval_src, kernel_id).c_str());
  if (e != hipSuccess) { (void)hipModuleUnload(mod); throw std::runtime_error(std::string("jit: hipModuleGetFunction: ") + hipGetErrorString(e)); }
  if (!from_disk && !cpath.empty()) cache_write(cpath, code);
  { std::lock_guard<std::mutex> lk(g_mu); if (from_disk) ++g_disk_hits; else ++g_compiles; }
  f.module = mod; f.fn = fn;
  return f;
}

// cache key: a loaded module belongs to the device it was loaded on
static std::string jit_key(const std::string& eval_src, int kernel_id) {
  int dev = 0; (void)hipGetDevice(&dev);
  return std::to_string(dev) + "|" + std::to_string(kernel_id) + "|" + eval_src;
}

const JitFn* jit_get(const std::string& eval_src, int kernel_id) {
  const std::string key = jit_key(eval_src, kernel_id);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) return &it->second;
  }
  const JitFn f = compile_and_load(eval_src, kernel_id);      // two threads may compile the same key; the first insert wins
  std::lock_guard<std::mutex> lk(g_mu);
  auto ins = g_cache.emplace(key, f);
  if (!ins.second) (void)hipModuleUnload((hipModule_t)f.module);      // lost the race: drop the duplicate module
  return &ins.first->second;
}

// ---- background tier: operators that run again and again on small inputs (a stage's many small partitions, the final stage
// of a two-phase aggregate) are worth specialising too -- a lone wave interprets ~100 instructions in ~20 us, the generated
// code runs them in ~3 -- but nobody should wait for the compiler: the request is queued, a worker thread compiles, callers
// keep using the interpreter kernels until the function is there.
namespace {
struct BgJob { std::string key, src; int kernel_id; int device; };
struct Bg {
  std::mutex mu; std::condition_variable cv, idle;
  std::deque<BgJob> queue; std::set<std::string> known;      // queued, running, done or failed: never requested twice
  bool running = false, started = false, stop = false;
};
Bg& bg() { static Bg* b = new Bg(); return *b; }      // leaked on purpose: the worker may outlive static destruction
void bg_worker() {
  Bg& B = bg();
  for (;;) {
    BgJob job;
    {
      std::unique_lock<std::mutex> lk(B.mu);
      B.cv.wait(lk, [&] { return B.stop || !B.queue.empty(); });
      if (B.stop) { B.running = false; B.idle.notify_all(); return; }
      job = std::move(B.queue.front()); B.queue.pop_front(); B.running = true;
    }
    try {
      (void)hipSetDevice(job.device);
      const JitFn f = compile_and_load(job.src, job.kernel_id);
      std::lock_guard<std::mutex> lk(g_mu);
      if (!g_cache.emplace(job.key, f).second) (void)hipModuleUnload((hipModule_t)f.module);
    } catch (const std::exception&) { /* stays on the interpreter kernels */ }
    {
      std::lock_guard<std::mutex> lk(B.mu);
      B.running = false;
      if (B.queue.empty()) B.idle.notify_all();
    }
  }
}
void bg_stop_at_exit() {
  Bg& B = bg();
  std::unique_lock<std::mutex> lk(B.mu);
  B.queue.clear(); B.stop = true; B.cv.notify_all();
  B.idle.wait_for(lk, std::chrono::seconds(20), [&] { return !B.running; });      // do not tear the runtime down under a compile
}
}  // namespace

const JitFn* jit_try_get(const std::string& eval_src, int kernel_id) {
  const std::string key = jit_key(eval_src, kernel_id);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) return &it->second;
  }
  if (!rtc().ok) return nullptr;
  Bg& B = bg();
  std::lock_guard<std::mutex> lk(B.mu);
  if (B.stop || !B.known.insert(key).second) return nullptr;
  int dev = 0; (void)hipGetDevice(&dev);
  B.queue.push_back({key, eval_src, kernel_id, dev});
  if (!B.started) { B.started = true; std::thread(bg_worker).detach(); std::atexit(bg_stop_at_exit); }
  B.cv.notify_one();
  return nullptr;
}

void jit_drain() {
  Bg& B = bg();
  std::unique_lock<std::mutex> lk(B.mu);
  if (!B.started) return;
  B.idle.wait(lk, [&] { return B.queue.empty() && !B.running; });
}

}  // namespace gpuq
