// Diagnostic preload: print the C stack when std::terminate runs (tools/gpu_r03z.sh).  Not part of the product.
#include <execinfo.h>
#include <exception>
#include <cstdlib>
#include <unistd.h>
#include <signal.h>
static void on_terminate() {
  void* fr[64]; int n = backtrace(fr, 64);
  const char m[] = "---- std::terminate: C stack ----\n"; (void)!write(2, m, sizeof(m) - 1);
  backtrace_symbols_fd(fr, n, 2);
  signal(SIGABRT, SIG_DFL); abort();
}
__attribute__((constructor)) static void install() { std::set_terminate(on_terminate); }
