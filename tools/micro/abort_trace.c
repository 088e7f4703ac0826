// LD_PRELOAD shim: print the C call stack when the process receives SIGABRT (who called abort()?).
//   gcc -shared -fPIC -o tools/micro/libs/abort_trace.so tools/micro/abort_trace.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_abort(int sig) {
  void* frames[64];
  int n = backtrace(frames, 64);
  static const char msg[] = "\n[abort_trace] SIGABRT, C stack:\n";
  write(2, msg, sizeof msg - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void install(void) { signal(SIGABRT, on_abort); }
