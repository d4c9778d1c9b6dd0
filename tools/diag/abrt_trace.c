/* diagnostics: LD_PRELOAD this to get the native call stack of whatever calls abort() (glibc backtrace on SIGABRT, then the default action) */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abrt(int sig) {
  void* frames[96];
  int n = backtrace(frames, 96);
  const char msg[] = "\n==== SIGABRT: native backtrace ====\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = on_abrt;
  sigaction(SIGABRT, &sa, 0);
}
