/* diagnostics: LD_PRELOAD this to get the native call stack of whatever calls abort() or faults (SIGABRT / SIGSEGV / SIGBUS): glibc
 * backtrace into a FILE OF ITS OWN, opened in the constructor -- not fd 2, which a test runner's fd-level capture may own at that moment
 * (round 4: the trace went into pytest's temp file and died with the process).  OCF_ABRT_TRACE_FILE names the file (default
 * ./abrt_trace.txt); the handler also copies /proc/self/maps so that the module+offset frames can be symbolised afterwards.
 *   gcc -shared -fPIC -O1 -o tools/diag/abrt_trace.so tools/diag/abrt_trace.c                                                        */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int g_fd = -1;

static void put(const char* s) { (void)!write(g_fd, s, strlen(s)); }

static void on_fatal(int sig) {
  void* frames[128];
  int n = backtrace(frames, 128);
  if (g_fd < 0) g_fd = 2;
  put("\n==== fatal signal ");
  put(sig == SIGABRT ? "SIGABRT" : sig == SIGSEGV ? "SIGSEGV" : "SIGBUS");
  put(": native backtrace of the raising thread ====\n");
  backtrace_symbols_fd(frames, n, g_fd);
  put("==== /proc/self/maps (executable mappings) ====\n");
  int m = open("/proc/self/maps", O_RDONLY);
  if (m >= 0) {
    static char buf[1 << 16];
    ssize_t k;
    while ((k = read(m, buf, sizeof(buf))) > 0) (void)!write(g_fd, buf, (size_t)k);
    close(m);
  }
  fsync(g_fd);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  const char* path = getenv("OCF_ABRT_TRACE_FILE");
  g_fd = open(path && *path ? path : "abrt_trace.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
  void* warm[4];
  backtrace(warm, 4); /* loads libgcc now: backtrace() must not call the loader from inside a signal handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = on_fatal;
  sigaction(SIGABRT, &sa, 0);
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}
