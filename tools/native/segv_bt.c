// Diagnostic helper: print the native backtrace of a SIGSEGV / SIGABRT to stderr, then let the default action run.
//   gcc -shared -fPIC -O1 -o tools/native/libsegv_bt.so tools/native/segv_bt.c ; ctypes.CDLL(...).segv_bt_install()
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t* si, void* ctx) {
  void* frames[64];
  const char msg[] = "\n==== native backtrace (segv_bt) ====\n";
  (void)si; (void)ctx;
  if (write(2, msg, sizeof(msg) - 1) < 0) {}
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

void segv_bt_install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}
