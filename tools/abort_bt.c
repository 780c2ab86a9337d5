/* LD_PRELOAD shim for the flake hunt: a native backtrace on SIGABRT / SIGSEGV (glibc's heap checks abort; which library was freeing / allocating?).
   gcc -shared -fPIC -o /tmp/abort_bt.so tools/abort_bt.c */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_fatal(int sig) {
  void* frames[64];
  const char msg[] = "\n---- native backtrace (tools/abort_bt.c) ----\n";
  (void)!write(2, msg, sizeof msg - 1);
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  void* warm[4];
  backtrace(warm, 4); /* loads libgcc now: not from inside the handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_fatal;
  sa.sa_flags = SA_NODEFER;
  sigaction(SIGABRT, &sa, 0);
  sigaction(SIGSEGV, &sa, 0);
}
