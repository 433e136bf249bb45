/* Loaded into a process that dies of SIGSEGV now and then: prints the native backtrace (addresses with the mapped
 * object's name and offset) to stderr and lets the signal take its course. */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_segv(int sig, siginfo_t* si, void* ctx) {
  (void)ctx;
  void* frames[64];
  const char msg[] = "segv_bt: SIGSEGV, native frames:\n";
  write(2, msg, sizeof msg - 1);
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  (void)si;
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, 0);
}
