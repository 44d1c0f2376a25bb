// Native backtrace on SIGSEGV / SIGABRT / SIGBUS (no gdb on the GPU boxes).  gcc -shared -fPIC -o segv_bt.so segv_bt.c
// Load with LD_PRELOAD, or from Python through ctypes.CDLL (tests/conftest.py: POEM_NATIVE_BT=<path>).  The handler runs on an
// alternate stack, so a stack overflow (runaway recursion) is reported too.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void on_sig(int sig, siginfo_t* si, void* ctx) {
  void* bt[96];
  (void)ctx;
  int n = backtrace(bt, 96);
  char msg[128];
  int m = snprintf(msg, sizeof msg, "\n== native backtrace (signal %d, fault address %p) ==\n", sig, si ? si->si_addr : (void*)0);
  if (write(2, msg, (size_t)m) < 0) {}
  backtrace_symbols_fd(bt, n, 2);
  _exit(128 + sig);
}
__attribute__((constructor)) static void init(void) {
  static char alt[1 << 16];
  stack_t ss;
  ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0;
  sigaltstack(&ss, NULL);
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_sig;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, NULL);
  sigaction(SIGABRT, &sa, NULL);
  sigaction(SIGBUS, &sa, NULL);
}
