// LD_PRELOAD shim: native backtrace on SIGSEGV / SIGABRT (no gdb on the GPU boxes).  gcc -shared -fPIC -o segv_bt.so segv_bt.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
static void on_sig(int sig) {
  void* bt[64];
  int n = backtrace(bt, 64);
  fprintf(stderr, "\n== native backtrace (signal %d) ==\n", sig);
  backtrace_symbols_fd(bt, n, 2);
  _exit(128 + sig);
}
__attribute__((constructor)) static void init(void) { signal(SIGSEGV, on_sig); signal(SIGABRT, on_sig); signal(SIGBUS, on_sig); }
