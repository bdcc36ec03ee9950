/* Test-harness aid: a SIGSEGV / SIGBUS / SIGABRT handler that writes the NATIVE backtrace of the faulting thread to
 * stderr - with the thread id, so a fault on a runtime helper thread is told from one on the main thread - and then hands over to
 * the handler that was installed before it (pytest's faulthandler: the Python frames).  Built by tests/conftest.py with gcc into the
 * pytest temp directory on GPU runs; not part of the product. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static struct sigaction g_prev[65];

static void on_fault(int sig, siginfo_t* info, void* ctx) {
  char buf[160];
  const long tid = syscall(SYS_gettid);
  const int n = snprintf(buf, sizeof(buf), "\n[crash_backtrace] signal %d at address %p on thread %ld (process %d): native frames follow\n", sig,
                         info ? info->si_addr : (void*)0, tid, (int)getpid());
  if (n > 0) { ssize_t w = write(2, buf, (size_t)n); (void)w; }
  void* frames[64];
  const int nf = backtrace(frames, 64);
  backtrace_symbols_fd(frames, nf, 2);
  const struct sigaction* p = &g_prev[sig];
  if (p->sa_flags & SA_SIGINFO) { if (p->sa_sigaction) { p->sa_sigaction(sig, info, ctx); return; } }
  else if (p->sa_handler != SIG_DFL && p->sa_handler != SIG_IGN) { p->sa_handler(sig); return; }
  signal(sig, SIG_DFL);
  raise(sig);
}

void crash_backtrace_install(void) {
  static char altstack[1 << 16];
  stack_t ss; ss.ss_sp = altstack; ss.ss_size = sizeof(altstack); ss.ss_flags = 0;
  sigaltstack(&ss, 0);
  void* warm[2]; backtrace(warm, 2);   /* loads libgcc now, not inside the handler */
  const int sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};
  for (unsigned i = 0; i < sizeof(sigs) / sizeof(sigs[0]); ++i) {
    struct sigaction sa; memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_fault; sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    sigaction(sigs[i], &sa, &g_prev[sigs[i]]);
  }
}
