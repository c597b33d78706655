// Debug aid: LD_PRELOAD=tools/dev/abrt_trace.so prints the native stack of the thread that raises SIGABRT / SIGSEGV.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_sig(int sig)
{
	void* bt[64]; int n = backtrace(bt, 64);
	const char* m = sig == SIGABRT ? "\n[abrt_trace] SIGABRT, native stack:\n" : "\n[abrt_trace] SIGSEGV, native stack:\n";
	write(2, m, strlen(m));
	backtrace_symbols_fd(bt, n, 2);
	signal(sig, SIG_DFL); raise(sig);
}
__attribute__((constructor)) static void init(void) { signal(SIGABRT, on_sig); signal(SIGSEGV, on_sig); }
