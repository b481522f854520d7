/*
 * abort_trace.c -- LD_PRELOAD helper of tests/test_gpu_exit.py: prints the
 * native call stack when the process aborts (glibc's "double free or
 * corruption" ends in abort()), so that a teardown bug names its frame
 * instead of just "dumped core".  Test tooling, not part of the product.
 *   gcc -shared -fPIC -O1 tools/abort_trace.c -o spmv_scpa_amd/bin/libabort_trace.so
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig) {
    static const char head[] = "\n== abort_trace: native stack at SIGABRT ==\n";
    void *frames[64];
    (void)!write(2, head, sizeof head - 1);
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    void *warm[2];
    backtrace(warm, 2); /* loads libgcc now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_abort;
    sigaction(SIGABRT, &sa, NULL);
}
