/* LD_PRELOAD helper for runs on the GPU box: prints a native backtrace (glibc backtrace_symbols_fd) to stderr when the process
 * receives SIGABRT / SIGSEGV / SIGBUS — e.g. glibc's "double free or corruption" abort — and then lets the default action happen.
 *   gcc -O1 -g -shared -fPIC -o tools/abort_bt.so tools/abort_bt.c
 *   LD_PRELOAD=$PWD/tools/abort_bt.so python -X faulthandler tools/scale_replay.py ...
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig) {
    void* frames[96];
    const char* msg = sig == SIGABRT ? "\n[abort_bt] SIGABRT, native backtrace:\n" : "\n[abort_bt] fatal signal, native backtrace:\n";
    (void)!write(2, msg, strlen(msg));
    const int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = handler;
    sa.sa_flags = SA_NODEFER;
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
}
