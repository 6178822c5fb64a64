/* Debug aid: print the C call stack when the process receives SIGABRT (a silent abort() inside a native library).
 * gcc -shared -fPIC -o libabort_trace.so abort_trace.c ; load with ctypes and call install_abort_trace(). */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig) {
    void* frames[96];
    const char msg[] = "\n=== SIGABRT: native call stack ===\n";
    int n = backtrace(frames, 96);
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

void install_abort_trace(void) { signal(SIGABRT, on_abort); }
