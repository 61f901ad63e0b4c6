/* TEST INFRASTRUCTURE (loaded by tests/conftest.py only; the product never links it).
 *
 * When the test process dies of a signal (GPUTEST_r05: SIGSEGV, 0 passed, nothing in the log's tail to say where), the last
 * thing written to the log is: the signal, the NATIVE backtrace of the faulting thread as module(+offset) pairs
 * (resolve with `llvm-symbolizer -e <module> <offset>` / addr2line), and the node id of the running test.
 *
 * Installed BEFORE Python's faulthandler: faulthandler keeps this handler as "previous", dumps the Python stacks, restores
 * this one and raises the signal again - so this report comes after the Python dump and is what a `tail` of the log shows.
 * Afterwards the default action is restored and the signal re-raised: the exit status stays 128 + signal.
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static char g_test[768] = "(no test started)";
static int g_fd = 2;

void crashname_set(const char* node_id) {
    if (!node_id) return;
    strncpy(g_test, node_id, sizeof g_test - 1);
    g_test[sizeof g_test - 1] = 0;
}

static void put(const char* s) {
    size_t n = strlen(s);
    while (n > 0) {
        ssize_t w = write(g_fd, s, n);
        if (w <= 0) return;
        s += w;
        n -= (size_t)w;
    }
}

static void put_int(long v) {
    char b[24];
    int i = 23;
    b[i] = 0;
    if (v == 0) b[--i] = '0';
    int neg = v < 0;
    if (neg) v = -v;
    while (v > 0 && i > 1) { b[--i] = (char)('0' + v % 10); v /= 10; }
    if (neg) b[--i] = '-';
    put(b + i);
}

static void on_signal(int sig, siginfo_t* si, void* uc) {
    (void)uc;
    put("\n[crash] signal ");
    put_int(sig);
    put(" (");
    put(sig == SIGSEGV ? "SIGSEGV" : sig == SIGBUS ? "SIGBUS" : sig == SIGABRT ? "SIGABRT" : sig == SIGFPE ? "SIGFPE" : sig == SIGILL ? "SIGILL" : "?");
    put(si && si->si_code > 0 ? "), fault address 0x" : "), re-raised by the fault handler before this one (no address), 0x");
    {
        unsigned long a = (unsigned long)(si && si->si_code > 0 ? si->si_addr : 0);
        char h[20];
        int i = 19;
        h[i] = 0;
        if (a == 0) h[--i] = '0';
        while (a > 0 && i > 0) { h[--i] = "0123456789abcdef"[a & 15]; a >>= 4; }
        put(h + i);
    }
    put(", thread ");
    put_int((long)gettid());
    put(" of process ");
    put_int((long)getpid());
    put("\n[crash] native backtrace of the faulting thread, module(+offset):\n");
    void* frames[30];                            /* the innermost frames are the ones that matter; deeper it is the interpreter loop */
    int n = backtrace(frames, 30);
    backtrace_symbols_fd(frames, n, g_fd);
    put("[crash] running test: ");
    put(g_test);
    put("\n");
    signal(sig, SIG_DFL);
    raise(sig);
}

/* fd: where the report goes (a dup of the real stderr, taken before pytest redirects fd 2 for capture) */
int crashname_install(int fd) {
    void* warm[4];
    (void)backtrace(warm, 4);                 /* loads libgcc's unwinder now: not something to do inside a signal handler */
    if (fd >= 0) g_fd = fd;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_signal;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
    sigemptyset(&sa.sa_mask);
    int rc = 0;
    const int sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};
    for (unsigned i = 0; i < sizeof sigs / sizeof sigs[0]; ++i) rc |= sigaction(sigs[i], &sa, 0);
    return rc;
}
