// Diagnostics that are not kernels: a native back trace when the process dies of a signal (the debug aids of
// the build).  The step replays multi-stream hipGraphs; when a fresh process dies inside the HIP runtime (DESIGN section 3,
// "lessons") Python's faulthandler only shows `graphs.py: replay` -- this handler prints the frames below it, per thread
// that takes the signal, with async-signal-safe calls only (backtrace_symbols_fd writes straight to the descriptor).
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>

#include "s2ag_common.h"

namespace {
int g_fd = 2;
void put(const char* s) { (void)!write(g_fd, s, strlen(s)); }
void put_int(long v) {
    char buf[24];
    int n = 0;
    if (v < 0) { put("-"); v = -v; }
    do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v && n < 23);
    while (n--) (void)!write(g_fd, &buf[n], 1);
}
void on_fatal(int sig, siginfo_t* info, void*) {
    put("\n[s2ag crash handler] signal ");
    put_int(sig);
    put(" in thread ");
    put_int((long)syscall(SYS_gettid));
    put(" (pid ");
    put_int((long)getpid());
    put("), fault address 0x");
    {
        unsigned long a = (unsigned long)info->si_addr;
        char hex[17];
        for (int i = 15; i >= 0; --i) { hex[i] = "0123456789abcdef"[a & 15]; a >>= 4; }
        hex[16] = 0;
        put(hex);
    }
    put("\n");
    void* frames[96];
    const int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, g_fd);
    put("[s2ag crash handler] end of native back trace\n");
    // back to the default action (or whoever was there before is gone: faulthandler chains itself if enabled later)
    struct sigaction dfl;
    memset(&dfl, 0, sizeof dfl);
    dfl.sa_handler = SIG_DFL;
    sigaction(sig, &dfl, nullptr);
    raise(sig);
}
}  // namespace

extern "C" int s2ag_install_crash_handler(int fd) {
    if (fd < 0) return S2AG_E_BADARG;
    g_fd = fd;
    void* warm[4];
    backtrace(warm, 4);          // loads libgcc now: dlopen inside a signal handler is not safe
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fatal;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
    static char alt[1 << 16];
    stack_t ss;
    ss.ss_sp = alt;
    ss.ss_size = sizeof alt;
    ss.ss_flags = 0;
    sigaltstack(&ss, nullptr);
    for (int s : {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL})
        if (sigaction(s, &sa, nullptr) != 0) return S2AG_E_BADARG;
    return 0;
}
