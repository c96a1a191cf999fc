/* LD_PRELOAD helper for runs on the GPU box: when the process receives SIGABRT / SIGSEGV / SIGBUS — e.g. glibc's "double free or
 * corruption" abort — it prints where it happened and then lets the default action happen.
 *
 * glibc's backtrace() cannot be used here: the abort this was written for (round 3 / round 4: a heap abort AFTER main() has returned) fires
 * inside the exit-time destructors, i.e. with the dynamic loader's lock held, and the unwinder's dl_iterate_phdr() then waits for that lock
 * for ever (two captures ended in "native backtrace:" and a 7-minute hang).  So the handler uses nothing but open / read / write: it reads
 * /proc/self/maps, then walks the stack upwards from the signal frame and prints every word that points into an executable mapping as
 * "module+offset" — a superset of the return addresses, innermost first; resolve with `addr2line -f -C -e <module> <offset>` / `nm -D`.
 * An alarm ends the process if anything in here blocks after all.
 *   gcc -O1 -g -shared -fPIC -o tools/abort_bt.so tools/abort_bt.c
 *   LD_PRELOAD=$PWD/tools/abort_bt.so python3 tools/scale_replay.py ...
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <string.h>
#include <ucontext.h>
#include <unistd.h>

#define MAX_MAPS 1024
static struct { uintptr_t lo, hi, off; char name[96]; } g_maps[MAX_MAPS];
static int g_nmaps;
static uintptr_t g_probe, g_probe_hi;   /* an address (the stack pointer) and the end of the mapping that holds it */
static char g_buf[1 << 18];

static void put(const char* s) { (void)!write(2, s, strlen(s)); }
static void put_hex(uintptr_t v) {
    char b[19] = "0x";
    for (int i = 0; i < 16; ++i) b[2 + i] = "0123456789abcdef"[(v >> (60 - 4 * i)) & 15];
    b[18] = 0;
    put(b);
}
static uintptr_t hex(const char** p) {
    uintptr_t v = 0;
    for (;; ++*p) {
        const char c = **p;
        if (c >= '0' && c <= '9') v = v * 16 + (uintptr_t)(c - '0');
        else if (c >= 'a' && c <= 'f') v = v * 16 + (uintptr_t)(c - 'a' + 10);
        else return v;
    }
}
static void read_maps(void) {
    const int fd = open("/proc/self/maps", O_RDONLY);
    if (fd < 0) return;
    size_t n = 0;
    for (;;) {
        const ssize_t r = read(fd, g_buf + n, sizeof(g_buf) - 1 - n);
        if (r <= 0) break;
        n += (size_t)r;
        if (n >= sizeof(g_buf) - 1) break;
    }
    close(fd);
    g_buf[n] = 0;
    g_nmaps = 0;
    for (const char* p = g_buf; *p && g_nmaps < MAX_MAPS;) {
        const char* line = p;
        const uintptr_t lo = hex(&p);
        if (*p != '-') break;
        ++p;
        const uintptr_t hi = hex(&p);
        const char* perms = p + 1;       /* " r-xp " */
        const int exec = perms[2] == 'x';
        p = perms + 5;
        const uintptr_t off = hex(&p);
        const char* eol = strchr(line, '\n');
        if (!eol) eol = line + strlen(line);
        if (g_probe >= lo && g_probe < hi) g_probe_hi = hi;
        if (exec) {
            const char* name = eol;
            while (name > line && name[-1] != ' ' && name[-1] != '/') --name;
            size_t len = (size_t)(eol - name);
            if (len > sizeof(g_maps[0].name) - 1) len = sizeof(g_maps[0].name) - 1;
            g_maps[g_nmaps].lo = lo; g_maps[g_nmaps].hi = hi; g_maps[g_nmaps].off = off;
            memcpy(g_maps[g_nmaps].name, name, len);
            g_maps[g_nmaps].name[len] = 0;
            ++g_nmaps;
        }
        p = *eol ? eol + 1 : eol;
    }
}
static int lookup(uintptr_t a) {
    for (int i = 0; i < g_nmaps; ++i)
        if (a >= g_maps[i].lo && a < g_maps[i].hi) return i;
    return -1;
}
static void show(const char* tag, uintptr_t a) {
    const int i = lookup(a);
    if (i < 0) return;
    put(tag); put(g_maps[i].name[0] ? g_maps[i].name : "[anon]"); put("+"); put_hex(a - g_maps[i].lo + g_maps[i].off); put("\n");
}

static void handler(int sig, siginfo_t* si, void* uc_) {
    (void)si;
    alarm(20);  /* SIGALRM keeps its default action: whatever blocks below, the process ends */
    put(sig == SIGABRT ? "\n[abort_bt] SIGABRT — code addresses on the stack, innermost first (module+file offset):\n" : "\n[abort_bt] fatal signal — code addresses on the stack:\n");
    const ucontext_t* uc = (const ucontext_t*)uc_;
    uintptr_t sp = (uintptr_t)&uc, pc = 0;
#if defined(__x86_64__)
    if (uc) {
        pc = (uintptr_t)uc->uc_mcontext.gregs[REG_RIP];
        sp = (uintptr_t)uc->uc_mcontext.gregs[REG_RSP];
    }
#endif
    g_probe = sp;
    g_probe_hi = 0;
    read_maps();
    if (pc) show("  pc  ", pc);
    uintptr_t end = sp + (1u << 16);
    if (g_probe_hi && end > g_probe_hi) end = g_probe_hi;   /* never past the end of the stack's own mapping */
    if (!g_probe_hi) end = sp;                              /* mapping not found: print nothing rather than fault */
    int shown = 0;
    for (uintptr_t a = sp & ~(uintptr_t)7; shown < 160 && a + 8 <= end; a += 8) {
        const uintptr_t w = *(const uintptr_t*)a;
        if (lookup(w) >= 0) { show("  ", w); ++shown; }
    }
    put("[abort_bt] end of list\n");
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_NODEFER | SA_SIGINFO;
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
}
