/* LD_PRELOAD heap guard for the flake hunt of DESIGN section 7 (host heap damaged during the eval-latent fit tests, noticed by glibc much
   later, in whoever allocates next).  glibc's MALLOC_CHECK_ does not arm under this image's python and GPU sanitizers are not available on
   the pool; this is the host-only stand-in:

     * every block gets a 32-byte header (size, the caller's return address, a canary) and a 16-byte tail canary; both are verified at
       free(), at realloc() and by heap_guard_sweep(), which walks EVERY live block (call it from Python between steps through ctypes:
       ctypes.CDLL(None).heap_guard_sweep(b"after fit 2")) -- a write behind a block is reported with the library that allocated the block;
     * freed blocks are filled with 0xDD and parked in a quarantine (HEAP_GUARD_QUARANTINE blocks, default 16384, blocks over 64 KB pass
       straight through); the fill is verified when a block leaves the quarantine and by the sweep -- a write through a stale pointer is
       reported with the library that had allocated the block;
     * a free() of a pointer this guard never handed out is reported with a backtrace before glibc sees it.

   Build and use (host code only, nothing here touches the GPU):
     gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl
     LD_PRELOAD=/tmp/heap_guard.so python tools/flake_seq.py test_gpu_eval_latents.py
   Self-test on the CPU: tools/heap_guard_selftest.sh */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <errno.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

extern void* __libc_malloc(size_t);
extern void __libc_free(void*);
extern void* __libc_memalign(size_t, size_t);

#define HDR 32
#define TAIL 16
#define MAGIC 0x48454150475541ull /* "HEAPGUA" */
#define TAIL_BYTE 0xA5
#define DEAD_BYTE 0xDD
#define QUARANTINE_MAX_BLOCK (64u << 10)

typedef struct Block {
  uint64_t magic; /* MAGIC ^ user pointer; ~ of it once freed */
  uint64_t size;  /* bytes the caller asked for */
  void* caller;   /* return address of the allocating call */
  void* base;     /* what glibc handed out */
} Block;          /* sits in the 32 bytes in front of the user pointer */

static atomic_flag g_lock = ATOMIC_FLAG_INIT;
static void** g_live = 0; /* registry of live blocks: open-addressed set of user pointers (0 = empty, 1 = tombstone), grown by doubling */
static size_t g_live_cap = 0, g_live_n = 0, g_live_used = 0;
static void** g_quar = 0; /* ring of freed user pointers */
static size_t g_quar_cap = 0, g_quar_head = 0, g_quar_n = 0;
static long g_violations = 0;
static __thread int g_inside = 0; /* re-entrancy (backtrace_symbols_fd, dladdr allocate) */

static void lock(void) { while (atomic_flag_test_and_set_explicit(&g_lock, memory_order_acquire)) { } }
static void unlock(void) { atomic_flag_clear_explicit(&g_lock, memory_order_release); }

static void say(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
#include <stdarg.h>
static void say(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (n > (int)sizeof buf - 1) n = sizeof buf - 1;
  if (n > 0) (void)!write(2, buf, n);
}

static void name_of(void* addr, char* out, size_t cap) {
  Dl_info info;
  g_inside++;
  if (addr && dladdr(addr, &info) && info.dli_fname) {
    const char* slash = strrchr(info.dli_fname, '/');
    snprintf(out, cap, "%s+0x%lx%s%s", slash ? slash + 1 : info.dli_fname, (unsigned long)((char*)addr - (char*)info.dli_fbase),
             info.dli_sname ? " " : "", info.dli_sname ? info.dli_sname : "");
  } else {
    snprintf(out, cap, "%p", addr);
  }
  g_inside--;
}

static void backtrace_here(void) {
  void* frames[48];
  g_inside++;
  int n = backtrace(frames, 48);
  backtrace_symbols_fd(frames, n, 2);
  g_inside--;
}

/* ---- registry (under the lock) ---- */
static size_t slot_of(void* p, size_t cap) { return (size_t)(((uintptr_t)p >> 4) * 0x9E3779B97F4A7C15ull) & (cap - 1); }

static void live_grow(void) {
  size_t ncap = g_live_cap ? g_live_cap * 2 : (1u << 16);
  void** nt = (void**)__libc_malloc(ncap * sizeof(void*));
  memset(nt, 0, ncap * sizeof(void*));
  for (size_t i = 0; i < g_live_cap; ++i) {
    void* p = g_live[i];
    if ((uintptr_t)p > 1) {
      size_t s = slot_of(p, ncap);
      while (nt[s]) s = (s + 1) & (ncap - 1);
      nt[s] = p;
    }
  }
  if (g_live) __libc_free(g_live);
  g_live = nt;
  g_live_cap = ncap;
  g_live_used = g_live_n;
}

static void live_add(void* p) {
  if ((g_live_used + 1) * 2 > g_live_cap) live_grow();
  size_t s = slot_of(p, g_live_cap);
  while ((uintptr_t)g_live[s] > 1) s = (s + 1) & (g_live_cap - 1);
  if (g_live[s] == 0) g_live_used++;
  g_live[s] = p;
  g_live_n++;
}

static int live_remove(void* p) {
  if (!g_live_cap) return 0;
  size_t s = slot_of(p, g_live_cap);
  while (g_live[s]) {
    if (g_live[s] == p) { g_live[s] = (void*)1; g_live_n--; return 1; }
    s = (s + 1) & (g_live_cap - 1);
  }
  return 0;
}

/* ---- checks ---- */
static int check_live(void* user, const char* when) {
  Block* b = (Block*)((char*)user - HDR);
  int bad = 0;
  char who[256];
  if (b->magic != (MAGIC ^ (uint64_t)(uintptr_t)user)) {
    say("heap_guard: %s: HEADER of block %p damaged (a write in FRONT of it, or behind its left neighbour)\n", when, user);
    return 1;
  }
  const unsigned char* t = (const unsigned char*)user + b->size;
  for (int i = 0; i < TAIL; ++i)
    if (t[i] != TAIL_BYTE) { bad = 1; break; }
  if (bad) {
    name_of(b->caller, who, sizeof who);
    say("heap_guard: %s: write BEHIND block %p of %lu bytes allocated by %s; tail:", when, user, (unsigned long)b->size, who);
    for (int i = 0; i < TAIL; ++i) say(" %02x", t[i]);
    say("\n");
  }
  return bad;
}

static int check_dead(void* user, const char* when) {
  Block* b = (Block*)((char*)user - HDR);
  char who[256];
  if (b->magic != ~(MAGIC ^ (uint64_t)(uintptr_t)user)) {
    say("heap_guard: %s: HEADER of freed block %p damaged\n", when, user);
    return 1;
  }
  const unsigned char* u = (const unsigned char*)user;
  for (size_t i = 0; i < b->size + TAIL; ++i)
    if (u[i] != (i < b->size ? DEAD_BYTE : TAIL_BYTE)) {
      name_of(b->caller, who, sizeof who);
      say("heap_guard: %s: write THROUGH A STALE POINTER into freed block %p (%lu bytes, allocated by %s) at offset %lu: %02x\n", when, user,
          (unsigned long)b->size, who, (unsigned long)i, u[i]);
      return 1;
    }
  return 0;
}

static void violation(void) {
  g_violations++;
  backtrace_here();
  if (!getenv("HEAP_GUARD_CONTINUE")) abort();
}

/* ---- fenced blocks: HEAP_GUARD_FENCE_SIZE=<n> gives every allocation of exactly n bytes a page of its own (mmap); free() makes the page
   inaccessible for good instead of parking the block in the quarantine, so a write -- or read -- through a stale pointer FAULTS at the
   instruction that does it: the SIGSEGV handler prints the faulting address and the native backtrace.  free() of a fenced block prints
   who frees it.  (Found with it in round 6: the 920-byte roc::VirtualGPU of a destroyed stream, decremented at offset 152 after its free.) */
#include <signal.h>
#include <sys/mman.h>
#define FENCE_MAGIC 0x46454e4345444221ull /* "FENCEDB!" */
static size_t g_fence_size = 0;
static long g_fence_n = 0;

static void fence_segv(int sig, siginfo_t* si, void* uc) {
  (void)uc;
  say("heap_guard: signal %d at address %p -- an access to a FREED fenced block if the address lies in one of the pages listed at their free; native backtrace:\n", sig, si ? si->si_addr : 0);
  backtrace_here();
  signal(sig, SIG_DFL);
  raise(sig);
}

static void* fence_alloc(size_t n, void* caller) {
  size_t len = (HDR + n + TAIL + 4095) & ~(size_t)4095;
  char* page = (char*)mmap(0, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (page == MAP_FAILED) return 0;
  char* user = page + HDR;
  Block* b = (Block*)page;
  b->magic = FENCE_MAGIC ^ (uint64_t)(uintptr_t)user;
  b->size = n;
  b->caller = caller;
  b->base = page;
  memset(user + n, TAIL_BYTE, TAIL);
  __atomic_add_fetch(&g_fence_n, 1, __ATOMIC_RELAXED);
  return user;
}

static int fence_is(void* user) {
  if (!g_fence_size || ((uintptr_t)user & 4095) != HDR) return 0;
  Block* b = (Block*)((char*)user - HDR);
  return b->magic == (FENCE_MAGIC ^ (uint64_t)(uintptr_t)user);
}

static void fence_free(void* user, void* caller) {
  Block* b = (Block*)((char*)user - HDR);
  char who[256], by[256];
  name_of(b->caller, who, sizeof who);
  name_of(caller, by, sizeof by);
  size_t len = (HDR + b->size + TAIL + 4095) & ~(size_t)4095;
  say("heap_guard: fenced block %p (%lu bytes, allocated by %s) FREED by %s; page %p..%p is now inaccessible; backtrace of the free:\n", user,
      (unsigned long)b->size, who, by, (void*)b->base, (void*)((char*)b->base + len));
  backtrace_here();
  mprotect(b->base, len, PROT_NONE);
}

/* ---- allocation ---- */
static void* guard_alloc(size_t align, size_t n, void* caller) {
  if (n > (SIZE_MAX >> 1)) { errno = ENOMEM; return 0; }
  if (g_fence_size && n == g_fence_size && align <= 16 && !g_inside) return fence_alloc(n, caller);
  size_t lead = align > HDR ? align : HDR;
  char* base = (char*)(align > 16 ? __libc_memalign(lead, lead + n + TAIL) : __libc_malloc(lead + n + TAIL));
  if (!base) return 0;
  char* user = base + lead;
  Block* b = (Block*)(user - HDR);
  b->magic = MAGIC ^ (uint64_t)(uintptr_t)user;
  b->size = n;
  b->caller = caller;
  b->base = base;
  memset(user + n, TAIL_BYTE, TAIL);
  lock();
  live_add(user);
  unlock();
  return user;
}

static void guard_free(void* user, void* caller) {
  if (!user) return;
  if (fence_is(user)) { fence_free(user, caller); return; }
  lock();
  int known = live_remove(user);
  unlock();
  if (!known) {
    if (g_inside) { __libc_free(user); return; }
    char who[256];
    name_of(caller, who, sizeof who);
    say("heap_guard: free(%p) by %s of a pointer this guard never handed out (or a second free)\n", user, who);
    g_violations++;
    backtrace_here();
    if (!getenv("HEAP_GUARD_CONTINUE")) abort();
    return; /* leak it */
  }
  if (check_live(user, "free")) violation();
  Block* b = (Block*)((char*)user - HDR);
  b->magic = ~b->magic;
  if (b->size > QUARANTINE_MAX_BLOCK || g_quar_cap == 0) { __libc_free(b->base); return; }
  memset(user, DEAD_BYTE, b->size);
  void* evict = 0;
  lock();
  if (g_quar_n == g_quar_cap) {
    evict = g_quar[g_quar_head];
    g_quar[g_quar_head] = user;
    g_quar_head = (g_quar_head + 1) % g_quar_cap;
  } else {
    g_quar[(g_quar_head + g_quar_n++) % g_quar_cap] = user;
  }
  unlock();
  if (evict) {
    if (check_dead(evict, "leaving the quarantine")) violation();
    __libc_free(((Block*)((char*)evict - HDR))->base);
  }
}

__attribute__((constructor)) static void heap_guard_init(void) {
  const char* q = getenv("HEAP_GUARD_QUARANTINE");
  size_t cap = q ? (size_t)strtoul(q, 0, 10) : 16384;
  if (cap) {
    void** ring = (void**)__libc_malloc(cap * sizeof(void*));
    memset(ring, 0, cap * sizeof(void*));
    g_quar = ring;
    g_quar_cap = cap;
  }
  void* warm[4];
  g_inside++;
  backtrace(warm, 4); /* loads libgcc's unwinder now */
  g_inside--;
  const char* f = getenv("HEAP_GUARD_FENCE_SIZE");
  if (f && strtoul(f, 0, 10) > 0) {
    g_fence_size = (size_t)strtoul(f, 0, 10);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = fence_segv;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
  }
}

/* quiet forms of the two checks (no report, no allocation): usable under the lock */
static int live_ok(void* user) {
  Block* b = (Block*)((char*)user - HDR);
  if (b->magic != (MAGIC ^ (uint64_t)(uintptr_t)user)) return 0;
  const unsigned char* t = (const unsigned char*)user + b->size;
  for (int i = 0; i < TAIL; ++i)
    if (t[i] != TAIL_BYTE) return 0;
  return 1;
}
static int dead_ok(void* user) {
  Block* b = (Block*)((char*)user - HDR);
  if (b->magic != ~(MAGIC ^ (uint64_t)(uintptr_t)user)) return 0;
  const unsigned char* u = (const unsigned char*)user;
  for (size_t i = 0; i < b->size + TAIL; ++i)
    if (u[i] != (i < b->size ? DEAD_BYTE : TAIL_BYTE)) return 0;
  return 1;
}

/* every live block and every quarantined block; -> violations found (printed); never aborts.  The checks run UNDER the lock (other
   threads -- the GPU runtime's, torch's -- allocate and free all the time: a block freed between a snapshot and its check looks
   damaged); a free() of another thread waits at its registry update meanwhile.  The reports (which allocate) come after the unlock. */
long heap_guard_sweep(const char* tag) {
  enum { MAXBAD = 16 };
  void* bad_live[MAXBAD];
  void* bad_dead[MAXBAD];
  long n_bad_live = 0, n_bad_dead = 0;
  lock();
  size_t n_live = g_live_n, n_quar = g_quar_n;
  for (size_t i = 0; i < g_live_cap; ++i)
    if ((uintptr_t)g_live[i] > 1 && !live_ok(g_live[i])) { if (n_bad_live < MAXBAD) bad_live[n_bad_live] = g_live[i]; n_bad_live++; }
  for (size_t i = 0; i < n_quar; ++i) {
    void* u = g_quar[(g_quar_head + i) % g_quar_cap];
    if (!dead_ok(u)) { if (n_bad_dead < MAXBAD) bad_dead[n_bad_dead] = u; n_bad_dead++; }
  }
  unlock();
  long bad = n_bad_live + n_bad_dead;
  for (long i = 0; i < n_bad_live && i < MAXBAD; ++i) check_live(bad_live[i], tag ? tag : "sweep");
  for (long i = 0; i < n_bad_dead && i < MAXBAD; ++i) check_dead(bad_dead[i], tag ? tag : "sweep");
  if (bad || getenv("HEAP_GUARD_VERBOSE")) say("heap_guard: sweep '%s': %lu live blocks, %lu quarantined, %ld damaged\n", tag ? tag : "", (unsigned long)n_live, (unsigned long)n_quar, bad);
  g_violations += bad;
  return bad;
}

long heap_guard_violations(void) { return g_violations; }

/* ---- the interposed entry points ---- */
void* malloc(size_t n) { return guard_alloc(16, n, __builtin_return_address(0)); }
void free(void* p) { guard_free(p, __builtin_return_address(0)); }
void cfree(void* p) { guard_free(p, __builtin_return_address(0)); }

void* calloc(size_t a, size_t b) {
  size_t n;
  if (__builtin_mul_overflow(a, b, &n)) { errno = ENOMEM; return 0; }
  void* p = guard_alloc(16, n, __builtin_return_address(0));
  if (p) memset(p, 0, n);
  return p;
}

void* realloc(void* p, size_t n) {
  void* caller = __builtin_return_address(0);
  if (!p) return guard_alloc(16, n, caller);
  if (n == 0) { guard_free(p, caller); return 0; }
  Block* b = (Block*)((char*)p - HDR);
  if (fence_is(p)) {
    void* q2 = guard_alloc(16, n, caller);
    if (!q2) return 0;
    memcpy(q2, p, b->size < n ? b->size : n);
    fence_free(p, caller);
    return q2;
  }
  if (b->magic != (MAGIC ^ (uint64_t)(uintptr_t)p)) { /* not ours or damaged: let free() say which */
    guard_free(p, caller);
    return guard_alloc(16, n, caller);
  }
  void* q = guard_alloc(16, n, caller);
  if (!q) return 0;
  memcpy(q, p, b->size < n ? b->size : n);
  guard_free(p, caller);
  return q;
}

void* memalign(size_t align, size_t n) { return guard_alloc(align, n, __builtin_return_address(0)); }
void* aligned_alloc(size_t align, size_t n) { return guard_alloc(align, n, __builtin_return_address(0)); }
void* valloc(size_t n) { return guard_alloc((size_t)sysconf(_SC_PAGESIZE), n, __builtin_return_address(0)); }
void* pvalloc(size_t n) {
  size_t pg = (size_t)sysconf(_SC_PAGESIZE);
  return guard_alloc(pg, (n + pg - 1) / pg * pg, __builtin_return_address(0));
}
int posix_memalign(void** out, size_t align, size_t n) {
  if (align < sizeof(void*) || (align & (align - 1))) return EINVAL;
  void* p = guard_alloc(align, n, __builtin_return_address(0));
  if (!p) return ENOMEM;
  *out = p;
  return 0;
}
size_t malloc_usable_size(void* p) {
  if (!p) return 0;
  Block* b = (Block*)((char*)p - HDR);
  if (fence_is(p)) return b->size;
  return b->magic == (MAGIC ^ (uint64_t)(uintptr_t)p) ? b->size : 0;
}
