/* deliberate heap errors for tools/heap_guard.c (tools/heap_guard_selftest.sh): 0 clean, 1 write behind a block, 2 write through a stale pointer,
   3 second free, 4 write behind a block found by the sweep, 5 write in front of a block */
#define _GNU_SOURCE
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <dlfcn.h>
typedef long (*sweep_t)(const char*);
int main(int argc, char** argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;
  char* p = malloc(24);
  char* q = malloc(100);
  void* a; posix_memalign(&a, 256, 1000); memset(a, 1, 1000); free(a);
  q = realloc(q, 5000); memset(q, 2, 5000); free(q);
  if (mode == 1) memset(p, 0x41, 26);           /* overrun by 2 */
  if (mode == 2) { free(p); p[3] = 7; for (int i = 0; i < 40000; ++i) free(malloc(32)); puts("evicted"); return 0; } /* stale write */
  if (mode == 3) { free(p); free(p); }          /* double free */
  if (mode == 4) { memset(p, 0x41, 26); printf("sweep -> %ld\n", ((sweep_t)dlsym(RTLD_DEFAULT, "heap_guard_sweep"))("selftest")); return 0; }
  if (mode == 5) { p[-30] = 9; }                 /* underrun */
  free(p);
  puts("clean exit");
  return 0;
}
