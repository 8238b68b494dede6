/* malloc_pad.c -- TEST INFRASTRUCTURE ONLY (LD_PRELOAD shim for tests/test_ref_loaders.py).
 *
 * The reference's ScenarioTree constructor allocates N / N+1 ints for nodesPerStage / nodesPerStageCumul and then
 * copies the N+1 / N+2 entries its JSON carries (/root/reference/src/ScenarioTree.cu:66-75): a 4-byte heap overrun
 * that glibc reports as "malloc(): invalid size (unsorted)" whenever the neighbouring chunk is inspected later
 * (deterministic for N = 5, the `toy` problem).  The child process that runs the reference's loaders is started with
 * this shim preloaded: every request is padded by 64 bytes, so the overrun lands in padding and the cross-check of
 * the JSON formats does not depend on the heap layout.  Nothing in the product loads this file. */
#include <stddef.h>

extern void *__libc_malloc(size_t);
extern void *__libc_calloc(size_t, size_t);
extern void *__libc_realloc(void *, size_t);
extern void *__libc_memalign(size_t, size_t);

#define PAD 64

void *malloc(size_t n) { return __libc_malloc(n + PAD); }
void *calloc(size_t a, size_t b) {
    if (b != 0 && a > ((size_t)-1 - PAD) / b) return 0;
    return __libc_calloc(1, a * b + PAD);
}
void *realloc(void *p, size_t n) { return __libc_realloc(p, n + PAD); }
void *memalign(size_t al, size_t n) { return __libc_memalign(al, n + PAD); }
void *aligned_alloc(size_t al, size_t n) { return __libc_memalign(al, n + PAD); }
int posix_memalign(void **out, size_t al, size_t n) {
    void *p = __libc_memalign(al, n + PAD);
    if (!p) return 12; /* ENOMEM */
    *out = p;
    return 0;
}
