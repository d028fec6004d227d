/* The reference's per-record loop against the C-ABI, from C: `calls` x kmerminhash_add_sequence of random `length`-base
 * records (force = false) into one k=31 scaled=1000 sketch, then kmerminhash_get_mins_size (which settles the queued
 * records).  Prints "<seconds> <hashes>".  usage: small_calls_loop <libsourmash_amd.so> <length> <calls> */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

typedef void* (*new_fn)(uint64_t, uint32_t, uint32_t, uint64_t, int, uint32_t);
typedef void (*add_fn)(void*, const char*, int);
typedef uintptr_t (*size_fn)(const void*);
typedef uint32_t (*code_fn)(void);

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    new_fn mk = (new_fn)dlsym(lib, "kmerminhash_new");
    add_fn add = (add_fn)dlsym(lib, "kmerminhash_add_sequence");
    size_fn size = (size_fn)dlsym(lib, "kmerminhash_get_mins_size");
    code_fn code = (code_fn)dlsym(lib, "sourmash_err_get_last_code");
    const long length = atol(argv[2]), calls = atol(argv[3]);
    const int n_seq = 256;
    char** seqs = malloc(sizeof(char*) * n_seq);
    uint64_t x = 88172645463325252ull;
    for (int s = 0; s < n_seq; ++s) {
        seqs[s] = malloc(length + 1);
        for (long i = 0; i < length; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; seqs[s][i] = "ACGT"[x & 3]; }
        seqs[s][length] = 0;
    }
    void* warm = mk(1000, 31, 1, 42, 0, 0);
    add(warm, seqs[0], 0);
    (void)size(warm);                                   /* device context, code objects */
    void* mh = mk(1000, 31, 1, 42, 0, 0);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long i = 0; i < calls; ++i) add(mh, seqs[i % n_seq], 0);
    const uintptr_t n = size(mh);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (code()) { fprintf(stderr, "error code %u\n", code()); return 1; }
    printf("%.6f %lu\n", (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec), (unsigned long)n);
    return 0;
}
