/* One string per call, from plain C: the reference's calling pattern (default_tokenizer.py:137-160 tokenizes ONE str) on
 * the C ABI.  Every argument is tokenized on its own with latok_token_spans_batch (UTF-32 code points in, stripped token
 * spans out, int32 records), the tokens are printed, and the call is timed over `reps` repetitions.
 *   gcc -std=c99 -O2 -Iinclude examples/tokenize_one.c -Llatok_amd -llatok_hip -Wl,-rpath,$PWD/latok_amd -o /tmp/tokenize_one
 *   /tmp/tokenize_one [reps] "first string" "second string" ...
 * A string of up to 4096 chars is one single-wavefront launch in the library; the call returns when the kernel's completion
 * word has arrived in pinned memory.  Needs a HIP device at run time (there is no CPU fallback). */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "latok_hip.h"

/* UTF-8 -> code points (valid input assumed); returns the number of code points */
static int64_t decode_utf8(const char* s, uint32_t* out) {
    int64_t n = 0;
    for (const unsigned char* p = (const unsigned char*)s; *p;) {
        uint32_t c = *p++;
        int more = c >= 0xF0 ? 3 : c >= 0xE0 ? 2 : c >= 0xC0 ? 1 : 0;
        if (more) c &= 0x3Fu >> more;
        while (more-- > 0 && *p) c = (c << 6) | (*p++ & 0x3Fu);
        out[n++] = c;
    }
    return n;
}

/* print code points [a, b) as UTF-8 */
static void print_cps(const uint32_t* cps, int64_t a, int64_t b) {
    for (int64_t i = a; i < b; ++i) {
        const uint32_t c = cps[i];
        if (c < 0x80) putchar((int)c);
        else if (c < 0x800) { putchar(0xC0 | (c >> 6)); putchar(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { putchar(0xE0 | (c >> 12)); putchar(0x80 | ((c >> 6) & 0x3F)); putchar(0x80 | (c & 0x3F)); }
        else { putchar(0xF0 | (c >> 18)); putchar(0x80 | ((c >> 12) & 0x3F)); putchar(0x80 | ((c >> 6) & 0x3F)); putchar(0x80 | (c & 0x3F)); }
    }
}

int main(int argc, char** argv) {
    int first = 1, reps = 1000;
    if (argc > 1 && argv[1][0] >= '0' && argv[1][0] <= '9') { reps = atoi(argv[1]); first = 2; }
    const char* dflt[] = {"This is a #test! Testing, Testing, 1 2 3 -- see http://example.com/x or mail bob@host.org, camelCaseWord."};
    const char** texts = first < argc ? (const char**)(argv + first) : dflt;
    const int n_texts = first < argc ? argc - first : 1;
    if (latok_init(0) != LATOK_OK) {
        fprintf(stderr, "latok_init: %s\n", latok_last_error());
        return 1;
    }
    for (int t = 0; t < n_texts; ++t) {
        const size_t bytes = strlen(texts[t]);
        uint32_t* cps = (uint32_t*)malloc((bytes + 1) * sizeof(uint32_t));
        int32_t* spans = (int32_t*)malloc((bytes + 1) * 2 * sizeof(int32_t));
        const int64_t n = decode_utf8(texts[t], cps);
        const int64_t row[2] = {0, n};
        int32_t count = 0;
        int64_t n_tok = 0;
        if (n == 0) { printf("%d:\n", t); free(cps); free(spans); continue; }
        struct timespec t0, t1;
        int rc = LATOK_OK;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int r = 0; r < reps && rc == LATOK_OK; ++r)
            rc = latok_token_spans_batch(cps, row, 1, n, (int64_t*)&count, (int64_t*)spans, n, &n_tok, LATOK_OUT_INT32, NULL);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (rc != LATOK_OK) {
            fprintf(stderr, "latok_token_spans_batch: %s\n", latok_last_error());
            return 1;
        }
        printf("%d:", t);
        for (int64_t k = 0; k < n_tok; ++k) {
            printf(" [");
            print_cps(cps, spans[2 * k], spans[2 * k + 1]);
            printf("]");
        }
        printf("\n");
        fprintf(stderr, "string %d: %lld chars, %lld tokens, %.1f us per call (%d calls)\n", t, (long long)n, (long long)n_tok,
                ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1e3 / reps, reps);
        free(cps);
        free(spans);
    }
    latok_shutdown();
    return 0;
}
