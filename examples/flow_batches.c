/* Many batches through ONE context with two in flight (include/latok_hip.h "batch flow"): what a C caller does whose batches
 * already live in device memory -- a loader thread fills HBM, this thread tokenizes.  Every batch is submitted with
 * latok_flow_token_spans (nothing waits for the item total), the results are read after one latok_flow_wait: per batch the
 * two result words (token total, error word), the per-string token counts and the span records.
 * The reference tokenizes one str per call (latok/core/default_tokenizer.py:137-160); a batch here is a list of such strings
 * as UTF-8 in byte space, and the printed tokens are the reference's.
 *   gcc -std=c99 -Iinclude examples/flow_batches.c -Llatok_amd -llatok_hip -Wl,-rpath,$PWD/latok_amd -o /tmp/flow_batches */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "latok_hip.h"

#define N_BATCH 5
#define CHECK(call)                                                          \
    do {                                                                     \
        if ((call) != LATOK_OK) {                                            \
            fprintf(stderr, "%s: %s\n", #call, latok_last_error());          \
            return 1;                                                        \
        }                                                                    \
    } while (0)

typedef struct {
    const char* const* texts;
    int n;
    int64_t byte_off[8];
    void *d_u8, *d_off, *d_counts, *d_spans, *d_result;
    int64_t cap;
} batch_t;

int main(void) {
    static const char* const b0[] = {"This is a #test! Testing, Testing, 1 2 3"};
    static const char* const b1[] = {"see http://a.b/c or mail me@x.org", "camelCase \xE6\x97\xA5\xE6\x9C\xAC\xE8\xAA\x9E \xF0\x9F\xA4\x93"};
    static const char* const b2[] = {"", "x", "  "};
    static const char* const b3[] = {"foo@bar.com, .@user hi", "$#@^:a./", "camelCaseXMLParser"};
    static const char* const b4[] = {"one more batch: the flow takes any number"};
    batch_t B[N_BATCH] = {{b0, 1}, {b1, 2}, {b2, 3}, {b3, 3}, {b4, 1}};
    CHECK(latok_init(0));
    for (int k = 0; k < N_BATCH; ++k) {               /* the batches become device resident (the loader's job) */
        batch_t* b = &B[k];
        b->byte_off[0] = 0;
        for (int i = 0; i < b->n; ++i) b->byte_off[i + 1] = b->byte_off[i] + (int64_t)strlen(b->texts[i]);
        const int64_t bytes = b->byte_off[b->n];
        char* joined = (char*)malloc((size_t)bytes + 1);
        for (int i = 0; i < b->n; ++i) memcpy(joined + b->byte_off[i], b->texts[i], strlen(b->texts[i]));
        b->cap = bytes > 0 ? bytes : 1;               /* a string of n bytes has at most n tokens */
        b->d_u8 = latok_dev_alloc((size_t)bytes + 16);
        b->d_off = latok_dev_alloc((size_t)(b->n + 1) * 8);
        b->d_counts = latok_dev_alloc((size_t)b->n * 4 + 16);
        b->d_spans = latok_dev_alloc((size_t)b->cap * 8 + 16);
        b->d_result = latok_dev_alloc(16);
        if (!b->d_u8 || !b->d_off || !b->d_counts || !b->d_spans || !b->d_result) return 1;
        if (bytes > 0) CHECK(latok_memcpy_h2d(b->d_u8, joined, (size_t)bytes));
        CHECK(latok_memcpy_h2d(b->d_off, b->byte_off, (size_t)(b->n + 1) * 8));
        free(joined);
    }
    /* submit everything, wait once */
    for (int k = 0; k < N_BATCH; ++k)
        CHECK(latok_flow_token_spans(B[k].d_u8, 0 /* UTF-8 bytes, byte space */, (const int64_t*)B[k].d_off, B[k].n, B[k].byte_off[B[k].n],
                                     B[k].d_counts, B[k].d_spans, B[k].cap, (int64_t*)B[k].d_result, LATOK_OUT_INT32));
    CHECK(latok_flow_wait());
    for (int k = 0; k < N_BATCH; ++k) {
        batch_t* b = &B[k];
        int64_t result[2];
        CHECK(latok_memcpy_d2h(result, b->d_result, 16));
        if (result[1] != 0 || result[0] > b->cap) {
            fprintf(stderr, "batch %d: total %lld, error word %lld\n", k, (long long)result[0], (long long)result[1]);
            return 1;
        }
        int32_t counts[8];
        int32_t* spans = (int32_t*)malloc((size_t)(result[0] > 0 ? result[0] : 1) * 8);
        CHECK(latok_memcpy_d2h(counts, b->d_counts, (size_t)b->n * 4));
        if (result[0] > 0) CHECK(latok_memcpy_d2h(spans, b->d_spans, (size_t)result[0] * 8));
        int64_t t = 0;
        for (int i = 0; i < b->n; ++i) {
            printf("%d.%d:", k, i);
            for (int c = 0; c < counts[i]; ++c, ++t) printf(" [%.*s]", (int)(spans[2 * t + 1] - spans[2 * t]), b->texts[i] + spans[2 * t]);
            printf("\n");
        }
        free(spans);
        latok_dev_free(b->d_u8); latok_dev_free(b->d_off); latok_dev_free(b->d_counts); latok_dev_free(b->d_spans); latok_dev_free(b->d_result);
    }
    CHECK(latok_shutdown());
    return 0;
}
