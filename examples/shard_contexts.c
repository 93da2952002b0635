/* One batch, several library contexts, one host thread each (include/latok_hip.h "contexts"): the way a C caller
 * spreads a batch over the GPUs of a node -- or, on a single GPU as here by default, overlaps the shards' copies with each
 * other's kernels.  Strings are independent (reference tokenize() takes one str, latok/core/default_tokenizer.py:137),
 * so the batch is cut into contiguous string ranges and the shards' per-string results simply follow each other.
 *   gcc -std=c99 -pthread -Iinclude examples/shard_contexts.c -Llatok_amd -llatok_hip -Wl,-rpath,$PWD/latok_amd -o /tmp/shard_contexts
 *   /tmp/shard_contexts [device ...]            (default: two contexts on device 0)
 *   /tmp/shard_contexts resident [device ...]   the shard is uploaded ONCE into its context's HBM and the call uses device
 *                                               pointers (LATOK_DEVICE_PTRS) -- twice, to show the batch stays put; only the
 *                                               records come back.  What a caller with data already on the GPUs does.
 * Prints, per shard, the boundary offsets of its strings (np.nonzero(gen_split_mask(...)) of the reference,
 * default_tokenizer.py:146-148) as 32-bit records. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "latok_hip.h"

typedef struct {
    int device;
    const uint8_t* units;      /* Latin-1 code units of the whole batch (PEP 393 kind 1: latok.c:53-55,79) */
    const int64_t* row_off;    /* of the whole batch */
    int64_t s0, s1;            /* my strings */
    int32_t* counts;           /* [s1 - s0] */
    int32_t* offsets;          /* capacity = my chars */
    int64_t n_offsets;
    int resident;              /* keep the shard in HBM and call with device pointers */
    int rc;
    char err[256];
} shard_t;

static void* run_shard(void* arg) {
    shard_t* sh = (shard_t*)arg;
    latok_ctx* ctx = NULL;
    sh->rc = latok_ctx_create(sh->device, &ctx);
    if (sh->rc == LATOK_OK) {
        latok_ctx_set_current(ctx);                      /* every call of this thread now runs on ctx */
        const int64_t n = sh->s1 - sh->s0, c0 = sh->row_off[sh->s0], chars = sh->row_off[sh->s1] - c0;
        int64_t* row = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
        for (int64_t i = 0; i <= n; ++i) row[i] = sh->row_off[sh->s0 + i] - c0;      /* a shard's offsets start at 0 */
        if (!sh->resident) {
            sh->rc = latok_split_offsets_kind_batch(sh->units + c0, 1, row, n, chars, (int64_t*)sh->counts, (int64_t*)sh->offsets,
                                                    chars, &sh->n_offsets, LATOK_OUT_INT32, NULL);
        } else {
            /* device memory belongs to the device of the context that is current on this thread */
            void* d_units = latok_dev_alloc((size_t)(chars > 0 ? chars : 1));
            void* d_row = latok_dev_alloc((size_t)(n + 1) * sizeof(int64_t));
            void* d_counts = latok_dev_alloc((size_t)(n + 1) * sizeof(int32_t));
            void* d_offs = latok_dev_alloc((size_t)(chars > 0 ? chars : 1) * sizeof(int32_t));
            sh->rc = (d_units && d_row && d_counts && d_offs) ? LATOK_OK : LATOK_ERR_NOMEM;
            if (sh->rc == LATOK_OK) sh->rc = latok_memcpy_h2d(d_units, sh->units + c0, (size_t)chars);          /* once */
            if (sh->rc == LATOK_OK) sh->rc = latok_memcpy_h2d(d_row, row, (size_t)(n + 1) * sizeof(int64_t));
            for (int pass = 0; pass < 2 && sh->rc == LATOK_OK; ++pass)   /* any number of passes: nothing is uploaded again */
                sh->rc = latok_split_offsets_kind_batch(d_units, 1, (const int64_t*)d_row, n, chars, (int64_t*)d_counts, (int64_t*)d_offs,
                                                        chars, &sh->n_offsets, LATOK_OUT_INT32 | LATOK_DEVICE_PTRS, NULL);
            if (sh->rc == LATOK_OK) sh->rc = latok_memcpy_d2h(sh->counts, d_counts, (size_t)n * sizeof(int32_t));
            if (sh->rc == LATOK_OK && sh->n_offsets > 0)
                sh->rc = latok_memcpy_d2h(sh->offsets, d_offs, (size_t)sh->n_offsets * sizeof(int32_t));
            latok_dev_free(d_units); latok_dev_free(d_row); latok_dev_free(d_counts); latok_dev_free(d_offs);
        }
        free(row);
    }
    if (sh->rc != LATOK_OK) snprintf(sh->err, sizeof(sh->err), "%s", latok_last_error());   /* the message is per thread */
    latok_ctx_set_current(NULL);
    latok_ctx_destroy(ctx);
    return NULL;
}

int main(int argc, char** argv) {
    const char* texts[] = {"This is a #test! Testing, Testing, 1 2 3", "see http://a.b/c or mail me@x.org", "camelCaseXMLParser",
                           "foo@bar.com, .@user hi", "x\t\ny", "$#@^:a./"};
    const int64_t n_str = 6;
    const int resident = argc > 1 && strcmp(argv[1], "resident") == 0;
    if (resident) { --argc; ++argv; }
    int n_shards = argc > 1 ? argc - 1 : 2;
    if (n_shards > n_str) n_shards = (int)n_str;
    int64_t row_off[7] = {0};
    for (int i = 0; i < n_str; ++i) row_off[i + 1] = row_off[i] + (int64_t)strlen(texts[i]);
    uint8_t* units = (uint8_t*)malloc((size_t)row_off[n_str]);
    for (int i = 0; i < n_str; ++i) memcpy(units + row_off[i], texts[i], (size_t)(row_off[i + 1] - row_off[i]));

    shard_t sh[8];
    pthread_t th[8];
    if (n_shards > 8) n_shards = 8;
    for (int r = 0; r < n_shards; ++r) {   /* contiguous ranges with about the same number of chars each */
        memset(&sh[r], 0, sizeof(sh[r]));
        sh[r].device = argc > 1 ? atoi(argv[1 + r]) : 0;
        sh[r].resident = resident;
        sh[r].units = units;
        sh[r].row_off = row_off;
        sh[r].s0 = r == 0 ? 0 : sh[r - 1].s1;
        int64_t s1 = sh[r].s0;
        while (s1 < n_str && (r == n_shards - 1 || row_off[s1] < row_off[n_str] * (r + 1) / n_shards)) ++s1;
        if (s1 == sh[r].s0 && s1 < n_str) ++s1;
        sh[r].s1 = s1;
        const int64_t chars = row_off[sh[r].s1] - row_off[sh[r].s0];
        sh[r].counts = (int32_t*)calloc((size_t)(sh[r].s1 - sh[r].s0) + 1, sizeof(int32_t));
        sh[r].offsets = (int32_t*)malloc((size_t)(chars > 0 ? chars : 1) * sizeof(int32_t));
        pthread_create(&th[r], NULL, run_shard, &sh[r]);
    }
    int bad = 0;
    for (int r = 0; r < n_shards; ++r) {
        pthread_join(th[r], NULL);
        if (sh[r].rc != LATOK_OK) {
            fprintf(stderr, "shard %d on device %d: %s\n", r, sh[r].device, sh[r].err);
            bad = 1;
            continue;
        }
        int64_t k = 0;
        for (int64_t s = sh[r].s0; s < sh[r].s1; ++s) {
            printf("%lld:", (long long)s);
            for (int32_t j = 0; j < sh[r].counts[s - sh[r].s0]; ++j) printf(" %d", sh[r].offsets[k++]);
            printf("\n");
        }
        free(sh[r].counts);
        free(sh[r].offsets);
    }
    free(units);
    return bad;
}
