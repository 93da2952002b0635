/* Minimal C caller of liblatok_hip.so: tokenize a few UTF-8 strings in byte space and print the tokens.
 *   gcc -std=c99 -Iinclude examples/tokenize_utf8.c -Llatok_amd -llatok_hip -Wl,-rpath,$PWD/latok_amd -o /tmp/tokenize_utf8
 * Needs a HIP device at run time (there is no CPU fallback); compiling it only needs the header. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "latok_hip.h"

int main(void) {
    const char* texts[] = {"This is a #test! Testing, Testing, 1 2 3", "see http://a.b/c or mail me@x.org",
                           "camelCase \xE6\x97\xA5\xE6\x9C\xAC\xE8\xAA\x9E \xF0\x9F\xA4\x93"};
    const int64_t n = 3;
    int64_t off[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) off[i + 1] = off[i] + (int64_t)strlen(texts[i]);
    uint8_t* buf = (uint8_t*)malloc((size_t)off[n]);
    for (int i = 0; i < n; ++i) memcpy(buf + off[i], texts[i], (size_t)(off[i + 1] - off[i]));

    if (latok_init(0) != LATOK_OK) {
        fprintf(stderr, "latok_init: %s\n", latok_last_error());
        return 1;
    }
    int64_t counts[3], n_tok = 0;
    int64_t* spans = (int64_t*)malloc((size_t)off[n] * 2 * sizeof(int64_t));
    if (latok_token_spans_utf8_bytes_batch(buf, off, n, off[n], counts, spans, off[n], &n_tok, 0, NULL) != LATOK_OK) {
        fprintf(stderr, "latok_token_spans_utf8_bytes_batch: %s\n", latok_last_error());
        return 1;
    }
    int64_t k = 0;
    for (int i = 0; i < n; ++i) {
        printf("%d:", i);
        for (int64_t j = 0; j < counts[i]; ++j, ++k)
            printf(" [%.*s]", (int)(spans[2 * k + 1] - spans[2 * k]), (const char*)buf + off[i] + spans[2 * k]);
        printf("\n");
    }
    latok_shutdown();
    free(spans);
    free(buf);
    return 0;
}
