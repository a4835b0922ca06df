/* A C host of the m3l_amd ABI (include/m3l_amd.h): what a non-Python binding of the MAE path starts from.
 *
 *   gcc -std=c99 -Iinclude examples/c_host.c -Lm3l_amd/lib -lm3l_amd -Wl,-rpath,$PWD/m3l_amd/lib -o /tmp/c_host && /tmp/c_host
 *
 * Runs the host-side half of the mask sampler (the Python-double count rule of pretrain_models.py:223-227) for BASELINE cfg 2 and for
 * the reference defaults, asks for the workspace a ViT-Tiny encoder step needs, and shows the error convention (non-zero return +
 * thread-local text, nothing thrown).  No GPU is touched, so it also runs in the build container. */
#include <stdio.h>

#include "m3l_amd.h"

int main(void) {
    char err[256];
    int c[6];
    m3l_geom cfg2 = {64, 64, 8, 3, 32, 32, 4, 3, 2, 1, 1};
    printf("m3l_amd ABI version %d\n", m3l_version());
    if (m3l_mask_counts(&cfg2, 0.75, c)) {
        m3l_last_error(err, sizeof err);
        fprintf(stderr, "m3l_mask_counts: %s\n", err);
        return 1;
    }
    printf("cfg2 mask 0.75: masked %d unmasked %d (image %d, per sensor %d) of n_img %d n_tac %d\n", c[0], c[1], c[2], c[3], c[4], c[5]);
    if (m3l_mask_counts(&cfg2, 0.95, c)) return 1;
    printf("ref  mask 0.95: masked %d unmasked %d (image %d, per sensor %d)\n", c[0], c[1], c[2], c[3]);

    m3l_tf_cfg vit_tiny = {192, 12, 3, 768, 1, 1};
    printf("encoder workspace for B=256, n=48 (bf16): %zu bytes\n", m3l_transformer_ws_bytes(&vit_tiny, 256, 48));

    m3l_tf_cfg bad = {100, 1, 1, 64, 1, 1};          /* dim not a multiple of 64: rejected, with a message */
    if (m3l_transformer_ws_bytes(&bad, 1, 1) == 0) {
        m3l_last_error(err, sizeof err);
        printf("error convention: \"%s\"\n", err);
    }
    return 0;
}
