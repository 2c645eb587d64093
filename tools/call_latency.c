/* tools/call_latency.c — per-call latency of the host-buffer entry points from plain C (no Python in the loop):
 * what Controller::PerformCL* pays per frame at the reference's own image sizes.  Pageable host memory, as the
 * reference uses.  Build: gcc -O2 tools/call_latency.c -Iinclude -L<lib> -lmi355_imgfilter -o tools/bin/call_latency
 * usage: call_latency [iterations]                                                                         */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mi355_imgfilter.h"

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    mi355_ctx* ctx = NULL;
    if (mi355_ctx_create(0, &ctx) != MI355_OK) {
        fprintf(stderr, "no GPU\n");
        return 1;
    }
    static const int sizes[][2] = {{75, 75}, {240, 192}, {640, 480}, {640, 512}, {1023, 819}, {1280, 720}, {1920, 1080}};
    printf("%-10s %-10s %9s %9s %9s %9s %9s\n", "size", "filter", "call us", "noprof us", "write us", "kernel us", "read us");
    for (unsigned s = 0; s < sizeof(sizes) / sizeof(sizes[0]); s++) {
        const int w = sizes[s][0], h = sizes[s][1];
        uint8_t* in = malloc((size_t)w * h * 4);
        uint8_t* out = malloc((size_t)w * h * 4);
        for (size_t i = 0; i < (size_t)w * h * 4; i++)
            in[i] = (uint8_t)(i * 2654435761u >> 24) | ((i & 3) == 3 ? 0xFF : 0);
        for (int f = 0; f < 4; f++) {
            static const char* names[4] = {"gray", "sobel", "gauss k5", "gauss k17"};
            uint64_t prof[6];
            double tw = 0, tk = 0, tr = 0, t_call = 0, t_noprof = 0;
            for (int pass = 0; pass < 2; pass++) {
                uint64_t* pp = pass == 0 ? prof : NULL;
                for (int i = -10; i < iters; i++) {
                    const double t0 = now_us();
                    int rc;
                    switch (f) {
                    case 0: rc = mi355_gray_rgba8(ctx, in, out, w, h, pp); break;
                    case 1: rc = mi355_sobel_rgba8(ctx, in, out, w, h, pp); break;
                    case 2: rc = mi355_gauss_rgba8(ctx, in, out, w, h, 5, 1.5f, pp); break;
                    default: rc = mi355_gauss_rgba8(ctx, in, out, w, h, 17, 6.0f, pp); break;
                    }
                    const double t1 = now_us();
                    if (rc != MI355_OK)
                        return 2;
                    if (i < 0)
                        continue;
                    if (pass == 0) {
                        t_call += t1 - t0;
                        tw += (prof[1] - prof[0]) * 1e-3;
                        tk += (prof[3] - prof[2]) * 1e-3;
                        tr += (prof[5] - prof[4]) * 1e-3;
                    } else {
                        t_noprof += t1 - t0;
                    }
                }
            }
            char sz[32];
            snprintf(sz, sizeof(sz), "%dx%d", w, h);
            printf("%-10s %-10s %9.1f %9.1f %9.1f %9.1f %9.1f\n", sz, names[f], t_call / iters, t_noprof / iters, tw / iters,
                   tk / iters, tr / iters);
        }
        free(in);
        free(out);
    }
    mi355_ctx_destroy(ctx);
    return 0;
}
