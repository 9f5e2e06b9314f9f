/* A plain-C client of include/optable_hip.h: no Python, no torch — device memory comes from the HIP runtime.
 * Scene: one circular mirror (radius 1) at x = 2 facing the source, written as a single ot_node the way the
 * scene compiler would.  Four rays leave the origin along +x with small tilts; each must come back:
 * segment 0 ends on the mirror (surface 0, length 2/cos), segment 1 escapes (surface -1, length +inf) with
 * dx mirrored.  Build (tests/test_gpu_abi_client.py does this):
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/abi_client.c \
 *       -Loptable_amd/csrc -loptable_hip -L/opt/rocm/lib -lamdhip64 -lm
 * The reference-side analogue is a ctypes binding; this file shows the boundary carries only
 * C types. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "optable_hip.h"

#define N 4
#define K 2
#define KT 4   /* cap of the ray-tree trace below: max_trace_num */
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_OT(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "optable_hip error %d: %s (%s:%d)\n", r_, ot_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static int upload(void** dev, const void* host, size_t bytes) {
    CHECK_HIP(hipMalloc(dev, bytes));
    if (host) CHECK_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return 0;
}

int main(void) {
    if (ot_abi_version() != OT_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    ot_ctx* ctx = NULL;
    CHECK_OT(ot_ctx_create(0, NULL, &ctx));

    ot_node mirror;
    memset(&mirror, 0, sizeof mirror);
    mirror.M[0] = -1.0; mirror.M[4] = -1.0; mirror.M[8] = 1.0;   /* RotZ(pi): the local +x normal faces the source */
    mirror.origin[0] = 2.0;
    mirror.p[0] = 1.0;                                           /* OT_SHAPE_CIRCLE: radius */
    mirror.reflectivity = 1.0;
    mirror.kind = OT_NODE_LEAF; mirror.end = 1; mirror.flags = 0; mirror.shape = OT_SHAPE_CIRCLE;
    mirror.interaction = OT_INT_MIRROR; mirror.mat1 = mirror.mat2 = -1; mirror.roc_kind = OT_ROC_INF;
    mirror.max_interact_count = -1; mirror.count_slot = -1; mirror.aux = -1; mirror.leaf_id = 0;
    ot_scene_desc scene;
    memset(&scene, 0, sizeof scene);
    scene.nodes = &mirror; scene.n_nodes = 1; scene.max_children = 1; scene.unit = 1e-2; scene.root_grid = -1;
    CHECK_OT(ot_scene_upload(ctx, &scene));

    double h[12][N];
    int32_t id[N], flags[N];
    memset(h, 0, sizeof h);
    for (int i = 0; i < N; ++i) {
        const double ty = 0.05 * i, norm = sqrt(1.0 + ty * ty);
        h[3][i] = 1.0 / norm; h[4][i] = ty / norm;               /* direction */
        h[6][i] = 780e-7;                                        /* wavelength */
        h[9][i] = 1.0; h[10][i] = 1.0;                           /* intensity, n */
        id[i] = i; flags[i] = 0;
    }
    ot_rays rays;
    memset(&rays, 0, sizeof rays);
    void** rf[12] = {&rays.ox, &rays.oy, &rays.oz, &rays.dx, &rays.dy, &rays.dz, &rays.wavelength, &rays.q_re, &rays.q_im,
                     &rays.intensity, &rays.n, &rays.pathlength};
    for (int f = 0; f < 12; ++f) if (upload(rf[f], h[f], sizeof(double) * N)) return 2;
    if (upload((void**)&rays.id, id, sizeof id) || upload((void**)&rays.flags, flags, sizeof flags)) return 2;

    ot_segments segs;
    memset(&segs, 0, sizeof segs);
    void** sf[12] = {&segs.ox, &segs.oy, &segs.oz, &segs.dx, &segs.dy, &segs.dz, &segs.length, &segs.intensity, &segs.q_re,
                     &segs.q_im, &segs.n, &segs.pathlength};
    for (int f = 0; f < 12; ++f) if (upload(sf[f], NULL, sizeof(double) * N * KT)) return 2;   /* (room for the ray trees below) */
    if (upload((void**)&segs.ray, NULL, sizeof(int32_t) * N * KT) || upload((void**)&segs.surface, NULL, sizeof(int32_t) * N * KT)) return 2;
    int32_t* seg_count = NULL;
    if (upload((void**)&seg_count, NULL, sizeof(int32_t) * N)) return 2;

    CHECK_OT(ot_trace_f64(ctx, &rays, N, K, &segs, seg_count, NULL, 0));
    CHECK_OT(ot_ctx_synchronize(ctx));

    double len[N * K], dx[N * K], ox[N * K];
    int32_t surf[N * K], cnt[N];
    CHECK_HIP(hipMemcpy(len, segs.length, sizeof len, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(dx, segs.dx, sizeof dx, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(ox, segs.ox, sizeof ox, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(surf, segs.surface, sizeof surf, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(cnt, seg_count, sizeof cnt, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < N; ++i) {
        const double expect = 2.0 / h[3][i];   /* distance to the plane x = 2 along the ray */
        const int k0 = i, k1 = N + i;          /* slot k*N + i */
        if (cnt[i] != 2 || surf[k0] != 0 || surf[k1] != -1) bad++;
        if (fabs(len[k0] - expect) > 1e-12 || !isinf(len[k1])) bad++;
        if (fabs(dx[k1] + h[3][i]) > 1e-15 || fabs(ox[k1] - 2.0) > 1e-12) bad++;
        printf("ray %d: hit at t = %.12f (expected %.12f), returns with dx = %+.12f\n", i, len[k0], expect, dx[k1]);
    }
    /* Ray trees: the same mirror half transmitting — every hit emits two rays (reflected, then transmitted:
     * optical_component.py:536-570).  One launch, a lane per tree, [k][tree] slots in the reference's FIFO order:
     * the input ray up to the mirror, the reflected ray, the transmitted ray. */
    mirror.reflectivity = 0.5; mirror.transmission = 0.5;
    scene.max_children = 2;
    CHECK_OT(ot_scene_upload(ctx, &scene));
    int32_t plan[8];
    CHECK_OT(ot_trace_trees_plan(ctx, 8, KT, N, plan));
    printf("lane-per-tree kernel: %s, queue entries per lane %d (%d of them in LDS), enough for every tree: %s\n",
           (plan[0] & 1) ? "yes" : "no", plan[1], plan[3], plan[2] ? "yes" : "no");
    if (!(plan[0] & 2) || !plan[2]) bad++;                /* this scene (planar preset) also writes [k][tree] slots; a cap of KT needs KT / 2 entries */
    CHECK_OT(ot_trace_trees_f64(ctx, &rays, N, KT, &segs, seg_count, NULL, 0));
    CHECK_OT(ot_ctx_synchronize(ctx));
    double inten[N * KT], tdx[N * KT];
    int32_t tsurf[N * KT];
    CHECK_HIP(hipMemcpy(tdx, segs.dx, sizeof tdx, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(inten, segs.intensity, sizeof inten, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(tsurf, segs.surface, sizeof tsurf, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(cnt, seg_count, sizeof cnt, hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i) {
        if (cnt[i] != 3 || tsurf[i] != 0 || tsurf[N + i] != -1 || tsurf[2 * N + i] != -1) bad++;
        if (fabs(tdx[N + i] + h[3][i]) > 1e-15 || fabs(tdx[2 * N + i] - h[3][i]) > 1e-15) bad++;   /* reflected, then transmitted */
        if (fabs(inten[N + i] - 0.5) > 1e-15 || fabs(inten[2 * N + i] - 0.5) > 1e-15) bad++;
    }
    printf("ray trees: %d rays per tree, children at half the intensity\n", (int)cnt[0]);
    /* error path: a NULL field is refused with a message, nothing is launched */
    ot_rays broken = rays;
    broken.q_im = NULL;
    if (ot_trace_f64(ctx, &broken, N, K, &segs, seg_count, NULL, 0) != OT_ERR_INVALID) bad++;
    printf("refused call says: %s\n", ot_last_error());
    CHECK_OT(ot_ctx_destroy(ctx));
    puts(bad ? "FAILED" : "OK");
    return bad ? 1 : 0;
}
