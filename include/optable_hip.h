/*
 * optable_hip.h — C-ABI of liboptable_hip.so, the MI355X (gfx950) trace engine that
 * sits behind optable's OpticalTable.ray_tracing().
 *
 * The reference (tim4431/optable) is pure Python and has no FFI of its own; the boundary
 * this library replaces is the body of
 *     OpticalTable.ray_tracing / _single_ray_tracing      optable/optical_table.py:57-147
 * and everything that loop calls per (segment, surface):
 *     OpticalComponent.interact / intersect_point_local   optable/optical_component.py:151-233,337-378
 *     ComponentGroup.interact (AABB prune + first-min)    optable/component_group.py:93-122
 *     solve_ray_bboxes_intersections                      optable/solver.py:5-48
 *     BaseMirror / BaseRefraciveSurface / Lens .interact_local
 *                                                         optable/optical_component.py:536-570,617-717,930-948
 *     Surface.f / normal / within_boundary                optable/surfaces.py
 *     GaussianBeam.q_at_z, Ray.pathlength, Sellmeier n    optable/ray.py:17-19,145-147; material.py:106-120
 *     Monitor.record                                      optable/monitor.py:183-193
 *
 * Conventions
 *   - plain C, no torch types; every pointer in ot_rays / ot_segments is a DEVICE pointer
 *     (hipMalloc'd or a torch tensor's data_ptr()), every pointer in the scene structs is HOST.
 *   - every entry point returns 0 on success or a negative ot_status; the message for the last
 *     failure on the calling thread is ot_last_error().
 *   - the caller owns all ray / segment buffers; the library owns the uploaded scene and its
 *     scratch.  Calls on one ot_ctx are serialised on the ctx's hipStream_t.
 *   - there is NO CPU fallback in this library.
 */
#ifndef OPTABLE_HIP_H
#define OPTABLE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OT_ABI_VERSION 13  /* 13: OT_OPT_TREES_GLOBAL_IMAGE, tree kernels for scenes beyond the LDS; 12: OT_OPT_GEN_PARENT_INDEX; 11: OT_OPT_GEN_AHEAD, ot_trace_trees_*, ot_trace_trees_append_*, OT_OPT_TREES_LDS_ENTRIES, OT_OPT_TREES_REFILL_AT, OT_OPT_TREES_FLAT; 10: ot_trace_plan, ot_probe_layouts, ot_runtime_info, OT_OPT_REFILL, OT_OPT_REFILL_TICKET, OT_OPT_POOL_JITTER, OT_OPT_GEN_ONEPASS; 9: ot_trace_tree_*, OT_OPT_BLOCK_POOL, OT_OPT_GEN_DROP_DOOMED, ot_trace_append_* holes per workgroup chunk; 8: OT_SHAPE_ASPHERE_CHEB, OT_MAT_CHEB, OT_NODE_BOX_TRUSTED, ot_trace_tiled_*, ot_bench_stream_tiled_*; 3: ot_trace_generation_f32; 4: ot_bench_stream_f32; 5: OT_OPT_LIST_CAP, ray flags bits 8..31, ot_debug_generation_mismatches; 6: ot_debug_last_launch, OT_OPT_FLAT_QUEUE, OT_OPT_LDS_RECORDS; 7: ot_trace_append_*, ot_segment_block, OT_OPT_APPEND_CHUNK, OT_OPT_INSTANCING */

/* ---- status codes ------------------------------------------------------------------- */
enum ot_status {
    OT_OK = 0,
    OT_ERR_INVALID = -1,      /* bad argument / shape mismatch                            */
    OT_ERR_HIP = -2,          /* a HIP runtime call failed (message has hipGetErrorString) */
    OT_ERR_UNSUPPORTED = -3,  /* scene needs a feature this build does not have           */
    OT_ERR_CAPACITY = -4,     /* an output buffer was too small (branching trace)         */
    OT_ERR_NOSCENE = -5       /* trace called before ot_scene_upload                      */
};

/* ---- scene description (host PODs, copied by ot_scene_upload) ------------------------- */

/* node kinds: the scene is the depth-first flattening of OpticalTable.components
 * (optical_table.py:119-123 iterates top-level components in list order; a ComponentGroup
 * iterates its children in list order, component_group.py:110-120).  With a strict '<'
 * nearest-hit update in that order the first minimum wins at every level, exactly as
 * np.argmin + the strict '<' at optical_table.py:121 do. */
enum ot_node_kind { OT_NODE_GROUP = 0, OT_NODE_LEAF = 1 };

/* surface shapes in the leaf's local frame, normal along +x (optical_component.py:43-47) */
enum ot_shape_kind {
    OT_SHAPE_CIRCLE = 0,     /* surfaces.py:139-161   p[0]=radius                                   */
    OT_SHAPE_RECT = 1,       /* surfaces.py:164-209   p[0]=width/2 p[1]=height/2                    */
    OT_SHAPE_POLYGON2D = 2,  /* surfaces.py:426-568, planar=True: aux -> polygon record             */
    OT_SHAPE_POLYGON3D = 3,  /* same, planar=False (plane not x=0): general root solve              */
    OT_SHAPE_SPHERE = 4,     /* surfaces.py:284-336   p[0]=R p[1]=height                            */
    OT_SHAPE_ASPHERE_PARAM = 5, /* component_group.py:1092-1097 p[0]=aperture radius p[1]=R p[2]=kappa p[3]=a4 p[4]=a6 p[5]=a8 */
    OT_SHAPE_ASPHERE_EXACT = 6, /* component_group.py:1061-1064 p[0]=aperture radius p[1]=EFL p[2]=n */
    OT_SHAPE_CYLINDER = 7,   /* surfaces.py:212-281   p[0]=R p[1]=height/2 p[2]=theta0 p[3]=theta1  */
    OT_SHAPE_POINT = 8,      /* surfaces.py:68-86     never hit                                     */
    OT_SHAPE_CSG = 9,        /* Plane.union/subtract  surfaces.py:100-136: aux -> postfix program   */
    OT_SHAPE_ASPHERE_CHEB = 10 /* ASphere with an arbitrary sag F(r) (component_group.py:1014-1055), given as a verified
                                Chebyshev series: p[0]=aperture radius, aux -> [N, lo, hi, c[N], c'[N], c''[N]]: F, dF/dr and
                                d2F/dr2 as series in t = (2r - lo - hi) / (hi - lo); [lo, hi] covers -2h .. radius*sqrt2 + 2h,
                                h = 1e-4 radius (the reference's finite-difference step, surfaces.py:355-369)            */
};

/* interaction kinds (what interact_local does) */
enum ot_interaction_kind {
    OT_INT_MIRROR = 0,   /* BaseMirror.interact_local            optical_component.py:536-570 */
    OT_INT_REFRACT = 1,  /* BaseRefraciveSurface.interact_local  optical_component.py:617-717 */
    OT_INT_LENS = 2,     /* Lens.interact_local (thin lens)      optical_component.py:930-948 */
    OT_INT_BLOCK = 3     /* Block.interact_local (absorb)        optical_component.py:501-503 */
};

/* radius-of-curvature source for the refractive ABCD update (optical_component.py:633-639) */
enum ot_roc_kind {
    OT_ROC_INF = 0,      /* flat: ROC = +inf                                   */
    OT_ROC_CONST = 1,    /* SphereRefractive: ROC = radius (:847)              */
    OT_ROC_ASPHERE = 2   /* ASphere.roc(P) finite differences (surfaces.py:355-373) */
};

enum ot_node_flags {
    OT_NODE_CHECK_AABB = 1, /* set for groups and for children of groups (component_group.py:98-107) */
    OT_NODE_GRID = 2,       /* group: aux -> 2-D grid over the children's AABBs (acceleration only: the
                               children found through it still pass their own AABB test)             */
    OT_NODE_BOX_TRUSTED = 4 /* the node's aabb contains its geometry.  The reference caches boxes and never updates
                               them (optical_component.py:62-67): a box that went stale is still THE pass / fail gate,
                               but nothing may be skipped because of where it lies.  The kernels prune a node whose
                               box starts behind the best hit only when this flag is set; callers that are not sure
                               leave it clear (results are the same, a little slower).                  */
};

typedef struct ot_node {
    double M[9];       /* leaf: transform_matrix, local->lab rotation, row major          */
    double origin[3];  /* leaf: lab origin                                                */
    double aabb[6];    /* lab AABB (xmin,xmax,ymin,ymax,zmin,zmax) = component.bbox       */
    double lbox[6];    /* leaf, non-planar: surface.get_bbox_local()                      */
    double p[8];       /* shape parameters, see ot_shape_kind                             */
    double reflectivity;
    double transmission;
    double focal_length; /* OT_INT_LENS                                                   */
    double roc;          /* OT_ROC_CONST                                                  */
    int32_t kind;        /* ot_node_kind                                                  */
    int32_t end;         /* index one past the last descendant (leaf: own index + 1)      */
    int32_t flags;       /* ot_node_flags                                                 */
    int32_t shape;       /* ot_shape_kind                                                 */
    int32_t interaction; /* ot_interaction_kind                                           */
    int32_t mat1, mat2;  /* material table indices of _n1 (x>0 side) and _n2 (x<0 side)   */
    int32_t roc_kind;    /* ot_roc_kind                                                   */
    int32_t max_interact_count; /* <0: unlimited (optical_component.py:140-149)           */
    int32_t count_slot;  /* row in the interact-count table, -1 when unlimited            */
    int32_t aux;         /* offset (in doubles) into the aux table, -1 when unused        */
    int32_t leaf_id;     /* index among leaves (what ot_segments.surface reports)         */
} ot_node;

enum ot_material_kind {
    OT_MAT_CONST = 0,
    OT_MAT_SELLMEIER = 1,
    OT_MAT_CHEB = 2      /* n(wavelength in metres) as a verified Chebyshev series of a user function (material.py:4-21):
                            ot_material.n holds the aux offset of [N, lo, hi, c[N]]; wavelengths outside [lo, hi] are clamped
                            to the interval (callers check their rays against it: optable_amd/engine.py)                   */
};

/* material.py:48-85 (RefractiveIndex), :106-120 (Sellmeier, wavelength in metres -> microns) */
typedef struct ot_material {
    double n;        /* OT_MAT_CONST; OT_MAT_CHEB: aux offset  */
    double B[3];     /* OT_MAT_SELLMEIER                    */
    double C[3];     /* microns^2                           */
    int32_t kind;
    int32_t _pad;
} ot_material;

typedef struct ot_scene_desc {
    const ot_node* nodes;
    int32_t n_nodes;
    const ot_material* materials;
    int32_t n_materials;
    const double* aux;     /* polygon records and CSG programs, see DESIGN.md               */
    int32_t n_aux;
    int32_t n_count_slots; /* leaves with max_interact_count set                            */
    int32_t max_children;  /* most rays any interaction can emit (1 => non-branching scene) */
    double unit;           /* metres per model length unit (OpticalTable(unit=...), 1e-2)   */
    int32_t root_grid;     /* aux offset of a 2-D grid over the top-level components (walked
                              cell by cell instead of the linear pass), -1 for none.  Record:
                              [a0 a1 g0 g1 org0 org1 inv0 inv1 margin size0 size1 | start | items] */
    int32_t _pad;
} ot_scene_desc;

/* ---- ray / segment streams (device pointers, structure of arrays) -------------------- */

enum ot_ray_flags {
    OT_RAY_HAS_Q = 1,   /* qo is not None (ray.py:98-104)        */
    OT_RAY_DEAD = 2     /* alive == False on input (optical_component.py:349) */
    /* bits 8..31: zero for a caller's rays.  In the `next` buffers of ot_trace_generation_* they carry
     * (node index + 1) of the surface a ray was emitted on — a ray is never tested against the plane it
     * starts on (the reference rejects that root, t = 0, by |t| < 1e-9) — so a generation's output can be
     * handed back as the next call's input unchanged. */
};

/* One ray = one Ray object of the reference (ray.py:63-104): 12 reals + id + flags.
 * "real" is double for the _f64 entry points and float for _f32. */
typedef struct ot_rays {
    void* ox; void* oy; void* oz;   /* origin                                 */
    void* dx; void* dy; void* dz;   /* unit direction                         */
    void* wavelength;               /* model units (0 when unset)             */
    void* q_re; void* q_im;         /* Gaussian q at origin (ignored w/o HAS_Q) */
    void* intensity;
    void* n;                        /* refractive index of the medium the ray is in, _n(wavelength*unit) */
    void* pathlength;               /* _pathlength                            */
    int32_t* id;                    /* interact-count class of the ray (rays sharing a Python _id share a class) */
    int32_t* flags;                 /* ot_ray_flags                           */
    void* length;                   /* optional: finite input length, +inf = None; NULL = all None */
} ot_rays;

/* One segment = one element of the list OpticalTable.ray_tracing returns
 * (optical_table.py:125-134): the parent truncated at the hit (alive=False, length=t) or the
 * unchanged escaping ray (alive=True, length=None -> +inf here, surface = -1). */
typedef struct ot_segments {
    void* ox; void* oy; void* oz;
    void* dx; void* dy; void* dz;
    void* length;
    void* intensity;
    void* q_re; void* q_im;
    void* n;
    void* pathlength;
    int32_t* ray;       /* index of the input ray whose tree this segment belongs to */
    int32_t* surface;   /* leaf_id that terminated the segment, -1 = escaped        */
} ot_segments;

typedef struct ot_ctx ot_ctx;

/* ---- lifecycle ------------------------------------------------------------------------ */
int ot_abi_version(void);
const char* ot_last_error(void);
/* The HIP runtime this library is bound to: file name of the libamdhip64 that resolves its symbols, and hipRuntimeGetVersion.
 * A process must hold ONE HIP runtime.  The library names libamdhip64.so.7 by soname, so a copy that is already in the process
 * is used (PyTorch ships its own: load it, or `import torch`, BEFORE this library — INTEGRATION.md); loaded first and alone, the
 * library falls back to the ROCm installation's copy.  Never touches the device. */
int ot_runtime_info(char* path, int32_t path_capacity, int32_t* runtime_version);

/* stream: a hipStream_t the caller owns (e.g. torch.cuda.current_stream().cuda_stream);
 * NULL is the device's default (null) stream — which is what torch uses unless told otherwise.
 * Every launch, and the hipEvents of ot_timing_*, go to this stream. */
int ot_ctx_create(int device, void* stream, ot_ctx** out);
int ot_ctx_destroy(ot_ctx* ctx);
int ot_ctx_synchronize(ot_ctx* ctx);
/* Re-target the ctx at another stream of the same device (synchronises the old one first). */
int ot_ctx_set_stream(ot_ctx* ctx, void* stream);

/* Copies the scene (converted to the f32 layout as well) to the device. */
int ot_scene_upload(ot_ctx* ctx, const ot_scene_desc* scene);

/* ---- the hot path ----------------------------------------------------------------------- */

/* Non-branching trace (scene.max_children <= 1): one lane per ray, all segments of a ray in
 * one launch.  Segment k of ray i is written at slot k*n_rays + i of every ot_segments array
 * (capacity max_segments*n_rays); seg_count[i] = number of slots ray i used.  max_segments is
 * the reference's perfomance_limit["max_trace_num"] (optical_table.py:87-97).
 * Scenes in which a hit CAN emit two rays (max_children == 2) may still be traced here
 * speculatively: a ray whose tree actually branches at its k-th segment gets
 * seg_count[i] = -(k+1) (its first k+1 slots are valid, the tree is incomplete) and must be
 * re-traced with ot_trace_generation_f64.
 * counts: int32 [n_count_slots][n_count_classes] interact-count table indexed by rays.id, or
 * NULL when the scene has no limited surface.  A ray whose id lies outside [0, n_count_classes)
 * is not counted (every limited surface stays open to it); nothing is indexed out of range.
 * Precondition for the reference's order (optical_table.py:66-70 finishes one input ray before the
 * next): at most ONE ray per id in a launch — callers with rays that share an id trace them in
 * successive launches over the same table (round r = the r-th ray of every id), as table.py does.
 * If the precondition is broken nothing is corrupted: the gate is an atomic increment-below-cap,
 * a counter never passes its cap; only WHICH of the sharing rays get the remaining counts is then
 * unspecified (ot_trace_generation_*: the trees of one generation all see the table as it stood
 * before the generation, and the table is closed at the cap afterwards). */
int ot_trace_f64(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_segments,
                 const ot_segments* out, int32_t* seg_count, int32_t* counts,
                 int32_t n_count_classes);
int ot_trace_f32(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_segments,
                 const ot_segments* out, int32_t* seg_count, int32_t* counts,
                 int32_t n_count_classes);

/* The same trace with the TILED layout: the max_segments * n_rays slots of ot_trace_* in tiles of 64 consecutive slots.
 * Slot s = k * n_rays + i lives in tile s / 64 at lane s % 64; a tile is OT_TILE_BYTES(real) = 64 * (12 * sizeof(real) + 8)
 * bytes: the 12 real fields of ot_segments, 64 values each, then int32 ray[64], int32 surface[64].  A wave of the kernel
 * then writes ONE contiguous 6656-byte (fp32: 3584) block per segment instead of 14 runs in 14 arrays that lie
 * max_segments * n_rays elements apart: the HBM streams of a light scene run 9 % faster (tools/stream_layouts.hip:
 * 5.68 instead of 5.22 TB/s for cfg 2's 1 record in, 5 out).  `tiles` is a device pointer (16-byte aligned) to
 * capacity / 64 tiles; capacity is a multiple of 64 and >= max_segments * n_rays rounded up to 64.  seg_count, counts:
 * as for ot_trace_*.  Light scenes only (the lane-per-ray kernel): a scene that ot_trace_* would send to the rolling
 * lists (24 nodes or more) gets OT_ERR_UNSUPPORTED here — its dense output is ot_trace_append_*. */
#define OT_TILE_BYTES(real_bytes) (64 * (12 * (real_bytes) + 8))
int ot_trace_tiled_f64(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_segments, void* tiles,
                       int64_t capacity, int32_t* seg_count, int32_t* counts, int32_t n_count_classes);
int ot_trace_tiled_f32(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_segments, void* tiles,
                       int64_t capacity, int32_t* seg_count, int32_t* counts, int32_t n_count_classes);

/* The same trace with the APPEND layout: a dense list of segment records instead of max_segments * n_rays slots.
 * The reference returns a list with one entry per processed segment (optical_table.py:125-134); the [k][ray] slots of
 * ot_trace_* hold that list with a hole for every segment a ray did not live to (cfg 3: 11 GB of slots for 2.8 GB of
 * records; cfg 5: 35 GB for 17 GB), and their late planes are written a few elements per cache line.  Here the output is
 * ONE allocation of 14 planes of `capacity` slots each — the 12 real fields in the order of ot_segments, then int32
 * ray[capacity], int32 surface[capacity] — filled in append order: every wave of the kernel claims chunks of slots from
 * a device-wide cursor and writes 64 consecutive records per pass.
 *   - *n_slots (device int64) receives the number of slots claimed.  Slots [0, *n_slots) hold records, except for HOLES,
 *     marked ray = -1: the unused tail of each wave's last chunk (at most OT_OPT_APPEND_CHUNK - 1 slots per wave); with
 *     the block pool (OT_OPT_BLOCK_POOL: curved scenes in single precision) chunks of 16 * OT_OPT_APPEND_CHUNK slots belong
 *     to a workgroup: up to 63 holes at the end of every chunk and the tail of each workgroup's last one.
 *   - order: the records of one ray lie at increasing slot numbers in segment order, so a STABLE sort by `ray` yields the
 *     reference's order (input ray major, then segment order) — the contract of ot_trace_generation_*'s flat list.  The
 *     order of rays among each other is not deterministic.  seg_count[i] is written as by ot_trace_*.
 *   - capacity (a multiple of 64, below 2^30 slots in single and 2^29 in double precision; base 16-byte aligned): if *n_slots > capacity the records that did not fit are lost
 *     (nothing is written outside the block); *n_slots is still exact, so the caller can retry with enough room.
 *     (*n_slots >= 2^62: a wave of the block-pool kernel stopped at one of its internal bounds — a defect, never a capacity matter.)
 *     sum(seg_count) + chunk * (number of waves launched) always suffices (block pool: sum(seg_count) * (1 + 1/128) +
 *     (16 * chunk + 64) * (number of workgroups launched)); ot_debug_last_launch names the launch shape, and
 *     max_segments * n_rays plus that slack never fails.
 * Works for every scene ot_trace_* accepts, always on the heavy-scene kernels (rolling lists or block pool). */
typedef struct ot_segment_block {
    void* base;        /* device pointer: 12 planes of `real`[capacity], then int32 ray[capacity], int32 surface[capacity] */
    int64_t capacity;  /* slots per plane */
} ot_segment_block;
int ot_trace_append_f64(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_segments,
                        const ot_segment_block* out, int64_t* n_slots, int32_t* seg_count, int32_t* counts,
                        int32_t n_count_classes);
int ot_trace_append_f32(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_segments,
                        const ot_segment_block* out, int64_t* n_slots, int32_t* seg_count, int32_t* counts,
                        int32_t n_count_classes);

/* Branching trace, one generation of the breadth-first ray tree per call
 * (optical_table.py:115-134).  Input: the generation's alive rays with their tree index
 * (rays_tree) in BFS order.  budget[i] = how many more segments tree i may still process
 * (max_trace_num minus what it already used); rays beyond the budget are dropped as the
 * reference drops them (optical_table.py:138-144).  Output: one segment per processed ray
 * appended at *seg_cursor, and the next generation's rays (stable order: parent order, then
 * child order) in next/next_tree with *n_next.  All counters are device int64/int32 scalars.  A tree that uses up its
 * budget in this call emits no children (OT_OPT_GEN_DROP_DOOMED): the next call would drop them all. */
int ot_trace_generation_f64(ot_ctx* ctx, const ot_rays* rays, const int32_t* rays_tree,
                            int64_t n_rays, int32_t* budget, const ot_segments* out,
                            int64_t out_capacity, int64_t* seg_cursor, const ot_rays* next,
                            int32_t* next_tree, int64_t next_capacity, int64_t* n_next,
                            int32_t* counts, int32_t n_count_classes);
int ot_trace_generation_f32(ot_ctx* ctx, const ot_rays* rays, const int32_t* rays_tree,
                            int64_t n_rays, int32_t* budget, const ot_segments* out,
                            int64_t out_capacity, int64_t* seg_cursor, const ot_rays* next,
                            int32_t* next_tree, int64_t next_capacity, int64_t* n_next,
                            int32_t* counts, int32_t n_count_classes);

/* Whole ray trees in ONE launch, a lane per tree (k_trace_trees): the reference's loop — pop the oldest ray, archive it,
 * push its children, stop after max_trace_num rays (optical_table.py:115-147) — runs per lane with the queue on the chip (its
 * front in LDS, the rest in an L2-resident scratch of the library), and only segment records go to memory.  Output as
 * ot_trace_*: slot k * n_rays + i of `out` (max_trace_num * n_rays slots) is the k-th ray of tree i in the reference's FIFO
 * order, seg_count[i] the rays tree i processed (== max_trace_num: cut short by the cap, or ended exactly there).  A queue of
 * ceil(max_trace_num / 2) rays per lane always suffices; ot_trace_trees_plan says whether the scene has such a kernel
 * (info[0] bit 0: every scene — one whose image and queue fronts fit the CU's LDS through the kernel of its preset, a larger one (thousands
 * of leaves) through the all-features kernel that keeps the node records in LDS, or nothing but the queues, and reads the tables from
 * global memory: dense list only; bit 1: it also writes these [k][tree] slots — scenes
 * of the planar preset; every such kernel writes the dense list of ot_trace_trees_append_*), how many entries its queues get
 * (info[1]), whether that is enough for every tree (info[2]: always, up to caps of ~170 in double precision for a batch that
 * fills the device and of 510 for a few hundred trees — the scratch is per workgroup of the launch) how many of them are in
 * LDS (info[3]), and for ot_trace_trees_append_* the slots a wave claims at a time (info[4]) and the waves of the launch (info[5]):
 * a block of n_rays * max_trace_num + info[4] * info[5] slots holds any trace of the batch.  If not, a tree whose queue overflows reports seg_count[i] = -(rays processed so far) and the
 * caller takes ot_trace_tree_*.  OT_ERR_UNSUPPORTED when info[0] would be 0.  counts / n_count_classes as for ot_trace_*:
 * a tree meets count-limited surfaces in the reference's FIFO order; trees that share a column (rays.id) in one launch get
 * the remaining counts in unspecified order — the caller's rounds, as for ot_trace_*. */
int ot_trace_trees_f64(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_trace_num, const ot_segments* out, int32_t* seg_count,
                       int32_t* counts, int32_t n_count_classes);
int ot_trace_trees_f32(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_trace_num, const ot_segments* out, int32_t* seg_count,
                       int32_t* counts, int32_t n_count_classes);
/* ... into the append layout of ot_trace_append_* (a dense list; *n_slots = slots claimed, records and holes): the records of
 * a step go to consecutive slots whatever trees the lanes are on, so batches whose trees differ widely in size (lanes refill
 * from their wave's share as their trees end) still write whole lines.  A tree's records lie at increasing addresses in FIFO
 * order: a stable sort by `ray` is the reference's order.  seg_count / counts as above. */
int ot_trace_trees_append_f64(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_trace_num, const ot_segment_block* out, int64_t* n_slots,
                              int32_t* seg_count, int32_t* counts, int32_t n_count_classes);
int ot_trace_trees_append_f32(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_trace_num, const ot_segment_block* out, int64_t* n_slots,
                              int32_t* seg_count, int32_t* counts, int32_t n_count_classes);
int ot_trace_trees_plan(ot_ctx* ctx, int32_t real_bytes, int32_t max_trace_num, int64_t n_rays /* 0: a batch that fills the device */,
                        int32_t* info /* int32[8] */);

/* The whole breadth-first trace of a batch of ray trees: the loop over ot_trace_generation_* (optical_table.py:115-147) run
 * by the library — per generation one launch sequence and ONE 16-byte read-back.  `state` is device int64[2] = {segment
 * cursor, rays of the pending generation}: zero it before the first call.  buf_a / buf_b (+ tree_a / tree_b) are two
 * generation buffers of buf_capacity rays each.  result (host int64[5]): [0] segments written so far, [1] rays of the
 * pending generation (0: finished), [2] where they are (0: the caller's `rays`, 1: buf_a, 2: buf_b), [3] generations run by
 * this call, [4] why it stopped: 0 queue empty, 1 the segment arrays cannot take the pending generation, 2 the buffers
 * cannot take its children (max_children x rays), 3 max_seconds (>= 0) ran out.  After 1 or 2 the caller provides more
 * room and calls again with the pending generation as `rays` / `rays_tree` (and the same state and budget). */
int ot_trace_tree_f64(ot_ctx* ctx, const ot_rays* rays, const int32_t* rays_tree, int64_t n_rays, int32_t* budget,
                      const ot_segments* out, int64_t out_capacity, int64_t* state, const ot_rays* buf_a, int32_t* tree_a,
                      const ot_rays* buf_b, int32_t* tree_b, int64_t buf_capacity, int32_t* counts, int32_t n_count_classes,
                      double max_seconds, int64_t* result);
int ot_trace_tree_f32(ot_ctx* ctx, const ot_rays* rays, const int32_t* rays_tree, int64_t n_rays, int32_t* budget,
                      const ot_segments* out, int64_t out_capacity, int64_t* state, const ot_rays* buf_a, int32_t* tree_a,
                      const ot_rays* buf_b, int32_t* tree_b, int64_t buf_capacity, int32_t* counts, int32_t n_count_classes,
                      double max_seconds, int64_t* result);

/* What the library would launch for the uploaded scene and a batch of n_rays rays of `real_bytes` (4 / 8) precision, and
 * which output layout its kernels write fastest — what a caller with no preference should ask for:
 * info[0] kernel family of ot_trace_* (1 lane per ray, 2 rolling lists / block pool); [1] 1 = ot_trace_tiled_* accepts the
 * scene; [2] log2 of the first capacity ot_trace_append_* refuses (30 / 29); [3] the recommended layout: 0 slot arrays
 * (ot_trace_*), 1 tiles (ot_trace_tiled_*), 2 append (ot_trace_append_*); [5], [6] light scenes: hundredths of a microsecond
 * per launch of the stream companion in slot arrays / tiles (ot_probe_layouts).  Heavy scenes: append, unless the worst case
 * max_segments * n_rays would not fit the capacity limit (then slots, which have none).  Light scenes: the layout this DEVICE
 * streams faster — the 64-slot tiles run up to 12 % faster than the 14 arrays on some boxes and 8 % slower on others, so
 * it is measured (once per context and precision, ~15 ms, 1.3 GB of temporary buffers in double precision). */
int ot_trace_plan(ot_ctx* ctx, int32_t real_bytes, int64_t n_rays, int32_t max_segments, int32_t info[8]);
/* The measurement behind it: microseconds per launch of cfg 2's streams (2^20 rays, 5 segments, no tracing) into the 14 slot
 * arrays and into 64-slot tiles; cached in the context. */
int ot_probe_layouts(ot_ctx* ctx, int32_t real_bytes, double* us_slots, double* us_tiled);

/* Diagnostic: how often the two passes of a generation (count, then emit: both run the same trace) disagreed about a
 * ray since the ctx was created.  Expected 0; a disagreement is contained (nothing is written outside the slots the
 * count pass reserved) and shows up as a zero-intensity dead ray in the next generation. */
int ot_debug_generation_mismatches(ot_ctx* ctx, int64_t* count);

/* Diagnostic: the shape of the last ot_trace_* launch on this ctx, for profiles and tuning notes.
 * info[0] kernel (1 lane per ray, 2 rolling lists, 3 refill, 4 lane per tree: [5] / [6] = its queue entries in LDS / in the
 * scratch ring), [1] threads per workgroup, [2] workgroups per CU the occupancy
 * query allowed, [3] workgroups launched, [4] dynamic LDS bytes per workgroup, [5] list capacity per wave (rolling),
 * [6] 1 = mixed generations, [7] bit 0 = candidate pair queue (OT_OPT_FLAT_QUEUE took effect), bit 1 = records in LDS,
 * bit 2 = append layout, bit 3 = tiled layout, bit 4 = workgroup-wide block pool (OT_OPT_BLOCK_POOL; [5] = its slots). */
int ot_debug_last_launch(ot_ctx* ctx, int32_t info[8]);

/* Monitor.record (monitor.py:183-193): intersect finished segments with a rectangular
 * monitor plane, honouring segment length.  hit_index receives the slot indices of the segments
 * that hit (ascending), P the local hit point, t the distance; *n_hits the count.
 * seg_count/n_rays: pass ot_trace_*'s seg_count and n_rays to scan a [k][ray] slot array
 * (n_segments = max_segments*n_rays; unused slots are skipped); NULL/0 for a flat list (ot_trace_generation_*);
 * NULL/-1 for a list with holes (ot_trace_append_*: slots whose `ray` is negative are skipped). */
typedef struct ot_monitor {
    double M[9];
    double origin[3];
    double half_width, half_height;
} ot_monitor;
int ot_monitor_record_f64(ot_ctx* ctx, const ot_monitor* mon, const ot_segments* segs,
                          int64_t n_segments, const int32_t* seg_count, int64_t n_rays,
                          int64_t* hit_index, void* Px, void* Py, void* Pz, void* t, int64_t* n_hits);

/* ---- measurement ------------------------------------------------------------------------ */
/* When enabled every trace launch is bracketed by hipEvents on the ctx stream. */
int ot_timing_enable(ot_ctx* ctx, int enabled);
/* Sum and count of kernel durations since the last reset (synchronises the stream). */
int ot_timing_read(ot_ctx* ctx, double* total_ms, int64_t* launches);
int ot_timing_reset(ot_ctx* ctx);

/* Tuning knobs (tools/tune.py, tests); the defaults are the measured best on MI355X. */
enum ot_option {
    OT_OPT_NT_STORES = 1,      /* segment records written with non-temporal stores (0/1)        */
    OT_OPT_MIN_WAVES = 2,      /* 0: compiler's choice; 4: cap registers for 4 waves per SIMD  */
    OT_OPT_BLOCKS_PER_CU = 3,  /* persistent-grid size in 256-thread blocks per CU (0 = auto)  */
    OT_OPT_KERNEL = 4,         /* 0 auto; 1 lane-per-ray kernel; 2 rolling-list kernel (heavy scenes) */
    OT_OPT_LDS_LIMIT_KB = 5,   /* scene images above this many KB are read from global memory   */
    OT_OPT_LIST_CAP = 6,       /* heavy scenes: live rays per wave in the rolling list: a multiple of 64, 128..1024 (mixed lists use it only if it is a power of two) */
    OT_OPT_PAIR_STORES = 7,    /* lane-per-ray kernel: lane pairs write two fields per 16-byte store (0/1) */
    OT_OPT_MIX_GENERATIONS = 8,/* heavy scenes: -1 auto, 0 generation-pure lists even under a top-level grid */
    OT_OPT_FLAT_QUEUE = 9,     /* planar scenes under a top-level grid: wave-wide candidate queue (0 / 1; a multiple of 64 in 192..8192:
                                  on, with room for that many (ray, leaf) pairs per round instead of 512 — lanes that do not fit wait a round) */
    OT_OPT_LDS_RECORDS = 10,   /* heavy scenes, fp32: records of the live rays in LDS (the first 128 positions of every wave's list; the rest
                                  of a generation-pure list of OT_OPT_LIST_CAP entries spills to global scratch): -1 auto, 0 never, 1 whenever it fits */
    OT_OPT_APPEND_CHUNK = 11,  /* ot_trace_append_*: slots a wave claims per atomic (multiple of 64, default 512) */
    OT_OPT_INSTANCING = 12,    /* fold identical lattice children (MMA / MLA / DMD) into one record + per-member pose at upload (0/1) */
    OT_OPT_GEN_REUSE = 13,     /* ot_trace_generation_*: the emit pass rebuilds the hit the count pass found (node + distance kept per
                                  ray) instead of searching the scene again: -1 auto (scenes of 12 nodes or more), 0 never, 1 always */
    OT_OPT_BLOCK_POOL = 14,    /* heavy scenes with curved surfaces, fp32: the live rays of a workgroup in one pool of 64-ray blocks in
                                  LDS, shared by its sixteen waves (k_trace_pool), instead of a list per wave: -1 auto, 0 never, 1 whenever it fits */
    OT_OPT_REFILL = 16,        /* heavy scenes under a top-level grid (mixed generations): the live rays in registers, every lane takes its
                                  next fresh ray in place (k_trace_refill) instead of a list per wave: 0 (default: the lists; the two tie on cfg 3), 1 whenever a kernel exists */
    OT_OPT_REFILL_TICKET = 17, /* ... rays a wave draws from the device-wide queue per atomic: 0 = by batch size (64..256), or a multiple of 64 */
    OT_OPT_GEN_ONEPASS = 19,   /* ot_trace_tree_*: every generation in ONE pass — each workgroup traces a tile of rays once, keeps the children in
                                  registers and takes its output offsets from a decoupled look-back over per-tile descriptors (k_gen_one) —
                                  instead of count + scan + emit: -1 (default) for generations of up to 65536 rays (one launch instead of six: what
                                  small ray trees cost), 0 never, 1 always.  Large generations are faster in two passes (every tile of the one-pass
                                  kernel waits for the slowest of its predecessors).  Scenes with count-limited surfaces always take the two
                                  passes.  Identical output either way. */
    OT_OPT_GEN_AHEAD = 20,     /* ot_trace_tree_*, light scenes without count-limited surfaces: the emit pass of a generation also counts the
                                  children of the children it writes, and the next generation replaces its count pass over the ray records by a pass
                                  over one byte per ray (k_gen_recount): 1 (default) / 0.  Identical output either way. */
    OT_OPT_TREES_GLOBAL_IMAGE = 25, /* ot_trace_trees_*: scenes whose image no LDS holds (thousands of leaves) are traced by the all-features tree kernel
                                  that reads the image from global memory; 1 = every scene takes that kernel (test knob: results must not change), 0 default */
    OT_OPT_GEN_PARENT_INDEX = 24, /* ot_trace_generation_*: 1 = next_tree[] receives, for every emitted ray, the INDEX of its parent in this call's
                                  `rays` instead of the parent's tree id (0, default).  For a host that keeps an object per ray and needs to know whose
                                  child a ray is while `tree` groups the rays for the FIFO-exact count gates: the object API's trace through scenes
                                  with user-defined components (optical_component.py:235-240; table.py _trace_hooked).  ot_trace_tree_* ignores it. */
    OT_OPT_TREES_FLAT = 23,    /* ot_trace_trees_*: planar scenes under a top-level grid of leaves search through the wave-wide pair queue of the
                                  heavy non-branching kernel (OT_OPT_FLAT_QUEUE) instead of a grid walk per lane: 1 (default) / 0.  Identical records. */
    OT_OPT_TREES_REFILL_AT = 22, /* ot_trace_trees_*: idle lanes of a wave at which they take the next trees of the wave's share (default 16: lanes
                                  kept busy; 64: a wave takes 64 trees at a time and its lanes stay at the same depth of their trees — faster
                                  when trees differ moderately in size, slower when most are tiny).  Identical records either way. */
    OT_OPT_TREES_LDS_ENTRIES = 21, /* ot_trace_trees_*: queue entries per lane kept in LDS (the rest of a tree's queue lives in a global scratch):
                                  more entries, fewer waves per CU.  0 (default): two under caps of up to 16, three above */
    OT_OPT_POOL_JITTER = 18,   /* test knob of the block pool's cross-wave protocol: one in `value` publications of a state or control word is
                                  held back ~8000 cycles after the records it announces were written (0 = off).  Results must not change. */
    OT_OPT_GEN_DROP_DOOMED = 15 /* ot_trace_generation_*: a tree whose budget ends with this generation gets no children in `next` (they
                                  could never be processed: optical_table.py:138-144) — 1 (default) / 0: emit them, for a caller who
                                  wants to go on with a larger budget */
};
int ot_set_option(ot_ctx* ctx, int32_t option, int32_t value);

/* Roofline companion of ot_trace_f64: the same streams (one ray record in, max_segments segment
 * records out per ray) with no tracing in between — the ceiling of this access pattern. */
int ot_bench_stream_f64(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_segments,
                        const ot_segments* out, int32_t* seg_count);
int ot_bench_stream_f32(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_segments,
                        const ot_segments* out, int32_t* seg_count);
/* ... and of ot_trace_tiled_*: the same records into 64-slot tiles. */
int ot_bench_stream_tiled_f64(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_segments, void* tiles,
                              int64_t capacity, int32_t* seg_count);
int ot_bench_stream_tiled_f32(ot_ctx* ctx, const ot_rays* rays, int64_t n_rays, int32_t max_segments, void* tiles,
                              int64_t capacity, int32_t* seg_count);

#ifdef __cplusplus
}
#endif
#endif /* OPTABLE_HIP_H */
