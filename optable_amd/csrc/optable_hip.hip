// optable_hip.hip — the C-ABI of liboptable_hip.so (include/optable_hip.h): context, scene validation and upload,
// kernel selection and launch.  The kernels are in kernels.h, the device math in trace_core.h.
// No CPU fallback lives here: without a GPU every entry point fails with OT_ERR_HIP.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

#include "kernels.h"
#include "misc_kernels.h"
#include "tables.h"

// ------------------------------------------------------------------------------------------
// error plumbing
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(OT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)

// ------------------------------------------------------------------------------------------
// context
struct Scratch {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        if (hipMalloc(&p, need) != hipSuccess) return -1;
        bytes = need;
        return 0;
    }
};

struct ot_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cus = 256;
    size_t lds_limit = 64 * 1024;
    // scene
    bool has_scene = false;
    void *blob64 = nullptr, *blob32 = nullptr;
    size_t bytes64 = 0, bytes32 = 0;
    size_t head64 = 0, head32 = 0;  // node + material records at the front of the image (k_trace_trees IMG = 2 keeps these in LDS)
    int32_t n_nodes = 0, n_mats = 0, n_aux = 0, n_slots = 0, max_children = 0;
    int32_t n_phys = 0, n_runs = 0, runs_word64 = 0, runs_word32 = 0;  // instanced runs folded by fill_blob (trace_core.h NodeRef)
    int32_t opt_append_chunk = 512;  // append layout: slots per claim
    int32_t opt_instancing = 1;      // fold lattice children into instanced runs at upload
    bool opt_gen_parent = false;     // ot_trace_generation_*: next_tree[] = index of the parent ray (OT_OPT_GEN_PARENT_INDEX)
    int32_t opt_gen_drop = 1;        // generation kernels: children of a tree whose budget ends with this generation are not emitted
    int32_t opt_gen_reuse = -1;      // generation kernels: emit pass rebuilds the count pass's hit instead of searching again (-1 auto)
    double unit = 1e-2;
    uint32_t features = 0;
    int32_t root_max_items = 0;  // most items in one cell of the top-level grid
    int64_t root_n_items = 0;    // entries of all its cells together
    int32_t root_pack = -1;      // aux offset of the packed cells (fill_blob), -1 when not built
    int32_t root_grid = -1;  // aux offset of the top-level grid
    int32_t cache_mat = -1;  // first Sellmeier material
    int32_t* slot_max = nullptr;  // device [n_slots]: max_interact_count per count slot
    // timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    double total_ms = 0.0;
    int64_t launches = 0;
    // knobs
    // scene images above this stay in global memory (L2): a 100+ KB LDS image leaves one block per CU,
    // and on cfg 5 the lost occupancy cost 1.5x (tools/bench_configs.py)
    int32_t opt_lds_limit_kb = 64;
    int32_t opt_kernel = 0;  // 0 auto, 1 fused (lane per ray), 2 rolling lists (the heavy-scene kernel)
    int32_t last_launch[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ot_debug_last_launch
    // heavy-scene launch plan per precision (a dozen occupancy queries): recomputed after an upload or an option change
    struct RollingPlan { uint64_t epoch = 0; int wpb = 4, per_cu = 1; int32_t cap = 128, capl = 0, flat_cap = 0; bool lds = false, rec_lds = false; size_t lds_bytes = 0; };
    RollingPlan plan[2][2];  // [precision][output layout]
    uint64_t plan_epoch = 1;
    int32_t opt_refill = 0;      // mixed scenes: rays in registers, refilled in place (k_trace_refill): 0 never (the lists; default: the two tie on cfg 3
                                 // in the append layout, 2.08-2.11 ms, and the lists win into [k][ray] slots, 3.7 against 4.7 ms), 1 whenever a kernel exists
    int32_t opt_refill_ticket = 0;  // rays per ticket of k_trace_refill (0 = by batch size)
    int32_t opt_pool_jitter = 0; // k_trace_pool: one in this many state-word publications is held back ~8000 cycles (protocol test; 0 = off)
    int32_t opt_pool = -1;       // curved-surface scenes, fp32: workgroup-wide block pool (-1 auto, 0 never, 1 whenever it fits)
    int32_t opt_rec_lds = -1;    // pair-queue scenes: records of the live rays in LDS (-1 auto, 0 never, 1 whenever it fits)
    int32_t opt_list_cap = 128;  // k_trace_rolling: live rays per wave (cfg 3: 128 beats 256 and 512)
    int32_t opt_flat = 1;  // fp32 planar top-level-grid scenes: wave-wide pair queue (flat_grid_hit)
    int32_t opt_mix = -1;  // -1 auto (scenes under a top-level grid mix generations), 0 never
    int32_t opt_list_cap_pure = 0;  // generation-pure lists: 0 = the chunk rule below
    Scratch blocked;
    size_t blocked_queue_off = 0;
    int32_t opt_pair = 1;  // paired 16-byte segment stores in the lane-per-ray kernel
    int32_t opt_nt = 1, opt_minw = 4, opt_blocks_per_cu = 0;  // defaults from tools/tune.py on MI355X
    Scratch gen, scan_tmp, mon, gen_rem, gen_ahead, trees;
    int32_t opt_trees_global = 0;      // k_trace_trees: 1 = every scene through the all-features kernel that reads its image from global memory (test knob)
    int32_t opt_trees_flat = 1;        // k_trace_trees: planar scenes under a top-level grid of leaves search through the wave-wide pair queue
    int32_t opt_trees_refill_at = 16;  // k_trace_trees: idle lanes of a wave at which they take their next trees (64: a wave takes 64 trees at a time)
    int32_t opt_trees_lds = 0;  // k_trace_trees: queue entries per lane kept in LDS (the rest of a tree's queue lives in a global scratch); 0: by the cap
    int32_t opt_gen_ahead = 1;  // ot_trace_tree_*: the emit pass counts its children's children, the next generation skips its count pass (k_gen_pass MODE 2)
    int32_t opt_gen_onepass = -1;  // ot_trace_tree_*: one pass per generation with a decoupled look-back (k_gen_one): -1 (default) generations of up to
                                   // 65536 rays (one launch instead of six), 0 never, 1 always.  Large generations keep count + scan + emit: the
                                   // one-pass kernel moves 22 % fewer bytes but every tile waits for the slowest of its predecessors (cfg 4 with
                                   // R = 0.2: 18.4 ms against 13.6, tools/ab_onepass.py).  Scenes with count-limited surfaces: always two passes
    double probe_us[2][2] = {{0, 0}, {0, 0}};  // ot_probe_layouts: [precision][0 slot arrays, 1 tiles] microseconds per launch of the stream companion; 0 = not measured
    int64_t* gen_chain = nullptr;                // k_gen_one in chained launches: the generation sizes, on the device
    unsigned long long* gen_mismatch = nullptr;  // count / emit disagreements of k_gen_pass (expected: 0)
    int64_t* pinned_state = nullptr;             // ot_trace_tree_*: page-locked landing place of the per-generation read-back
};

static int flush_events(ot_ctx* c) {
    if (c->events_used == 0) return 0;
    HIP_TRY(hipEventSynchronize(c->events[c->events_used - 1].second));
    for (size_t k = 0; k < c->events_used; ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->events[k].first, c->events[k].second));
        c->total_ms += ms;
        c->launches += 1;
    }
    c->events_used = 0;
    return 0;
}
static int timing_begin(ot_ctx* c) {
    if (!c->timing) return 0;
    if (c->events_used == c->events.size()) {
        if (c->events.size() >= 1024) {
            int rc = flush_events(c);
            if (rc) return rc;
        } else {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            c->events.push_back({a, b});
        }
    }
    HIP_TRY(hipEventRecord(c->events[c->events_used].first, c->stream));
    return 0;
}
// Single-kernel launches attach the event pair to the dispatch itself (hipExtLaunchKernelGGL): the
// timestamps are the kernel's own begin/end and no extra marker packets sit between launches.
static int timing_pair(ot_ctx* c, hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (!c->timing) return 0;
    if (c->events_used == c->events.size()) {
        if (c->events.size() >= 1024) {
            int rc = flush_events(c);
            if (rc) return rc;
        } else {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            c->events.push_back({a, b});
        }
    }
    *start = c->events[c->events_used].first;
    *stop = c->events[c->events_used].second;
    c->events_used += 1;
    return 0;
}

static int timing_end(ot_ctx* c) {
    if (!c->timing) return 0;
    HIP_TRY(hipEventRecord(c->events[c->events_used].second, c->stream));
    c->events_used += 1;
    return 0;
}

// host -> device node conversion
// Cells of the top-level grid in one word each (first item | count << 11) behind the caller's aux array, for the pair-queue
// walk (trace_core.h flat_grid_hit): possible when the item list has at most 1023 entries and no cell more than 42.
// Returns the number of cells to append, 0 when the grid is absent or too large for the packing.
static int64_t packable_cells(const ot_scene_desc* s) {
    if (s->root_grid < 0) return 0;
    const double* g = s->aux + s->root_grid;
    const int64_t cells = (int64_t)g[2] * (int64_t)g[3];
    if ((int64_t)g[11 + cells] > 1023) return 0;  // a queue marker is (lane << 10 | index into the item list) + 1, in 16 bits
    for (int64_t k = 0; k < cells; ++k)
        if (g[11 + k + 1] - g[11 + k] > 42) return 0;
    return cells;
}
// Aspheres whose implicit function g = x + F(r) is convex (or concave) along every line through the local box — F convex and
// non-decreasing on [0, box diagonal] (or -F) — get a device flag: a ray that starts on such a surface can skip the root
// scan in the two cases trace_core.h hit_leaf spells out.  Checked numerically on 512 samples of the reference's own F
// (component_group.py:1061-1064, 1092-1097); anything not clearly convex gets no flag and the full scan.
static double host_sag(const ot_node& h, double r) {
    const double r2 = r * r;
    if (h.shape == OT_SHAPE_ASPHERE_PARAM) {
        const double R = h.p[1], k = h.p[2], r4 = r2 * r2;
        return r2 / (R * (1.0 + std::sqrt(1.0 - (1.0 + k) * r2 / (R * R)))) + h.p[3] * r4 + h.p[4] * r4 * r2 + h.p[5] * r4 * r4;
    }
    const double EFL = h.p[1], n = h.p[2];
    return (EFL / (n + 1.0)) * (-1.0 + std::sqrt(1.0 + (n + 1.0) / (n - 1.0) * r2 / (EFL * EFL)));
}
static int32_t asphere_convexity(const ot_node& h) {
    if (h.kind != OT_NODE_LEAF || (h.shape != OT_SHAPE_ASPHERE_PARAM && h.shape != OT_SHAPE_ASPHERE_EXACT)) return 0;
    constexpr int N = 512;
    const double rmax = 1.4143 * h.p[0] * 1.001, dr = rmax / N;
    if (!(rmax > 0)) return 0;
    double f[N + 1];
    for (int j = 0; j <= N; ++j) {
        f[j] = host_sag(h, j * dr);
        if (!std::isfinite(f[j])) return 0;
    }
    double scale = 0.0;
    for (int j = 0; j <= N; ++j) scale = std::fmax(scale, std::fabs(f[j]));
    const double tol = 1e-13 * (scale + 1e-300);
    bool pos = true, neg = true;  // F (resp. -F) non-decreasing and convex
    for (int j = 1; j <= N; ++j) {
        const double d1 = f[j] - f[j - 1];
        if (d1 < -tol) pos = false;
        if (d1 > tol) neg = false;
        if (j < N) {
            const double d2 = f[j + 1] - 2.0 * f[j] + f[j - 1];
            if (d2 < -tol) pos = false;
            if (d2 > tol) neg = false;
        }
    }
    return pos ? DN_CONVEX_POS : (neg ? DN_CONVEX_NEG : 0);
}

// Instanced runs: consecutive leaf children of a gridded group that differ only in origin, lab AABB and leaf id — the caps
// of an MMA, the lenslets of an MLA, the mirrors of a DMD (component_group.py:228-304, 367).  The device image keeps one
// record per run plus 9 reals per member (trace_core.h NodeRef); everything is compared bit for bit, so a lattice with
// per-element drift (roc_drift, focal_drift) simply has no runs.  Children of gridded groups are only ever reached
// through the group's grid, never by the linear walk, which is what makes folding them safe.
struct NodeRun { int32_t first, count, pnode, geo; };
static bool same_but_pose(const ot_node& a, const ot_node& b) {
    return !memcmp(a.M, b.M, sizeof a.M) && !memcmp(a.lbox, b.lbox, sizeof a.lbox) && !memcmp(a.p, b.p, sizeof a.p) &&
           !memcmp(&a.reflectivity, &b.reflectivity, 4 * sizeof(double)) && a.kind == b.kind && a.flags == b.flags && a.shape == b.shape &&
           a.interaction == b.interaction && a.mat1 == b.mat1 && a.mat2 == b.mat2 && a.roc_kind == b.roc_kind &&
           a.max_interact_count < 0 && b.max_interact_count < 0 && a.aux < 0 && b.aux < 0 && b.leaf_id == a.leaf_id + 1;
}
static std::vector<NodeRun> find_runs(const ot_scene_desc* s) {
    std::vector<NodeRun> runs;
    constexpr int MIN_RUN = 8, MAX_RUNS = 8;
    for (int g = 0; g < s->n_nodes; ++g) {
        const ot_node& grp = s->nodes[g];
        if (grp.kind != OT_NODE_GROUP || !(grp.flags & OT_NODE_GRID)) continue;
        for (int j = g + 1; j < grp.end;) {  // (validate_scene: every child of a gridded group is a leaf)
            int e = j + 1;
            while (e < grp.end && s->nodes[e].kind == OT_NODE_LEAF && s->nodes[e - 1].kind == OT_NODE_LEAF && same_but_pose(s->nodes[e - 1], s->nodes[e])) ++e;
            if (e - j >= MIN_RUN && (int)runs.size() < MAX_RUNS) runs.push_back({j, e - j, 0, 0});
            j = e;
        }
    }
    return runs;  // ascending in `first` (groups are visited in node order, children in order)
}

template <class T> static void fill_blob(const ot_scene_desc* s, const std::vector<NodeRun>& runs_in, std::vector<uint8_t>& out, int32_t& n_phys,
                                         int32_t& runs_word) {
    std::vector<NodeRun> runs = runs_in;
    const int64_t pack = packable_cells(s);
    int folded = 0, geo_reals = 0;
    for (const NodeRun& r : runs) { folded += r.count - 1; geo_reals += 9 * r.count; }
    n_phys = s->n_nodes - folded;
    const size_t nb = sizeof(DNode<T>) * n_phys, mb = sizeof(DMat<T>) * s->n_materials, ab = sizeof(T) * (s->n_aux + pack + geo_reals);
    const size_t rb = sizeof(int32_t) * 4 * runs.size();
    out.assign(((nb + mb + ab + rb + 15) / 16) * 16, 0);
    DNode<T>* nodes = reinterpret_cast<DNode<T>*>(out.data());
    size_t next_run = 0;
    int phys = 0;
    for (int i = 0; i < s->n_nodes; ++i) {
        if (next_run < runs.size() && i == runs[next_run].first) runs[next_run].pnode = phys;
        if (next_run < runs.size() && i > runs[next_run].first) {  // a folded member
            if (i == runs[next_run].first + runs[next_run].count - 1) ++next_run;
            continue;
        }
        const ot_node& h = s->nodes[i];
        DNode<T>& d = nodes[phys++];
        for (int k = 0; k < 9; ++k) d.M[k] = (T)h.M[k];
        for (int k = 0; k < 3; ++k) d.org[k] = (T)h.origin[k];
        for (int k = 0; k < 6; ++k) { d.aabb[k] = (T)h.aabb[k]; d.lbox[k] = (T)h.lbox[k]; }
        for (int k = 0; k < 8; ++k) d.p[k] = (T)h.p[k];
        d.refl = (T)h.reflectivity; d.trans = (T)h.transmission; d.focal = (T)h.focal_length; d.roc = (T)h.roc;
        d.inv_focal = (T)(h.focal_length != 0.0 ? 1.0 / h.focal_length : 0.0);
        d.r2 = (T)(h.p[0] * h.p[0]);
        if (h.shape == OT_SHAPE_SPHERE) {  // cap aperture radius squared when the cap is shallower than a hemisphere, else 0
            const double R = h.p[0], ht = h.p[1];
            d.r2 = (T)(ht < R ? R * R - (R - ht) * (R - ht) : 0.0);
        }
        d.rad2 = (T)(h.p[0] * h.p[0]);  // sphere / cylinder radius squared
        if (h.shape == OT_SHAPE_ASPHERE_PARAM) {  // folded constants of the root search (trace_core.h sag_search)
            d.p[6] = (T)((1.0 + h.p[2]) / (h.p[1] * h.p[1]));
            d.p[7] = (T)(1.0 / h.p[1]);
        } else if (h.shape == OT_SHAPE_ASPHERE_EXACT) {
            d.p[6] = (T)((h.p[2] + 1.0) / ((h.p[2] - 1.0) * h.p[1] * h.p[1]));
            d.p[7] = (T)(h.p[1] / (h.p[2] + 1.0));
        }
        d.pad1 = (T)0;
        d.kind = h.kind; d.end = h.end; d.flags = h.flags | asphere_convexity(h); d.shape = h.shape; d.inter = h.interaction;
        d.mat1 = h.mat1; d.mat2 = h.mat2; d.roc_kind = h.roc_kind; d.max_count = h.max_interact_count;
        d.slot = h.count_slot; d.aux = h.aux; d.leaf_id = h.leaf_id;
    }
    DMat<T>* mats = reinterpret_cast<DMat<T>*>(out.data() + nb);
    for (int i = 0; i < s->n_materials; ++i) {
        const ot_material& h = s->materials[i];
        mats[i].n = (T)h.n;
        for (int k = 0; k < 3; ++k) { mats[i].B[k] = (T)h.B[k]; mats[i].C[k] = (T)h.C[k]; }
        mats[i].kind = h.kind;
        mats[i].pad = 0;
    }
    T* aux = reinterpret_cast<T*>(out.data() + nb + mb);
    for (int i = 0; i < s->n_aux; ++i) aux[i] = (T)s->aux[i];
    if (sizeof(T) == 4) {
        // The cell lookup of a gridded group widens the ray's footprint by the grid's margin (trace_core.h grid_children);
        // in single precision the footprint itself carries a few ulp of the coordinates, far above the 1e-7 the host
        // chose for double: widen to 64 ulp of the group's largest coordinate (a fraction of a percent of a cell).
        for (int i = 0; i < s->n_nodes; ++i) {
            const ot_node& g = s->nodes[i];
            if (g.kind != OT_NODE_GROUP || !(g.flags & OT_NODE_GRID)) continue;
            double big = 0.0;
            for (int k = 0; k < 6; ++k) big = std::fmax(big, std::fabs(g.aabb[k]));
            const double m = 64.0 * 1.1920929e-7 * big;
            if ((double)aux[g.aux + 8] < m) aux[g.aux + 8] = (T)m;
        }
    }
    for (int64_t k = 0; k < pack; ++k) {
        const double* start = s->aux + s->root_grid + 11;
        aux[s->n_aux + k] = (T)(start[k] + 2048.0 * (start[k + 1] - start[k]));
    }
    int32_t geo = (int32_t)(s->n_aux + pack);
    for (NodeRun& r : runs) {  // per member: origin[3], lab AABB[6]
        r.geo = geo;
        for (int m = 0; m < r.count; ++m) {
            const ot_node& h = s->nodes[r.first + m];
            for (int k = 0; k < 3; ++k) aux[geo++] = (T)h.origin[k];
            for (int k = 0; k < 6; ++k) aux[geo++] = (T)h.aabb[k];
        }
    }
    runs_word = (int32_t)((nb + mb + ab) / 4);
    int32_t* rt = reinterpret_cast<int32_t*>(out.data() + nb + mb + ab);
    for (size_t k = 0; k < runs.size(); ++k) { rt[4 * k] = runs[k].first; rt[4 * k + 1] = runs[k].count; rt[4 * k + 2] = runs[k].pnode; rt[4 * k + 3] = runs[k].geo; }
}

// which code paths the scene needs (trace_core.h feature mask)
static uint32_t scene_features(const ot_scene_desc* s) {
    uint32_t f = 0;
    for (int i = 0; i < s->n_nodes; ++i) {
        const ot_node& nd = s->nodes[i];
        if (nd.flags & OT_NODE_CHECK_AABB) f |= F_AABB;
        if (nd.flags & OT_NODE_GRID) f |= F_GRID;
        if (nd.kind != OT_NODE_LEAF) continue;
        if (nd.shape == OT_SHAPE_POLYGON2D || nd.shape == OT_SHAPE_CSG) f |= F_POLY;
        if (nd.shape != OT_SHAPE_CIRCLE && nd.shape != OT_SHAPE_RECT && nd.shape != OT_SHAPE_POLYGON2D &&
            nd.shape != OT_SHAPE_CSG)
            f |= F_CURVED;
        if (nd.shape == OT_SHAPE_ASPHERE_CHEB) f |= F_MISC;
        if (nd.shape == OT_SHAPE_POLYGON3D) f |= F_POLY | F_MISC;
        if (nd.shape == OT_SHAPE_CYLINDER) f |= F_MISC;
        if (nd.interaction == OT_INT_REFRACT) f |= F_REFRACT;
        if (nd.interaction == OT_INT_LENS) f |= F_LENS;
        if (nd.max_interact_count >= 0) f |= F_LIMIT;
    }
    for (int i = 0; i < s->n_materials; ++i)
        if (s->materials[i].kind == OT_MAT_CHEB) f |= F_MISC;  // series materials: evaluated by the all-features kernels only
    return f;
}

static int validate_root_grid(const ot_scene_desc* s) {
    if (s->root_grid < 0) return 0;
    if (s->root_grid + 11 > s->n_aux) return fail(OT_ERR_INVALID, "root grid out of range");
    const double* g = s->aux + s->root_grid;
    const int a0 = (int)g[0], a1 = (int)g[1], g0 = (int)g[2], g1 = (int)g[3];
    if (a0 < 0 || a0 > 2 || a1 < 0 || a1 > 2 || a0 == a1 || g0 < 1 || g1 < 1 || (int64_t)g0 * g1 > 1 << 20)
        return fail(OT_ERR_INVALID, "bad root grid header");
    const int64_t cells = (int64_t)g0 * g1;
    if (s->root_grid + 11 + cells + 1 > s->n_aux) return fail(OT_ERR_INVALID, "root grid starts out of range");
    const double* start = g + 11;
    const int64_t n_items = (int64_t)start[cells];
    if (start[0] != 0 || s->root_grid + 11 + cells + 1 + n_items > s->n_aux) return fail(OT_ERR_INVALID, "root grid items out of range");
    for (int64_t k = 0; k < cells; ++k)
        if (start[k + 1] < start[k]) return fail(OT_ERR_INVALID, "root grid starts not monotone");
    const double* items = start + cells + 1;
    for (int64_t k = 0; k < n_items; ++k) {
        const int ni = (int)items[k];
        if (ni < 0 || ni >= s->n_nodes) return fail(OT_ERR_INVALID, "root grid item out of range");
    }
    for (int i = 0; i < s->n_nodes; ++i)
        if (s->nodes[i].kind == OT_NODE_LEAF && s->nodes[i].max_interact_count >= 0)
            return fail(OT_ERR_INVALID, "a root grid cannot be combined with count-limited leaves");
    for (int i = 0; i < s->n_nodes; ++i)
        if (!(s->nodes[i].flags & OT_NODE_BOX_TRUSTED)) return fail(OT_ERR_INVALID, "a root grid needs OT_NODE_BOX_TRUSTED on every node (the walk stops early on the strength of the boxes)");
    return 0;
}

// polygon record: [nv, plane normal(3), v0(3), e1(3), e2(3) | nv x (x, y)]; returns its length or -1
static int64_t polygon_record_len(const ot_scene_desc* s, int64_t off) {
    if (off < 0 || off + 13 > s->n_aux) return -1;
    const double nv = s->aux[off];
    if (!(nv >= 3 && nv <= 4096) || nv != (double)(int)nv) return -1;
    const int64_t len = 13 + 2 * (int64_t)nv;
    return off + len <= s->n_aux ? len : -1;
}

// Chebyshev series record: [N, lo, hi | blocks x N coefficients] (trace_core.h cheb_eval); returns its length or -1
static int64_t series_record_len(const ot_scene_desc* s, int64_t off, int blocks) {
    if (off < 0 || off + 3 > s->n_aux) return -1;
    const double n = s->aux[off];
    if (!(n >= 1 && n <= 512) || n != (double)(int)n || !(s->aux[off + 2] > s->aux[off + 1])) return -1;
    const int64_t len = 3 + blocks * (int64_t)n;
    return off + len <= s->n_aux ? len : -1;
}

// CSG record: [ntok | ntok x (kind, len, body[len])], postfix over a 32-deep bit stack (trace_core.h csg_inside)
static int validate_csg(const ot_scene_desc* s, int64_t off) {
    if (off < 0 || off + 1 > s->n_aux) return fail(OT_ERR_INVALID, "CSG program out of range");
    const double ntok = s->aux[off];
    if (!(ntok >= 1 && ntok <= 1024) || ntok != (double)(int)ntok) return fail(OT_ERR_INVALID, "bad CSG token count");
    int64_t t = off + 1;
    int sp = 0;
    for (int k = 0; k < (int)ntok; ++k) {
        if (t + 2 > s->n_aux) return fail(OT_ERR_INVALID, "CSG token out of range");
        const int kind = (int)s->aux[t];
        const double len = s->aux[t + 1];
        if (!(len >= 0 && len <= 16384) || len != (double)(int)len || t + 2 + (int64_t)len > s->n_aux)
            return fail(OT_ERR_INVALID, "CSG token body out of range");
        if (kind == 100 || kind == 101) {  // union / subtract
            if (sp < 2) return fail(OT_ERR_INVALID, "CSG operator without two operands");
            sp -= 1;
        } else {
            if (kind == OT_SHAPE_CIRCLE) { if (len < 1) return fail(OT_ERR_INVALID, "CSG circle needs a radius"); }
            else if (kind == OT_SHAPE_RECT) { if (len < 2) return fail(OT_ERR_INVALID, "CSG rectangle needs two half sizes"); }
            else if (kind == OT_SHAPE_POLYGON2D) { if (polygon_record_len(s, t + 2) != (int64_t)len) return fail(OT_ERR_INVALID, "bad CSG polygon record"); }
            else return fail(OT_ERR_UNSUPPORTED, "unknown CSG primitive");
            if (++sp > 32) return fail(OT_ERR_UNSUPPORTED, "CSG program deeper than 32");
        }
        t += 2 + (int64_t)len;
    }
    if (sp != 1) return fail(OT_ERR_INVALID, "CSG program does not reduce to one value");
    return 0;
}

static int validate_scene(const ot_scene_desc* s) {
    if (!s || s->n_nodes < 0 || s->n_materials < 0 || s->n_aux < 0 || s->n_count_slots < 0) return fail(OT_ERR_INVALID, "bad scene sizes");
    if (s->n_nodes && !s->nodes) return fail(OT_ERR_INVALID, "nodes is NULL");
    if (s->n_materials && !s->materials) return fail(OT_ERR_INVALID, "materials is NULL");
    if (s->n_aux && !s->aux) return fail(OT_ERR_INVALID, "aux is NULL");
    if (s->max_children < 0 || s->max_children > 2) return fail(OT_ERR_INVALID, "max_children must be 0, 1 or 2");
    if (!(s->unit > 0)) return fail(OT_ERR_INVALID, "unit must be positive");
    for (int i = 0; i < s->n_materials; ++i) {
        const ot_material& m = s->materials[i];
        if (m.kind != OT_MAT_CONST && m.kind != OT_MAT_SELLMEIER && m.kind != OT_MAT_CHEB)
            return fail(OT_ERR_UNSUPPORTED, "unknown material kind");
        if (m.kind == OT_MAT_CHEB && series_record_len(s, (int64_t)m.n, 1) < 0) return fail(OT_ERR_INVALID, "material series record out of range");
    }
    for (int i = 0; i < s->n_nodes; ++i) {
        const ot_node& nd = s->nodes[i];
        if (nd.end <= i || nd.end > s->n_nodes) return fail(OT_ERR_INVALID, "node.end out of range at " + std::to_string(i));
        if (nd.kind == OT_NODE_LEAF) {
            if (nd.end != i + 1) return fail(OT_ERR_INVALID, "leaf.end must be index+1");
            if (nd.shape < 0 || nd.shape > OT_SHAPE_ASPHERE_CHEB) return fail(OT_ERR_UNSUPPORTED, "unknown shape kind");
            if (nd.shape == OT_SHAPE_ASPHERE_CHEB && series_record_len(s, nd.aux, 3) < 0)
                return fail(OT_ERR_INVALID, "asphere series record out of range at node " + std::to_string(i));
            if (nd.interaction < 0 || nd.interaction > OT_INT_BLOCK) return fail(OT_ERR_UNSUPPORTED, "unknown interaction kind");
            if (nd.interaction == OT_INT_REFRACT &&
                (nd.mat1 < 0 || nd.mat1 >= s->n_materials || nd.mat2 < 0 || nd.mat2 >= s->n_materials))
                return fail(OT_ERR_INVALID, "material index out of range");
            if ((nd.shape == OT_SHAPE_POLYGON2D || nd.shape == OT_SHAPE_POLYGON3D) && polygon_record_len(s, nd.aux) < 0)
                return fail(OT_ERR_INVALID, "polygon record out of range at node " + std::to_string(i));
            if (nd.shape == OT_SHAPE_CSG) {
                const int rc = validate_csg(s, nd.aux);
                if (rc) return rc;
            }
            if (nd.max_interact_count >= 0 && (nd.count_slot < 0 || nd.count_slot >= s->n_count_slots))
                return fail(OT_ERR_INVALID, "count_slot out of range");
        } else if (nd.kind != OT_NODE_GROUP) {
            return fail(OT_ERR_INVALID, "unknown node kind");
        } else if (nd.flags & OT_NODE_GRID) {  // grid record: [a0 a1 g0 g1 org0 org1 inv0 inv1 margin | start[] | items]
            if (nd.aux < 0 || nd.aux + 9 > s->n_aux) return fail(OT_ERR_INVALID, "grid record out of range");
            const double* g = s->aux + nd.aux;
            const int a0 = (int)g[0], a1 = (int)g[1], g0 = (int)g[2], g1 = (int)g[3];
            if (a0 < 0 || a0 > 2 || a1 < 0 || a1 > 2 || a0 == a1 || g0 < 1 || g1 < 1 || (int64_t)g0 * g1 > 1 << 20)
                return fail(OT_ERR_INVALID, "bad grid header");
            const int64_t cells = (int64_t)g0 * g1;
            if (nd.aux + 9 + cells + 1 > s->n_aux) return fail(OT_ERR_INVALID, "grid starts out of range");
            const double* start = g + 9;
            const int64_t n_items = (int64_t)start[cells];
            if (start[0] != 0 || nd.aux + 9 + cells + 1 + n_items > s->n_aux) return fail(OT_ERR_INVALID, "grid items out of range");
            for (int64_t k = 0; k < cells; ++k)
                if (start[k + 1] < start[k]) return fail(OT_ERR_INVALID, "grid starts not monotone");
            const double* items = start + cells + 1;
            for (int64_t k = 0; k < n_items; ++k) {
                const int ci = (int)items[k];
                if (ci <= i || ci >= nd.end || s->nodes[ci].kind != OT_NODE_LEAF) return fail(OT_ERR_INVALID, "grid item is not a leaf child");
            }
            if (!(nd.flags & OT_NODE_BOX_TRUSTED)) return fail(OT_ERR_INVALID, "a gridded group needs OT_NODE_BOX_TRUSTED on itself and its children");
            for (int ci = i + 1; ci < nd.end; ++ci)
                if (!(s->nodes[ci].flags & OT_NODE_BOX_TRUSTED)) return fail(OT_ERR_INVALID, "a gridded group needs OT_NODE_BOX_TRUSTED on itself and its children");
        }
    }
    return 0;
}

extern "C" {

int ot_abi_version(void) { return OT_ABI_VERSION; }

// Which HIP runtime did the dynamic linker bind this library to?  (a process must hold ONE libamdhip64; the
// library names it by soname only, so a copy that is already loaded — PyTorch ships its own — is the one that is used.)
int ot_runtime_info(char* path, int32_t path_capacity, int32_t* runtime_version) {
    if (runtime_version) {
        int v = 0;
        if (hipRuntimeGetVersion(&v) != hipSuccess) { v = 0; (void)hipGetLastError(); }
        *runtime_version = v;
    }
    if (path && path_capacity > 0) {
        path[0] = 0;
        Dl_info info;
        if (dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::strncpy(path, info.dli_fname, (size_t)path_capacity - 1);
            path[path_capacity - 1] = 0;
        }
    }
    return 0;
}
const char* ot_last_error(void) { return g_err.c_str(); }

int ot_ctx_create(int device, void* stream, ot_ctx** out) {
    if (!out) return fail(OT_ERR_INVALID, "out is NULL");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(OT_ERR_INVALID, "no such device");
    HIP_TRY(hipSetDevice(device));
    ot_ctx* c = new ot_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;  // NULL is the device's default (null) stream, e.g. torch's default
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        c->n_cus = prop.multiProcessorCount;
        c->lds_limit = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : prop.sharedMemPerBlock;
    }
    *out = c;
    return 0;
}

int ot_ctx_destroy(ot_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& e : c->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (c->blob64) (void)hipFree(c->blob64);
    if (c->blob32) (void)hipFree(c->blob32);
    if (c->slot_max) (void)hipFree(c->slot_max);
    if (c->gen.p) (void)hipFree(c->gen.p);
    if (c->gen_rem.p) (void)hipFree(c->gen_rem.p);
    if (c->gen_ahead.p) (void)hipFree(c->gen_ahead.p);
    if (c->trees.p) (void)hipFree(c->trees.p);
    if (c->gen_mismatch) (void)hipFree(c->gen_mismatch);
    if (c->gen_chain) (void)hipFree(c->gen_chain);
    if (c->pinned_state) (void)hipHostFree(c->pinned_state);
    if (c->scan_tmp.p) (void)hipFree(c->scan_tmp.p);
    if (c->mon.p) (void)hipFree(c->mon.p);
    if (c->blocked.p) (void)hipFree(c->blocked.p);
    delete c;
    return 0;
}

int ot_ctx_synchronize(ot_ctx* c) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int ot_ctx_set_stream(ot_ctx* c, void* stream) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    int rc = flush_events(c);  // pending timing events belong to the old stream
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)stream;
    return 0;
}

int ot_scene_upload(ot_ctx* c, const ot_scene_desc* s) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    int rc = validate_scene(s);
    if (rc) return rc;
    rc = validate_root_grid(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<uint8_t> b64, b32;
    const std::vector<NodeRun> runs = c->opt_instancing ? find_runs(s) : std::vector<NodeRun>();
    int32_t n_phys = 0;
    fill_blob<double>(s, runs, b64, n_phys, c->runs_word64);
    fill_blob<float>(s, runs, b32, n_phys, c->runs_word32);
    c->n_phys = n_phys;
    c->n_runs = (int32_t)runs.size();
    HIP_TRY(hipStreamSynchronize(c->stream));  // previous launches may still read the old scene
    if (c->blob64) { (void)hipFree(c->blob64); c->blob64 = nullptr; }
    if (c->blob32) { (void)hipFree(c->blob32); c->blob32 = nullptr; }
    HIP_TRY(hipMalloc(&c->blob64, b64.size() ? b64.size() : 16));
    HIP_TRY(hipMalloc(&c->blob32, b32.size() ? b32.size() : 16));
    if (!b64.empty()) HIP_TRY(hipMemcpy(c->blob64, b64.data(), b64.size(), hipMemcpyHostToDevice));
    if (!b32.empty()) HIP_TRY(hipMemcpy(c->blob32, b32.data(), b32.size(), hipMemcpyHostToDevice));
    c->bytes64 = b64.size(); c->bytes32 = b32.size();
    c->head64 = sizeof(DNode<double>) * (size_t)n_phys + sizeof(DMat<double>) * (size_t)s->n_materials;
    c->head32 = sizeof(DNode<float>) * (size_t)n_phys + sizeof(DMat<float>) * (size_t)s->n_materials;
    c->n_nodes = s->n_nodes; c->n_mats = s->n_materials; c->n_aux = s->n_aux;
    c->n_slots = s->n_count_slots; c->max_children = s->max_children; c->unit = s->unit;
    c->features = scene_features(s);
    c->root_grid = s->root_grid;
    c->root_pack = packable_cells(s) > 0 ? s->n_aux : -1;
    c->cache_mat = -1;
    for (int i = 0; i < s->n_materials; ++i)
        if (s->materials[i].kind != OT_MAT_CONST) { c->cache_mat = i; break; }  // the first dispersive material: evaluated once per ray
    if (c->root_grid >= 0) {
        c->features |= F_ROOT | F_AABB;
        const double* g = s->aux + s->root_grid;
        const int64_t cells = (int64_t)g[2] * (int64_t)g[3];
        const double* items = g + 11 + cells + 1;
        const int64_t n_items = (int64_t)g[11 + cells];
        for (int64_t k = 0; k < n_items; ++k)
            if (s->nodes[(int)items[k]].kind != OT_NODE_LEAF) { c->features |= F_SUBTREE; break; }
        c->root_max_items = 0;
        c->root_n_items = n_items;
        for (int64_t k = 0; k < cells; ++k) {
            const int m = (int)(g[11 + k + 1] - g[11 + k]);
            if (m > c->root_max_items) c->root_max_items = m;
        }
    }
    if (c->slot_max) { (void)hipFree(c->slot_max); c->slot_max = nullptr; }
    if (s->n_count_slots > 0) {
        std::vector<int32_t> smax(s->n_count_slots, 0);
        for (int i = 0; i < s->n_nodes; ++i)
            if (s->nodes[i].kind == OT_NODE_LEAF && s->nodes[i].max_interact_count >= 0) smax[s->nodes[i].count_slot] = s->nodes[i].max_interact_count;
        HIP_TRY(hipMalloc((void**)&c->slot_max, sizeof(int32_t) * smax.size()));
        HIP_TRY(hipMemcpy(c->slot_max, smax.data(), sizeof(int32_t) * smax.size(), hipMemcpyHostToDevice));
    }
    c->has_scene = true;
    ++c->plan_epoch;
    return 0;
}

}  // extern "C"

static int check_rays(const ot_rays* r, const char* what) {
    if (!r) return fail(OT_ERR_INVALID, std::string(what) + " is NULL");
    const void* f[] = {r->ox, r->oy, r->oz, r->dx, r->dy, r->dz, r->wavelength, r->q_re, r->q_im, r->intensity, r->n,
                       r->pathlength, r->id, r->flags};
    for (const void* p : f)
        if (!p) return fail(OT_ERR_INVALID, std::string(what) + " has a NULL field");
    return 0;
}
static int check_segs(const ot_segments* s) {
    if (!s) return fail(OT_ERR_INVALID, "segments is NULL");
    const void* f[] = {s->ox, s->oy, s->oz, s->dx, s->dy, s->dz, s->length, s->intensity, s->q_re, s->q_im, s->n,
                       s->pathlength, s->ray, s->surface};
    for (const void* p : f)
        if (!p) return fail(OT_ERR_INVALID, "segments has a NULL field");
    return 0;
}

static size_t align_up(size_t x) { return (x + 255) / 256 * 256; }

// paired segment stores (kernels.h store_segment_paired): slot k*n + i is even on even lanes iff n is even, and every
// segment array must be aligned to two elements
// fp64 only: measured on cfg 4 (1.6e8 pairs) 12.73 -> 12.08 ms (5230 -> 5510 GB/s, 97 % of the stream ceiling), cfg 2
// 0.108 -> 0.106 ms; in fp32 the pair is an 8-byte store and the kernel is VALU-bound: the exchange costs 7-13 %.
template <class T> static int32_t pair_ok(const ot_ctx* c, const ot_segments* s, int64_t n) {
    if (!c->opt_pair || (n & 1) || sizeof(T) != 8) return 0;
    const void* real[] = {s->ox, s->oy, s->oz, s->dx, s->dy, s->dz, s->length, s->intensity, s->q_re, s->q_im, s->n, s->pathlength};
    for (const void* p : real)
        if ((uintptr_t)p % (2 * sizeof(T))) return 0;
    if ((uintptr_t)s->ray % 8 || (uintptr_t)s->surface % 8) return 0;
    return 1;
}

template <class T> static SceneBlob make_blob(const ot_ctx* c) {
    constexpr bool f64 = sizeof(T) == 8;
    SceneBlob blob;
    blob.words = (const uint32_t*)(f64 ? c->blob64 : c->blob32);
    blob.n_words = (int32_t)((f64 ? c->bytes64 : c->bytes32) / 4);
    blob.n_nodes = c->n_nodes;
    blob.n_phys = c->n_phys;
    blob.n_mats = c->n_mats;
    blob.root = c->root_grid;
    blob.root_pack = c->root_pack;
    blob.cache_mat = c->cache_mat;
    blob.n_runs = c->n_runs;
    blob.runs_word = f64 ? c->runs_word64 : c->runs_word32;
    return blob;
}

// Heavy scenes: persistent waves with their own lists of live rays (k_trace_rolling).  OUT = SegsT<T>: the [k][ray] slots of
// ot_trace_*; SegPlanes<T>: the append layout of ot_trace_append_*.
template <class T, class OUT>
static int launch_rolling(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const OUT& out, const AppendCtl& ac, int32_t* seg_count,
                          int32_t* counts, int32_t n_classes) {
    using namespace preset;  // tables.h
    constexpr bool f64 = sizeof(T) == 8, append = std::is_same<OUT, SegPlanes<T>>::value;
    const SceneBlob blob = make_blob<T>(c);
    const size_t bytes = f64 ? c->bytes64 : c->bytes32;
    const uint32_t need = c->features;
    // Scenes under a top-level grid (many separate components, rays of a wave unrelated after the first bounce)
    // mix generations in a list and top it up continuously.  Scenes whose rays all run through the same sequence
    // of surfaces (cfg 5) keep generation-pure lists: mixing costs them more than the tails do (cfg 5 fp32:
    // 36.6 vs 31.6 ms; cfg 3 fp32: 5.1 vs 5.5 ms).
    const bool mix = c->opt_mix < 0 ? c->root_grid >= 0 : (c->opt_mix != 0 && c->root_grid >= 0);
    const size_t img = ((bytes + 15) / 16) * 16;
    const bool img_fits = c->opt_lds_limit_kb != 0 && img <= 140 * 1024;  // else: read from L2, by the all-features preset
    const int fr = !img_fits ? 3 : ((c->root_grid >= 0 && (need & ~FR) == 0) ? 0 : ((c->root_grid >= 0 && (need & ~FRP) == 0) ? 4 : ((need & ~FC) == 0 ? 1 : ((need & ~FD) == 0 ? 2 : 3))));
    // planar scenes under a top-level grid of leaves: candidates through a wave-wide pair queue (flat_grid_hit)
    // pairs a round of the pair queue can hold: 512, not the worst case of 64 lanes x the fullest cells (cfg 3: 1152) — a round
    // that would overflow defers lanes (flat_grid_hit), and the 1.3 KB per wave are what lets 16 waves per CU run instead of 12
    const int32_t flat_full = 64 * FLAT_CELLS * (c->root_max_items > 0 ? c->root_max_items : 1);
    const int32_t flat_room = c->opt_flat > 1 ? c->opt_flat : 512;  // (>= one lane's worst case: 2 * FLAT_CELLS * 42 items = 168)
    const int32_t flat_cap = flat_full < flat_room ? flat_full : flat_room;
    const bool flat_ok = c->opt_flat && mix && (fr == 0 || fr == 4) && c->root_pack >= 0 && flat_cap <= 8192 && c->n_runs == 0;  // queue entry = lane << 10 | index into the grid's item list
    // Where the scene image and the records of the live rays live, and how many waves share an image.  The waves never
    // synchronise after staging, so the workgroup size is only packaging: take what keeps most waves resident per CU
    // (registers and LDS decide).  Preference: image + records in LDS (a pass then waits for nothing in global memory)
    // when enough waves still fit; else image in LDS, records in the per-wave global scratch (L2); images beyond what
    // LDS holds next to the lists are read from L2 (all-features preset only).
    const size_t entry = mix ? 8 : 4;  // list entry: (ray index | segment index << 32), or the ray index alone (kernels.h)
    const size_t flat_bytes = flat_ok ? (((size_t)(FlatLds<T>::fixed_bytes + (size_t)flat_cap * 2) + 15) & ~(size_t)15) : 0;  // per wave (kernels.h)
    const size_t rec_bytes = 12 * sizeof(T) + 4 * (fr == 3 ? 2 : 1);  // per record of a live ray (kernels.h rec_int_words: the all-features preset keeps the count class)
    // List capacity.  Mixed lists (rings, a power of two): 128 (256: +4 %, 512: +25 % on cfg 3).  Generation-pure lists: the
    // longer the better for the lanes (a list shrinks as its rays die and every round ends in a partial pass: 45 lanes per
    // pass at 128 entries, 53 at 256, 58 at 512 on cfg 5), but only the first 128 positions keep their records in LDS and a
    // pass over the global part waits for its loads behind the segment stores of the pass before (one in-order counter):
    // cfg 5 fp32, append layout, 16 waves per CU: 14.9 ms at 128, 13.5 at 256, 16+ at 384 and beyond.
    // Mixed scenes: the live rays in registers, refilled in place (k_trace_refill) — no list, no records, 1.5 KB of LDS per wave
    // (pair queue) + 1.5 KB (the parked ride-along fields), so registers alone decide how many waves share a CU.
    if (mix && img_fits && c->opt_refill > 0) {
        const auto kf = refill_kernel<T, OUT>(fr, flat_ok);
        if (kf) {
            const int threads = refill_max_threads<T>(fr, flat_ok), waves = threads / 64;
            const size_t park_bytes = flat_ok ? 6 * 64 * sizeof(T) : 0;
            const size_t lds_f = img + (size_t)waves * (flat_bytes + park_bytes);
            int per_cu = 0;
            if (lds_f <= 158 * 1024) {
                if (lds_f > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kf, threads, lds_f) != hipSuccess) per_cu = 0;
            }
            if (per_cu >= 1) {
                if (c->opt_blocks_per_cu > 0) per_cu = c->opt_blocks_per_cu;
                const int64_t want = (n + 64 * (int64_t)waves - 1) / (64 * (int64_t)waves);
                const int64_t capf = (int64_t)c->n_cus * per_cu;
                const int gridf = (int)(want < capf ? want : capf);
                // rays per ticket (one atomic on the device-wide queue each): 256, less when the batch is small enough that
                // whole tickets would leave waves without work
                int64_t per_wave4 = n / ((int64_t)gridf * waves * 4);
                int32_t ticket = c->opt_refill_ticket > 0 ? c->opt_refill_ticket : (int32_t)(per_wave4 >= 256 ? 256 : (per_wave4 < 64 ? 64 : (per_wave4 / 64) * 64));
                if (c->blocked.ensure(256)) return fail(OT_ERR_HIP, "hipMalloc of the ticket counter failed");
                c->blocked_queue_off = 0;
                unsigned long long* queue = (unsigned long long*)c->blocked.p;
#ifdef OT_STAMP
                HIP_TRY(hipMemsetAsync(queue, 0, 24 * sizeof(unsigned long long), c->stream));
#else
                HIP_TRY(hipMemsetAsync(queue, 0, sizeof(unsigned long long), c->stream));
#endif
                if (append) HIP_TRY(hipMemsetAsync(ac.cursor, 0, sizeof(unsigned long long), c->stream));
                hipEvent_t ev0, ev1;
                int rc = timing_pair(c, &ev0, &ev1);
                if (rc) return rc;
                WaveScratch<T> ws = {nullptr, 0};
                hipExtLaunchKernelGGL(kf, dim3(gridf), dim3(threads), (uint32_t)lds_f, c->stream, ev0, ev1, 0u, blob, (T)c->unit, view<T>(rays), n, K, out,
                                      ac, seg_count, counts, n_classes, ws, ticket, 0, queue, 1, flat_ok ? flat_cap : 0);
                HIP_TRY(hipGetLastError());
                const int32_t shape[8] = {2, threads, per_cu, gridf, (int32_t)lds_f, ticket, 1, (flat_ok ? 1 : 0) | (append ? 4 : 0) | 32};  // bit 5: rays in registers
                for (int q = 0; q < 8; ++q) c->last_launch[q] = shape[q];
                return 0;
            }
        }
    }
    const int32_t cap0 = mix ? c->opt_list_cap : (c->opt_list_cap_pure > 0 ? c->opt_list_cap_pure : 256);
    // Generation-pure scenes of the curved-surface preset, single precision, append layout: the workgroup-wide block pool
    // (k_trace_pool) when at least 24 blocks of 64 records fit next to the image (cfg 5: 15 KB image, 41 blocks; 12.1 ms
    // against 13.8 with the per-wave lists).  Not for the [k][ray] slots unless asked for (OT_OPT_BLOCK_POOL = 1): blocks
    // that merge mix rays of many tickets, a pass then stores 64 scattered elements per plane instead of runs (20 ms
    // against 14.4).
    if (!mix && !f64 && img_fits && (c->opt_pool > 0 || (c->opt_pool < 0 && append)) && c->opt_rec_lds != 0 && K < (1 << 20)) {  // (a block's generation has 20 bits)
        const auto kp = pool_kernel<T, OUT>(fr);
        const size_t fixed = img + (64 + 16) * sizeof(uint32_t);
        const int64_t nb_fit = fixed < 158 * 1024 ? (int64_t)((158 * 1024 - fixed) / (POOL_BLOCK_WORDS * 4)) : 0;
        const int32_t NB = (int32_t)(nb_fit > 64 ? 64 : nb_fit);
        if (kp && NB >= (c->opt_pool > 0 ? 16 : 24)) {
            const size_t lds_p = fixed + (size_t)NB * POOL_BLOCK_WORDS * 4;
            HIP_TRY(hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
            const int64_t want = (n + 1023) / 1024;
            const int gridp = (int)(want < c->n_cus ? want : c->n_cus);
            if (c->blocked.ensure(256)) return fail(OT_ERR_HIP, "hipMalloc of the ticket counter failed");
            c->blocked_queue_off = 0;
            unsigned long long* queue = (unsigned long long*)c->blocked.p;
#ifdef OT_STAMP
            HIP_TRY(hipMemsetAsync(queue, 0, 24 * sizeof(unsigned long long), c->stream));
#else
            HIP_TRY(hipMemsetAsync(queue, 0, sizeof(unsigned long long), c->stream));
#endif
            if (append) HIP_TRY(hipMemsetAsync(ac.cursor, 0, sizeof(unsigned long long), c->stream));
            hipEvent_t ev0, ev1;
            int rc = timing_pair(c, &ev0, &ev1);
            if (rc) return rc;
            WaveScratch<T> ws = {nullptr, 0};
            hipExtLaunchKernelGGL(kp, dim3(gridp), dim3(1024), (uint32_t)lds_p, c->stream, ev0, ev1, 0u, blob, (T)c->unit, view<T>(rays), n, K, out,
                                  ac, seg_count, counts, n_classes, ws, NB, 0, queue, 0, c->opt_pool_jitter);
            HIP_TRY(hipGetLastError());
            const int32_t shape[8] = {2, 1024, 1, gridp, (int32_t)lds_p, NB * 64, 0, 2 | 16 | (append ? 4 : 0)};  // bit 4: block pool
            for (int q = 0; q < 8; ++q) c->last_launch[q] = shape[q];
            return 0;
        }
    }
    ot_ctx::RollingPlan& plan = c->plan[f64 ? 1 : 0][append ? 1 : 0];
    if (plan.epoch != c->plan_epoch) {
        struct Try { int waves = 0, wpb = 0, per_cu = 0; int32_t cap = 0, capl = 0; size_t lds = 0; };
        // most waves per CU for one placement: image in LDS or not, the first `capl` records of every list in LDS
        size_t flat_b = flat_bytes;  // per wave, for the queue room under evaluation (below)
        auto evaluate = [&](bool lds_img, int32_t CAP, int32_t CAPL, Try& best) -> int {
            const void* k = (const void*)rolling_kernel<T, OUT>(fr, flat_ok, lds_img, CAPL > 0);
            if (!k) return 0;
            const size_t per_wave = (size_t)CAP * entry + flat_b + rec_bytes * CAPL;
            for (int wpb = 4; wpb * 64 <= rolling_max_threads<T>(fr, flat_ok, CAPL > 0); wpb += 4) {
                const size_t lds_b = (lds_img ? img : 0) + (size_t)wpb * per_wave;
                if (lds_b > 158 * 1024) continue;
                if (lds_b > 48 * 1024) HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
                int per_cu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 64 * wpb, lds_b) != hipSuccess) per_cu = 0;
                if (per_cu * wpb > best.waves) { best.waves = per_cu * wpb; best.wpb = wpb; best.per_cu = per_cu; best.cap = CAP; best.capl = CAPL; best.lds = lds_b; }
            }
            return 0;
        };
        Try chosen;
        bool lds_img = false;
        int32_t room_sel = flat_cap;
        // The pair queue's room is LDS that every wave holds: 512 pairs are 1 KB, and with the records of the live rays in LDS the
        // sixteenth wave of a CU can hang on the last few hundred bytes (cfg 3's scene under grids whose fullest cell holds four or
        // five leaves instead of three: 12 waves per CU instead of 16, 2.6-3.0 ms instead of 2.15-2.25 — tools/sweep_root_grid.py).
        // A round that overflows a smaller queue defers lanes and costs little (cfg 3 with room for 192 pairs where its worst case is
        // 384: 2.165 against 2.13-2.17 ms at 256 ... 512); a lost quarter of the waves costs 25 %.
        // So the room is the largest of 512, 448, ... 192 (>= one lane's worst case of 168) that keeps the most waves resident;
        // OT_OPT_FLAT_QUEUE > 1 still sets it by hand.
        auto choose = [&]() -> int {
        chosen = Try();
        lds_img = false;
        // (1) image in LDS, records in LDS: all of them (mixed lists) or the front of the list (generation-pure lists).
        //     Taken when at least 12 waves per CU still fit (OT_OPT_LDS_RECORDS = 1: whenever it fits at all).
        const int rec_lds_min_waves = c->opt_rec_lds > 0 ? 4 : 12;
        if (img_fits && c->opt_rec_lds != 0) {
            Try t;
            // (the LDS part of a list is REC_LDS_POSITIONS = 128 entries, a compile-time constant of the kernels: a mixed list
            // is then exactly that long, a generation-pure one keeps the rest of its cap0 entries in global scratch)
            const int rc = evaluate(true, mix || cap0 < REC_LDS_POSITIONS ? REC_LDS_POSITIONS : cap0, REC_LDS_POSITIONS, t);
            if (rc) return rc;
            if (t.waves >= rec_lds_min_waves) { chosen = t; lds_img = true; }
        }
        // (2) image in LDS, records in global scratch
        if (!chosen.waves && img_fits) {
            for (int32_t CAP = cap0; CAP >= 128 && !chosen.waves; CAP >>= 1) {
                const int rc = evaluate(true, CAP, 0, chosen);
                if (rc) return rc;
            }
            lds_img = chosen.waves > 0;
        }
        // (3) image read from L2
        if (!chosen.waves) {
            const int rc = evaluate(false, cap0, 0, chosen);
            if (rc) return rc;
        }
        return 0;
        };
        int rc_choose = choose();
        if (rc_choose) return rc_choose;
        if (flat_ok && c->opt_flat == 1 && flat_cap > 192) {
            const Try first = chosen;
            const bool first_img = lds_img;
            Try best_t = first;
            bool best_img = first_img;
            for (int32_t room = flat_cap - 64; room >= 192; room -= 64) {
                flat_b = ((size_t)(FlatLds<T>::fixed_bytes + (size_t)room * 2) + 15) & ~(size_t)15;
                rc_choose = choose();
                if (rc_choose) return rc_choose;
                // (the policy of choose() first — records in LDS when at least 12 waves fit — then the number of waves)
                const bool rec_new = chosen.capl > 0, rec_old = best_t.capl > 0;
                if ((rec_new && !rec_old) || (rec_new == rec_old && chosen.waves > best_t.waves)) { best_t = chosen; best_img = lds_img; room_sel = room; }
            }
            chosen = best_t;
            lds_img = best_img;
        }
        if (!chosen.waves) return fail(OT_ERR_UNSUPPORTED, "no k_trace_rolling launch configuration fits this scene image");
        plan.flat_cap = room_sel;
        plan.epoch = c->plan_epoch; plan.wpb = chosen.wpb; plan.per_cu = chosen.per_cu; plan.cap = chosen.cap; plan.capl = chosen.capl; plan.lds = lds_img;
        plan.rec_lds = chosen.capl > 0; plan.lds_bytes = chosen.lds;
    }
    const int wpb = plan.wpb;
    const int32_t CAP = plan.cap;
    const auto kr = rolling_kernel<T, OUT>(fr, flat_ok, plan.lds, plan.rec_lds);
    if (!kr) return fail(OT_ERR_UNSUPPORTED, "no k_trace_rolling instantiation for this scene / option combination");
    const size_t lds_r = plan.lds_bytes;
    if (lds_r > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
    int per_cu_r = plan.per_cu;
    if (c->opt_blocks_per_cu > 0) per_cu_r = c->opt_blocks_per_cu;
    const int64_t want = (n + 64 * (int64_t)wpb - 1) / (64 * (int64_t)wpb);  // one ticket per wave at least
    const int64_t capr = (int64_t)c->n_cus * per_cu_r;
    const int gridr = (int)(want < capr ? want : capr);
    // per-wave record scratch (by list position) + the ticket counter
    const int32_t CAPL = plan.capl;
    const size_t wave_bytes = align_up((size_t)(CAP - CAPL) * rec_bytes);  // what the lists keep outside LDS
    const size_t scratch_bytes = wave_bytes * (size_t)gridr * wpb;
    if (c->blocked.ensure(scratch_bytes + 256)) return fail(OT_ERR_HIP, "hipMalloc of rolling-trace scratch failed");
    c->blocked_queue_off = scratch_bytes;
    WaveScratch<T> ws = {(uint8_t*)c->blocked.p, (int64_t)wave_bytes};
    unsigned long long* queue = (unsigned long long*)((uint8_t*)c->blocked.p + scratch_bytes);
#ifdef OT_STAMP
    HIP_TRY(hipMemsetAsync(queue, 0, 24 * sizeof(unsigned long long), c->stream));
#else
    HIP_TRY(hipMemsetAsync(queue, 0, sizeof(unsigned long long), c->stream));
#endif
    if (append) HIP_TRY(hipMemsetAsync(ac.cursor, 0, sizeof(unsigned long long), c->stream));
    hipEvent_t ev0, ev1;
    int rc = timing_pair(c, &ev0, &ev1);
    if (rc) return rc;
    hipExtLaunchKernelGGL(kr, dim3(gridr), dim3(64 * wpb), (uint32_t)lds_r, c->stream, ev0, ev1, 0u, blob, (T)c->unit, view<T>(rays), n,
                          K, out, ac, seg_count, counts, n_classes, ws, CAP, CAPL, queue, mix ? 1 : 0, flat_ok ? plan.flat_cap : 0);
    HIP_TRY(hipGetLastError());
    const int32_t shape[8] = {2, 64 * wpb, per_cu_r, gridr, (int32_t)lds_r, CAP, mix ? 1 : 0, (flat_ok ? 1 : 0) | (plan.rec_lds ? 2 : 0) | (append ? 4 : 0)};
    for (int q = 0; q < 8; ++q) c->last_launch[q] = shape[q];
    return 0;
}

static int check_trace_args(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const int32_t* seg_count, const int32_t* counts, int32_t n_classes) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    if (!c->has_scene) return fail(OT_ERR_NOSCENE, "ot_scene_upload has not been called");
    if (c->max_children > 2) return fail(OT_ERR_UNSUPPORTED, "more than two children per hit");
    int rc = check_rays(rays, "rays");
    if (rc) return rc;
    if (n < 0 || K < 1 || !seg_count) return fail(OT_ERR_INVALID, "bad n / max_segments / seg_count");
    if (n >= (int64_t)1 << 31) return fail(OT_ERR_INVALID, "n must be < 2^31 per launch (int32 ray index)");
    if (c->n_slots > 0 && (!counts || n_classes < 1)) return fail(OT_ERR_INVALID, "scene has limited surfaces: counts table required");
    return 0;
}

// does this scene take the rolling lists (heavy: many nodes per segment => VALU-bound, uneven path lengths) or one lane per
// ray with perfectly coalesced streams (light: HBM-bound)?  The all-features preset has no lane-per-ray form in double
// precision (it would need more than 256 registers): those scenes always take the lists.
static int fused_preset(uint32_t need) {  // smallest lane-per-ray preset that covers the scene (tables.h): 0 FA, 1 FB, 2 FE, 3 FM, 4 F_ALL
    using namespace preset;
    return (need & ~FA) == 0 ? 0 : ((need & ~FB) == 0 ? 1 : ((need & ~FE) == 0 ? 2 : ((need & ~FM) == 0 ? 3 : 4)));
}
template <class T> static bool wants_rolling(const ot_ctx* c, int32_t K) {
    using namespace preset;
    const bool f64 = sizeof(T) == 8;
    const int fi = fused_preset(c->features);
    const size_t bytes = f64 ? c->bytes64 : c->bytes32;
    const bool in_lds = bytes <= (size_t)c->opt_lds_limit_kb * 1024;
    return c->opt_kernel == 2 || (c->opt_kernel == 0 && c->n_nodes >= 24 && K > 2) || (f64 && (fi == 4 || !in_lds));
}

// one lane per ray (k_trace_fused); OUT = SegsT<T> or SegTiles<T>
template <class T, class OUT>
static int launch_fused(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const OUT& out, int32_t pair, int32_t* seg_count, int32_t* counts,
                        int32_t n_classes) {
    int rc = 0;
    using namespace preset;  // tables.h
    const bool f64 = sizeof(T) == 8;
    const uint32_t need = c->features;
    const int fi = fused_preset(need);
    const size_t bytes = f64 ? c->bytes64 : c->bytes32;
    const bool in_lds = bytes <= (size_t)c->opt_lds_limit_kb * 1024;
    const SceneBlob blob = make_blob<T>(c);
    const int block = 256;
    const int64_t blocks_needed = (n + block - 1) / block;
    // Grid: small scenes (staging the blob costs nothing) get up to 256 blocks per CU, i.e. one ray per
    // lane up to 1.7e7 rays and a short grid-stride loop beyond: fresh blocks replace finished ones, which
    // balances better than 16 long-lived blocks per CU (cfg 4, 1.6e8 rays: 12.7 -> 11.8 ms fp64; flat from
    // 64 per CU on).  Scenes with a large LDS image run persistent, as many blocks per CU as the image allows.
    int per_cu = 256;
    if (in_lds && bytes > 16 * 1024) {  // beyond 64 B of staging per ray a short-lived workgroup no longer pays
        const int fit = (int)((160 * 1024) / (bytes + 512));
        per_cu = fit < 1 ? 1 : (fit > 8 ? 8 : fit);
    }
    if (c->opt_blocks_per_cu > 0) per_cu = c->opt_blocks_per_cu;
    const int64_t cap = (int64_t)c->n_cus * per_cu;
    const int grid = (int)(blocks_needed < cap ? blocks_needed : cap);
    hipEvent_t ev0, ev1;
    rc = timing_pair(c, &ev0, &ev1);
    if (rc) return rc;
    // smallest instantiation that covers the scene's features; the 128-register cap pays for the mirror / lens kernel
    // and the fp32 Snell kernel only (the fp64 Snell kernel would spill: 145 VGPRs)
    const bool mw = c->opt_minw == 4 && (fi == 0 || (fi == 1 && !f64));
    FusedKern<T, OUT> kern = fused_kernel<T, OUT>(fi, in_lds, mw, c->opt_nt != 0);
    if (!kern) return fail(OT_ERR_UNSUPPORTED, "no kernel instantiation for this scene / option combination");
    const size_t lds_bytes = in_lds ? bytes : 0;
    if (lds_bytes > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), (uint32_t)lds_bytes, c->stream, ev0, ev1, 0u, blob, (T)c->unit, view<T>(rays), n, K,
                          out, seg_count, counts, n_classes, pair);
    HIP_TRY(hipGetLastError());
    const int32_t shape[8] = {1, (int32_t)block, 0, (int32_t)grid, (int32_t)lds_bytes, 0, 0, std::is_same<OUT, SegTiles<T>>::value ? 8 : 0};
    for (int q = 0; q < 8; ++q) c->last_launch[q] = shape[q];
    return 0;
}

template <class T>
static int trace_fused(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segments* out, int32_t* seg_count,
                       int32_t* counts, int32_t n_classes) {
    int rc = check_trace_args(c, rays, n, K, seg_count, counts, n_classes);
    if (rc) return rc;
    rc = check_segs(out);
    if (rc) return rc;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    if (wants_rolling<T>(c, K)) return launch_rolling<T, SegsT<T>>(c, rays, n, K, view<T>(out), AppendCtl{nullptr, 0, 0}, seg_count, counts, n_classes);
    return launch_fused<T, SegsT<T>>(c, rays, n, K, view<T>(out), pair_ok<T>(c, out, n), seg_count, counts, n_classes);
}

// Tiled layout: the lane-per-ray kernel only (light scenes; heavy ones have the append layout)
template <class T> static int check_tiles(const void* tiles, int64_t capacity, int64_t n, int32_t K) {
    if (!tiles || (uintptr_t)tiles % 16) return fail(OT_ERR_INVALID, "tiles must be a 16-byte aligned device pointer");
    if (capacity % 64 || capacity < ((n * K + 63) / 64) * 64) return fail(OT_ERR_CAPACITY, "tiled layout: capacity must be a multiple of 64 and hold max_segments * n_rays slots");
    return 0;
}
template <class T>
static int trace_tiled(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, void* tiles, int64_t capacity, int32_t* seg_count, int32_t* counts,
                       int32_t n_classes) {
    int rc = check_trace_args(c, rays, n, K, seg_count, counts, n_classes);
    if (rc) return rc;
    rc = check_tiles<T>(tiles, capacity, n, K);
    if (rc) return rc;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    if (wants_rolling<T>(c, K))
        return fail(OT_ERR_UNSUPPORTED, "the tiled layout belongs to the lane-per-ray kernel (light scenes); heavy scenes write [k][ray] slots (ot_trace_*) or the append layout (ot_trace_append_*)");
    const int32_t pair = (c->opt_pair && !(n & 1) && sizeof(T) == 8) ? 1 : 0;  // lane pairs write 16 bytes: even slot on the even lane
    return launch_fused<T, SegTiles<T>>(c, rays, n, K, SegTiles<T>{(uint8_t*)tiles}, pair, seg_count, counts, n_classes);
}

// Append layout: always the rolling lists (a light scene takes the planar preset FC).
template <class T>
static int trace_append(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segment_block* out, int64_t* n_slots,
                        int32_t* seg_count, int32_t* counts, int32_t n_classes) {
    int rc = check_trace_args(c, rays, n, K, seg_count, counts, n_classes);
    if (rc) return rc;
    if (!out || !out->base || out->capacity < 0 || !n_slots) return fail(OT_ERR_INVALID, "bad segment block / n_slots");
    if ((uintptr_t)out->base % 16 || out->capacity % 64) return fail(OT_ERR_INVALID, "segment block: base must be 16-byte aligned, capacity a multiple of 64");
    if (out->capacity >= ((int64_t)1 << 30) / (int64_t)(sizeof(T) / 4)) return fail(OT_ERR_INVALID, "segment block: capacity must stay below 2^30 slots (2^29 in double precision) per launch");
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(n_slots, 0, sizeof(int64_t), c->stream));
        return 0;
    }
    // slots per claim: one device-wide atomic per chunk.  512 keeps a cfg 3 trace (5e7 records) at 1e5 claims per launch;
    // the price is the unused tail of every wave's last chunk (holes, ray = -1: half a chunk per wave on average).
    const int32_t chunk = c->opt_append_chunk;
    const SegPlanes<T> planes = {(uint8_t*)out->base, out->capacity};
    const AppendCtl ac = {(unsigned long long*)n_slots, out->capacity, chunk};
    return launch_rolling<T, SegPlanes<T>>(c, rays, n, K, planes, ac, seg_count, counts, n_classes);
}

extern "C" {

int ot_trace_f64(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segments* out, int32_t* seg_count,
                 int32_t* counts, int32_t n_classes) {
    return trace_fused<double>(c, rays, n, K, out, seg_count, counts, n_classes);
}
int ot_trace_f32(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segments* out, int32_t* seg_count,
                 int32_t* counts, int32_t n_classes) {
    return trace_fused<float>(c, rays, n, K, out, seg_count, counts, n_classes);
}
int ot_trace_tiled_f64(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, void* tiles, int64_t capacity, int32_t* seg_count, int32_t* counts,
                       int32_t n_classes) {
    return trace_tiled<double>(c, rays, n, K, tiles, capacity, seg_count, counts, n_classes);
}
int ot_trace_tiled_f32(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, void* tiles, int64_t capacity, int32_t* seg_count, int32_t* counts,
                       int32_t n_classes) {
    return trace_tiled<float>(c, rays, n, K, tiles, capacity, seg_count, counts, n_classes);
}
int ot_trace_append_f64(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segment_block* out, int64_t* n_slots,
                        int32_t* seg_count, int32_t* counts, int32_t n_classes) {
    return trace_append<double>(c, rays, n, K, out, n_slots, seg_count, counts, n_classes);
}
int ot_trace_append_f32(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segment_block* out, int64_t* n_slots,
                        int32_t* seg_count, int32_t* counts, int32_t n_classes) {
    return trace_append<float>(c, rays, n, K, out, n_slots, seg_count, counts, n_classes);
}

}  // extern "C"

static int gen_preset(uint32_t need) {  // 0 FB (planar, no count gates), 1 FC (planar under grids), 2 FE, 3 FM, 4 F_ALL
    using namespace preset;
    return (need & ~FB) == 0 ? 0 : ((need & ~FC) == 0 ? 1 : ((need & ~FE) == 0 ? 2 : ((need & ~FM) == 0 ? 3 : 4)));
}

template <class T>
static int trace_generation(ot_ctx* c, const ot_rays* rays, const int32_t* tree, int64_t n, int32_t* budget,
                            const ot_segments* out, int64_t out_capacity, int64_t* seg_cursor, const ot_rays* next,
                            int32_t* next_tree, int64_t next_capacity, int64_t* n_next, int32_t* counts,
                            int32_t n_classes, uint8_t* ahead_in = nullptr, uint8_t* ahead_out = nullptr, bool parent_index = false) {
    // parent_index (OT_OPT_GEN_PARENT_INDEX, the single-generation entry points only): next_tree[] receives the input index of each
    // child's parent instead of its tree id.
    // ahead_in: the bytes the emit pass of the generation before left for these rays (children per ray if processed): no count
    // pass over the rays, k_gen_recount instead.  ahead_out: where this emit pass leaves them for the next generation (of
    // next_capacity bytes); NULL: plain emit.  Both live outside c->gen, which may be reallocated between generations.
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    if (!c->has_scene) return fail(OT_ERR_NOSCENE, "ot_scene_upload has not been called");
    int rc = check_rays(rays, "rays");
    if (rc) return rc;
    rc = check_rays(next, "next");
    if (rc) return rc;
    rc = check_segs(out);
    if (rc) return rc;
    if (n < 1 || !tree || !budget || !seg_cursor || !next_tree || !n_next) return fail(OT_ERR_INVALID, "bad generation arguments");
    if (n >= (int64_t)1 << 30) return fail(OT_ERR_INVALID, "generation too large");
    const int fan = c->max_children < 1 ? 1 : c->max_children;
    if (fan > 2) return fail(OT_ERR_UNSUPPORTED, "more than two children per hit");
    if (next_capacity < n * fan) return fail(OT_ERR_CAPACITY, "next_capacity < n * max_children");
    if (c->n_slots > 0 && (!counts || n_classes < 1)) return fail(OT_ERR_INVALID, "scene has limited surfaces: counts table required");
    HIP_TRY(hipSetDevice(c->device));
    // scratch carve-up
    const int ns = c->n_slots;
    const size_t sz_slot = align_up(sizeof(int32_t) * n * (ns > 0 ? ns : 1));
    const size_t sz_tot = align_up(sizeof(int64_t) * 4);
    const int64_t n_waves = (n + 63) / 64;
    const size_t sz_code = align_up((size_t)n), sz_wave = align_up(sizeof(unsigned long long) * n_waves);
    // heavy scenes keep the count pass's decision per ray for the emit pass (kernels.h k_gen_pass); OT_OPT_GEN_REUSE: -1 auto
    const bool reuse = (c->opt_gen_reuse < 0 ? c->n_nodes >= 12 : c->opt_gen_reuse != 0) && !ahead_in;
    const size_t sz_hn = reuse ? align_up(sizeof(int32_t) * n) : 0, sz_ht = reuse ? align_up(sizeof(T) * n) : 0;
    const size_t total = sz_tot + sz_code + 2 * sz_wave + (ns > 0 ? 3 * sz_slot : 0) + sz_hn + sz_ht;
    if (c->gen.ensure(total)) return fail(OT_ERR_HIP, "hipMalloc of generation scratch failed");
    uint8_t* p = (uint8_t*)c->gen.p;
    int64_t* totals = (int64_t*)p;
    unsigned long long* mismatch = nullptr;
    p += sz_tot;
    uint8_t* code = p; p += sz_code;
    unsigned long long* wave_total = (unsigned long long*)p; p += sz_wave;
    unsigned long long* wave_prefix = (unsigned long long*)p; p += sz_wave;
    int32_t *probe = nullptr, *probe_ex = nullptr, *rank = nullptr;
    if (ns > 0) {
        probe = (int32_t*)p; p += sz_slot;
        probe_ex = (int32_t*)p; p += sz_slot;
        rank = (int32_t*)p; p += sz_slot;
    }
    int32_t* hit_node = reuse ? (int32_t*)p : nullptr;
    p += sz_hn;
    T* hit_t = reuse ? (T*)p : nullptr;
    const size_t tmp = ns > 0 ? scan_tmp_bytes<int32_t>(n) : 0, tmp_w = scan_tmp_bytes<unsigned long long>(n_waves);
    if (c->scan_tmp.ensure((tmp > tmp_w ? tmp : tmp_w) + 256)) return fail(OT_ERR_HIP, "hipMalloc of scan scratch failed");
    const int block = 256;
    const int g1 = (int)((n + block - 1) / block);
    rc = timing_begin(c);
    if (rc) return rc;
    if (!c->gen_mismatch) {
        HIP_TRY(hipMalloc((void**)&c->gen_mismatch, sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(c->gen_mismatch, 0, sizeof(unsigned long long), c->stream));
    }
    mismatch = c->gen_mismatch;  // lives with the ctx: accumulated over all generations (ot_debug_generation_mismatches)
    constexpr bool f64 = sizeof(T) == 8;
    const size_t bytes = f64 ? c->bytes64 : c->bytes32;
    const SceneBlob blob = make_blob<T>(c);
    const bool in_lds = bytes <= (size_t)c->opt_lds_limit_kb * 1024;
    const size_t lds_bytes = in_lds ? bytes : 0;
    const int fg = gen_preset(c->features);  // smallest generation preset that covers the scene (tables.h)
    const ProbeKern<T> k_probe = probe_kernel<T>(fg, in_lds);
    const GenKern<T> k_count = gen_kernel<T>(fg, in_lds, false);
    const GenKern<T> k_emit = ahead_out ? gen_ahead_kernel<T>(fg, in_lds) : gen_kernel<T>(fg, in_lds, true);
    if (!k_emit) return fail(OT_ERR_UNSUPPORTED, "no look-ahead emit kernel for this scene");
    if (ahead_in) code = ahead_in;  // rewritten in place by k_gen_recount
    if (lds_bytes > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        HIP_TRY(hipFuncSetAttribute((const void*)k_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        HIP_TRY(hipFuncSetAttribute((const void*)k_emit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    if (ns > 0) {  // FIFO-exact interact-count gating: probe -> per-slot scan -> rank within the tree
        HIP_TRY(hipMemsetAsync(probe, 0, sizeof(int32_t) * n * ns, c->stream));
        hipLaunchKernelGGL(k_probe, dim3(g1), dim3(block), lds_bytes, c->stream, blob, (T)c->unit, view<T>(rays), tree, n, budget,
                           counts, n_classes, probe);
        for (int s = 0; s < ns; ++s)
            exclusive_scan<int32_t, int32_t>(c->scan_tmp.p, probe + (int64_t)s * n, probe_ex + (int64_t)s * n, n, c->stream);
        hipLaunchKernelGGL(k_gen_rank, dim3(g1), dim3(block), 0, c->stream, tree, n, ns, probe_ex, rank);
    }
    // count -> scan of the wave totals -> emit (kernels.h: k_gen_pass)
    if (ahead_in)
        hipLaunchKernelGGL(k_gen_recount, dim3(g1), dim3(block), 0, c->stream, tree, n, (const int32_t*)budget, code, wave_total, c->opt_gen_drop ? 1 : 0);
    else
        hipLaunchKernelGGL(k_count, dim3(g1), dim3(block), lds_bytes, c->stream, blob, (T)c->unit, view<T>(rays), tree, n, budget,
                           (const int64_t*)seg_cursor, view<T>(out), out_capacity, view_out<T>(next), next_tree, next_capacity, code, wave_total,
                           (const unsigned long long*)wave_prefix, counts, n_classes, (const int32_t*)rank, mismatch, hit_node, hit_t,
                           c->opt_gen_drop ? 1 : 0, (uint8_t*)nullptr);
    exclusive_scan<unsigned long long, unsigned long long>(c->scan_tmp.p, wave_total, wave_prefix, n_waves, c->stream);
    hipLaunchKernelGGL(k_gen_totals, dim3(1), dim3(64), 0, c->stream, (const unsigned long long*)wave_total,
                       (const unsigned long long*)wave_prefix, n_waves, totals, seg_cursor, n_next);
    hipLaunchKernelGGL(k_emit, dim3(g1), dim3(block), lds_bytes, c->stream, blob, (T)c->unit, view<T>(rays), tree, n, budget,
                       (const int64_t*)(totals + 2), view<T>(out), out_capacity, view_out<T>(next), next_tree, next_capacity, code, wave_total,
                       (const unsigned long long*)wave_prefix, counts, n_classes, (const int32_t*)rank, mismatch, hit_node, hit_t,
                       (c->opt_gen_drop ? 1 : 0) | (parent_index ? 2 : 0), ahead_out);
    if (ns > 0)
        hipLaunchKernelGGL(k_gen_counts, dim3(g1), dim3(block), 0, c->stream, tree, rays->id, n, ns, rank, probe, c->slot_max, counts,
                           n_classes);
    HIP_TRY(hipGetLastError());
    return timing_end(c);
}

// Whole ray trees, a lane per tree (kernels.h k_trace_trees).  The plan: workgroups of 256 threads; QL entries per lane in LDS
// (OT_OPT_TREES_LDS_ENTRIES, default 3: eight waves per CU in double precision, sixteen in single) and the rest of the
// ceil(cap / 2) entries a tree can need in a per-wave global scratch, as long as the scratch of all resident waves stays
// within 1 GiB (caps up to ~170 in double precision); beyond that the queues are what fits and the launch is a speculation
// on small trees (full = 0).
struct TreesPlan { int32_t QL, QG, full, groups_per_cu, grid, chunk, preset, flat_cap, img_global; size_t lds_bytes; };
// Planar scenes under a top-level grid of leaves search through the wave-wide pair queue of the heavy non-branching kernel
// (launch_rolling's flat_ok, flat_grid_hit): the tree kernel's presets 5 (FR) / 6 (FRP).  0: the scene does not qualify.
static int32_t trees_flat_cap(const ot_ctx* c, int* preset) {
    using namespace preset;
    const uint32_t need = c->features;
    const int fr = (c->root_grid >= 0 && (need & ~FR) == 0) ? 5 : ((c->root_grid >= 0 && (need & ~FRP) == 0) ? 6 : 0);
    const int32_t flat_full = 64 * FLAT_CELLS * (c->root_max_items > 0 ? c->root_max_items : 1);
    const int32_t flat_room = c->opt_flat > 1 ? c->opt_flat : 512;
    const int32_t flat_cap = flat_full < flat_room ? flat_full : flat_room;
    if (!c->opt_flat || !fr || c->root_pack < 0 || flat_cap > 8192 || c->n_runs != 0 || c->n_slots > 0) return 0;
    *preset = fr;
    return flat_cap;
}
template <class T> static bool trees_plan(const ot_ctx* c, int32_t cap, int64_t n, TreesPlan* p) {
    const size_t image = sizeof(T) == 8 ? c->bytes64 : c->bytes32;
    *p = TreesPlan{};
    if (!c->has_scene || c->max_children > 2 || cap < 1) return false;
    p->preset = gen_preset(c->features);
    p->flat_cap = c->opt_trees_flat ? trees_flat_cap(c, &p->preset) : 0;
    if (!tree_kernel<T, SegPlanes<T>>(p->preset)) return false;
    const size_t flat_bytes = p->flat_cap ? (((size_t)(FlatLds<T>::fixed_bytes + (size_t)p->flat_cap * 2) + 15) & ~(size_t)15) : 0;  // per wave (kernels.h)
    const size_t room = 160 * 1024 - 1024, entry = (size_t)tree_entry_bytes<T>();
    size_t img = ((image + 15) & ~(size_t)15) + 4 * flat_bytes;
    if (img + 4 * entry > room || c->opt_trees_global) {  // (OT_OPT_TREES_GLOBAL_IMAGE: test knob, every scene takes this path)
        // an image no LDS holds (thousands of leaves): the all-features kernel reads it from global memory (L2), the LDS holds queues alone
        if (!tree_kernel<T, SegPlanes<T>>(4, 0)) return false;
        p->preset = 4; p->flat_cap = 0; p->img_global = 1;
        img = 0;
        // ... and the node and material records alone in LDS when they leave room for the queues (instanced runs fold thousands of
        // lattice members into a few records: the walk's dependent reads then come from LDS, only poses and grids from L2)
        const size_t head = ((size_t)(sizeof(T) == 8 ? c->head64 : c->head32) + 16 * (size_t)c->n_runs + 15) & ~(size_t)15;  // records + the run table
        if (head > 0 && head + 4 * 3 * entry <= room / 2 && c->opt_trees_global != 2 && tree_kernel<T, SegPlanes<T>>(4, 2)) { p->img_global = 2; img = head; }
    }
    const int64_t need = ((int64_t)cap + 1) / 2, fit = (int64_t)((room - img) / (4 * entry));
    // entries in LDS: two under small caps (queues stay short: a third workgroup per CU is worth more than the third entry —
    // cfg 4 R = 0.2: 4.05 vs 4.2 ms, bushy trees under a cap of 12: 0.56 vs 0.62), three above (cap 48: 3.35 vs 3.99)
    // ... and ONE where the search is the pair queue (heavy planar scenes: their image and queues take the LDS, and a fourth workgroup
    // per CU is worth more than entries — cfg 3 with reflecting slabs 1.38 / 1.49 / 1.85 ms with one / two / three)
    const int64_t want = c->opt_trees_lds > 0 ? c->opt_trees_lds : (p->flat_cap ? 1 : (cap <= 16 ? 2 : 3));
    int64_t ql = want < need ? want : need;
    if (ql > fit) ql = fit;
    if (ql > 255) ql = 255;
    // (... but never a third entry that leaves a CU with ONE workgroup where two would fit with two entries: cfg 3 with reflecting
    // slabs in double precision, image + pair queues + rings: 4.1 ms with three entries, 3.3-3.7 with two)
    if (c->opt_trees_lds <= 0 && ql == 3 && (160 * 1024) / (img + 4 * 3 * entry + 256) < 2 && (160 * 1024) / (img + 4 * 2 * entry + 256) >= 2) ql = 2;
    p->QL = (int32_t)ql;
    p->lds_bytes = img + 4 * (size_t)ql * entry;
    const int by_lds = (int)((160 * 1024) / (p->lds_bytes + 256)), by_regs = tree_groups_by_registers<T>(p->preset);  // (waves per SIMD the kernel's registers allow)
    p->groups_per_cu = by_lds < by_regs ? (by_lds < 1 ? 1 : by_lds) : by_regs;
    // (the scratch is per workgroup of the LAUNCH: a batch of a few trees is a few workgroups, and gets long queues out of the same 1 GiB)
    const int64_t groups_needed = n > 0 ? (n + 255) / 256 : (int64_t)1 << 40, groups_most = (int64_t)c->n_cus * p->groups_per_cu;
    p->grid = (int32_t)(groups_needed < groups_most ? groups_needed : groups_most);
    const int64_t waves = (int64_t)p->grid * 4;
    // Append output: slots per claim.  A wave waits for its claim (one atomic on ONE device-wide cursor, ~2 us) at the occupancy of this
    // kernel: 512-slot chunks were 12 % of a cfg 4 R = 0.2 trace (4.06 -> 3.55 ms with 2048+).  As large as leaves the unused chunk tails
    // (one per wave) within a sixteenth of the output, 8192 at most, OT_OPT_APPEND_CHUNK at least.
    {
        const int64_t sixteenth = n > 0 ? (n * (int64_t)cap) / (waves * 16) / 64 * 64 : 8192;
        const int64_t most_chunk = sixteenth < 8192 ? sixteenth : 8192;
        p->chunk = (int32_t)(most_chunk > c->opt_append_chunk ? most_chunk : c->opt_append_chunk);
    }
    const int64_t scratch_entry = 64 * (sizeof(T) == 8 ? 96 : 48);  // lane-major records of 12 words (kernels.h)
    int64_t most = ((int64_t)1 << 30) / (waves * scratch_entry);
    if (most > 255) most = 255;  // (the kernel keeps ring positions in bytes)
    // (the scratch ring alone must hold a whole queue: pushes keep going there while the LDS entries in front of them drain)
    const int64_t qg = need > ql ? need : 0;
    p->QG = (int32_t)(qg < most ? qg : most);
    p->full = qg <= most;
    return true;
}
template <class T, class OUT>
static int launch_trees(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t cap, const OUT& out, const AppendCtl& ac, int32_t* seg_count, int32_t* counts,
                        int32_t n_classes) {
    TreesPlan p;
    if (!trees_plan<T>(c, cap, n, &p)) return fail(OT_ERR_UNSUPPORTED, "no tree kernel for this scene (a hit that emits more than two rays, or no room for the queues): use ot_trace_tree_*");
    HIP_TRY(hipSetDevice(c->device));
    const TreeKern<T, OUT> kern = tree_kernel<T, OUT>(p.preset, p.img_global == 0 ? 1 : (p.img_global == 2 ? 2 : 0));
    if (!kern) return fail(OT_ERR_UNSUPPORTED, "this scene's tree kernel writes the append layout only (ot_trace_trees_append_*)");
    const SceneBlob blob = make_blob<T>(c);
    const int grid = p.grid;  // persistent: the scratch is per workgroup
    const size_t scratch = (size_t)grid * 4 * (size_t)p.QG * 64 * (sizeof(T) == 8 ? 96 : 48);
    if (c->trees.ensure(scratch + 256)) return fail(OT_ERR_HIP, "hipMalloc of the tree queues failed");
    if (ac.cursor) HIP_TRY(hipMemsetAsync(ac.cursor, 0, sizeof(unsigned long long), c->stream));
    AppendCtl ctl = ac;
    if (ac.cursor) ctl.chunk = p.chunk;
    hipEvent_t ev0, ev1;
    int rc = timing_pair(c, &ev0, &ev1);
    if (rc) return rc;
    if (p.lds_bytes > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes));
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), (uint32_t)p.lds_bytes, c->stream, ev0, ev1, 0u, blob, (T)c->unit, view<T>(rays), n, cap, p.QL, p.QG,
                          (uint8_t*)c->trees.p, out, ctl, seg_count, counts, n_classes, c->opt_trees_refill_at, p.flat_cap);
    HIP_TRY(hipGetLastError());
    const int32_t shape[8] = {4, 256, p.groups_per_cu, (int32_t)grid, (int32_t)p.lds_bytes, p.QL, p.QG, (std::is_same<OUT, SegPlanes<T>>::value ? 4 : 0) | (p.flat_cap ? 1 : 0)};
    for (int q = 0; q < 8; ++q) c->last_launch[q] = shape[q];
    return 0;
}
template <class T>
static int trace_trees(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t cap, const ot_segments* out, int32_t* seg_count, int32_t* counts,
                       int32_t n_classes) {
    int rc = check_trace_args(c, rays, n, cap, seg_count, counts, n_classes);
    if (rc) return rc;
    rc = check_segs(out);
    if (rc) return rc;
    if (n == 0) return 0;
    return launch_trees<T, SegsT<T>>(c, rays, n, cap, view<T>(out), AppendCtl{nullptr, 0, 0}, seg_count, counts, n_classes);
}
template <class T>
static int trace_trees_append(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t cap, const ot_segment_block* out, int64_t* n_slots, int32_t* seg_count,
                              int32_t* counts, int32_t n_classes) {
    int rc = check_trace_args(c, rays, n, cap, seg_count, counts, n_classes);
    if (rc) return rc;
    if (!out || !out->base || out->capacity < 0 || !n_slots) return fail(OT_ERR_INVALID, "bad segment block / n_slots");
    if ((uintptr_t)out->base % 16 || out->capacity % 64) return fail(OT_ERR_INVALID, "segment block: base must be 16-byte aligned, capacity a multiple of 64");
    if (out->capacity >= ((int64_t)1 << 30) / (int64_t)(sizeof(T) / 4)) return fail(OT_ERR_INVALID, "segment block: capacity must stay below 2^30 slots (2^29 in double precision) per launch");
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(n_slots, 0, sizeof(int64_t), c->stream));
        return 0;
    }
    const SegPlanes<T> planes = {(uint8_t*)out->base, out->capacity};
    const AppendCtl ac = {(unsigned long long*)n_slots, out->capacity, c->opt_append_chunk};
    return launch_trees<T, SegPlanes<T>>(c, rays, n, cap, planes, ac, seg_count, counts, n_classes);
}

// One generation in one pass (kernels.h k_gen_one): zero the tile descriptors and the ticket, launch.  `rem`: what is left of
// its tree's budget for every ray of the generation; `next_rem` receives the children's.
template <class T>
static int trace_generation_one(ot_ctx* c, const ot_rays* rays, const int32_t* tree, const int32_t* rem, int64_t n, int32_t* budget, const ot_segments* out,
                                int64_t out_capacity, int64_t* state, const ot_rays* next, int32_t* next_tree, int32_t* next_rem, int64_t next_capacity,
                                int32_t* counts, int32_t n_classes, const int64_t* n_in = nullptr, int64_t* n_out = nullptr) {
    constexpr bool f64 = sizeof(T) == 8;
    const int64_t n_tiles = (n + 63) / 64, n_groups = (n + 255) / 256;  // a tile = the 64 rays of one wave
    const size_t sz_desc = align_up(sizeof(unsigned long long) * n_tiles + 8);
    // (at least 64 KB: growing the scratch frees it, which waits for the device — not between the launches of a chain)
    if (c->gen.ensure(sz_desc < 65536 ? 65536 : sz_desc)) return fail(OT_ERR_HIP, "hipMalloc of generation scratch failed");
    unsigned long long* desc = (unsigned long long*)c->gen.p;
    uint32_t* ticket = (uint32_t*)(desc + n_tiles);
    int rc = timing_begin(c);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(desc, 0, sizeof(unsigned long long) * n_tiles + 8, c->stream));
    const size_t bytes = f64 ? c->bytes64 : c->bytes32;
    const SceneBlob blob = make_blob<T>(c);
    const bool in_lds = bytes <= (size_t)c->opt_lds_limit_kb * 1024;
    const size_t lds_bytes = in_lds ? bytes : 0;
    const int fg = gen_preset(c->features);
    const GenOneKern<T> k = gen_one_kernel<T>(fg, in_lds);
    if (lds_bytes > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(k, dim3((unsigned)n_groups), dim3(256), lds_bytes, c->stream, blob, (T)c->unit, view<T>(rays), tree, rem, n, budget, state, view<T>(out),
                       out_capacity, view_out<T>(next), next_tree, next_rem, next_capacity, desc, ticket, counts, n_classes, c->opt_gen_drop ? 1 : 0, n_in, n_out);
    HIP_TRY(hipGetLastError());
    return timing_end(c);
}

// The generation loop of a whole ray tree batch (optical_table.py:115-147) on the host side of the library: one
// trace_generation per generation, the two counters read back (16 bytes, one stream synchronisation) and the two generation
// buffers swapped — the loop the Python shell used to run with a dozen ctypes conversions per turn.  It stops when the
// queue is empty, when the next generation does not fit the buffers or the segment arrays (the caller grows them and calls
// again with the pending generation as input), or when the wall clock runs out.
template <class T>
static int trace_tree(ot_ctx* c, const ot_rays* rays, const int32_t* tree, int64_t n, int32_t* budget, const ot_segments* out,
                      int64_t out_capacity, int64_t* state, const ot_rays* buf_a, int32_t* tree_a, const ot_rays* buf_b, int32_t* tree_b,
                      int64_t buf_capacity, int32_t* counts, int32_t n_classes, double max_seconds, int64_t* result) {
    if (!c || !state || !result || !buf_a || !buf_b || !tree_a || !tree_b) return fail(OT_ERR_INVALID, "NULL argument");
    if (!c->has_scene) return fail(OT_ERR_NOSCENE, "ot_scene_upload has not been called");
    HIP_TRY(hipSetDevice(c->device));
    const int fan = c->max_children < 1 ? 1 : c->max_children;
    const auto t0 = std::chrono::steady_clock::now();
    if (!c->pinned_state) HIP_TRY(hipHostMalloc((void**)&c->pinned_state, 2 * sizeof(int64_t), hipHostMallocDefault));
    int64_t* const host_state = c->pinned_state;  // (page-locked: the copy is a DMA the stream waits for, not a staged memcpy)
    HIP_TRY(hipMemcpyAsync(host_state, state, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int64_t written = host_state[0], cur_n = n, generations = 0;
    const ot_rays* cur = rays;
    const int32_t* cur_tree = tree;
    int where = 0;  // which buffer holds the pending generation: 0 the caller's rays, 1 buf_a, 2 buf_b
    int64_t reason = 0;
    // One pass per generation (k_gen_one) where OT_OPT_GEN_ONEPASS allows it and the scene has no count-limited leaves (their
    // gate needs the scans between a probe pass and the trace).  Its per-ray budgets live in the library: seeded from budget[]
    // for the first generation of this call, two buffers of buf_capacity for the children.
    const bool one_pass_kernel = [&] {
        const size_t image = sizeof(T) == 8 ? c->bytes64 : c->bytes32;
        return gen_one_kernel<T>(gen_preset(c->features), image <= (size_t)c->opt_lds_limit_kb * 1024) != nullptr;
    }();
    // OT_OPT_GEN_ONEPASS: 1 every generation, 0 none, -1 (default) the SMALL ones: a generation of a few thousand rays is a
    // handful of tiles that are all resident at once — nothing to wait for in the look-back — and one launch instead of six
    // (count, three scan kernels, totals, emit) is what a tree of six rays costs: 1.08 -> ms per table.ray_tracing call.
    constexpr int64_t ONE_PASS_SMALL = 1 << 16;
    const bool one_pass_ok = c->n_slots == 0 && c->opt_gen_onepass != 0 && n > 0 && one_pass_kernel;
    int32_t *rem_cur = nullptr, *rem_in = nullptr, *rem_a = nullptr, *rem_b = nullptr;
    bool rem_valid = false;  // rem_cur holds the budgets of the CURRENT generation's rays (seeded, or written by a one-pass generation)
    int chained_to = 0;      // buffer (1 / 2) the last generation of a chain of small generations left its children in, 0 = no chain ran
    int64_t* chain_n = nullptr;  // two device words: sizes handed from one chained generation to the next
    if (one_pass_ok) {
        if (c->gen_rem.ensure(sizeof(int32_t) * (size_t)(n + 2 * buf_capacity) + 256)) return fail(OT_ERR_HIP, "hipMalloc of the per-ray budgets failed");
        rem_in = (int32_t*)c->gen_rem.p;
        rem_a = rem_in + n;
        rem_b = rem_a + buf_capacity;
        if (!c->gen_chain) HIP_TRY(hipMalloc((void**)&c->gen_chain, 2 * sizeof(int64_t)));
        chain_n = c->gen_chain;
    }
    // Look-ahead (k_gen_pass MODE 2; OT_OPT_GEN_AHEAD): the emit pass of a two-pass generation leaves, per child, the number of
    // children that child will have; the next two-pass generation replaces its count pass over the rays by k_gen_recount over
    // those bytes.  Light scenes (no decision reuse: their search is a few planes) without count gates whose generation
    // buffers carry no `len`; two byte arrays of buf_capacity used in turn.
    const bool ahead_ok = [&] {
        const size_t image = sizeof(T) == 8 ? c->bytes64 : c->bytes32;
        const bool reuse = c->opt_gen_reuse < 0 ? c->n_nodes >= 12 : c->opt_gen_reuse != 0;
        return c->opt_gen_ahead != 0 && c->n_slots == 0 && !reuse && !buf_a->length && !buf_b->length &&
               gen_ahead_kernel<T>(gen_preset(c->features), image <= (size_t)c->opt_lds_limit_kb * 1024) != nullptr;
    }();
    uint8_t *ahead_a = nullptr, *ahead_b = nullptr, *ahead_cur = nullptr;  // ahead_cur: the bytes of the CURRENT generation, if its producer left them
    if (ahead_ok) {
        const size_t each = align_up((size_t)buf_capacity + 64);
        if (c->gen_ahead.ensure(2 * each)) return fail(OT_ERR_HIP, "hipMalloc of the look-ahead bytes failed");
        ahead_a = (uint8_t*)c->gen_ahead.p;
        ahead_b = ahead_a + each;
    }
    while (cur_n > 0) {
        if (written + cur_n > out_capacity) { reason = 1; break; }        // the segment arrays are too small for this generation
        if (cur_n * fan > buf_capacity) { reason = 2; break; }           // ... the generation buffers for the next one
        const bool to_a = where != 1;
        const bool one_pass = one_pass_ok && (c->opt_gen_onepass > 0 || cur_n <= ONE_PASS_SMALL);
        int rc;
        if (one_pass) {
            if (!rem_valid) {  // first generation of the call, or the one before took the two passes: per-ray budgets from the tree table
                rem_cur = where == 0 ? rem_in : (where == 1 ? rem_a : rem_b);
                hipLaunchKernelGGL(k_gen_seed_rem, dim3((unsigned)((cur_n + 255) / 256)), dim3(256), 0, c->stream, cur_tree, (const int32_t*)budget, cur_n, rem_cur);
                HIP_TRY(hipGetLastError());
            }
            ahead_cur = nullptr;
            rc = trace_generation_one<T>(c, cur, cur_tree, rem_cur, cur_n, budget, out, out_capacity, state, to_a ? buf_a : buf_b, to_a ? tree_a : tree_b,
                                         to_a ? rem_a : rem_b, buf_capacity, counts, n_classes, nullptr, chain_n);
            rem_cur = to_a ? rem_a : rem_b;
            rem_valid = true;
            // Small trees: further generations are enqueued WITHOUT reading anything back — each launch sized for the most rays
            // the one before can have emitted, its real size taken on the device from where that one left it (k_gen_one's n_in /
            // n_out).  One read-back per chain instead of one per generation: a tree of a handful of rays is launch and
            // synchronisation latency, nothing else.
            if (rc == 0 && (max_seconds < 0 || max_seconds > 1.0)) {  // (a chain is at most fifteen launches of a few microseconds)
                int64_t bound = cur_n * fan, written_bound = written + cur_n;
                int here = to_a ? 1 : 2, slot = 0;
                for (int link = 0; link < 15 && bound <= 4096 && bound * fan <= buf_capacity && written_bound + bound <= out_capacity; ++link) {
                    const bool from_a = here == 1;
                    rc = trace_generation_one<T>(c, from_a ? buf_a : buf_b, from_a ? tree_a : tree_b, from_a ? rem_a : rem_b, bound, budget, out, out_capacity,
                                                 state, from_a ? buf_b : buf_a, from_a ? tree_b : tree_a, from_a ? rem_b : rem_a, buf_capacity, counts, n_classes,
                                                 chain_n + slot, chain_n + (slot ^ 1));
                    if (rc) break;
                    written_bound += bound;
                    bound *= fan;
                    here = from_a ? 2 : 1;
                    slot ^= 1;
                    ++generations;
                    chained_to = here;
                }
            }
        } else {
            uint8_t* const ahead_next = ahead_ok ? (to_a ? ahead_a : ahead_b) : nullptr;
            rc = trace_generation<T>(c, cur, cur_tree, cur_n, budget, out, out_capacity, state, to_a ? buf_a : buf_b, to_a ? tree_a : tree_b,
                                     buf_capacity, state + 1, counts, n_classes, ahead_cur, ahead_next);
            ahead_cur = ahead_next;
            rem_valid = false;
        }
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(host_state, state, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // the one host synchronisation per generation (per chain of small generations)
        written = host_state[0];
        cur_n = host_state[1];
        ++generations;
        where = chained_to ? chained_to : (to_a ? 1 : 2);
        if (chained_to) rem_cur = where == 1 ? rem_a : rem_b;
        chained_to = 0;
        cur = where == 1 ? buf_a : buf_b;
        cur_tree = where == 1 ? tree_a : tree_b;
        if (max_seconds >= 0 && cur_n > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= max_seconds) { reason = 3; break; }
    }
    result[0] = written; result[1] = cur_n; result[2] = where; result[3] = generations; result[4] = reason;
    return 0;
}

extern "C" {

int ot_trace_trees_f64(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t max_trace_num, const ot_segments* out, int32_t* seg_count, int32_t* counts,
                       int32_t n_classes) {
    return trace_trees<double>(c, rays, n, max_trace_num, out, seg_count, counts, n_classes);
}
int ot_trace_trees_f32(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t max_trace_num, const ot_segments* out, int32_t* seg_count, int32_t* counts,
                       int32_t n_classes) {
    return trace_trees<float>(c, rays, n, max_trace_num, out, seg_count, counts, n_classes);
}
int ot_trace_trees_append_f64(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t max_trace_num, const ot_segment_block* out, int64_t* n_slots, int32_t* seg_count,
                              int32_t* counts, int32_t n_classes) {
    return trace_trees_append<double>(c, rays, n, max_trace_num, out, n_slots, seg_count, counts, n_classes);
}
int ot_trace_trees_append_f32(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t max_trace_num, const ot_segment_block* out, int64_t* n_slots, int32_t* seg_count,
                              int32_t* counts, int32_t n_classes) {
    return trace_trees_append<float>(c, rays, n, max_trace_num, out, n_slots, seg_count, counts, n_classes);
}
int ot_trace_trees_plan(ot_ctx* c, int32_t real_bytes, int32_t max_trace_num, int64_t n_rays, int32_t* info) {
    if (!c || !info || (real_bytes != 4 && real_bytes != 8)) return fail(OT_ERR_INVALID, "bad ot_trace_trees_plan arguments");
    TreesPlan p;
    const bool ok = real_bytes == 8 ? trees_plan<double>(c, max_trace_num, n_rays, &p) : trees_plan<float>(c, max_trace_num, n_rays, &p);
    info[0] = ok ? 1 : 0; info[1] = p.QL + p.QG; info[2] = p.full; info[3] = p.QL;
    info[4] = p.chunk; info[5] = p.grid * 4; info[6] = info[7] = 0;
    if (ok && (real_bytes == 8 ? tree_kernel<double, SegsT<double>>(p.preset, p.img_global ? 0 : 1) != nullptr : tree_kernel<float, SegsT<float>>(p.preset, p.img_global ? 0 : 1) != nullptr))
        info[0] |= 2;  // ... and writes the [k][tree] slots too
    return 0;
}
int ot_trace_tree_f64(ot_ctx* c, const ot_rays* rays, const int32_t* tree, int64_t n, int32_t* budget, const ot_segments* out,
                      int64_t out_capacity, int64_t* state, const ot_rays* buf_a, int32_t* tree_a, const ot_rays* buf_b, int32_t* tree_b,
                      int64_t buf_capacity, int32_t* counts, int32_t n_classes, double max_seconds, int64_t* result) {
    return trace_tree<double>(c, rays, tree, n, budget, out, out_capacity, state, buf_a, tree_a, buf_b, tree_b, buf_capacity, counts, n_classes,
                              max_seconds, result);
}
int ot_trace_tree_f32(ot_ctx* c, const ot_rays* rays, const int32_t* tree, int64_t n, int32_t* budget, const ot_segments* out,
                      int64_t out_capacity, int64_t* state, const ot_rays* buf_a, int32_t* tree_a, const ot_rays* buf_b, int32_t* tree_b,
                      int64_t buf_capacity, int32_t* counts, int32_t n_classes, double max_seconds, int64_t* result) {
    return trace_tree<float>(c, rays, tree, n, budget, out, out_capacity, state, buf_a, tree_a, buf_b, tree_b, buf_capacity, counts, n_classes,
                             max_seconds, result);
}

int ot_trace_generation_f64(ot_ctx* c, const ot_rays* rays, const int32_t* tree, int64_t n, int32_t* budget,
                            const ot_segments* out, int64_t out_capacity, int64_t* seg_cursor, const ot_rays* next,
                            int32_t* next_tree, int64_t next_capacity, int64_t* n_next, int32_t* counts,
                            int32_t n_classes) {
    return trace_generation<double>(c, rays, tree, n, budget, out, out_capacity, seg_cursor, next, next_tree, next_capacity, n_next,
                                    counts, n_classes, nullptr, nullptr, c && c->opt_gen_parent);
}
int ot_trace_generation_f32(ot_ctx* c, const ot_rays* rays, const int32_t* tree, int64_t n, int32_t* budget,
                            const ot_segments* out, int64_t out_capacity, int64_t* seg_cursor, const ot_rays* next,
                            int32_t* next_tree, int64_t next_capacity, int64_t* n_next, int32_t* counts,
                            int32_t n_classes) {
    return trace_generation<float>(c, rays, tree, n, budget, out, out_capacity, seg_cursor, next, next_tree, next_capacity, n_next,
                                   counts, n_classes, nullptr, nullptr, c && c->opt_gen_parent);
}

int ot_monitor_record_f64(ot_ctx* c, const ot_monitor* mon, const ot_segments* segs, int64_t n, const int32_t* seg_count,
                          int64_t n_rays, int64_t* hit_index, void* Px, void* Py, void* Pz, void* t, int64_t* n_hits) {
    if (!c || !mon || !hit_index || !Px || !Py || !Pz || !t || !n_hits) return fail(OT_ERR_INVALID, "NULL argument");
    int rc = check_segs(segs);
    if (rc) return rc;
    if (n < 0 || n >= (int64_t)1 << 31) return fail(OT_ERR_INVALID, "bad segment count");
    if (seg_count && (n_rays < 1 || n % n_rays != 0)) return fail(OT_ERR_INVALID, "n_segments must be a multiple of n_rays");
    if (!seg_count && n_rays > 0) return fail(OT_ERR_INVALID, "n_rays without seg_count: pass 0 for a list, -1 for a list with holes");
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(n_hits, 0, sizeof(int64_t), c->stream));
        return 0;
    }
    const size_t sz_hit = align_up(sizeof(int32_t) * n), sz_off = align_up(sizeof(int64_t) * n), sz_P = align_up(sizeof(double) * 3 * n),
                 sz_t = align_up(sizeof(double) * n);
    if (c->mon.ensure(sz_hit + sz_off + sz_P + sz_t)) return fail(OT_ERR_HIP, "hipMalloc of monitor scratch failed");
    uint8_t* p = (uint8_t*)c->mon.p;
    int32_t* hit = (int32_t*)p; p += sz_hit;
    int64_t* off = (int64_t*)p; p += sz_off;
    double* P = (double*)p; p += sz_P;
    double* tt = (double*)p;
    if (c->scan_tmp.ensure(scan_tmp_bytes<int64_t>(n) + 256)) return fail(OT_ERR_HIP, "hipMalloc of scan scratch failed");
    const int block = 256, grid = (int)((n + block - 1) / block);
    hipLaunchKernelGGL(k_mon_test, dim3(grid), dim3(block), 0, c->stream, *mon, view<double>(segs), n, seg_count, n_rays, hit, P, tt);
    exclusive_scan<int32_t, int64_t>(c->scan_tmp.p, hit, off, n, c->stream);
    hipLaunchKernelGGL(k_mon_compact, dim3(grid), dim3(block), 0, c->stream, hit, off, P, tt, n, hit_index, (double*)Px, (double*)Py,
                       (double*)Pz, (double*)t, n_hits);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ot_debug_last_launch(ot_ctx* c, int32_t info[8]) {
    if (!c || !info) return fail(OT_ERR_INVALID, "NULL argument");
    for (int q = 0; q < 8; ++q) info[q] = c->last_launch[q];
    return 0;
}

int ot_debug_generation_mismatches(ot_ctx* c, int64_t* out) {
    if (!c || !out) return fail(OT_ERR_INVALID, "NULL argument");
    *out = 0;
    if (!c->gen_mismatch) return 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, c->gen_mismatch, sizeof(v), hipMemcpyDeviceToHost));
    *out = (int64_t)v;
    return 0;
}

}  // extern "C"

// Which of the two [k][ray] slot layouts do THIS device's memory channels like better?  The lane-per-ray kernel is bound by
// its streams, and the stream rate of the two layouts differs by box: on some the 64-slot tiles run 12 % faster than the 14
// arrays, on others 8 % slower (same code, same clocks; stable within a box to half a percent: tools/stream_layouts2.hip).
// So it is measured, once per context and precision: cfg 2's shape (2^20 rays, 5 segments) through k_stream_ceiling in
// both layouts, interleaved, on buffers of the library's own that are freed again (1.3 GB in double precision, ~15 ms).
template <class T> static int probe_layouts(ot_ctx* c) {
    constexpr int64_t n = 1 << 20;
    constexpr int32_t K = 5;
    const int pi = sizeof(T) == 8 ? 1 : 0;
    HIP_TRY(hipSetDevice(c->device));
    const size_t in_bytes = (size_t)n * (12 * sizeof(T) + 8), slots_bytes = (size_t)n * K * (12 * sizeof(T) + 8);
    const size_t tiles_bytes = (size_t)(n * K / 64) * SegTiles<T>::TILE_BYTES;
    uint8_t* buf = nullptr;
    if (hipMalloc((void**)&buf, in_bytes + slots_bytes + tiles_bytes + 4 * n + 4096) != hipSuccess) {
        (void)hipGetLastError();
        return fail(OT_ERR_HIP, "ot_probe_layouts: no room for the probe buffers");
    }
    HIP_TRY(hipMemsetAsync(buf, 0, in_bytes, c->stream));
    ot_rays in;
    void** inf[12] = {&in.ox, &in.oy, &in.oz, &in.dx, &in.dy, &in.dz, &in.wavelength, &in.q_re, &in.q_im, &in.intensity, &in.n, &in.pathlength};
    uint8_t* p = buf;
    for (int f = 0; f < 12; ++f) { *inf[f] = p; p += n * sizeof(T); }
    in.id = (int32_t*)p; p += 4 * n;
    in.flags = (int32_t*)p; p += 4 * n;
    in.length = nullptr;
    ot_segments sg;
    void** of[12] = {&sg.ox, &sg.oy, &sg.oz, &sg.dx, &sg.dy, &sg.dz, &sg.length, &sg.intensity, &sg.q_re, &sg.q_im, &sg.n, &sg.pathlength};
    for (int f = 0; f < 12; ++f) { *of[f] = p; p += n * K * sizeof(T); }
    sg.ray = (int32_t*)p; p += 4 * n * K;
    sg.surface = (int32_t*)p; p += 4 * n * K;
    p = (uint8_t*)(((uintptr_t)p + 4095) & ~(uintptr_t)4095);
    uint8_t* tiles = p; p += tiles_bytes;
    int32_t* seg_count = (int32_t*)p;
    const int block = 256, grid = (int)((n + block - 1) / block);
    const int32_t pair = (c->opt_pair && sizeof(T) == 8) ? 1 : 0;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    double best[2] = {1e30, 1e30};
    auto launch = [&](int layout) {
        if (layout == 0) hipLaunchKernelGGL((k_stream_ceiling<T, true, SegsT<T>>), dim3(grid), dim3(block), 0, c->stream, view<T>(&in), n, K, view<T>(&sg), seg_count, pair);
        else hipLaunchKernelGGL((k_stream_ceiling<T, true, SegTiles<T>>), dim3(grid), dim3(block), 0, c->stream, view<T>(&in), n, K, SegTiles<T>{tiles}, seg_count, pair);
    };
    int rc = 0;
    for (int round = 0; round < 3 && !rc; ++round)
        for (int layout = 0; layout < 2 && !rc; ++layout) {
            for (int w = 0; w < (round == 0 ? 30 : 5); ++w) launch(layout);  // clocks up, code object loaded
            if (hipEventRecord(e0, c->stream) != hipSuccess) { rc = 1; break; }
            for (int w = 0; w < 10; ++w) launch(layout);
            float ms = 0.f;
            if (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = 1; break; }
            const double us = ms * 1e3 / 10;
            if (us < best[layout]) best[layout] = us;
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(buf);
    if (rc || hipGetLastError() != hipSuccess) return fail(OT_ERR_HIP, "ot_probe_layouts: a probe launch failed");
    c->probe_us[pi][0] = best[0];
    c->probe_us[pi][1] = best[1];
    return 0;
}

extern "C" int ot_probe_layouts(ot_ctx* c, int32_t real_bytes, double* us_slots, double* us_tiled) {
    if (!c || (real_bytes != 4 && real_bytes != 8)) return fail(OT_ERR_INVALID, "ot_probe_layouts: ctx / real_bytes (4 or 8)");
    const int pi = real_bytes == 8 ? 1 : 0;
    if (c->probe_us[pi][0] == 0) {
        const int rc = real_bytes == 8 ? probe_layouts<double>(c) : probe_layouts<float>(c);
        if (rc) return rc;
    }
    if (us_slots) *us_slots = c->probe_us[pi][0];
    if (us_tiled) *us_tiled = c->probe_us[pi][1];
    return 0;
}

// What would the library launch for this scene and batch, and which output layout do its kernels write fastest?
extern "C" int ot_trace_plan(ot_ctx* c, int32_t real_bytes, int64_t n, int32_t K, int32_t info[8]) {
    if (!c || !info || (real_bytes != 4 && real_bytes != 8) || n < 0 || K < 1) return fail(OT_ERR_INVALID, "ot_trace_plan: bad argument");
    if (!c->has_scene) return fail(OT_ERR_NOSCENE, "ot_scene_upload has not been called");
    const bool f64 = real_bytes == 8;
    const bool heavy = f64 ? wants_rolling<double>(c, K) : wants_rolling<float>(c, K);
    const int64_t limit = ((int64_t)1 << 30) / (f64 ? 2 : 1);  // append capacity: the byte offset of a slot inside a plane stays below 2^32
    for (int q = 0; q < 8; ++q) info[q] = 0;
    info[0] = heavy ? 2 : 1;
    info[1] = heavy ? 0 : 1;                      // ot_trace_tiled_* takes the scene
    info[2] = f64 ? 29 : 30;                      // log2 of the first append capacity that is refused
    int layout = 0;
    if (heavy) {
        // the dense list — unless even a tight block could not hold the worst case of this batch (the caller then falls back
        // to the slots, which have no such limit, or splits the batch)
        layout = (n * (int64_t)K + ((int64_t)1 << 23) < limit) ? 2 : 0;
    } else {
        const int pi = f64 ? 1 : 0;
        if (c->probe_us[pi][0] == 0) {
            const int rc = f64 ? probe_layouts<double>(c) : probe_layouts<float>(c);
            if (rc) return rc;
        }
        layout = c->probe_us[pi][1] < 0.985 * c->probe_us[pi][0] ? 1 : 0;  // tiles only where they win by more than the probe's noise
        info[5] = (int32_t)(c->probe_us[pi][0] * 100);  // hundredths of a microsecond per probe launch: slot arrays, tiles
        info[6] = (int32_t)(c->probe_us[pi][1] * 100);
    }
    info[3] = layout;  // 0 slot arrays (ot_trace_*), 1 tiles (ot_trace_tiled_*), 2 append (ot_trace_append_*)
    return 0;
}

extern "C" {

int ot_set_option(ot_ctx* c, int32_t option, int32_t value) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    ++c->plan_epoch;  // launch plans depend on the options
    switch (option) {
        case OT_OPT_NT_STORES: c->opt_nt = value != 0; return 0;
        case OT_OPT_PAIR_STORES: c->opt_pair = value != 0; return 0;
        case OT_OPT_MIX_GENERATIONS: c->opt_mix = value < 0 ? -1 : (value != 0); return 0;
        case OT_OPT_FLAT_QUEUE:
            if (value < 0 || (value > 1 && (value < 192 || value > 8192 || value % 64))) return fail(OT_ERR_INVALID, "OT_OPT_FLAT_QUEUE takes 0, 1, or the pairs a round may hold: a multiple of 64, 192..8192");
            c->opt_flat = value; ++c->plan_epoch; return 0;
        case OT_OPT_MIN_WAVES: 
            if (value != 0 && value != 4) return fail(OT_ERR_INVALID, "OT_OPT_MIN_WAVES takes 0 or 4");
            c->opt_minw = value; return 0;
        case OT_OPT_KERNEL:
            if (value < 0 || value > 2) return fail(OT_ERR_INVALID, "OT_OPT_KERNEL takes 0 (auto), 1 (lane per ray) or 2 (rolling lists)");
            c->opt_kernel = value; return 0;
        case OT_OPT_LDS_LIMIT_KB:
            if (value < 0 || value > 150) return fail(OT_ERR_INVALID, "OT_OPT_LDS_LIMIT_KB takes 0..150");
            c->opt_lds_limit_kb = value; return 0;
        case OT_OPT_LIST_CAP:  // generation-pure lists never wrap: any multiple of 64; mixed lists are rings: powers of two only
            if (value < 128 || value > 1024 || value % 64) return fail(OT_ERR_INVALID, "OT_OPT_LIST_CAP takes a multiple of 64, 128..1024");
            if (!(value & (value - 1))) c->opt_list_cap = value;
            c->opt_list_cap_pure = value; return 0;
        case OT_OPT_LDS_RECORDS:
            if (value < -1 || value > 1) return fail(OT_ERR_INVALID, "OT_OPT_LDS_RECORDS takes -1 (auto), 0 or 1");
            c->opt_rec_lds = value; return 0;
        case OT_OPT_APPEND_CHUNK:
            if (value < 64 || value > (1 << 20) || value % 64) return fail(OT_ERR_INVALID, "OT_OPT_APPEND_CHUNK takes a multiple of 64, 64..1048576");
            c->opt_append_chunk = value; return 0;
        case OT_OPT_GEN_REUSE:
            if (value < -1 || value > 1) return fail(OT_ERR_INVALID, "OT_OPT_GEN_REUSE takes -1 (auto), 0 or 1");
            c->opt_gen_reuse = value; return 0;
        case OT_OPT_GEN_DROP_DOOMED: c->opt_gen_drop = value != 0; return 0;
        case OT_OPT_BLOCK_POOL:
            if (value < -1 || value > 1) return fail(OT_ERR_INVALID, "OT_OPT_BLOCK_POOL takes -1 (auto), 0 or 1");
            c->opt_pool = value; return 0;
        case OT_OPT_INSTANCING: c->opt_instancing = value != 0; return 0;  // takes effect at the next ot_scene_upload
        case OT_OPT_GEN_AHEAD: c->opt_gen_ahead = value != 0; return 0;
        case OT_OPT_TREES_FLAT: c->opt_trees_flat = value != 0; return 0;
        case OT_OPT_GEN_PARENT_INDEX: c->opt_gen_parent = value != 0; return 0;
        case OT_OPT_TREES_GLOBAL_IMAGE: c->opt_trees_global = value < 0 ? 0 : (value > 2 ? 2 : value); return 0;  // 1: records in LDS when they fit; 2: everything from global memory
        case OT_OPT_TREES_REFILL_AT:
            if (value < 1 || value > 64) return fail(OT_ERR_INVALID, "OT_OPT_TREES_REFILL_AT takes 1..64");
            c->opt_trees_refill_at = value; return 0;
        case OT_OPT_TREES_LDS_ENTRIES:
            if (value < 0 || value > 64) return fail(OT_ERR_INVALID, "OT_OPT_TREES_LDS_ENTRIES takes 0 (by the cap) or 1..64");
            c->opt_trees_lds = value; return 0;
        case OT_OPT_GEN_ONEPASS:
            if (value < -1 || value > 1) return fail(OT_ERR_INVALID, "OT_OPT_GEN_ONEPASS takes -1 (small generations), 0 or 1");
            c->opt_gen_onepass = value; return 0;
        case OT_OPT_POOL_JITTER:
            if (value < 0 || value > (1 << 20)) return fail(OT_ERR_INVALID, "OT_OPT_POOL_JITTER takes 0 (off) or a period up to 2^20");
            c->opt_pool_jitter = value; return 0;
        case OT_OPT_REFILL:
            if (value < 0 || value > 1) return fail(OT_ERR_INVALID, "OT_OPT_REFILL takes 0 or 1");
            c->opt_refill = value; return 0;
        case OT_OPT_REFILL_TICKET:
            if (value < 0 || value > 4096 || value % 64) return fail(OT_ERR_INVALID, "OT_OPT_REFILL_TICKET takes 0 (by batch size) or a multiple of 64 up to 4096");
            c->opt_refill_ticket = value; return 0;
        case OT_OPT_BLOCKS_PER_CU:
            if (value < 0 || value > 65536) return fail(OT_ERR_INVALID, "OT_OPT_BLOCKS_PER_CU out of range");
            c->opt_blocks_per_cu = value; return 0;
        default: return fail(OT_ERR_INVALID, "unknown option");
    }
}

}  // extern "C"

template <class T, class OUT>
static int bench_stream(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const OUT& out, int32_t pair, int32_t* seg_count) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    int rc = check_rays(rays, "rays");
    if (rc) return rc;
    if (n < 1 || K < 1 || !seg_count || n >= (int64_t)1 << 31) return fail(OT_ERR_INVALID, "bad n / K / seg_count");
    HIP_TRY(hipSetDevice(c->device));
    const int block = 256;
    const int64_t need = (n + block - 1) / block, cap = (int64_t)c->n_cus * (c->opt_blocks_per_cu > 0 ? c->opt_blocks_per_cu : 256);  // the fused kernel's grid rule
    const int grid = (int)(need < cap ? need : cap);
    rc = timing_begin(c);
    if (rc) return rc;
    if (c->opt_nt)
        hipLaunchKernelGGL((k_stream_ceiling<T, true, OUT>), dim3(grid), dim3(block), 0, c->stream, view<T>(rays), n, K, out, seg_count, pair);
    else
        hipLaunchKernelGGL((k_stream_ceiling<T, false, OUT>), dim3(grid), dim3(block), 0, c->stream, view<T>(rays), n, K, out, seg_count, pair);
    HIP_TRY(hipGetLastError());
    return timing_end(c);
}

extern "C" {

int ot_bench_stream_f64(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segments* out, int32_t* seg_count) {
    const int rc = check_segs(out);
    return rc ? rc : bench_stream<double, SegsT<double>>(c, rays, n, K, view<double>(out), pair_ok<double>(c, out, n), seg_count);
}
int ot_bench_stream_f32(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segments* out, int32_t* seg_count) {
    const int rc = check_segs(out);
    return rc ? rc : bench_stream<float, SegsT<float>>(c, rays, n, K, view<float>(out), pair_ok<float>(c, out, n), seg_count);
}
int ot_bench_stream_tiled_f64(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, void* tiles, int64_t capacity, int32_t* seg_count) {
    const int rc = check_tiles<double>(tiles, capacity, n, K);
    return rc ? rc : bench_stream<double, SegTiles<double>>(c, rays, n, K, SegTiles<double>{(uint8_t*)tiles}, (c && c->opt_pair && !(n & 1)) ? 1 : 0, seg_count);
}
int ot_bench_stream_tiled_f32(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, void* tiles, int64_t capacity, int32_t* seg_count) {
    const int rc = check_tiles<float>(tiles, capacity, n, K);
    return rc ? rc : bench_stream<float, SegTiles<float>>(c, rays, n, K, SegTiles<float>{(uint8_t*)tiles}, 0, seg_count);
}

#ifdef OT_STAMP
// diagnostic builds only (make STAMP=1): wave-cycles the last k_trace_rolling launch spent per phase
int ot_debug_stamps(ot_ctx* c, unsigned long long* out5) {
    if (!c || !c->blocked.p) return fail(OT_ERR_INVALID, "no rolling launch yet");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out5, (uint8_t*)c->blocked.p + c->blocked_queue_off + 8 * sizeof(unsigned long long), 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif
int ot_timing_enable(ot_ctx* c, int enabled) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    if (!enabled) {
        int rc = flush_events(c);
        if (rc) return rc;
    }
    c->timing = enabled != 0;
    return 0;
}
int ot_timing_read(ot_ctx* c, double* total_ms, int64_t* launches) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    int rc = flush_events(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->total_ms;
    if (launches) *launches = c->launches;
    return 0;
}
int ot_timing_reset(ot_ctx* c) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    int rc = flush_events(c);
    if (rc) return rc;
    c->total_ms = 0.0;
    c->launches = 0;
    return 0;
}
}  // extern "C"
