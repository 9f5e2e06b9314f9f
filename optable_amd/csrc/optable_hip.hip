// optable_hip.hip — the C-ABI of liboptable_hip.so (include/optable_hip.h): context, scene validation and upload,
// kernel selection and launch.  The kernels are in kernels.h, the device math in trace_core.h.
// No CPU fallback lives here: without a GPU every entry point fails with OT_ERR_HIP.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"

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
    int32_t n_nodes = 0, n_mats = 0, n_aux = 0, n_slots = 0, max_children = 0;
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
    struct RollingPlan { uint64_t epoch = 0; int wpb = 4, per_cu = 1; int32_t cap = 128; bool lds = false, rec_lds = false; };
    RollingPlan plan[2];
    uint64_t plan_epoch = 1;
    int32_t opt_rec_lds = -1;    // pair-queue scenes: records of the live rays in LDS (-1 auto, 0 never, 1 whenever it fits)
    int32_t opt_list_cap = 128;  // k_trace_rolling: live rays per wave (cfg 3: 128 beats 256 and 512)
    int32_t opt_flat = 1;  // fp32 planar top-level-grid scenes: wave-wide pair queue (flat_grid_hit)
    int32_t opt_mix = -1;  // -1 auto (scenes under a top-level grid mix generations), 0 never
    int32_t opt_list_cap_pure = 0;  // generation-pure lists: 0 = the chunk rule below
    Scratch blocked;
    size_t blocked_queue_off = 0;
    int32_t opt_pair = 1;  // paired 16-byte segment stores in the lane-per-ray kernel
    int32_t opt_nt = 1, opt_minw = 4, opt_blocks_per_cu = 0;  // defaults from tools/tune.py on MI355X
    Scratch gen, scan_tmp, mon;
    unsigned long long* gen_mismatch = nullptr;  // count / emit disagreements of k_gen_pass (expected: 0)
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
// walk (trace_core.h flat_grid_hit): possible when the item list has at most 1024 entries and no cell more than 42.
// Returns the number of cells to append, 0 when the grid is absent or too large for the packing.
static int64_t packable_cells(const ot_scene_desc* s) {
    if (s->root_grid < 0) return 0;
    const double* g = s->aux + s->root_grid;
    const int64_t cells = (int64_t)g[2] * (int64_t)g[3];
    if ((int64_t)g[11 + cells] > 1024) return 0;
    for (int64_t k = 0; k < cells; ++k)
        if (g[11 + k + 1] - g[11 + k] > 42) return 0;
    return cells;
}
template <class T> static void fill_blob(const ot_scene_desc* s, std::vector<uint8_t>& out) {
    const int64_t pack = packable_cells(s);
    const size_t nb = sizeof(DNode<T>) * s->n_nodes, mb = sizeof(DMat<T>) * s->n_materials, ab = sizeof(T) * (s->n_aux + pack);
    out.assign(((nb + mb + ab + 15) / 16) * 16, 0);
    DNode<T>* nodes = reinterpret_cast<DNode<T>*>(out.data());
    for (int i = 0; i < s->n_nodes; ++i) {
        const ot_node& h = s->nodes[i];
        DNode<T>& d = nodes[i];
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
        d.kind = h.kind; d.end = h.end; d.flags = h.flags; d.shape = h.shape; d.inter = h.interaction;
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
    for (int64_t k = 0; k < pack; ++k) {
        const double* start = s->aux + s->root_grid + 11;
        aux[s->n_aux + k] = (T)(start[k] + 2048.0 * (start[k + 1] - start[k]));
    }
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
        if (nd.shape == OT_SHAPE_POLYGON3D) f |= F_POLY | F_MISC;
        if (nd.shape == OT_SHAPE_CYLINDER) f |= F_MISC;
        if (nd.interaction == OT_INT_REFRACT) f |= F_REFRACT;
        if (nd.interaction == OT_INT_LENS) f |= F_LENS;
        if (nd.max_interact_count >= 0) f |= F_LIMIT;
    }
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
    for (int i = 0; i < s->n_materials; ++i)
        if (s->materials[i].kind != OT_MAT_CONST && s->materials[i].kind != OT_MAT_SELLMEIER)
            return fail(OT_ERR_UNSUPPORTED, "unknown material kind");
    for (int i = 0; i < s->n_nodes; ++i) {
        const ot_node& nd = s->nodes[i];
        if (nd.end <= i || nd.end > s->n_nodes) return fail(OT_ERR_INVALID, "node.end out of range at " + std::to_string(i));
        if (nd.kind == OT_NODE_LEAF) {
            if (nd.end != i + 1) return fail(OT_ERR_INVALID, "leaf.end must be index+1");
            if (nd.shape < 0 || nd.shape > OT_SHAPE_CSG) return fail(OT_ERR_UNSUPPORTED, "unknown shape kind");
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
        }
    }
    return 0;
}

extern "C" {

int ot_abi_version(void) { return OT_ABI_VERSION; }
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
    if (c->gen_mismatch) (void)hipFree(c->gen_mismatch);
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
    fill_blob<double>(s, b64);
    fill_blob<float>(s, b32);
    HIP_TRY(hipStreamSynchronize(c->stream));  // previous launches may still read the old scene
    if (c->blob64) { (void)hipFree(c->blob64); c->blob64 = nullptr; }
    if (c->blob32) { (void)hipFree(c->blob32); c->blob32 = nullptr; }
    HIP_TRY(hipMalloc(&c->blob64, b64.size() ? b64.size() : 16));
    HIP_TRY(hipMalloc(&c->blob32, b32.size() ? b32.size() : 16));
    if (!b64.empty()) HIP_TRY(hipMemcpy(c->blob64, b64.data(), b64.size(), hipMemcpyHostToDevice));
    if (!b32.empty()) HIP_TRY(hipMemcpy(c->blob32, b32.data(), b32.size(), hipMemcpyHostToDevice));
    c->bytes64 = b64.size(); c->bytes32 = b32.size();
    c->n_nodes = s->n_nodes; c->n_mats = s->n_materials; c->n_aux = s->n_aux;
    c->n_slots = s->n_count_slots; c->max_children = s->max_children; c->unit = s->unit;
    c->features = scene_features(s);
    c->root_grid = s->root_grid;
    c->root_pack = packable_cells(s) > 0 ? s->n_aux : -1;
    c->cache_mat = -1;
    for (int i = 0; i < s->n_materials; ++i)
        if (s->materials[i].kind == OT_MAT_SELLMEIER) { c->cache_mat = i; break; }
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

// Address of one k_trace_fused instantiation, or nullptr for the combinations the launch logic never selects: the
// 128-register cap (MINW = 4) on the fp64 Snell kernel would spill (145 VGPRs wanted), so it is not even compiled.
template <class T, uint32_t FM, bool L, int W, bool N>
static auto fused_ptr() {
    using Kern = void (*)(SceneBlob, T, RaysT<T>, int64_t, int32_t, SegsT<T>, int32_t*, int32_t*, int32_t, int32_t);
    if constexpr (sizeof(T) == 8 && W == 4 && (FM & F_REFRACT) != 0) return (Kern) nullptr;
    else return (Kern)k_trace_fused<T, FM, L, W, N>;
}

// the pair-queue variants of the heavy-scene kernel (trace_core.h flat_grid_hit): planar scenes under a top-level grid
// of leaves, with circular / rectangular apertures only (FM = FR) or with polygon / boolean ones as well (FR | F_POLY)
template <class T, uint32_t FM, bool L>
static auto rolling_flat_ptr() {
    using KernR = void (*)(SceneBlob, T, RaysT<T>, int64_t, int32_t, SegsT<T>, int32_t*, int32_t*, int32_t, WaveScratch<T>, int32_t,
                           unsigned long long*, int32_t, int32_t);
    return (KernR)k_trace_rolling<T, (FM | F_FLAT), L, false>;
}
// ... and with the records of the live rays in LDS next to the scene image (single precision)
template <class T, uint32_t FM>
static auto rolling_flat_lds_ptr() {
    using KernR = void (*)(SceneBlob, T, RaysT<T>, int64_t, int32_t, SegsT<T>, int32_t*, int32_t*, int32_t, WaveScratch<T>, int32_t,
                           unsigned long long*, int32_t, int32_t);
    if constexpr (sizeof(T) == 4) return (KernR)k_trace_rolling<T, (FM | F_FLAT), true, false, true>;
    else return (KernR) nullptr;
}

template <class T>
static int trace_fused(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segments* out, int32_t* seg_count,
                       int32_t* counts, int32_t n_classes) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    if (!c->has_scene) return fail(OT_ERR_NOSCENE, "ot_scene_upload has not been called");
    if (c->max_children > 2) return fail(OT_ERR_UNSUPPORTED, "more than two children per hit");
    int rc = check_rays(rays, "rays");
    if (rc) return rc;
    rc = check_segs(out);
    if (rc) return rc;
    if (n < 0 || K < 1 || !seg_count) return fail(OT_ERR_INVALID, "bad n / max_segments / seg_count");
    if (n >= (int64_t)1 << 31) return fail(OT_ERR_INVALID, "n must be < 2^31 per launch (int32 ray index)");
    if (c->n_slots > 0 && (!counts || n_classes < 1)) return fail(OT_ERR_INVALID, "scene has limited surfaces: counts table required");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    const bool f64 = sizeof(T) == 8;
    SceneBlob blob;
    blob.words = (const uint32_t*)(f64 ? c->blob64 : c->blob32);
    const size_t bytes = f64 ? c->bytes64 : c->bytes32;
    blob.n_words = (int32_t)(bytes / 4);
    blob.n_nodes = c->n_nodes;
    blob.n_mats = c->n_mats;
    blob.root = c->root_grid;
    blob.root_pack = c->root_pack;
    blob.cache_mat = c->cache_mat;
    const int block = 256;
    const bool in_lds = bytes <= (size_t)c->opt_lds_limit_kb * 1024;
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
    // smallest instantiation that covers the scene's features, then the launch options
    constexpr uint32_t FA = F_AABB | F_LENS, FB = F_AABB | F_LENS | F_REFRACT, FC = FB | F_GRID | F_ROOT | F_SUBTREE,
                       FR = FB | F_ROOT,  // planar scenes under a top-level grid that lists leaves only (cfg 3)
                       FRP = FR | F_POLY,  // ... with polygon / boolean apertures (prisms with polygonal caps, blocks with holes)
                       FD = F_AABB | F_REFRACT | F_CURVED | F_GRID;  // spherical / aspheric optics in gridded groups (cfg 5)
    const uint32_t need = c->features;
    // Heavy scenes (many nodes per segment => VALU-bound, uneven path lengths) use the blocked
    // kernel; light ones are HBM-bound and keep one lane per ray with perfectly coalesced streams.
    const bool use_blocked = c->opt_kernel == 2 || (c->opt_kernel == 0 && c->n_nodes >= 24 && K > 2);
    if (use_blocked) {
        // Heavy scenes: persistent waves with their own lists of live rays (k_trace_rolling).
        using KernR = void (*)(SceneBlob, T, RaysT<T>, int64_t, int32_t, SegsT<T>, int32_t*, int32_t*, int32_t, WaveScratch<T>, int32_t,
                               unsigned long long*, int32_t, int32_t);
        // Scenes under a top-level grid (many separate components, rays of a wave unrelated after the first bounce)
        // mix generations in a list and top it up continuously.  Scenes whose rays all run through the same sequence
        // of surfaces (cfg 5) keep generation-pure lists: mixing costs them more than the tails do (cfg 5 fp32:
        // 36.6 vs 31.6 ms; cfg 3 fp32: 5.1 vs 5.5 ms).
        const bool mix = c->opt_mix < 0 ? c->root_grid >= 0 : (c->opt_mix != 0 && c->root_grid >= 0);
        const int fr = (c->root_grid >= 0 && (need & ~FR) == 0) ? 0 : ((c->root_grid >= 0 && (need & ~FRP) == 0) ? 4 : ((need & ~FC) == 0 ? 1 : ((need & ~FD) == 0 ? 2 : 3)));
        // [preset][image in LDS][non-temporal segment stores].  Mixed lists write the [k][ray] slots of a pass in fragments
        // of several tickets: partial lines that the L2 can merge with what neighbouring passes write if the stores are
        // PLAIN (cfg 3 fp32: 5.62 ms with non-temporal stores, 4.38 ms with plain ones, interleaved A/B); generation-pure
        // lists write longer runs and keep the non-temporal stores (cfg 5: 19.3 vs 19.6 ms).
#define OT_R(FM) {{k_trace_rolling<T, FM, false, false>, k_trace_rolling<T, FM, false, true>}, {k_trace_rolling<T, FM, true, false>, k_trace_rolling<T, FM, true, true>}}
        static const KernR tr[5][2][2] = {OT_R(FR), OT_R(FC), OT_R(FD), OT_R(F_ALL), OT_R(FRP)};
#undef OT_R
        const int nt_r = (mix || !c->opt_nt) ? 0 : 1;
        // planar scenes under a top-level grid of leaves: candidates through a wave-wide pair queue (flat_grid_hit)
        const int32_t flat_cap = 64 * FLAT_CELLS * (c->root_max_items > 0 ? c->root_max_items : 1);
        const bool flat_ok = c->opt_flat && mix && (fr == 0 || fr == 4) && c->root_pack >= 0 && flat_cap <= 8192;  // queue entry = lane << 10 | index into the grid's item list
        static const KernR flat_k[2][2] = {{rolling_flat_ptr<T, FR, false>(), rolling_flat_ptr<T, FR, true>()},
                                           {rolling_flat_ptr<T, FRP, false>(), rolling_flat_ptr<T, FRP, true>()}};
        const int fp = fr == 4 ? 1 : 0;
        static const int max_threads[5] = {blocked_threads<T, FR>(), blocked_threads<T, FC>(), blocked_threads<T, FD>(), blocked_threads<T, F_ALL>(),
                                           blocked_threads<T, FRP>()};
        // Where the scene image lives and how many waves share it.  The waves never synchronise after staging, so the
        // workgroup size is only packaging: take the one that keeps most waves resident per CU (registers and LDS
        // decide; cfg 5 fp32: 54 KB image, 152 VGPRs -> one 768-thread workgroup = 12 waves, against 2 x 256 threads
        // = 8 waves).  Images beyond what LDS holds next to the lists are read from L2.
        const size_t img = ((bytes + 15) / 16) * 16;
        const size_t entry = sizeof(unsigned long long);
        const size_t flat_bytes = ((size_t)(FlatLds<T>::fixed_bytes + (size_t)flat_cap * 2) + 15) & ~(size_t)15;  // per wave (kernels.h)
        constexpr int REC_LDS_MIN_WAVES = 12;
        const size_t rec_bytes = 12 * sizeof(T) + 12;  // per record of a live ray
        const KernR kl = (flat_ok && !f64) ? (fp ? rolling_flat_lds_ptr<T, FRP>() : rolling_flat_lds_ptr<T, FR>()) : (KernR) nullptr;  // pair queue + records in LDS (fp32: a record is 60 bytes)
        int32_t cap0 = mix ? c->opt_list_cap : (c->opt_list_cap_pure > 0 ? c->opt_list_cap_pure : 256);
        int best_wpb = 4, best_waves = 0, best_per_cu = 1, best_cap = cap0;
        bool best_lds = false, rec_lds = false;
        ot_ctx::RollingPlan& plan = c->plan[f64 ? 1 : 0];
        if (plan.epoch == c->plan_epoch) {
            best_wpb = plan.wpb; best_per_cu = plan.per_cu; best_cap = plan.cap; best_lds = plan.lds; rec_lds = plan.rec_lds; best_waves = 1;
        } else {
            for (int pass = 0; pass < 2 && best_waves == 0; ++pass) {  // pass 0: image in LDS; pass 1: image in L2
                const bool lds_img = pass == 0;
                if (lds_img && (c->opt_lds_limit_kb == 0 || img > 140 * 1024)) continue;
                for (int wpb = 4; wpb * 64 <= (flat_ok ? blocked_threads<T, FR | F_FLAT>() : max_threads[fr]); wpb += 4) {
                    int32_t CAP = cap0;
                    while (CAP > 128 && lds_img && img + (size_t)wpb * CAP * entry > 156 * 1024) CAP >>= 1;
                    const size_t lds_b = (lds_img ? img : 0) + (size_t)wpb * CAP * entry + (flat_ok ? (size_t)wpb * flat_bytes : 0);
                    if (lds_b > 158 * 1024) continue;
                    KernR kq = flat_ok ? flat_k[fp][lds_img ? 1 : 0] : tr[fr][lds_img ? 1 : 0][nt_r];
                    if (lds_b > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
                    int per_cu = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kq, 64 * wpb, lds_b) != hipSuccess) per_cu = 0;
                    if (per_cu * wpb > best_waves) { best_waves = per_cu * wpb; best_wpb = wpb; best_per_cu = per_cu; best_cap = CAP; best_lds = lds_img; }
                }
            }
            // Pair-queue scenes: the records of the live rays (15 words x CAP per wave) in LDS as well, when at least
            // REC_LDS_MIN_WAVES waves per CU still fit.  A pass then touches global memory only for a ray's first load and
            // for the segment records it writes — nothing it has to wait for (gfx9 retires loads and stores in order, so with
            // the records in global memory every pass's loads queue behind the 29 stores of the pass before).
            // (Offered to the curved-surface preset too, cfg 5 fp32: its 74 KB image leaves room for 8 such waves only —
            // 31.2 instead of 19.7 ms — so the variant is compiled for the pair-queue preset alone.)
            if (kl && c->opt_rec_lds != 0 && c->opt_lds_limit_kb != 0 && img <= 140 * 1024) {
                int rw = 0, rwpb = 0, rcap = 0, rper = 0;
                for (int32_t capl = cap0; capl >= 128; capl >>= 1) {
                    const size_t per_wave = (size_t)capl * entry + (flat_ok ? flat_bytes : 0) + rec_bytes * capl;
                    for (int wpb = 4; wpb <= 12; wpb += 4) {
                        const size_t lds_b = img + (size_t)wpb * per_wave;
                        if (lds_b > 158 * 1024) continue;
                        HIP_TRY(hipFuncSetAttribute((const void*)kl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
                        int per_cu = 0;
                        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kl, 64 * wpb, lds_b) != hipSuccess) per_cu = 0;
                        if (per_cu * wpb > rw) { rw = per_cu * wpb; rwpb = wpb; rcap = capl; rper = per_cu; }
                    }
                    if (rw >= REC_LDS_MIN_WAVES) break;  // the largest list that still gives the waves
                }
                if (rw >= (c->opt_rec_lds > 0 ? 4 : REC_LDS_MIN_WAVES)) {
                    rec_lds = true; best_wpb = rwpb; best_cap = rcap; best_lds = true; best_waves = rw; best_per_cu = rper;
                }
            }
            if (best_waves == 0) return fail(OT_ERR_HIP, "no launch configuration fits this scene image");
            plan.epoch = c->plan_epoch; plan.wpb = best_wpb; plan.per_cu = best_per_cu; plan.cap = best_cap; plan.lds = best_lds; plan.rec_lds = rec_lds;
        }
        const int wpb = best_wpb;
        const int32_t CAP = best_cap;
        KernR kr = rec_lds ? kl : (flat_ok ? flat_k[fp][best_lds ? 1 : 0] : tr[fr][best_lds ? 1 : 0][nt_r]);
        const size_t lds_r = (best_lds ? img : 0) + (size_t)wpb * CAP * entry + (flat_ok ? (size_t)wpb * flat_bytes : 0) + (rec_lds ? (size_t)wpb * rec_bytes * CAP : 0);
        if (lds_r > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
        int per_cu_r = best_per_cu;
        if (c->opt_blocks_per_cu > 0) per_cu_r = c->opt_blocks_per_cu;
        const int64_t want = (n + 64 * (int64_t)wpb - 1) / (64 * (int64_t)wpb);  // one ticket per wave at least
        const int64_t capr = (int64_t)c->n_cus * per_cu_r;
        const int gridr = (int)(want < capr ? want : capr);
        // per-wave record scratch (by list position) + the ticket counter
        const size_t wave_bytes = align_up((size_t)CAP * (12 * sizeof(T) + 12));
        const size_t scratch_bytes = wave_bytes * (size_t)gridr * wpb;
        if (c->blocked.ensure(scratch_bytes + 256)) return fail(OT_ERR_HIP, "hipMalloc of rolling-trace scratch failed");
        c->blocked_queue_off = scratch_bytes;
        WaveScratch<T> ws = {(uint8_t*)c->blocked.p, (int64_t)wave_bytes, CAP};
        unsigned long long* queue = (unsigned long long*)((uint8_t*)c->blocked.p + scratch_bytes);
#ifdef OT_STAMP
        HIP_TRY(hipMemsetAsync(queue, 0, 24 * sizeof(unsigned long long), c->stream));
#else
        HIP_TRY(hipMemsetAsync(queue, 0, sizeof(unsigned long long), c->stream));
#endif
        hipEvent_t ev0, ev1;
        rc = timing_pair(c, &ev0, &ev1);
        if (rc) return rc;
        hipExtLaunchKernelGGL(kr, dim3(gridr), dim3(64 * wpb), (uint32_t)lds_r, c->stream, ev0, ev1, 0u, blob, (T)c->unit, view<T>(rays), n,
                              K, view<T>(out), seg_count, counts, n_classes, ws, CAP, queue, mix ? 1 : 0, flat_ok ? flat_cap : 0);
        HIP_TRY(hipGetLastError());
        const int32_t shape[8] = {2, 64 * wpb, per_cu_r, gridr, (int32_t)lds_r, CAP, mix ? 1 : 0, (flat_ok ? 1 : 0) | (rec_lds ? 2 : 0)};
        for (int q = 0; q < 8; ++q) c->last_launch[q] = shape[q];
        return 0;
    }
    hipEvent_t ev0, ev1;
    rc = timing_pair(c, &ev0, &ev1);
    if (rc) return rc;
    using Kern = void (*)(SceneBlob, T, RaysT<T>, int64_t, int32_t, SegsT<T>, int32_t*, int32_t*, int32_t, int32_t);
    const int fi = (need & ~FA) == 0 ? 0 : ((need & ~FB) == 0 ? 1 : 2);
    // the 128-register cap pays for the mirror/lens kernel only; the Snell kernel would spill (fp64: 145 VGPRs)
    const int mw = (c->opt_minw == 4 && (fi == 0 || (fi == 1 && !f64))) ? 1 : 0, nt = c->opt_nt ? 1 : 0;
#define OT_K(FM, L, W, N) fused_ptr<T, FM, L, W, N>()
#define OT_ROW(FM) {{{OT_K(FM, false, 1, false), OT_K(FM, false, 1, true)}, {OT_K(FM, false, 4, false), OT_K(FM, false, 4, true)}}, \
                    {{OT_K(FM, true, 1, false), OT_K(FM, true, 1, true)}, {OT_K(FM, true, 4, false), OT_K(FM, true, 4, true)}}}
    static const Kern table[3][2][2][2] = {OT_ROW(FA), OT_ROW(FB),
                                           {{{OT_K(F_ALL, false, 1, false), OT_K(F_ALL, false, 1, true)}, {nullptr, nullptr}},
                                            {{OT_K(F_ALL, true, 1, false), OT_K(F_ALL, true, 1, true)}, {nullptr, nullptr}}}};
#undef OT_ROW
#undef OT_K
    Kern kern = table[fi][in_lds ? 1 : 0][mw][nt];
    if (!kern) return fail(OT_ERR_UNSUPPORTED, "no kernel instantiation for this scene / option combination");
    const size_t lds_bytes = in_lds ? bytes : 0;
    if (lds_bytes > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), (uint32_t)lds_bytes, c->stream, ev0, ev1, 0u, blob, (T)c->unit, view<T>(rays), n, K,
                          view<T>(out), seg_count, counts, n_classes, pair_ok<T>(c, out, n));
    HIP_TRY(hipGetLastError());
    const int32_t shape[8] = {1, (int32_t)block, 0, (int32_t)grid, (int32_t)lds_bytes, 0, 0, 0};
    for (int q = 0; q < 8; ++q) c->last_launch[q] = shape[q];
    return 0;
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

}  // extern "C"

template <class T>
static int trace_generation(ot_ctx* c, const ot_rays* rays, const int32_t* tree, int64_t n, int32_t* budget,
                            const ot_segments* out, int64_t out_capacity, int64_t* seg_cursor, const ot_rays* next,
                            int32_t* next_tree, int64_t next_capacity, int64_t* n_next, int32_t* counts,
                            int32_t n_classes) {
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
    const size_t total = sz_tot + sz_code + 2 * sz_wave + (ns > 0 ? 3 * sz_slot : 0);
    if (c->gen.ensure(total)) return fail(OT_ERR_HIP, "hipMalloc of generation scratch failed");
    uint8_t* p = (uint8_t*)c->gen.p;
    int64_t* totals = (int64_t*)p;
    unsigned long long* mismatch = nullptr;
    p += sz_tot;
    uint8_t* code = p; p += sz_code;
    unsigned long long* wave_total = (unsigned long long*)p; p += sz_wave;
    unsigned long long* wave_prefix = (unsigned long long*)p; p += sz_wave;
    int32_t* probe = (int32_t*)p; p += sz_slot;
    int32_t* probe_ex = (int32_t*)p; p += sz_slot;
    int32_t* rank = (int32_t*)p;
    size_t tmp = 0, tmp_w = 0;
    if (ns > 0) hipcub::DeviceScan::ExclusiveSum((void*)nullptr, tmp, probe, probe_ex, (int)n, c->stream);
    hipcub::DeviceScan::ExclusiveSum((void*)nullptr, tmp_w, wave_total, wave_prefix, (int)n_waves, c->stream);
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
    SceneBlob blob;
    constexpr bool f64 = sizeof(T) == 8;
    const size_t bytes = f64 ? c->bytes64 : c->bytes32;
    blob.words = (const uint32_t*)(f64 ? c->blob64 : c->blob32);
    blob.n_words = (int32_t)(bytes / 4);
    blob.n_nodes = c->n_nodes;
    blob.n_mats = c->n_mats;
    blob.root = c->root_grid;
    blob.root_pack = c->root_pack;
    blob.cache_mat = c->cache_mat;
    const bool in_lds = bytes <= (size_t)c->opt_lds_limit_kb * 1024;
    const size_t lds_bytes = in_lds ? bytes : 0;
    // beam splitters and partially reflecting slabs are planar scenes: they get the small instantiation
    // (145 instead of 255 VGPRs, 3 waves/SIMD instead of 1); count gates need the full one
    constexpr uint32_t FG = F_AABB | F_LENS | F_REFRACT;
    const bool small = (c->features & ~FG) == 0;
    auto k_probe = in_lds ? k_gen_probe<T, F_ALL, true> : k_gen_probe<T, F_ALL, false>;
    auto k_count = small ? (in_lds ? k_gen_pass<T, FG, true, false> : k_gen_pass<T, FG, false, false>)
                         : (in_lds ? k_gen_pass<T, F_ALL, true, false> : k_gen_pass<T, F_ALL, false, false>);
    auto k_emit = small ? (in_lds ? k_gen_pass<T, FG, true, true> : k_gen_pass<T, FG, false, true>)
                        : (in_lds ? k_gen_pass<T, F_ALL, true, true> : k_gen_pass<T, F_ALL, false, true>);
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
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->scan_tmp.p, tmp, probe + (int64_t)s * n, probe_ex + (int64_t)s * n, (int)n, c->stream));
        hipLaunchKernelGGL(k_gen_rank, dim3(g1), dim3(block), 0, c->stream, tree, n, ns, probe_ex, rank);
    }
    // count -> scan of the wave totals -> emit (kernels.h: k_gen_pass)
    hipLaunchKernelGGL(k_count, dim3(g1), dim3(block), lds_bytes, c->stream, blob, (T)c->unit, view<T>(rays), tree, n, budget,
                       (const int64_t*)seg_cursor, view<T>(out), out_capacity, view_out<T>(next), next_tree, next_capacity, code, wave_total,
                       (const unsigned long long*)wave_prefix, counts, n_classes, (const int32_t*)rank, mismatch);
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->scan_tmp.p, tmp_w, wave_total, wave_prefix, (int)n_waves, c->stream));
    hipLaunchKernelGGL(k_gen_totals, dim3(1), dim3(64), 0, c->stream, (const unsigned long long*)wave_total,
                       (const unsigned long long*)wave_prefix, n_waves, totals, seg_cursor, n_next);
    hipLaunchKernelGGL(k_emit, dim3(g1), dim3(block), lds_bytes, c->stream, blob, (T)c->unit, view<T>(rays), tree, n, budget,
                       (const int64_t*)(totals + 2), view<T>(out), out_capacity, view_out<T>(next), next_tree, next_capacity, code, wave_total,
                       (const unsigned long long*)wave_prefix, counts, n_classes, (const int32_t*)rank, mismatch);
    if (ns > 0)
        hipLaunchKernelGGL(k_gen_counts, dim3(g1), dim3(block), 0, c->stream, tree, rays->id, n, ns, rank, probe, c->slot_max, counts,
                           n_classes);
    HIP_TRY(hipGetLastError());
    return timing_end(c);
}

extern "C" {

int ot_trace_generation_f64(ot_ctx* c, const ot_rays* rays, const int32_t* tree, int64_t n, int32_t* budget,
                            const ot_segments* out, int64_t out_capacity, int64_t* seg_cursor, const ot_rays* next,
                            int32_t* next_tree, int64_t next_capacity, int64_t* n_next, int32_t* counts,
                            int32_t n_classes) {
    return trace_generation<double>(c, rays, tree, n, budget, out, out_capacity, seg_cursor, next, next_tree, next_capacity, n_next,
                                    counts, n_classes);
}
int ot_trace_generation_f32(ot_ctx* c, const ot_rays* rays, const int32_t* tree, int64_t n, int32_t* budget,
                            const ot_segments* out, int64_t out_capacity, int64_t* seg_cursor, const ot_rays* next,
                            int32_t* next_tree, int64_t next_capacity, int64_t* n_next, int32_t* counts,
                            int32_t n_classes) {
    return trace_generation<float>(c, rays, tree, n, budget, out, out_capacity, seg_cursor, next, next_tree, next_capacity, n_next,
                                   counts, n_classes);
}

int ot_monitor_record_f64(ot_ctx* c, const ot_monitor* mon, const ot_segments* segs, int64_t n, const int32_t* seg_count,
                          int64_t n_rays, int64_t* hit_index, void* Px, void* Py, void* Pz, void* t, int64_t* n_hits) {
    if (!c || !mon || !hit_index || !Px || !Py || !Pz || !t || !n_hits) return fail(OT_ERR_INVALID, "NULL argument");
    int rc = check_segs(segs);
    if (rc) return rc;
    if (n < 0 || n >= (int64_t)1 << 31) return fail(OT_ERR_INVALID, "bad segment count");
    if (seg_count && (n_rays < 1 || n % n_rays != 0)) return fail(OT_ERR_INVALID, "n_segments must be a multiple of n_rays");
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
    size_t tmp = 0;
    hipcub::DeviceScan::ExclusiveSum((void*)nullptr, tmp, hit, off, (int)n, c->stream);
    if (c->scan_tmp.ensure(tmp + 256)) return fail(OT_ERR_HIP, "hipMalloc of scan scratch failed");
    const int block = 256, grid = (int)((n + block - 1) / block);
    hipLaunchKernelGGL(k_mon_test, dim3(grid), dim3(block), 0, c->stream, *mon, view<double>(segs), n, seg_count, n_rays, hit, P, tt);
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->scan_tmp.p, tmp, hit, off, (int)n, c->stream));
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

int ot_set_option(ot_ctx* c, int32_t option, int32_t value) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    ++c->plan_epoch;  // launch plans depend on the options
    switch (option) {
        case OT_OPT_NT_STORES: c->opt_nt = value != 0; return 0;
        case OT_OPT_PAIR_STORES: c->opt_pair = value != 0; return 0;
        case OT_OPT_MIX_GENERATIONS: c->opt_mix = value < 0 ? -1 : (value != 0); return 0;
        case OT_OPT_FLAT_QUEUE: c->opt_flat = value != 0; return 0;
        case OT_OPT_MIN_WAVES: 
            if (value != 0 && value != 4) return fail(OT_ERR_INVALID, "OT_OPT_MIN_WAVES takes 0 or 4");
            c->opt_minw = value; return 0;
        case OT_OPT_KERNEL:
            if (value < 0 || value > 2) return fail(OT_ERR_INVALID, "OT_OPT_KERNEL takes 0 (auto), 1 (lane per ray) or 2 (rolling lists)");
            c->opt_kernel = value; return 0;
        case OT_OPT_LDS_LIMIT_KB:
            if (value < 0 || value > 150) return fail(OT_ERR_INVALID, "OT_OPT_LDS_LIMIT_KB takes 0..150");
            c->opt_lds_limit_kb = value; return 0;
        case OT_OPT_LIST_CAP:
            if (value < 64 || value > 1024 || (value & (value - 1))) return fail(OT_ERR_INVALID, "OT_OPT_LIST_CAP takes a power of two, 64..1024");
            c->opt_list_cap = value; c->opt_list_cap_pure = value; return 0;
        case OT_OPT_LDS_RECORDS:
            if (value < -1 || value > 1) return fail(OT_ERR_INVALID, "OT_OPT_LDS_RECORDS takes -1 (auto), 0 or 1");
            c->opt_rec_lds = value; return 0;
        case OT_OPT_BLOCKS_PER_CU:
            if (value < 0 || value > 65536) return fail(OT_ERR_INVALID, "OT_OPT_BLOCKS_PER_CU out of range");
            c->opt_blocks_per_cu = value; return 0;
        default: return fail(OT_ERR_INVALID, "unknown option");
    }
}

}  // extern "C"

template <class T>
static int bench_stream(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segments* out, int32_t* seg_count) {
    if (!c) return fail(OT_ERR_INVALID, "ctx is NULL");
    int rc = check_rays(rays, "rays");
    if (rc) return rc;
    rc = check_segs(out);
    if (rc) return rc;
    if (n < 1 || K < 1 || !seg_count || n >= (int64_t)1 << 31) return fail(OT_ERR_INVALID, "bad n / K / seg_count");
    HIP_TRY(hipSetDevice(c->device));
    const int block = 256;
    const int64_t need = (n + block - 1) / block, cap = (int64_t)c->n_cus * (c->opt_blocks_per_cu > 0 ? c->opt_blocks_per_cu : 256);  // the fused kernel's grid rule
    const int grid = (int)(need < cap ? need : cap);
    rc = timing_begin(c);
    if (rc) return rc;
    if (c->opt_nt)
        hipLaunchKernelGGL((k_stream_ceiling<T, true>), dim3(grid), dim3(block), 0, c->stream, view<T>(rays), n, K, view<T>(out), seg_count, pair_ok<T>(c, out, n));
    else
        hipLaunchKernelGGL((k_stream_ceiling<T, false>), dim3(grid), dim3(block), 0, c->stream, view<T>(rays), n, K, view<T>(out), seg_count, pair_ok<T>(c, out, n));
    HIP_TRY(hipGetLastError());
    return timing_end(c);
}

extern "C" {

int ot_bench_stream_f64(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segments* out, int32_t* seg_count) {
    return bench_stream<double>(c, rays, n, K, out, seg_count);
}
int ot_bench_stream_f32(ot_ctx* c, const ot_rays* rays, int64_t n, int32_t K, const ot_segments* out, int32_t* seg_count) {
    return bench_stream<float>(c, rays, n, K, out, seg_count);
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
