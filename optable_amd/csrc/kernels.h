// kernels.h — the gfx950 __global__ kernels of liboptable_hip.so and the typed views they take.
// Device math (nearest hit, shapes, interactions) is in trace_core.h; the C-ABI, validation and launch logic in
// optable_hip.hip.
//   k_trace_fused<T>     one lane per ray, every segment of the ray in one launch (non-branching scenes, light
//                        scene tables).  HBM-bound: SoA streams, 64 consecutive elements per wave instruction.
//   k_trace_rolling<T>   the same trace for heavy scenes: persistent waves, each with its own list of live rays that
//                        is compacted every segment and refilled from a device-wide queue.
//   k_trace_pool<float>  heavy scenes whose rays all run through the same sequence of surfaces, append layout: the live
//                        rays of a workgroup in one pool of 64-ray blocks in LDS, shared by its sixteen waves.
//   k_stream_ceiling<T>  the fused kernel's streams with no tracing (roofline companion).
//   k_gen_pass<T>        one breadth-first generation of branching ray trees (optical_table.py:115-134) in two
//                        streaming passes: count (rank within the tree, trace) -> scan of wave totals -> emit (trace
//                        again, ordered slots); k_gen_probe / k_gen_rank / k_gen_counts keep interact-count gates
//                        FIFO-exact; k_gen_totals publishes the generation's totals between the two passes.
//   k_mon_*              Monitor.record over a segment stream (monitor.py:183-193).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <type_traits>

#include "trace_core.h"

using namespace ot;

// ------------------------------------------------------------------------------------------
// typed views of the C structs
template <class T> struct RaysT {
    const T *ox, *oy, *oz, *dx, *dy, *dz, *wl, *qr, *qi, *I, *n, *pl;
    const int32_t *id, *flags;
    const T* len;
};
template <class T> struct RaysOutT {
    T *ox, *oy, *oz, *dx, *dy, *dz, *wl, *qr, *qi, *I, *n, *pl;
    int32_t *id, *flags;
};
template <class T> struct SegsT {
    T *ox, *oy, *oz, *dx, *dy, *dz, *len, *I, *qr, *qi, *n, *pl;
    int32_t *ray, *surface;
};
template <class T> static RaysT<T> view(const ot_rays* r) {
    return {(const T*)r->ox, (const T*)r->oy, (const T*)r->oz, (const T*)r->dx, (const T*)r->dy, (const T*)r->dz,
            (const T*)r->wavelength, (const T*)r->q_re, (const T*)r->q_im, (const T*)r->intensity, (const T*)r->n,
            (const T*)r->pathlength, r->id, r->flags, (const T*)r->length};
}
template <class T> static RaysOutT<T> view_out(const ot_rays* r) {
    return {(T*)r->ox, (T*)r->oy, (T*)r->oz, (T*)r->dx, (T*)r->dy, (T*)r->dz, (T*)r->wavelength, (T*)r->q_re,
            (T*)r->q_im, (T*)r->intensity, (T*)r->n, (T*)r->pathlength, r->id, r->flags};
}
template <class T> static SegsT<T> view(const ot_segments* s) {
    return {(T*)s->ox, (T*)s->oy, (T*)s->oz, (T*)s->dx, (T*)s->dy, (T*)s->dz, (T*)s->length, (T*)s->intensity,
            (T*)s->q_re, (T*)s->q_im, (T*)s->n, (T*)s->pathlength, s->ray, s->surface};
}

// ------------------------------------------------------------------------------------------
// scene blob: [DNode<T> x n_phys][DMat<T> x n_mats][T x n_aux][int32 x 4 n_runs], staged into LDS word by word.
// n_nodes counts the caller's (virtual) nodes, n_phys the records kept after instanced runs were folded (trace_core.h NodeRef).
struct SceneBlob {
    const uint32_t* words;
    int32_t n_words, n_nodes, n_phys, n_mats, root, cache_mat, root_pack, n_runs, runs_word;
};

template <class T> __device__ __forceinline__ Scene<T> bind_scene(const uint32_t* base, const SceneBlob& b, T unit) {
    Scene<T> sc;
    sc.nodes = reinterpret_cast<const DNode<T>*>(base);
    sc.mats = reinterpret_cast<const DMat<T>*>(sc.nodes + b.n_phys);
    sc.aux = reinterpret_cast<const T*>(sc.mats + b.n_mats);
    sc.runs = reinterpret_cast<const int32_t*>(base + b.runs_word);
    sc.n_nodes = b.n_nodes;
    sc.n_runs = b.n_runs;
    for (int k = 0; k < 4; ++k) sc.run0[k] = b.n_runs > 0 ? __builtin_amdgcn_readfirstlane(sc.runs[k]) : 0;
    sc.n_mats = b.n_mats;
    sc.cache_mat = b.cache_mat;
    sc.root = b.root;
    sc.root_pack = b.root_pack;
    sc.unit = unit;
    return sc;
}

// Segment records are written once and never re-read by the trace: NT = true marks the stores
// non-temporal so they stream past L2 / Infinity Cache instead of evicting the scene and inputs.
template <bool NT, class V> __device__ __forceinline__ void st(V* p, V v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <class T, bool NT = false>
__device__ __forceinline__ void store_segment(const SegsT<T>& out, int64_t slot, const RayState<T>& r, T len, int32_t tree,
                                              int32_t surface) {
    st<NT>(out.ox + slot, r.ox); st<NT>(out.oy + slot, r.oy); st<NT>(out.oz + slot, r.oz);
    st<NT>(out.dx + slot, r.dx); st<NT>(out.dy + slot, r.dy); st<NT>(out.dz + slot, r.dz);
    st<NT>(out.len + slot, len); st<NT>(out.I + slot, r.I);
    st<NT>(out.qr + slot, r.qr); st<NT>(out.qi + slot, r.qi);
    st<NT>(out.n + slot, r.n); st<NT>(out.pl + slot, r.pl);
    st<NT>(out.ray + slot, tree); st<NT>(out.surface + slot, surface);
}

// Append layout (ot_trace_append_*): ONE allocation of 14 planes of `cap` slots — the 12 real fields in ot_segments order,
// then int32 ray[cap], int32 surface[cap] — instead of 14 independent arrays: two scalar registers instead of 28 live
// across the pass loop of a kernel that has 102 of them.
template <class T> struct SegPlanes {
    uint8_t* base;
    int64_t cap;
};
// Every plane is addressed as (wave-uniform plane base) + (one 32-bit lane offset): the store takes its base from a
// scalar register pair and the lane offset from ONE vector register computed once per record, instead of a 64-bit vector
// add per field (the host keeps capacity below 2^30 slots so that the byte offset fits 32 bits).
template <class T, bool NT = false>
__device__ __forceinline__ void store_segment(const SegPlanes<T>& out, int64_t slot, const RayState<T>& r, T len, int32_t tree,
                                              int32_t surface) {
#ifdef OT_EXP_NOSTORE  // measurement only: what the record stores cost a kernel (never in the product library)
    if (slot != 0x7fffffffffffll) return;
#endif
    const uint32_t s = (uint32_t)slot;
    uint8_t* b = out.base;
    asm volatile("" : "+s"(b));  // the fourteen plane bases are formed HERE, two scalar adds each, not kept in 28 registers across the pass loop
    T* p = reinterpret_cast<T*>(b);
    const int64_t cap = out.cap;
    st<NT>(p + s, r.ox); p += cap; st<NT>(p + s, r.oy); p += cap; st<NT>(p + s, r.oz); p += cap;
    st<NT>(p + s, r.dx); p += cap; st<NT>(p + s, r.dy); p += cap; st<NT>(p + s, r.dz); p += cap;
    st<NT>(p + s, len); p += cap; st<NT>(p + s, r.I); p += cap;
    st<NT>(p + s, r.qr); p += cap; st<NT>(p + s, r.qi); p += cap;
    st<NT>(p + s, r.n); p += cap; st<NT>(p + s, r.pl); p += cap;
    int32_t* q = reinterpret_cast<int32_t*>(p);
    st<NT>(q + s, tree); st<NT>(q + cap + s, surface);
}
template <class T> __device__ __forceinline__ int32_t* ray_plane(const SegPlanes<T>& out) { return reinterpret_cast<int32_t*>(reinterpret_cast<T*>(out.base) + 12 * out.cap); }

// Tiled layout (ot_trace_tiled_*): slot s lives in tile s / 64 at lane s % 64; a tile holds the 14 fields of 64 consecutive
// slots field by field — 12 x 64 reals, then int32 ray[64], int32 surface[64] — so the 64 lanes of a wave write ONE
// contiguous 6656-byte (fp32: 3584) block per segment instead of 14 runs in 14 arrays that lie n_rays * K elements apart.
// Same bytes, fewer open DRAM pages per wave: the streams of cfg 2 (1 record in, 5 out per ray) run at 5.68 instead of
// 5.22 TB/s that way (tools/stream_layouts.hip).
template <class T> struct SegTiles {
    uint8_t* base;
    static constexpr int64_t TILE_BYTES = 64 * (12 * (int64_t)sizeof(T) + 8);
};
template <class T, bool NT = false>
__device__ __forceinline__ void store_segment(const SegTiles<T>& out, int64_t slot, const RayState<T>& r, T len, int32_t tree,
                                              int32_t surface) {
    uint8_t* tb = out.base + (slot >> 6) * SegTiles<T>::TILE_BYTES;
    T* p = reinterpret_cast<T*>(tb) + (slot & 63);
    st<NT>(p, r.ox); st<NT>(p + 64, r.oy); st<NT>(p + 128, r.oz);
    st<NT>(p + 192, r.dx); st<NT>(p + 256, r.dy); st<NT>(p + 320, r.dz);
    st<NT>(p + 384, len); st<NT>(p + 448, r.I);
    st<NT>(p + 512, r.qr); st<NT>(p + 576, r.qi);
    st<NT>(p + 640, r.n); st<NT>(p + 704, r.pl);
    int32_t* q = reinterpret_cast<int32_t*>(tb + 768 * sizeof(T)) + (slot & 63);
    st<NT>(q, tree); st<NT>(q + 64, surface);
}

// Paired stores: lanes 2j and 2j+1 hold records for two ADJACENT slots.  Instead of fourteen stores of one element
// per lane, the pair takes the fields two at a time: after one DPP exchange the even lane holds both lanes' values of
// field A and writes them as ONE 16-byte store (fp64; 8 bytes in fp32), the odd lane does the same for field B.
// Half as many store instructions, each twice as wide: the widest coalesced form a structure-of-arrays layout allows
// (MI355X_MICROARCH.md: 16 bytes per lane is what streaming stores want).  Needs an even slot on the even lane, both
// lanes storing, and arrays aligned to the vector: the caller checks and falls back to store_segment otherwise.
__device__ __forceinline__ int dpp_xor1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }  // quad_perm [1,0,3,2]
__device__ __forceinline__ float xchg1(float v) { return __int_as_float(dpp_xor1(__float_as_int(v))); }
__device__ __forceinline__ double xchg1(double v) {
    return __hiloint2double(dpp_xor1(__double2hiint(v)), dpp_xor1(__double2loint(v)));
}
__device__ __forceinline__ int32_t xchg1(int32_t v) { return dpp_xor1(v); }
template <bool NT, class V>
__device__ __forceinline__ void store_pair(V* A, V* B, int64_t slot, V a, V b, bool odd) {
    typedef V vec2 __attribute__((ext_vector_type(2)));
    const V got = xchg1(odd ? a : b);  // what the partner does not store itself
    vec2 v;
    v.x = odd ? got : a;
    v.y = odd ? b : got;
    vec2* p = reinterpret_cast<vec2*>(odd ? B + (slot - 1) : A + slot);
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
// ... the same inside a tile: both lanes of a pair lie in one tile (64 is even), the planes are 64 elements apart
template <class T, bool NT>
__device__ __forceinline__ void store_segment_paired(const SegTiles<T>& out, int64_t slot, const RayState<T>& r, T len, int32_t tree,
                                                     int32_t surface, bool odd) {
    uint8_t* tb = out.base + (slot >> 6) * SegTiles<T>::TILE_BYTES;
    T* p = reinterpret_cast<T*>(tb);
    const int64_t l = slot & 63;
    store_pair<NT>(p, p + 64, l, r.ox, r.oy, odd);
    store_pair<NT>(p + 128, p + 192, l, r.oz, r.dx, odd);
    store_pair<NT>(p + 256, p + 320, l, r.dy, r.dz, odd);
    store_pair<NT>(p + 384, p + 448, l, len, r.I, odd);
    store_pair<NT>(p + 512, p + 576, l, r.qr, r.qi, odd);
    store_pair<NT>(p + 640, p + 704, l, r.n, r.pl, odd);
    int32_t* q = reinterpret_cast<int32_t*>(tb + 768 * sizeof(T));
    store_pair<NT>(q, q + 64, l, tree, surface, odd);
}
template <class T, bool NT>
__device__ __forceinline__ void store_segment_paired(const SegsT<T>& out, int64_t slot, const RayState<T>& r, T len, int32_t tree,
                                                     int32_t surface, bool odd) {
    store_pair<NT>(out.ox, out.oy, slot, r.ox, r.oy, odd);
    store_pair<NT>(out.oz, out.dx, slot, r.oz, r.dx, odd);
    store_pair<NT>(out.dy, out.dz, slot, r.dy, r.dz, odd);
    store_pair<NT>(out.len, out.I, slot, len, r.I, odd);
    store_pair<NT>(out.qr, out.qi, slot, r.qr, r.qi, odd);
    store_pair<NT>(out.n, out.pl, slot, r.n, r.pl, odd);
    store_pair<NT>(out.ray, out.surface, slot, tree, surface, odd);
}

// set bits of a wave-uniform mask below this lane (v_mbcnt: the mask comes from scalar registers, no per-lane lane-mask constant to keep)
__device__ __forceinline__ int rank_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// The caller's ray records are read exactly once per trace: non-temporal loads keep them from displacing what is
// re-read (per-wave records, scene image in L2 mode).  Worth 1 % on cfg 4 (11.79 -> 11.67 ms, interleaved A/B), nothing on cfg 3 / 5.
template <class V> __device__ __forceinline__ V ld_once(const V* p) { return __builtin_nontemporal_load(p); }
template <class T> __device__ __forceinline__ RayState<T> load_ray(const RaysT<T>& in, int64_t i, int32_t flags) {
    RayState<T> r;
    r.ox = ld_once(in.ox + i); r.oy = ld_once(in.oy + i); r.oz = ld_once(in.oz + i);
    r.dx = ld_once(in.dx + i); r.dy = ld_once(in.dy + i); r.dz = ld_once(in.dz + i);
    r.wl = ld_once(in.wl + i); r.qr = ld_once(in.qr + i); r.qi = ld_once(in.qi + i);
    r.I = ld_once(in.I + i); r.n = ld_once(in.n + i); r.pl = ld_once(in.pl + i);
    r.len = in.len ? in.len[i] : Num<T>::inf();
    r.has_q = (flags & OT_RAY_HAS_Q) != 0;
    r.last = (int32_t)((uint32_t)flags >> 8) - 1;  // bits 8..31: node the ray was emitted on, plus one (generation buffers; 0 for a caller's ray)
    return r;
}

// ------------------------------------------------------------------------------------------
// k_trace_fused: the hot kernel
// OUT: SegsT<T> (the [k][ray] slots of ot_trace_*: 14 arrays) or SegTiles<T> (the same slots in 64-slot tiles, ot_trace_tiled_*)
template <class T, uint32_t F, bool SCENE_IN_LDS, int MINW, bool NT, class OUT>
__global__ __launch_bounds__(256, MINW) void k_trace_fused(SceneBlob blob, T unit, RaysT<T> in, int64_t n, int32_t K, OUT out,
                                                     int32_t* __restrict__ seg_count, int32_t* counts, int32_t n_classes, int32_t pair) {
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t* base = blob.words;
    if (SCENE_IN_LDS) {
        for (int w = threadIdx.x; w < blob.n_words; w += blockDim.x) lds[w] = blob.words[w];
        __syncthreads();
        base = lds;
    }
    const Scene<T> sc = bind_scene<T>(base, blob, unit);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += stride) {
        const int64_t i = i0 + threadIdx.x;
        bool active = i < n;
        RayState<T> r = {};
        int32_t cls = 0, used = 0;
        MatCache<T> mc = {T(1)};
        if (active) {
            const int32_t fl = in.flags[i];
            r = load_ray(in, i, fl);
            if ((uint32_t)r.last >= (uint32_t)sc.n_nodes) r.last = -1;  // bits 8.. of a caller's flags that name no node of this scene
            if constexpr (F & F_REFRACT) mc = make_matcache<T, F>(sc, r.wl);
            cls = in.id[i];
            if (fl & OT_RAY_DEAD) {  // optical_component.py:349: a dead ray hits nothing and is returned as is
                store_segment<T, NT>(out, i, r, r.len, (int32_t)i, -2);
                used = 1;
                active = false;
            }
        }
        for (int32_t k = 0; k < K; ++k) {  // wave-uniform trip count: lanes never leave the loop alone
            if (!__any(active)) break;
            const GateCtx gate = {counts, n_classes, cls, nullptr, nullptr, 0, 0};
            const Hit<T> h = nearest_hit<T, F, GATE_PLAIN>(sc, r, active, gate);
            // segment record: pairs of lanes that both store write two fields per 16-byte store (store_segment_paired)
            const unsigned long long storing = __ballot(active);
            const bool both = pair && ((storing >> (threadIdx.x & 62)) & 3ull) == 3ull;
            if (both) {
                const bool hit = h.node >= 0;
                store_segment_paired<T, NT>(out, (int64_t)k * n + i, r, hit ? h.t : r.len, (int32_t)i, hit ? leaf_id_of<T, F>(sc, h.node) : -1,
                                            (threadIdx.x & 1) != 0);
            }
            if (active) {
                const int64_t slot = (int64_t)k * n + i;
                used = k + 1;
                if (h.node < 0) {  // escaped: archived unchanged (optical_table.py:132-134)
                    if (!both) store_segment<T, NT>(out, slot, r, r.len, (int32_t)i, -1);
                    active = false;
                } else {
                    if (!both) store_segment<T, NT>(out, slot, r, h.t, (int32_t)i, leaf_id_of<T, F>(sc, h.node));
                    RayState<T> child;
                    const int nk = interact<T, F, 1>(sc, r, h, &child, mc);
                    if (nk == 1) r = child;
                    else {
                        active = false;
                        if (nk > 1) used = -(k + 1);  // the tree branches here: not representable in [k][ray] slots
                    }
                }
            }
        }
        if (i < n) seg_count[i] = used;
    }
}

// ------------------------------------------------------------------------------------------
// k_trace_trees: whole ray TREES, a lane per tree (round 4).  The breadth-first loop of the reference (optical_table.py:
// 115-147: pop the oldest ray, archive it, push its children, stop after max_trace_num rays) runs per lane with the FIFO
// on the chip; only segment records leave it — the generation kernels write every child to HBM and read it back (271 of
// the 375-484 bytes they move per processed ray).  A tree that processes at most `cap` rays never needs more than
// ceil(cap / 2) queued rays: after j rays the queue holds at most j + 1, and only its first cap - j entries can still be
// processed; children that would queue up behind that are not stored.
// The queue of a lane is two rings, [entry][field][lane] each (11 reals + the node the ray starts on), so a wave's pushes
// and pops are conflict-free / coalesced whatever the lanes' positions: QL entries in LDS hold the FRONT of the queue, QG
// entries in a per-wave global scratch (L2-resident: written and read back by the same lane within microseconds) the rest.
// A push goes to LDS while nothing is queued in the scratch and LDS has room, else to the scratch; a pop takes LDS first.
// Short queues — a chain of partial reflections (cfg 4 with R = 0.2: three rays at most) — hardly leave the LDS; three entries
// per lane leave room for 8 waves per CU in double precision, two for 12, which is what the kernel's speed hangs on (one wave
// per SIMD issues an instruction every ~4.5 cycles: 5.8-6.2 ms on cfg 4 R = 0.2 with six entries in LDS; three entries: 4.2-4.3;
// two: 4.0-4.1; with claims of 2048+ slots instead of 512, below: 3.6-3.9).  Every child is queued the moment the interaction has
// formed it (SINK, trace_core.h interact) and the next ray is popped behind them: both children of a hit live in registers at
// once were 40 of the kernel's 200.  The scratch ring alone holds a whole queue (pushes keep going there while the LDS entries in
// front of them drain).  QL + QG < ceil(cap / 2) is allowed (large caps, small trees): a tree whose rings overflow reports
// -(segments so far) and the caller takes the generation path.
// Output: the [k][tree] slots of ot_trace_* — slot k * n + i is the k-th ray of tree i in FIFO order, which IS the
// reference's order; seg_count[i] = rays processed (== cap: the cap cut the tree short or the tree ended exactly there,
// as `budget <= 0` on the generation path).  Count-limited leaves: one column of the counts table per tree (rays that share
// an id are the caller's successive launches, as for ot_trace_*).
struct AppendCtl {
    unsigned long long* cursor;  // slots claimed so far (device); the caller reads it back as *n_slots
    int64_t capacity;
    int32_t chunk;               // slots per claim, a multiple of 64
};
template <class T> constexpr int tree_entry_bytes() { return 64 * (11 * (int)sizeof(T) + 4); }
// OUT = SegsT<T>: the [k][tree] slots described above.  OUT = SegPlanes<T>: the append layout of ot_trace_append_* — the records of a
// step go to consecutive slots of the wave's current chunk (claimed from a device-wide cursor, `ac`), whatever trees and
// positions the lanes are on: dense, whole lines per field.  A tree stays with its lane and a wave's chunks are claimed in
// address order, so a stable sort by `ray` is the reference's order; the unused tail of a wave's last chunk is marked ray = -1.
// IMG (the all-features preset only has the other two): 1 = the whole scene image in LDS; 2 = the node and material records in LDS,
// the aux tables (poses of instanced runs, grids, polygons, series) read where the upload left them (global memory: L2); 0 = everything
// read from global memory.  Scenes whose image no LDS holds (the reference's largest example, examples/ripa_gen2_lensless.py: 7,689
// leaves, ONE ray reflected three thousand times) get their trees in one launch too, instead of a launch sequence per generation
// with three searches each (fixture g27: 324 ms -> see DESIGN 4.5a).
template <class T, uint32_t F, int MINW, class OUT, int IMG = 1>
__global__ __launch_bounds__(256, MINW) void k_trace_trees(SceneBlob blob, T unit, RaysT<T> in, int64_t n, int32_t cap, int32_t QL, int32_t QG,
                                                           uint8_t* __restrict__ scratch, OUT out, AppendCtl ac, int32_t* __restrict__ seg_count,
                                                           int32_t* counts, int32_t n_classes, int32_t refill_at, int32_t flat_cap) {
    constexpr bool APPEND = std::is_same<OUT, SegPlanes<T>>::value;
    static_assert(IMG == 1 || (F & F_FLAT) == 0, "the pair queue reads the image from LDS");
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t* base = blob.words;
    size_t img_bytes = 0;
    if constexpr (IMG != 0) {
        const int head_words = IMG == 1 ? blob.n_words : (int)(((size_t)blob.n_phys * sizeof(DNode<T>) + (size_t)blob.n_mats * sizeof(DMat<T>)) / 4);
        for (int w = threadIdx.x; w < head_words; w += blockDim.x) lds[w] = blob.words[w];
        // IMG = 2: the run table too (four words per run) — node_ref walks it for every node index behind the first run, one
        // dependent read per run: from global memory that was a chain of L2 round trips per node visited
        const int run_words = IMG == 2 ? 4 * blob.n_runs : 0;
        for (int w = threadIdx.x; w < run_words; w += blockDim.x) lds[head_words + w] = blob.words[blob.runs_word + w];
        __syncthreads();
        base = lds;
        img_bytes = ((size_t)(head_words + run_words) * 4 + 15) & ~(size_t)15;
    }
    Scene<T> sc_bound = bind_scene<T>(IMG == 2 ? blob.words : base, blob, unit);
    if constexpr (IMG == 2) {  // the aux tables from global memory; records and the run table from LDS
        sc_bound.nodes = reinterpret_cast<const DNode<T>*>(lds);
        sc_bound.mats = reinterpret_cast<const DMat<T>*>(sc_bound.nodes + blob.n_phys);
        sc_bound.runs = reinterpret_cast<const int32_t*>(lds) + ((size_t)blob.n_phys * sizeof(DNode<T>) + (size_t)blob.n_mats * sizeof(DMat<T>)) / 4;
    }
    const Scene<T> sc = sc_bound;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr size_t EB = tree_entry_bytes<T>();
    uint8_t* const ring = reinterpret_cast<uint8_t*>(lds) + img_bytes + (size_t)wave * QL * EB;
    // F_FLAT (planar scenes under a top-level grid of leaves: cfg 3 with reflecting slabs): the search of a step is the wave-wide pair
    // queue of the heavy non-branching kernel (flat_grid_hit) — candidates of all 64 current rays, tested with full lanes — instead
    // of a grid walk per lane; its key table and queue sit behind the rings of all waves.
    FlatLds<T> flat = {nullptr, nullptr, nullptr, 0};
    FlatGrid<T> flat_grid = {};
    if constexpr ((F & F_FLAT) != 0) {
        const int per_wave = (FlatLds<T>::fixed_bytes + flat_cap * 2 + 15) & ~15;
        uint8_t* fb = reinterpret_cast<uint8_t*>(lds) + img_bytes + (size_t)(blockDim.x >> 6) * QL * EB + (size_t)wave * per_wave;
        flat.key = reinterpret_cast<unsigned long long*>(fb);
        if constexpr (sizeof(T) == 8) flat.node = reinterpret_cast<int32_t*>(fb + 64 * 8);
        flat.queue = reinterpret_cast<uint16_t*>(fb + FlatLds<T>::fixed_bytes);
        flat.queue_cap = flat_cap;
        for (int q = lane; q < flat_cap; q += 64) flat.queue[q] = 0;  // markers only; every round leaves it zeroed again
        flat_grid = flat_grid_header<T, true>(sc);
    }
#ifdef OT_STAMP
    unsigned long long tree_st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tree_st_last = 0;  // (the diagnostic build's stamps of flat_grid_hit: unused here)
#endif
    // The scratch ring is LANE-major: an entry of a lane is one contiguous record (12 words: 48 bytes in single precision, 96 in
    // double), written and read with 16-byte accesses — three or six instructions per ray instead of twelve, and whole records
    // per line when only some lanes of a wave push (field-major, a push of twenty lanes dirtied twenty-four 128-byte lines for 960
    // bytes: cfg 3 with reflecting slabs moved 7.6 x its algorithmic bytes).
    constexpr size_t GEB = 64 * (sizeof(T) == 8 ? 96 : 48);
    uint8_t* const gring = scratch + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * (size_t)QG * GEB;
    auto real_at = [&](uint8_t* base, int e, int f) -> T* { return reinterpret_cast<T*>(base + (size_t)e * EB) + f * 64 + lane; };
    auto int_at = [&](uint8_t* base, int e) -> int32_t* { return reinterpret_cast<int32_t*>(base + (size_t)e * EB + 64 * 11 * sizeof(T)) + lane; };
    auto put = [&](uint8_t* base, int e, const RayState<T>& c) {
        *real_at(base, e, 0) = c.ox; *real_at(base, e, 1) = c.oy; *real_at(base, e, 2) = c.oz;
        *real_at(base, e, 3) = c.dx; *real_at(base, e, 4) = c.dy; *real_at(base, e, 5) = c.dz;
        *real_at(base, e, 6) = c.qr; *real_at(base, e, 7) = c.qi; *real_at(base, e, 8) = c.I;
        *real_at(base, e, 9) = c.n; *real_at(base, e, 10) = c.pl;
        *int_at(base, e) = c.last;
    };
    auto put_g = [&](int e, const RayState<T>& c) {
        uint8_t* rec = gring + (size_t)e * GEB + (size_t)lane * (sizeof(T) == 8 ? 96 : 48);
        if constexpr (sizeof(T) == 4) {
            float4* q = reinterpret_cast<float4*>(rec);
            q[0] = make_float4(c.ox, c.oy, c.oz, c.dx);
            q[1] = make_float4(c.dy, c.dz, c.qr, c.qi);
            q[2] = make_float4(c.I, c.n, c.pl, __int_as_float(c.last));
        } else {
            double2* q = reinterpret_cast<double2*>(rec);
            q[0] = make_double2(c.ox, c.oy); q[1] = make_double2(c.oz, c.dx); q[2] = make_double2(c.dy, c.dz);
            q[3] = make_double2(c.qr, c.qi); q[4] = make_double2(c.I, c.n);
            q[5] = make_double2(c.pl, __longlong_as_double((long long)c.last));
        }
    };
    auto get_g = [&](int e, RayState<T>& r) {
        const uint8_t* rec = gring + (size_t)e * GEB + (size_t)lane * (sizeof(T) == 8 ? 96 : 48);
        if constexpr (sizeof(T) == 4) {
            const float4* q = reinterpret_cast<const float4*>(rec);
            const float4 a = q[0], b = q[1], c = q[2];
            r.ox = a.x; r.oy = a.y; r.oz = a.z; r.dx = a.w; r.dy = b.x; r.dz = b.y; r.qr = b.z; r.qi = b.w;
            r.I = c.x; r.n = c.y; r.pl = c.z; r.last = __float_as_int(c.w);
        } else {
            const double2* q = reinterpret_cast<const double2*>(rec);
            const double2 a = q[0], b = q[1], c = q[2], d = q[3], e2 = q[4], f = q[5];
            r.ox = a.x; r.oy = a.y; r.oz = b.x; r.dx = b.y; r.dy = c.x; r.dz = c.y; r.qr = d.x; r.qi = d.y;
            r.I = e2.x; r.n = e2.y; r.pl = f.x; r.last = (int32_t)__double_as_longlong(f.y);
        }
    };
    auto get = [&](uint8_t* base, int e, RayState<T>& r) {
        r.ox = *real_at(base, e, 0); r.oy = *real_at(base, e, 1); r.oz = *real_at(base, e, 2);
        r.dx = *real_at(base, e, 3); r.dy = *real_at(base, e, 4); r.dz = *real_at(base, e, 5);
        r.qr = *real_at(base, e, 6); r.qi = *real_at(base, e, 7); r.I = *real_at(base, e, 8);
        r.n = *real_at(base, e, 9); r.pl = *real_at(base, e, 10);
        r.last = *int_at(base, e);
    };
    // The trees of this wave: a contiguous share of the batch.  A lane whose tree has ended takes the next tree of the share in
    // place (no atomics: the share is the wave's own), so a wave does not live as long as its longest tree with the other
    // lanes idle — trees of a batch can differ by the whole cap.  Refills wait until `refill_at` lanes are idle (or none works).
    // 16: the lanes are kept busy, at the price of a wave whose lanes sit at different depths of their trees — a step then mixes
    // rays that reflect, refract and leave.  64: a wave takes 64 trees at a time and stays in step.  Which is faster depends on
    // how uneven the trees are: 90 % single-ray trees 1.33 vs 2.13 ms, trees of 13-30 rays 2.63 vs 1.90 (OT_OPT_TREES_REFILL_AT;
    // Engine.trace_branching times both on its 1 % sample).
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t per_wave = ((n + n_waves - 1) / n_waves + 63) / 64 * 64;
    int64_t next = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * per_wave;  // wave-uniform
    const int64_t end = next + per_wave < n ? next + per_wave : n;
    bool active = false, overflow = false;
    RayState<T> r = {};
    int32_t i = 0, k = 0, cls = 0;  // the lane's tree; rays it has processed; its column of the interact-count table
    // the two rings in one register (QL, QG <= 255): LDS ring head | entries << 8 | scratch ring head << 16 | entries << 24
    uint32_t qs = 0;
    auto lhead = [&]() -> int { return (int)(qs & 255u); };
    auto llen = [&]() -> int { return (int)((qs >> 8) & 255u); };
    auto ghead = [&]() -> int { return (int)((qs >> 16) & 255u); };
    auto glen = [&]() -> int { return (int)(qs >> 24); };
    auto queued = [&]() -> int { return llen() + glen(); };
    MatCache<T> mc = {T(1)};
    int64_t chunk_pos = 0;  // append layout: next free slot of this wave's chunk, and how many are left in it
    int32_t chunk_left = 0;
    // where a lane's record of this step goes (append: every lane of the wave calls it, `writes` = the lane has a record)
    auto place = [&](bool writes, int64_t kn_plus_i) -> int64_t {
        if constexpr (!APPEND) return kn_plus_i;
        else {
            const unsigned long long writers = __ballot(writes);
            const int need = __popcll(writers), rank = rank_below(writers);
            int64_t fresh_pos = 0;
            if (need > chunk_left) {  // wave-uniform: claim the next chunk; the step may straddle the two
                const unsigned long long c0 = lane == 0 ? atomicAdd(ac.cursor, (unsigned long long)ac.chunk) : 0ull;
                fresh_pos = (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(c0 >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(c0 & 0xffffffffull)));
            }
            const int64_t slot = rank < chunk_left ? chunk_pos + rank : fresh_pos + (rank - chunk_left);
            if (need > chunk_left) { chunk_pos = fresh_pos + (need - chunk_left); chunk_left = ac.chunk - (need - chunk_left); }
            else { chunk_pos += need; chunk_left -= need; }
            return slot < ac.capacity ? slot : -1;  // an output that is too small loses records, never writes outside (the cursor tells)
        }
    };
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (next < end && (__popcll(idle) >= refill_at || idle == ~0ull)) {
            const int64_t cand = next + rank_below(idle);
            bool dead = false;
            if (!active && cand < end) {
                i = (int32_t)cand;
                const int32_t fl = in.flags[i];
                r = load_ray(in, i, fl);
                if ((uint32_t)r.last >= (uint32_t)sc.n_nodes) r.last = -1;
                if constexpr (F & F_REFRACT) mc = make_matcache<T, F>(sc, r.wl);
                cls = in.id[i];
                k = 0; qs = 0; overflow = false;
                active = true;
                dead = (fl & OT_RAY_DEAD) != 0;
            }
            if (__any(dead)) {  // optical_component.py:349: a dead ray hits nothing and is returned as is
                const int64_t slot = place(dead, i);
                if (dead) {
                    if (slot >= 0) store_segment<T, false>(out, slot, r, r.len, i, -2);
                    seg_count[i] = 1;
                    active = false;
                }
            }
            next += __popcll(idle);
        }
        if (!__any(active)) {
            if (next >= end) break;
            continue;
        }
        // count-limited leaves (optical_component.py:140-149): a tree's rays meet them one after the other in FIFO order, as
        // the reference's loop does — exact as long as no other tree of the launch shares the column (the host API's rounds)
        const GateCtx gate = {counts, n_classes, cls, nullptr, nullptr, 0, 0};
        Hit<T> h;
        if constexpr ((F & F_FLAT) != 0) {
#ifdef OT_STAMP
            h = flat_grid_hit<T, F, GATE_PLAIN>(sc, flat_grid, r, active, gate, flat, lane, tree_st_acc, tree_st_last);
#else
            h = flat_grid_hit<T, F, GATE_PLAIN>(sc, flat_grid, r, active, gate, flat, lane);
#endif
        } else {
            h = nearest_hit<T, F, GATE_PLAIN>(sc, r, active, gate);
        }
        const int64_t slot = place(active, (int64_t)k * n + i);
        if (active) {
            ++k;
            const int32_t left = cap - k;  // rays this tree may still process
            // Every child goes to the queue the moment the interaction has formed it (the first is queued before the second is
            // computed: two children live together were 40 of the kernel's registers), and the next ray is popped behind them.
            auto push = [&](const RayState<T>& c) {
                if (queued() >= left) return;  // would never be processed (optical_table.py:138-144 drops it with the queue)
                if (glen() == 0 && llen() < QL) {
                    int e = lhead() + llen();
                    if (e >= QL) e -= QL;
                    put(ring, e, c);
                    qs += 1u << 8;
                } else if (glen() < QG) {
                    int e = ghead() + glen();
                    if (e >= QG) e -= QG;
                    put_g(e, c);
                    qs += 1u << 24;
                } else {
                    overflow = true;
                }
            };
            const T wl = r.wl;
            const int32_t has_q = r.has_q;
            if (h.node < 0) {
                if (slot >= 0) store_segment<T, false>(out, slot, r, r.len, i, -1);
            } else {
                if (slot >= 0) store_segment<T, false>(out, slot, r, h.t, i, leaf_id_of<T, F>(sc, h.node));
                interact<T, F, 2, decltype(push)>(sc, r, h, nullptr, mc, &push);
            }
            if ((qs & 0xff00ff00u) != 0 && left > 0 && !overflow) {
                if (llen() > 0) {
                    const int e = lhead();
                    get(ring, e, r);
                    qs = ((qs & ~255u) | (uint32_t)(e + 1 == QL ? 0 : e + 1)) - (1u << 8);
                } else {
                    const int e = ghead();
                    get_g(e, r);
                    qs = ((qs & ~(255u << 16)) | ((uint32_t)(e + 1 == QG ? 0 : e + 1) << 16)) - (1u << 24);
                    if (glen() == 0) qs &= ~(255u << 16);  // an empty ring starts over at its first entry: the scratch a lane touches is its longest queue, not QG
                }
                r.wl = wl; r.has_q = has_q; r.len = Num<T>::inf();
            } else {
                active = false;
            }
            if (overflow) active = false;
            if (!active) seg_count[i] = overflow ? -k : k;  // the tree is done (or its queue overflowed: the caller takes the generations)
        }
    }
    if constexpr (APPEND) {  // the unused tail of this wave's last chunk: holes
        int32_t* rp = ray_plane(out);
        for (int64_t s = chunk_pos + lane; s < chunk_pos + chunk_left; s += 64)
            if (s < ac.capacity) rp[s] = -1;
    }
}

// ------------------------------------------------------------------------------------------
// k_trace_rolling: heavy scenes (many nodes per segment: VALU- and latency-bound, uneven path lengths).
// A fixed chunk of rays per wave lives as long as its longest ray, so its late passes run with a handful of lanes, and
// a pass costs about the same whether 64 lanes work in it or 3.  Here every wave owns ONE list of up to CAP live rays: a
// ring in LDS (entry = ray index | segment index << 32).  A pass takes the 64 OLDEST entries and its survivors go to the
// tail; fresh rays come in tickets of 64 consecutive rays from a device-wide queue (one atomic per ticket).
//   mixed lists (scenes under a top-level grid: the rays of a wave are unrelated after the first bounce — cfg 3): a ticket
//     is drawn whenever 64 slots are free and traced at once, as a pass of its own (coalesced loads of the caller's arrays,
//     and a batch's rays usually start alike: a coherent pass); only its survivors enter the ring.  Every pass is full
//     until the queue is empty, rays of all generations share a pass.
//   generation-pure lists (scenes whose rays all run through the same sequence of surfaces — cfg 5): only an EMPTY list is
//     refilled, CAP rays at once, and worked off in rounds; a pass then tests one kind of surface (mixing cost cfg 5 16 %).
// Workgroups are persistent and their waves never synchronise after the scene image is staged: no wave waits for the
// slowest chunk of its workgroup.
// The records of the live rays (12 reals + one word of flags and start node, + the count class in scenes with limited
// surfaces) are kept BY LIST POSITION, [field][CAP] per wave: a pass reads 64 consecutive ring positions and writes its
// survivors to the tail (generation-pure lists: to the start of the round's range, all read already, so the compaction
// happens in place).  REC_LDS puts them in LDS next to the scene image — a pass then touches global memory only for a
// ray's first load and for the segment records it writes, nothing it has to wait for (gfx9 retires loads and stores
// through ONE in-order counter, so with the records in global memory every pass's loads queue behind the stores of
// the pass before); otherwise they live in a per-wave global scratch that stays in L2 between the pass that writes it
// and the pass that reads it (fp64, large images).
// Output, by the type of `out`:
//   SegsT<T>      the [k][ray] slots of ot_trace_*: segment k of ray i at k * n + i.  Survivors of different tickets are
//                 scattered over the late planes: 4-byte stores into lines whose other elements belong to dead rays
//                 (cfg 3: 6.3 GB written for 2.8 GB of records, and a store instruction that touches 64 lines).
//   SegPlanes<T>  the append layout of ot_trace_append_*: every wave claims chunks of `chunk` slots from a device-wide
//                 cursor (one atomic per chunk) and fills them pass by pass, 64 records = whole lines per field.  A ray
//                 stays with its wave and chunks are claimed in address order, so the records of one ray lie at
//                 increasing addresses: a stable sort by `ray` is the reference's order (optical_table.py:125-134), as for
//                 the breadth-first trace.  The unused tail of a wave's last chunk is marked ray = -1.
// Tried and dropped (cfg 3, fp32, 1e7 rays): a ring buffer with the next pass's records prefetched before this pass's
// stores (19 more live registers and per-lane source selects cost more than the hidden latency returns); a branch-free
// planar test in the per-lane cell loop (the early exits do pay there); a resumable grid walk that visits at most 1 / 2 /
// 4 cells per pass (every extra pass pays the pass's load -> trace -> store latency again); lists worked off in rounds
// with a remainder pass instead of the FIFO ring (44 instead of 62 lanes per pass).
static constexpr int REC_LDS_POSITIONS = 128;  // REC_LDS kernels: the first 128 positions of every wave's list keep their records in LDS
template <class T> struct WaveScratch {
    uint8_t* base;
    int64_t wave_bytes;  // bytes per wave: CAP * record bytes, rounded up
};
template <uint32_t F> constexpr int rec_int_words() { return (F & F_LIMIT) ? 2 : 1; }
// one kernel argument, loaded from the kernel-argument segment HERE (the asm keeps the load from being hoisted and kept live)
template <class V> __device__ __forceinline__ V karg(size_t offset) {
    typedef const __attribute__((address_space(4))) V* ArgPtr;
    ArgPtr p = (ArgPtr)((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offset);
    asm volatile("" : "+s"(p));
    return *p;
}

// Largest workgroup an instantiation may be launched with, and the workgroups per CU the compiler has to leave registers
// for.  The waves never synchronise after staging, so the workgroup size only decides how many waves share one image:
//   pair queue, records in LDS: LDS decides the occupancy — up to one 1024-thread workgroup per CU (128 registers, ~110 used);
//   pair queue, records in global memory: four 256-thread workgroups = 4 waves per SIMD (128 registers; it needs ~110 since the
//     grid header is kept in registers across passes), fp64 three;
//   curved-surface preset (cfg 5) in fp32: one 1024-thread workgroup = 4 waves per SIMD on one image (128 registers);
//   all features in fp64: 256 threads (255 registers); everything else 512 = 2 waves per SIMD.
template <class T, uint32_t F, bool REC_LDS> constexpr int rolling_threads() {
    if ((F & F_FLAT) != 0) return REC_LDS ? 1024 : 256;
    if (sizeof(T) == 8 && F == F_ALL) return 256;
    if (sizeof(T) == 4 && F == (F_AABB | F_REFRACT | F_CURVED | F_GRID)) return 1024;
    return 512;
}
template <class T, uint32_t F, bool REC_LDS> constexpr int rolling_minw() { return ((F & F_FLAT) != 0 && !REC_LDS) ? (sizeof(T) == 4 ? 4 : 3) : 1; }

template <class T, uint32_t F, bool SCENE_IN_LDS, bool NT, bool REC_LDS, class OUT>
__global__ __launch_bounds__((rolling_threads<T, F, REC_LDS>()), (rolling_minw<T, F, REC_LDS>())) void k_trace_rolling(
    SceneBlob blob, T unit, RaysT<T> in, int64_t n, int32_t K, OUT out, AppendCtl ac, int32_t* __restrict__ seg_count, int32_t* counts,
    int32_t n_classes, WaveScratch<T> ws, int32_t CAP, int32_t capl_arg, unsigned long long* queue, int32_t mix, int32_t flat_cap) {
    // list positions whose records live in LDS: a compile-time constant, so that the twelve field planes of a record are
    // immediate offsets of one ds_read / ds_write address instead of twelve scalar registers and an add each
    constexpr int CAPL = REC_LDS ? REC_LDS_POSITIONS : 0;
    (void)capl_arg;  // the host passes the same number (optable_hip.hip checks it against REC_LDS_POSITIONS)
    constexpr bool APPEND = std::is_same<OUT, SegPlanes<T>>::value;
    constexpr int W = (int)(sizeof(T) / 4), RI = rec_int_words<F>();
    // Where the records of the live rays are: list positions below CAPL in LDS (REC_LDS), the others in this wave's global
    // scratch.  Mixed lists with REC_LDS keep everything in LDS (CAPL == CAP; the pair-queue kernels are only ever mixed
    // and do not compile the overflow path); generation-pure lists keep the FRONT of the list there: they compact in place
    // towards position 0, so a long list (full passes: 58 lanes per pass at 512 entries, 45 at 128) spills its far end to
    // global memory only while it is young, and once it has thinned out to CAPL rays every pass is LDS only.
    constexpr bool OVERFLOW = !REC_LDS || (F & F_FLAT) == 0;
    // The caller's 15 array pointers are needed once per ray, for its first segment — and as a by-value kernel argument
    // they would sit in 30 of the 102 scalar registers for the whole pass loop (the kernel spills scalars into vector lanes
    // as it is).  They are read from the kernel-argument segment where they are used instead; `in` itself is never touched.
    // (The segment is laid out like a struct of the parameters in order, each at its natural alignment: LeadArgs below;
    // tools/kernel_resources.sh --args prints the offsets the code object records.)
    struct LeadArgs {  // the kernel's parameter list, in order: the layout of the kernel-argument segment
        SceneBlob blob; T unit; RaysT<T> in; int64_t n; int32_t K; OUT out; AppendCtl ac; int32_t* seg_count; int32_t* counts;
        int32_t n_classes; WaveScratch<T> ws; int32_t CAP; int32_t capl; unsigned long long* queue; int32_t mix; int32_t flat_cap;
    };
    // ... and so are the arguments a pass needs once or less: the ticket queue, the append cursor and its
    // bounds, seg_count.  A scalar that is live across the nearest-hit search is spilled into a vector lane before it
    // and read back after it (one VALU instruction each way, per pass); one that is loaded from the argument segment
    // when it is needed costs a scalar load.
#define OT_KARG(field) karg<decltype(LeadArgs::field)>(offsetof(LeadArgs, field))
    (void)ac; (void)seg_count; (void)queue;
    typedef const __attribute__((address_space(4))) RaysT<T>* RaysArgPtr;
    const RaysArgPtr in_arg = (RaysArgPtr)((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LeadArgs, in));
    (void)in;
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t* base = blob.words;
    uint32_t* lds_tail = lds;
    if (SCENE_IN_LDS) {
        for (int w = threadIdx.x; w < blob.n_words; w += blockDim.x) lds[w] = blob.words[w];
        base = lds;
        lds_tail = lds + ((blob.n_words + 3) & ~3);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    // wave-private list of CAP entries: mixed lists a ring of (ray index | segment index << 32); generation-pure lists the ray
    // indices alone (every ray of the list is on the same segment: the list is refilled only when it is empty)
    const int ring_bytes = CAP * (mix ? 8 : 4);
    unsigned long long* ring64 = reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(lds_tail) + wave * ring_bytes);
    uint32_t* ring32 = reinterpret_cast<uint32_t*>(ring64);
    uint8_t* lds_next = reinterpret_cast<uint8_t*>(lds_tail) + n_waves * ring_bytes;
    // F_FLAT: per-wave key table and pair queue of flat_grid_hit, behind the lists of all waves
    FlatLds<T> flat = {nullptr, nullptr, nullptr, 0};
    if constexpr ((F & F_FLAT) != 0) {
        const int per_wave = (FlatLds<T>::fixed_bytes + flat_cap * 2 + 15) & ~15;
        uint8_t* fb = lds_next + wave * per_wave;
        flat.key = reinterpret_cast<unsigned long long*>(fb);
        if constexpr (sizeof(T) == 8) flat.node = reinterpret_cast<int32_t*>(fb + 64 * 8);
        flat.queue = reinterpret_cast<uint16_t*>(fb + FlatLds<T>::fixed_bytes);
        flat.queue_cap = flat_cap;
        for (int q = lane; q < flat_cap; q += 64) flat.queue[q] = 0;  // markers only; every round leaves it zeroed again
        lds_next += n_waves * per_wave;
    }
    __syncthreads();  // the only workgroup barrier: the scene image is staged
    const Scene<T> sc = bind_scene<T>(base, blob, unit);
    FlatGrid<T> flat_grid = {};
    if constexpr ((F & F_FLAT) != 0) flat_grid = flat_grid_header<T>(sc);
    // ONE base pointer per wave; field f of ring position p is element p + f * CAP (twelve reals, then the integer words
    // behind them).  Separate base pointers per field cost scalar registers that the kernel does not have.
    const int64_t gw = (int64_t)blockIdx.x * n_waves + wave;
    T* const lrec = reinterpret_cast<T*>(lds_next) + (int64_t)wave * ((12 * W + RI) * CAPL) / W;  // REC_LDS: field stride CAPL
    int32_t* const lint = reinterpret_cast<int32_t*>(lrec + 12 * CAPL);
    const int CAPG = CAP - (REC_LDS ? CAPL : 0);                                                   // global part: field stride CAPG
    T* const grec = reinterpret_cast<T*>(ws.base + gw * ws.wave_bytes);
    int32_t* const gint = reinterpret_cast<int32_t*>(grec + 12 * CAPG);
    int32_t kround = 0;  // generation-pure lists: the segment index of every ray in the list
    const int M = mix ? CAP - 1 : -1;  // mixed lists: a ring, CAP is a power of two (host); generation-pure lists always start at position 0 and never wrap: any multiple of 64
    int head = 0, tail = 0, alive = 0, round_left = 0;  // wave-uniform: `alive` entries from ring position `head`; survivors and tickets go to `tail`
    bool exhausted = false;
    int64_t chunk_pos = 0;  // append layout: next free slot of this wave's chunk, and how many are left in it
    int32_t chunk_left = 0;
#ifdef OT_STAMP
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define OT_STAMP_AT(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long _t = __builtin_amdgcn_s_memtime(); st_acc[k] += _t - st_last; st_last = _t; } while (0)
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#define OT_FLAT_STAMP_ARGS , st_acc, st_last
#elif defined(OT_MARK)  // static census (tools/isa_census.py): the stamp sites as comments in the assembly
#define OT_STAMP_AT(k) asm volatile("; OT_MARK pass " #k)
#define OT_FLAT_STAMP_ARGS
#else
#define OT_STAMP_AT(k) do {} while (0)
#define OT_FLAT_STAMP_ARGS
#endif
    // (Tickets drawn one ahead, the atomic's round trip under a pass instead of in front of one: measured, cfg 3 2.63
    // against 2.59 ms, cfg 5 14.4 against 14.0 — the result register and the wait it drags into the pass cost more.)
    auto draw_ticket = [&]() -> unsigned long long {
        unsigned long long first = 0;
        if (lane == 0) first = atomicAdd(OT_KARG(queue), 64ull);
        return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(first & 0xffffffffull));
    };

    for (;;) {
        bool fresh = false;
        unsigned long long fresh_first = 0;
        int fresh_cnt = 0;
        if (mix) {
            if (!exhausted && alive + 64 <= CAP) {
                const unsigned long long first = draw_ticket();
                const unsigned long long n_now = (unsigned long long)n;
                if (first >= n_now) {
                    exhausted = true;
                } else {
                    fresh = true;
                    fresh_first = first;
                    fresh_cnt = (int)(n_now - first < 64ull ? n_now - first : 64ull);
                }
            }
        } else if (round_left == 0 && alive == 0) {
            while (!exhausted && alive + 64 <= CAP) {
                const unsigned long long first = draw_ticket();
                const unsigned long long n_now = (unsigned long long)n;
                if (first >= n_now) { exhausted = true; break; }
                const int cnt = (int)(n_now - first < 64ull ? n_now - first : 64ull);
                if (lane < cnt) ring32[tail + lane] = (uint32_t)first + (uint32_t)lane;  // segment index 0
                tail += cnt;
                alive += cnt;
            }
            kround = 0;
        }
        if (!fresh && alive == 0) break;  // queue and list are empty
        if (!fresh && round_left == 0) {
            round_left = alive;
            // generation-pure lists start every round at position 0 and compact IN PLACE (survivors go to positions
            // already read): the records a round touches are the shrinking prefix the last round wrote
            if (!mix) tail = head;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // One pass: a fresh ticket, or the 64 oldest entries of the ring (FIFO: no ray waits behind younger ones).
        const int avail = mix ? alive : round_left;
        const int take = fresh ? fresh_cnt : (avail < 64 ? avail : 64);
        {
            const int p = (head + lane) & M;
            const bool entry = lane < take;
            // (ray index, segment index): 32 bits each — n < 2^31 per launch
            int32_t i = (int32_t)(uint32_t)fresh_first + lane, k = 0;
            if (!fresh) {
                i = 0;
                k = kround;
                if (entry) {
                    if (mix) { const unsigned long long e = ring64[p]; i = (int32_t)(e & 0x7fffffffull); k = (int32_t)(e >> 32); }
                    else i = (int32_t)ring32[p];
                }
            }
            RayState<T> r = {};
            int32_t cls = 0, fl = 0;
            // A ray's record in two parts: what the nearest-hit search needs (origin, direction, length, start node) and
            // what only rides along to the interaction and the next record (q, intensity, index, path length, wavelength).
            // Kernels with their records in LDS fetch the second part AFTER the search: six registers less at the search's
            // peak (the curved-surface kernel has 128 at 16 waves per CU), for LDS reads that cost the same either way.
            constexpr bool LATE = REC_LDS;
            auto load_part = [&](const bool geom) {
                if (!entry) return;
                if (k == 0) {  // first segment: the caller's arrays (a ticket is 64 consecutive rays)
                    RaysArgPtr ip = in_arg;
                    asm volatile("" : "+s"(ip));  // opaque here: the pointers are loaded now, not hoisted out of the pass loop
                    RaysT<T> in_now;
                    {
                        static_assert(sizeof(RaysT<T>) == 15 * sizeof(uint64_t), "RaysT is fifteen pointers");
                        const __attribute__((address_space(4))) uint64_t* src = (const __attribute__((address_space(4))) uint64_t*)ip;
                        uint64_t words[15];
#pragma unroll
                        for (int w = 0; w < 15; ++w) words[w] = src[w];
                        __builtin_memcpy(&in_now, words, sizeof(in_now));
                    }
                    if (geom) {
                        fl = in_now.flags[i];
                        if constexpr ((F & F_LIMIT) != 0) cls = in_now.id[i];
                        r.ox = ld_once(in_now.ox + i); r.oy = ld_once(in_now.oy + i); r.oz = ld_once(in_now.oz + i);
                        r.dx = ld_once(in_now.dx + i); r.dy = ld_once(in_now.dy + i); r.dz = ld_once(in_now.dz + i);
                        r.len = in_now.len ? in_now.len[i] : Num<T>::inf();
                        r.has_q = (fl & OT_RAY_HAS_Q) != 0;
                        r.last = (int32_t)((uint32_t)fl >> 8) - 1;  // bits 8..31: node the ray was emitted on, plus one (generation buffers; 0 for a caller's ray)
                        if ((uint32_t)r.last >= (uint32_t)sc.n_nodes) r.last = -1;  // ... that name no node of this scene
                        fl &= 0xff;
                    } else {
                        r.wl = ld_once(in_now.wl + i); r.qr = ld_once(in_now.qr + i); r.qi = ld_once(in_now.qi + i);
                        r.I = ld_once(in_now.I + i); r.n = ld_once(in_now.n + i); r.pl = ld_once(in_now.pl + i);
                    }
                } else if (REC_LDS && (!OVERFLOW || head < CAPL)) {  // later ones: this wave's records, by list position (a pass lies in one region: wave-uniform branch)
                    if (geom) {
                        r.ox = lrec[p]; r.oy = lrec[p + CAPL]; r.oz = lrec[p + 2 * CAPL];
                        r.dx = lrec[p + 3 * CAPL]; r.dy = lrec[p + 4 * CAPL]; r.dz = lrec[p + 5 * CAPL];
                        const int32_t meta = lint[p];
                        if constexpr ((F & F_LIMIT) != 0) cls = lint[p + CAPL];
                        r.last = (meta >> 8) - 1;
                        r.has_q = meta & OT_RAY_HAS_Q;  // (a dead ray never gets a record: the flag word of a record is HAS_Q or nothing)
                        r.len = Num<T>::inf();
                    } else {
                        r.qr = lrec[p + 6 * CAPL]; r.qi = lrec[p + 7 * CAPL]; r.I = lrec[p + 8 * CAPL];
                        r.n = lrec[p + 9 * CAPL]; r.pl = lrec[p + 10 * CAPL]; r.wl = lrec[p + 11 * CAPL];
                    }
                } else if constexpr (OVERFLOW) {
                    const int g = p - (REC_LDS ? CAPL : 0);
                    if (geom) {
                        r.ox = grec[g]; r.oy = grec[g + CAPG]; r.oz = grec[g + 2 * CAPG];
                        r.dx = grec[g + 3 * CAPG]; r.dy = grec[g + 4 * CAPG]; r.dz = grec[g + 5 * CAPG];
                        const int32_t meta = gint[g];
                        if constexpr ((F & F_LIMIT) != 0) cls = gint[g + CAPG];
                        r.last = (meta >> 8) - 1;
                        r.has_q = meta & OT_RAY_HAS_Q;
                        r.len = Num<T>::inf();
                    } else {
                        r.qr = grec[g + 6 * CAPG]; r.qi = grec[g + 7 * CAPG]; r.I = grec[g + 8 * CAPG];
                        r.n = grec[g + 9 * CAPG]; r.pl = grec[g + 10 * CAPG]; r.wl = grec[g + 11 * CAPG];
                    }
                }
            };
            load_part(true);
            if constexpr (!LATE) load_part(false);
            OT_STAMP_AT(0);
            const bool active = entry && !(fl & OT_RAY_DEAD);  // optical_component.py:349: a dead ray is returned as it came
            const GateCtx gate = {counts, n_classes, cls, nullptr, nullptr, 0, 0};
            Hit<T> h;
            if constexpr ((F & F_FLAT) != 0) h = flat_grid_hit<T, F, GATE_PLAIN>(sc, flat_grid, r, active, gate, flat, lane OT_FLAT_STAMP_ARGS);
            else {
#ifdef OT_STAMP
                h = nearest_hit<T, F, GATE_PLAIN>(sc, r, active, gate, st_acc, &st_last);
#else
                h = nearest_hit<T, F, GATE_PLAIN>(sc, r, active, gate);
#endif
            }
            if constexpr (LATE) load_part(false);
            OT_STAMP_AT(1);
            // the segment record: every entry of the pass writes exactly one
            const bool hit = active && h.node >= 0;
            int64_t slot = APPEND ? 0 : (int64_t)k * n + (int64_t)i;
            bool room = true;
            if constexpr (APPEND) {
                const unsigned long long writers = __ballot(entry);
                const int need = __popcll(writers), rank = __popcll(writers & ((1ull << lane) - 1ull));
                int64_t fresh_pos = 0;
                if (need > chunk_left) {  // wave-uniform: claim the next chunk; the pass may straddle the two
                    const unsigned long long c0 = lane == 0 ? atomicAdd(OT_KARG(ac.cursor), (unsigned long long)OT_KARG(ac.chunk)) : 0ull;
                    fresh_pos = (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(c0 >> 32)) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(c0 & 0xffffffffull)));
                }
                slot = rank < chunk_left ? chunk_pos + rank : fresh_pos + (rank - chunk_left);
                if (need > chunk_left) { chunk_pos = fresh_pos + (need - chunk_left); chunk_left = OT_KARG(ac.chunk) - (need - chunk_left); }
                else { chunk_pos += need; chunk_left -= need; }
                room = slot < OT_KARG(ac.capacity);  // an output that is too small loses records, never writes outside (the cursor tells)
            }
            if constexpr (APPEND) {
                if (entry && room) store_segment<T, NT>(out, slot, r, hit ? h.t : r.len, (int32_t)i, hit ? leaf_id_of<T, F>(sc, h.node) : (active ? -1 : -2));
            } else {  // the fourteen [k][ray] array pointers: from the kernel-argument segment, like the caller's ray pointers
                typedef const __attribute__((address_space(4))) uint64_t* ArgWords;
                ArgWords src = (ArgWords)((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LeadArgs, out));
                asm volatile("" : "+s"(src));
                static_assert(sizeof(SegsT<T>) == 14 * sizeof(uint64_t), "SegsT is fourteen pointers");
                uint64_t words[14];
#pragma unroll
                for (int w = 0; w < 14; ++w) words[w] = src[w];
                SegsT<T> out_now;
                __builtin_memcpy(&out_now, words, sizeof(out_now));
                if (entry) store_segment<T, NT>(out_now, slot, r, hit ? h.t : r.len, (int32_t)i, hit ? leaf_id_of<T, F>(sc, h.node) : (active ? -1 : -2));
            }
            bool survive = false;
            RayState<T> child = {};
            if (entry) {
                int32_t used = k + 1;
                if (hit) {
                    MatCache<T> mc = {T(1)};
                    if constexpr (F & F_REFRACT) mc = make_matcache<T, F>(sc, r.wl);
                    const int nk = interact<T, F, 1>(sc, r, h, &child, mc);
                    if (nk == 1) survive = k + 1 < K;
                    else if (nk > 1) used = -(k + 1);  // the tree branches here: the caller re-traces it generation by generation
                }
                if (!survive) OT_KARG(seg_count)[i] = used;
            }
            const unsigned long long mk = __ballot(survive);
            if (survive) {
                // to the tail of the ring.  It may wrap onto the positions this pass has just read (at most `take` of
                // them: alive <= CAP and survivors <= take); every record load of the pass was issued before these stores
                // and accesses of one wave complete in issue order
                const int q = (tail + __popcll(mk & ((1ull << lane) - 1ull))) & M;
                if (mix) ring64[q] = ((unsigned long long)(uint32_t)(k + 1) << 32) | (unsigned long long)(uint32_t)i;
                else ring32[q] = (uint32_t)i;
                const int32_t meta = (r.has_q ? OT_RAY_HAS_Q : 0) | ((child.last + 1) << 8);
                if (REC_LDS && (!OVERFLOW || q < CAPL)) {
                    lrec[q] = child.ox; lrec[q + CAPL] = child.oy; lrec[q + 2 * CAPL] = child.oz;
                    lrec[q + 3 * CAPL] = child.dx; lrec[q + 4 * CAPL] = child.dy; lrec[q + 5 * CAPL] = child.dz;
                    lrec[q + 6 * CAPL] = child.qr; lrec[q + 7 * CAPL] = child.qi; lrec[q + 8 * CAPL] = child.I;
                    lrec[q + 9 * CAPL] = child.n; lrec[q + 10 * CAPL] = child.pl; lrec[q + 11 * CAPL] = r.wl;
                    lint[q] = meta;
                    if constexpr ((F & F_LIMIT) != 0) lint[q + CAPL] = cls;
                }
                if constexpr (OVERFLOW) {
                    if (!REC_LDS || q >= CAPL) {
                        const int g = q - (REC_LDS ? CAPL : 0);
                        grec[g] = child.ox; grec[g + CAPG] = child.oy; grec[g + 2 * CAPG] = child.oz;
                        grec[g + 3 * CAPG] = child.dx; grec[g + 4 * CAPG] = child.dy; grec[g + 5 * CAPG] = child.dz;
                        grec[g + 6 * CAPG] = child.qr; grec[g + 7 * CAPG] = child.qi; grec[g + 8 * CAPG] = child.I;
                        grec[g + 9 * CAPG] = child.n; grec[g + 10 * CAPG] = child.pl; grec[g + 11 * CAPG] = r.wl;
                        gint[g] = meta;
                        if constexpr ((F & F_LIMIT) != 0) gint[g + CAPG] = cls;
                    }
                }
            }
            tail = (tail + __popcll(mk)) & M;
            if (fresh) {
                alive += __popcll(mk);
            } else {
                head = (head + take) & M;
                alive += __popcll(mk) - take;
                round_left -= take;
            }
            if (!mix && round_left == 0) { head = (tail - alive) & M; ++kround; }  // the next round reads what this one wrote
            OT_STAMP_AT(2);
#ifdef OT_STAMP
            st_acc[4] += 1;
#endif
        }
        // the ring and the records written above are read by other lanes of this wave in a later pass: LDS and global
        // accesses of one wave complete in issue order, the fence only stops the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if constexpr (APPEND) {  // the unused tail of this wave's last chunk: holes
        int32_t* rp = ray_plane(out);
        for (int64_t s = chunk_pos + lane; s < chunk_pos + chunk_left; s += 64)
            if (s < OT_KARG(ac.capacity)) rp[s] = -1;
    }
#ifdef OT_STAMP
    if (lane == 0) for (int q = 0; q < 12; ++q) atomicAdd(&queue[8 + q], st_acc[q]);
#endif
}

// ------------------------------------------------------------------------------------------
// k_trace_refill — mixed scenes (everything under a top-level grid: the rays of a wave are unrelated after their first
// bounce — cfg 3) with the live rays IN REGISTERS.  A lane keeps its ray from segment to segment; when the ray ends the
// lane takes the next fresh ray of the wave's ticket IN PLACE (one atomic on the device-wide queue per TICKET rays, the
// fresh rays of a pass are consecutive: their loads coalesce).  No list, no compaction, no records in LDS or in global
// scratch: the only LDS a wave owns is the pair queue's key table and markers (1.5 KB), so the workgroups that fit a CU
// are decided by registers alone — the round-4 calibration (profiles/r04_issue_calibration.json) showed k_trace_rolling at
// 4 waves per SIMD issuing one vector instruction per 5.1 cycles where the pipe takes one per 2.3: latency-bound, and its
// 8.9 KB of LDS per wave were what kept more waves out.  Every pass is full until the queue runs dry (a lane is idle for
// at most the pass in which its ticket ran out); the drain at the end is the same as the lists'.
// Output and order contract as k_trace_rolling: a ray never leaves its wave, a wave's passes claim slots in address
// order, so the records of one ray lie at increasing slots (append layout: a stable sort by `ray` is the reference's
// order, optical_table.py:125-134); [k][ray] slots are what they are.  Nothing about a ray's arithmetic depends on the
// lane or pass that carries it: bit-identical to the lists (tests/test_gpu_refill.py).
#ifndef OT_REFILL_FLAT_WG
#define OT_REFILL_FLAT_WG 256
#define OT_REFILL_FLAT_MINW 4
#endif
template <class T, uint32_t F> constexpr int refill_threads() {
    // (whole multiples of four waves, one per SIMD: the waves of a workgroup are dealt to the SIMDs in turn, and two workgroups of
    // ten waves put six on the first SIMD — where five fit — so only one of them ran: 2.4 waves per SIMD measured, for 5 asked)
    if (sizeof(T) == 4) return (F & F_FLAT) != 0 ? OT_REFILL_FLAT_WG : 256;
    return 256;
}
template <class T, uint32_t F> constexpr int refill_minw() {  // waves per SIMD the registers are capped for
    if (sizeof(T) == 4) return (F & F_FLAT) != 0 ? OT_REFILL_FLAT_MINW : (F == F_ALL ? 2 : 5);
    return F == F_ALL ? 1 : ((F & F_FLAT) != 0 ? 3 : 2);
}
template <class T, uint32_t F, bool NT, class OUT>
__global__ __launch_bounds__((refill_threads<T, F>()), (refill_minw<T, F>())) void k_trace_refill(
    SceneBlob blob, T unit, RaysT<T> in, int64_t n, int32_t K, OUT out, AppendCtl ac, int32_t* __restrict__ seg_count, int32_t* counts,
    int32_t n_classes, WaveScratch<T> ws, int32_t TICKET, int32_t capl_arg, unsigned long long* queue, int32_t mix, int32_t flat_cap) {
    constexpr bool APPEND = std::is_same<OUT, SegPlanes<T>>::value;
    struct LeadArgs {  // the parameter list: the layout of the kernel-argument segment (see k_trace_rolling)
        SceneBlob blob; T unit; RaysT<T> in; int64_t n; int32_t K; OUT out; AppendCtl ac; int32_t* seg_count; int32_t* counts;
        int32_t n_classes; WaveScratch<T> ws; int32_t CAP; int32_t capl; unsigned long long* queue; int32_t mix; int32_t flat_cap;
    };
    (void)ac; (void)seg_count; (void)queue; (void)in; (void)ws; (void)capl_arg; (void)mix;
    typedef const __attribute__((address_space(4))) RaysT<T>* RaysArgPtr;
    const RaysArgPtr in_arg = (RaysArgPtr)((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LeadArgs, in));
    extern __shared__ __align__(16) uint32_t lds[];
    for (int w = threadIdx.x; w < blob.n_words; w += blockDim.x) lds[w] = blob.words[w];
    uint32_t* const lds_tail = lds + ((blob.n_words + 3) & ~3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    FlatLds<T> flat = {nullptr, nullptr, nullptr, 0};
    if constexpr ((F & F_FLAT) != 0) {
        const int per_wave = (FlatLds<T>::fixed_bytes + flat_cap * 2 + 15) & ~15;
        uint8_t* fb = reinterpret_cast<uint8_t*>(lds_tail) + wave * per_wave;
        flat.key = reinterpret_cast<unsigned long long*>(fb);
        if constexpr (sizeof(T) == 8) flat.node = reinterpret_cast<int32_t*>(fb + 64 * 8);
        flat.queue = reinterpret_cast<uint16_t*>(fb + FlatLds<T>::fixed_bytes);
        flat.queue_cap = flat_cap;
        for (int q = lane; q < flat_cap; q += 64) flat.queue[q] = 0;  // markers only; every round leaves it zeroed again
    }
    // PARK: what only rides along to the interaction and the next record (wavelength, q, intensity, index, path length) waits
    // in LDS while the search runs — [6][64] reals per wave, a lane's own column — instead of in six of the registers that
    // decide how many waves fit a SIMD (12 LDS instructions per pass; the lists read and wrote 26)
    constexpr bool PARK = (F & F_FLAT) != 0;
    T* park = nullptr;
    if constexpr (PARK) {
        const int per_wave_flat = (FlatLds<T>::fixed_bytes + flat_cap * 2 + 15) & ~15;
        park = reinterpret_cast<T*>(reinterpret_cast<uint8_t*>(lds_tail) + (blockDim.x >> 6) * per_wave_flat) + wave * (6 * 64) + lane;
    }
    __syncthreads();  // the only workgroup barrier: the scene image is staged
    const Scene<T> sc = bind_scene<T>(lds, blob, unit);
    FlatGrid<T> flat_grid = {};
    if constexpr ((F & F_FLAT) != 0) flat_grid = flat_grid_header<T, true>(sc);
    RayState<T> r = {};
    int32_t i = 0, k = 0, cls = 0, fl = 0;
    bool busy = false;
    uint32_t tk_next = 0, tk_end = 0;  // this wave's ticket: fresh rays [tk_next, tk_end) (wave-uniform)
    bool exhausted = false;
    int64_t chunk_pos = 0;  // append layout: next free slot of this wave's chunk, and how many are left in it
    int32_t chunk_left = 0;
    // Fresh rays are fetched ONE PASS AHEAD into registers of their own (f*), as soon as the interaction has told which lanes
    // will not carry their ray on, and BEFORE the pass's fourteen segment stores: gfx9 retires a wave's loads and stores through one
    // in-order counter, so a load issued behind the stores has its data held back until every one of them is acknowledged by
    // memory — with the fetch at the top of the loop every pass waited for the stores of the pass before (2.24 ms at 4 waves
    // per SIMD where the lists, which only load in one pass of five, took 2.10).  The stores cover the fetch; the lanes take
    // their fresh rays over at the top of the next pass.  (Fetching before the interaction, for the lanes without a hit, would
    // cover more of it, but sixteen more registers are live through the interaction then: 23 spilled at 96.)
    T fox = T(0), foy = T(0), foz = T(0), fdx = T(0), fdy = T(0), fdz = T(0), fwl = T(0), fqr = T(0), fqi = T(0), fI = T(0), fn = T(0), fpl = T(0), flen = T(0);
    int32_t fi = 0, ffl = 0, fcls = 0;
    bool pending = false;
    auto fetch = [&](const unsigned long long freem) {  // freem: the lanes that take a fresh ray (wave-uniform mask)
        // (the registers hold nothing between the take-over at the top of a pass and here: said explicitly, or they would be
        // carried around the loop, through the search, for the lanes the fetch below does not write)
        fox = foy = foz = fdx = fdy = fdz = fwl = fqr = fqi = fI = fn = fpl = flen = T(0);
        fi = ffl = fcls = 0;
        if (freem == 0ull || exhausted) return;
        if (tk_next == tk_end) {
            unsigned long long first = 0;
            if (lane == 0) first = atomicAdd(OT_KARG(queue), (unsigned long long)TICKET);
            first = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32) |
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(first & 0xffffffffull));
            const unsigned long long n_now = (unsigned long long)n;
            if (first >= n_now) { exhausted = true; return; }
            tk_next = (uint32_t)first;
            tk_end = (uint32_t)(first + (unsigned long long)TICKET < n_now ? first + (unsigned long long)TICKET : n_now);
        }
        const int need = __popcll(freem), avail = (int)(tk_end - tk_next), take = need < avail ? need : avail;
        const int rank = rank_below(freem);
        if (((freem >> lane) & 1ull) && rank < take) {
            fi = (int32_t)(tk_next + (uint32_t)rank);
            RaysArgPtr ip = in_arg;
            asm volatile("" : "+s"(ip));  // opaque here: the pointers are loaded now, not hoisted out of the pass loop
            RaysT<T> in_now;
            {
                static_assert(sizeof(RaysT<T>) == 15 * sizeof(uint64_t), "RaysT is fifteen pointers");
                const __attribute__((address_space(4))) uint64_t* src = (const __attribute__((address_space(4))) uint64_t*)ip;
                uint64_t words[15];
#pragma unroll
                for (int w = 0; w < 15; ++w) words[w] = src[w];
                __builtin_memcpy(&in_now, words, sizeof(in_now));
            }
            // (plain loads: the fresh rays of a pass are a dozen consecutive records, a fraction of a cache line per field — the
            // rest of the line is wanted a pass or two later and should still be in L2 then; non-temporal loads fetched every
            // line three times: 1.55 GB read for 0.52 GB of rays)
            ffl = in_now.flags[fi];
            if constexpr ((F & F_LIMIT) != 0) fcls = in_now.id[fi];
            fox = in_now.ox[fi]; foy = in_now.oy[fi]; foz = in_now.oz[fi];
            fdx = in_now.dx[fi]; fdy = in_now.dy[fi]; fdz = in_now.dz[fi];
            fwl = in_now.wl[fi]; fqr = in_now.qr[fi]; fqi = in_now.qi[fi];
            fI = in_now.I[fi]; fn = in_now.n[fi]; fpl = in_now.pl[fi];
            flen = in_now.len ? in_now.len[fi] : Num<T>::inf();
            pending = true;
        }
        tk_next += (uint32_t)take;
    };
#ifdef OT_STAMP
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    fetch(~0ull);
    for (;;) {
        if (pending) {  // the rays fetched during the last pass
            r.ox = fox; r.oy = foy; r.oz = foz; r.dx = fdx; r.dy = fdy; r.dz = fdz;
            r.wl = fwl; r.qr = fqr; r.qi = fqi; r.I = fI; r.n = fn; r.pl = fpl; r.len = flen;
            i = fi; k = 0; cls = fcls;
            r.has_q = (ffl & OT_RAY_HAS_Q) != 0;
            r.last = (int32_t)((uint32_t)ffl >> 8) - 1;  // bits 8..31: node the ray was emitted on, plus one (generation buffers; 0 for a caller's ray)
            if ((uint32_t)r.last >= (uint32_t)sc.n_nodes) r.last = -1;  // ... that name no node of this scene
            fl = ffl & 0xff;
            busy = true;
            pending = false;
        }
        if (!__any(busy)) {
            if (exhausted) break;  // the queue is exhausted and every ray of this wave has ended
            fetch(~0ull);          // (the ticket ran out in the middle of a fetch and no ray was left alive)
            continue;
        }
        // ---- one pass: one segment of every live ray
        OT_STAMP_AT(0);
        const bool entry = busy;
        const bool active = entry && !(fl & OT_RAY_DEAD);  // optical_component.py:349: a dead ray is returned as it came
        const GateCtx gate = {counts, n_classes, cls, nullptr, nullptr, 0, 0};
        Hit<T> h;
        if constexpr (PARK) {
            park[0] = r.wl; park[64] = r.qr; park[128] = r.qi; park[192] = r.I; park[256] = r.n; park[320] = r.pl;
        }
        if constexpr ((F & F_FLAT) != 0) h = flat_grid_hit<T, F, GATE_PLAIN>(sc, flat_grid, r, active, gate, flat, lane OT_FLAT_STAMP_ARGS);
        else {
#ifdef OT_STAMP
            h = nearest_hit<T, F, GATE_PLAIN>(sc, r, active, gate, st_acc, &st_last);
#else
            h = nearest_hit<T, F, GATE_PLAIN>(sc, r, active, gate);
#endif
        }
        if constexpr (PARK) {
            r.wl = park[0]; r.qr = park[64]; r.qi = park[128]; r.I = park[192]; r.n = park[256]; r.pl = park[320];
        }
        OT_STAMP_AT(1);
        const bool hit = active && h.node >= 0;
        // the interaction first: it tells which lanes carry a ray on, and the fresh rays for the others are fetched BEFORE
        // the segment stores (see above); the parent is kept for its record
        bool survive = false;
        RayState<T> child = {};
        int32_t used = k + 1;
        if (hit) {
            MatCache<T> mc = {T(1)};
            if constexpr (F & F_REFRACT) mc = make_matcache<T, F>(sc, r.wl);
            const int nk = interact<T, F, 1>(sc, r, h, &child, mc);
            if (nk == 1) survive = k + 1 < K;
            else if (nk > 1) used = -(k + 1);  // the tree branches here: the caller re-traces it generation by generation
        }
        OT_STAMP_AT(2);
        fetch(__ballot(!survive));
        OT_STAMP_AT(3);
        // the segment record: every entry of the pass writes exactly one
        int64_t slot = APPEND ? 0 : (int64_t)k * n + (int64_t)i;
        bool room = true;
        if constexpr (APPEND) {
            const unsigned long long writers = __ballot(entry);
            const int need = __popcll(writers), rank = rank_below(writers);
            int64_t fresh_pos = 0;
            if (need > chunk_left) {  // wave-uniform: claim the next chunk; the pass may straddle the two
                const unsigned long long c0 = lane == 0 ? atomicAdd(OT_KARG(ac.cursor), (unsigned long long)OT_KARG(ac.chunk)) : 0ull;
                fresh_pos = (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(c0 >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(c0 & 0xffffffffull)));
            }
            slot = rank < chunk_left ? chunk_pos + rank : fresh_pos + (rank - chunk_left);
            if (need > chunk_left) { chunk_pos = fresh_pos + (need - chunk_left); chunk_left = OT_KARG(ac.chunk) - (need - chunk_left); }
            else { chunk_pos += need; chunk_left -= need; }
            room = slot < OT_KARG(ac.capacity);  // an output that is too small loses records, never writes outside (the cursor tells)
            if (entry && room) store_segment<T, NT>(out, slot, r, hit ? h.t : r.len, (int32_t)i, hit ? leaf_id_of<T, F>(sc, h.node) : (active ? -1 : -2));
        } else {  // the fourteen [k][ray] array pointers: from the kernel-argument segment, like the caller's ray pointers
            typedef const __attribute__((address_space(4))) uint64_t* ArgWords;
            ArgWords src = (ArgWords)((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LeadArgs, out));
            asm volatile("" : "+s"(src));
            static_assert(sizeof(SegsT<T>) == 14 * sizeof(uint64_t), "SegsT is fourteen pointers");
            uint64_t words[14];
#pragma unroll
            for (int w = 0; w < 14; ++w) words[w] = src[w];
            SegsT<T> out_now;
            __builtin_memcpy(&out_now, words, sizeof(out_now));
            if (entry) store_segment<T, NT>(out_now, slot, r, hit ? h.t : r.len, (int32_t)i, hit ? leaf_id_of<T, F>(sc, h.node) : (active ? -1 : -2));
        }
        if (entry && !survive) OT_KARG(seg_count)[i] = used;
        if (survive) { r = child; ++k; }  // (child.wl, .has_q are the parent's; .len = inf, .last = the node it was emitted on)
        busy = survive;
        OT_STAMP_AT(3);  // (stores, waited: charged to the fetch / store phase)
#ifdef OT_STAMP
        st_acc[4] += 1;
#endif
    }
    if constexpr (APPEND) {  // the unused tail of this wave's last chunk: holes
        int32_t* rp = ray_plane(out);
        for (int64_t s = chunk_pos + lane; s < chunk_pos + chunk_left; s += 64)
            if (s < OT_KARG(ac.capacity)) rp[s] = -1;
    }
#ifdef OT_STAMP
    if (lane == 0) for (int q = 0; q < 12; ++q) atomicAdd(&queue[8 + q], st_acc[q]);
#endif
}

// k_trace_pool — generation-pure tracing with the live rays of a WORKGROUP in one pool of blocks (single precision, the
// curved-surface preset: cfg 5).
// k_trace_rolling's generation-pure lists are private to a wave: 256 rays that die at different times, worked off in
// rounds, every round ending in a partial pass — 53 lanes per pass on cfg 5 — and only the first 128 positions of a list
// have their records in LDS.  Here the sixteen waves of the workgroup share ONE pool of NB blocks in LDS; a block holds up
// to 64 rays that are all on the same segment index (its generation), their records with them:
//     block = [ray index][ox oy oz dx dy dz qr qi I n pl wl][meta], 14 x 64 words
// and a state word (count | generation << 8 | LOCKED).  A wave that needs work looks at all state words at once (one lane
// per block), takes the block of the LOWEST generation with the fewest rays (A), and a second block of the same generation
// (B) to fill its lanes from: all of A's rays and as many from the END of B as make 64.  Both are locked with a
// compare-and-swap; B is released, shortened, as soon as the pass has read its records; the survivors go back into A — one
// generation on — and A is released.  Young generations first: a cohort that was filled later catches up with the older
// ones and merges with them, so partners are rarely missing.  Free blocks are refilled from the ticket queue several at a
// time (one atomic for the lot), which is what makes cohorts.
// No wave ever waits for another one while it has anything to do: a wave that finds nothing to trace and nothing to
// fill sleeps a few hundred cycles and looks again (somebody holds the blocks that are left), and leaves when the queue
// is exhausted and every block is free.
// Ordering between waves needs no waiting.  The argument, in the terms of the ISA: every access to a block or to a state /
// control word is an LDS instruction (ds_read, ds_write, ds_cmpst, ds_add) of SOME wave of this workgroup; a CU has ONE LDS
// pipeline, it executes the DS instructions it is handed one at a time, and it is handed those of one wave in program order
// (a wave's DS instructions leave through one in-order queue: that is what lets `s_waitcnt lgkmcnt(N)` mean "all but the
// N youngest are done").  So (i) a wave's record writes, issued before its state-word write, are executed by the pipeline
// before it; (ii) any wave whose ds_read of that state word returns the new value had that read executed after the write,
// hence after the record writes, and its own record reads, issued after the state read (it waits for the value: lgkmcnt),
// execute later still: they see the records.  (iii) The other direction: a wave reads B's records into registers, then
// writes B's state word; whoever sees the new word and overwrites B does so after those reads have executed.  No step
// depends on how long anything takes, only on the order within one wave and on the pipeline being single — which the
// jitter test (OT_OPT_POOL_JITTER: publications held back by thousands of cycles at random) exercises.  The fences below are
// therefore wave-scope: they only keep the COMPILER from reordering the accesses.  A workgroup-scope release would also
// wait for the pass's fourteen global segment stores — 1.8x the pass time.
// Two policies matter more than anything else here (cfg 5, 12.3 ms as built): refilling EAGERLY, as soon as 16 / 8 / 4 blocks
// are free instead of when a wave has nothing to trace, makes small cohorts whose generations never meet (13.7 / 15.1 /
// 16.6 ms); the OLDEST generation first instead of the youngest lets every cohort run ahead on its own (21.7 ms).
// Passes are generation-pure by construction, so the nearest-hit search runs as in k_trace_rolling; nothing about a ray's
// arithmetic depends on which wave or pass carries it: results are bit-identical (tests/test_gpu_pool.py).
static constexpr int POOL_BLOCK_WORDS = 14 * 64;
static constexpr uint32_t POOL_LOCKED = 0x80000000u;
template <class T, uint32_t F, bool NT, class OUT>
__global__ __launch_bounds__(1024, 1) void k_trace_pool(
    SceneBlob blob, T unit, RaysT<T> in, int64_t n, int32_t K, OUT out, AppendCtl ac, int32_t* __restrict__ seg_count, int32_t* counts,
    int32_t n_classes, WaveScratch<T> ws, int32_t NB, int32_t capl_arg, unsigned long long* queue, int32_t mix, int32_t flat_cap) {
    static_assert(sizeof(T) == 4, "a double-precision record is 100 bytes: too few blocks would fit");
    static_assert((F & (F_LIMIT | F_FLAT)) == 0, "no count classes, no pair queue");
    constexpr bool APPEND = std::is_same<OUT, SegPlanes<T>>::value;
    struct LeadArgs {  // the parameter list: the layout of the kernel-argument segment (see k_trace_rolling)
        SceneBlob blob; T unit; RaysT<T> in; int64_t n; int32_t K; OUT out; AppendCtl ac; int32_t* seg_count; int32_t* counts;
        int32_t n_classes; WaveScratch<T> ws; int32_t CAP; int32_t capl; unsigned long long* queue; int32_t mix; int32_t flat_cap;
    };
    (void)ac; (void)seg_count; (void)queue; (void)in; (void)ws; (void)capl_arg; (void)mix;  // (flat_cap: the jitter period of the protocol test, 0 = off)
    typedef const __attribute__((address_space(4))) RaysT<T>* RaysArgPtr;
    const RaysArgPtr in_arg = (RaysArgPtr)((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LeadArgs, in));
    extern __shared__ __align__(16) uint32_t lds[];
    for (int w = threadIdx.x; w < blob.n_words; w += blockDim.x) lds[w] = blob.words[w];
    uint32_t* const state = lds + ((blob.n_words + 3) & ~3);   // [64] block states, then [4] control words
    uint32_t* const ctl = state + 64;                          // [0] the ticket queue is exhausted; [1], [2], [4..11]: append layout, see claim()
    uint32_t* const pool = ctl + 16;                           // NB blocks of POOL_BLOCK_WORDS
    if (threadIdx.x < 80) state[threadIdx.x] = threadIdx.x == 65 && APPEND ? (uint32_t)min(16 * OT_KARG(ac.chunk), 1 << 19) : 0u;  // (ctl[1]: see claim())
    const int lane = threadIdx.x & 63;
    __syncthreads();  // the only workgroup barrier: image staged, pool empty
    const Scene<T> sc = bind_scene<T>(lds, blob, unit);
#ifdef OT_STAMP  // diagnostic build: [0] passes [1] rays in them [2] sleeps [3] lost locks [4] fills [5] blocks filled [6] passes with a second block [9] sleeps inside claim()
    unsigned long long pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [7] cycles between passes, [8] cycles in passes
    unsigned long long pt = __builtin_amdgcn_s_memtime();
#define OT_POOL_COUNT(k, v) pc[k] += (v)
#define OT_POOL_TIME(k) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); pc[k] += _t - pt; pt = _t; } while (0)
#else
#define OT_POOL_COUNT(k, v) do {} while (0)
#define OT_POOL_TIME(k) do {} while (0)
#endif
    // Append layout: slots are claimed per WORKGROUP.  A ray changes waves from pass to pass; if every wave filled chunks
    // of its own (as in k_trace_rolling), the records of one ray would not lie at increasing addresses, and "a stable
    // sort by ray is the reference's order" (include/optable_hip.h) would not hold.  So the workgroup fills ONE chunk
    // of wg_chunk slots at a time: ctl[1] = (epoch << 20 | next free offset), bumped by every pass with one LDS atomic;
    // the pass whose claim crosses the end of the chunk marks what is left of it as holes, claims the next chunk from the
    // device-wide cursor, enters its base in ctl[4 + 2 * (epoch & 3)] and opens the new epoch; passes that claimed beyond
    // the end wait for that (a few microseconds, for a wave that itself waits for nobody).  Claims within a workgroup are
    // in time order and chunks are claimed in address order: segment k + 1 of a ray, whose pass begins after the pass of
    // segment k has released its block, lies behind segment k.
    const uint32_t wg_chunk = APPEND ? (uint32_t)min(16 * OT_KARG(ac.chunk), 1 << 19) : 0u;
    // (ctl[1] starts as epoch 0, offset wg_chunk — a full chunk: the first claim opens the first real one)
    bool broken = false;  // a bound that "cannot trigger" did: the launch is reported as failed (see the end of the kernel)
    uint32_t pass_no = 0;
    // OT_OPT_POOL_JITTER (tests): before one in `flat_cap` publications of a state or control word the wave stalls for ~8000
    // cycles — the records are written, the word that says so is late.  Whatever the protocol leaves to timing shows up as
    // a wrong record under this; with the option off it is one scalar compare per site.
    auto jitter = [&](uint32_t site) {
        if (flat_cap > 0) {
            const uint32_t h = ((pass_no * 2654435761u) ^ (site * 40503u) ^ ((uint32_t)(threadIdx.x >> 6) * 9176u) ^ (blockIdx.x * 131u)) >> 7;
            if (h % (uint32_t)flat_cap == 0u)
                for (int q = 0; q < 64; ++q) __builtin_amdgcn_s_sleep(127);
        }
    };

    auto chunk_base = [&](uint32_t epoch) -> int64_t {
        const uint32_t lo = __hip_atomic_load(&ctl[4 + 2 * (epoch & 3u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t hi = __hip_atomic_load(&ctl[5 + 2 * (epoch & 3u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)lo));
    };
    auto claim = [&](int take) -> int64_t {  // first of `take` consecutive slots (wave-uniform)
      if constexpr (APPEND) {
        for (int tries = 0; tries < (1 << 16); ++tries) {
            uint32_t old = 0;
            if (lane == 0) old = atomicAdd(&ctl[1], (uint32_t)take);
            old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
            const uint32_t epoch = old >> 20, off = old & 0xfffffu;
            if (off + (uint32_t)take <= wg_chunk) return chunk_base(epoch) + (int64_t)off;
            if (off <= wg_chunk) {  // this claim crosses the end of the chunk
                if (off < wg_chunk) {
                    const int64_t hole = chunk_base(epoch) + (int64_t)off + lane;  // fewer than `take` <= 64 slots are left
                    if (off + (uint32_t)lane < wg_chunk && hole < OT_KARG(ac.capacity)) ray_plane(out)[hole] = -1;
                }
                unsigned long long nb = 0;
                if (lane == 0) nb = atomicAdd(OT_KARG(ac.cursor), (unsigned long long)wg_chunk);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(nb & 0xffffffffull)), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(nb >> 32));
                if (lane == 0) {
                    __hip_atomic_store(&ctl[4 + 2 * ((epoch + 1u) & 3u)], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(&ctl[5 + 2 * ((epoch + 1u) & 3u)], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                jitter(4);
                if (lane == 0) __hip_atomic_store(&ctl[1], ((epoch + 1u) << 20) | (uint32_t)take, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return (int64_t)(((unsigned long long)hi << 32) | lo);
            }
            for (int w = 0; w < (1 << 16); ++w) {  // claimed beyond the end: the wave that crossed it is opening the next chunk
                const uint32_t now = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if ((now >> 20) != epoch) break;
                __builtin_amdgcn_s_sleep(1);
                OT_POOL_COUNT(9, 1);
            }
        }
        broken = true;
        return OT_KARG(ac.capacity);  // (unreachable; a claim that never succeeds drops its records instead of writing anywhere)
      } else {
        return 0;
      }
    };
    int idle = 0;
    uint32_t prev_st = 0;    // the state word this lane saw at the last poll: a poll that sees ANY word changed is progress
    for (;;) {
        // ---- choose: every lane looks at one block
        const uint32_t st = lane < NB ? __hip_atomic_load(&state[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : POOL_LOCKED;
        if (__any(st != prev_st)) idle = 0;  // somebody released, filled or advanced a block since the last look: the workgroup is alive
        prev_st = st;
        const int cnt_l = (int)(st & 127u), gen_l = (int)((st >> 8) & 0xfffffu);
        const bool elig = !(st & POOL_LOCKED) && cnt_l > 0;
        // lowest generation first; among its blocks the first one from a starting point of this wave's own (sixteen waves
        // that all went for "the smallest block" would fight over the same one and fifteen would choose again)
        const int rot = (lane - 4 * (int)(threadIdx.x >> 6)) & 63;
        const int key = elig ? ((gen_l << 6) | rot) : 0x7fffffff;
        const int kmin = wave_all_min_i32(key);
        if (kmin == 0x7fffffff) {  // nothing to trace right now
            const unsigned long long freeb = __ballot(lane < NB && st == 0u);
            const bool exhausted = __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;
            if (freeb != 0ull && !exhausted) {
                // ---- fill: lock the free blocks, ONE ticket atomic for all of them, 64 consecutive rays each
                bool got = false;
                if ((freeb >> lane) & 1ull) got = atomicCAS(&state[lane], 0u, POOL_LOCKED) == 0u;
                const unsigned long long gm = __ballot(got);
                const int m = __popcll(gm);
                if (m > 0) {
                    unsigned long long first = 0;
                    if (lane == 0) first = atomicAdd(OT_KARG(queue), 64ull * (unsigned long long)m);
                    first = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(first & 0xffffffffull));
                    const unsigned long long n_now = (unsigned long long)n;
                    unsigned long long left = gm;
                    int j = 0;
                    while (left) {  // wave-uniform: the blocks this wave locked, in index order
                        const int b = __builtin_ctzll(left);
                        left &= left - 1ull;
                        const unsigned long long start = first + 64ull * (unsigned long long)j;
                        const int c = start >= n_now ? 0 : (int)(n_now - start < 64ull ? n_now - start : 64ull);
                        if (lane < c) pool[b * POOL_BLOCK_WORDS + lane] = (uint32_t)start + (uint32_t)lane;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        jitter(1);
                        if (lane == 0) __hip_atomic_store(&state[b], (uint32_t)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // generation 0, unlocked (0 rays: free)
                        ++j;
                    }
                    if (first + 64ull * (unsigned long long)m >= n_now && lane == 0)
                        __hip_atomic_store(&ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    idle = 0;
                    OT_POOL_COUNT(4, 1);
                    OT_POOL_COUNT(5, m);
                }
                continue;
            }
            if (exhausted && !__any(lane < NB && st != 0u)) break;  // nothing left anywhere
            // blocks exist but other waves hold them (or are filling them): look again shortly
            __builtin_amdgcn_s_sleep(4);
            OT_POOL_COUNT(2, 1);
            // (only polls that saw NOTHING change count: a long tail in which other waves work off the last blocks — a cavity with
            // 1e5 bounces — is not a failure; 2^22 polls of an unchanged pool are, a wave never holds a block longer than one pass)
            if (++idle > (1 << 22)) { broken = true; break; }
            continue;
        }
        idle = 0;
        const int gen = kmin >> 6;
        const int A = ((kmin & 63) + 4 * (int)(threadIdx.x >> 6)) & 63;
        const int cntA = __builtin_amdgcn_readlane(cnt_l, A);
        int B = -1, cntB = 0, takeB = 0;
        if (cntA < 64) {  // a second block of the same generation to fill the lanes from
            const int k2 = wave_all_min_i32((elig && gen_l == gen && lane != A) ? rot : 0x7fffffff);
            if (k2 != 0x7fffffff) {
                B = (k2 + 4 * (int)(threadIdx.x >> 6)) & 63;
                cntB = __builtin_amdgcn_readlane(cnt_l, B);
                takeB = min(64 - cntA, cntB);
            }
        }
        // lock A (and B): the states must still be what this wave saw
        bool mine = true;
        if (lane == A || lane == B) mine = atomicCAS(&state[lane], st, st | POOL_LOCKED) == st;
        const unsigned long long failed = __ballot(!mine);
        if (failed != 0ull) {  // somebody was faster: give back what was locked and choose again
            if ((lane == A || lane == B) && mine) __hip_atomic_store(&state[lane], st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            OT_POOL_COUNT(3, 1);
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- one pass: all of A, the last takeB rays of B
        {
            const int take = cntA + takeB;
            OT_POOL_TIME(7);
            OT_POOL_COUNT(0, 1);
            OT_POOL_COUNT(1, take);
            OT_POOL_COUNT(6, B >= 0 ? 1 : 0);
            const bool entry = lane < take;
            const bool from_b = lane >= cntA;
            const uint32_t* src = pool + (from_b ? B * POOL_BLOCK_WORDS + (cntB - takeB) + (lane - cntA) : A * POOL_BLOCK_WORDS + lane);
            const int32_t k = gen;
            int32_t i = 0;
            RayState<T> r = {};
            int32_t fl = 0;
            if (entry) {
                i = (int32_t)src[0];
                if (k == 0) {  // first segment: the caller's arrays
                    RaysArgPtr ip = in_arg;
                    asm volatile("" : "+s"(ip));
                    RaysT<T> in_now;
                    {
                        static_assert(sizeof(RaysT<T>) == 15 * sizeof(uint64_t), "RaysT is fifteen pointers");
                        const __attribute__((address_space(4))) uint64_t* w = (const __attribute__((address_space(4))) uint64_t*)ip;
                        uint64_t words[15];
#pragma unroll
                        for (int q = 0; q < 15; ++q) words[q] = w[q];
                        __builtin_memcpy(&in_now, words, sizeof(in_now));
                    }
                    fl = in_now.flags[i];
                    r.ox = ld_once(in_now.ox + i); r.oy = ld_once(in_now.oy + i); r.oz = ld_once(in_now.oz + i);
                    r.dx = ld_once(in_now.dx + i); r.dy = ld_once(in_now.dy + i); r.dz = ld_once(in_now.dz + i);
                    r.len = in_now.len ? in_now.len[i] : Num<T>::inf();
                    r.has_q = (fl & OT_RAY_HAS_Q) != 0;
                    r.last = (int32_t)((uint32_t)fl >> 8) - 1;
                    if ((uint32_t)r.last >= (uint32_t)sc.n_nodes) r.last = -1;
                    fl &= 0xff;
                    r.wl = ld_once(in_now.wl + i); r.qr = ld_once(in_now.qr + i); r.qi = ld_once(in_now.qi + i);
                    r.I = ld_once(in_now.I + i); r.n = ld_once(in_now.n + i); r.pl = ld_once(in_now.pl + i);
                } else {
                    const T* rec = reinterpret_cast<const T*>(src);
                    r.ox = rec[64]; r.oy = rec[2 * 64]; r.oz = rec[3 * 64];
                    r.dx = rec[4 * 64]; r.dy = rec[5 * 64]; r.dz = rec[6 * 64];
                    r.qr = rec[7 * 64]; r.qi = rec[8 * 64]; r.I = rec[9 * 64];
                    r.n = rec[10 * 64]; r.pl = rec[11 * 64]; r.wl = rec[12 * 64];
                    const int32_t meta = (int32_t)src[13 * 64];
                    r.last = (meta >> 8) - 1;
                    r.has_q = meta & OT_RAY_HAS_Q;
                    r.len = Num<T>::inf();
                }
            }
            // B's rays are in registers: give it back, shorter (its remaining rays sit where they sat)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            jitter(2);
            if (B >= 0 && lane == 0) {
                const int rest = cntB - takeB;
                __hip_atomic_store(&state[B], rest > 0 ? (uint32_t)((gen << 8) | rest) : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            const bool active = entry && !(fl & OT_RAY_DEAD);
            const GateCtx gate = {counts, n_classes, 0, nullptr, nullptr, 0, 0};
            const Hit<T> h = nearest_hit<T, F, GATE_PLAIN>(sc, r, active, gate);
            const bool hit = active && h.node >= 0;
            int64_t slot = APPEND ? 0 : (int64_t)k * n + (int64_t)i;
            bool room = true;
            if constexpr (APPEND) {
                slot = claim(take) + lane;  // the entries of a pass are its first `take` lanes
                room = slot < OT_KARG(ac.capacity);
                if (entry && room) store_segment<T, NT>(out, slot, r, hit ? h.t : r.len, (int32_t)i, hit ? leaf_id_of<T, F>(sc, h.node) : (active ? -1 : -2));
            } else {
                typedef const __attribute__((address_space(4))) uint64_t* ArgWords;
                ArgWords w = (ArgWords)((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LeadArgs, out));
                asm volatile("" : "+s"(w));
                static_assert(sizeof(SegsT<T>) == 14 * sizeof(uint64_t), "SegsT is fourteen pointers");
                uint64_t words[14];
#pragma unroll
                for (int q = 0; q < 14; ++q) words[q] = w[q];
                SegsT<T> out_now;
                __builtin_memcpy(&out_now, words, sizeof(out_now));
                if (entry) store_segment<T, NT>(out_now, slot, r, hit ? h.t : r.len, (int32_t)i, hit ? leaf_id_of<T, F>(sc, h.node) : (active ? -1 : -2));
            }
            bool survive = false;
            RayState<T> child = {};
            if (entry) {
                int32_t used = k + 1;
                if (hit) {
                    MatCache<T> mc = {T(1)};
                    if constexpr (F & F_REFRACT) mc = make_matcache<T, F>(sc, r.wl);
                    const int nk = interact<T, F, 1>(sc, r, h, &child, mc);
                    if (nk == 1) survive = k + 1 < K;
                    else if (nk > 1) used = -(k + 1);  // the tree branches here: the caller re-traces it generation by generation
                }
                if (!survive) OT_KARG(seg_count)[i] = used;
            }
            const unsigned long long mk = __ballot(survive);
            if (survive) {  // into A, from its first position on (every ray of A is in this pass's registers)
                uint32_t* dst = pool + A * POOL_BLOCK_WORDS + __popcll(mk & ((1ull << lane) - 1ull));
                T* rec = reinterpret_cast<T*>(dst);
                dst[0] = (uint32_t)i;
                rec[64] = child.ox; rec[2 * 64] = child.oy; rec[3 * 64] = child.oz;
                rec[4 * 64] = child.dx; rec[5 * 64] = child.dy; rec[6 * 64] = child.dz;
                rec[7 * 64] = child.qr; rec[8 * 64] = child.qi; rec[9 * 64] = child.I;
                rec[10 * 64] = child.n; rec[11 * 64] = child.pl; rec[12 * 64] = r.wl;
                dst[13 * 64] = (uint32_t)((r.has_q ? OT_RAY_HAS_Q : 0) | ((child.last + 1) << 8));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            jitter(3);
            ++pass_no;
            if (lane == 0) {
                const int left = __popcll(mk);
                __hip_atomic_store(&state[A], left > 0 ? (uint32_t)(((gen + 1) << 8) | left) : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            OT_POOL_TIME(8);
        }
    }
    if constexpr (APPEND) {
        // Should one of the bounds above ever stop a wave, rays may be left untraced: the slot count becomes impossible, and
        // whoever reads it (ot_trace_append_*'s *n_slots; SegmentBatch.n_valid raises) sees that the launch failed
        if (broken && lane == 0) atomicMax(OT_KARG(ac.cursor), 1ull << 62);
    }
    if constexpr (APPEND) {  // the last wave to leave marks the unused tail of the workgroup's last chunk: holes
        uint32_t gone = 0;
        if (lane == 0) gone = atomicAdd(&ctl[2], 1u);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)gone) + 1u == (blockDim.x >> 6)) {
            const uint32_t last = __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)last) >> 20, off = (uint32_t)__builtin_amdgcn_readfirstlane((int)last) & 0xfffffu;
            if (off < wg_chunk) {
                const int64_t b = chunk_base(epoch);
                int32_t* rp = ray_plane(out);
                for (int64_t s = b + off + lane; s < b + wg_chunk; s += 64)
                    if (s < OT_KARG(ac.capacity)) rp[s] = -1;
            }
        }
    }
#ifdef OT_STAMP
    OT_POOL_TIME(7);
    if (lane == 0) for (int q = 0; q < 10; ++q) atomicAdd(&queue[8 + q], pc[q]);
#endif
}

// k_stream_ceiling: the fused kernel's memory traffic with no tracing — reads one ray record,
// writes K segment records per ray through the same SoA streams.  What this access pattern can
// reach on the device; reported next to the trace kernel (bench.py).
template <class T, bool NT, class OUT>
__global__ __launch_bounds__(256) void k_stream_ceiling(RaysT<T> in, int64_t n, int32_t K, OUT out, int32_t* seg_count, int32_t pair) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        RayState<T> r = load_ray(in, i, in.flags[i]);
        const int32_t cls = in.id[i];
        const bool both = pair && (i | 1) < n;  // n is even when `pair` is set: both lanes of a pair are in range
        for (int32_t k = 0; k < K; ++k) {
            if (both) store_segment_paired<T, NT>(out, (int64_t)k * n + i, r, r.len, (int32_t)i, cls, (threadIdx.x & 1) != 0);
            else store_segment<T, NT>(out, (int64_t)k * n + i, r, r.len, (int32_t)i, cls);
            r.ox += T(1);  // keep the K records distinct so the stores cannot be merged
        }
        seg_count[i] = K;
    }
}

// ------------------------------------------------------------------------------------------
// breadth-first generation step
// A generation lists its rays tree by tree (parent order), so `tree[]` is non-decreasing and the first ray of
// ray i's tree is a lower bound: rank within the tree = i - tree_head(i).  ~log2(n) L2-resident loads per ray
// replace the head-flag kernel, the max-scan and the mark kernel of the first version.
__device__ __forceinline__ int64_t tree_head(const int32_t* __restrict__ tree, int64_t i) {
    const int32_t t = tree[i];
    for (int k = 1; k <= 3; ++k)  // most trees have a handful of rays per generation: look at the neighbours first
        if (i - k < 0 || tree[i - k] != t) return i - k + 1;
    int64_t lo = 0, hi = i - 3;  // tree[hi] == t
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (tree[mid] < t) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// k_gen_probe: the pre-pass of a generation in a scene with count-limited leaves.  It only records which limited
// leaves each ray hits GEOMETRICALLY (probe[slot][i] = 1; no outputs, no children): a per-slot scan of these flags
// then gives every ray the number of EARLIER rays of its tree that hit the leaf, which is what makes the gate
// FIFO-exact inside a generation (optical_component.py:140-149, 359-362; k_gen_rank / k_gen_counts below).
template <class T, uint32_t F, bool SCENE_IN_LDS>
__global__ __launch_bounds__(256) void k_gen_probe(SceneBlob blob, T unit, RaysT<T> in, const int32_t* __restrict__ tree, int64_t n,
                                                   const int32_t* __restrict__ budget, int32_t* counts, int32_t n_classes, int32_t* probe) {
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t* base = blob.words;
    if (SCENE_IN_LDS) {
        for (int w = threadIdx.x; w < blob.n_words; w += blockDim.x) lds[w] = blob.words[w];
        __syncthreads();
        base = lds;
    }
    const Scene<T> sc = bind_scene<T>(base, blob, unit);
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int32_t my_tree = -1;
    int start_lane = -1;
    if (i < n) {
        my_tree = tree[i];
        if (lane == 0 || tree[i - 1] != my_tree) start_lane = lane;
    }
    const int head_lane = wave_incl_max_i32(start_lane);  // see k_gen_pass
    int64_t head = (i - lane) + head_lane;
    if (i < n && head_lane == 0) head = tree_head(tree, i - lane);
    const bool active = i < n && (i - head) < (int64_t)budget[my_tree];
    RayState<T> r = {};
    int32_t cls = 0, fl = 0;
    if (active) {
        fl = in.flags[i];
        r = load_ray(in, i, fl);
        if ((uint32_t)r.last >= (uint32_t)sc.n_nodes) r.last = -1;
        cls = in.id[i];
    }
    const GateCtx gate = {counts, n_classes, cls, nullptr, probe, n, i};
    (void)nearest_hit<T, F, GATE_PROBE>(sc, r, active && !(fl & OT_RAY_DEAD), gate);
}

// ------------------------------------------------------------------------------------------
// k_gen_pass: a generation in TWO streaming passes instead of one pass with an ordered look-back.
//   COUNT  (EMIT = false): rank within the tree, trace, interaction -> per ray one byte (processed | children << 1)
//          and per WAVE one packed total (processed << 32 | children).
//   (host) exclusive scan over the wave totals (hipCUB), n / 64 words.
//   EMIT   (EMIT = true):  the same trace again; slots = scanned wave prefix + rank inside the wave (two ballot /
//          shuffle scans); segment and children written once, in stable order, straight from registers.
// Neither pass has a workgroup barrier, a ticket or a dependency on another wave, so both run at the occupancy
// their registers allow and stream at memory speed; the price is reading a generation's rays twice (116 of ~440
// bytes per processed ray).  The single-pass kernel of round 1 (tiles handed out by an atomic ticket, decoupled look-back
// over packed (segments, children) words, children held in registers across it) spent 70 % of its wave-cycles waiting: a tile's 90 KB of stores, the next tile's loads behind them (gfx9
// counts loads and stores in one in-order counter), the trace and the look-back were strictly serial per workgroup,
// and three workgroups per CU (168 VGPRs: both children of every ray live across the look-back) could not hide it —
// cfg 4 with reflectivity 0.2: 26.6 ms in round 1, 20.0 ms with four look-back windows in flight and no scratch,
// and now the two passes.
// Both passes run the same device code on the same inputs, so they take the same decisions; should a last-bit
// difference between the two compilations ever flip one (a hit within an ulp of an aperture edge, sin(theta_t)
// within an ulp of 1), EMIT still never writes outside the slots COUNT reserved: it emits at most the counted number
// of children, fills a missing one with a zero-intensity dead ray, and counts the event in `mismatch` (tests
// assert 0).
// The emit pass stores PLAIN: its children are the next generation's input, read back within the same millisecond, and
// the L2 / Infinity Cache serve part of that; with non-temporal stores cfg 4 (R = 0.2) took 19.7 instead of 15.0 ms.
// LOOK-AHEAD (MODE 2, round 4): the emit pass has every child of its rays in registers, so it also does what the count pass
// of the NEXT generation would do with it — the search and the interaction, for the number of children only — and leaves that
// number per child (`ahead`, one byte at the child's slot).  The next generation then needs no count pass over its rays:
// k_gen_recount (below) turns those bytes, tree[] and budget[] into the code bytes and wave totals — 10 bytes per ray read and
// written instead of the 109 of a ray record.  A ray is still traced twice (as a child here, as a parent in its own emit pass),
// but read once: 385 instead of 484 bytes per processed ray on cfg 4 with reflectivity 0.2.  Should the two traces ever
// disagree, the emit pass fills or drops as described above and counts it in `mismatch`.  Light scenes without count gates only.
static constexpr bool GEN_NT = false;
template <class T, uint32_t F, bool SCENE_IN_LDS, int MODE>  // 0 count, 1 emit, 2 emit + look-ahead
__global__ __launch_bounds__(256) void k_gen_pass(SceneBlob blob, T unit, RaysT<T> in, const int32_t* __restrict__ tree, int64_t n,
                                                  int32_t* __restrict__ budget, const int64_t* cursor, SegsT<T> out,
                                                  int64_t out_capacity, RaysOutT<T> next, int32_t* next_tree, int64_t next_capacity,
                                                  uint8_t* code, unsigned long long* wave_total, const unsigned long long* wave_prefix,
                                                  int32_t* counts, int32_t n_classes, const int32_t* rank, unsigned long long* mismatch,
                                                  int32_t* hit_node, T* hit_t, int32_t drop_doomed, uint8_t* __restrict__ ahead) {
    constexpr bool EMIT = MODE != 0, AHEAD = MODE == 2;
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t* base = blob.words;
    if (SCENE_IN_LDS) {
        for (int w = threadIdx.x; w < blob.n_words; w += blockDim.x) lds[w] = blob.words[w];
        __syncthreads();
        base = lds;
    }
    const Scene<T> sc = bind_scene<T>(base, blob, unit);
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gwave = i >> 6;
    // rank of the ray inside its tree (as in k_gen_probe): rays beyond the tree's remaining budget are dropped.  A
    // generation lists its rays tree by tree, so the head of ray i's tree is the nearest earlier start in its wave (a
    // max-scan of lane numbers: six DPP steps) or, when the tree began before the wave, a lower bound of tree[].
    int32_t my_tree = -1;
    int start_lane = -1;
    if (i < n) {
        my_tree = tree[i];
        if (lane == 0 || tree[i - 1] != my_tree) start_lane = lane;
    }
    const int head_lane = wave_incl_max_i32(start_lane);
    int64_t head = (i - lane) + head_lane;
    if (i < n && head_lane == 0) head = tree_head(tree, i - lane);
    const int32_t c = (EMIT && i < n) ? (int32_t)code[i] : 0;  // EMIT: what the count pass decided
    const int64_t bud = (!EMIT && i < n) ? (int64_t)budget[my_tree] : 0;  // (only the count pass reads budgets)
    const bool active = EMIT ? (c & 1) != 0 : (i < n && (i - head) < bud);
    // A tree whose budget ends with this generation — it has a ray of rank budget - 1 here — will have every ray of the
    // next generation dropped (optical_table.py:138-144): what its rays emit now can never be processed.  Such children
    // are not emitted at all (OT_OPT_GEN_DROP_DOOMED): the last generation of a capped tree is its largest, and writing
    // its children was a sixth of the bytes of a cfg 4 (R = 0.2) trace.  The count pass decides (it is the one that may
    // read budgets), bit 3 of the code byte tells the emit pass.
    bool doomed = (c & 8) != 0;
    if (!EMIT && active && (drop_doomed & 1)) {
        const int64_t last_needed = head + bud - 1;
        doomed = last_needed < n && tree[last_needed] == my_tree;
    }
    if (EMIT && i < n && (i == n - 1 || tree[i + 1] != my_tree)) {
        // Last ray of its tree in this generation: the tree's budget shrinks by the rays processed (optical_table.py:
        // 138-144).  Only the count pass reads budgets, and it has finished: one writer per tree, no reader.
        const int64_t in_gen = i - head + 1;
        const int32_t b = budget[my_tree];
        budget[my_tree] = b - (int32_t)(in_gen < b ? in_gen : b);
    }
    RayState<T> r = {};
    int32_t cls = 0, fl = 0;
    if (active) {
        fl = in.flags[i];
        r = load_ray(in, i, fl);
        if ((uint32_t)r.last >= (uint32_t)sc.n_nodes) r.last = -1;
        cls = in.id[i];
    }
    const bool dead = active && (fl & OT_RAY_DEAD);
    const GateCtx gate = {counts, n_classes, cls, rank, nullptr, n, i};
    // Heavy scenes (hit_node != NULL): the count pass leaves its decision per ray — hit node and distance, 12 bytes in
    // double precision — and the emit pass rebuilds the hit from them instead of searching the scene a second time: the
    // search is what a generation of such a scene costs, and the two passes cannot disagree.  Light scenes trace twice:
    // for them the search is a few planes and the passes run at memory speed either way.
    Hit<T> h;
    if (EMIT && hit_node) {
        h = rebuild_hit<T, F>(sc, r, (active && !dead) ? hit_node[i] : -1, (active && !dead) ? hit_t[i] : Num<T>::inf());
    } else {
        h = nearest_hit<T, F, GATE_TABLE>(sc, r, active && !dead, gate);
        if (!EMIT && hit_node && i < n) { hit_node[i] = h.node; hit_t[i] = h.t; }
    }
    int32_t nk = 0;
    RayState<T> ch[2];  // indexed by constants only
    if (active && !dead && h.node >= 0) nk = interact<T, F, 2>(sc, r, h, ch, make_matcache<T, F>(sc, r.wl));
    if (doomed) nk = 0;
    if (!EMIT) {
        if (i < n) code[i] = (uint8_t)((active ? 1 : 0) | (nk << 1) | (doomed ? 8 : 0));
        // wave totals: processed rays and children
        const int n_act = __popcll(__ballot(active));
        int kids;
        wave_excl_scan_i32(nk, kids);
        if (lane == 0 && (gwave << 6) < n) wave_total[gwave] = ((unsigned long long)n_act << 32) | (unsigned long long)kids;
        return;
    }
    // EMIT: slots from the scanned wave prefix and the counted codes
    const bool c_active = active;
    const int32_t c_nk = (c >> 1) & 3;
    const unsigned long long act_mask = __ballot(c_active);
    const int seg_rank = __popcll(act_mask & ((1ull << lane) - 1ull));
    int kids_total;
    const int kid_excl = wave_excl_scan_i32(c_nk, kids_total);
    const unsigned long long before = (gwave << 6) < n ? wave_prefix[gwave] : 0ull;
    if (c_active && c_nk != nk) atomicAdd(mismatch, 1ull);  // see the header: never expected
    if (c_active) {
        const int64_t slot = *cursor + (int64_t)(before >> 32) + seg_rank;
        if (slot < out_capacity) {
            const int32_t t = my_tree;
            if (dead) store_segment<T, GEN_NT>(out, slot, r, r.len, t, -2);
            else if (h.node < 0) store_segment<T, GEN_NT>(out, slot, r, r.len, t, -1);
            else store_segment<T, GEN_NT>(out, slot, r, h.t, t, leaf_id_of<T, F>(sc, h.node));
        }
    }
    const int64_t d0 = (int64_t)(before & 0xffffffffull) + kid_excl;
    auto put = [&](const RayState<T>& k, int64_t d, bool real) {
        if (d >= next_capacity) return;
        st<GEN_NT>(next.ox + d, k.ox); st<GEN_NT>(next.oy + d, k.oy); st<GEN_NT>(next.oz + d, k.oz);
        st<GEN_NT>(next.dx + d, k.dx); st<GEN_NT>(next.dy + d, k.dy); st<GEN_NT>(next.dz + d, k.dz);
        st<GEN_NT>(next.wl + d, k.wl); st<GEN_NT>(next.qr + d, k.qr); st<GEN_NT>(next.qi + d, k.qi);
        st<GEN_NT>(next.I + d, real ? k.I : T(0)); st<GEN_NT>(next.n + d, k.n); st<GEN_NT>(next.pl + d, k.pl);
        next.flags[d] = real ? ((fl & OT_RAY_HAS_Q) | ((k.last + 1) << 8)) : OT_RAY_DEAD;
        next.id[d] = cls;
        next_tree[d] = (drop_doomed & 2) ? (int32_t)i : my_tree;  // bit 1 (OT_OPT_GEN_PARENT_INDEX): the parent's input index, for hosts that keep an object per ray
    };
    if (c_nk > 0) put(nk > 0 ? ch[0] : r, d0, nk > 0);
    if (c_nk > 1) put(nk > 1 ? ch[1] : r, d0 + 1, nk > 1);
    if constexpr (AHEAD) {
        // the children as the next generation will load them (load_ray: no `len` in a generation buffer, flags as `put` wrote them)
        auto count_of = [&](RayState<T> q, bool real) -> int32_t {
            q.len = Num<T>::inf();
            q.has_q = (fl & OT_RAY_HAS_Q) != 0;
            if ((uint32_t)q.last >= (uint32_t)sc.n_nodes) q.last = -1;
            const Hit<T> h2 = nearest_hit<T, F, GATE_TABLE>(sc, q, real, gate);
            RayState<T> gc[2];
            int32_t n2 = 0;
            if (real && h2.node >= 0) n2 = interact<T, F, 2>(sc, q, h2, gc, make_matcache<T, F>(sc, q.wl));
            return n2;
        };
        const bool real0 = c_nk > 0 && nk > 0, real1 = c_nk > 1 && nk > 1;
        if (__any(c_nk > 0)) {  // (wave-uniform: the search is a wave-wide walk)
            const int32_t a0 = count_of(ch[0], real0);
            if (c_nk > 0 && d0 < next_capacity) ahead[d0] = (uint8_t)a0;
        }
        if (__any(c_nk > 1)) {
            const int32_t a1 = count_of(ch[1], real1);
            if (c_nk > 1 && d0 + 1 < next_capacity) ahead[d0 + 1] = (uint8_t)a1;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_gen_one: a generation in ONE pass (round 4).  Every WAVE traces a tile of 64 consecutive rays of the generation
// once, keeps the 0-2 children of each ray in registers, and learns where its segments and children go from a DECOUPLED
// LOOK-BACK over one 64-bit descriptor per tile (flag | segments << 31 | children): the tile publishes its aggregate as
// soon as its trace is done, then reads the descriptors of its predecessors 64 at a time until it meets one that
// already holds an inclusive prefix, publishes its own inclusive prefix and writes.  Tiles are numbered by workgroup and
// wave and workgroups are dispatched in index order, so every predecessor of a running tile is itself running or finished: no
// wait can last longer than a predecessor's trace, and nothing is serial per workgroup — the difference to round 1's
// single-pass kernel (persistent workgroups that loaded, traced, looked back and stored tile after tile: 26.6 ms on cfg 4
// with reflectivity 0.2).  Against the two passes of k_gen_pass a generation's rays are read and traced ONCE (375 instead of
// 484 bytes per processed ray in double precision), there is no scan, no totals kernel, no code byte, no stored decision,
// and count and emit cannot disagree because there is only one of them.
// Budgets (optical_table.py:138-144).  The two-pass kernels read budget[tree] in the count pass and let the tree's last ray
// write it in the emit pass; in one pass that is a race (a tree's rays can span workgroups).  Here every RAY carries what is
// left of its tree's budget at the start of the generation (rem[], 4 bytes per ray: seeded from budget[] by k_gen_seed_rem
// before the first generation of a call, handed to the children as rem - min(rays of the tree in this generation, rem));
// budget[] itself is only WRITTEN (by the tree's last ray) and stays what callers read between calls.
// Scenes with count-limited leaves keep the two-pass path (their gate needs the per-slot scans between probe and trace).
static constexpr unsigned long long GEN1_AGG = 1ull << 62, GEN1_INC = 2ull << 62, GEN1_MASK = (1ull << 62) - 1ull;
// index one past the last ray of ray i's tree (tree[] is non-decreasing): neighbours first, then a binary search
__device__ __forceinline__ int64_t tree_tail(const int32_t* __restrict__ tree, int64_t i, int64_t n) {
    const int32_t t = tree[i];
    for (int k = 1; k <= 3; ++k)
        if (i + k >= n || tree[i + k] != t) return i + k;
    int64_t lo = i + 3, hi = n;  // tree[lo] == t; first index in (lo, n] whose tree differs
    while (lo + 1 < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (tree[mid] == t) lo = mid; else hi = mid;
    }
    return hi;
}
template <class T, uint32_t F> constexpr int gen_one_minw() { return (sizeof(T) == 8 && F == (F_AABB | F_LENS | F_REFRACT)) ? 4 : 1; }  // (4 waves per SIMD where 128 registers suffice)
template <class T, uint32_t F, bool SCENE_IN_LDS>
__global__ __launch_bounds__(256, (gen_one_minw<T, F>())) void k_gen_one(SceneBlob blob, T unit, RaysT<T> in, const int32_t* __restrict__ tree, const int32_t* __restrict__ rem,
                                                 int64_t n, int32_t* __restrict__ budget, int64_t* state, SegsT<T> out, int64_t out_capacity,
                                                 RaysOutT<T> next, int32_t* next_tree, int32_t* next_rem, int64_t next_capacity,
                                                 unsigned long long* desc, uint32_t* ticket, int32_t* counts, int32_t n_classes, int32_t drop_doomed,
                                                 const int64_t* n_in, int64_t* n_out) {
    // CHAINED launches (small ray trees: ot_trace_tree_* enqueues several generations before it reads anything back): the
    // generation's size is not known on the host when the launch is enqueued — the grid covers an upper bound (`n`), the real
    // size comes from where the generation before left it (*n_in) and this one leaves its own in *n_out (two words used in
    // turn, so that a late wave of this launch never reads what its last tile has just written).
    if (n_in) {
        const int64_t n_real = *n_in;
        if (n_real <= 0) {  // the trees ended generations ago: nothing to trace, nothing follows
            if (blockIdx.x == 0 && threadIdx.x == 0) { state[1] = 0; *n_out = 0; }
            return;
        }
        n = n_real < n ? n_real : n;
    }
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t* base = blob.words;
    if (SCENE_IN_LDS) {
        for (int w = threadIdx.x; w < blob.n_words; w += blockDim.x) lds[w] = blob.words[w];
        __syncthreads();  // the only workgroup barrier: the scene image is staged; from here on the four waves are on their own
        base = lds;
    }
    const Scene<T> sc = bind_scene<T>(base, blob, unit);
    const int lane = threadIdx.x & 63;
    // A tile is the 64 rays of ONE WAVE (numbered by workgroup and wave: workgroups are dispatched in index order, so every
    // predecessor of a running wave is running or done).  With a tile per workgroup, three waves sat at a barrier while
    // the fourth looked back (cfg 4 R = 0.2: 17.7 ms against 14.7 for the two passes); per wave nothing ever waits for a
    // neighbour and a CU has sixteen independent look-backs in flight.
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tile = i >> 6;
    // (the segment cursor as it stands BEFORE this generation: the last tile rewrites it at the very end, which it can only do
    // after every other tile has published — so the value is pinned here, ahead of anything this tile publishes)
    const int64_t cursor = (int64_t)__hip_atomic_load(reinterpret_cast<unsigned long long*>(state), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"((int32_t)cursor), "v"((int32_t)(cursor >> 32)) : "memory");
    // rank within the tree, budget cut, doomed children: as k_gen_pass, with the ray's own rem[] in place of budget[tree]
    int32_t my_tree = -1;
    int start_lane = -1;
    if (i < n) {
        my_tree = tree[i];
        if (lane == 0 || tree[i - 1] != my_tree) start_lane = lane;
    }
    const int head_lane = wave_incl_max_i32(start_lane);
    int64_t head = (i - lane) + head_lane;
    if (i < n && head_lane == 0) head = tree_head(tree, i - lane);
    const int64_t bud = i < n ? (int64_t)rem[i] : 0;
    const bool active = i < n && (i - head) < bud;
    bool doomed = false;
    if (active && drop_doomed) {
        const int64_t last_needed = head + bud - 1;
        doomed = last_needed < n && tree[last_needed] == my_tree;
    }
    RayState<T> r = {};
    int32_t cls = 0, fl = 0;
    if (active) {
        fl = in.flags[i];
        r = load_ray(in, i, fl);
        if ((uint32_t)r.last >= (uint32_t)sc.n_nodes) r.last = -1;
        cls = in.id[i];
    }
    const bool dead = active && (fl & OT_RAY_DEAD);
    const GateCtx gate = {counts, n_classes, cls, nullptr, nullptr, n, i};
    const Hit<T> h = nearest_hit<T, F, GATE_TABLE>(sc, r, active && !dead, gate);
    // Double precision: two children are 2 x 11 doubles = 44 registers that would sit through the barrier and the look-back (133
    // registers, 3 waves per SIMD).  The interaction is therefore run for its COUNT here (its unused results are dead code) and
    // again for the children after the look-back, behind an opaque copy of one input so that the two calls are not merged:
    // 124 registers, 4 waves per SIMD, ~5 % more arithmetic in a pass that is bound by its streams.
    constexpr bool LATE_KIDS = sizeof(T) == 8;
    int32_t nk = 0;
    RayState<T> ch[2];  // indexed by constants only
    if (active && !dead && h.node >= 0) {
        if constexpr (LATE_KIDS) {
            RayState<T> unused[2];
            nk = interact<T, F, 2>(sc, r, h, unused, make_matcache<T, F>(sc, r.wl));
        } else {
            nk = interact<T, F, 2>(sc, r, h, ch, make_matcache<T, F>(sc, r.wl));
        }
    }
    if (doomed) nk = 0;
    // what the tree has left after this generation: for the children, and (its last ray) for budget[]
    int32_t rem_after = 0;
    if (i < n && (nk > 0 || i == n - 1 || tree[i + 1] != my_tree)) {
        const int64_t in_gen = tree_tail(tree, i, n) - head;
        rem_after = (int32_t)(bud - (in_gen < bud ? in_gen : bud));
        if (i == n - 1 || tree[i + 1] != my_tree) budget[my_tree] = rem_after;  // one writer per tree, no reader in this kernel
    }
    // ---- the tile's aggregate
    const unsigned long long act_mask = __ballot(active);
    const int seg_rank = rank_below(act_mask), n_act = __popcll(act_mask);
    int kids_total;
    const int kid_excl = wave_excl_scan_i32(nk, kids_total);
    const unsigned long long agg = ((unsigned long long)n_act << 31) | (unsigned long long)kids_total;
    // ---- decoupled look-back: exclusive prefix over the tiles before this one
    unsigned long long excl = 0;
    if (tile * 64 < n) {  // (wave-uniform; the waves behind the end of the generation have nothing to say)
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(&desc[tile], GEN1_AGG | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t pos = tile - 1;  // nearest predecessor not yet accounted for
            for (;;) {
                const int64_t j = pos - lane;
                const unsigned long long d = j >= 0 ? __hip_atomic_load(&desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : GEN1_INC;  // (before tile 0: an inclusive zero)
                const unsigned long long m_inc = __ballot((d >> 62) == 2ull), m_not = __ballot((d >> 62) == 0ull);
                const int first_inc = m_inc ? __builtin_ctzll(m_inc) : 64, first_not = m_not ? __builtin_ctzll(m_not) : 64;
                if (first_not < first_inc) {  // a predecessor in the window has not published yet: it is tracing (it was dispatched before this wave)
                    __builtin_amdgcn_s_sleep(4);
                    continue;
                }
                unsigned long long v = lane <= first_inc ? (d & GEN1_MASK) : 0ull;  // aggregates up to the first inclusive prefix, and that prefix
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                excl += v;
                if (first_inc < 64) break;
                pos -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&desc[tile], GEN1_INC | (excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((tile + 1) * 64 >= n) {  // the last tile: the generation's totals (every other tile has read the cursor: it published before this one could finish)
                const unsigned long long all = excl + agg;
                state[0] = cursor + (int64_t)(all >> 31);
                state[1] = (int64_t)(all & 0x7fffffffull);
                if (n_out) *n_out = (int64_t)(all & 0x7fffffffull);
            }
        }
    }
    const unsigned long long mine = excl;
    // ---- write: segment record and children, once, in stable order
    if (active) {
        const int64_t slot = cursor + (int64_t)(mine >> 31) + seg_rank;
        if (slot < out_capacity) {
            if (dead) store_segment<T, GEN_NT>(out, slot, r, r.len, my_tree, -2);
            else if (h.node < 0) store_segment<T, GEN_NT>(out, slot, r, r.len, my_tree, -1);
            else store_segment<T, GEN_NT>(out, slot, r, h.t, my_tree, leaf_id_of<T, F>(sc, h.node));
        }
    }
    const int64_t d0 = (int64_t)(mine & 0x7fffffffull) + kid_excl;
    if constexpr (LATE_KIDS) {
        if (nk > 0) {
            RayState<T> r2 = r;
            asm volatile("" : "+v"(r2.dx));  // (the same value, opaque to the optimiser: this call is not the one above)
            (void)interact<T, F, 2>(sc, r2, h, ch, make_matcache<T, F>(sc, r2.wl));
        }
    }
    auto put = [&](const RayState<T>& k, int64_t d) {
        if (d >= next_capacity) return;
        st<GEN_NT>(next.ox + d, k.ox); st<GEN_NT>(next.oy + d, k.oy); st<GEN_NT>(next.oz + d, k.oz);
        st<GEN_NT>(next.dx + d, k.dx); st<GEN_NT>(next.dy + d, k.dy); st<GEN_NT>(next.dz + d, k.dz);
        st<GEN_NT>(next.wl + d, k.wl); st<GEN_NT>(next.qr + d, k.qr); st<GEN_NT>(next.qi + d, k.qi);
        st<GEN_NT>(next.I + d, k.I); st<GEN_NT>(next.n + d, k.n); st<GEN_NT>(next.pl + d, k.pl);
        next.flags[d] = (fl & OT_RAY_HAS_Q) | ((k.last + 1) << 8);
        next.id[d] = cls;
        next_tree[d] = my_tree;
        next_rem[d] = rem_after;
    };
    if (nk > 0) put(ch[0], d0);
    if (nk > 1) put(ch[1], d0 + 1);
}
