// tables.h — which kernel instantiations exist, and the lookups the launch code (optable_hip.hip) uses to get them.
// The instantiations are compiled in their own translation units, one per kernel family and precision
// (inst_fused.hip, inst_rolling.hip, inst_gen.hip, each built with -DOT_REAL=double and -DOT_REAL=float), so the
// library builds in parallel and the launch code only ever sees function pointers.  A lookup returns nullptr for a
// combination that is not compiled; the launch code never asks for one.
#pragma once

#include "kernels.h"

// scene feature presets (trace_core.h feature mask): the launch code picks the smallest one that covers the scene
namespace preset {
constexpr uint32_t FA = F_AABB | F_LENS;                       // mirrors + thin lenses (cfg 2)
constexpr uint32_t FB = FA | F_REFRACT;                        // + Snell interfaces, materials (cfg 4)
constexpr uint32_t FC = FB | F_GRID | F_ROOT | F_SUBTREE;      // planar scenes with gridded groups / subtrees under a top-level grid
constexpr uint32_t FR = FB | F_ROOT;                           // planar scenes under a top-level grid that lists leaves only (cfg 3)
constexpr uint32_t FRP = FR | F_POLY;                          // ... with polygon / boolean apertures
constexpr uint32_t FD = F_AABB | F_REFRACT | F_CURVED | F_GRID;  // spherical / aspheric optics in gridded groups (cfg 5)
// light scenes made of the reference's everyday parts — what used to fall into the all-features preset (169-181 registers and
// 146-186 scalar spills in single precision, no lane-per-ray form at all in double):
constexpr uint32_t FE = FB | F_POLY | F_CURVED | F_LIMIT;      // + polygon / boolean apertures, spheres and aspheres, count gates: TriangularPrism,
                                                               //   Block(hole=), BiConvexLens, doublets (fp32 118 registers, fp64 186)
constexpr uint32_t FM = FE | F_MISC;                           // + cylinders, polygons in tilted planes (DovePrism), sag / index series (138 / 244)
}  // namespace preset

template <class T, class OUT>
using FusedKern = void (*)(SceneBlob, T, RaysT<T>, int64_t, int32_t, OUT, int32_t*, int32_t*, int32_t, int32_t);
template <class T, class OUT>
using RollingKern = void (*)(SceneBlob, T, RaysT<T>, int64_t, int32_t, OUT, AppendCtl, int32_t*, int32_t*, int32_t, WaveScratch<T>, int32_t,
                             int32_t, unsigned long long*, int32_t, int32_t);
template <class T>
using GenKern = void (*)(SceneBlob, T, RaysT<T>, const int32_t*, int64_t, int32_t*, const int64_t*, SegsT<T>, int64_t, RaysOutT<T>, int32_t*,
                         int64_t, uint8_t*, unsigned long long*, const unsigned long long*, int32_t*, int32_t, const int32_t*,
                         unsigned long long*, int32_t*, T*, int32_t, uint8_t*);
template <class T, class OUT>
using TreeKern = void (*)(SceneBlob, T, RaysT<T>, int64_t, int32_t, int32_t, int32_t, uint8_t*, OUT, AppendCtl, int32_t*, int32_t*, int32_t, int32_t, int32_t);
template <class T>
using GenOneKern = void (*)(SceneBlob, T, RaysT<T>, const int32_t*, const int32_t*, int64_t, int32_t*, int64_t*, SegsT<T>, int64_t, RaysOutT<T>, int32_t*,
                            int32_t*, int64_t, unsigned long long*, uint32_t*, int32_t*, int32_t, int32_t, const int64_t*, int64_t*);
template <class T>
using ProbeKern = void (*)(SceneBlob, T, RaysT<T>, const int32_t*, int64_t, const int32_t*, int32_t*, int32_t, int32_t*);

// k_trace_fused: fi = 0 FA, 1 FB, 2 FE, 3 FM, 4 F_ALL (single precision only); image in LDS; 128-register cap (4 waves per SIMD); non-temporal segment stores
template <class T, class OUT> FusedKern<T, OUT> fused_kernel(int fi, bool lds, bool minw4, bool nt);
// k_trace_rolling: fr = 0 FR, 1 FC, 2 FD, 3 F_ALL, 4 FRP; flat = the pair-queue walk (FR / FRP only); image in LDS or read
// from L2 (all-features preset only); records of the live rays in LDS (fp32: pair queue and FD) or in the global scratch.
// OUT = SegsT<T> ([k][ray] slots) or SegPlanes<T> (append layout).  Non-temporal segment stores are part of the choice:
// the sparse [k][ray] slots of mixed lists want plain stores (partial lines merge in L2), everything else streams.
template <class T, class OUT> RollingKern<T, OUT> rolling_kernel(int fr, bool flat, bool lds, bool rec_lds);
template <class T> int rolling_max_threads(int fr, bool flat, bool rec_lds);
// k_trace_pool (same arguments; CAP carries the number of blocks): the curved-surface preset FD in single precision, else nullptr
template <class T, class OUT> RollingKern<T, OUT> pool_kernel(int fr);
// k_trace_refill (same arguments; CAP carries the rays per ticket): mixed scenes, rays in registers.  fr / flat as above;
// nullptr where no instantiation exists (the lists take the scene then)
template <class T, class OUT> RollingKern<T, OUT> refill_kernel(int fr, bool flat);
template <class T> int refill_max_threads(int fr, bool flat);
// k_gen_pass / k_gen_probe: fg = 0 the planar preset FB, 1 FC (planar scenes under grids: cfg 3 with splitting slabs), 2 FE, 3 FM, 4 F_ALL
template <class T> GenKern<T> gen_kernel(int fg, bool lds, bool emit);
// the emit pass that also counts the children's children (k_gen_pass MODE 2); nullptr where no instantiation exists
template <class T> GenKern<T> gen_ahead_kernel(int fg, bool lds);
template <class T> ProbeKern<T> probe_kernel(int fg, bool lds);
// k_trace_trees (a lane per tree, the FIFO in LDS): fg as above; nullptr where no instantiation exists
template <class T, class OUT> TreeKern<T, OUT> tree_kernel(int fg, int img = 1);  // fg 5 / 6: the planar presets FR / FRP with the wave-wide pair queue (F_FLAT); img: 1 whole image in LDS, 2 node records in LDS + tables in global memory, 0 all global (2 / 0: fg 4, append output)
// ... and the waves per SIMD its registers are capped for = the workgroups per CU it can have (256 threads: one wave per SIMD
// each).  With every child queued the moment the interaction has formed it the kernels need 71-74 registers in single precision
// (FB; FC / FE 97, FM 111) and 124-130 in double (FE 165, FM 255): the queues' LDS decides, not the registers.
template <class T> constexpr int tree_minw(int fg) { return sizeof(T) == 4 ? (fg == 0 ? 6 : (fg == 4 ? 3 : 4)) : (fg == 4 ? 1 : (fg == 3 ? 2 : 3)); }
template <class T> constexpr int tree_groups_by_registers(int fg) { return tree_minw<T>(fg); }
// k_gen_one (one pass per generation, decoupled look-back): fg as above; nullptr where no instantiation exists
template <class T> GenOneKern<T> gen_one_kernel(int fg, bool lds);

#define OT_DECLARE_TABLES(T)                                                       \
    template <> FusedKern<T, SegsT<T>> fused_kernel<T, SegsT<T>>(int, bool, bool, bool);         \
    template <> FusedKern<T, SegTiles<T>> fused_kernel<T, SegTiles<T>>(int, bool, bool, bool);   \
    template <> RollingKern<T, SegsT<T>> rolling_kernel<T, SegsT<T>>(int, bool, bool, bool);          \
    template <> RollingKern<T, SegPlanes<T>> rolling_kernel<T, SegPlanes<T>>(int, bool, bool, bool);  \
    template <> int rolling_max_threads<T>(int, bool, bool);                       \
    template <> RollingKern<T, SegsT<T>> pool_kernel<T, SegsT<T>>(int);            \
    template <> RollingKern<T, SegPlanes<T>> pool_kernel<T, SegPlanes<T>>(int);    \
    template <> RollingKern<T, SegsT<T>> refill_kernel<T, SegsT<T>>(int, bool);       \
    template <> RollingKern<T, SegPlanes<T>> refill_kernel<T, SegPlanes<T>>(int, bool); \
    template <> int refill_max_threads<T>(int, bool);                              \
    template <> GenKern<T> gen_kernel<T>(int, bool, bool);                         \
    template <> GenKern<T> gen_ahead_kernel<T>(int, bool);                         \
    template <> ProbeKern<T> probe_kernel<T>(int, bool);                          \
    template <> TreeKern<T, SegsT<T>> tree_kernel<T, SegsT<T>>(int, int);              \
    template <> TreeKern<T, SegPlanes<T>> tree_kernel<T, SegPlanes<T>>(int, int);      \
    template <> GenOneKern<T> gen_one_kernel<T>(int, bool);
OT_DECLARE_TABLES(double)
OT_DECLARE_TABLES(float)
#undef OT_DECLARE_TABLES
