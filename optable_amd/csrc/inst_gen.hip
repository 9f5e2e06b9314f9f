// inst_gen.hip — the generation kernels (k_gen_pass count / emit, k_gen_probe) of one precision and their lookup.
#include "tables.h"

using T = OT_REAL;
using namespace preset;

template <uint32_t FM> static GenKern<T> pick(bool lds, bool emit) {
    if (lds) return emit ? k_gen_pass<T, FM, true, 1> : k_gen_pass<T, FM, true, 0>;
    return emit ? k_gen_pass<T, FM, false, 1> : k_gen_pass<T, FM, false, 0>;
}
// beam splitters and partially reflecting slabs are planar scenes: they get the small instantiation (145 instead of 255
// registers in fp64); count gates, curved shapes and polygons need the full one
// ... and planar scenes big enough for grids (a top-level grid, gridded groups, subtrees) the planar preset with the grid
// walks: the all-features kernel spends registers and branches on curved shapes, polygons and count gates they do not have
template <> GenKern<T> gen_kernel<T>(int fg, bool lds, bool emit) {
    if (fg == 0) return pick<FB>(lds, emit);
    if constexpr (sizeof(T) == 4) {  // (single precision, image in LDS: the combinations whose count / emit pair compiles without a stack slot)
        if (fg == 1 && lds) return emit ? k_gen_pass<T, FC, true, 1> : k_gen_pass<T, FC, true, 0>;
    }
    // the everyday parts (count gates, polygons, spheres, aspheres; + the rarer shapes): 94 / 114 registers in single precision
    // (the all-features emit kernel: 150-161), 148 / 236 in double (255-262); image in LDS only
    if (fg == 2 && lds) return emit ? k_gen_pass<T, FE, true, 1> : k_gen_pass<T, FE, true, 0>;
    if (fg == 3 && lds) return emit ? k_gen_pass<T, FM, true, 1> : k_gen_pass<T, FM, true, 0>;
    return pick<F_ALL>(lds, emit);
}
// the emit pass with look-ahead (the next generation needs no count pass): light planar scenes whose image is in LDS
template <> GenKern<T> gen_ahead_kernel<T>(int fg, bool lds) {
    if (fg == 0 && lds) return k_gen_pass<T, FB, true, 2>;
    return nullptr;
}
template <> ProbeKern<T> probe_kernel<T>(int fg, bool lds) {
    if (fg == 2 && lds) return k_gen_probe<T, FE, true>;
    if (fg == 3 && lds) return k_gen_probe<T, FM, true>;
    return lds ? k_gen_probe<T, F_ALL, true> : k_gen_probe<T, F_ALL, false>;
}

// k_gen_one (opt-in, OT_OPT_GEN_ONEPASS): the planar presets; nullptr elsewhere (the two passes take the generation)
template <> GenOneKern<T> gen_one_kernel<T>(int fg, bool lds) {
    if (fg == 0) return lds ? k_gen_one<T, FB, true> : k_gen_one<T, FB, false>;
    if constexpr (sizeof(T) == 4) {
        if (fg == 1 && lds) return k_gen_one<T, FC, true>;
    }
    return nullptr;
}

// k_trace_trees: every preset (the all-features one included: 148 registers in single precision, 263 in double — one wave per SIMD,
// as its generation kernels); 2 waves per SIMD in double precision (the rarer shapes: 1 — 256 registers would spill), 3 in single: the register
// caps the queues' LDS leaves room for (tables.h tree_minw)
template <class OUT> static TreeKern<T, OUT> pick_tree(int fg, int img) {
    if (img != 1) {  // scenes no LDS holds: the all-features preset, dense list; img 2 = node records in LDS, tables in global memory; 0 = all global
        if constexpr (std::is_same<OUT, SegPlanes<T>>::value) {
            if (fg == 4 && img == 2) return k_trace_trees<T, F_ALL, tree_minw<T>(4), OUT, 2>;
            if (fg == 4 && img == 0) return k_trace_trees<T, F_ALL, tree_minw<T>(4), OUT, 0>;
        }
        return nullptr;
    }
    if (fg == 0) return k_trace_trees<T, FB, tree_minw<T>(0), OUT>;
    if constexpr (std::is_same<OUT, SegPlanes<T>>::value) {  // (the [k][tree] slots: the planar preset only — every preset writes the dense list)
        if (fg == 1) return k_trace_trees<T, FC, tree_minw<T>(1), OUT>;
        if (fg == 2) return k_trace_trees<T, FE, tree_minw<T>(2), OUT>;
        if (fg == 3) return k_trace_trees<T, FM, tree_minw<T>(3), OUT>;
        if (fg == 4) return k_trace_trees<T, F_ALL, tree_minw<T>(4), OUT>;  // grids AND curved / exotic shapes: every scene has a lane-per-tree kernel
        if (fg == 5) return k_trace_trees<T, FR | F_FLAT, tree_minw<T>(5), OUT>;   // planar scenes under a top-level grid of leaves: the pair queue
        if (fg == 6) return k_trace_trees<T, FRP | F_FLAT, tree_minw<T>(6), OUT>;  // ... with polygon / boolean apertures
    }
    return nullptr;
}
template <> TreeKern<T, SegsT<T>> tree_kernel<T, SegsT<T>>(int fg, int img) { return pick_tree<SegsT<T>>(fg, img); }
template <> TreeKern<T, SegPlanes<T>> tree_kernel<T, SegPlanes<T>>(int fg, int img) { return pick_tree<SegPlanes<T>>(fg, img); }
