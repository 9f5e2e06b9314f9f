// inst_fused.hip — the k_trace_fused instantiations of one precision (-DOT_REAL=double | float) and their lookup.
#include "tables.h"

using T = OT_REAL;
using namespace preset;

// The 128-register cap (MINW = 4) pays for the mirror / lens kernel and for the fp32 Snell kernel; the fp64 Snell kernel
// wants 145 registers and would spill: that combination is not compiled.  FE / FM (everyday parts: polygons, spheres, aspheres,
// count gates; + the rarer shapes) exist in both precisions; the all-features kernel in single precision only (in double it
// needs more than 256 registers; those scenes take the rolling lists), with non-temporal stores only, and is also the one
// that reads images beyond the LDS limit from L2.  Every kernel in two output layouts:
// the [k][ray] arrays of ot_trace_* and the 64-slot tiles of ot_trace_tiled_*.
template <uint32_t FM, int W, bool N, class OUT> static FusedKern<T, OUT> one() {
    if constexpr (W == 4 && sizeof(T) == 8 && (FM & F_REFRACT) != 0) return nullptr;
    else return k_trace_fused<T, FM, true, W, N, OUT>;
}
template <uint32_t FM, class OUT> static FusedKern<T, OUT> pick(bool minw4, bool nt) {
    return minw4 ? (nt ? one<FM, 4, true, OUT>() : one<FM, 4, false, OUT>()) : (nt ? one<FM, 1, true, OUT>() : one<FM, 1, false, OUT>());
}
template <class OUT> static FusedKern<T, OUT> lookup(int fi, bool lds, bool minw4, bool nt) {
    if (lds && fi == 0) return pick<FA, OUT>(minw4, nt);
    if (lds && fi == 1) return pick<FB, OUT>(minw4, nt);
    if (lds && fi == 2) return k_trace_fused<T, FE, true, 1, true, OUT>;  // (non-temporal stores only, like the all-features kernel)
    if (lds && fi == 3) return k_trace_fused<T, FM, true, 1, true, OUT>;
    if constexpr (sizeof(T) == 4) return lds ? k_trace_fused<T, F_ALL, true, 1, true, OUT> : k_trace_fused<T, F_ALL, false, 1, true, OUT>;
    else return nullptr;
}
template <> FusedKern<T, SegsT<T>> fused_kernel<T, SegsT<T>>(int fi, bool lds, bool minw4, bool nt) { return lookup<SegsT<T>>(fi, lds, minw4, nt); }
template <> FusedKern<T, SegTiles<T>> fused_kernel<T, SegTiles<T>>(int fi, bool lds, bool minw4, bool nt) { return lookup<SegTiles<T>>(fi, lds, minw4, nt); }
