// inst_rolling.hip — the k_trace_rolling instantiations of one precision and one output layout and their lookup
// (-DOT_REAL=double | float, -DOT_APPEND=0 | 1).
#include "tables.h"

using T = OT_REAL;
using namespace preset;
#ifndef OT_APPEND
#define OT_APPEND 0
#endif
#if OT_APPEND
using OUT = SegPlanes<T>;
#else
using OUT = SegsT<T>;
#endif
constexpr bool APPEND = OT_APPEND != 0;

// records in LDS: single precision only (a double-precision record is 100 bytes: too few waves would fit)
template <uint32_t FM, bool NT, bool REC_OK> static RollingKern<T, OUT> pick(bool lds, bool rec_lds) {
    if (rec_lds) {
        if constexpr (REC_OK && sizeof(T) == 4) return lds ? k_trace_rolling<T, FM, true, NT, true, OUT> : nullptr;
        else return nullptr;
    }
    if (lds) return k_trace_rolling<T, FM, true, NT, false, OUT>;
    return nullptr;
}
template <> RollingKern<T, OUT> rolling_kernel<T, OUT>(int fr, bool flat, bool lds, bool rec_lds) {
    // [k][ray] slots written by mixed lists (the presets under a top-level grid): plain stores; everything else non-temporal
    constexpr bool NT_GRID = APPEND;
    if (flat) {
        if (fr == 0) return pick<FR | F_FLAT, NT_GRID, true>(lds, rec_lds);
        if (fr == 4) return pick<FRP | F_FLAT, NT_GRID, true>(lds, rec_lds);
        return nullptr;
    }
    switch (fr) {
        case 0: return pick<FR, NT_GRID, false>(lds, rec_lds);
        case 4: return pick<FRP, NT_GRID, false>(lds, rec_lds);
        case 1: return pick<FC, NT_GRID, false>(lds, rec_lds);
        case 2: return pick<FD, true, true>(lds, rec_lds);
        default:
            if (rec_lds) return nullptr;
            return lds ? k_trace_rolling<T, F_ALL, true, true, false, OUT> : k_trace_rolling<T, F_ALL, false, true, false, OUT>;
    }
}
template <> RollingKern<T, OUT> pool_kernel<T, OUT>(int fr) {
    if constexpr (sizeof(T) == 4) {
        if (fr == 2) return k_trace_pool<T, FD, true, OUT>;
    }
    return nullptr;
}
#if !OT_APPEND
template <> int rolling_max_threads<T>(int fr, bool flat, bool rec_lds) {
    if (flat) return rec_lds ? rolling_threads<T, FR | F_FLAT, true>() : rolling_threads<T, FR | F_FLAT, false>();
    switch (fr) {
        case 0: return rolling_threads<T, FR, false>();
        case 4: return rolling_threads<T, FRP, false>();
        case 1: return rolling_threads<T, FC, false>();
        case 2: return rolling_threads<T, FD, false>();
        default: return rolling_threads<T, F_ALL, false>();
    }
}
#endif
