// inst_refill.hip — the k_trace_refill instantiations of one precision and one output layout and their lookup
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

// fr as for rolling_kernel (0 FR, 1 FC, 3 F_ALL, 4 FRP); flat = the pair-queue walk (FR / FRP).  Non-temporal stores as for the
// lists: the sparse [k][ray] slots want plain stores (partial lines merge in L2), the dense append list streams.
template <> RollingKern<T, OUT> refill_kernel<T, OUT>(int fr, bool flat) {
    constexpr bool NT = APPEND;
    if (flat) {
        if (fr == 0) return k_trace_refill<T, FR | F_FLAT, NT, OUT>;
        if (fr == 4) return k_trace_refill<T, FRP | F_FLAT, NT, OUT>;
        return nullptr;
    }
    if (fr == 1) return k_trace_refill<T, FC, NT, OUT>;
    if (fr == 3) return k_trace_refill<T, F_ALL, true, OUT>;
    return nullptr;
}
#if !OT_APPEND
template <> int refill_max_threads<T>(int fr, bool flat) {
    if (flat) return refill_threads<T, FR | F_FLAT>();
    return fr == 1 ? refill_threads<T, FC>() : refill_threads<T, F_ALL>();
}
#endif
