// inst_refill.hip — the k_trace_refill instantiations of one precision and their lookup (-DOT_REAL=double | float).
#include "tables.h"

using T = OT_REAL;
using namespace preset;

// fr as for rolling_kernel (0 FR, 4 FRP); the pair-queue walk only, and only the append layout: into the sparse [k][ray] slots the
// lists are faster (3.7 against 4.7 ms on cfg 3), and an opt-in kernel that ties with the default is not worth twelve more
// instantiations.  nullptr: the lists take the launch.
template <> RollingKern<T, SegPlanes<T>> refill_kernel<T, SegPlanes<T>>(int fr, bool flat) {
    if (flat && fr == 0) return k_trace_refill<T, FR | F_FLAT, true, SegPlanes<T>>;
    if (flat && fr == 4) return k_trace_refill<T, FRP | F_FLAT, true, SegPlanes<T>>;
    return nullptr;
}
template <> RollingKern<T, SegsT<T>> refill_kernel<T, SegsT<T>>(int, bool) { return nullptr; }
template <> int refill_max_threads<T>(int, bool) { return refill_threads<T, FR | F_FLAT>(); }
