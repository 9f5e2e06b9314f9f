// trace_core.h — device-side ray/scene arithmetic for the gfx950 trace kernels.
//
// One lane owns one ray.  The scene (skip-list of nodes, materials, aux records) is staged
// into LDS once per workgroup; the plain pass reads it with wave-uniform indices (every lane the
// same LDS words: broadcast, no bank conflicts), the grid walks with per-lane candidates.  The
// per-ray state lives in VGPRs.  Templated on the real type (double / float) and on a scene FEATURE MASK: the host
// looks at what the uploaded scene contains and launches the smallest instantiation that covers
// it, so a mirrors-and-lenses scene does not carry the registers of the asphere root solver.
//
// Semantics follow the reference (tim4431/optable); the arithmetic is organised differently:
//   nearest_hit  — one forward pass over the depth-first node list with a per-lane
//                  `skip_until` and a strict `<`: equals ComponentGroup.interact's AABB prune +
//                  np.argmin and the table loop's first-strict-minimum
//                  (component_group.py:93-122, optical_table.py:119-123).
//   hit_leaf     — OpticalComponent.intersect_point_local (optical_component.py:151-233); the
//                  reference's 10-point sign scan is kept bracket for bracket, the root inside a
//                  bracket is polished by a safeguarded Newton iteration instead of brentq; for
//                  spheres the same search in closed form (two line-sphere roots + the scan's
//                  "exactly one root per sample interval" rule).
//   grid_children, root_grid_hit — acceleration only: 2-D grids over a large group's children and
//                  over the top-level components; every candidate found through them still gets the
//                  tests the plain pass applies, ties go to the lower node index.
//   interact     — BaseMirror / BaseRefraciveSurface / Lens .interact_local
//                  (optical_component.py:536-570, 617-717, 930-948).
// Deliberate differences in rounding only (all far inside the 1e-6 parity tolerance):
//   * redundant renormalisations of already-unit vectors (Ray.direction setter on every copy)
//     are not repeated per surface test; each child direction is normalised once;
//   * 1/d per axis is computed once per segment and shared by every AABB slab test (the
//     reference also forms 1/d, solver.py:34, but per box);
//   * x/f is evaluated as x*(1/f) with 1/f formed on the host; |P| <= r as |P|^2 <= r^2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/optable_hip.h"

namespace ot {

// scene feature mask
enum : uint32_t {
    F_POLY = 1,      // polygon / boolean apertures
    F_CURVED = 2,    // sphere, asphere, cylinder, tilted polygon: bracketed root solve
    F_REFRACT = 4,   // Snell interfaces (and materials)
    F_LENS = 8,      // thin lenses
    F_AABB = 16,     // groups: AABB prune
    F_LIMIT = 32,    // max_interact_count gates
    F_GRID = 64,     // large groups carry a 2-D grid over their children's boxes
    F_ROOT = 128,    // the top-level component list carries a 2-D grid walked cell by cell (DDA)
    F_MISC = 256,    // the rarer curved shapes: cylinder walls, polygons in a tilted plane (needs F_CURVED)
    F_SUBTREE = 512, // cells of the top-level grid may list whole groups (stale boxes, gridded groups) besides leaves
    F_ALL = 1023,
    F_FLAT = 1024    // (not part of F_ALL) planar scenes under a top-level grid: candidates go through a wave-wide
                     // queue of (ray, leaf) pairs and are tested with full lanes (flat_grid_hit)
};

template <class T> struct Num;
template <> struct Num<double> {
    static __device__ __forceinline__ double inf() { return __builtin_inf(); }
    static __device__ __forceinline__ double eps_t() { return 1e-9; }  // EPS, optical_component.py:163
    static __device__ __forceinline__ double root_tol() { return 4.4e-16; }
    static __device__ __forceinline__ double tiny() { return 1e-300; }
};
template <> struct Num<float> {
    static __device__ __forceinline__ float inf() { return __builtin_inff(); }
    // 1e-9 is below fp32 resolution at unit scale; the fp32 self-hit guard is 1e-5
    static __device__ __forceinline__ float eps_t() { return 1e-5f; }
    static __device__ __forceinline__ float root_tol() { return 2.4e-7f; }
    static __device__ __forceinline__ float tiny() { return 1e-37f; }
};

// Device copy of ot_node in the kernel's real type (built by the host in ot_scene_upload).
// Record stride: 53 dwords in fp32, 94 in fp64.  When 64 lanes read the same field of 64 DIFFERENT nodes (the grid walks),
// the LDS bank of lane l is (stride * node_l + field) mod 32: an odd stride (53) sends different nodes to different banks,
// where the natural 52 dwords (a multiple of 4) left only 8 distinct banks — 21 % of the LDS cycles of cfg 3 were bank
// conflicts, 4 % with the pad (tools/lds_probe.sh); cfg 3 +1 %, cfg 5 fp32 19.65 -> 19.2 ms.
// flags of a device record beyond ot_node_flags (set at upload, optable_hip.hip fill_blob)
enum : int32_t {
    DN_CONVEX_POS = 256,  // asphere: g(t) = x + F(r) is convex along every line inside the local box (F convex and non-decreasing there)
    DN_CONVEX_NEG = 512   // ... -g is
};
template <class T> struct DNode {
    T M[9];
    T org[3];
    T aabb[6];
    T lbox[6];
    T p[8];
    T refl, trans, focal, roc;
    T inv_focal, r2, rad2, pad1;  // 1/focal_length; circle radius^2 or cap aperture^2; sphere/cylinder R^2
    int32_t kind, end, flags, shape, inter, mat1, mat2, roc_kind, max_count, slot, aux, leaf_id;
    int32_t bank_pad[sizeof(T) == 4 ? 1 : 2];
};
template <class T> struct DMat {
    T n;
    T B[3];
    T C[3];
    int32_t kind, pad;
};

template <class T> struct Scene {
    const DNode<T>* nodes;  // PHYSICAL records: every node once, except that a run of instanced leaves keeps one prototype
    const DMat<T>* mats;
    const T* aux;
    const int32_t* runs;    // [n_runs][4]: first virtual index, count, physical index of the prototype, aux offset of the geo table
    int32_t run0[4];        // the first run, read once per workgroup: node_ref is called a dozen times per pass and an LDS round trip each adds up
    int32_t n_nodes;        // VIRTUAL node count: the caller's depth-first list (indices in hits, ties, skip lists, grid items)
    int32_t n_runs;
    int32_t n_mats;
    int32_t cache_mat;  // first Sellmeier material, -1 when every material is a constant
    int32_t root;  // aux offset of the top-level grid, -1 when the scene has none
    int32_t root_pack;  // aux offset of its cells as (first item | count << 11), one word per cell; -1 when not built
    T unit;
};

template <class T> struct RayState {
    T ox, oy, oz, dx, dy, dz;
    T wl, qr, qi, I, n, pl;
    T len;  // +inf == None
    int32_t has_q;
    int32_t last;  // node the ray starts on (it was emitted there), -1 for a caller's ray
};

template <class T> struct Hit {
    T t;           // distance (local frame == lab frame: M is a rotation)
    T px, py, pz;  // local hit point
    int32_t node;  // -1: none
};

// A node as the kernels see it: the record (shared by all members of an instanced run) and the pose that is the member's
// own — origin[3] and lab AABB[6], contiguous — plus its leaf id.  Instanced runs: the children of a lattice group (MMA caps,
// MLA lenslets, DMD mirrors: component_group.py:228-304, 367) differ only in origin, box and leaf id, so the device image
// keeps ONE record per run and a 9-real table entry per member instead of 53 / 94 words each (cfg 5: 74 KB -> 16 KB, which is
// what lets the records of the live rays sit in LDS next to the image).  The virtual index space is the caller's: hits,
// tie-breaks, `last` and grid items never see the compression.
template <class T> struct NodeRef {
    const DNode<T>* nd;
    const T* geo;      // org[0..2], aabb[3..8]
    int32_t inst;      // member number inside its run, 0 for ordinary nodes
};
// Record i of a table, for a PER-LANE index: the byte offset by a 24-bit multiply (v_mul_u32_u24, full rate).  Plain pointer
// arithmetic on a 212-byte record compiles to v_mul_lo_u32, which issues at a quarter of that; indices are far below 2^24.
// Only where the index differs from lane to lane (the pair-queue kernels, LANES = true): in the linear pass every lane
// holds the same node index, the compiler multiplies in the scalar unit, and the intrinsic would drag that into vector
// registers (cfg 5: 14.4 against 14.1 ms).
template <bool LANES, class R> __device__ __forceinline__ const R* record_at(const R* table, int i) {
    if constexpr (LANES) return reinterpret_cast<const R*>(reinterpret_cast<const char*>(table) + __umul24((unsigned)i, (unsigned)sizeof(R)));
    else return table + i;
}
template <class T, uint32_t F> __device__ __forceinline__ NodeRef<T> node_ref(const Scene<T>& sc, int v) {
    NodeRef<T> r;
    r.inst = 0;
    if constexpr ((F & F_GRID) != 0) {  // runs only exist below gridded groups
        int shift = 0;
        if (sc.n_runs > 0) {  // scene-uniform
            const int first = sc.run0[0], cnt = sc.run0[1];
            if (v >= first) {
                if (v < first + cnt) {
                    r.nd = sc.nodes + sc.run0[2];
                    r.inst = v - first;
                    r.geo = sc.aux + sc.run0[3] + 9 * r.inst;
                    return r;
                }
                shift = cnt - 1;
                for (int k = 1; k < sc.n_runs; ++k) {  // further runs: rare
                    const int f2 = sc.runs[4 * k], c2 = sc.runs[4 * k + 1];
                    if (v < f2) break;
                    if (v < f2 + c2) {
                        r.nd = sc.nodes + sc.runs[4 * k + 2];
                        r.inst = v - f2;
                        r.geo = sc.aux + sc.runs[4 * k + 3] + 9 * r.inst;
                        return r;
                    }
                    shift += c2 - 1;
                }
            }
        }
        r.nd = sc.nodes + (v - shift);
    } else {
        r.nd = record_at<(F & F_FLAT) != 0>(sc.nodes, v);
    }
    r.geo = r.nd->org;  // org[3] and aabb[6] are adjacent in the record
    return r;
}

// what ot_segments.surface reports for a hit on virtual node v (the members of a run are consecutive leaves)
template <class T, uint32_t F> __device__ __forceinline__ int32_t leaf_id_of(const Scene<T>& sc, int v) {
    const NodeRef<T> nr = node_ref<T, F>(sc, v);
    return nr.nd->leaf_id + nr.inst;
}

template <class T> __device__ __forceinline__ T sqrt_t(T x);
template <> __device__ __forceinline__ double sqrt_t<double>(double x) { return sqrt(x); }
// fp32: the bare v_sqrt_f32 (1 ulp).  sqrtf() wraps it in a scaling sequence for denormal arguments (a compare, a multiply,
// two selects and an ldexp per call: ~150 of the 6000 instructions of the curved-surface kernel); every argument here is a
// squared length or 1 +- something of order one.
template <> __device__ __forceinline__ float sqrt_t<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <class T> __device__ __forceinline__ T abs_t(T x) { return x < T(0) ? -x : x; }
template <class T> __device__ __forceinline__ T min_t(T a, T b) { return a < b ? a : b; }
template <class T> __device__ __forceinline__ T max_t(T a, T b) { return a < b ? b : a; }

// 1/sqrt(x): hardware estimate + Newton steps (x is a squared length, never denormal here)
__device__ __forceinline__ double rsqrt_t(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    y = y * fma(-hx * y, y, 1.5);
    return y;
}
__device__ __forceinline__ float rsqrt_t(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    return y * fmaf(-0.5f * x * y, y, 1.5f);
}

// a/b and 1/b without the IEEE special-case scaffolding (v_div_scale / v_div_fmas / v_div_fixup):
// hardware reciprocal estimate + two Newton steps + one residual correction, <= 1 ulp for normal
// operands.  Callers never pass b == 0 (guarded) or denormals (lengths, direction components).
__device__ __forceinline__ double rcp_t(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double div_t(double a, double b) {
    const double r = rcp_t(b);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
// fp32: the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the ~10-instruction IEEE division sequence.  The
// single-precision kernels are VALU-bound where the double-precision ones are HBM-bound (cfg 2: 80 % of the fp32
// stream ceiling), and their contract is a tolerance against fp64, not correctly rounded quotients.
__device__ __forceinline__ float rcp_t(float b) { return __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float div_t(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
// a / b as the build's precision wants it where the source of the reference has a plain division: IEEE in double (the
// oracle's bits), the hardware reciprocal in single — the compiler's fp32 "/" guards against denormal and huge operands
// with a frexp / ldexp sequence of ~8 instructions, and the kernels divide lengths and indices of order one.
__device__ __forceinline__ double qd(double a, double b) { return a / b; }
__device__ __forceinline__ float qd(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
// a0*b0 + a1*b1 + a2*b2 with the roundings WRITTEN DOWN (one product rounded, two fused steps).  Left to the compiler's
// contraction the same source expression fuses differently in different surroundings, and the planar leaf test exists in
// two shapes (test_leaf with early exits, the branch-free slot of flat_grid_hit) whose results must agree bit for bit.
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <class T> __device__ __forceinline__ T dot3_t(T a0, T b0, T a1, T b1, T a2, T b2) { return fma_t(a2, b2, fma_t(a1, b1, a0 * b0)); }

// ---------------------------------------------------------------------------------------------
// solver.py:5-48 with the per-axis reciprocal hoisted out (RayInv is built once per segment).
template <class T> struct RayInv {
    T inv[3];
    bool par[3];
};
template <class T> __device__ __forceinline__ RayInv<T> make_inv(T dx, T dy, T dz) {
    RayInv<T> r;
    const T d[3] = {dx, dy, dz};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        r.par[ax] = abs_t(d[ax]) <= T(1e-8);  // np.isclose(d, 0)
        r.inv[ax] = rcp_t(d[ax]);  // unused when par[ax] (d ~ 0)
    }
    return r;
}
template <class T>
__device__ __forceinline__ bool slab_inv(T ox, T oy, T oz, const RayInv<T>& ri, const T* box, T& t1, T& t2) {
    t1 = T(0);
    t2 = Num<T>::inf();
    const T o[3] = {ox, oy, oz};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const T lo = box[2 * ax], hi = box[2 * ax + 1];
        if (ri.par[ax]) {
            if (o[ax] < lo || o[ax] > hi) { t1 = T(1); t2 = T(0); }
        } else {
            const T a = (lo - o[ax]) * ri.inv[ax], b = (hi - o[ax]) * ri.inv[ax];
            t1 = max_t(t1, min_t(a, b));
            t2 = min_t(t2, max_t(a, b));
        }
    }
    return (t2 + T(1e-12) >= t1) && (t2 >= T(0));
}

// A box whose slab interval starts at t1 cannot hold a hit nearer than `best`: whatever lies inside it is at t >=
// t1 up to rounding (the margin is 16x the self-hit guard, relative: 1.6e-8 / 1.6e-4).  The reference has no such
// test — it evaluates everything and keeps the minimum — so skipping what cannot become the minimum changes
// nothing, EXCEPT for count-limited leaves, whose counters see every geometric hit (optical_component.py:359-362):
// callers apply it only where no limited leaf is behind the box — and, for a LAB box, only when the box is trusted
// (OT_NODE_BOX_TRUSTED): a component's cached box may have gone stale (the reference never updates it; it then still
// gates hits, but the geometry can lie in front of it).  The local box of a curved leaf is derived from the surface at
// upload and needs no such flag.
template <class T> __device__ __forceinline__ bool beyond_best(T t1, T best_t) {
    return t1 > best_t + T(16) * Num<T>::eps_t() * (T(1) + best_t);  // false while best_t is +inf
}

// ---------------------------------------------------------------------------------------------
// material.py:54-72, 106-120
// A user function as a verified Chebyshev series (optable_amd/cheb.py): record [N, lo, hi, c[N], ...further blocks of N],
// evaluated by Clenshaw's recurrence in t = (2x - lo - hi) / (hi - lo).  block 0: the function, 1 / 2: its derivatives.
template <class T> __device__ __forceinline__ T cheb_eval(const T* rec, T x, int block = 0) {
    const int n = (int)rec[0];
    const T lo = rec[1], hi = rec[2];
    const T t2 = T(2) * qd(T(2) * x - (lo + hi), hi - lo);
    const T* c = rec + 3 + block * n;
    T b1 = T(0), b2 = T(0);
    for (int k = n - 1; k >= 1; --k) {
        const T b0 = c[k] + t2 * b1 - b2;
        b2 = b1;
        b1 = b0;
    }
    return c[0] + T(0.5) * t2 * b1 - b2;
}

template <class T, uint32_t F> __device__ __forceinline__ T material_index(const Scene<T>& sc, const DMat<T>& m, T wavelength_m) {
    if (m.kind == OT_MAT_CONST) return m.n;
    if constexpr ((F & F_MISC) != 0) {
        if (m.kind == OT_MAT_CHEB) {  // Material(n = callable), material.py:4-21: the series, on its interval
            const T* rec = sc.aux + (int)m.n;
            return cheb_eval(rec, min_t(max_t(wavelength_m, rec[1]), rec[2]));
        }
    }
    // 1 + sum_i B_i L^2/(L^2 - C_i) over a common denominator: one division instead of three
    const T um = qd(wavelength_m, T(1e-6)), um2 = um * um;
    const T p0 = um2 - m.C[0], p1 = um2 - m.C[1], p2 = um2 - m.C[2];
    const T num = m.B[0] * p1 * p2 + m.B[1] * p0 * p2 + m.B[2] * p0 * p1;
    return sqrt_t(T(1) + qd(um2 * num, p0 * p1 * p2));
}

// n(lambda) depends only on the ray's wavelength and the material, and a ray keeps its wavelength
// (ray.py:441-444), so the kernels evaluate the scene's first dispersive material once per ray
// instead of at every refraction (the reference re-evaluates per hit, optical_component.py:627-628).
// One cached value: more would cost the registers that keep 4 waves per SIMD resident.
template <class T> struct MatCache {
    T v;
};
template <class T, uint32_t F> __device__ __forceinline__ MatCache<T> make_matcache(const Scene<T>& sc, T wl) {
    MatCache<T> m = {T(1)};
    if (sc.cache_mat >= 0) m.v = material_index<T, F>(sc, sc.mats[sc.cache_mat], wl * sc.unit);
    return m;
}
template <class T, uint32_t F> __device__ __forceinline__ T cached_index(const Scene<T>& sc, const MatCache<T>& m, int idx, T wl) {
    if (idx == sc.cache_mat) return m.v;
    return material_index<T, F>(sc, *record_at<(F & F_FLAT) != 0>(sc.mats, idx), wl * sc.unit);
}

// ---------------------------------------------------------------------------------------------
// Asphere sag F(r) and the reference's finite differences (surfaces.py:351-369).
// F as a function of r^2 (both sag formulas only contain even powers of r): the root scan needs no
// square root to get r.
template <class T> __device__ __forceinline__ T sag_r2(const DNode<T>& nd, T r2) {
    if (nd.shape == OT_SHAPE_ASPHERE_PARAM) {  // component_group.py:1092-1097
        const T R = nd.p[1], k = nd.p[2];
        const T r4 = r2 * r2;
        return r2 / (R * (T(1) + sqrt_t(T(1) - (T(1) + k) * r2 / (R * R)))) + nd.p[3] * r4 + nd.p[4] * r4 * r2 +
               nd.p[5] * r4 * r4;
    }
    const T EFL = nd.p[1], n = nd.p[2];  // component_group.py:1061-1064
    return (EFL / (n + T(1))) * (T(-1) + sqrt_t(T(1) + (n + T(1)) / (n - T(1)) * r2 / (EFL * EFL)));
}
template <class T> __device__ __forceinline__ T sag(const DNode<T>& nd, T r) { return sag_r2(nd, r * r); }
// The same F for the root search only (sign scan + Newton polish), with the constant quotients folded on the
// host (fill_blob): p[6] = (1+k)/R^2, p[7] = 1/R for the parametric form; p[6] = (n+1)/((n-1) EFL^2),
// p[7] = EFL/(n+1) for the exact one.  One division instead of three, none for the exact form.  The normal
// and curvature (sag_d1 / sag_d2 below) keep the reference's expression term by term: their finite
// differences amplify rounding by 1/h^2.
// A paraboloid base (kappa = -1: p[6] = (1 + kappa) / R^2 = 0) has sqrt(1 - 0) = 1 in all three forms below; the branch
// is wave-uniform wherever the node is, and skips the square root and the division without changing a bit of the
// result (1 + 1 = 2 and its reciprocal are exact).
template <class T> __device__ __forceinline__ T sag_search(const DNode<T>& nd, T r2) {
    if (nd.shape == OT_SHAPE_ASPHERE_PARAM) {
        const T r4 = r2 * r2;
        const T base = nd.p[6] == T(0) ? (r2 * nd.p[7]) * T(0.5) : div_t(r2 * nd.p[7], T(1) + sqrt_t(T(1) - nd.p[6] * r2));
        return base + nd.p[3] * r4 + nd.p[4] * r4 * r2 + nd.p[5] * r4 * r4;
    }
    return nd.p[7] * (sqrt_t(T(1) + nd.p[6] * r2) - T(1));
}
// A value with the SIGN of x + F(r) and no division: the 10-point scan only compares signs, and
// 1 + sqrt(..) > 0.  (A negative radicand gives NaN here as it does in F: no crossing is reported.)
template <class T> __device__ __forceinline__ T asphere_sign(const DNode<T>& nd, T x, T r2) {
    if (nd.shape == OT_SHAPE_ASPHERE_PARAM) {
        const T r4 = r2 * r2;
        const T poly = nd.p[3] * r4 + nd.p[4] * r4 * r2 + nd.p[5] * r4 * r4;
        if (nd.p[6] == T(0)) return (x + poly) * T(2) + r2 * nd.p[7];
        return (x + poly) * (T(1) + sqrt_t(T(1) - nd.p[6] * r2)) + r2 * nd.p[7];
    }
    return x + nd.p[7] * (sqrt_t(T(1) + nd.p[6] * r2) - T(1));
}
// dF/dr divided by r, analytic, as a function of r^2 (Newton slope of the root polish only: it
// steers the iteration, the root it converges to does not depend on it)
template <class T> __device__ __forceinline__ T sag_slope_over_r(const DNode<T>& nd, T r2) {
    if (nd.shape == OT_SHAPE_ASPHERE_PARAM) {
        const T dpoly = r2 * (T(4) * nd.p[3] + r2 * (T(6) * nd.p[4] + r2 * T(8) * nd.p[5]));
        if (nd.p[6] == T(0)) return nd.p[7] + dpoly;
        return div_t(nd.p[7], sqrt_t(T(1) - nd.p[6] * r2)) + dpoly;
    }
    return div_t(nd.p[6] * nd.p[7], sqrt_t(T(1) + nd.p[6] * r2));
}
// fp64: the reference's central differences, h = 1e-4*radius, reproduced term by term.
// fp32: the differences would lose 4-5 digits to cancellation (F ~ 0.3, h ~ 2.5e-4), far more than
// their own O(h^2) ~ 1e-8 truncation error, so the float kernel uses the analytic derivatives —
// closer to what the reference computes in double than a float finite difference could be.
template <class T> __device__ __forceinline__ T sag_d1(const DNode<T>& nd, T r) {
    if (sizeof(T) == 4) {
        const T r2 = r * r;
        if (nd.shape == OT_SHAPE_ASPHERE_PARAM) {
            const T R = nd.p[1], sq = sqrt_t(T(1) - qd((T(1) + nd.p[2]) * r2, R * R));
            return qd(r, R * sq) + r * r2 * (T(4) * nd.p[3] + r2 * (T(6) * nd.p[4] + r2 * T(8) * nd.p[5]));
        }
        const T E = nd.p[1], n = nd.p[2], A = qd(n + T(1), n - T(1));
        return qd(A * r, (n + T(1)) * E * sqrt_t(T(1) + qd(A * r2, E * E)));
    }
    const T h = T(1e-4) * nd.p[0];
    return (sag(nd, r + h) - sag(nd, r - h)) / (T(2) * h);
}
template <class T> __device__ __forceinline__ T sag_d2(const DNode<T>& nd, T r) {
    if (sizeof(T) == 4) {
        const T r2 = r * r;
        if (nd.shape == OT_SHAPE_ASPHERE_PARAM) {
            const T R = nd.p[1], k1 = T(1) + nd.p[2], sq = sqrt_t(T(1) - qd(k1 * r2, R * R));
            return qd(T(1), R * sq) + qd(k1 * r2, R * R * R * sq * sq * sq) +
                   r2 * (T(12) * nd.p[3] + r2 * (T(30) * nd.p[4] + r2 * T(56) * nd.p[5]));
        }
        const T E = nd.p[1], n = nd.p[2], A = qd(n + T(1), n - T(1)), u = sqrt_t(T(1) + qd(A * r2, E * E));
        return qd(A, (n + T(1)) * E) * (qd(T(1), u) - qd(A * r2, E * E * u * u * u));
    }
    const T h = T(1e-4) * nd.p[0];
    return (sag(nd, r + h) - T(2) * sag(nd, r) + sag(nd, r - h)) / (h * h);
}

// Slope and curvature of a sag given as a series (OT_SHAPE_ASPHERE_CHEB).  fp64: the reference's central differences of
// F with h = 1e-4 * radius, term by term (surfaces.py:355-369) — F being the series; fp32: the series of F' and F'', for
// the reason given at sag_d1.
template <class T> __device__ __forceinline__ T cheb_d1(const T* rec, T radius, T r) {
    if (sizeof(T) == 4) return cheb_eval(rec, r, 1);
    const T h = T(1e-4) * radius;
    return (cheb_eval(rec, r + h) - cheb_eval(rec, r - h)) / (T(2) * h);
}
template <class T> __device__ __forceinline__ T cheb_d2(const T* rec, T radius, T r) {
    if (sizeof(T) == 4) return cheb_eval(rec, r, 2);
    const T h = T(1e-4) * radius;
    return (cheb_eval(rec, r + h) - T(2) * cheb_eval(rec, r) + cheb_eval(rec, r - h)) / (h * h);
}

// polygon aux record: [nverts, n(3), v0(3), u(3), v(3), (x,y)*nverts]   (surfaces.py:534-558)
template <class T> __device__ __forceinline__ bool poly_inside(const T* rec, T Px, T Py, T Pz) {
    const T tol = T(1e-9);
    const T rx = Px - rec[4], ry = Py - rec[5], rz = Pz - rec[6];
    const T px = rx * rec[7] + ry * rec[8] + rz * rec[9];
    const T py = rx * rec[10] + ry * rec[11] + rz * rec[12];
    const int nv = (int)rec[0];
    const T* v = rec + 13;
    bool inside = false, on_edge = false;
    for (int i = 0; i < nv; ++i) {
        const int j = (i + 1 == nv) ? 0 : i + 1;
        const T x1 = v[2 * i], y1 = v[2 * i + 1], x2 = v[2 * j], y2 = v[2 * j + 1];
        const T cr = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1);
        if (abs_t(cr) <= tol && min_t(x1, x2) - tol <= px && px <= max_t(x1, x2) + tol && min_t(y1, y2) - tol <= py &&
            py <= max_t(y1, y2) + tol)
            on_edge = true;
        if ((y1 > py) != (y2 > py)) {
            const T xc = x1 + qd((py - y1) * (x2 - x1), y2 - y1);
            if (xc >= px) inside = !inside;
        }
    }
    return on_edge || inside;
}

template <class T> __device__ __forceinline__ bool prim_inside(int kind, const T* body, T Px, T Py, T Pz) {
    if (kind == OT_SHAPE_CIRCLE) return Px * Px + Py * Py + Pz * Pz <= body[0] * body[0];  // 3-norm, surfaces.py:144-145
    if (kind == OT_SHAPE_RECT) return abs_t(Py) <= body[0] && abs_t(Pz) <= body[1];
    return poly_inside(body, Px, Py, Pz);
}

// Plane.union / subtract as a postfix program (surfaces.py:100-136); stack kept in a bit mask.
template <class T> __device__ __forceinline__ bool csg_inside(const T* prog, T Px, T Py, T Pz) {
    uint32_t stack = 0;
    int sp = 0;
    const int ntok = (int)prog[0];
    const T* t = prog + 1;
    for (int k = 0; k < ntok; ++k) {
        const int kind = (int)t[0], len = (int)t[1];
        bool r;
        if (kind >= 100) {
            const bool b = (stack >> (sp - 1)) & 1u, a = (stack >> (sp - 2)) & 1u;
            sp -= 2;
            r = (kind == 100) ? (a || b) : (a && !b);
        } else {
            r = prim_inside(kind, t + 2, Px, Py, Pz);
        }
        stack = (stack & ~(1u << sp)) | (uint32_t(r) << sp);
        ++sp;
        t += 2 + len;
    }
    return stack & 1u;
}

template <class T, uint32_t F>
__device__ __forceinline__ bool planar_boundary(const Scene<T>& sc, const DNode<T>& nd, T Px, T Py, T Pz) {
    if (nd.shape == OT_SHAPE_CIRCLE) return dot3_t(Px, Px, Py, Py, Pz, Pz) <= nd.r2;
    if (nd.shape == OT_SHAPE_RECT) return abs_t(Py) <= nd.p[0] && abs_t(Pz) <= nd.p[1];
    if constexpr (F & F_POLY) {
        if (nd.shape == OT_SHAPE_POLYGON2D) return poly_inside(sc.aux + nd.aux, Px, Py, Pz);
        return csg_inside(sc.aux + nd.aux, Px, Py, Pz);
    }
    return false;
}

template <class T, uint32_t F>
__device__ __forceinline__ bool curved_boundary(const Scene<T>& sc, const DNode<T>& nd, T Px, T Py, T Pz) {
    if constexpr (F & F_MISC) {
        if (nd.shape == OT_SHAPE_POLYGON3D) return poly_inside(sc.aux + nd.aux, Px, Py, Pz);
        if (nd.shape == OT_SHAPE_CYLINDER) {
            const T th = atan2(Py, Px);
            return nd.p[2] <= th && th <= nd.p[3] && -nd.p[1] <= Pz && Pz <= nd.p[1];
        }
        if (nd.shape == OT_SHAPE_ASPHERE_CHEB) return sqrt_t(Py * Py + Pz * Pz) <= nd.p[0] + T(1e-12);  // surfaces.py:375-378
    }
    switch (nd.shape) {
        case OT_SHAPE_SPHERE:
            // surfaces.py:300-303 bounds the cap by x.  For a shallow cap (R >> aperture) that test is
            // ill-conditioned in single precision (dr = dx*R/r), so the fp32 kernel applies the same
            // bound to the radial distance instead: r^2 <= R^2 - (R-h)^2 (host-computed in fp64).
            if (sizeof(T) == 4 && nd.r2 > T(0)) return Px > T(0) && Py * Py + Pz * Pz <= nd.r2;
            return nd.p[0] - nd.p[1] - T(1e-12) <= Px && Px <= nd.p[0] + T(1e-12);
        case OT_SHAPE_ASPHERE_PARAM:
        case OT_SHAPE_ASPHERE_EXACT: return sqrt_t(Py * Py + Pz * Pz) <= nd.p[0] + T(1e-12);
        default: return false;
    }
}

// Implicit function g(t) = f(o + t d) of the non-planar shapes and its derivative.
template <class T, uint32_t F>
__device__ __forceinline__ T surf_g(const Scene<T>& sc, const DNode<T>& nd, T ox, T oy, T oz, T dx, T dy, T dz, T t, T* dg) {
    const T Px = ox + t * dx, Py = oy + t * dy, Pz = oz + t * dz;
    if constexpr (F & F_MISC) {
        if (nd.shape == OT_SHAPE_CYLINDER) {
            if (dg) *dg = T(2) * (Px * dx + Py * dy);
            return Px * Px + Py * Py - nd.rad2;
        }
        if (nd.shape == OT_SHAPE_POLYGON3D) {
            const T* rec = sc.aux + nd.aux;
            if (dg) *dg = rec[1] * dx + rec[2] * dy + rec[3] * dz;
            return rec[1] * (Px - rec[4]) + rec[2] * (Py - rec[5]) + rec[3] * (Pz - rec[6]);
        }
        if (nd.shape == OT_SHAPE_ASPHERE_CHEB) {  // x + F(r), F the series of the user's sag (surfaces.py:371-373)
            const T* rec = sc.aux + nd.aux;
            const T r = sqrt_t(Py * Py + Pz * Pz);
            if (dg) *dg = dx + (r > Num<T>::tiny() ? qd(cheb_eval(rec, r, 1) * (Py * dy + Pz * dz), r) : T(0));
            return Px + cheb_eval(rec, r);
        }
    }
    switch (nd.shape) {
        case OT_SHAPE_SPHERE: {  // |P| - R has the sign and the roots of |P|^2 - R^2: no square root
            if (dg) *dg = T(2) * (Px * dx + Py * dy + Pz * dz);
            return Px * Px + Py * Py + Pz * Pz - nd.rad2;
        }
        default: {  // aspheres: x + F(r), F even in r
            const T r2 = Py * Py + Pz * Pz;
            if (dg) *dg = dx + sag_slope_over_r(nd, r2) * (Py * dy + Pz * dz);
            return Px + sag_search(nd, r2);
        }
    }
}

// Root of g inside a bracket with a sign change: Newton steps kept inside the shrinking
// bracket, bisection when a step leaves it (the reference polishes the same bracket with
// scipy brentq to xtol 2e-12; both converge on the same root).
template <class T, uint32_t F>
__device__ __forceinline__ T polish_root(const Scene<T>& sc, const DNode<T>& nd, T ox, T oy, T oz, T dx, T dy, T dz, T a, T b,
                                         T ga, T gb) {
    T t = a - qd(ga * (b - a), gb - ga);  // false-position start
    if (!(t > a && t < b)) t = T(0.5) * (a + b);
    for (int it = 0; it < 48; ++it) {
        T dg;
        const T g = surf_g<T, F>(sc, nd, ox, oy, oz, dx, dy, dz, t, &dg);
        if (g == T(0)) return t;
        if ((g < T(0)) == (ga < T(0))) { a = t; ga = g; } else { b = t; gb = g; }
        T tn = t - qd(g, dg);
        // converged Newton step: accept BEFORE the bracket safeguard — a converged iterate sits on the
        // bracket end it has just moved (tn == a), and bisecting there would throw the root away and
        // spend the remaining iterations halving the interval
        if (abs_t(tn - t) <= Num<T>::root_tol() * abs_t(t) + Num<T>::tiny()) return (tn >= a && tn <= b) ? tn : t;
        if (!(tn > a && tn < b)) tn = T(0.5) * (a + b);
        t = tn;
        if (b - a <= Num<T>::root_tol() * abs_t(t)) break;
    }
    return t;
}

// intersect_point_local, non-planar branch (optical_component.py:197-233), ray already in the
// leaf's frame: local-box slab -> [t1, min(t2, 100)] -> 10-point sign scan -> first root that is
// far enough, inside the length and inside the shape's boundary.  Planar leaves: test_leaf.
template <class T, uint32_t F>
__device__ __forceinline__ bool hit_leaf(const Scene<T>& sc, const DNode<T>& nd, T ox, T oy, T oz, T dx, T dy, T dz, T len, T& t_out,
                                         T& Px, T& Py, T& Pz, T prune_t = Num<T>::inf(), bool own = false) {
    if constexpr (F & F_CURVED) {
        const T EPS = Num<T>::eps_t();
        if (nd.shape == OT_SHAPE_POINT) return false;  // f = |P| never changes sign (surfaces.py:73-80)
        T t1, t2;
        const RayInv<T> li = make_inv(dx, dy, dz);
        slab_inv(ox, oy, oz, li, nd.lbox, t1, t2);
        if (t2 + EPS < t1) return false;
        if (beyond_best(t1, prune_t)) return false;  // every root lies in [t1 - EPS, t2 + EPS]: none can beat the best hit
        t1 = max_t(t1, T(0));
        t2 = min_t(t2, T(100));
        // np.linspace(t1 - EPS, t2 + EPS, 10): sign change per sub-interval, roots ascending
        const T a = t1 - EPS, b = t2 + EPS, step = qd(b - a, T(9));
        if (nd.shape == OT_SHAPE_SPHERE) {
            // Closed form of the same search.  g(t) = |o + t d| - R is negative exactly between the two
            // intersections r0 < r1 of the line with the sphere, so sample interval i of the scan shows a sign
            // change iff exactly one of them lies inside it, and brentq would converge to that one.  The
            // candidates are therefore r0 and r1 (ascending) when they lie inside (a, b) and not both in the
            // same interval; filters and boundary as below.  Roots through the closest approach to the
            // centre (no cancellation between |o|^2 and R^2).  ~80 instead of ~380 instructions per test,
            // and every ray leaving a micro-mirror pays this test for the cap it just left (root at t = 0).
            const T tca = -(ox * dx + oy * dy + oz * dz);
            const T cx = ox + tca * dx, cy = oy + tca * dy, cz = oz + tca * dz;
            const T disc = nd.rad2 - (cx * cx + cy * cy + cz * cz);
            if (!(disc > T(0))) return false;  // the line stays outside: g never changes sign
            const T hc = sqrt_t(disc);
            const T r0 = tca - hc, r1 = tca + hc;
            const T inv_step = qd(T(9), b - a);
            const bool in0 = r0 > a && r0 < b, in1 = r1 > a && r1 < b;
            const int i0 = min((int)((r0 - a) * inv_step), 8), i1 = min((int)((r1 - a) * inv_step), 8);
            if (in0 && in1 && i0 == i1) return false;  // both inside one interval: equal signs at its ends
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const T t = k == 0 ? r0 : r1;
                if (!(k == 0 ? in0 : in1)) continue;
                if (t >= T(0) && abs_t(t) >= EPS && t <= len) {
                    const T X = ox + t * dx, Y = oy + t * dy, Z = oz + t * dz;
                    if (curved_boundary<T, F>(sc, nd, X, Y, Z)) {
                        t_out = t; Px = X; Py = Y; Pz = Z;
                        return true;
                    }
                }
            }
            return false;
        }
        // sign samples: aspheres through the division-free form (same sign, and the polish below only uses
        // the bracket values for its starting guess)
        const bool asph = nd.shape == OT_SHAPE_ASPHERE_PARAM || nd.shape == OT_SHAPE_ASPHERE_EXACT;
        auto sample = [&](T t) {
            if (asph) {
                const T Y = oy + t * dy, Z = oz + t * dz;
                return asphere_sign(nd, ox + t * dx, Y * Y + Z * Z);
            }
            return surf_g<T, F>(sc, nd, ox, oy, oz, dx, dy, dz, t, (T*)nullptr);
        };
        // A ray that STARTS on a convex asphere needs no scan in two cases (both exact in real arithmetic, and what the scan
        // below would find): g is convex along the ray with g(0) = 0, so (i) leaving to the positive side, g'(0) > 0, it
        // never returns; (ii) going to the negative side it returns at most once, and not before the end of the search
        // interval if g is still negative there.  The only crossing the reference's samples can then show is the one at the
        // start point, which its |t| < EPS filter throws away (optical_component.py:221-227).  Grazing starts (|g'| small,
        // where rounding of the start point could move that root beyond EPS) take the scan.  cfg 5: the rays leaving the
        // lens front towards the mirror and those entering the glass — two of six segments per round trip.
        if (own && asph && (nd.flags & (DN_CONVEX_POS | DN_CONVEX_NEG))) {
            const T sgn = (nd.flags & DN_CONVEX_POS) ? T(1) : T(-1);
            const T r20 = oy * oy + oz * oz;
            const T dg0 = sgn * (dx + sag_slope_over_r(nd, r20) * (oy * dy + oz * dz));
            if (dg0 > T(0.25)) return false;
            if (dg0 < T(-0.25) && sgn * sample(b) < T(0)) return false;
        }
        // Two steps, so that the instruction stream holds the ten samples once and ONE copy of the root polish: (1) the
        // ten samples, unrolled, leave a mask of the intervals with a sign change; (2) the marked intervals in ascending
        // order — almost always one — are polished and filtered as the reference does.  (Written as one loop, the compiler
        // unrolled nine polishes into every inlined copy of this function: 14000 instructions in the curved-surface kernel.)
        uint32_t crossings = 0;
        {
            T gl = sample(a);
#pragma unroll
            for (int i = 1; i < 10; ++i) {
                const T gr = sample((i == 9) ? b : a + T(i) * step);
                // fp64: the reference's strict product test (optical_component.py:131).  fp32: |P| - R is
                // quantised to ~2e-6 at R ~ 30, so a sample lands on g == 0 exactly for several percent of
                // the rays and the product test would drop those roots; compare signs instead.
                const bool crossing = sizeof(T) == 4 ? ((gl < T(0)) != (gr < T(0))) : (gl * gr < T(0));
                if (crossing) crossings |= 1u << i;
                gl = gr;
            }
        }
#pragma unroll 1
        while (crossings) {
            const int i = __builtin_ctz(crossings);
            crossings &= crossings - 1;
            const T tl = (i == 1) ? a : a + T(i - 1) * step, tr = (i == 9) ? b : a + T(i) * step;
            // A ray that was emitted ON this surface (own) has a root at its start point: the bracket that holds t = 0
            // shows a sign change, brentq converges to that root and the |t| < EPS filter throws it away
            // (optical_component.py:221-227).  The samples are taken as always — a second crossing further along still
            // shows in its own interval — but polishing the root at the start point only to discard it is skipped.
            // ... provided the crossing in that bracket IS the start point's: with the start point rounded to the far side of
            // the surface (fp32: up to 1e-5 off) its root falls before the first sample, and the crossing seen here is a
            // genuine second hit (a short chord); then g at the bracket's start already has the sign the ray is heading
            // for, i.e. g(tl) and dg/dt at the start point have the same sign.  Grazing starts are polished like any
            // other bracket and left to the |t| < EPS filter, as the reference does.
            if (own && tl <= T(0) && tr >= T(0)) {
                T dg0;
                (void)surf_g<T, F>(sc, nd, ox, oy, oz, dx, dy, dz, T(0), &dg0);
                if (abs_t(dg0) > T(0.05) && sample(tl) * dg0 < T(0)) continue;
            }
            const T t = polish_root<T, F>(sc, nd, ox, oy, oz, dx, dy, dz, tl, tr, sample(tl), sample(tr));
            if (t >= T(0) && abs_t(t) >= EPS && t <= len) {
                const T X = ox + t * dx, Y = oy + t * dy, Z = oz + t * dz;
                if (curved_boundary<T, F>(sc, nd, X, Y, Z)) {
                    t_out = t; Px = X; Py = Y; Pz = Z;
                    return true;
                }
            }
        }
    }
    return false;
}

template <class T> __device__ __forceinline__ void to_local(const DNode<T>& nd, T vx, T vy, T vz, T& lx, T& ly, T& lz) {
    lx = nd.M[0] * vx + nd.M[3] * vy + nd.M[6] * vz;  // M^T v  (inverse of a rotation)
    ly = nd.M[1] * vx + nd.M[4] * vy + nd.M[7] * vz;
    lz = nd.M[2] * vx + nd.M[5] * vy + nd.M[8] * vz;
}
template <class T> __device__ __forceinline__ void to_lab(const DNode<T>& nd, T lx, T ly, T lz, T& vx, T& vy, T& vz) {
    vx = nd.M[0] * lx + nd.M[1] * ly + nd.M[2] * lz;
    vy = nd.M[3] * lx + nd.M[4] * ly + nd.M[5] * lz;
    vz = nd.M[6] * lx + nd.M[7] * ly + nd.M[8] * lz;
}

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// Interact-count gate (optical_component.py:140-149, 359-362): a limited leaf that is
// geometrically hit consumes one count while count < max, otherwise it is transparent.
//   GATE_PLAIN  fused / blocked kernels: atomic increment-below-cap (exact for one ray per class per launch, the
//               host API's rounds; memory-safe and cap-exact for any sharing).
//   GATE_PROBE  pre-pass of a branching generation: only record which limited leaves the ray
//               hits geometrically (probe[slot*stride + idx] = 1); nothing else is evaluated.
//   GATE_TABLE  branching generation proper: pass iff counts + (number of EARLIER rays of this
//               tree in this generation that hit the leaf) < max, i.e. the reference's FIFO order;
//               the table is updated once per tree afterwards (k_gen_counts).
enum { GATE_PLAIN = 0, GATE_PROBE = 1, GATE_TABLE = 2 };
struct GateCtx {
    int32_t* counts;      // [n_slots][n_classes]
    int32_t n_classes, cls;
    const int32_t* rank;  // GATE_TABLE: [n_slots][stride]
    int32_t* probe;       // GATE_PROBE: [n_slots][stride]
    int64_t stride, idx;
};
template <int GATE> __device__ __forceinline__ bool count_gate(const GateCtx& g, int32_t slot, int32_t max_count) {
    if ((uint32_t)g.cls >= (uint32_t)g.n_classes) return true;  // id outside the table: not counted (never indexes out of range)
    int32_t* c = g.counts + (int64_t)slot * g.n_classes + g.cls;
    if (GATE == GATE_TABLE) return *c + g.rank[(int64_t)slot * g.stride + g.idx] < max_count;
    // "increment while below the cap" as one atomic step: rays of one class that meet in a launch (a direct C-ABI
    // caller that did not split them into rounds) can never push a counter past its cap or lose an update; WHICH
    // of them gets the remaining counts is then unspecified (the reference's order is the caller's rounds).
    int32_t v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (v < max_count) {
        if (__hip_atomic_compare_exchange_strong(c, &v, v + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
    }
    return false;
}

// One leaf against one ray; updates `best`.  ORDERED = the caller visits leaves in increasing
// node index (strict '<' keeps the first minimum); otherwise an exact tie goes to the lower index,
// which is the same rule.
// DEFER_AABB: the caller has NOT yet applied this leaf's own AABB test (component_group.py:104-107).
// A planar leaf is cheaper to reject by its sign/distance tests than by the slab test, and a hit
// needs both, so the slab test runs last and only for leaves that would otherwise be hits.
// KIND: 0 any leaf; 1 / 2: the caller has sorted planar from curved leaves itself and this call site only ever sees planar /
// curved ones (the other half of the function is not compiled into it).
template <class T, uint32_t F, int GATE, bool ORDERED, bool DEFER_AABB = false, int KIND = 0>
__device__ __forceinline__ void test_leaf(const Scene<T>& sc, const NodeRef<T>& nr, int idx, const RayState<T>& r, Hit<T>& best,
                                          const GateCtx& gate, const RayInv<T>* ri = nullptr) {
    const DNode<T>& nd = *nr.nd;
    if (GATE == GATE_PROBE && nd.max_count < 0) return;  // the probe pass only looks at limited leaves
    const T rx = r.ox - nr.geo[0], ry = r.oy - nr.geo[1], rz = r.oz - nr.geo[2];
    const int sh = nd.shape;
    const bool planar = sh == OT_SHAPE_CIRCLE || sh == OT_SHAPE_RECT || sh == OT_SHAPE_POLYGON2D || sh == OT_SHAPE_CSG;
    const bool limited = (F & F_LIMIT) && nd.max_count >= 0;
    T t, Px, Py, Pz;
    if (KIND != 2 && (!(F & F_CURVED) || KIND == 1 || planar)) {
        // A ray cannot meet the plane it starts on a second time: the only root is t = 0, which the reference's
        // |t| < 1e-9 guard rejects (optical_component.py:184).  Saying so by identity instead of by distance is
        // the same rule in exact arithmetic and immune to rounding of the start point — in single precision a
        // start point 1e-5 off its plane at coordinates ~30 produced a second "hit" just beyond the guard
        // (8 in 10^4 rays of cfg 3 left the fp64 path that way, tools/fp32_divergence.py).
        if (idx == r.last) return;
        // Planar leaf, evaluated lazily: only the local x row is needed to know t
        // (t = -o_x/d_x, optical_component.py:171-179).  Exact rejections first: parallel,
        // behind the ray (sign test == t < 0), t == 0.  Then a CONSERVATIVE far test (1e-9
        // slack) drops leaves that cannot beat the current best; it is skipped for
        // count-limited leaves, whose gate must see every geometric hit.
        const T lox = dot3_t(nd.M[0], rx, nd.M[3], ry, nd.M[6], rz);
        const T ldx = dot3_t(nd.M[0], r.dx, nd.M[3], r.dy, nd.M[6], r.dz);
        const T s = -lox;
        if (ldx == T(0) || s == T(0) || ((s > T(0)) != (ldx > T(0)))) return;
        if (!limited && abs_t(s) > best.t * abs_t(ldx) * (T(1) + (sizeof(T) == 4 ? T(1e-6) : T(1e-9)))) return;  // (1 + 1e-9 is 1 in single precision)
        t = div_t(s, ldx);
        if (abs_t(t) < Num<T>::eps_t() || t < T(0) || t > r.len) return;
        if (!limited && !(t < best.t || (!ORDERED && t == best.t && idx < best.node))) return;
        if (DEFER_AABB && (nd.flags & OT_NODE_CHECK_AABB)) {  // the leaf's own AABB test, after the cheap rejections
            T u1, u2;
            if (!slab_inv(r.ox, r.oy, r.oz, *ri, nr.geo + 3, u1, u2)) return;
        }
        const T loy = dot3_t(nd.M[1], rx, nd.M[4], ry, nd.M[7], rz), loz = dot3_t(nd.M[2], rx, nd.M[5], ry, nd.M[8], rz);
        const T ldy = dot3_t(nd.M[1], r.dx, nd.M[4], r.dy, nd.M[7], r.dz), ldz = dot3_t(nd.M[2], r.dx, nd.M[5], r.dy, nd.M[8], r.dz);
        Px = fma_t(t, ldx, lox); Py = fma_t(t, ldy, loy); Pz = fma_t(t, ldz, loz);
        if (!planar_boundary<T, F>(sc, nd, Px, Py, Pz)) return;
    } else if (KIND != 1) {
        const T prune_t = limited ? Num<T>::inf() : best.t;
        if (DEFER_AABB && (nd.flags & OT_NODE_CHECK_AABB)) {  // curved leaves: the slab test is the cheap one
            T u1, u2;
            if (!slab_inv(r.ox, r.oy, r.oz, *ri, nr.geo + 3, u1, u2)) return;
            if ((nd.flags & OT_NODE_BOX_TRUSTED) && beyond_best(u1, prune_t)) return;
        }
        T ox, oy, oz, dx, dy, dz;
        to_local(nd, rx, ry, rz, ox, oy, oz);
        to_local(nd, r.dx, r.dy, r.dz, dx, dy, dz);
        if (!hit_leaf<T, F>(sc, nd, ox, oy, oz, dx, dy, dz, r.len, t, Px, Py, Pz, prune_t, idx == r.last)) return;
    }
    if constexpr (F & F_LIMIT) {
        if (GATE == GATE_PROBE) {
            gate.probe[(int64_t)nd.slot * gate.stride + gate.idx] = 1;
            return;
        }
        if (limited && !count_gate<GATE>(gate, nd.slot, nd.max_count)) return;
    }
    if (t < best.t || (!ORDERED && t == best.t && idx < best.node)) {
        best.t = t; best.node = idx; best.px = Px; best.py = Py; best.pz = Pz;
    }
}

template <class T> __device__ __forceinline__ T pick(int axis, T x, T y, T z) { return axis == 0 ? x : (axis == 1 ? y : z); }

// Children of a gridded group (OT_NODE_GRID): instead of testing every child's AABB like
// ComponentGroup.interact does (component_group.py:104-115), look up the cells the ray can touch
// while it is inside the group's box and run the SAME per-child AABB + leaf tests on that
// superset only.  aux record: [a0 a1 g0 g1 org0 org1 inv0 inv1 margin | start[g0*g1+1] | items...].
template <class T, uint32_t F, int GATE>
__device__ __forceinline__ void grid_children(const Scene<T>& sc, const DNode<T>& grp, const RayState<T>& r, const RayInv<T>& ri,
                                              T t1, T t2, Hit<T>& best, const GateCtx& gate) {
    const T* g = sc.aux + grp.aux;
    const int a0 = (int)g[0], a1 = (int)g[1], g0 = (int)g[2], g1 = (int)g[3];
    const T margin = g[8];
    const T ta = max_t(t1, T(0)), tb = t2;
    const T o0 = pick(a0, r.ox, r.oy, r.oz), o1 = pick(a1, r.ox, r.oy, r.oz);
    const T d0 = pick(a0, r.dx, r.dy, r.dz), d1 = pick(a1, r.dx, r.dy, r.dz);
    const T p0a = o0 + ta * d0, p0b = o0 + tb * d0, p1a = o1 + ta * d1, p1b = o1 + tb * d1;
    const T lo0 = min_t(p0a, p0b) - margin, hi0 = max_t(p0a, p0b) + margin;
    const T lo1 = min_t(p1a, p1b) - margin, hi1 = max_t(p1a, p1b) + margin;
    auto cell = [](T v, T org, T inv, int n) {
        const T c = (v - org) * inv;
        return c <= T(0) ? 0 : (c >= T(n - 1) ? n - 1 : (int)c);
    };
    const int c0lo = cell(lo0, g[4], g[6], g0), c0hi = cell(hi0, g[4], g[6], g0);
    const int c1lo = cell(lo1, g[5], g[7], g1), c1hi = cell(hi1, g[5], g[7], g1);
    const T* start = g + 9;
    const T* items = start + (g0 * g1 + 1);
    // Phase 1 (cheap, lane-divergent): walk the cells, keep the children whose own AABB the ray hits.
    // Phase 2 (expensive, lock-step): every lane runs the leaf test of its c-th candidate at the same
    // time.  Testing a curved leaf the moment a lane finds it would serialise the 10-point scans of
    // the 64 lanes behind each other (measured on cfg 5: ~15k VALU instructions per wave-segment).
    // The walk is one flat loop with its position (cell, item) in registers, so that a lane whose four candidate slots
    // are full tests them and walks on — rare; a lattice lists one member and perhaps a back plate per cell — and the
    // instruction stream holds the leaf test once.
    int c1 = c1lo, c0 = c0lo;
    int k = (int)start[c1 * g0 + c0], ke = (int)start[c1 * g0 + c0 + 1];
    bool walking = true;
#pragma unroll 1
    while (__any(walking)) {  // rounds of up to four candidates per lane: almost always one
        int cand0 = -1, cand1 = -1, cand2 = -1, cand3 = -1, ncand = 0;
        while (walking && ncand < 4) {
            if (k >= ke) {  // next cell of the footprint
                if (++c0 > c0hi) {
                    c0 = c0lo;
                    if (++c1 > c1hi) { walking = false; break; }
                }
                k = (int)start[c1 * g0 + c0];
                ke = (int)start[c1 * g0 + c0 + 1];
                continue;
            }
            const int ci = (int)items[k++];
            if (ci == cand0 || ci == cand1 || ci == cand2 || ci == cand3) continue;  // listed in several cells
            const NodeRef<T> ch = node_ref<T, F>(sc, ci);
            T u1, u2;
            if (!slab_inv(r.ox, r.oy, r.oz, ri, ch.geo + 3, u1, u2)) continue;  // the child's own AABB test, unchanged
            if (beyond_best(u1, best.t)) continue;  // (gridded groups hold no count-limited child, and only trusted boxes: scene.py)
            if (ncand == 0) cand0 = ci;
            else if (ncand == 1) cand1 = ci;
            else if (ncand == 2) cand2 = ci;
            else cand3 = ci;
            ++ncand;
        }
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {  // one copy of the leaf test in the instruction stream
            const int ci = c == 0 ? cand0 : (c == 1 ? cand1 : (c == 2 ? cand2 : cand3));
            if (c < ncand) test_leaf<T, F, GATE, false>(sc, node_ref<T, F>(sc, ci), ci, r, best, gate);
        }
    }
}

// Per-lane walk of one top-level component (a leaf, or a group with everything below it): the
// skip-list logic of nearest_hit for a single subtree, with a lane-private index.
template <class T, uint32_t F, int GATE>
__device__ __forceinline__ void walk_subtree(const Scene<T>& sc, int first, const RayState<T>& r, const RayInv<T>& ri, Hit<T>& best,
                                             const GateCtx& gate) {
    const int last = node_ref<T, F>(sc, first).nd->end;
    for (int j = first; j < last;) {
        const NodeRef<T> nr = node_ref<T, F>(sc, j);
        const DNode<T>& nd = *nr.nd;
        T t1 = T(0), t2 = T(0);
        if (nd.flags & OT_NODE_CHECK_AABB) {
            if (!slab_inv(r.ox, r.oy, r.oz, ri, nr.geo + 3, t1, t2)) { j = nd.end; continue; }
            if (!(F & F_LIMIT) && (nd.flags & OT_NODE_BOX_TRUSTED) && beyond_best(t1, best.t)) { j = nd.end; continue; }
        }
        if (nd.kind == OT_NODE_GROUP) {
            if constexpr (F & F_GRID) {
                if (nd.flags & OT_NODE_GRID) {
                    grid_children<T, F, GATE>(sc, nd, r, ri, t1, t2, best, gate);
                    j = nd.end;
                    continue;
                }
            }
            ++j;
            continue;
        }
        test_leaf<T, F, GATE, false>(sc, nr, j, r, best, gate);
        ++j;
    }
}

// Nearest hit through the top-level grid: clip the ray to the scene box, visit the cells it
// crosses in order (2-D DDA) and stop as soon as the best hit lies inside the part of the ray
// already covered.  Candidates are top-level components binned by their lab AABB (with margin);
// each is examined with exactly the tests the linear pass applies, and ties go to the lower node
// index, so the winner is the one optical_table.py:119-123 picks.
// aux record: [a0 a1 g0 g1 org0 org1 inv0 inv1 margin size0 size1 | start[g0*g1+1] | items...]
template <class T, uint32_t F, int GATE>
__device__ __forceinline__ void root_grid_hit(const Scene<T>& sc, const RayState<T>& r, const RayInv<T>& ri, Hit<T>& best,
                                              const GateCtx& gate) {
    const T* g = sc.aux + sc.root;
    const int a0 = (int)g[0], a1 = (int)g[1], g0 = (int)g[2], g1 = (int)g[3];
    const T org0 = g[4], org1 = g[5], inv0 = g[6], inv1 = g[7], margin = g[8], size0 = g[9], size1 = g[10];
    const T o0 = pick(a0, r.ox, r.oy, r.oz), o1 = pick(a1, r.ox, r.oy, r.oz);
    const T d0 = pick(a0, r.dx, r.dy, r.dz), d1 = pick(a1, r.dx, r.dy, r.dz);
    const T i0 = pick(a0, ri.inv[0], ri.inv[1], ri.inv[2]), i1 = pick(a1, ri.inv[0], ri.inv[1], ri.inv[2]);
    const bool par0 = abs_t(d0) <= T(1e-12), par1 = abs_t(d1) <= T(1e-12);
    // clip to the grid rectangle
    T tin = T(0), tout = Num<T>::inf();
    const T hi0 = org0 + size0 * T(g0), hi1 = org1 + size1 * T(g1);
    if (par0) { if (o0 < org0 || o0 > hi0) return; }
    else { const T a = (org0 - o0) * i0, b = (hi0 - o0) * i0; tin = max_t(tin, min_t(a, b)); tout = min_t(tout, max_t(a, b)); }
    if (par1) { if (o1 < org1 || o1 > hi1) return; }
    else { const T a = (org1 - o1) * i1, b = (hi1 - o1) * i1; tin = max_t(tin, min_t(a, b)); tout = min_t(tout, max_t(a, b)); }
    if (!(tin <= tout)) return;
    // starting cell and stepping
    auto clampi = [](T c, int n) { return c <= T(0) ? 0 : (c >= T(n - 1) ? n - 1 : (int)c); };
    int c0 = clampi((o0 + tin * d0 - org0) * inv0, g0), c1 = clampi((o1 + tin * d1 - org1) * inv1, g1);
    const int s0 = d0 > T(0) ? 1 : -1, s1 = d1 > T(0) ? 1 : -1;
    T tmax0 = par0 ? Num<T>::inf() : (org0 + size0 * T(c0 + (s0 > 0 ? 1 : 0)) - o0) * i0;
    T tmax1 = par1 ? Num<T>::inf() : (org1 + size1 * T(c1 + (s1 > 0 ? 1 : 0)) - o1) * i1;
    const T dt0 = par0 ? Num<T>::inf() : size0 * abs_t(i0), dt1 = par1 ? Num<T>::inf() : size1 * abs_t(i1);
    const T* start = g + 11;
    const T* items = start + (g0 * g1 + 1);
    const T slack = T(4) * margin;
    for (int guard = 0; guard < g0 + g1 + 2; ++guard) {
        const int cidx = __mul24(c1, g0) + c0;
        const int kb = (int)start[cidx], ke = (int)start[cidx + 1];
        for (int k = kb; k < ke; ++k) {
            const int item = (int)items[k];
            const NodeRef<T> nr = node_ref<T, F>(sc, item);
            // the compiler lists leaves directly wherever it can (scene.py:_root_grid): cheap planar rejections
            // first, the leaf's own AABB test last; subtrees (stale boxes, gridded groups) take the general walk.
            // (A branch-free planar test was measured here, cfg 3 fp32: 5.54 against 5.09 ms with the early exits —
            // in a per-lane walk whole waves leave a candidate together often enough; in the slots of flat_grid_hit,
            // where 64 lanes hold 64 unrelated pairs, it is the other way round.)
            if constexpr (F & F_SUBTREE) {
                if (nr.nd->kind != OT_NODE_LEAF) { walk_subtree<T, F, GATE>(sc, item, r, ri, best, gate); continue; }
            }
            test_leaf<T, F, GATE, false, true>(sc, nr, item, r, best, gate, &ri);
        }
        const T texit = min_t(tmax0, tmax1);
        if (best.t + slack < texit) return;  // nothing in later cells can be nearer (or tie)
        if (tmax0 < tmax1) { c0 += s0; tmax0 += dt0; if (c0 < 0 || c0 >= g0) return; }
        else { c1 += s1; tmax1 += dt1; if (c1 < 0 || c1 >= g1) return; }
    }
}

// ---------------------------------------------------------------------------------------------
// flat_grid_hit: the nearest hit through the top-level grid for the 64 rays of a WAVE together.
// In root_grid_hit every lane walks its own cells and tests its own candidates, so the wave is as slow as its slowest
// lane at every step: on cfg 3 the longest of 64 walks is 6.2 cells (mean 1.95) and the wave runs 16.7 candidate
// tests per pass where a ray needs 4.0 — the cell loop is ~85 % of the kernel's VALU instructions at 21 % active
// lanes.  Here the walk and the tests are separated:
//   walk   every lane steps through up to FLAT_CELLS cells WITHOUT testing and notes the item ranges of those cells;
//          one wave-wide scan of the counts gives every lane its place in a queue of (lane, leaf) pairs in LDS (kept as
//          one marker per item range, see "queue" below);
//   test   the wave takes the pairs 64 at a time: lane q tests pair q (the ray's origin, direction, length and
//          start leaf come from the owner lane's registers by ds_bpermute), so every lane has a real
//          candidate; a hit goes into the ray's slot of a key table with ONE 64-bit LDS atomic minimum on
//          (t as ordered bits << 32 | node index): the smallest t wins, exact ties go to the lower node index — the
//          rule of optical_table.py:119-123 / component_group.py:118-120;
//   round  a lane is done when its best hit lies inside the part of the ray the walk has covered, or when the walk left
//          the grid; the others walk on (second and later rounds have few lanes, but also few pairs).
// A slot ends with its atomic and reads nothing back; the owner of a ray forms the winner's hit point after the last round
// from its own registers and the winner's t (rebuild_hit: the expressions of the test, the same bits).
// Planar leaves only (preset FR); results are bit-identical to root_grid_hit in both precisions
// (test_pair_queue_walk_equals_per_lane_walk, test_random_planar_scene_pair_queue_variants_agree).  In double precision
// t does not fit a 64-bit key next to the index: the key is t alone and the node index lives in a second table (see
// commit below).  cfg 3 fp64 gains 3 % from it (139 registers, 3 waves per SIMD), random planar scenes 12-29 %.
#ifndef OT_FLAT_CELLS
#define OT_FLAT_CELLS 2  // cells per round; cfg 3 fp32 with the final slot: 1 / 2 / 3 / 4 cells = 3.81 / 3.42-3.50 / 3.64 / 3.71 ms
#endif
static constexpr int FLAT_CELLS = OT_FLAT_CELLS;
template <class T> struct FlatLds {
    unsigned long long* key;   // [64]  fp32: t bits << 32 | node; fp64: t bits
    int32_t* node;             // [64]  fp64 only: lowest node index among the candidates at the key's t
    uint16_t* queue;           // [queue_cap]
    int32_t queue_cap;
    // bytes per wave, without the queue (kernels.h and the host size the LDS with this)
    static constexpr int fixed_bytes = sizeof(T) == 4 ? 64 * 8 : 64 * (8 + 4);
};
// exclusive add-scan over the 64 lanes in six DPP adds (row_shr 1 / 2 / 4 / 8 inside the rows of 16, then row_bcast 15 /
// 31 across the rows) — no LDS crossbar round trips; the wave total comes back in a scalar register.  Called with all 64
// lanes enabled.
__device__ __forceinline__ int wave_excl_scan_i32(int v, int& total) {
    int incl = v;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);  // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);  // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);  // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);  // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
    total = __builtin_amdgcn_readlane(incl, 63);
    return incl - v;
}
// inclusive max-scan over the 64 lanes, same DPP steps.  (Lanes without a source read INT_MIN, the identity of a signed
// maximum: with the identity as the fill value the compiler folds the cross-lane move into the v_max_i32 itself — one
// instruction per step instead of four.)
__device__ __forceinline__ int wave_incl_max_i32(int v) {
    constexpr int LOWEST = (int)0x80000000;
    v = max(v, __builtin_amdgcn_update_dpp(LOWEST, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(LOWEST, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(LOWEST, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(LOWEST, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(LOWEST, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(LOWEST, v, 0x143, 0xc, 0xf, false));
    return v;
}
__device__ __forceinline__ int wave_all_min_i32(int v) {  // DPP steps of wave_incl_max_i32, minimum; the result of lane 63 to all
    constexpr int HIGHEST = 0x7fffffff;
    v = min(v, __builtin_amdgcn_update_dpp(HIGHEST, v, 0x111, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(HIGHEST, v, 0x112, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(HIGHEST, v, 0x114, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(HIGHEST, v, 0x118, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(HIGHEST, v, 0x142, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(HIGHEST, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
#ifdef OT_STAMP  // diagnostic build: phases of a call into st_acc[5..] (walk, queue, test, verdict; [9] slots, [10] rounds)
#define OT_FLAT_STAMP_PARAMS , unsigned long long* st_acc, unsigned long long& st_last
#define OT_FLAT_AT(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long _t = __builtin_amdgcn_s_memtime(); st_acc[k] += _t - st_last; st_last = _t; } while (0)
#define OT_FLAT_COUNT(k) do { st_acc[k] += 1; } while (0)
#elif defined(OT_MARK)
#define OT_FLAT_STAMP_PARAMS
#define OT_FLAT_AT(k) asm volatile("; OT_MARK flat " #k)
#define OT_FLAT_COUNT(k) do {} while (0)
#else
#define OT_FLAT_STAMP_PARAMS
#define OT_FLAT_AT(k) do {} while (0)
#define OT_FLAT_COUNT(k) do {} while (0)
#endif
// The header of the top-level grid, decoded ONCE per wave before the pass loop (kernels.h): the table sits in LDS, which
// the pass loop writes to, so the compiler cannot keep these values across passes on its own — every call used to read and
// convert them again, ~35 instructions per pass for numbers that never change.  (Vector registers: the pair-queue kernels
// run at 3 waves per SIMD for their LDS, a third of the register file is unused.)
template <class T> struct FlatGrid {
    int a0, a1, g0, g1;
    T org0, org1, inv0, inv1, size0, size1, hi0, hi1, slack;
    const T* items;
    const T* cellpack;  // cells as (first item | count << 11)
};
// a wave-uniform value the compiler cannot know to be uniform (it came out of LDS): into a scalar register
__device__ __forceinline__ int uniform_t(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniform_t(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double uniform_t(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// SCALAR = true: the header's thirteen numbers go through readfirstlane — they sit in scalar registers (or their spill lanes)
// instead of thirteen vector registers for the whole pass loop (k_trace_refill, whose occupancy is decided by vector registers)
template <class T, bool SCALAR = false> __device__ __forceinline__ FlatGrid<T> flat_grid_header(const Scene<T>& sc) {
    FlatGrid<T> G;
    const T* g = sc.aux + sc.root;
    G.a0 = (int)g[0]; G.a1 = (int)g[1]; G.g0 = (int)g[2]; G.g1 = (int)g[3];
    G.org0 = g[4]; G.org1 = g[5]; G.inv0 = g[6]; G.inv1 = g[7]; G.size0 = g[9]; G.size1 = g[10];
    G.slack = T(4) * g[8];
    if constexpr (SCALAR) {
        G.a0 = uniform_t(G.a0); G.a1 = uniform_t(G.a1); G.g0 = uniform_t(G.g0); G.g1 = uniform_t(G.g1);
        G.org0 = uniform_t(G.org0); G.org1 = uniform_t(G.org1); G.inv0 = uniform_t(G.inv0); G.inv1 = uniform_t(G.inv1);
        G.size0 = uniform_t(G.size0); G.size1 = uniform_t(G.size1); G.slack = uniform_t(G.slack);
    }
    G.hi0 = G.org0 + G.size0 * T(G.g0); G.hi1 = G.org1 + G.size1 * T(G.g1);
    if constexpr (SCALAR) { G.hi0 = uniform_t(G.hi0); G.hi1 = uniform_t(G.hi1); }
    G.items = g + 11 + (G.g0 * G.g1 + 1);
    G.cellpack = sc.aux + sc.root_pack;  // (the host launches this walk only when the packed cells exist)
    return G;
}
template <class T, uint32_t F> __device__ __forceinline__ Hit<T> rebuild_hit(const Scene<T>& sc, const RayState<T>& r, int32_t node, T t);
template <class T, uint32_t F, int GATE>
__device__ __forceinline__ Hit<T> flat_grid_hit(const Scene<T>& sc, const FlatGrid<T>& G, const RayState<T>& r, bool active, const GateCtx& gate,
                                                const FlatLds<T>& L, int lane OT_FLAT_STAMP_PARAMS) {
    constexpr bool F32 = sizeof(T) == 4;
    Hit<T> best;
    best.t = Num<T>::inf(); best.node = -1; best.px = best.py = best.pz = T(0);
    const RayInv<T> ri = make_inv(r.dx, r.dy, r.dz);
    L.key[lane] = ~0ull;
    if constexpr (!F32) L.node[lane] = 0x7fffffff;
    // DDA start (root_grid_hit's, per lane)
    const int a0 = G.a0, a1 = G.a1, g0 = G.g0, g1 = G.g1;
    const T org0 = G.org0, org1 = G.org1, inv0 = G.inv0, inv1 = G.inv1, size0 = G.size0, size1 = G.size1;
    const T o0 = pick(a0, r.ox, r.oy, r.oz), o1 = pick(a1, r.ox, r.oy, r.oz);
    const T d0 = pick(a0, r.dx, r.dy, r.dz), d1 = pick(a1, r.dx, r.dy, r.dz);
    const T i0 = pick(a0, ri.inv[0], ri.inv[1], ri.inv[2]), i1 = pick(a1, ri.inv[0], ri.inv[1], ri.inv[2]);
    const bool par0 = abs_t(d0) <= T(1e-12), par1 = abs_t(d1) <= T(1e-12);
    bool walking = active;
    T tin = T(0), tout = Num<T>::inf();
    const T hi0 = G.hi0, hi1 = G.hi1;
    if (par0) { if (o0 < org0 || o0 > hi0) walking = false; }
    else { const T a = (org0 - o0) * i0, b = (hi0 - o0) * i0; tin = max_t(tin, min_t(a, b)); tout = min_t(tout, max_t(a, b)); }
    if (par1) { if (o1 < org1 || o1 > hi1) walking = false; }
    else { const T a = (org1 - o1) * i1, b = (hi1 - o1) * i1; tin = max_t(tin, min_t(a, b)); tout = min_t(tout, max_t(a, b)); }
    if (!(tin <= tout)) walking = false;
    auto clampi = [](T c, int n) { return c <= T(0) ? 0 : (c >= T(n - 1) ? n - 1 : (int)c); };
    int c0 = clampi((o0 + tin * d0 - org0) * inv0, g0), c1 = clampi((o1 + tin * d1 - org1) * inv1, g1);
    const int s0 = d0 > T(0) ? 1 : -1, s1 = d1 > T(0) ? 1 : -1;
    T tmax0 = par0 ? Num<T>::inf() : (org0 + size0 * T(c0 + (s0 > 0 ? 1 : 0)) - o0) * i0;
    T tmax1 = par1 ? Num<T>::inf() : (org1 + size1 * T(c1 + (s1 > 0 ? 1 : 0)) - o1) * i1;
    const T dt0 = par0 ? Num<T>::inf() : size0 * abs_t(i0), dt1 = par1 ? Num<T>::inf() : size1 * abs_t(i1);
    const T* items = G.items;
    const T* cellpack = G.cellpack;
    const T slack = G.slack;
    // (the host sizes the queue for 64 lanes x FLAT_CELLS x the fullest cell: the pairs of a round always fit)
    bool first_round = true;
    while (true) {  // rounds (wave-uniform)
        const unsigned long long wmask = __ballot(walking);
        if (wmask == 0ull) break;
        // ---- walk: item ranges only.  A round walks FLAT_CELLS cells per lane; once at most half of the lanes are still
        // walking it takes twice as many — their pairs fit the same queue, the slots of such a round are half empty
        // anyway, and a round less is a scan, two fences and a partly filled slot less.
        const int ncell = 2 * __popcll(wmask) <= 64 ? 2 * FLAT_CELLS : FLAT_CELLS;
        int kb[2 * FLAT_CELLS], ke[2 * FLAT_CELLS], cnt = 0;
        T covered = T(0);
        bool left = !walking;
        const int c0_was = c0, c1_was = c1;  // (a lane whose pairs do not fit the queue this round walks these cells again)
        const T tmax0_was = tmax0, tmax1_was = tmax1;
#pragma unroll
        for (int w = 0; w < 2 * FLAT_CELLS; ++w) {
            kb[w] = ke[w] = 0;
            if (w < ncell && !left) {  // (w < ncell: wave-uniform)
                const int pk = (int)cellpack[__mul24(c1, g0) + c0];  // first item | count << 11: one LDS word per cell instead of two
                kb[w] = pk & 2047;
                ke[w] = kb[w] + (pk >> 11);
                cnt += pk >> 11;
                covered = min_t(tmax0, tmax1);
                if (tmax0 < tmax1) { c0 += s0; tmax0 += dt0; if (c0 < 0 || c0 >= g0) left = true; }
                else { c1 += s1; tmax1 += dt1; if (c1 < 0 || c1 >= g1) left = true; }
            }
        }
        // The leaf a ray starts on lies in the first cell of its walk and can never be hit (test_leaf: idx == r.last): it is
        // taken out here, before it costs a pair — one candidate in six on cfg 3.  (The search runs to the fullest first
        // cell of the wave, four independent reads at a time; a leaf is listed once per cell.)
        int skip_at = -1;
        if (first_round) {
            const int c = ke[0] - kb[0];
            const int c0max = __builtin_amdgcn_readlane(wave_incl_max_i32(c), 63);
            for (int j0 = 0; j0 < c0max; j0 += 4) {
                int it[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) it[u] = j0 + u < c ? (int)items[kb[0] + j0 + u] : -2;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (it[u] == r.last) skip_at = j0 + u;
            }
            if (skip_at >= 0) cnt -= 1;
            first_round = false;
        }
        OT_FLAT_AT(5);
        OT_FLAT_COUNT(10);
        // ---- queue: one scan of the counts gives every lane the place of its pairs in the round's queue.  A lane does not
        // write its pairs one by one (a loop to the fullest cell with a predicated store per item was a sixth of the
        // kernel): it writes ONE 16-bit marker per item range — (lane << 10 | first index into the grid's item list) + 1 at
        // the queue position where the range begins; everything else in the queue is zero.  The lane that tests pair q finds
        // the last marker at or before q with a max-scan over the slot (see below) and takes the item (q - marker
        // position) places after the marker's.  At most five ranges per lane: the cells of the round, the first one split
        // around the start leaf.
        int total;
        int off = wave_excl_scan_i32(cnt, total);
        // The queue holds 512 pairs or more, not the worst case of 64 lanes with the fullest cells (the host sizes it:
        // LDS per wave decides how many waves run).  When a round's pairs do not fit (wave-uniform, rare: cfg 3 never),
        // the lanes from the first one that does not fit onward sit the round out: they go back to where their walk stood
        // and queue the same cells next round.  The pairs of one lane always fit (host), so every round gets somewhere.
        bool deferred = false;
        if (total > L.queue_cap) {
            deferred = cnt > 0 && off + cnt > L.queue_cap;
            total = wave_all_min_i32(deferred ? off : 0x7fffffff);  // offsets rise with the lane: everything before it fits
            if (deferred) { c0 = c0_was; c1 = c1_was; tmax0 = tmax0_was; tmax1 = tmax1_was; left = false; covered = T(0); }
        }
        auto mark = [&](int at, int first) { if (!deferred) L.queue[at] = (uint16_t)(((lane << 10) | first) + 1); };
#pragma unroll
        for (int w = 0; w < 2 * FLAT_CELLS; ++w) {
            if (w >= ncell) break;  // wave-uniform
            const int c = ke[w] - kb[w];
            if (w == 0 && skip_at >= 0) {
                if (skip_at > 0) mark(off, kb[0]);
                if (skip_at < c - 1) mark(off + skip_at, kb[0] + skip_at + 1);
                off += c - 1;
            } else {
                if (c > 0) mark(off, kb[w]);
                off += c;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        OT_FLAT_AT(6);
        // ---- test: 64 pairs at a time, every lane a real candidate
        // The planar test of test_leaf WITHOUT its early exits: the 64 lanes hold 64 different pairs at 64 different stages
        // of rejection, so an exit saves nothing — the wave runs on until its last lane is through — while every exit costs
        // an exec-mask save / restore (the slot was 336 instructions, 57 of them register moves and 21 such branches,
        // before).  Everything is evaluated, the verdict is one conjunction.  The arithmetic is test_leaf's, expression by
        // expression with the roundings written down (dot3_t, fma_t): bit-identical results
        // (test_pair_queue_walk_equals_per_lane_walk).
        struct Cand { int src, item; bool ok; T t; };
        auto eval = [&](int q, int src, int item) -> Cand {
            Cand cd;
            // the ray of the pair, straight from its owner's registers (ds_bpermute: a crossbar read, no bank conflicts)
            const T sx = __shfl(r.ox, src, 64), sy = __shfl(r.oy, src, 64), sz = __shfl(r.oz, src, 64);
            const T sdx = __shfl(r.dx, src, 64), sdy = __shfl(r.dy, src, 64), sdz = __shfl(r.dz, src, 64);
            const T slen = __shfl(r.len, src, 64);
            const int slast = __shfl(r.last, src, 64);
            const DNode<T>& nd = *record_at<true>(sc.nodes, item);
            OT_FLAT_AT(11);
            const T rx = sx - nd.org[0], ry = sy - nd.org[1], rz = sz - nd.org[2];
            const T lox = dot3_t(nd.M[0], rx, nd.M[3], ry, nd.M[6], rz);
            const T ldx = dot3_t(nd.M[0], sdx, nd.M[3], sdy, nd.M[6], sdz);
            const T s = -lox;
            const T t = div_t(s, ldx);
            const T loy = dot3_t(nd.M[1], rx, nd.M[4], ry, nd.M[7], rz), loz = dot3_t(nd.M[2], rx, nd.M[5], ry, nd.M[8], rz);
            const T ldy = dot3_t(nd.M[1], sdx, nd.M[4], sdy, nd.M[7], sdz), ldz = dot3_t(nd.M[2], sdx, nd.M[5], sdy, nd.M[8], sdz);
            const T Px = fma_t(t, ldx, lox), Py = fma_t(t, ldy, loy), Pz = fma_t(t, ldz, loz);
            // (operands read up front and the verdict formed with & and |, not && and ||: with the short-circuit forms the
            // compiler reads r2 / p[] inside branches on the shape and leaves the slot through a dozen exec-mask
            // branches — ~40 scalar instructions per slot for two LDS words saved)
            const int sh = nd.shape;
            const T r2 = nd.r2, p0 = nd.p[0], p1 = nd.p[1];
            const bool in_circle = dot3_t(Px, Px, Py, Py, Pz, Pz) <= r2;
            const bool in_rect = (abs_t(Py) <= p0) & (abs_t(Pz) <= p1);
            bool inside = sh == OT_SHAPE_CIRCLE ? in_circle : ((sh == OT_SHAPE_RECT) & in_rect);
            bool ok = (q < total) & (item != slast) & (ldx != T(0)) & (s != T(0)) & ((s > T(0)) == (ldx > T(0)));
            ok = ok & !((abs_t(t) < Num<T>::eps_t()) | (t < T(0)) | (t > slen));
            // (no look at the ray's best so far: it could only save the atomic, and costs an LDS read, the decoding and two
            // comparisons in every slot — cfg 3: 2.80 against 2.89 ms without it)
            if constexpr (F & F_POLY) {  // polygon / boolean apertures: only for the few pairs that got this far
                if (ok && sh != OT_SHAPE_CIRCLE && sh != OT_SHAPE_RECT) inside = planar_boundary<T, F>(sc, nd, Px, Py, Pz);
            }
            ok = ok && inside;
            // (the leaf's own AABB test, component_group.py:104-107, can only turn a would-be hit down.  Here it would run
            // in every slot for the one lane in six that holds a would-be hit — ~50 instructions at a sixth of the lanes, a
            // fifth of the kernel's VALU work on cfg 3.  The owner of the ray applies it to the WINNER after the last round
            // instead; see below for the ray whose winner fails it.)
            cd.src = src; cd.item = item; cd.ok = ok; cd.t = t;
            return cd;
        };
        auto commit = [&](const Cand& cd) {
            if (cd.ok) {  // the candidate beat what this lane saw: let the table decide
                if constexpr (F32) {
                    const unsigned long long mine = ((unsigned long long)__float_as_uint(cd.t) << 32) | (unsigned long long)(unsigned)cd.item;
                    // nothing is read back: the slot ends with the atomic, and the next slot's reads are already on their way.
                    // The owner of the ray forms the hit point of the winner itself after the last round (rebuild_hit)
                    __hip_atomic_fetch_min(&L.key[cd.src], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    // double precision: t does not fit a 64-bit key next to the index.  The key is t alone; a lane that
                    // LOWERS the minimum clears the node table's entry, then every lane whose t equals the minimum takes part
                    // in an atomic minimum over the node indices (instructions of one wave run in order: clear before vote)
                    const unsigned long long mine = (unsigned long long)__double_as_longlong(cd.t);  // t > 0: bits order like values
                    const unsigned long long old = atomicMin(&L.key[cd.src], mine);
                    if (mine < old) L.node[cd.src] = 0x7fffffff;
                    if (L.key[cd.src] == mine) atomicMin(&L.node[cd.src], cd.item);
                }
            }
        };
        // (two slots evaluated before either commits, so that the LDS round trips of one overlap the arithmetic of the
        // other: measured twice, no gain — the slot is bound by instruction issue, not by latency)
        int carry = -1;  // the last marker before this slot
        for (int q0 = 0; q0 < total; q0 += 64) {
            OT_FLAT_COUNT(9);
            const int q = q0 + lane;
            const int v = q < total ? (int)L.queue[q] : 0;
            if (v) L.queue[q] = 0;  // (the queue is all zeros again when the round is over)
            const int m = max(wave_incl_max_i32(v ? (q << 16) | v : -1), carry);  // position << 16 | marker: later ranges are larger
            carry = __builtin_amdgcn_readlane(m, 63);
            const int first = ((m & 0xffff) - 1) & 1023, src = ((m & 0xffff) - 1) >> 10;
            commit(eval(q, src, q < total ? (int)items[first + (q - (m >> 16))] : 0));
            OT_FLAT_AT(7);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        OT_FLAT_AT(7);
        // ---- verdict: done when the best hit lies inside the covered part of the ray, or the walk left the grid
        const unsigned long long mine = L.key[lane];
        if (mine != ~0ull) {
            if constexpr (F32) { best.t = __uint_as_float((unsigned)(mine >> 32)); best.node = (int)(mine & 0xffffffffull); }
            else { best.t = __longlong_as_double((long long)mine); best.node = L.node[lane]; }
        }
        if (walking) walking = !left && !(best.t + slack < covered);
        OT_FLAT_AT(8);
    }
    if (!(active && best.node >= 0)) { best.t = Num<T>::inf(); best.node = -1; }
    // The winner's own AABB test, by the owner of the ray.  The winner is the nearest candidate that passes every test BUT
    // this one: if it passes this one too it is the nearest that passes all of them.  If it does not (the top-level grid
    // is only built over boxes that hold their surfaces, so this is a hit within rounding of its box's face), the nearest
    // valid candidate is unknown and the ray takes the per-lane walk, which applies every test to every candidate.
    bool redo = false;
    if (best.node >= 0) {
        const DNode<T>& nd = *record_at<true>(sc.nodes, best.node);
        T u1, u2;
        if (nd.flags & OT_NODE_CHECK_AABB) redo = !slab_inv(r.ox, r.oy, r.oz, ri, nd.aabb, u1, u2);
    }
#ifndef OT_EXP_NOREDO
    if (__any(redo)) {
        if (redo) {
            best.t = Num<T>::inf(); best.node = -1; best.px = best.py = best.pz = T(0);
            root_grid_hit<T, F, GATE>(sc, r, ri, best, gate);
        }
    }
#endif
    // the hit point of the winner, from the ray's own values and the winner's t: the expressions of the test, the same bits
    if (best.node >= 0 && !redo) return rebuild_hit<T, F>(sc, r, best.node, best.t);
    return best;
}

// Nearest hit over the whole scene for one ray (all lanes of the wave walk the node list with
// the same index; a lane that pruned a group idles until the list leaves that group, and when
// every lane of the wave pruned it the wave jumps ahead to the smallest skip target).
#ifdef OT_STAMP  // diagnostic build: where the linear pass spends its cycles — st_acc[5] group boxes and loop, [6] planar leaves, [7] gridded
                 // groups (cell lookup + children), [8] deferred curved leaves
#define OT_NH_STAMP_PARAMS , unsigned long long* st_acc = nullptr, unsigned long long* st_last = nullptr
#define OT_NH_AT(k) do { if (st_acc) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long _t = __builtin_amdgcn_s_memtime(); st_acc[k] += _t - *st_last; *st_last = _t; } } while (0)
#elif defined(OT_MARK)
#define OT_NH_STAMP_PARAMS
#define OT_NH_AT(k) asm volatile("; OT_MARK nh " #k)
#else
#define OT_NH_STAMP_PARAMS
#define OT_NH_AT(k) do {} while (0)
#endif
template <class T, uint32_t F, int GATE>
__device__ __forceinline__ Hit<T> nearest_hit(const Scene<T>& sc, const RayState<T>& r, bool active, const GateCtx& gate OT_NH_STAMP_PARAMS) {
    Hit<T> best;
    best.t = Num<T>::inf();
    best.node = -1;
    best.px = best.py = best.pz = T(0);
    int skip_until = active ? 0 : 0x7fffffff;
    RayInv<T> ri;
    if constexpr (F & F_AABB) ri = make_inv(r.dx, r.dy, r.dz);
    if constexpr (F & F_ROOT) {
        // presets without F_SUBTREE are only launched for scenes that have a top-level grid over leaves: no linear pass
        if ((F & F_SUBTREE) == 0 || sc.root >= 0) {  // scene-uniform branch
            if (active) root_grid_hit<T, F, GATE>(sc, r, ri, best, gate);
            return best;
        }
    }
    // Curved leaves wait until the planar ones have been tested (scenes without count gates: the order of the tests is
    // free there, exact ties go to the lower node index either way).  A planar test costs ~50 instructions, an asphere
    // search ~400, and behind a nearer planar hit the curved leaf's AABB entry lies beyond the best hit: it is pruned
    // without a search (cfg 5: a ray coming back from the micro-mirrors meets the flat back of the lens before its
    // aspheric front, which the list names first).  The pending leaves are wave-uniform node indices with a per-lane
    // membership mask.
    constexpr bool DEFER_CURVED = (F & F_CURVED) != 0 && (F & F_LIMIT) == 0;
    int pend0 = 0, pend1 = 0, pend2 = 0, pend3 = 0, n_pend = 0;  // (four scalars: an array indexed by a loop variable would live in scratch)
    uint32_t my_pend = 0;
    int i = 0;
  more_nodes:
    for (; i < sc.n_nodes; ++i) {  // virtual indices; the children of an instanced run are only ever reached through their group's grid
        const NodeRef<T> nr = node_ref<T, F>(sc, i);
        const DNode<T>& nd = *nr.nd;
        if constexpr (F & F_AABB) {
            bool inside = false;
            T t1 = T(0), t2 = T(0);
            if (nd.kind == OT_NODE_GROUP && (nd.flags & OT_NODE_CHECK_AABB)) {
                if (i >= skip_until) {
                    inside = slab_inv(r.ox, r.oy, r.oz, ri, nr.geo + 3, t1, t2);
                    if (!(F & F_LIMIT) && (nd.flags & OT_NODE_BOX_TRUSTED) && beyond_best(t1, best.t)) inside = false;  // nothing in this group can be nearer
                    if (!inside) skip_until = nd.end;
                }
            }
            if (nd.kind == OT_NODE_GROUP) {
                if constexpr (F & F_GRID) {
                    if (nd.flags & OT_NODE_GRID) {  // wave-uniform branch; lanes that missed the box idle inside
                        OT_NH_AT(5);
                        if (inside) {
                            grid_children<T, F, GATE>(sc, nd, r, ri, t1, t2, best, gate);
                            skip_until = nd.end;
                        }
                        OT_NH_AT(7);
                    }
                }
                // Every lane of the wave takes this branch (nd is wave-uniform).  If no lane wants the group's first child the
                // wave jumps over the group: a lane that is skipping holds the end of this group or of one that encloses it
                // (inactive lanes INT_MAX), so none of them has business before nd.end.  One ballot; the first version took
                // the wave-wide minimum of skip_until here, six dependent cross-lane steps per group node.
                if (!__any(skip_until <= i + 1)) i = nd.end - 1;
                continue;
            }
        }
        if constexpr (DEFER_CURVED) {
            const int sh = nd.shape;
            if (!(sh == OT_SHAPE_CIRCLE || sh == OT_SHAPE_RECT || sh == OT_SHAPE_POLYGON2D || sh == OT_SHAPE_CSG)) {  // wave-uniform
                const bool want = i >= skip_until;
                if (__any(want)) {
                    if (n_pend == 4) break;  // (wave-uniform) test what is pending first: ONE call site of the curved-leaf test
                    if (n_pend == 0) pend0 = i; else if (n_pend == 1) pend1 = i; else if (n_pend == 2) pend2 = i; else pend3 = i;
                    if (want) my_pend |= 1u << n_pend;
                    ++n_pend;
                }
                continue;
            }
            if (i < skip_until) continue;
            OT_NH_AT(5);
            test_leaf<T, F, GATE, false, (F & F_AABB) != 0, 1>(sc, nr, i, r, best, gate, &ri);
            OT_NH_AT(6);
        } else {
            if (i < skip_until) continue;
            test_leaf<T, F, GATE, true, (F & F_AABB) != 0>(sc, nr, i, r, best, gate, &ri);
        }
    }
    OT_NH_AT(5);
    if constexpr (DEFER_CURVED) {
#pragma unroll 1
        for (int c = 0; c < n_pend; ++c) {  // one copy of the curved-leaf test in the instruction stream
            const int pn = c == 0 ? pend0 : (c == 1 ? pend1 : (c == 2 ? pend2 : pend3));
            if ((my_pend >> c) & 1u) test_leaf<T, F, GATE, false, (F & F_AABB) != 0, 2>(sc, node_ref<T, F>(sc, pn), pn, r, best, gate, &ri);
        }
        n_pend = 0;
        my_pend = 0;
        if (i < sc.n_nodes) goto more_nodes;  // (the list was full: the walk stopped before pushing node i)
    }
    OT_NH_AT(8);
    return best;
}

// The hit record of a decision taken earlier (k_gen_pass: the count pass stores node and t, the emit pass rebuilds the
// local hit point from them instead of searching again): the very expressions test_leaf / hit_leaf form the point with,
// so the interaction sees the same bits.
template <class T, uint32_t F> __device__ __forceinline__ Hit<T> rebuild_hit(const Scene<T>& sc, const RayState<T>& r, int32_t node, T t) {
    Hit<T> h;
    h.node = node; h.t = t; h.px = h.py = h.pz = T(0);
    if (node < 0) return h;
    const NodeRef<T> nr = node_ref<T, F>(sc, node);
    const DNode<T>& nd = *nr.nd;
    const T rx = r.ox - nr.geo[0], ry = r.oy - nr.geo[1], rz = r.oz - nr.geo[2];
    const int sh = nd.shape;
    const bool planar = sh == OT_SHAPE_CIRCLE || sh == OT_SHAPE_RECT || sh == OT_SHAPE_POLYGON2D || sh == OT_SHAPE_CSG;
    if (!(F & F_CURVED) || planar) {
        const T lox = dot3_t(nd.M[0], rx, nd.M[3], ry, nd.M[6], rz), ldx = dot3_t(nd.M[0], r.dx, nd.M[3], r.dy, nd.M[6], r.dz);
        const T loy = dot3_t(nd.M[1], rx, nd.M[4], ry, nd.M[7], rz), loz = dot3_t(nd.M[2], rx, nd.M[5], ry, nd.M[8], rz);
        const T ldy = dot3_t(nd.M[1], r.dx, nd.M[4], r.dy, nd.M[7], r.dz), ldz = dot3_t(nd.M[2], r.dx, nd.M[5], r.dy, nd.M[8], r.dz);
        h.px = fma_t(t, ldx, lox); h.py = fma_t(t, ldy, loy); h.pz = fma_t(t, ldz, loz);
    } else {
        T ox, oy, oz, dx, dy, dz;
        to_local(nd, rx, ry, rz, ox, oy, oz);
        to_local(nd, r.dx, r.dy, r.dz, dx, dy, dz);
        h.px = ox + t * dx; h.py = oy + t * dy; h.pz = oz + t * dz;
    }
    return h;
}

// a / b for complex numbers with one real division
template <class T> __device__ __forceinline__ void cdiv(T ar, T ai, T br, T bi, T& cr, T& ci) {
    const T inv = rcp_t(br * br + bi * bi);
    cr = (ar * br + ai * bi) * inv;
    ci = (ai * br - ar * bi) * inv;
}

template <class T, uint32_t F>
__device__ __forceinline__ void surf_normal(const Scene<T>& sc, const DNode<T>& nd, T Px, T Py, T Pz, T& nx, T& ny, T& nz) {
    nx = T(1); ny = T(0); nz = T(0);  // Plane._normal
    if constexpr (F & F_POLY) {
        if (nd.shape == OT_SHAPE_POLYGON2D) {
            const T* rec = sc.aux + nd.aux;
            nx = rec[1]; ny = rec[2]; nz = rec[3];
        }
    }
    if constexpr (F & F_CURVED) {
        switch (nd.shape) {
            case OT_SHAPE_SPHERE: nx = qd(Px, nd.p[0]); ny = qd(Py, nd.p[0]); nz = qd(Pz, nd.p[0]); return;
            case OT_SHAPE_CYLINDER:
                if constexpr (F & F_MISC) { nx = qd(Px, nd.p[0]); ny = qd(Py, nd.p[0]); nz = T(0); }
                return;
            case OT_SHAPE_POLYGON3D:
                if constexpr (F & F_MISC) {
                    const T* rec = sc.aux + nd.aux;
                    nx = rec[1]; ny = rec[2]; nz = rec[3];
                }
                return;
            case OT_SHAPE_ASPHERE_CHEB:
            case OT_SHAPE_ASPHERE_PARAM:
            case OT_SHAPE_ASPHERE_EXACT: {  // surfaces.py:380-388
                const T r = sqrt_t(Py * Py + Pz * Pz);
                if (r < T(1e-12)) return;
                T s;
                if constexpr ((F & F_MISC) != 0) s = nd.shape == OT_SHAPE_ASPHERE_CHEB ? cheb_d1(sc.aux + nd.aux, nd.p[0], r) : sag_d1(nd, r);
                else s = sag_d1(nd, r);
                const T ay = s * qd(Py, r), az = s * qd(Pz, r);
                const T inv = rsqrt_t(T(1) + ay * ay + az * az);
                nx = inv; ny = ay * inv; nz = az * inv;
                return;
            }
            default: return;
        }
    }
}

// Children of a hit.  Writes the first MAXK lab-frame children into kids[] and returns how many
// the interaction emits (which can exceed MAXK = 1: the fused kernel treats that as "this tree
// branches" and hands the ray back to the host, see k_trace_fused).
// SINK (MAXK > 1): a callable that takes every outgoing ray the moment it is complete, instead of `kids` — the lane-per-tree
// kernel queues the first child before the second is formed, so that the two are never live together (k_trace_trees).
struct NoSink {};
template <class T, uint32_t F, int MAXK, class SINK = NoSink>
__device__ __forceinline__ int interact(const Scene<T>& sc, const RayState<T>& r, const Hit<T>& h, RayState<T>* kids,
                                        const MatCache<T>& mc, SINK* sink = nullptr) {
    const NodeRef<T> nr = node_ref<T, F>(sc, h.node);
    const DNode<T>& nd = *nr.nd;
    if (nd.inter == OT_INT_BLOCK) return 0;
    T dx, dy, dz;  // incoming direction in the leaf frame
    to_local(nd, r.dx, r.dy, r.dz, dx, dy, dz);
    const T t = h.t;
    const T q1r = r.qr + t, q1i = r.qi;  // q_at_z (ray.py:17-19)
    const T pl_hit = r.pl + t * r.n;     // Ray.pathlength(t) (ray.py:145-147)
    T Ox, Oy, Oz;
    to_lab(nd, h.px, h.py, h.pz, Ox, Oy, Oz);
    Ox += nr.geo[0]; Oy += nr.geo[1]; Oz += nr.geo[2];
    int nk = 0;
    // One outgoing ray at most (the non-branching kernels): the branches below only choose the outgoing direction in the
    // leaf frame and what rides along; normalising, turning it into the lab frame and filling the record happen ONCE behind
    // them.  With `emit` expanded inside every branch a pass whose rays meet a lens, a mirror and a glass face ran that
    // tail three times at partial lanes.  Same expressions in the same order either way: the same bits.
    T out_x = T(0), out_y = T(0), out_z = T(0), out_I = T(0), out_qr = T(0), out_qi = T(0), out_n = T(0), out_pl = T(0);
    auto emit = [&](T lx, T ly, T lz, T I, T qr, T qi, T n, T pl) {
        if constexpr (MAXK == 1) {
            if (nk == 0) { out_x = lx; out_y = ly; out_z = lz; out_I = I; out_qr = qr; out_qi = qi; out_n = n; out_pl = pl; }
        } else if (nk < MAXK) {
            RayState<T> k;
            const T inv = rsqrt_t(lx * lx + ly * ly + lz * lz);
            to_lab(nd, lx * inv, ly * inv, lz * inv, k.dx, k.dy, k.dz);
            k.ox = Ox; k.oy = Oy; k.oz = Oz;
            k.wl = r.wl; k.has_q = r.has_q; k.len = Num<T>::inf(); k.last = h.node;
            k.I = I; k.qr = qr; k.qi = qi; k.n = n; k.pl = pl;
            // constant indices only: kids[nk] with a run-time nk would put both children into private scratch
            // (240 B per lane in the fp64 generation kernel of round 1)
            if constexpr (!std::is_same<SINK, NoSink>::value) (*sink)(k);
            else if (nk == 0) kids[0] = k;
            else kids[MAXK > 1 ? 1 : 0] = k;
        }
        ++nk;
    };
    auto finish = [&]() -> int {
        if constexpr (MAXK == 1) {
            if (nk > 0) {
                RayState<T> k;
                const T inv = rsqrt_t(out_x * out_x + out_y * out_y + out_z * out_z);
                to_lab(nd, out_x * inv, out_y * inv, out_z * inv, k.dx, k.dy, k.dz);
                k.ox = Ox; k.oy = Oy; k.oz = Oz;
                k.wl = r.wl; k.has_q = r.has_q; k.len = Num<T>::inf(); k.last = h.node;
                k.I = out_I; k.qr = out_qr; k.qi = out_qi; k.n = out_n; k.pl = out_pl;
                kids[0] = k;
            }
        }
        return nk;
    };
    if constexpr (F & F_LENS) {
        if (nd.inter == OT_INT_LENS) {  // optical_component.py:930-948
            T qr = r.qr, qi = r.qi;
            const T jf = nd.inv_focal;
            if (r.has_q) cdiv(q1r, q1i, T(1) - q1r * jf, -q1i * jf, qr, qi);
            emit(dx - h.px * jf, dy - h.py * jf, dz - h.pz * jf, r.I * nd.trans, qr, qi, r.n, r.pl);  // pathlength, n unchanged
            return finish();
        }
    }
    T nx, ny, nz;
    surf_normal<T, F>(sc, nd, h.px, h.py, h.pz, nx, ny, nz);
    const T dn = dx * nx + dy * ny + dz * nz;
    if (!(F & F_REFRACT) || nd.inter == OT_INT_MIRROR) {  // optical_component.py:536-570
        if (nd.refl > T(0)) emit(dx - T(2) * dn * nx, dy - T(2) * dn * ny, dz - T(2) * dn * nz, r.I * nd.refl, q1r, q1i, r.n, pl_hit);
        if (nd.trans > T(0)) emit(dx, dy, dz, r.I * nd.trans, q1r, q1i, r.n, pl_hit);
        return finish();
    }
    if constexpr (F & F_REFRACT) {  // optical_component.py:617-717
        const T n1 = cached_index<T, F>(sc, mc, nd.mat1, r.wl), n2 = cached_index<T, F>(sc, mc, nd.mat2, r.wl);
        T ROC = Num<T>::inf();
        if (nd.roc_kind == OT_ROC_CONST) ROC = nd.roc;
        if constexpr (F & F_CURVED) {
            if (nd.roc_kind == OT_ROC_ASPHERE) {  // surfaces.py:362-373
                const T rr = sqrt_t(h.py * h.py + h.pz * h.pz);
                T s, c;
                bool series = false;
                if constexpr ((F & F_MISC) != 0) series = nd.shape == OT_SHAPE_ASPHERE_CHEB;
                if (series) { s = cheb_d1(sc.aux + nd.aux, nd.p[0], rr); c = cheb_d2(sc.aux + nd.aux, nd.p[0], rr); }
                else { s = sag_d1(nd, rr); c = sag_d2(nd, rr); }
                const T w = T(1) + s * s;
                ROC = qd(w * sqrt_t(w), c);
            }
        }
        T nin = n1, nout = n2;
        if (!(dn < T(0))) { nin = n2; nout = n1; ROC = -ROC; }
        const T ratio = div_t(nin, nout);
        T qtr = r.qr, qti = r.qi, qrr = r.qr, qri = r.qi;
        if (r.has_q) {
            if (nd.roc_kind == OT_ROC_INF) {  // flat interface: C = 0, so q_t = q1 / (nin/nout) and q_r = q1
                const T back = rcp_t(ratio);
                qtr = q1r * back; qti = q1i * back; qrr = q1r; qri = q1i;
            } else {
                const T Cc = qd(nin - nout, ROC * nout), Cr = qd(T(2), ROC);
                cdiv(q1r, q1i, Cc * q1r + ratio, Cc * q1i, qtr, qti);
                cdiv(q1r, q1i, Cr * q1r + T(1), Cr * q1i, qrr, qri);
            }
        }
        const T ci = min_t(max_t(dn, T(-1)), T(1));
        const T si = sqrt_t(T(1) - ci * ci), st = ratio * si;
        if (st < T(1)) {
            if (nd.trans > T(0)) {
                const T ct = sqrt_t(T(1) - st * st), sgn = dn > T(0) ? T(1) : T(-1);
                emit(ratio * (dx - dn * nx) + ct * sgn * nx, ratio * (dy - dn * ny) + ct * sgn * ny,
                     ratio * (dz - dn * nz) + ct * sgn * nz, r.I * nd.trans, qtr, qti, nout, pl_hit);
            }
        } else {  // total internal reflection: full intensity
            emit(dx - T(2) * ci * nx, dy - T(2) * ci * ny, dz - T(2) * ci * nz, r.I, qrr, qri, r.n, pl_hit);
        }
        if (nd.refl > T(0))
            emit(dx - T(2) * ci * nx, dy - T(2) * ci * ny, dz - T(2) * ci * nz, r.I * nd.refl, qrr, qri, r.n, pl_hit);
    }
    return finish();
}

}  // namespace ot
