// misc_kernels.h — the small non-template kernels of the library (one translation unit: optable_hip.hip) and the
// device-wide exclusive scan they share.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.h"

// ------------------------------------------------------------------------------------------
// Exclusive prefix sum over n elements in three streaming launches: tile sums -> scan of the tile sums (one
// workgroup) -> tile-local scan + tile offset.  Every scan of the library is small next to the pass it serves (wave
// totals of a generation: n / 64 words; Monitor.record: 4 of the ~110 bytes its test reads per slot), so the second
// read of the input is in the noise, and the library carries three kernels of its own instead of a scan library's
// several hundred tuning variants.
//   In  int32 flags / counts, or packed 2 x 32-bit totals in one 64-bit word (the halves never carry into each other)
//   Out int32 / int64 / uint64
static constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <class V> __device__ __forceinline__ V wave_incl_scan(V v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const V o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}
// exclusive scan of one value per thread over a 256-thread workgroup; `total` = the workgroup's sum (all threads)
template <class V> __device__ __forceinline__ V block_excl_scan(V v, V& total) {
    __shared__ V wsum[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const V incl = wave_incl_scan(v, lane);
    __syncthreads();  // wsum may still be read by the previous call
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    V before = V(0), all = V(0);
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        if (w < wave) before += wsum[w];
        all += wsum[w];
    }
    total = all;
    return before + incl - v;
}
template <class In, class Out>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(const In* __restrict__ in, int64_t n, Out* __restrict__ tile_sum) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    Out s = Out(0);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) s += (Out)in[base + k];
    Out total;
    (void)block_excl_scan(s, total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}
template <class Out> __global__ __launch_bounds__(SCAN_THREADS) void k_scan_spine(Out* tile_sum, int64_t n_tiles) {
    Out carry = Out(0);
    for (int64_t b = 0; b < n_tiles; b += SCAN_THREADS) {
        const int64_t i = b + threadIdx.x;
        const Out v = i < n_tiles ? tile_sum[i] : Out(0);
        Out total;
        const Out ex = block_excl_scan(v, total);
        if (i < n_tiles) tile_sum[i] = carry + ex;
        carry += total;
    }
}
template <class In, class Out>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const In* __restrict__ in, int64_t n, const Out* __restrict__ tile_sum,
                                                             Out* __restrict__ out) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    Out v[SCAN_ITEMS];
    Out s = Out(0);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = base + k < n ? (Out)in[base + k] : Out(0);
        s += v[k];
    }
    Out total;
    Out run = block_excl_scan(s, total) + tile_sum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}
// bytes of temporary storage an exclusive_scan over n elements needs
template <class Out> static size_t scan_tmp_bytes(int64_t n) { return sizeof(Out) * (size_t)((n + SCAN_TILE - 1) / SCAN_TILE + 1); }
template <class In, class Out> static void exclusive_scan(void* tmp, const In* in, Out* out, int64_t n, hipStream_t stream) {
    if (n <= 0) return;
    const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    Out* tile_sum = (Out*)tmp;
    hipLaunchKernelGGL((k_scan_tiles<In, Out>), dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, stream, in, n, tile_sum);
    hipLaunchKernelGGL((k_scan_spine<Out>), dim3(1), dim3(SCAN_THREADS), 0, stream, tile_sum, tiles);
    hipLaunchKernelGGL((k_scan_apply<In, Out>), dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, stream, in, n, (const Out*)tile_sum, out);
}

// ------------------------------------------------------------------------------------------
// totals of a generation from the scanned wave totals: segments written and rays of the next generation.  Runs between
// the two passes: the emit pass takes its first slot from totals[2] (the cursor as it stood), so the caller's cursor
// and next-generation count can be published here and the generation needs no closing kernel.
__global__ void k_gen_totals(const unsigned long long* wave_total, const unsigned long long* wave_prefix, int64_t n_waves, int64_t* totals,
                             int64_t* cursor, int64_t* n_next) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long all = wave_prefix[n_waves - 1] + wave_total[n_waves - 1];
        totals[0] = (int64_t)(all >> 32);
        totals[1] = (int64_t)(all & 0xffffffffull);
        totals[2] = *cursor;
        *cursor += totals[0];
        *n_next = totals[1];
    }
}

// k_gen_one's per-ray budgets, seeded from the per-tree table before the first generation of a call (kernels.h)
__global__ void k_gen_seed_rem(const int32_t* __restrict__ tree, const int32_t* __restrict__ budget, int64_t n, int32_t* __restrict__ rem) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rem[i] = budget[tree[i]];
}

// k_gen_recount: the count pass of a generation whose rays were looked ahead by the emit pass that wrote them (k_gen_pass MODE 2):
// rank within the tree, budget cut and doomed children exactly as the count pass decides them, the number of children from the
// byte the emit pass left.  Writes the code byte in place of it and the wave totals.
__global__ __launch_bounds__(256) void k_gen_recount(const int32_t* __restrict__ tree, int64_t n, const int32_t* __restrict__ budget, uint8_t* code,
                                                     unsigned long long* __restrict__ wave_total, int32_t drop_doomed) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gwave = i >> 6;
    int32_t my_tree = -1;
    int start_lane = -1;
    if (i < n) {
        my_tree = tree[i];
        if (lane == 0 || tree[i - 1] != my_tree) start_lane = lane;
    }
    const int head_lane = wave_incl_max_i32(start_lane);
    int64_t head = (i - lane) + head_lane;
    if (i < n && head_lane == 0) head = tree_head(tree, i - lane);
    const int64_t bud = i < n ? (int64_t)budget[my_tree] : 0;
    const bool active = i < n && (i - head) < bud;
    bool doomed = false;
    if (active && drop_doomed) {
        const int64_t last_needed = head + bud - 1;
        doomed = last_needed < n && tree[last_needed] == my_tree;
    }
    int32_t nk = active ? (int32_t)(code[i] & 3) : 0;
    if (doomed) nk = 0;
    if (i < n) code[i] = (uint8_t)((active ? 1 : 0) | (nk << 1) | (doomed ? 8 : 0));
    const int n_act = __popcll(__ballot(active));
    int kids;
    wave_excl_scan_i32(nk, kids);
    if (lane == 0 && (gwave << 6) < n) wave_total[gwave] = ((unsigned long long)n_act << 32) | (unsigned long long)kids;
}

// rank[slot][i] = how many earlier rays of i's tree (this generation) hit limited leaf `slot`
__global__ void k_gen_rank(const int32_t* tree, int64_t n, int32_t n_slots, const int32_t* ex, int32_t* rank) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t head = tree_head(tree, i);
    for (int s = 0; s < n_slots; ++s) rank[(int64_t)s * n + i] = ex[(int64_t)s * n + i] - ex[(int64_t)s * n + head];
}
// after the trace: each tree's last ray of the generation folds the generation's hits into the table
__global__ void k_gen_counts(const int32_t* tree, const int32_t* ids, int64_t n, int32_t n_slots, const int32_t* rank,
                             const int32_t* probe, const int32_t* slot_max, int32_t* counts, int32_t n_classes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1 || tree[i + 1] != tree[i]) {
        for (int s = 0; s < n_slots; ++s) {
            if ((uint32_t)ids[i] >= (uint32_t)n_classes) break;  // id outside the table: not counted (count_gate's rule)
            int32_t* c = counts + (int64_t)s * n_classes + ids[i];
            const int32_t total = *c + rank[(int64_t)s * n + i] + probe[(int64_t)s * n + i];
            *c = total < slot_max[s] ? total : (*c > slot_max[s] ? *c : slot_max[s]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Monitor.record
__global__ void k_mon_test(ot_monitor mon, SegsT<double> s, int64_t n, const int32_t* seg_count, int64_t n_rays, int32_t* hit,
                           double* P, double* tt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (seg_count) {  // [k][ray] layout of ot_trace_*: slot i is valid iff k < |seg_count[ray]|
        const int32_t c = seg_count[i % n_rays];
        if (i / n_rays >= (c < 0 ? -c : c)) { hit[i] = 0; return; }
    } else if (n_rays < 0 && s.ray[i] < 0) {  // a list with holes: the append layout of ot_trace_append_*
        hit[i] = 0;
        return;
    }
    const double rx = s.ox[i] - mon.origin[0], ry = s.oy[i] - mon.origin[1], rz = s.oz[i] - mon.origin[2];
    const double* M = mon.M;
    const double ox = M[0] * rx + M[3] * ry + M[6] * rz, oy = M[1] * rx + M[4] * ry + M[7] * rz,
                 oz = M[2] * rx + M[5] * ry + M[8] * rz;
    double dx = M[0] * s.dx[i] + M[3] * s.dy[i] + M[6] * s.dz[i], dy = M[1] * s.dx[i] + M[4] * s.dy[i] + M[7] * s.dz[i],
           dz = M[2] * s.dx[i] + M[5] * s.dy[i] + M[8] * s.dz[i];
    const double inv = 1.0 / sqrt(dx * dx + dy * dy + dz * dz);  // ray_to_local_coordinates renormalises
    dx *= inv; dy *= inv; dz *= inv;
    int32_t ok = 0;
    if (dx != 0.0) {
        const double t = -ox / dx;
        if (!(fabs(t) < 1e-9 || t < 0.0 || t > s.len[i])) {
            const double Px = ox + t * dx, Py = oy + t * dy, Pz = oz + t * dz;
            if (fabs(Py) <= mon.half_width && fabs(Pz) <= mon.half_height) {
                ok = 1;
                P[3 * i] = Px; P[3 * i + 1] = Py; P[3 * i + 2] = Pz;
                tt[i] = t;
            }
        }
    }
    hit[i] = ok;
}
__global__ void k_mon_compact(const int32_t* hit, const int64_t* off, const double* P, const double* tt, int64_t n,
                              int64_t* hit_index, double* Px, double* Py, double* Pz, double* t, int64_t* n_hits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (hit[i]) {
        const int64_t d = off[i];
        hit_index[d] = i; Px[d] = P[3 * i]; Py[d] = P[3 * i + 1]; Pz[d] = P[3 * i + 2]; t[d] = tt[i];
    }
    if (i == n - 1) *n_hits = off[i] + hit[i];
}

