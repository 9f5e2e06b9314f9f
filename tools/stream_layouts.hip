// stream_layouts.hip — one experiment on the cfg 2 ceiling: does the OUTPUT LAYOUT bound the
// streaming rate of the trace's access pattern?  Reads one 104-byte ray record per lane (14 SoA streams) and writes K
// segment records per ray, in three layouts:
//   soa     14 separate arrays, slot k*n + i (what ot_trace_f64 writes; the library's k_stream_ceiling)
//   tiled   64-ray x 14-field tiles: a wave writes one contiguous 7 KB tile per segment (AoSoA)
//   soa K=1 the same arrays with a single segment per ray (read : write = 1 : 1)
// Four input / output sets are rotated so that nothing is served from the 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_layouts.hip -o /tmp/stream_layouts && /tmp/stream_layouts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct In { const double* f[12]; const int* id; const int* fl; };
struct OutSoa { double* f[12]; int* ray; int* surf; };

template <bool NT> __device__ __forceinline__ void st(double* p, double v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ void sti(int* p, int v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT> __global__ __launch_bounds__(256) void k_soa(In in, long n, int K, OutSoa out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v[12];
#pragma unroll
    for (int f = 0; f < 12; ++f) v[f] = __builtin_nontemporal_load(in.f[f] + i);
    const int id = in.id[i], fl = in.fl[i];
    for (int k = 0; k < K; ++k) {
        const long s = (long)k * n + i;
#pragma unroll
        for (int f = 0; f < 12; ++f) st<NT>(out.f[f] + s, v[f]);
        sti<NT>(out.ray + s, id); sti<NT>(out.surf + s, fl);
        v[0] += 1.0;
    }
}
// tile = 64 rays x (12 doubles + 2 ints) = 64 * 104 B = 6656 B, field-major inside the tile
template <bool NT> __global__ __launch_bounds__(256) void k_tiled(In in, long n, int K, char* out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v[12];
#pragma unroll
    for (int f = 0; f < 12; ++f) v[f] = __builtin_nontemporal_load(in.f[f] + i);
    const int id = in.id[i], fl = in.fl[i];
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < K; ++k) {
        const long tile = ((long)k * n + i) >> 6;
        char* base = out + tile * 6656;
#pragma unroll
        for (int f = 0; f < 12; ++f) st<NT>((double*)(base + f * 512) + lane, v[f]);
        sti<NT>((int*)(base + 6144) + lane, id); sti<NT>((int*)(base + 6400) + lane, fl);
        v[0] += 1.0;
    }
}

int main() {
    const long n = 1000000;
    const int SETS = 4;
    std::vector<In> ins(SETS);
    std::vector<OutSoa> outs(SETS);
    std::vector<char*> tiles(SETS);
    for (int s = 0; s < SETS; ++s) {
        for (int f = 0; f < 12; ++f) { double* p; CHECK(hipMalloc(&p, n * 8)); CHECK(hipMemset(p, 0, n * 8)); ins[s].f[f] = p; }
        int* q; CHECK(hipMalloc(&q, n * 4)); CHECK(hipMemset(q, 0, n * 4)); ins[s].id = q;
        CHECK(hipMalloc(&q, n * 4)); CHECK(hipMemset(q, 0, n * 4)); ins[s].fl = q;
        for (int f = 0; f < 12; ++f) CHECK(hipMalloc(&outs[s].f[f], 5 * n * 8));
        CHECK(hipMalloc(&outs[s].ray, 5 * n * 4)); CHECK(hipMalloc(&outs[s].surf, 5 * n * 4));
        CHECK(hipMalloc(&tiles[s], 5 * n * 104 + 6656));
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = (int)((n + 255) / 256);
    auto run = [&](const char* name, int K, int mode) -> int {
        for (int w = 0; w < 300; ++w) {  // clocks up
            if (mode == 0) hipLaunchKernelGGL(k_soa<true>, dim3(grid), dim3(256), 0, 0, ins[w % SETS], n, K, outs[w % SETS]);
            else if (mode == 1) hipLaunchKernelGGL(k_tiled<true>, dim3(grid), dim3(256), 0, 0, ins[w % SETS], n, K, tiles[w % SETS]);
            else if (mode == 2) hipLaunchKernelGGL(k_soa<false>, dim3(grid), dim3(256), 0, 0, ins[w % SETS], n, K, outs[w % SETS]);
            else hipLaunchKernelGGL(k_tiled<false>, dim3(grid), dim3(256), 0, 0, ins[w % SETS], n, K, tiles[w % SETS]);
        }
        CHECK(hipDeviceSynchronize());
        const int reps = 200;
        CHECK(hipEventRecord(e0));
        for (int w = 0; w < reps; ++w) {
            if (mode == 0) hipLaunchKernelGGL(k_soa<true>, dim3(grid), dim3(256), 0, 0, ins[w % SETS], n, K, outs[w % SETS]);
            else if (mode == 1) hipLaunchKernelGGL(k_tiled<true>, dim3(grid), dim3(256), 0, 0, ins[w % SETS], n, K, tiles[w % SETS]);
            else if (mode == 2) hipLaunchKernelGGL(k_soa<false>, dim3(grid), dim3(256), 0, 0, ins[w % SETS], n, K, outs[w % SETS]);
            else hipLaunchKernelGGL(k_tiled<false>, dim3(grid), dim3(256), 0, 0, ins[w % SETS], n, K, tiles[w % SETS]);
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms / reps * 1e3, bytes = (double)n * 104 * (1 + K);
        printf("%-34s K=%d  %8.2f us per launch  %7.1f GB/s  %.3f of 8 TB/s\n", name, K, us, bytes / us / 1e3, bytes / us / 1e3 / 8000);
        return 0;
    };
    if (run("soa, non-temporal stores", 5, 0)) return 1;
    if (run("tiled 64 x 14, non-temporal", 5, 1)) return 1;
    if (run("soa, plain stores", 5, 2)) return 1;
    if (run("tiled 64 x 14, plain stores", 5, 3)) return 1;
    if (run("soa, non-temporal stores", 1, 0)) return 1;
    if (run("tiled 64 x 14, non-temporal", 1, 1)) return 1;
    return 0;
}
