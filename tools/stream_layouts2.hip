// stream_layouts2.hip — round-4 follow-up to stream_layouts.hip: why is the 64-slot tile layout faster on
// some boxes and slower on others, and is there a tile arrangement that is reliably faster than the 14 slot arrays?
// The trace's streams with no tracing: one 104-byte ray record in per lane (14 SoA streams), K segment records out.
//   soa        14 arrays, slot k*n + i                                                 (ot_trace_f64)
//   tile6656   64-slot tiles of 6656 B back to back, tile = (k*n + i) / 64                 (ot_trace_tiled_f64 today)
//   tile8192   the same tiles on an 8 KiB stride (padded), base 4 KiB aligned
//   raytile    tiles grouped by RAY: tile = (i / 64) * K + k — the K tiles of one wave's 64 rays are adjacent: a wave
//              writes ONE contiguous K * 6656-byte run over its life instead of K runs that lie n * 104 bytes apart
// Four input / output sets are rotated so that nothing is served from the 256 MiB Infinity Cache; every variant runs
// ROUNDS times, interleaved with the others, so that drift of the box shows as spread inside a variant.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_layouts2.hip -o tools/bin/stream_layouts2 && tools/bin/stream_layouts2 [n] [K]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct In { const double* f[12]; const int* id; const int* fl; };
struct OutSoa { double* f[12]; int* ray; int* surf; };
__device__ __forceinline__ void st(double* p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void sti(int* p, int v) { __builtin_nontemporal_store(v, p); }

__global__ __launch_bounds__(256) void k_soa(In in, long n, int K, OutSoa out) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        double v[12];
#pragma unroll
        for (int f = 0; f < 12; ++f) v[f] = __builtin_nontemporal_load(in.f[f] + i);
        const int id = in.id[i], fl = in.fl[i];
        for (int k = 0; k < K; ++k) {
            const long s = (long)k * n + i;
#pragma unroll
            for (int f = 0; f < 12; ++f) st(out.f[f] + s, v[f]);
            sti(out.ray + s, id); sti(out.surf + s, fl);
            v[0] += 1.0;
        }
    }
}
// MODE 0: tile = (k*n + i) / 64 on `stride` bytes; MODE 1: tile = (i / 64) * K + k
template <int MODE> __global__ __launch_bounds__(256) void k_tiled(In in, long n, int K, char* out, long tstride) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        double v[12];
#pragma unroll
        for (int f = 0; f < 12; ++f) v[f] = __builtin_nontemporal_load(in.f[f] + i);
        const int id = in.id[i], fl = in.fl[i];
        const int lane = threadIdx.x & 63;
        for (int k = 0; k < K; ++k) {
            const long tile = MODE == 0 ? (((long)k * n + i) >> 6) : ((i >> 6) * K + k);
            char* base = out + tile * tstride;
#pragma unroll
            for (int f = 0; f < 12; ++f) st((double*)(base + f * 512) + lane, v[f]);
            sti((int*)(base + 6144) + lane, id); sti((int*)(base + 6400) + lane, fl);
            v[0] += 1.0;
        }
    }
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 1000000;  // (a multiple of 64)
    const int K = argc > 2 ? atoi(argv[2]) : 5;
    const int SETS = n * 104 * (1 + K) > 2000000000L ? 1 : 4, ROUNDS = 5;
    std::vector<In> ins(SETS);
    std::vector<OutSoa> outs(SETS);
    std::vector<char*> tiles(SETS);
    for (int s = 0; s < SETS; ++s) {
        for (int f = 0; f < 12; ++f) { double* p; CHECK(hipMalloc(&p, n * 8)); CHECK(hipMemset(p, 0, n * 8)); ins[s].f[f] = p; }
        int* q; CHECK(hipMalloc(&q, n * 4)); CHECK(hipMemset(q, 0, n * 4)); ins[s].id = q;
        CHECK(hipMalloc(&q, n * 4)); CHECK(hipMemset(q, 0, n * 4)); ins[s].fl = q;
        for (int f = 0; f < 12; ++f) CHECK(hipMalloc(&outs[s].f[f], (size_t)K * n * 8));
        CHECK(hipMalloc(&outs[s].ray, (size_t)K * n * 4)); CHECK(hipMalloc(&outs[s].surf, (size_t)K * n * 4));
        CHECK(hipMalloc(&tiles[s], (size_t)K * (n / 64 + 1) * 8192 + 8192));
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int full = (int)((n + 255) / 256);
    struct V { const char* name; int kind; long tstride; int grid; };
    const V vs[] = {{"soa (14 arrays)", 0, 0, full}, {"tile6656", 1, 6656, full}, {"tile8192 (padded)", 1, 8192, full}, {"tile7168 (padded)", 1, 7168, full},
                    {"raytile 6656", 2, 6656, full}, {"raytile 8192", 2, 8192, full},
                    {"soa, 4096 persistent blocks", 0, 0, full < 4096 ? full : 4096}, {"raytile 6656, 4096 persistent blocks", 2, 6656, full < 4096 ? full : 4096}};
    const int NV = sizeof(vs) / sizeof(vs[0]);
    auto launch = [&](const V& v, int w) {
        if (v.kind == 0) hipLaunchKernelGGL(k_soa, dim3(v.grid), dim3(256), 0, 0, ins[w % SETS], n, K, outs[w % SETS]);
        else if (v.kind == 1) hipLaunchKernelGGL(k_tiled<0>, dim3(v.grid), dim3(256), 0, 0, ins[w % SETS], n, K, tiles[w % SETS], v.tstride);
        else hipLaunchKernelGGL(k_tiled<1>, dim3(v.grid), dim3(256), 0, 0, ins[w % SETS], n, K, tiles[w % SETS], v.tstride);
    };
    std::vector<std::vector<double>> us(NV);
    const double bytes = (double)n * 104 * (1 + K);
    const int reps = bytes > 1e9 ? 10 : 200, warm = bytes > 1e9 ? 5 : 300;
    for (int r = 0; r < ROUNDS; ++r)
        for (int vi = 0; vi < NV; ++vi) {
            for (int w = 0; w < (r == 0 ? warm : 20); ++w) launch(vs[vi], w);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int w = 0; w < reps; ++w) launch(vs[vi], w);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            us[vi].push_back(ms / reps * 1e3);
        }
    printf("n = %ld rays, K = %d, %.0f MB per launch, %d sets rotated, %d rounds of %d launches\n", n, K, bytes / 1e6, SETS, ROUNDS, reps);
    for (int vi = 0; vi < NV; ++vi) {
        std::vector<double> s = us[vi];
        std::sort(s.begin(), s.end());
        const double med = s[s.size() / 2];
        printf("%-40s median %8.2f us = %7.1f GB/s = %.3f of 8 TB/s   (min %.2f max %.2f)\n", vs[vi].name, med, bytes / med / 1e3, bytes / med / 1e3 / 8000, s.front(), s.back());
    }
    return 0;
}
