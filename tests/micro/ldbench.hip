// Micro-benchmark of divergent vector-memory reads on gfx950: which unit bounds a wave that
// gathers one 64-byte record per lane?  Prints clocks per wave-level load instruction per CU.
//   hipcc --offload-arch=gfx950 -O3 tests/micro/ldbench.hip -o gpurun_out/ldbench && gpurun_out/ldbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f4 *GF4;
typedef const __attribute__((address_space(1))) f2 *GF2;
typedef const __attribute__((address_space(1))) float *GF1;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// mode 0: per-lane record, 4 x dwordx4       (what a plain BVH walk does)
// mode 1: quad-cooperative: 4 x dwordx4, lanes of a quad read the 4 pieces of one record
// mode 2: fully coalesced dwordx4 (lane i reads piece i)
// mode 3: per-lane record, 8 x dwordx2
// mode 4: per-lane record, 16 x dword
// mode 5: per-lane 16-byte piece only (1 x dwordx4, random)
// mode 6: per-lane 2 x dwordx4 (32-byte record)
template <int MODE>
__global__ __launch_bounds__(256) void k(const f4 *tab, uint32_t mask /* records - 1 */, int iters, float *out, unsigned long long *clk)
{
    const GF4 t = (GF4)tab;
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t j = threadIdx.x & 3u;
    f4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        const uint32_t r = lcg(s) & mask;
        if (MODE == 0) { acc += t[r * 4] + t[r * 4 + 1] + t[r * 4 + 2] + t[r * 4 + 3]; }
        else if (MODE == 1) {
            // the four records of the quad: derive from a quad-uniform seed
            uint32_t q = (s ^ (j * 0x9E3779B9u)) ; (void)q;
            uint32_t qs = __builtin_amdgcn_update_dpp(0, (int)r, 0x00, 0xF, 0xF, true);
            uint32_t r1 = __builtin_amdgcn_update_dpp(0, (int)r, 0x55, 0xF, 0xF, true);
            uint32_t r2 = __builtin_amdgcn_update_dpp(0, (int)r, 0xAA, 0xF, 0xF, true);
            uint32_t r3 = __builtin_amdgcn_update_dpp(0, (int)r, 0xFF, 0xF, 0xF, true);
            acc += t[qs * 4 + j] + t[r1 * 4 + j] + t[r2 * 4 + j] + t[r3 * 4 + j];
        }
        else if (MODE == 2) { const uint32_t b = (lcg(s) & mask & ~63u) * 4; const uint32_t l = threadIdx.x & 63u;
                              const uint32_t bb = __builtin_amdgcn_readfirstlane(b);
                              acc += t[bb + l] + t[bb + 64 + l] + t[bb + 128 + l] + t[bb + 192 + l]; }
        else if (MODE == 3) { const GF2 p = (GF2)(t + r * 4); f2 a = p[0] + p[1] + p[2] + p[3] + p[4] + p[5] + p[6] + p[7]; acc.x += a.x; acc.y += a.y; }
        else if (MODE == 4) { const GF1 p = (GF1)(t + r * 4); float a = 0;
#pragma unroll
                              for (int i = 0; i < 16; ++i) a += p[i];
                              acc.x += a; }
        else if (MODE == 5) { acc += t[r * 4]; }
        else if (MODE == 6) { acc += t[r * 4] + t[r * 4 + 1]; }
    }
    const unsigned long long t1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, const f4 *tab, uint32_t records, int blocks_per_cu, int loads_per_iter, float *out, unsigned long long *clk)
{
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<MODE><<<grid, 256>>>(tab, records - 1, 50, out, clk);
    CHECK(hipEventRecord(e0));
    k<MODE><<<grid, 256>>>(tab, records - 1, iters, out, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // wave-level load instructions per CU = blocks_per_cu * 4 waves * iters * loads_per_iter
    const double instr = (double)blocks_per_cu * 4 * iters * loads_per_iter;
    const double ghz = 2.4;
    printf("%-34s table %8u KB  waves/CU %2d  %8.3f ms  %7.1f clk/wave-load-instr/CU  %7.1f clk per 64 records  %7.1f GB/s/CU\n", name, records * 64 / 1024,
           blocks_per_cu * 4, ms, ms * 1e6 * ghz / instr, ms * 1e6 * ghz / ((double)blocks_per_cu * 4 * iters),
           (double)blocks_per_cu * 256 * iters * loads_per_iter * (MODE == 3 ? 8 : MODE == 4 ? 4 : 16) / (ms * 1e6));
}

int main()
{
    const uint32_t max_records = 1u << 22; // 256 MB
    f4 *tab; float *out; unsigned long long *clk;
    CHECK(hipMalloc(&tab, (size_t)max_records * 64));
    CHECK(hipMemset(tab, 0, (size_t)max_records * 64));
    CHECK(hipMalloc(&out, 256 * 16 * 256 * sizeof(float)));
    CHECK(hipMalloc(&clk, 256 * 16 * 8));
    for (uint32_t records : {128u, 256u, 16384u, 1u << 20}) { // 8 KB, 16 KB (L1), 1 MB (L2), 64 MB (beyond L2, MALL/HBM)
        for (int bpc : {4, 8}) {
            run<0>("per-lane 4 x dwordx4", tab, records, bpc, 4, out, clk);
            run<1>("quad-cooperative 4 x dwordx4", tab, records, bpc, 4, out, clk);
            run<2>("coalesced 4 x dwordx4", tab, records, bpc, 4, out, clk);
            run<3>("per-lane 8 x dwordx2", tab, records, bpc, 8, out, clk);
            run<4>("per-lane 16 x dword", tab, records, bpc, 16, out, clk);
            run<5>("per-lane 1 x dwordx4", tab, records, bpc, 1, out, clk);
            run<6>("per-lane 2 x dwordx4", tab, records, bpc, 2, out, clk);
        }
    }
    return 0;
}
