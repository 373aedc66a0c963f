// Microbenchmark: how many per-lane 16-byte vector loads a CU sustains, by address pattern.  Unlike tcp_coalesce.hip (one
// dependent chain per lane: a latency figure), every lane here keeps 8 independent loads in flight and 8 waves per SIMD are
// resident, so the vector memory pipeline (address unit + L1 tag lookup) is what saturates.
//   mode 0: every lane loads 16 bytes from a random 128-byte line of its own
//   mode 1: every lane loads the 7 pieces (112 bytes) of a random 128-byte record of its own with 7 loads — the traversal step's pattern
//   mode 2: the 4 lanes of a quad load 64 contiguous bytes of one random line (one load covers half a record of one lane of the quad)
//   mode 3: the wave loads 1 KB contiguous at a random offset
//   mode 4: all 64 lanes load the same 16 bytes (the coherent-ray case: one record for the whole wave)
//   mode 5: groups of 8 lanes share an address (8 distinct records per wave instruction)
//   mode 6: mode 1's pattern loaded straight into LDS (global_load_lds_dwordx4, 7 per record), then read back with ds_read_b128
// Printed: lane-loads per clock per CU (at the clock measured by the kernel's own s_memtime span) and bytes per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

template <int MODE>
__global__ void __launch_bounds__(256) k(const float4* table, uint32_t lineMask, int iters, float* out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    float acc = 0.0f;
    uint32_t h = pcg(gid);
    __shared__ float4 stage[4 * 7 * 64];
    for (int it = 0; it < iters; it++) {
        float4 v[8];
        if (MODE == 6) {
            float4* base = stage + (threadIdx.x >> 6) * 7 * 64;
            h = pcg(h);
            const float4* r = table + 8 * (size_t)(h & lineMask);
#pragma unroll
            for (int j = 0; j < 7; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(r + j), (__attribute__((address_space(3))) void*)(base + j * 64), 16, 0, 0);
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 7; j++) v[j] = base[j * 64 + lane];
            v[7] = v[0];
        } else
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) { h = pcg(h); v[j] = table[8 * (size_t)(h & lineMask) + (h >> 29)]; }
        } else if (MODE == 1) {
            h = pcg(h);
            const float4* r = table + 8 * (size_t)(h & lineMask);
#pragma unroll
            for (int j = 0; j < 7; j++) v[j] = r[j];
            v[7] = v[0];
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                h = pcg(h);
                const uint32_t hq = __shfl(h, (int)(lane & ~3u));     // the quad's line
                v[j] = table[8 * (size_t)(hq & lineMask) + 4 * ((hq >> 31) & 1u) + (lane & 3u)];
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                h = pcg(h);
                const uint32_t hw = __shfl(h, 0);
                v[j] = table[8 * (size_t)((hw & lineMask) & ~7u) + lane];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                h = pcg(h);
                const uint32_t hw = __shfl(h, MODE == 4 ? 0 : (int)(lane & ~7u));
                v[j] = table[8 * (size_t)(hw & lineMask) + (hw >> 29)];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j].x + v[j].w;
    }
    out[gid] = acc;
}

int main(int argc, char** argv) {
    const int logLines = argc > 1 ? atoi(argv[1]) : 15;        // 128-byte lines: 2^15 = 4 MB
    const uint32_t lines = 1u << logLines;
    const int iters = 256;
    float4* table; float* out;
    CHECK(hipMalloc(&table, (size_t)lines * 128 + 4096));
    std::vector<float> hbuf((size_t)lines * 32);
    for (size_t i = 0; i < hbuf.size(); i++) hbuf[i] = (float)(i % 5);
    CHECK(hipMemcpy(table, hbuf.data(), hbuf.size() * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8, threads = 256;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double clk = prop.clockRate * 1e3;   // Hz (nominal)
    for (int mode = 0; mode < 7; mode++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) k<0><<<blocks, threads>>>(table, lines - 1, iters, out);
            if (mode == 1) k<1><<<blocks, threads>>>(table, lines - 1, iters, out);
            if (mode == 2) k<2><<<blocks, threads>>>(table, lines - 1, iters, out);
            if (mode == 3) k<3><<<blocks, threads>>>(table, lines - 1, iters, out);
            if (mode == 4) k<4><<<blocks, threads>>>(table, lines - 1, iters, out);
            if (mode == 5) k<5><<<blocks, threads>>>(table, lines - 1, iters, out);
            if (mode == 6) k<6><<<blocks, threads>>>(table, lines - 1, iters, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double loadsPerLane = ((mode == 1 || mode == 6) ? 7.0 : 8.0) * iters;
        const double laneLoads = (double)blocks * threads * loadsPerLane;
        const double perClkCu = laneLoads / (best * 1e-3) / clk / cus;
        printf("lines 2^%d (%.1f MB) mode %d: %.3f ms, %.1f G lane-loads/s, %.2f lane-loads/clk/CU, %.1f B/clk/CU, %.2f TB/s (nominal clock %.2f GHz)\n", logLines,
               lines * 128.0 / 1e6, mode, best, laneLoads / best / 1e6, perClkCu, perClkCu * 16, laneLoads * 16 / best / 1e9, clk / 1e9);
    }
    return 0;
}
