// Microbenchmark + layout check: two ways for the 64 lanes of a wave to fetch one 128-byte record each (7 x 16 bytes used).
//   own:  every lane loads the 7 pieces of its record itself (7 wave instructions that each touch 64 cache lines)
//   coop: eight lanes share a record: in instruction k (0..7) lane L loads one 16-byte piece of the record of lane
//         8k + (L >> 3) straight into LDS (global_load_lds_dwordx4: lane L's bytes land at base + 16 L), so every wave
//         instruction covers 8 whole lines; afterwards each lane reads its own 7 pieces back (ds_read_b128).  The piece a
//         lane loads is XOR-swizzled with bits of the record's lane number so that the read-back is free of bank conflicts.
// Checks that both give the same bytes, then times both for tables of several sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

template <int MODE, bool SWIZZLE>
__global__ void __launch_bounds__(256) k(const char* __restrict__ table, uint32_t recMask, int iters, float* out, uint32_t activePercent) {
    extern __shared__ v4f stage[];   // 512 float4 (8 KiB) per wave
    const uint32_t lane = threadIdx.x & 63u;
    v4f* const mine = stage + (threadIdx.x >> 6) * 512;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t h = pcg(gid * 2654435761u + 99u);
    const uint32_t pieceEven = SWIZZLE ? (((lane & 7u) ^ (lane >> 4)) << 4) : ((lane & 7u) << 4);
    const uint32_t pieceOdd = SWIZZLE ? (pieceEven ^ 64u) : pieceEven;
    const uint32_t sR = SWIZZLE ? (((lane >> 1) & 7u) << 4) : 0u;
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        h = pcg(h);
        const bool active = (h >> 8) % 100u < activePercent;
        const uint32_t rec = active ? (h & recMask) : 0xFFFFFFFFu;
        v4f p[7];
        if (MODE == 0) {
            if (active) {
                const v4f* r = (const v4f*)(table + ((size_t)rec << 7));
#pragma unroll
                for (int j = 0; j < 7; j++) p[j] = r[j];
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const uint32_t r = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane >> 3) + 8u * kk) << 2), (int)rec);
                if (r != 0xFFFFFFFFu) {
                    const uint32_t off = (r << 7) + ((kk & 1) ? pieceOdd : pieceEven);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(table + off),
                                                     (__attribute__((address_space(3))) void*)(mine + kk * 64), 16, 0, 0);
                }
            }
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (active) {
                const char* img = (const char*)mine + lane * 128u;
#pragma unroll
                for (int j = 0; j < 7; j++) p[j] = *(const v4f*)(img + (((uint32_t)j << 4) ^ sR));
            }
        }
        if (active) {
#pragma unroll
            for (int j = 0; j < 7; j++) acc += p[j].x * (float)(j + 1) + p[j].w;
        }
    }
    out[gid] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t maxBytes = (size_t)128 << 20;
    char* table;
    CHECK(hipMalloc(&table, maxBytes));
    {
        std::vector<float> hbuf(maxBytes / 4);
        uint32_t s = 12345u;
        for (size_t i = 0; i < hbuf.size(); i++) { s = s * 1664525u + 1013904223u; hbuf[i] = (float)(s >> 8) * (1.0f / 16777216.0f); }
        CHECK(hipMemcpy(table, hbuf.data(), maxBytes, hipMemcpyHostToDevice));
    }
    const int maxBlocks = cus * 8;
    float *outA, *outB;
    CHECK(hipMalloc(&outA, (size_t)maxBlocks * 256 * 4));
    CHECK(hipMalloc(&outB, (size_t)maxBlocks * 256 * 4));
    const size_t lds = 4 * 512 * 16;
    // ---- same bytes?
    {
        const int blocks = cus, iters = 8;
        hipLaunchKernelGGL((k<0, false>), dim3(blocks), dim3(256), lds, 0, table, (1u << 20) - 1u, iters, outA, 70u);
        for (int swz = 0; swz < 2; swz++) {
            if (swz) hipLaunchKernelGGL((k<1, true>), dim3(blocks), dim3(256), lds, 0, table, (1u << 20) - 1u, iters, outB, 70u);
            else hipLaunchKernelGGL((k<1, false>), dim3(blocks), dim3(256), lds, 0, table, (1u << 20) - 1u, iters, outB, 70u);
            CHECK(hipDeviceSynchronize());
            std::vector<float> a((size_t)blocks * 256), b((size_t)blocks * 256);
            CHECK(hipMemcpy(a.data(), outA, a.size() * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(b.data(), outB, b.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
            printf("layout check (%s): %zu of %zu lanes differ from the per-lane fetch\n", swz ? "swizzled" : "plain", bad, a.size());
        }
    }
    // ---- rates
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int logRecs : {14, 17, 20}) {   // 2 MB, 16 MB, 128 MB of records
        for (uint32_t pct : {100u, 60u}) {
            for (int bpc : {2, 3, 4, 6}) {
                for (int mode = 0; mode < 3; mode++) {
                    const int blocks = cus * bpc, iters = 512 / bpc;
                    float best = 1e30f;
                    for (int rep = 0; rep < 3; rep++) {
                        CHECK(hipEventRecord(e0));
                        if (mode == 0) hipLaunchKernelGGL((k<0, false>), dim3(blocks), dim3(256), lds, 0, table, (1u << logRecs) - 1u, iters, outA, pct);
                        if (mode == 1) hipLaunchKernelGGL((k<1, false>), dim3(blocks), dim3(256), lds, 0, table, (1u << logRecs) - 1u, iters, outA, pct);
                        if (mode == 2) hipLaunchKernelGGL((k<1, true>), dim3(blocks), dim3(256), lds, 0, table, (1u << logRecs) - 1u, iters, outA, pct);
                        CHECK(hipEventRecord(e1));
                        CHECK(hipEventSynchronize(e1));
                        float ms;
                        CHECK(hipEventElapsedTime(&ms, e0, e1));
                        if (rep && ms < best) best = ms;
                    }
                    const double recs = (double)blocks * 256 * iters * pct / 100.0;
                    printf("records %5.1f MB  active %3u%%  waves/CU %2d  %-12s %7.3f ms  %6.2f G records/s  %5.2f records/ns/CU-equivalent TB/s %5.2f\n", (double)(128u << logRecs) / 1e6, pct,
                           bpc * 4, mode == 0 ? "own" : mode == 1 ? "coop" : "coop+swizzle", best, recs / (best * 1e-3) / 1e9, recs / (best * 1e-3) / 1e9 / cus,
                           recs * 128 / (best * 1e-3) / 1e12);
                    fflush(stdout);
                }
            }
        }
    }
    return 0;
}
