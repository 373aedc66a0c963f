// Microbenchmark: the rate at which a CU's vector memory pipeline returns 16-byte loads, swept over the things that could
// explain a factor of two between measurements: waves resident per CU, independent loads in flight per lane, the address
// pattern, the form of the address (64-bit per-lane pointer or scalar base + 32-bit per-lane offset), and the
// destination (registers or LDS).
//   pattern 0  "row":    the wave loads 1 KiB contiguous of a random 1,152-byte row (a wave-coalesced row gather)
//   pattern 1  "own":    every lane loads the 7 pieces of a random 128-byte record of its own (a traversal step)
//   pattern 2  "oct":    8 lanes share a random 128-byte record, 16 bytes each (8 records per wave instruction)
//   pattern 3  "same":   all lanes load the same 16 bytes of a random record
//   pattern 4  "row>lds": pattern 0 through global_load_lds_dwordx4 (destination LDS, no VGPR write-back)
// Printed per line: GB/s per CU, lane-loads per clock per CU at the measured shader clock, chip-wide TB/s.
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

typedef float v4f __attribute__((ext_vector_type(4)));

// U independent loads per lane and iteration.  SADDR: byte offsets are 32-bit and the base is wave-uniform, which lets the
// compiler emit global_load_dwordx4 v, v_off, s[base:base+1].
template <int PATTERN, int U, bool SADDR>
__global__ void __launch_bounds__(256) k(const char* __restrict__ table, uint32_t rows, uint32_t rowBytes, int iters, float* out, unsigned long long* clocks) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    extern __shared__ v4f stage[];
    float acc = 0.0f;
    uint32_t h = pcg(gid * 2654435761u + 12345u);
    const __attribute__((address_space(1))) char* vbase = (const __attribute__((address_space(1))) char*)table;
    if (!SADDR) asm volatile("" : "+v"(vbase));   // the base lives in vector registers: the loads take the 64-bit per-lane address form
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long r0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        v4f v[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            uint32_t off;
            if (PATTERN == 0 || PATTERN == 4) {
                h = pcg(h);
                const uint32_t hw = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
                off = (hw % rows) * rowBytes + lane * 16u;
            } else if (PATTERN == 1) {
                if (j % 7 == 0) h = pcg(h);
                off = (h % rows) * rowBytes + (uint32_t)(j % 7) * 16u;
            } else if (PATTERN == 2) {
                h = pcg(h);
                const uint32_t hq = (uint32_t)__shfl((int)h, (int)(lane & ~7u));
                off = (hq % rows) * rowBytes + (lane & 7u) * 16u;
            } else {
                h = pcg(h);
                const uint32_t hw = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
                off = (hw % rows) * rowBytes;
            }
            if (PATTERN == 4) {
                v4f* dst = stage + ((threadIdx.x >> 6) * U + j) * 64;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(table + off), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else if (SADDR) {
                v[j] = *(const v4f*)(table + off);
            } else {
                v[j] = *(const __attribute__((address_space(1))) v4f*)(vbase + (size_t)off);
            }
        }
        if (PATTERN == 4) {
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += stage[(threadIdx.x >> 6) * U * 64 + lane].x;
        } else {
#pragma unroll
            for (int j = 0; j < U; j++) acc += v[j].x + v[j].w;
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    out[gid] = acc;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

struct Cfg { int pattern, u, saddr, blocksPerCu; };

template <int PATTERN, int U, bool SADDR>
static float run(const char* table, uint32_t rows, uint32_t rowBytes, int iters, int blocks, float* out, unsigned long long* clocks, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e30f;
    const size_t lds = PATTERN == 4 ? (size_t)4 * U * 64 * 16 : 0;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<PATTERN, U, SADDR>), dim3(blocks), dim3(256), lds, 0, table, rows, rowBytes, iters, out, clocks);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int wallKhz = 0;
    CHECK(hipDeviceGetAttribute(&wallKhz, hipDeviceAttributeWallClockRate, 0));
    printf("device %s, %d CUs, nominal clock %.2f GHz, wall clock %d kHz\n", prop.name, cus, prop.clockRate / 1e6, wallKhz);
    const size_t tableSizes[] = {(size_t)2048 * 1152, (size_t)38 << 20, (size_t)151 << 20};
    const size_t maxBytes = tableSizes[2] + 4096;
    char* table;
    CHECK(hipMalloc(&table, maxBytes));
    CHECK(hipMemset(table, 1, maxBytes));
    float* out;
    unsigned long long* clocks;
    const int maxBlocks = cus * 8;
    CHECK(hipMalloc(&out, (size_t)maxBlocks * 256 * 4));
    CHECK(hipMalloc(&clocks, (size_t)maxBlocks * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<unsigned long long> hc((size_t)maxBlocks * 2);
    const char* names[] = {"row", "own", "oct", "same", "row>lds"};
    const int onlyTable = getenv("GR_TABLE") ? atoi(getenv("GR_TABLE")) : -1;
    for (int ts = 0; ts < 3; ts++) {
        if (onlyTable >= 0 && ts != onlyTable) continue;
        for (int pattern = 0; pattern < 5; pattern++) {
            const uint32_t rowBytes = (pattern == 0 || pattern == 4) ? 1152u : 128u;
            const uint32_t rows = (uint32_t)(tableSizes[ts] / rowBytes);
            for (int bpc : {1, 2, 4, 8}) {
                for (int u : {4, 8, 16}) {
                    if (pattern == 1 && u == 4) continue;
                    for (int saddr = 0; saddr < 2; saddr++) {
                        if (pattern == 4 && saddr) continue;
                        if (pattern >= 2 && pattern <= 3 && (saddr || u != 8)) continue;   // the side cases: one form each
                        const int blocks = cus * bpc;
                        const int scale = getenv("GR_ITERS_SCALE") ? atoi(getenv("GR_ITERS_SCALE")) : 1;   // longer runs: does the rate hold once the clock settles?
                        const int iters = scale * (4096 / (u * bpc) < 16 ? 16 : 4096 / (u * bpc));
                        float ms = 0;
#define RUN(P, UU, S) if (pattern == P && u == UU && saddr == S) ms = run<P, UU, (S != 0)>(table, rows, rowBytes, iters, blocks, out, clocks, e0, e1);
                        RUN(0, 4, 0) RUN(0, 4, 1) RUN(0, 8, 0) RUN(0, 8, 1) RUN(0, 16, 0) RUN(0, 16, 1)
                        RUN(1, 8, 0) RUN(1, 8, 1) RUN(1, 16, 0) RUN(1, 16, 1)
                        RUN(2, 8, 0) RUN(3, 8, 0)
                        RUN(4, 4, 0) RUN(4, 8, 0) RUN(4, 16, 0)
#undef RUN
                        if (ms == 0) continue;
                        CHECK(hipMemcpy(hc.data(), clocks, (size_t)blocks * 16, hipMemcpyDeviceToHost));
                        double cyc = 0, wall = 0;
                        for (int b = 0; b < blocks; b++) { cyc += (double)hc[2 * b]; wall += (double)hc[2 * b + 1]; }
                        const double ghz = cyc / wall * (wallKhz * 1e3) / 1e9;   // shader clock while the kernel ran
                        const double laneLoads = (double)blocks * 256 * u * iters;
                        const double gbs = laneLoads * 16 / (ms * 1e-3) / 1e9;
                        printf("table %6.1f MB  %-8s waves/CU %2d  in flight/lane %2d  %s  %7.3f ms  %6.1f GB/s/CU  %5.2f lane-loads/clk/CU  %6.2f TB/s  (clock %.2f GHz)\n",
                               tableSizes[ts] / 1e6, names[pattern], bpc * 4, u, saddr ? "saddr" : "vaddr", ms, gbs / cus, laneLoads / (ms * 1e-3) / (ghz * 1e9) / cus,
                               gbs / 1e3, ghz);
                        fflush(stdout);
                    }
                }
            }
        }
    }
    return 0;
}
