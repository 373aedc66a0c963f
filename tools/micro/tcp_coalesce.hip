// Microbenchmark: how the vector L1 handles 16-byte loads of random 64-byte records.
//   mode 0: every lane reads the 4 pieces of its own record with 4 loads (the traversal's pattern)
//   mode 1: the 4 lanes of a quad read the 4 pieces of one record per load, 4 loads cover the quad's 4 records
//   mode 2: as mode 1, loaded straight into LDS and read back with 4 ds_read_b128
//   mode 3: as mode 0 through buffer loads with 32-bit offsets
//   mode 4: own record as 8 x 8-byte loads
//   mode 5: only the first half of the own record (2 x 16 bytes)
//   mode 6: own record straight into LDS (4 loads) and 4 ds_read_b128 back
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
__global__ void __launch_bounds__(256) k(const float4* table, uint32_t mask, int iters, float* out) {
    __shared__ float4 stage[4 * 4 * 65];   // 4 waves x 4 regions x 1040 bytes
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, p = lane & 3u, q = lane >> 2;
    float acc = 0.0f;
    uint32_t h = pcg(gid);
    for (int it = 0; it < iters; it++) {
        h = pcg(h + (uint32_t)__float_as_uint(acc) % 3u);   // depends on the loaded data, like a traversal step
        const uint32_t rec = h & mask;
        float4 a, b, c, d;
        if (MODE == 0) {
            const float4* r = table + 4 * (size_t)rec;
            a = r[0]; b = r[1]; c = r[2]; d = r[3];
        } else if (MODE == 1) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t rj = __shfl(rec, (int)(lane & ~3u) + j);
                v[j] = table[4 * (size_t)rj + p];
            }
            a = v[0]; b = v[1]; c = v[2]; d = v[3];   // (pieces of four different records: the sum below is all that matters here)
        } else if (MODE == 3) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)((mask + 1u) * 64u), 0x00020000);
            typedef float v4f __attribute__((ext_vector_type(4)));
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            v4u r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, rec * 64u, 0, 0);
            v4u r1 = __builtin_amdgcn_raw_buffer_load_b128(rs, rec * 64u + 16u, 0, 0);
            v4u r2 = __builtin_amdgcn_raw_buffer_load_b128(rs, rec * 64u + 32u, 0, 0);
            v4u r3 = __builtin_amdgcn_raw_buffer_load_b128(rs, rec * 64u + 48u, 0, 0);
            a.x = __uint_as_float(r0.x); b.y = __uint_as_float(r1.y); c.z = __uint_as_float(r2.z); d.w = __uint_as_float(r3.w);
        } else if (MODE == 4) {
            const float2* r = (const float2*)(table + 4 * (size_t)rec);
            float2 v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3], v4 = r[4], v5 = r[5], v6 = r[6], v7 = r[7];
            a.x = v0.x + v1.y; b.y = v2.x + v3.y; c.z = v4.x + v5.y; d.w = v6.x + v7.y;
        } else if (MODE == 5) {
            const float4* r = table + 4 * (size_t)rec;
            a = r[0]; b = r[1]; c = a; d = b;
        } else if (MODE == 6) {
            float4* base = stage + wave * 4 * 65;
#pragma unroll
            for (int j = 0; j < 4; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(table + 4 * (size_t)rec + j),
                                                 (__attribute__((address_space(3))) void*)(base + j * 65), 16, 0, 0);
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            a = base[0 * 65 + lane]; b = base[1 * 65 + lane]; c = base[2 * 65 + lane]; d = base[3 * 65 + lane];
        } else {
            float4* base = stage + wave * 4 * 65;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t rj = __shfl(rec, (int)(lane & ~3u) + j);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(table + 4 * (size_t)rj + p),
                                                 (__attribute__((address_space(3))) void*)(base + j * 65), 16, 0, 0);
            }
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float4* mine = base + p * 65 + q * 4;
            a = mine[0]; b = mine[1]; c = mine[2]; d = mine[3];
        }
        acc += a.x + b.y + c.z + d.w;
    }
    out[gid] = acc;
}

int main(int argc, char** argv) {
    const uint32_t nrec = 1u << (argc > 1 ? atoi(argv[1]) : 20);   // 64-byte records
    const int iters = 64;
    float4* table; float* out;
    CHECK(hipMalloc(&table, (size_t)nrec * 64));
    std::vector<float> h((size_t)nrec * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 7);
    CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int blocks = 256 * 8, threads = 256;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 7; mode++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) k<0><<<blocks, threads>>>(table, nrec - 1, iters, out);
            if (mode == 1) k<1><<<blocks, threads>>>(table, nrec - 1, iters, out);
            if (mode == 2) k<2><<<blocks, threads>>>(table, nrec - 1, iters, out);
            if (mode == 3) k<3><<<blocks, threads>>>(table, nrec - 1, iters, out);
            if (mode == 4) k<4><<<blocks, threads>>>(table, nrec - 1, iters, out);
            if (mode == 5) k<5><<<blocks, threads>>>(table, nrec - 1, iters, out);
            if (mode == 6) k<6><<<blocks, threads>>>(table, nrec - 1, iters, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        std::vector<float> o(16);
        CHECK(hipMemcpy(o.data(), out, 64, hipMemcpyDeviceToHost));
        const double recs = (double)blocks * threads * iters;
        printf("records 2^%d (%u MB) mode %d: %.3f ms, %.1f G records/s, %.2f TB/s  (out %g)\n", argc > 1 ? atoi(argv[1]) : 20, nrec >> 14, mode, best, recs / best / 1e6, recs * 64 / best / 1e9, o[5]);
    }
    return 0;
}
