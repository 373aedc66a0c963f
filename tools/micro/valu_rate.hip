// Microbenchmark: how many wave64 vector-ALU instructions a SIMD issues per clock, by instruction kind and by the number of waves
// that share the SIMD.  The traversal kernels are priced against "one wave64 instruction per SIMD every 4 clocks"; this measures
// whether several resident waves get more than that out of a SIMD, and what the instructions the traversal is made of cost.
// Kinds: the float and integer instructions the interior step is made of, selects on VCC and on an SGPR pair, compares, and the
// compare + select pair of a compare-exchange (counted as one instruction per line of the pair: see the per-kind note printed).
// Every lane keeps 8 independent chains, 64 instructions per loop iteration; blocks of 256 threads = one wave per SIMD per block.
// Printed: wave-instructions per clock and SIMD (clock = the kernel's own s_memtime span over its wall time), and the chip-wide
// rate in T lane-operations per second (64 lanes per instruction; a v_pk instruction counted as 128).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_rate tools/micro/valu_rate.hip && tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(256) k(int iters, float seed, float* out, unsigned long long* clocks) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float b = seed * 0.5f + 1.0f, c = seed * 0.25f;
    const v2f pb = {b, b};
    uint32_t u0 = __float_as_uint(a0), u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const uint32_t ub = u0 | 1u, uc = 3u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#define F(i, TEXT) asm volatile(TEXT : "+v"(a##i) : "v"(b), "v"(c));
#define U(i, TEXT) asm volatile(TEXT : "+v"(u##i) : "v"(ub), "v"(uc));
#define ALLF(TEXT) F(0, TEXT) F(1, TEXT) F(2, TEXT) F(3, TEXT) F(4, TEXT) F(5, TEXT) F(6, TEXT) F(7, TEXT)
#define ALLU(TEXT) U(0, TEXT) U(1, TEXT) U(2, TEXT) U(3, TEXT) U(4, TEXT) U(5, TEXT) U(6, TEXT) U(7, TEXT)
            if (KIND == 0) { ALLF("v_add_f32 %0, %0, %1") }
            else if (KIND == 1) { ALLF("v_mul_f32 %0, %0, %1") }
            else if (KIND == 2) { ALLF("v_fma_f32 %0, %0, %1, %2") }
            else if (KIND == 3) { ALLF("v_max3_f32 %0, %0, %1, %2") }
            else if (KIND == 4) { ALLF("v_min_f32 %0, %0, %1") }
            else if (KIND == 5) {   // vcc set once, then eight selects on it
                asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a0), "v"(b) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a4) : "v"(b) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a5) : "v"(b) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a6) : "v"(b) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a7) : "v"(b) : "vcc");
            }
            else if (KIND == 6) {
#define P(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##i) : "v"(pb));
                P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7)
#undef P
            }
            else if (KIND == 7) { ALLU("v_and_or_b32 %0, %0, %1, %2") }
            else if (KIND == 8) { ALLF("v_sub_f32 %0, %0, %1") }
            else if (KIND == 9) { ALLU("v_min_u32 %0, %0, %1") }
            else if (KIND == 10) { ALLU("v_max_i32 %0, %0, %1") }
            else if (KIND == 11) { ALLU("v_add_u32 %0, %0, %1") }
            else if (KIND == 12) { ALLU("v_bfi_b32 %0, %0, %1, %2") }
            else if (KIND == 13) { ALLU("v_and_b32 %0, %0, %1") }
            else if (KIND == 14) { ALLU("v_lshl_add_u32 %0, %0, 1, %1") }
            else if (KIND == 15) {   // a compare into an SGPR pair and a select on it, alternating (the compare-exchange's shape)
#define CS(i) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(u##i) : "v"(ub), "v"(uc) : "s20", "s21");
                CS(0) CS(1) CS(2) CS(3) CS(4) CS(5) CS(6) CS(7)
#undef CS
            }
            else if (KIND == 16) {   // compares only (VOP3 form into an SGPR pair)
#define C(i) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(u##i), "v"(ub) : "s20", "s21");
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7)
#undef C
            }
            else if (KIND == 17) { ALLF("v_max_f32 %0, %0, %1") }
            else if (KIND == 18) { ALLU("v_mov_b32 %0, %1") }
            else if (KIND == 20) { ALLU("v_or_b32 %0, %0, %1") }
            else if (KIND == 21) { ALLU("v_lshlrev_b32 %0, 1, %0") }
            else if (KIND == 22) { ALLU("v_ashrrev_i32 %0, 1, %0") }
            else if (KIND == 23) { ALLU("v_sub_u32 %0, %0, %1") }
            else if (KIND == 24) { ALLU("v_cvt_f32_u32 %0, %0") }
            else if (KIND == 25) { ALLU("v_add3_u32 %0, %0, %1, %2") }
            else if (KIND == 26) { ALLU("v_xor_b32 %0, %0, %1") }
            else if (KIND == 27) { ALLU("v_not_b32 %0, %0") }
            else if (KIND == 28) { ALLF("v_fmac_f32 %0, %1, %2") }
            else if (KIND == 29) { ALLU("v_bfe_u32 %0, %0, 1, 30") }
            else if (KIND == 30) {   // selects on VCC (set before the loop; no clobber, so the compiler adds no padding)
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u##i) : "v"(ub));
                S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
            }
            else if (KIND == 31) {   // the same selects in the 64-bit encoding, VCC named as an ordinary scalar operand
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u##i) : "v"(ub));
                S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
            }
            else if (KIND == 32) {   // 64-bit encoding on an SGPR pair, explicitly
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(u##i) : "v"(ub));
                S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
            }
            else if (KIND == 33 || KIND == 34) {   // one select among seven additions: the 32-bit encoding (33) against the 64-bit one (34)
                F(0, "v_add_f32 %0, %0, %1") F(1, "v_add_f32 %0, %0, %1") F(2, "v_add_f32 %0, %0, %1") F(3, "v_add_f32 %0, %0, %1")
                F(4, "v_add_f32 %0, %0, %1") F(5, "v_add_f32 %0, %0, %1") F(6, "v_add_f32 %0, %0, %1")
                if (KIND == 33) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u7) : "v"(ub));
                else asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u7) : "v"(ub));
            }
            else if (KIND == 35) { ALLU("v_or_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") }
            else if (KIND == 36) { ALLU("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1") }
            else if (KIND == 37) { ALLU("v_lshrrev_b32 %0, 1, %0") }
            else if (KIND == 38) { ALLU("v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0") }
            else if (KIND == 19) {   // selects on a fixed SGPR pair
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, s[22:23]" : "+v"(u##i) : "v"(ub));
                S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + __uint_as_float(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clocks[0] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int cus, int wavesPerSimd, float* out, unsigned long long* clocks) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = cus * wavesPerSimd;   // 256 threads = 4 waves = one per SIMD
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, 64, 1.0f, out, clocks);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, out, clocks);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long span = 0;
    CHECK(hipMemcpy(&span, clocks, 8, hipMemcpyDeviceToHost));
    // s_memtime ticks at a fixed 100 MHz on this chip; the wall time and the nominal clock give clocks per instruction
    const double instrPerWave = (double)iters * 64.0;
    const double wallClocks = ms * 1e-3 * 2.4e9;   // at the nominal 2.4 GHz
    const double perSimdPerClock = instrPerWave * wavesPerSimd / wallClocks;
    const double lanes = (KIND == 6 ? 128.0 : 64.0);
    const double tops = instrPerWave * wavesPerSimd * 4.0 * cus * lanes / (ms * 1e-3) / 1e12;
    printf("%-14s waves/SIMD %d  %.3f ms  %.3f wave-instr/clk/SIMD at 2.4 GHz (1 per %.2f clk)  %.1f T lane-op/s\n", name, wavesPerSimd, ms, perSimdPerClock,
           1.0 / perSimdPerClock, tops);
    (void)span;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    float* out;
    unsigned long long* clocks;
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMalloc(&clocks, 64));
    for (int w : {1, 6}) {
        run<0>("v_add_f32", cus, w, out, clocks);
        run<8>("v_sub_f32", cus, w, out, clocks);
        run<1>("v_mul_f32", cus, w, out, clocks);
        run<2>("v_fma_f32", cus, w, out, clocks);
        run<6>("v_pk_mul_f32", cus, w, out, clocks);
        run<4>("v_min_f32", cus, w, out, clocks);
        run<17>("v_max_f32", cus, w, out, clocks);
        run<3>("v_max3_f32", cus, w, out, clocks);
        run<9>("v_min_u32", cus, w, out, clocks);
        run<10>("v_max_i32", cus, w, out, clocks);
        run<11>("v_add_u32", cus, w, out, clocks);
        run<13>("v_and_b32", cus, w, out, clocks);
        run<18>("v_mov_b32", cus, w, out, clocks);
        run<7>("v_and_or_b32", cus, w, out, clocks);
        run<12>("v_bfi_b32", cus, w, out, clocks);
        run<14>("v_lshl_add_u32", cus, w, out, clocks);
        run<5>("cmp+8 cndmask", cus, w, out, clocks);
        run<19>("cndmask sgpr", cus, w, out, clocks);
        run<30>("cndmask vcc", cus, w, out, clocks);
        run<33>("7 add + e32 sel", cus, w, out, clocks);
        run<34>("7 add + e64 sel", cus, w, out, clocks);
        run<31>("cndmask e64 vcc", cus, w, out, clocks);
        run<32>("cndmask e64 sgpr", cus, w, out, clocks);
        run<20>("v_or_b32", cus, w, out, clocks);
        run<26>("v_xor_b32", cus, w, out, clocks);
        run<27>("v_not_b32", cus, w, out, clocks);
        run<21>("v_lshlrev_b32", cus, w, out, clocks);
        run<22>("v_ashrrev_i32", cus, w, out, clocks);
        run<23>("v_sub_u32", cus, w, out, clocks);
        run<24>("v_cvt_f32_u32", cus, w, out, clocks);
        run<25>("v_add3_u32", cus, w, out, clocks);
        run<28>("v_fmac_f32", cus, w, out, clocks);
        run<29>("v_bfe_u32", cus, w, out, clocks);
        run<16>("v_cmp e64", cus, w, out, clocks);
        run<15>("cmp,cndmask", cus, w, out, clocks);
        run<37>("v_lshrrev_b32", cus, w, out, clocks);
        run<35>("v_or_b32 sdwa", cus, w, out, clocks);
        run<38>("v_and_b32 sdwa", cus, w, out, clocks);
        run<36>("v_cvt_f32_u32 sdwa", cus, w, out, clocks);
    }
    return 0;
}
