// The wavefront stages as HIP kernels: generate -> { extend -> shade -> connect } x bounces -> resolve.
// Ray queues are arrays of 16-byte columns in HBM (origin+birth, direction+flags, throughput, hit),
// so every queue access is one coalesced dwordx4 per lane.
#pragma once
#include "pt_device.hpp"

#define PT_BLOCK 256

// Queue traffic is streamed once per stage: non-temporal accesses keep it from displacing the tree and the triangles in L2.
#ifdef PT_NO_NT
__device__ __forceinline__ float4 ntLoad(const float4* p) { return *p; }
__device__ __forceinline__ void ntStore(float4* p, float4 v) { *p = v; }
#else
typedef float pt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntLoad(const float4* p) {
    const pt_v4f v = __builtin_nontemporal_load((const pt_v4f*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void ntStore(float4* p, float4 v) {
    const pt_v4f w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, (pt_v4f*)p);
}
#endif
#ifndef PT_SHADE_BLOCK
#define PT_SHADE_BLOCK 768
#endif
#ifndef PT_TRAVERSE_WAVES
#define PT_TRAVERSE_WAVES 5   // minimum waves per SIMD the traversal kernels are compiled for (caps their VGPRs)
#endif
#ifndef PT_POOL
#define PT_POOL 0             // whether the waves of a traversal block trade thinned-out rays through LDS (see PT_POOL_SLOTS)
#endif

struct RenderParams {
    int32_t width, height;
    float sky[4];
    // FrameState.Camera (State.cs:9-18)
    float p1[3], p2[3], p3[3], pos[3], right[3], up[3];
    uint32_t frames0;          // FrameState.Frames of the batch's first sample
    uint32_t slots;            // local pixel slots (local groups * 64)
    uint32_t samples;          // samples in this batch
    int32_t rank, world;
    int32_t rng_mode, nee;
    int32_t plane_fresh;       // the batch's planes were not cleared: shade of bounce 0 stores every path's first value
    int32_t bounce, last_bounce;
    int32_t group_major;       // order of the camera rays in the queue (the birth index is sample-major either way)
};

struct Queues {
    float4* o[2];      // origin.xyz, birth index (bits)
    float4* d[2];      // direction.xyz, flags: bit0 = previous vertex was specular; else, with multiple importance sampling, the sampled direction's density (low bit cleared)
    float4* T[2];      // throughput rgba
    float4* hit;       // t, triangle (bits), u, v
    float4* so;        // shadow origin.xyz, tmax
    float4* sd;        // shadow direction.xyz, birth (bits)
    float4* sc;        // shadow contribution rgba
    float4* plane;     // per-path radiance of the batch, indexed by birth
};

// ------------------------------------------------------------------ generate
// generate.glsl:135-156, 170-200.  Thread k of the batch: sample k / slots, local slot k % slots.
// DENSE (image sides divisible by 8: every thread emits a ray) writes ray k to queue slot k and
// needs no counter; otherwise the rays are appended by wave-level compaction.
template <bool DENSE>
__global__ void __launch_bounds__(PT_BLOCK) k_generate(RenderParams rp, Queues q, Ctl* ctl) {
    const uint32_t total = rp.slots * rp.samples;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = k < total;
    uint32_t pixel = 0, gidx = 0;
    uint32_t sample = 0, slot = 0;
    if (valid) {
        if (rp.group_major) {   // the samples of one 8x8 pixel group follow each other: a wave's range of the queue stays inside one group
            const uint32_t per = 64u * rp.samples, group = k / per, within = k - group * per;
            sample = within >> 6;
            slot = group * 64u + (within & 63u);
        } else {
            sample = k / rp.slots;
            slot = k - sample * rp.slots;
        }
        valid = pixelOfSlot(slot, rp.width, rp.height, rp.rank, rp.world, &pixel, &gidx);
    }
    f3 origin = mk3(0, 0, 0), dir = mk3(0, 0, 0);
    if (valid) {
        uint32_t seed = streamSeed(gidx, rp.frames0 + sample, PT_STREAM_GENERATE, 0, rp.rng_mode);
        const uint32_t sx = pixel % (uint32_t)rp.width, sy = pixel / (uint32_t)rp.width;
        const float r0 = randomFloat(seed);
        const float r1 = randomFloat(seed);
        const float r2 = randomFloat(seed) - 0.5f;
        const float r3 = randomFloat(seed) - 0.5f;
        const float x = ((float)sx + r0) / (float)rp.width;
        const float y = ((float)((uint32_t)rp.height - sy) + r1) / (float)rp.height;
        const f3 P1 = mk3(rp.p1[0], rp.p1[1], rp.p1[2]), P2 = mk3(rp.p2[0], rp.p2[1], rp.p2[2]), P3 = mk3(rp.p3[0], rp.p3[1], rp.p3[2]);
        const f3 pos = mk3(rp.pos[0], rp.pos[1], rp.pos[2]), right = mk3(rp.right[0], rp.right[1], rp.right[2]), up = mk3(rp.up[0], rp.up[1], rp.up[2]);
        const f3 t = (P1 + x * (P2 - P1)) + y * (P3 - P1);
        const f3 p = pos + 0.05f * (r2 * right + r3 * up);
        dir = normalize3(t - p);
        origin = p;
    }
    const uint32_t out = DENSE ? k : waveAppend(valid, &ctl->count[0]);
    if (valid) {
        ntStore(&q.o[0][out], make_float4(origin.x, origin.y, origin.z, __uint_as_float(sample * rp.slots + slot)));   // birth: sample-major, the planes' index
        ntStore(&q.d[0][out], make_float4(dir.x, dir.y, dir.z, __uint_as_float(1u)));
        // the throughput of a fresh path is (1,1,1,1) (generate.glsl:199): shade knows that at bounce 0, so it is not stored
    }
}

// -------------------------------------------------------------------- extend
// extend.glsl:301-309: closest hit for every ray of the current queue.  A fixed grid of
// persistent waves (no host round trip to size the launch); see traverseWave for how rays are
// handed to lanes.
struct ExtendSource {
    const float4* o;
    const float4* d;
    float4* hit;
    __device__ __forceinline__ void load(uint32_t i, f3* ro, f3* rd, float* tlimit) const {
        const float4 a = ntLoad(&o[i]), b = ntLoad(&d[i]);
        *ro = xyz(a);
        *rd = xyz(b);
        *tlimit = PT_INF;
    }
    __device__ __forceinline__ void store(uint32_t i, const HitRec& h, bool) const {
        ntStore(&hit[i], make_float4(h.t, __int_as_float(h.tri), h.u, h.v));
    }
};

// the instanced walk carries a second ray (the one in the mesh's space) and is compiled for one wave per SIMD less
template <bool STATS, int MODE>
__global__ void __launch_bounds__(PT_BLOCK, MODE == PT_WALK_INST ? PT_TRAVERSE_WAVES : ((MODE == PT_WALK_QUANT || PT_POOL != 0) ? PT_TRAVERSE_WAVES + 1 : PT_TRAVERSE_WAVES)) k_extend(DevScene sc, Queues q, Ctl* ctl, int cur, uint32_t refillMin) {
    __shared__ uint2 lds_stack[(PT_BLOCK / PT_WAVE) * PT_STACK_LDS * PT_WAVE];
    __shared__ unsigned long long lds_leafq[(PT_BLOCK / PT_WAVE) * PT_LEAFQ_WORDS];
    uint2* lstack = lds_stack + (threadIdx.x / PT_WAVE) * (PT_STACK_LDS * PT_WAVE);
    unsigned long long* leafq = lds_leafq + (threadIdx.x / PT_WAVE) * PT_LEAFQ_WORDS;
    constexpr bool POOL = PT_POOL != 0 && MODE != PT_WALK_INST;
    __shared__ __attribute__((aligned(16))) uint32_t lds_pool[POOL ? PT_POOL_WORDS : 4];
    if (POOL) {
        if (threadIdx.x < 4) lds_pool[threadIdx.x] = 0u;
        __syncthreads();
    }
    const uint32_t n = ctl->count[cur];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // counters of the stages that follow this launch
        ctl->count[cur ^ 1] = 0;
        ctl->shadow_count[cur] = 0;
        atomicAdd(&ctl->extend_rays, (unsigned long long)n);
    }
    uint32_t steps = 0, tris = 0, overflow = 0;
    ExtendSource src;
    src.o = q.o[cur];
    src.d = q.d[cur];
    src.hit = q.hit;
    uint32_t sched[6] = {0, 0, 0, 0, 0, 0};
    traverseWave<false, STATS, MODE, POOL>(sc, n, &ctl->cursor_extend, lstack, leafq, src, &steps, &tris, &overflow, sched, refillMin, &ctl->special_rays, lds_pool);
    if (STATS) {
        waveAdd64(&ctl->extend_steps, steps);
        waveAdd64(&ctl->extend_tris, tris);
        if (laneId() == 0) {
            atomicAdd(&ctl->sched_interior, (unsigned long long)sched[0]);
            atomicAdd(&ctl->sched_leaf, (unsigned long long)sched[1]);
            atomicAdd(&ctl->sched_refill, (unsigned long long)sched[2]);
            atomicAdd(&ctl->unique_steps, (unsigned long long)sched[3]);
            atomicAdd(&ctl->unique_tris, (unsigned long long)sched[4]);
        }
    }
    if (__any(overflow != 0)) waveAdd64(&ctl->stack_overflows, overflow);
}

// ------------------------------------------------------------------- connect
// Shadow rays of next-event estimation (no reference counterpart).  Unoccluded rays add
// their contribution to the path's radiance.
struct ConnectSource {
    const float4* so;
    const float4* sd;
    const float4* sc;
    float4* plane;
    __device__ __forceinline__ void load(uint32_t i, f3* ro, f3* rd, float* tlimit) const {
        const float4 a = ntLoad(&so[i]), b = ntLoad(&sd[i]);
        *ro = xyz(a);
        *rd = xyz(b);
        *tlimit = a.w;
    }
    __device__ __forceinline__ void store(uint32_t i, const HitRec&, bool occluded) const {
        if (!occluded) {
            const uint32_t birth = __float_as_uint(sd[i].w);
            plane[birth] = to4(from4(plane[birth]) + from4(sc[i]));
        }
    }
};

template <bool STATS, int MODE>
__global__ void __launch_bounds__(PT_BLOCK, MODE == PT_WALK_INST ? PT_TRAVERSE_WAVES : (PT_POOL != 0 ? PT_TRAVERSE_WAVES + 1 : PT_TRAVERSE_WAVES)) k_connect(DevScene sc, Queues q, Ctl* ctl, int cur, uint32_t refillMin) {
    __shared__ uint2 lds_stack[(PT_BLOCK / PT_WAVE) * PT_STACK_LDS * PT_WAVE];
    __shared__ unsigned long long lds_leafq[(PT_BLOCK / PT_WAVE) * PT_LEAFQ_WORDS];
    uint2* lstack = lds_stack + (threadIdx.x / PT_WAVE) * (PT_STACK_LDS * PT_WAVE);
    unsigned long long* leafq = lds_leafq + (threadIdx.x / PT_WAVE) * PT_LEAFQ_WORDS;
    constexpr bool POOL = PT_POOL != 0 && MODE != PT_WALK_INST;
    __shared__ __attribute__((aligned(16))) uint32_t lds_pool[POOL ? PT_POOL_WORDS : 4];
    if (POOL) {
        if (threadIdx.x < 4) lds_pool[threadIdx.x] = 0u;
        __syncthreads();
    }
    const uint32_t n = ctl->shadow_count[cur];
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctl->shadow_rays, (unsigned long long)n);
    uint32_t steps = 0, tris = 0, overflow = 0;
    ConnectSource src;
    src.so = q.so;
    src.sd = q.sd;
    src.sc = q.sc;
    src.plane = q.plane;
    uint32_t sched[6] = {0, 0, 0, 0, 0, 0};
    traverseWave<true, STATS, MODE, POOL>(sc, n, &ctl->cursor_connect, lstack, leafq, src, &steps, &tris, &overflow, sched, refillMin, &ctl->special_rays, lds_pool);
    if (STATS) {
        waveAdd64(&ctl->shadow_steps, steps);
        waveAdd64(&ctl->shadow_tris, tris);
        if (laneId() == 0) atomicAdd(&ctl->shadow_occluded, (unsigned long long)sched[5]);
    }
    if (__any(overflow != 0)) waveAdd64(&ctl->stack_overflows, overflow);
}

// ---------------------------------------------------- connect + extend in one launch
// The shadow rays of bounce b and the path rays of bounce b + 1 in one grid: every wave first takes shadow rays until that queue is dry,
// then path rays.  The two stages only meet the shade launches around them (see k_connect), so they may share a launch, and the few
// long shadow rays that would end a connect launch of their own finish under the path rays of the waves that are through.
template <bool STATS, int MODE>
__global__ void __launch_bounds__(PT_BLOCK, MODE == PT_WALK_INST ? PT_TRAVERSE_WAVES : ((MODE == PT_WALK_QUANT || PT_POOL != 0) ? PT_TRAVERSE_WAVES + 1 : PT_TRAVERSE_WAVES))
k_connect_extend(DevScene sc, Queues q, Ctl* ctl, int cur, uint32_t refillMin, uint32_t refillShadow) {
    __shared__ uint2 lds_stack[(PT_BLOCK / PT_WAVE) * PT_STACK_LDS * PT_WAVE];
    __shared__ unsigned long long lds_leafq[(PT_BLOCK / PT_WAVE) * PT_LEAFQ_WORDS];
    uint2* lstack = lds_stack + (threadIdx.x / PT_WAVE) * (PT_STACK_LDS * PT_WAVE);
    unsigned long long* leafq = lds_leafq + (threadIdx.x / PT_WAVE) * PT_LEAFQ_WORDS;
    constexpr bool POOL = PT_POOL != 0 && MODE != PT_WALK_INST;
    __shared__ __attribute__((aligned(16))) uint32_t lds_pool[POOL ? PT_POOL_WORDS : 4];
    if (POOL) {
        if (threadIdx.x < 4) lds_pool[threadIdx.x] = 0u;
        __syncthreads();
    }
    const uint32_t n = ctl->count[cur], nShadow = ctl->shadow_count[cur ^ 1];   // the shadow rays are those of the bounce before
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->count[cur ^ 1] = 0;
        ctl->shadow_count[cur] = 0;
        atomicAdd(&ctl->extend_rays, (unsigned long long)n);
        atomicAdd(&ctl->shadow_rays, (unsigned long long)nShadow);
    }
    uint32_t overflow = 0;
    {
        uint32_t steps = 0, tris = 0;
        ConnectSource src;
        src.so = q.so;
        src.sd = q.sd;
        src.sc = q.sc;
        src.plane = q.plane;
        uint32_t sched[6] = {0, 0, 0, 0, 0, 0};
        traverseWave<true, STATS, MODE, POOL>(sc, nShadow, &ctl->cursor_connect, lstack, leafq, src, &steps, &tris, &overflow, sched, refillShadow, &ctl->special_rays, lds_pool);
        if (STATS) {
            waveAdd64(&ctl->shadow_steps, steps);
            waveAdd64(&ctl->shadow_tris, tris);
            if (laneId() == 0) atomicAdd(&ctl->shadow_occluded, (unsigned long long)sched[5]);
        }
    }
    {
        uint32_t steps = 0, tris = 0;
        ExtendSource src;
        src.o = q.o[cur];
        src.d = q.d[cur];
        src.hit = q.hit;
        uint32_t sched[6] = {0, 0, 0, 0, 0, 0};
        traverseWave<false, STATS, MODE, POOL>(sc, n, &ctl->cursor_extend, lstack, leafq, src, &steps, &tris, &overflow, sched, refillMin, &ctl->special_rays, lds_pool);
        if (STATS) {
            waveAdd64(&ctl->extend_steps, steps);
            waveAdd64(&ctl->extend_tris, tris);
            if (laneId() == 0) {
                atomicAdd(&ctl->sched_interior, (unsigned long long)sched[0]);
                atomicAdd(&ctl->sched_leaf, (unsigned long long)sched[1]);
                atomicAdd(&ctl->sched_refill, (unsigned long long)sched[2]);
                atomicAdd(&ctl->unique_steps, (unsigned long long)sched[3]);
                atomicAdd(&ctl->unique_tris, (unsigned long long)sched[4]);
            }
        }
    }
    if (__any(overflow != 0)) waveAdd64(&ctl->stack_overflows, overflow);
}

// --------------------------------------------------------------------- shade
// shade.glsl:150-162
__device__ __forceinline__ float schlick(f3 direction, f3 normal, float n1, float n2) {
    float r0 = (n1 - n2) / (n1 + n2);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5(1.0f - dot3(normal, neg3(direction)));
}
__device__ __forceinline__ f3 schlick3(f3 ks, f3 l, f3 n) {
    const float ldotn = clamp01(dot3(l, n));
    const float p = pow5(1.0f - ldotn);
    return mk3(ks.x + (1.0f - ks.x) * p, ks.y + (1.0f - ks.y) * p, ks.z + (1.0f - ks.z) * p);
}
// shade.glsl:164-194
__device__ __forceinline__ float smithGgxMasking(f3 v, f3 n, float a2) {
    const float ndotv = clamp01(dot3(n, v));
    const float c = sqrtf(a2 + (1.0f - a2) * ndotv * ndotv) + ndotv;
    if (c == 0.0f) return 0.0f;
    return 2.0f * ndotv / c;
}
__device__ __forceinline__ float smithGgxShadowMasking(f3 l, f3 v, f3 n, float a2) {
    const float ndotl = clamp01(dot3(n, l));
    const float ndotv = clamp01(dot3(n, v));
    const float a = ndotv * sqrtf(a2 + (1.0f - a2) * ndotl * ndotl);
    const float b = ndotl * sqrtf(a2 + (1.0f - a2) * ndotv * ndotv);
    const float denom = a + b;
    if (denom == 0.0f) return 0.0f;
    return 2.0f * ndotl * ndotv / denom;
}
__device__ __forceinline__ f3 reflect3(f3 i, f3 n) { float d2 = 2.0f * dot3(n, i); return i - d2 * n; }

// shade.glsl:210-225
__device__ __forceinline__ f3 lambertImportanceSample(f3 normal, uint32_t& seed) {
    const f3 axis = fabsf(normal.x) > 0.95f ? mk3(0, 1, 0) : mk3(1, 0, 0);
    const f3 u = normalize3(cross3(normal, axis));
    const f3 v = cross3(u, normal);
    const float r0 = randomFloat(seed);
    const float r1 = randomFloat(seed);
    const float r = sqrtf(r0);
    const float theta = 2.0f * PT_PI * r1;
    float st, ct;
    det_sincos(theta, &st, &ct);
    return ((r * ct) * u + (r * st) * v) + sqrtf(1.0f - r0) * normal;
}

// shade.glsl:238-257
__device__ __forceinline__ f3 ggxImportance(f3 view, float alpha, uint32_t& seed) {
    const float r1 = randomFloat(seed);
    const float r2 = randomFloat(seed);
    const f3 wo = normalize3(mk3(view.x * alpha, view.y * alpha, view.z));
    const f3 t1 = (wo.z < 0.999f) ? normalize3(cross3(wo, mk3(0, 0, 1))) : mk3(1, 0, 0);
    const f3 t2 = cross3(t1, wo);
    float a = 1.0f / (1.0f + wo.z);
    float r = sqrtf(r1);
    float phi = (r2 < a) ? (r2 / a) * PT_PI : PT_PI + (r2 - a) / (1.0f - a) * PT_PI;
    float sphi, cphi;
    det_sincos(phi, &sphi, &cphi);
    float p1 = r * cphi;
    float p2 = r * sphi * ((r2 < a) ? 1.0f : wo.z);
    f3 n = (p1 * t1 + p2 * t2) + sqrtf(max0(1.0f - p1 * p1 - p2 * p2)) * wo;
    return normalize3(mk3(alpha * n.x, alpha * n.y, max0(n.z)));
}

// shade.glsl:296-330
__device__ __forceinline__ f4 sampleGgx(f3 color, float roughness, f3 world_n, f3 world_v, f3* world_l, uint32_t& seed) {
    f3 W = fabsf(world_n.x) > 0.9f ? mk3(0, 1, 0) : mk3(1, 0, 0);
    f3 N = world_n;
    f3 T = normalize3(cross3(N, W));
    f3 B = cross3(N, T);
    f3 v = mk3(dot3(world_v, T), dot3(world_v, B), dot3(world_v, N));
    f3 m = ggxImportance(v, roughness, seed);
    f3 l = reflect3(neg3(v), m);
    if (dot3(l, m) < 0.0f) return mk4(0, 0, 0, 0);
    *world_l = (l.x * T + l.y * B) + l.z * N;
    float a2 = roughness * roughness;
    f3 f = schlick3(color, m, l);
    float g1 = smithGgxMasking(v, m, a2);
    float g2 = smithGgxShadowMasking(l, v, m, a2);
    if (g1 == 0.0f) return mk4(0, 0, 0, 0);
    float g = g2 / g1;
    return mk4(f.x * g, f.y * g, f.z * g, 0.0f * g);
}

// smallest i with r * total < running_sum[i] (area-proportional light choice)
__device__ __forceinline__ int pickLight(const DevScene& sc, float r) {
    const float target = r * sc.light_total;
    int lo = 0, hi = sc.n_lights - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (target < sc.lights[3 * mid].w) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Dense output slots in two queues at once for the threads of the block with a / b set: wave ballots, a prefix
// over the waves' counts in LDS, and ONE atomic per queue and block (a single counter word sustains only
// ~90 atomics/us, so per-wave atomics would bound the whole stage).  All threads of the block must call it
// together; `lds` holds 2 x (BLOCK / 64) counts and 2 bases, and callers alternate between two such areas
// from one call to the next, which is what makes two barriers enough.
template <int BLOCK>
__device__ __forceinline__ void blockAppend2(bool a, bool b, uint32_t* counterA, uint32_t* counterB, uint32_t* lds, uint32_t* slotA, uint32_t* slotB) {
    constexpr uint32_t W = BLOCK / PT_WAVE;
    const unsigned long long maskA = __ballot(a), maskB = __ballot(b);
    const uint32_t lane = laneId(), wave = threadIdx.x / PT_WAVE;
    if (lane == 0) { lds[wave] = (uint32_t)__popcll(maskA); lds[W + wave] = (uint32_t)__popcll(maskB); }
    __syncthreads();
    if (lane == 0 && wave < 2u) {   // one thread per queue, in different waves: the two atomics overlap
        uint32_t* cnt = lds + wave * W;
        uint32_t total = 0;
        for (uint32_t w = 0; w < W; w++) { const uint32_t c = cnt[w]; cnt[w] = total; total += c; }
        lds[2 * W + wave] = total ? atomicAdd(wave == 0u ? counterA : counterB, total) : 0u;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    *slotA = lds[2 * W] + lds[wave] + (uint32_t)__popcll(maskA & below);
    *slotB = lds[2 * W + 1] + lds[W + wave] + (uint32_t)__popcll(maskB & below);
}

// shade.glsl:349-459 plus next-event estimation.  Terminated paths add their radiance to
// plane[birth] (one live path per birth index, so a plain read-modify-write is race free);
// survivors are appended to the other queue by wave-level ballot/prefix compaction.
// 6 waves per SIMD (80 VGPRs, 12 bytes of scratch) = two resident 768-thread blocks per CU: 5 % faster than the 96 VGPRs
// the compiler takes when left alone, and the larger block needs fewer appends to the one queue counter.
// A second instantiation with 256-thread blocks is for frames that share the GPU with another frame's traversal: its blocks are one
// wave per SIMD and fit beside four resident traversal waves, so this stage (bound by memory) runs under the other frame's extend
// (bound by instruction issue) instead of waiting for its blocks to leave.  Alone it is slower than the large block (three times
// the appends: 2.2 against 1.3 ms on a 1080p x 16 spp frame).
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 6
#endif
#ifndef PT_SHADE_BLOCK_SMALL
#define PT_SHADE_BLOCK_SMALL 256
#endif
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, PT_SHADE_WAVES) k_shade(DevScene sc, RenderParams rp, Queues q, Ctl* ctl, int cur) {
    __shared__ uint32_t s_append[2][2 * (BLOCK / PT_WAVE) + 2];
    const uint32_t n = ctl->count[cur];
    const int nxt = cur ^ 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->cursor_extend = 0;
        ctl->cursor_connect = 0;
    }
    const uint32_t stride = gridDim.x * blockDim.x;
    // every lane of a wave runs the same number of iterations: the compaction below is wave-wide
    const uint32_t rounds = (n + stride - 1) / stride;
    for (uint32_t it = 0; it < rounds; it++) {
        const uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool active = i < n;
        bool survive = false, shadow = false;
        f3 newO = mk3(0, 0, 0), newD = mk3(0, 0, 0);
        f4 newT = mk4(0, 0, 0, 0);
        uint32_t newFlags = 0, birth = 0;
        f3 sO = mk3(0, 0, 0), sD = mk3(0, 0, 0);
        float sTmax = 0.0f;
        f4 sC = mk4(0, 0, 0, 0);
        if (active) {
            const float4 ro = ntLoad(&q.o[cur][i]), rd = ntLoad(&q.d[cur][i]);
            const f3 o = xyz(ro), d = xyz(rd);
            birth = __float_as_uint(ro.w);
            const bool lastSpecular = (__float_as_uint(rd.w) & 1u) != 0u;
            const float lastPdf = __uint_as_float(__float_as_uint(rd.w) & ~1u);   // density of the direction that led here (multiple importance sampling only)
            const f4 T = rp.bounce == 0 ? mk4(1.0f, 1.0f, 1.0f, 1.0f) : from4(ntLoad(&q.T[cur][i]));
            const float4 hr = ntLoad(&q.hit[i]);
            const int32_t tri = __float_as_int(hr.y);
            const uint32_t sample = birth / rp.slots;
            const uint32_t slot = birth - sample * rp.slots;
            const uint32_t gidx = ((slot >> 6) * (uint32_t)rp.world + (uint32_t)rp.rank) * 64u + (slot & 63u);
            const uint32_t frames = rp.frames0 + sample;
            uint32_t seed = streamSeed(gidx, frames, PT_STREAM_SHADE, (uint32_t)rp.bounce, rp.rng_mode);
            f4 add = mk4(0, 0, 0, 0);
            bool addIt = false;
            if (tri < 0) {
                add = T * mk4(rp.sky[0], rp.sky[1], rp.sky[2], rp.sky[3]);   // shade.glsl:358-362
                addIt = true;
            } else {
                const float4 h0 = sc.tri_hot[3 * (size_t)tri];
                const int32_t matId = __float_as_int(h0.w);
                const float4 m0 = sc.mats[3 * matId], m1 = sc.mats[3 * matId + 1], m2 = sc.mats[3 * matId + 2];
                const f4 color = from4(m0);
                const int32_t type = __float_as_int(m2.x);
                const float index = m2.y, roughness = m2.z;
                if (m1.x > 0.0f || m1.y > 0.0f || m1.z > 0.0f) {   // shade.glsl:366-370
                    if (!rp.nee || lastSpecular) { add = T * from4(m1); addIt = true; }
                    else if (rp.nee == 2) {
                        // reached by sampling the BRDF at a vertex that also sampled the lights: balance heuristic
                        // against the density the light sampling has for this very point
                        const f3 ng = cross3(xyz(sc.tri_hot[3 * (size_t)tri + 1]), xyz(sc.tri_hot[3 * (size_t)tri + 2]));
                        const float cosy = fabsf(dot3(d, ng)) / sqrtf(dot3(ng, ng));
                        const float dist2 = hr.x * hr.x;
                        if (cosy > 0.0f && dist2 > 0.0f) {
                            const float pl = dist2 / (cosy * sc.light_total);
                            add = (T * from4(m1)) * (lastPdf / (lastPdf + pl));
                            addIt = true;
                        }
                    }
                } else {
                    const float4 c0 = sc.tri_cold[3 * (size_t)tri], c1 = sc.tri_cold[3 * (size_t)tri + 1], c2 = sc.tri_cold[3 * (size_t)tri + 2];
                    const float hu = hr.z, hv = hr.w;
                    const float w = 1.0f - hu - hv;
                    f3 normal = normalize3((xyz(c0) * w + xyz(c1) * hu) + xyz(c2) * hv);
                    const float dist = hr.x;
                    f3 bounce = mk3(0, 0, 0);
                    f4 throughput = T;
                    const f3 hitP = o + d * dist;
                    bool dead = false;
                    if (type == 2) {   // dielectric, shade.glsl:375-405
                        float n1 = 1.0f, n2 = index;
                        if (dot3(d, normal) > 0.0f) {
                            n1 = index;
                            n2 = 1.0f;
                            normal = normal * -1.0f;
                            throughput.x = throughput.x * det_exp(-color.x * dist);
                            throughput.y = throughput.y * det_exp(-color.y * dist);
                            throughput.z = throughput.z * det_exp(-color.z * dist);
                        }
                        const float r = randomFloat(seed);
                        const float sch = schlick(d, normal, n1, n2);
                        if (r < sch) {
                            bounce = reflect3(d, normal);
                        } else {
                            const float eta = n1 / n2;
                            const float dn = dot3(normal, d);
                            const float kk = 1.0f - eta * eta * (1.0f - dn * dn);
                            if (kk < 0.0f) bounce = reflect3(d, normal);   // total internal reflection
                            else bounce = eta * d - (eta * dn + sqrtf(kk)) * normal;
                        }
                        throughput = throughput * color;
                    }
                    if (type == 1) {   // mirror, shade.glsl:407-411
                        bounce = reflect3(d, normal);
                        throughput = throughput * color;
                    }
                    if (type == 0) {   // diffuse, shade.glsl:413-426
                        bounce = lambertImportanceSample(normal, seed);
                        const f4 brdf = color * PT_INVPI;
                        const float pdf = clamp01(dot3(normal, bounce)) * PT_INVPI;
                        if (rp.nee && sc.n_lights > 0) {
                            uint32_t ns = streamSeed(gidx, frames, PT_STREAM_NEE, (uint32_t)rp.bounce, rp.rng_mode);
                            const float rl = randomFloat(ns);
                            const float ru = randomFloat(ns);
                            const float rv = randomFloat(ns);
                            const int li = pickLight(sc, rl);
                            const float4 l0 = sc.lights[3 * li], l1 = sc.lights[3 * li + 1], l2 = sc.lights[3 * li + 2];
                            const f3 lv1 = xyz(l0), lv2 = xyz(l1), lv3 = xyz(l2);
                            const float su = sqrtf(ru);
                            const float b0 = 1.0f - su;
                            const float b1 = rv * su;
                            const float b2 = 1.0f - b0 - b1;
                            const f3 yp = (lv1 * b0 + lv2 * b1) + lv3 * b2;
                            const f3 wv = yp - hitP;
                            const float dist2 = dot3(wv, wv);
                            const float dl = sqrtf(dist2);
                            const f3 wn = wv / dl;
                            const float cosx = dot3(wn, normal);
                            const f3 ng = cross3(lv2 - lv1, lv3 - lv1);
                            const float ngl = sqrtf(dot3(ng, ng));
                            const float cosy = fabsf(dot3(wn, ng)) / ngl;
                            if (cosx > 0.0f && cosy > 0.0f && dist2 > 0.0f) {
                                const float geom = cosx * cosy * sc.light_total / dist2;
                                const int32_t lm = __float_as_int(l1.w);
                                sC = (T * brdf) * from4(sc.mats[3 * lm + 1]) * geom;
                                if (rp.nee == 2) {   // balance heuristic against the BRDF sampling's density for this direction
                                    const float pl = dist2 / (cosy * sc.light_total);
                                    sC = sC * (pl / (pl + clamp01(cosx) * PT_INVPI));
                                }
                                sO = hitP + wn * PT_EPSILON;
                                sD = wn;
                                sTmax = dl * 0.999f;
                                shadow = true;
                            }
                        }
                        if (pdf == 0.0f) dead = true;   // shade.glsl:419-423
                        else throughput = throughput * brdf * dot3(bounce, normal) / pdf;
                    }
                    if (type == 3) {   // metal, shade.glsl:428-431
                        throughput = throughput * sampleGgx(mk3(color.x, color.y, color.z), roughness, normal, neg3(d), &bounce, seed);
                    }
                    if (!dead) {
                        float roulette = (throughput.x < throughput.y) ? throughput.y : throughput.x;
                        roulette = (roulette < throughput.z) ? throughput.z : roulette;
                        if (!(randomFloat(seed) > roulette)) {   // shade.glsl:433-441
                            newT = throughput * (1.0f / roulette);
                            newO = hitP + bounce * PT_EPSILON;
                            newD = bounce;
                            newFlags = (type != 0) ? 1u : 0u;
                            if (type == 0 && rp.nee == 2) newFlags = __float_as_uint(clamp01(dot3(normal, bounce)) * PT_INVPI) & ~1u;
                            survive = !rp.last_bounce;
                        }
                    }
                }
            }
            if (rp.plane_fresh && rp.bounce == 0) ntStore(&q.plane[birth], to4(add));   // every path passes here once: no clear, no read
            else if (addIt) q.plane[birth] = to4(from4(q.plane[birth]) + add);
        }
        uint32_t slotOut, sOut;
        blockAppend2<BLOCK>(survive, shadow, &ctl->count[nxt], &ctl->shadow_count[cur], s_append[it & 1u], &slotOut, &sOut);
        if (survive) {
            ntStore(&q.o[nxt][slotOut], make_float4(newO.x, newO.y, newO.z, __uint_as_float(birth)));
            ntStore(&q.d[nxt][slotOut], make_float4(newD.x, newD.y, newD.z, __uint_as_float(newFlags)));
            ntStore(&q.T[nxt][slotOut], to4(newT));
        }
        if (shadow) {
            ntStore(&q.so[sOut], make_float4(sO.x, sO.y, sO.z, sTmax));
            ntStore(&q.sd[sOut], make_float4(sD.x, sD.y, sD.z, __uint_as_float(birth)));
            ntStore(&q.sc[sOut], to4(sC));
        }
    }
}

// ------------------------------------------------------------------- resolve
// sum[slot] += plane[s][slot] for the batch's samples in order: the image does not depend
// on how many samples were traced concurrently.
__global__ void __launch_bounds__(PT_BLOCK) k_resolve(float4* sum, const float4* plane, uint32_t slots, uint32_t samples) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= slots) return;
    f4 acc = from4(sum[slot]);
    for (uint32_t s = 0; s < samples; s++) acc = acc + from4(plane[(size_t)s * slots + slot]);
    sum[slot] = to4(acc);
}

// De-interleaves `world` rank-local accumulators (slots_padded float4 each) into a row-major
// image and divides by the sample count (the read-out of shade.glsl:89-96's running mean).
__global__ void __launch_bounds__(PT_BLOCK) k_assemble(const float4* gathered, uint32_t slots_padded, int world, int width, int height, float samples, float4* image) {
    const uint32_t gcx = (uint32_t)(width + 7) / 8, gcy = (uint32_t)(height + 7) / 8;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= gcx * gcy * 64u) return;
    const uint32_t group = k >> 6, lid = k & 63u;
    const uint32_t sx = (group % gcx) * 8 + compactBits(lid), sy = (group / gcx) * 8 + compactBits(lid >> 1);
    if (sx >= (uint32_t)width || sy >= (uint32_t)height) return;
    const uint32_t rank = group % (uint32_t)world, local = group / (uint32_t)world;
    const f4 v = from4(gathered[(size_t)rank * slots_padded + local * 64u + lid]);
    image[sx + sy * (uint32_t)width] = to4(v / samples);
}

// ------------------------------------------------------- standalone queries
struct TraceSource {
    const float* o_d;
    const float* tmax;
    int32_t* tri;
    float* t;
    float* uv;
    uint8_t* occluded;
    __device__ __forceinline__ void load(uint32_t i, f3* ro, f3* rd, float* tlimit) const {
        *ro = mk3(o_d[6 * i], o_d[6 * i + 1], o_d[6 * i + 2]);
        *rd = mk3(o_d[6 * i + 3], o_d[6 * i + 4], o_d[6 * i + 5]);
        *tlimit = tmax ? tmax[i] : PT_INF;
    }
    __device__ __forceinline__ void store(uint32_t i, const HitRec& h, bool occ) const {
        if (occluded) occluded[i] = occ ? 1 : 0;
        if (tri) tri[i] = h.tri;
        if (t) t[i] = h.t;
        if (uv) { uv[2 * i] = h.u; uv[2 * i + 1] = h.v; }
    }
};

// One wave per 64 rays (grid sized by the caller), no shared cursor.
template <bool ANYHIT, int MODE>
__global__ void __launch_bounds__(PT_BLOCK) k_trace(DevScene sc, TraceSource src, int n, Ctl* ctl) {
    __shared__ uint2 lds_stack[(PT_BLOCK / PT_WAVE) * PT_STACK_LDS * PT_WAVE];
    __shared__ unsigned long long lds_leafq[(PT_BLOCK / PT_WAVE) * PT_LEAFQ_WORDS];
    uint2* lstack = lds_stack + (threadIdx.x / PT_WAVE) * (PT_STACK_LDS * PT_WAVE);
    unsigned long long* leafq = lds_leafq + (threadIdx.x / PT_WAVE) * PT_LEAFQ_WORDS;
    uint32_t steps = 0, tris = 0, overflow = 0;
    uint32_t sched[6] = {0, 0, 0, 0, 0, 0};
    // every wave traces its own 64 rays and never refills: the threshold is not used
    traverseWave<ANYHIT, false, MODE, false>(sc, (uint32_t)n, nullptr, lstack, leafq, src, &steps, &tris, &overflow, sched, 64u, &ctl->special_rays);
    if (__any(overflow != 0)) waveAdd64(&ctl->stack_overflows, overflow);
}

__global__ void k_set_ctl(Ctl* ctl, uint32_t count0, unsigned long long paths) {
    ctl->count[0] = count0;
    ctl->count[1] = 0;
    ctl->shadow_count[0] = ctl->shadow_count[1] = 0;
    ctl->cursor_extend = 0;
    ctl->cursor_connect = 0;
    ctl->paths += paths;
}
