// Device-side building blocks of the wavefront path tracer (gfx950, wave64).
//
// Numerics contract: this file is compiled with -ffp-contract=off and without
// fast-math, and uses only IEEE +,-,*,/,sqrt,floor and comparisons, so every
// value is reproducible bit for bit on a CPU.  Division and sqrt are the
// correctly rounded forms (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define PT_WAVE 64
#ifndef PT_STACK_LDS
#define PT_STACK_LDS 10      // traversal stack entries (8 bytes each) kept in LDS per lane: the most that leaves room for 6 blocks per CU
#endif
#ifndef PT_STACK_SPILL
#define PT_STACK_SPILL 86    // further entries in per-lane scratch
#endif
#define PT_LEAFQ_WORDS 160   // 8-byte words of LDS per wave for the leaf phase: 64 result keys, 64 (u, v), 64 owner lanes
#define PT_STACK_TOTAL (PT_STACK_LDS + PT_STACK_SPILL)
#define PT_EPSILON 0.00001f  // extend.glsl:4 / shade.glsl:4
#define PT_PI 3.14159274101257324f
#define PT_INVPI 0.318309873342514038f
#define PT_INF __builtin_huge_valf()

typedef float v2f __attribute__((ext_vector_type(2)));
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
// System.Numerics.Vector3.Dot order: (x*x' + y*y') + z*z'   (Triangle.cs:21-40 call sites)
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ f3 normalize3(f3 a) { float l = sqrtf(dot3(a, a)); return mk3(a.x / l, a.y / l, a.z / l); }
__device__ __forceinline__ f3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }

struct f4 { float x, y, z, w; };
__device__ __forceinline__ f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
__device__ __forceinline__ f4 from4(float4 v) { return mk4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 to4(f4 v) { return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ f4 operator*(f4 a, f4 b) { return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ f4 operator*(f4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ f4 operator/(f4 a, float s) { return mk4(a.x / s, a.y / s, a.z / s, a.w / s); }
__device__ __forceinline__ f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ------------------------------------------------------------ scene on device
// pairs:    one 64-byte record per sibling pair = the host's Node[2k+1], Node[2k+2] (Node.cs:5-10),
//           64-byte aligned so one traversal step is one cache line.  The boxes are verbatim; the
//           leftFirst word of each node is replaced at upload by the node's traversal entry (see
//           ptrepack::leafEntry), so a step forms no entries.
// quads:    one 128-byte record per interior node at an even depth: the boxes of its (up to four)
//           grandchildren — a child that is a leaf stands for itself — as six float4 columns
//           (min x, min y, min z, max x, max y, max z), then the four traversal entries and 16
//           bytes of padding.  Slot order: the left child's expansion first.  An empty slot holds the
//           point box at the origin, which no ray passes: its near and far distances coincide on
//           every axis, so "entry distance < exit distance" cannot hold.
// big:      (first, count) of the leaves too large for a packed entry, indexed by that entry.
// tri_hot:  48 B per triangle: (V1.xyz, material bits) (V2-V1, 0) (V3-V1, 0) — all the hit test reads.
// tri_cold: 48 B per triangle: the three vertex normals, read once per shaded hit.
// mats:     the host's Material[] verbatim (48 B = 3 x float4).
// lights:   48 B per emissive triangle: (V1, running area sum) (V2, material bits) (V3, 0).
// qquads:   the quad records again in 64 bytes, for trees that do not fit the L2 caches: the same four entries, but the
//           boxes as 16-bit cell numbers of one grid over the scene (cell = extent / 65000 per axis), every plane moved
//           outward by at least one whole cell.  Three uint4 of planes — (lo_x, lo_y) (lo_z, hi_x) (hi_y, hi_z), each
//           plane two dwords = children (0 | 1 << 16), (2 | 3 << 16) — then the entries.  Such a box only decides where the
//           walk goes; whether a leaf's triangles are eligible is decided by the leaf's exact box, which travels with each
//           of its triangles in tri_gate (below).  Empty slots hold cell 0 on every plane: a point box, which no ray passes
//           (entry and exit distance coincide on every axis).
// tri_gate: 64 B per triangle: (V1.xyz, material bits) (V2-V1, box.min.x) (V3-V1, box.min.y) (box.min.z, box.max.xyz),
//           the box being that of the triangle's leaf, verbatim from the host's Node.
struct DevScene {
    const float4* pairs;
    const float4* quads;
    const uint2* big;
    const float4* tri_hot;
    const float4* tri_cold;
    const float4* mats;
    const float4* lights;
    const uint4* qquads;
    const float4* tri_gate;
    int32_t n_nodes, n_tris, n_lights;
    float light_total;
    float root_min[3], root_max[3];
    int32_t root_left, root_count;
    // the grid of qquads: plane = grid_min + cell number * grid_cell; rays that start farther than grid_reach from
    // grid_mid on some axis do not take the quantised walk (see rayOffGrid)
    float grid_min[3], grid_cell[3], grid_mid[3], grid_reach[3];
    // ---- the same scene as instances of meshes (optional; pt_build_scene_instanced): a tree over the instances' world boxes whose
    // leaves name instances, and one tree per mesh in the mesh's own space.  No reference counterpart: the reference flattens.
    // iquads:   quad records (128 B, exact format) of the top-level tree, then of every mesh's tree; boxes widened by the error bound
    //           of the ray's transformation (see rayOffInstances)
    // ibig:     big-leaf rows of those trees
    // itris:    48 B per mesh triangle in its tree's order: the three object-space positions; .w of the first = the triangle's
    //           position in the mesh as the host listed it
    // iinst:    128 B per instance in the top-level tree's order, eight float4: the instance matrix by columns with the translation
    //           (M11 M21 M31 M41) (M12 M22 M32 M42) (M13 M23 M33 M43); the inverse A of its linear part by columns with the translation
    //           again (A11 A21 A31 M41) (A12 A22 A32 M42) (A13 A23 A33 M43); (entry of the mesh tree's root, first flattened triangle of
    //           the instance, mesh box min x, min y); (mesh box min z, max x, max y, max z) — the box widened like the tree's
    // built_of: flattened triangle as the host listed it -> its index in the built flattened array (the id results carry)
    const float4* iquads;
    const uint2* ibig;
    const float4* itris;
    const float4* iinst;
    const uint32_t* built_of;
    uint32_t itop_entry;           // entry of the top-level tree's root
    float itop_min[3], itop_max[3];
    float inst_reach[3];           // rays starting farther than this from grid_mid take the flattened exact walk (< 0: all of them)
};

// Counters and queue sizes shared by the stages of one context (device memory).
struct Ctl {
    uint32_t count[2];        // live rays in the ping/pong path queues
    uint32_t cursor_extend;   // next unclaimed ray of the running extend launch
    uint32_t cursor_connect;
    unsigned long long paths, extend_rays, shadow_rays, extend_steps, extend_tris, shadow_steps, shadow_tris, stack_overflows;
    unsigned long long sched_interior, sched_leaf, sched_refill;   // wave iterations by kind, extend stage, instrumented builds
    unsigned long long special_rays;                               // rays that took the two-wide, NaN-rejecting walk
    unsigned long long unique_steps, unique_tris;                  // extend stage, instrumented builds: DISTINCT records fetched per wave instruction, summed
    unsigned long long shadow_occluded;                            // connect stage, instrumented builds: shadow rays that found an occluder
    // rays in the shadow queue, by the parity of the bounce that filled it (connect of one bounce may still be reading its count while
    // extend of the next bounce clears the other).  Far from count[]: every block of the shade stage adds to one of each per round,
    // and two atomics on one cache line take turns where two on distant lines proceed side by side.
    alignas(4096) uint32_t shadow_count[2];
};

// ------------------------------------------------ deterministic elementary fns
// Fixed polynomials (Cephes single-precision coefficients, three-constant
// Cody-Waite reduction); arguments here are in [0, 2*pi] for sin/cos.
__device__ __forceinline__ void det_sincos(float x, float* s, float* c) {
    float k = floorf(x * 0.636619772f + 0.5f);
    float r = x - k * 1.5703125f;
    r = r - k * 4.837512969970703125e-4f;
    r = r - k * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    int q = ((int)k) & 3;
    float ss = (q & 1) ? cp : sp;
    float cc = (q & 1) ? sp : cp;
    if (q == 1 || q == 2) cc = -cc;
    if (q >= 2) ss = -ss;
    *s = ss;
    *c = cc;
}

__device__ __forceinline__ float det_exp(float x) {
    if (x > 88.0f) return PT_INF;
    if (x < -87.0f) return 0.0f;
    float n = floorf(x * 1.44269504088896341f + 0.5f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float z = r * r;
    float p = ((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r + 1.6666665459e-1f) * r + 5.0000001201e-1f;
    float y = p * z + r + 1.0f;
    return y * __uint_as_float((uint32_t)((int)n + 127) << 23);
}

__device__ __forceinline__ float pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
__device__ __forceinline__ float clamp01(float x) { float a = (x < 0.0f) ? 0.0f : x; return (1.0f < a) ? 1.0f : a; }
__device__ __forceinline__ float max0(float x) { return (0.0f < x) ? x : 0.0f; }

// ------------------------------------------------------------------ random
// generate.glsl:10-21
__device__ __forceinline__ void xorShift(uint32_t& seed) {
    seed ^= seed << 13;
    seed ^= seed >> 17;
    seed ^= seed << 5;
}
__device__ __forceinline__ float randomFloat(uint32_t& seed) {
    xorShift(seed);
    return (float)seed * 2.3283064365386963e-10f;   // float(1.0 / 4294967295.0) is 2^-32
}
__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
#define PT_STREAM_GENERATE 2983742873u   // generate.glsl:174
#define PT_STREAM_SHADE 984524691u       // shade.glsl:352
#define PT_STREAM_NEE 0x68E31DA4u
// One stream per (path, frame, stage, bounce).  Mode 0 is the reference's formula with the
// bounce mixed in; mode 1 hashes so that frame 0 and thread 0 are not degenerate.
__device__ __forceinline__ uint32_t streamSeed(uint32_t gidx, uint32_t frames, uint32_t stream, uint32_t bounce, int mode) {
    uint32_t seed;
    if (mode == 0) {
        seed = gidx * 57028723u * frames * 87029659u + stream + bounce * 0x9E3779B9u;
    } else {
        seed = pcg(pcg(gidx + pcg(frames)) + stream + bounce * 0x9E3779B9u);
        if (seed == 0) seed = 1;
    }
    xorShift(seed); xorShift(seed); xorShift(seed);
    return seed;
}

// generate.glsl:160-168
__device__ __forceinline__ uint32_t compactBits(uint32_t x) {
    x &= 0x55555555;
    x = (x ^ (x >> 1)) & 0x33333333;
    x = (x ^ (x >> 2)) & 0x0f0f0f0f;
    x = (x ^ (x >> 4)) & 0x00ff00ff;
    x = (x ^ (x >> 8)) & 0x0000ffff;
    return x;
}

// Local slot -> pixel.  64 consecutive slots are one 8x8 Morton tile (generate.glsl:177-187);
// global tile g belongs to rank g % world as that rank's local tile g / world.  gidx is the
// thread index a single-GPU run gives the same pixel; it keys the random streams, which
// makes the image independent of the number of GPUs.
__device__ __forceinline__ bool pixelOfSlot(uint32_t slot, int width, int height, int rank, int world, uint32_t* pixel, uint32_t* gidx) {
    uint32_t gcx = (uint32_t)(width + 7) / 8, gcy = (uint32_t)(height + 7) / 8;
    uint32_t lid = slot & 63u;
    uint32_t group = (slot >> 6) * (uint32_t)world + (uint32_t)rank;
    if (group >= gcx * gcy) return false;
    uint32_t gx = group % gcx, gy = group / gcx;
    uint32_t sx = gx * 8 + compactBits(lid), sy = gy * 8 + compactBits(lid >> 1);
    if (sx >= (uint32_t)width || sy >= (uint32_t)height) return false;
    *pixel = sx + sy * (uint32_t)width;
    *gidx = group * 64u + lid;
    return true;
}

// --------------------------------------------------------------- wave helpers
__device__ __forceinline__ uint32_t laneId() { return __lane_id(); }

// Dense output slot for every lane with alive == true: ballot + prefix popcount, one
// atomic per wave.  Lanes with alive == false get an unspecified value.
__device__ __forceinline__ uint32_t waveAppend(bool alive, uint32_t* counter) {
    unsigned long long mask = __ballot(alive);
    if (mask == 0ull) return 0;
    uint32_t lane = laneId();
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ void waveAdd64(unsigned long long* counter, uint32_t v) {
    // sum over the active lanes, one atomic per wave
    unsigned long long total = v;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if (laneId() == 0) atomicAdd(counter, total);
}

// Inclusive prefix sum over the 64 lanes (all lanes must be active).
__device__ __forceinline__ uint32_t waveInclusiveScan(uint32_t v) {
    // within rows of 16 lanes: row_shr:n with out-of-row lanes reading 0
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    // across rows
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
    const uint32_t r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const uint32_t row = laneId() >> 4;
    return v + (row > 0u ? r0 : 0u) + (row > 1u ? r1 : 0u) + (row > 2u ? r2 : 0u);
}

// A slab distance can be NaN only for a ray with a zero/denormal/non-finite direction
// component or a non-finite origin; such rays take the NaN-rejecting traversal.
__device__ __forceinline__ bool raySpecial(f3 o, f3 d) {
    const bool dOk = fabsf(d.x) >= 1.0e-37f && fabsf(d.x) <= 3.0e38f && fabsf(d.y) >= 1.0e-37f && fabsf(d.y) <= 3.0e38f &&
                     fabsf(d.z) >= 1.0e-37f && fabsf(d.z) <= 3.0e38f;
    const bool oOk = fabsf(o.x) <= 3.0e38f && fabsf(o.y) <= 3.0e38f && fabsf(o.z) <= 3.0e38f;
    return !(dOk && oOk);
}

// The quantised walk (traverseWave<..., QUANT = true>) evaluates a grid plane's distance as fma(cell number, A, B) with
// A = cell * (1/d) and B = (grid_min - o) * (1/d), both rounded once per ray.  Against the reference's (plane - o) * (1/d)
// on the exact plane that differs by at most 6 * 2^-24 * R in units of position, R being the largest distance on that
// axis between the origin and the scene's box; the records keep every plane at least one cell = extent / 65000 away
// from anything inside, which covers that error while R <= 42 extents.  Rays that start farther out than 16 extents
// (grid_reach) or whose direction has a component below 1e-20 (A or B could overflow) take the exact two-wide walk.
__device__ __forceinline__ bool rayOffGrid(const DevScene& sc, f3 o, f3 d) {
    const bool dOk = fabsf(d.x) >= 1.0e-20f && fabsf(d.y) >= 1.0e-20f && fabsf(d.z) >= 1.0e-20f;
    const bool oOk = fabsf(o.x - sc.grid_mid[0]) <= sc.grid_reach[0] && fabsf(o.y - sc.grid_mid[1]) <= sc.grid_reach[1] &&
                     fabsf(o.z - sc.grid_mid[2]) <= sc.grid_reach[2];
    return !(dOk && oOk);
}

// Which records traverseWave walks.
#define PT_WALK_EXACT 0
#define PT_WALK_QUANT 1
#define PT_WALK_INST 2
// The instanced walk moves the ray into a mesh's space: o' = (o - T) A, d' = d A with A the inverse of the instance matrix's
// linear part, no renormalisation, so distances along the ray stay what they are in the world.  The rounding of that
// transformation and of the slab test on it displaces a box plane by at most ~10 * 2^-24 * (|o - T| / scale + 2) mesh units; the
// mesh trees' boxes are widened by 16 * 2^-24 * that bound taken at inst_reach (two scene extents around the scene's middle), and the
// top-level boxes by the same amount in world units.  Rays that start farther out, or can produce NaN slab distances, take the
// exact two-wide walk over the flattened scene.  Triangles are always tested in world space, on vertices re-derived from the
// mesh vertex and the instance matrix with the host's own expression (Vector3.Transform: ((x M11 + y M21) + z M31) + M41), which
// reproduces the flattened triangle bit for bit: a hit's distance and barycentrics are those of the flattened walk.
__device__ __forceinline__ bool rayOffInstances(const DevScene& sc, f3 o) {
    return !(fabsf(o.x - sc.grid_mid[0]) <= sc.inst_reach[0] && fabsf(o.y - sc.grid_mid[1]) <= sc.inst_reach[1] &&
             fabsf(o.z - sc.grid_mid[2]) <= sc.inst_reach[2]);
}
#define PT_SENTINEL 0x7FFFFFFFu   // stack entry below the entries of a mesh's tree: popping it takes the ray back to the world

// ------------------------------------------------------------------ traversal
struct HitRec { float t; int32_t tri; float u, v; };

// Box.cs:51-79.  The host's MathF.Min/Max propagate NaN, which makes a box test with a
// NaN slab distance a miss; v_min/v_max drop NaN, so the NANSAFE form rejects it
// explicitly.  A NaN can only come from 0 * inf, i.e. a ray with a zero (or denormal)
// direction component, so rays without one take the cheaper form with identical results.
template <bool NANSAFE>
__device__ __forceinline__ bool slabTest(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, f3 o, f3 inv, float* tmin) {
    float t1 = (mnx - o.x) * inv.x;
    float t2 = (mxx - o.x) * inv.x;
    float t3 = (mny - o.y) * inv.y;
    float t4 = (mxy - o.y) * inv.y;
    float t5 = (mnz - o.z) * inv.z;
    float t6 = (mxz - o.z) * inv.z;
    float tmn = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    float tmx = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    bool hit = tmx > 0.0f && tmn < tmx;
    if (NANSAFE) hit = hit && !((t1 != t1) | (t2 != t2) | (t3 != t3) | (t4 != t4) | (t5 != t5) | (t6 != t6));
    *tmin = tmn;
    return hit;
}

// The two distances of slabTest<false> without its verdict (the wide step folds the verdict into its order key).
__device__ __forceinline__ void slabRange(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, f3 o, f3 inv, float* tmin, float* tmax) {
    float t1 = (mnx - o.x) * inv.x;
    float t2 = (mxx - o.x) * inv.x;
    float t3 = (mny - o.y) * inv.y;
    float t4 = (mxy - o.y) * inv.y;
    float t5 = (mnz - o.z) * inv.z;
    float t6 = (mxz - o.z) * inv.z;
    *tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    *tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
}

// Triangle.cs:15-47 with edge1 = V2-V1 and edge2 = V3-V1 taken from the hot record (the
// same two subtractions, done once at upload).  Returns dist > 0 acceptance; the caller
// compares against its current best.
__device__ __forceinline__ bool triangleTest(float4 q0, float4 q1, float4 q2, f3 o, f3 d, float* dist, float* u_out, float* v_out) {
    f3 v1 = xyz(q0), edge1 = xyz(q1), edge2 = xyz(q2);
    f3 h = cross3(d, edge2);
    float a = dot3(edge1, h);
    float f = 1.0f / a;
    f3 s = o - v1;
    float u = f * dot3(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    f3 q = cross3(s, edge1);
    float v = f * dot3(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = f * dot3(edge2, q);
    *dist = t;
    *u_out = u;
    *v_out = v;
    return t > 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Per-ray traversal (the order every result depends on; the CPU oracle restates it step for step)
//
//   A ray holds one current entry `cur` and a stack of entries.  An entry is either an interior
//   reference (node index of a left child; the pair cur, cur+1 is one 64-byte record) or a leaf
//   (first triangle, count) packed into 32 bits.
//   Interior step: fetch the pair, slab-test both children, drop a child whose entry distance lies
//   beyond the best hit (closest: tmin > best, so a child entered exactly at the best distance is
//   still visited and an equal-distance triangle with a smaller index can win; any-hit:
//   tmin >= tlimit).  Of two surviving children the nearer becomes `cur`, the other is pushed;
//   with none, `cur` is popped (the ray ends on an empty stack).  Pushed entries carry their entry
//   distance; a popped entry whose distance has meanwhile fallen beyond the best hit is discarded
//   without being visited (closest hit only).
//   Leaf step: test its triangles (all of them for a closest-hit ray; up to the first occluder in
//   index order for an any-hit ray), then pop.
//   A packed leaf holds count-1 in 4 bits and the first triangle in 27; a leaf that does not fit
//   (more than 15 triangles, or first >= 2^27) is entered through an index into a side table of
//   (first, count) instead.
//   Closest-hit ties go to the smaller triangle index.
//
// Wide traversal (every ray whose slab distances cannot be NaN)
//
//   The same walk over the tree with every other level skipped: an interior entry refers to a
//   quad record, a step slab-tests its (up to four) boxes, drops the culled ones as above, visits
//   the survivors nearest first and pushes the rest, farthest first.  The order key of a child is
//   its entry distance clamped at zero, with the two lowest mantissa bits replaced by the slot
//   number (keys are unique, any sorting network gives the one order); pushed entries carry the
//   key with those two bits cleared as their distance.
//   Skipping a level is exact: for a ray without NaN slab distances a box that passes the slab test
//   lies inside a parent that passes it too (subtraction and multiplication round monotonically),
//   and the parent's entry distance is not larger, so the two-wide walk reaches the same leaves.
//   A ray that can produce NaN (zero, denormal or non-finite components: raySpecial) does not
//   take this walk: its lane runs the two-wide, NaN-rejecting step over the pair records inside
//   the same loop (such rays are a few in a hundred million, so the divergence costs nothing).
//
// Per-wave scheduling (free to change: it cannot alter any ray's result)
//
//   Each iteration of the wave does ONE kind of work under a wave-uniform branch: an interior
//   step for every lane holding an interior entry, or a leaf phase: the triangles of the waiting
//   leaves (at most 64, in lane then index order) are dealt one to a lane, each lane tests its
//   triangle against the owner's ray (fetched with lane permutes), results return to the owners
//   through an LDS minimum on (distance, triangle index), and owners whose leaf is finished pop.  Lanes that reach a leaf wait until PT_LEAF_BATCH lanes are waiting (or nobody is left
//   stepping), so both code paths run with many lanes active instead of a few.
//   Lanes are persistent: a lane whose ray has ended stores the result and takes the next ray of
//   the wave's private range; the range is replenished from the shared cursor (first range static,
//   no atomic) 256 rays at a time in large launches and 64 in small ones, so long rays do not hold
//   63 finished lanes hostage and the single cursor word is not the bottleneck.
//   The stack keeps PT_STACK_LDS entries (entry, distance: 8 bytes) per lane in LDS
//   (lstack[e * 64 + lane]: conflict free), deeper entries in scratch.
#define PT_LEAF_FLAG 0x80000000u
#define PT_LEAF_MAX_COUNT 15
#define PT_LEAF_MAX_FIRST (1u << 27)
#ifndef PT_REFILL_MIN
#define PT_REFILL_MIN 16
#endif
#ifndef PT_LEAF_BATCH
#define PT_LEAF_BATCH 12
#endif
#ifndef PT_ENTER_BATCH
#define PT_ENTER_BATCH 8     // INST: lanes waiting to enter an instance that make the wave run an enter phase
#endif
#define PT_NO_RAY 0xFFFFFFFFu
// The waves of a block share a pool of parked rays in LDS.  A wave whose lanes have thinned out hands its remaining rays
// to the pool and starts again with 64 fresh ones (which walk the top of the tree together); a thinned-out wave that finds
// enough parked rays takes them instead and runs full.  A ray's state is four 16-byte words and PT_POOL_STACK stack entries;
// rays with a deeper stack, or in the middle of a leaf, stay where they are.  One wave at a time works on the pool (a lock it
// tries once and never waits for), nobody ever waits for a ray to arrive, and a wave parks rays only while it has fresh ones
// to go on with, so the last wave alive finds the pool's remainder when it runs out of work.
#ifndef PT_POOL_SLOTS
#define PT_POOL_SLOTS 64
#endif
#ifndef PT_POOL_STACK
#define PT_POOL_STACK 8
#endif
#ifndef PT_POOL_MIN
#define PT_POOL_MIN 20       // idle lanes at which a wave turns to the pool
#endif
#define PT_POOL_WORDS (4 + 4 * 4 * PT_POOL_SLOTS + 2 * PT_POOL_STACK * PT_POOL_SLOTS)   // 32-bit words of LDS per block: lock, count, pad; state; stacks
#ifndef PT_GRAB_LARGE
#define PT_GRAB_LARGE 256u
#endif
#ifndef PT_GRAB_SWITCH
#define PT_GRAB_SWITCH 1024u   // rays per wave above which a launch counts as large
#endif

// Source: struct with
//   void load(uint32_t i, f3* o, f3* d, float* tlimit) const   — ray i of the queue
//   void store(uint32_t i, const HitRec& h, bool occluded) const
//
// Lane state: `cur` is the current entry: 0 = no work (ray ended or lane idle), sign bit clear = interior
// reference, sign bit set = leaf whose remaining triangles are [leafJ, leafEnd).
#define PT_KEY_MISS 0xFFFFFFFFu
// Markers for tools/isa_cost.py: with -DPT_ISA_MARKS the assembly carries comments around one interior iteration and one
// leaf phase, between which the tool counts vector-ALU instructions, and around the rarely taken part of the interior step
// (the stack spilling out of LDS), which the tool leaves out of the count.  The shipped build has none.
#ifdef PT_ISA_MARKS
#define PT_MARK(name) asm volatile("; PT_MARK " name)
#else
#define PT_MARK(name)
#endif

template <bool ANYHIT, bool STATS, int MODE, bool POOL, class Source>
__device__ __forceinline__ void traverseWave(const DevScene& sc, uint32_t n, uint32_t* cursor, uint2* lstack, unsigned long long* leafq, const Source& src,
                                             uint32_t* nsteps, uint32_t* ntris, uint32_t* overflow, uint32_t* sched, uint32_t refillMin,
                                             unsigned long long* specialRays, uint32_t* pool = nullptr) {
    static_assert(!(POOL && MODE == PT_WALK_INST), "the pool does not carry the instanced walk's second ray");
    static_assert(PT_POOL_STACK <= PT_STACK_LDS, "parked stacks are copied from the LDS part of a lane's stack");
    // sched[0..2] (STATS builds): wave iterations spent on interior steps, leaf steps, retire/refill; sched[3..4]: distinct quad
    // records / triangles fetched per wave instruction, summed (wave-uniform counts); sched[5]: any-hit rays that ended occluded
    constexpr bool QUANT = MODE == PT_WALK_QUANT, INST = MODE == PT_WALK_INST;
    const uint32_t lane = laneId();
    const uint32_t totalWaves = gridDim.x * (blockDim.x / PT_WAVE);
    const uint32_t waveId = blockIdx.x * (blockDim.x / PT_WAVE) + threadIdx.x / PT_WAVE;
    // Ranges are handed out in queue order (first one static, no atomic): rays cost very different amounts (sky against
    // geometry), and equal static shares per wave measured 30 % slower on primary rays.  Large launches take
    // PT_GRAB_LARGE rays per atomic to spare the cursor word, small ones 64 so that they end balanced.
    const uint32_t grab = (cursor != nullptr && n >= totalWaves * PT_GRAB_SWITCH) ? PT_GRAB_LARGE : 64u;
    const uint32_t dynBase = totalWaves * grab;
    uint32_t rangeLo = waveId * grab, rangeHi = rangeLo + grab;
    if (rangeLo > n) rangeLo = n;
    if (rangeHi > n) rangeHi = n;
    bool exhausted = (cursor == nullptr) || dynBase >= n;

    uint32_t rayIdx = PT_NO_RAY;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 0), inv = mk3(0, 0, 0);
    f3 qa = mk3(0, 0, 0), qb = mk3(0, 0, 0);   // QUANT: distance of grid plane number q on an axis = fma(q, qa, qb)
    // INST: origin and reciprocal direction in the space the ray is in (world, or a mesh's); the other walks have the world's only
    f3 ocStore = mk3(0, 0, 0), invcStore = mk3(0, 0, 0);
    f3& oc = INST ? ocStore : o;
    f3& invc = INST ? invcStore : inv;
    int32_t instSlot = -1;                     // INST: the instance the ray is inside, -1 at the top level
    float best = PT_INF, bu = 0.0f, bv = 0.0f, tlimit = PT_INF;
    int32_t bestTri = -1;
    bool occluded = false, special = false;
    uint32_t cur = 0;
    int32_t leafJ = 0, leafEnd = 0;
    int sp = 0;
    uint2 spill[PT_STACK_SPILL];
    uint2* const mystack = lstack + lane;

    // makes `e` current.  A leaf's range [leafJ, leafEnd) is unpacked when its first leaf phase begins (leafEnd == 0 until then: every
    // real range ends at 1 or later), so interior steps — most entries are interior — do not pay for it.
    auto enter = [&](uint32_t e) {
        cur = e;
        leafEnd = 0;
    };
    auto unpackLeaf = [&]() {
        const uint32_t e = cur;
        const uint32_t low = e & (PT_LEAF_MAX_FIRST - 1u), cm1 = (e >> 27) & 15u;
        leafJ = (int32_t)low;
        leafEnd = (int32_t)(low + cm1 + 1u);
        if ((e >> 27) == 31u) {   // leaf too big to pack: `low` is its row in the side table
            const uint2 rec = (INST && !special) ? sc.ibig[low] : sc.big[low];
            leafJ = (int32_t)rec.x;
            leafEnd = (int32_t)(rec.x + rec.y);
        }
    };
    // pops until an entry whose entry distance is not beyond the best hit turns up (closest hit only:
    // an any-hit limit never shrinks, so everything pushed stays eligible); 0 when the stack runs out
    auto popNext = [&]() {
        uint32_t next = 0;
        while (sp > 0) {
            sp--;
            const int slot = sp < PT_STACK_LDS ? sp : PT_STACK_LDS - 1;
            uint2 r = mystack[slot * PT_WAVE];
            // opaque to the optimiser: otherwise the LDS and scratch reads are merged into one flat access
            asm volatile("" : "+v"(r.x), "+v"(r.y));
            if (sp >= PT_STACK_LDS) r = spill[sp - PT_STACK_LDS];
            if (INST && r.x == PT_SENTINEL) { instSlot = -1; oc = o; invc = inv; continue; }   // the mesh's tree is done: back to the world
            if (ANYHIT || !(__uint_as_float(r.y) > best)) { next = r.x; break; }
        }
        enter(next);
    };
    auto push = [&](uint32_t e, float t) {
        const bool room = sp < PT_STACK_TOTAL;
        const uint2 r = make_uint2(e, __float_as_uint(t));
        if (sp < PT_STACK_LDS) mystack[sp * PT_WAVE] = r;
        else if (room) spill[sp - PT_STACK_LDS] = r;
        sp += room ? 1 : 0;
        if (!room) *overflow += 1u;   // the subtree is dropped; counted, never silent
    };
    // INST: a ray on a leaf of the top-level tree enters the leaf's first instance (the others go back on the stack as a shorter
    // leaf), with a marker below everything the mesh's tree will push.  An instance whose mesh box the transformed ray misses is left
    // again on the spot.  Like the triangle tests this runs as a phase of the wave, for all the lanes that wait for it.
    // whether the lane's current entry is a leaf of the top-level tree (INST): such lanes wait for an enter phase, not a leaf phase
    auto atInstance = [&]() { return INST && (int32_t)cur < 0 && instSlot < 0 && !special; };
    auto enterInstances = [&]() {
        if (atInstance()) {
            const uint32_t e = cur;
            uint32_t first = e & (PT_LEAF_MAX_FIRST - 1u), count = ((e >> 27) & 15u) + 1u;
            if ((e >> 27) == 31u) { const uint2 rec = sc.ibig[first]; first = rec.x; count = rec.y; }
            // the rest of the leaf as a packed entry (leaves of the top-level tree hold at most PT_LEAF_MAX_COUNT + 1 instances: a scene
            // whose top-level tree has a longer one is not installed as instances)
            if (count > 1u) push(PT_LEAF_FLAG | ((count - 2u) << 27) | (first + 1u), 0.0f);
            push(PT_SENTINEL, 0.0f);
            const float4* rec = sc.iinst + 8 * (size_t)first;
            const float4 a0 = rec[3], a1 = rec[4], a2 = rec[5], m6 = rec[6], m7 = rec[7];
            instSlot = (int32_t)first;
            const f3 rel = mk3(o.x - a0.w, o.y - a1.w, o.z - a2.w);
            oc = mk3((rel.x * a0.x + rel.y * a0.y) + rel.z * a0.z, (rel.x * a1.x + rel.y * a1.y) + rel.z * a1.z, (rel.x * a2.x + rel.y * a2.y) + rel.z * a2.z);
            const f3 dc = mk3((d.x * a0.x + d.y * a0.y) + d.z * a0.z, (d.x * a1.x + d.y * a1.y) + d.z * a1.z, (d.x * a2.x + d.y * a2.y) + d.z * a2.z);
            // the reciprocal need not be the correctly rounded one: the mesh trees' boxes are widened for more than its last bit
            invc = mk3(__builtin_amdgcn_rcpf(dc.x), __builtin_amdgcn_rcpf(dc.y), __builtin_amdgcn_rcpf(dc.z));
            float tm, tx;
            slabRange(m6.z, m6.w, m7.x, m7.y, m7.z, m7.w, oc, invc, &tm, &tx);
            const float tc = tm > 0.0f ? tm : 0.0f;
            // (a direction component that is exactly zero in the mesh's space, with the origin exactly on a box plane of that axis, makes
            // 0 * inf = NaN, which v_min / v_max drop: that axis then constrains nothing or everything — a set of rays of measure zero)
            const bool in = tc < tx && (ANYHIT ? tm < tlimit : !(tc > best));
            if (in) enter(__float_as_uint(m6.x));
            else popNext();
        }
    };

    // POOL: the block's parked rays (see PT_POOL_SLOTS): word 0 lock, word 1 count, then four 16-byte state words per slot
    // ([word][slot]) and PT_POOL_STACK stack entries per slot ([entry][slot])
    volatile uint32_t* const poolCtl = pool;
    uint4* const poolState = (uint4*)(pool + 4);
    uint2* const poolStack = (uint2*)(pool + 4 + 16 * PT_POOL_SLOTS);
    uint32_t breakAt = POOL ? (uint32_t)PT_POOL_MIN : refillMin;   // idle lanes at which the stepping loops hand over to the block below
    for (;;) {
        // ---- retire finished rays and hand the idle lanes the next rays: when nothing is in flight, or when
        //      refillMin lanes are idle and rays remain to be handed out; POOL: also when the wave has thinned out
        //      enough to trade rays with the block's pool
        {
            const uint32_t nActive0 = (uint32_t)__popcll(__ballot(cur != 0u));
            const bool moreRays = !(exhausted && rangeLo == rangeHi);
            uint32_t parked = 0;
            if (POOL) parked = poolCtl[1];
            bool service = nActive0 == 0u || (moreRays && 64u - nActive0 >= refillMin);
            if (POOL) service = service || (64u - nActive0 >= breakAt && (moreRays || parked != 0u));
            if (service) {
                if (STATS) sched[2]++;
                if (STATS && ANYHIT) sched[5] += (uint32_t)__popcll(__ballot(rayIdx != PT_NO_RAY && cur == 0u && occluded));
                if (rayIdx != PT_NO_RAY && cur == 0u) {
                    HitRec h;
                    h.t = best; h.tri = bestTri; h.u = bu; h.v = bv;
                    // the instanced walk names a triangle by its place in the host's list; results carry its index in the built array
                    if (INST && !special && bestTri >= 0) h.tri = (int32_t)sc.built_of[bestTri];
                    src.store(rayIdx, h, occluded);
                    rayIdx = PT_NO_RAY;
                }
                uint32_t nActive = nActive0;
                if (POOL && (parked != 0u || (moreRays && nActive0 != 0u))) {
                    uint32_t locked = 0;
                    if (lane == 0u) locked = atomicCAS((uint32_t*)pool, 0u, 1u) == 0u ? 1u : 0u;
                    locked = (uint32_t)__builtin_amdgcn_readfirstlane((int)locked);
                    if (locked != 0u) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const uint32_t count = poolCtl[1];
                        const uint32_t nIdle = 64u - nActive0;
                        if (count != 0u && (count >= nIdle || !moreRays)) {
                            // take: the idle lanes pick up the rays parked last
                            const uint32_t take = count < nIdle ? count : nIdle;
                            const unsigned long long idle = __ballot(rayIdx == PT_NO_RAY);
                            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                            if (rayIdx == PT_NO_RAY && rank < take) {
                                const uint32_t slot = count - 1u - rank;
                                const uint4 A = poolState[slot], B = poolState[PT_POOL_SLOTS + slot], C = poolState[2 * PT_POOL_SLOTS + slot],
                                            D = poolState[3 * PT_POOL_SLOTS + slot];
                                o = mk3(__uint_as_float(A.x), __uint_as_float(A.y), __uint_as_float(A.z));
                                rayIdx = A.w;
                                d = mk3(__uint_as_float(B.x), __uint_as_float(B.y), __uint_as_float(B.z));
                                enter(B.w);
                                inv = mk3(__uint_as_float(C.x), __uint_as_float(C.y), __uint_as_float(C.z));
                                sp = (int)(C.w & 0xffu);
                                special = (C.w & 0x100u) != 0u;
                                occluded = false;
                                if (ANYHIT) {
                                    tlimit = __uint_as_float(D.x);
                                    best = tlimit;
                                } else {
                                    tlimit = PT_INF;
                                    best = __uint_as_float(D.x); bu = __uint_as_float(D.y); bv = __uint_as_float(D.z); bestTri = (int32_t)D.w;
                                }
                                if (QUANT) {
                                    qa = mk3(sc.grid_cell[0] * inv.x, sc.grid_cell[1] * inv.y, sc.grid_cell[2] * inv.z);
                                    qb = mk3((sc.grid_min[0] - o.x) * inv.x, (sc.grid_min[1] - o.y) * inv.y, (sc.grid_min[2] - o.z) * inv.z);
                                }
#pragma unroll
                                for (int e = 0; e < PT_POOL_STACK; e++)
                                    if (e < sp) mystack[e * PT_WAVE] = poolStack[e * PT_POOL_SLOTS + slot];
                            }
                            if (lane == 0u) poolCtl[1] = count - take;
                            nActive += take;
                        } else if (moreRays && nActive0 != 0u) {
                            // park: every ray that can travel, if the pool has room for all of them
                            const bool parkable = cur != 0u && sp <= PT_POOL_STACK && !((int32_t)cur < 0 && leafEnd != 0);
                            const unsigned long long pm = __ballot(parkable);
                            const uint32_t np = (uint32_t)__popcll(pm);
                            if (np != 0u && count + np <= (uint32_t)PT_POOL_SLOTS) {
                                if (parkable) {
                                    const uint32_t slot = count + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
                                    poolState[slot] = make_uint4(__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), rayIdx);
                                    poolState[PT_POOL_SLOTS + slot] = make_uint4(__float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z), cur);
                                    poolState[2 * PT_POOL_SLOTS + slot] =
                                        make_uint4(__float_as_uint(inv.x), __float_as_uint(inv.y), __float_as_uint(inv.z), (uint32_t)sp | (special ? 0x100u : 0u));
                                    poolState[3 * PT_POOL_SLOTS + slot] = ANYHIT ? make_uint4(__float_as_uint(tlimit), 0u, 0u, 0u)
                                                                                 : make_uint4(__float_as_uint(best), __float_as_uint(bu), __float_as_uint(bv), (uint32_t)bestTri);
#pragma unroll
                                    for (int e = 0; e < PT_POOL_STACK; e++)
                                        if (e < sp) poolStack[e * PT_POOL_SLOTS + slot] = mystack[e * PT_WAVE];
                                    rayIdx = PT_NO_RAY;
                                    cur = 0u;
                                    sp = 0;
                                }
                                if (lane == 0u) poolCtl[1] = count + np;
                                nActive -= np;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0u) poolCtl[0] = 0u;
                    }
                }
                const bool fresh = nActive == 0u || (moreRays && 64u - nActive >= refillMin);
                if (fresh && rangeLo == rangeHi && !exhausted) {
                    uint32_t got = 0;
                    if (lane == 0) got = atomicAdd(cursor, grab);
                    got = __shfl(got, 0);
                    const uint32_t lo = dynBase + got;
                    if (lo >= n) exhausted = true;
                    else { rangeLo = lo; rangeHi = lo + grab < n ? lo + grab : n; }
                }
                const uint32_t avail = fresh ? rangeHi - rangeLo : 0u;
                if (fresh && avail == 0u && nActive == 0u) {   // nothing in flight and nothing left to hand out
                    if (!POOL) break;
                    if (poolCtl[1] == 0u) break;               // (rays that a wave parks from now on are that wave's to pick up)
                    __builtin_amdgcn_s_sleep(2);
                }
                const unsigned long long idleMask = __ballot(rayIdx == PT_NO_RAY);
                const uint32_t rank = (uint32_t)__popcll(idleMask & ((1ull << lane) - 1ull));
                if (rayIdx == PT_NO_RAY && rank < avail) {
                    rayIdx = rangeLo + rank;
                    src.load(rayIdx, &o, &d, &tlimit);
                    inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);   // Ray.cs:18
                    best = tlimit;
                    bestTri = -1;
                    bu = bv = 0.0f;
                    occluded = false;
                    sp = 0;
                    special = raySpecial(o, d);
                    if (QUANT) {
                        special = special || rayOffGrid(sc, o, d);
                        qa = mk3(sc.grid_cell[0] * inv.x, sc.grid_cell[1] * inv.y, sc.grid_cell[2] * inv.z);
                        qb = mk3((sc.grid_min[0] - o.x) * inv.x, (sc.grid_min[1] - o.y) * inv.y, (sc.grid_min[2] - o.z) * inv.z);
                    }
                    if (INST) {
                        special = special || rayOffInstances(sc, o);
                        oc = o;
                        invc = inv;
                        instSlot = -1;
                    }
                    float tr;
                    // the root test is done once per ray: always the NaN-propagating form
                    cur = 0;
                    if (INST && !special) {
                        if (sc.n_nodes > 0 && slabTest<true>(sc.itop_min[0], sc.itop_min[1], sc.itop_min[2], sc.itop_max[0], sc.itop_max[1], sc.itop_max[2], o, inv, &tr)) {
                            enter(sc.itop_entry);
                        }
                    } else {
                        const bool in = sc.n_nodes > 0 &&
                                        slabTest<true>(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], o, inv, &tr);
                        if (in) {
                            if (sc.root_count > 0) { cur = PT_LEAF_FLAG; leafJ = sc.root_left; leafEnd = sc.root_left + sc.root_count; }
                            else cur = special ? (uint32_t)sc.root_left : 1u;   // the node index of the root's left child (two-wide walk), or quad 0
                        }
                    }
                }
                {
                    const unsigned long long sm = __ballot(rayIdx != PT_NO_RAY && (idleMask >> lane & 1ull) && special);
                    if (sm != 0ull && lane == 0u) atomicAdd(specialRays, (unsigned long long)__popcll(sm));
                }
                const uint32_t nIdle = (uint32_t)__popcll(idleMask);
                rangeLo += nIdle < avail ? nIdle : avail;
                if (POOL) {
                    // the next visit: once enough lanes are idle to trade with the pool; if this visit left that many idle
                    // (nothing to take, no room to park), after eight more have finished
                    const uint32_t idleNow = 64u - (uint32_t)__popcll(__ballot(rayIdx != PT_NO_RAY));
                    breakAt = idleNow < (uint32_t)PT_POOL_MIN ? (uint32_t)PT_POOL_MIN : idleNow + 8u;
                }
            }
        }
        // (the pool's count is read afresh here and again at the top of the loop: a wave that leaves the stepping loops because the
        // pool had rays either trades with it there, which moves breakAt beyond its idle lanes, or comes back here and finds it empty)
        const bool refillable = !(exhausted && rangeLo == rangeHi) || (POOL && poolCtl[1] != 0u);
        // ---- interior steps, for as long as that is the wave's kind of work
        for (;;) {
            const unsigned long long interiorMask = __ballot((int32_t)cur > 0);
            const uint32_t nEnter = INST ? (uint32_t)__popcll(__ballot(atInstance())) : 0u;
            const uint32_t nLeaf = (uint32_t)__popcll(__ballot((int32_t)cur < 0)) - nEnter;
            if (interiorMask == 0ull || nLeaf >= PT_LEAF_BATCH || nEnter >= PT_ENTER_BATCH) break;
            if (refillable && 64u - (uint32_t)__popcll(interiorMask) - nLeaf - nEnter >= breakAt) break;
            if (STATS) sched[0]++;
            PT_MARK("interior_begin");
            if (STATS) {
                // distinct quad records among the lanes that step now: what the vector memory pipeline pays for (lanes that
                // ask for the same 16 bytes are served together)
                const bool stepping = (int32_t)cur > 0 && !special;
                unsigned long long m = __ballot(stepping);
                while (m != 0ull) {
                    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)__ffsll((long long)m) - 1);
                    m &= ~__ballot(stepping && cur == c);
                    sched[3]++;
                }
            }
            {
                if ((int32_t)cur > 0 && !special) {
                    if (STATS) (*nsteps)++;
                    uint32_t k0, k1, k2, k3;
                    uint32_t e0, e1, e2, e3;
                    // Order key of a child from its entry and exit distances.  Box.cs:51-79 passes when tx > 0 and tm < tx; with
                    // t0 = max(tm, +0) that is t0 < tx (t0 >= 0 makes tx positive, tm <= t0; negative distances and -0 clamp to +0).
                    // The best distance is positive, so "not beyond the best hit" reads the same on t0 as on tm.
#define PT_QUAD_KEY(K, TM, TX, SLOT)                                                                          \
                    {                                                                                          \
                        const int32_t tb = __float_as_int(TM);                                                 \
                        const int32_t t0 = tb > 0 ? tb : 0;                                                    \
                        const float tc = __int_as_float(t0);                                                   \
                        const bool h = tc < (TX) && (ANYHIT ? (TM) < tlimit : !(tc > best));                   \
                        K = h ? (((uint32_t)t0 & ~3u) | SLOT) : PT_KEY_MISS;                                   \
                    }
                    if (QUANT) {
                        const uint4* p = sc.qquads + 4 * (size_t)(cur - 1u);
                        const uint4 w0 = p[0], w1 = p[1], w2 = p[2], we = p[3];
                        e0 = we.x; e1 = we.y; e2 = we.z; e3 = we.w;
                        // distances of the six grid planes of a child: fma(cell number, cell / d, (grid_min - o) / d), see rayOffGrid.
                        // The fma is monotone in the cell number, so the nearer of an axis' two planes is the low one for d > 0 and the
                        // high one for d < 0 (as for the exact records below): the words holding the near and the far cells are picked by
                        // the sign once per step, and the pairwise min / max per child fall away.
                        const bool ngx = qa.x < 0.0f, ngy = qa.y < 0.0f, ngz = qa.z < 0.0f;
                        const uint32_t nxa = ngx ? w1.z : w0.x, nxb = ngx ? w1.w : w0.y, fxa = ngx ? w0.x : w1.z, fxb = ngx ? w0.y : w1.w;
                        const uint32_t nya = ngy ? w2.x : w0.z, nyb = ngy ? w2.y : w0.w, fya = ngy ? w0.z : w2.x, fyb = ngy ? w0.w : w2.y;
                        const uint32_t nza = ngz ? w2.z : w1.x, nzb = ngz ? w2.w : w1.y, fza = ngz ? w1.x : w2.z, fzb = ngz ? w1.y : w2.w;
#define PT_QUANT_CHILD(K, NX, NY, NZ, FX, FY, FZ, SLOT)                                                       \
                    {                                                                                          \
                        const float tm = fmaxf(fmaxf(fmaf((float)(NX), qa.x, qb.x), fmaf((float)(NY), qa.y, qb.y)), fmaf((float)(NZ), qa.z, qb.z)); \
                        const float tx = fminf(fminf(fmaf((float)(FX), qa.x, qb.x), fmaf((float)(FY), qa.y, qb.y)), fmaf((float)(FZ), qa.z, qb.z)); \
                        PT_QUAD_KEY(K, tm, tx, SLOT)                                                           \
                    }
                        PT_QUANT_CHILD(k0, nxa & 0xffffu, nya & 0xffffu, nza & 0xffffu, fxa & 0xffffu, fya & 0xffffu, fza & 0xffffu, 0u)
                        PT_QUANT_CHILD(k1, nxa >> 16, nya >> 16, nza >> 16, fxa >> 16, fya >> 16, fza >> 16, 1u)
                        PT_QUANT_CHILD(k2, nxb & 0xffffu, nyb & 0xffffu, nzb & 0xffffu, fxb & 0xffffu, fyb & 0xffffu, fzb & 0xffffu, 2u)
                        PT_QUANT_CHILD(k3, nxb >> 16, nyb >> 16, nzb >> 16, fxb >> 16, fyb >> 16, fzb >> 16, 3u)
#undef PT_QUANT_CHILD
                    } else {
                        // Box.cs:51-79 takes min and max of each axis' two plane distances.  (plane - o) * (1/d) is monotone in the plane
                        // (subtraction and multiplication round monotonically, the ray cannot produce NaN here), so for d > 0 the smaller
                        // distance is the min plane's and for d < 0 the max plane's: each lane fetches the column that is its near plane
                        // and the one that is its far plane (columns 48 bytes apart) and the twelve pairwise min / max of a step fall away.
                        // The entry distance is then the largest near distance, the exit distance the smallest far one: the same values.
                        const char* const qbase = (const char*)(INST ? sc.iquads : sc.quads);
                        const uint32_t rec = (cur - 1u) << 7;   // the host installs exact records only while they stay below 4 GB
                        const f3& iv = INST ? invc : inv;
                        const f3& ov = INST ? oc : o;
                        // (all-ones masks from the sign bits of the reciprocals pick between the record's offset and that plus 48)
                        const int32_t mx = __float_as_int(iv.x) >> 31, my = __float_as_int(iv.y) >> 31, mz = __float_as_int(iv.z) >> 31;
                        const uint32_t rec48 = rec + 48u;
#define PT_PICK(M, A, B) (((uint32_t)(M) & (A)) | (~(uint32_t)(M) & (B)))   /* one v_bfi_b32 */
                        const float4 nx = *(const float4*)(qbase + (size_t)PT_PICK(mx, rec48, rec)), fx = *(const float4*)(qbase + (size_t)PT_PICK(mx, rec, rec48));
                        const float4 ny = *(const float4*)(qbase + (size_t)PT_PICK(my, rec48, rec) + 16), fy = *(const float4*)(qbase + (size_t)PT_PICK(my, rec, rec48) + 16);
                        const float4 nz = *(const float4*)(qbase + (size_t)PT_PICK(mz, rec48, rec) + 32), fz = *(const float4*)(qbase + (size_t)PT_PICK(mz, rec, rec48) + 32);
#undef PT_PICK
                        const float4 eb = *(const float4*)(qbase + (size_t)rec + 96);
                        e0 = __float_as_uint(eb.x); e1 = __float_as_uint(eb.y); e2 = __float_as_uint(eb.z); e3 = __float_as_uint(eb.w);
#define PT_QUAD_CHILD(K, C, SLOT)                                                                             \
                    {                                                                                          \
                        const float tm = fmaxf(fmaxf((nx.C - ov.x) * iv.x, (ny.C - ov.y) * iv.y), (nz.C - ov.z) * iv.z); \
                        const float tx = fminf(fminf((fx.C - ov.x) * iv.x, (fy.C - ov.y) * iv.y), (fz.C - ov.z) * iv.z); \
                        PT_QUAD_KEY(K, tm, tx, SLOT)                                                           \
                    }
                        PT_QUAD_CHILD(k0, x, 0u)
                        PT_QUAD_CHILD(k1, y, 1u)
                        PT_QUAD_CHILD(k2, z, 2u)
                        PT_QUAD_CHILD(k3, w, 3u)
#undef PT_QUAD_CHILD
                    }
#undef PT_QUAD_KEY
#define PT_CSWAP(KA, EA, KB, EB)                                                  \
                    {                                                              \
                        const bool sw = KB < KA;                                   \
                        const uint32_t ka = sw ? KB : KA, kb = sw ? KA : KB;       \
                        const uint32_t ea = sw ? EB : EA, eb2 = sw ? EA : EB;      \
                        KA = ka; KB = kb; EA = ea; EB = eb2;                       \
                    }
                    PT_CSWAP(k0, e0, k1, e1)
                    PT_CSWAP(k2, e2, k3, e3)
                    PT_CSWAP(k0, e0, k2, e2)
                    PT_CSWAP(k1, e1, k3, e3)
                    PT_CSWAP(k1, e1, k2, e2)
#undef PT_CSWAP
                    // survivors beyond the nearest: pushed farthest first (sorted: k3 present implies k2, k2 implies k1)
                    if (k1 != PT_KEY_MISS) {
                        const int np = 1 + (k2 != PT_KEY_MISS ? 1 : 0) + (k3 != PT_KEY_MISS ? 1 : 0);
                        if (sp + np <= PT_STACK_LDS) {
                            uint2* const at = mystack + sp * PT_WAVE;
                            if (np == 3) at[0] = make_uint2(e3, k3 & ~3u);
                            if (np >= 2) at[(np - 2) * PT_WAVE] = make_uint2(e2, k2 & ~3u);
                            at[(np - 1) * PT_WAVE] = make_uint2(e1, k1 & ~3u);
                            sp += np;
                        } else {   // the stack leaves LDS: a few steps in a thousand
                            PT_MARK("cold_begin");
                            if (np == 3) push(e3, __uint_as_float(k3 & ~3u));
                            if (np >= 2) push(e2, __uint_as_float(k2 & ~3u));
                            push(e1, __uint_as_float(k1 & ~3u));
                            PT_MARK("cold_end");
                        }
                    }
                    if (k0 != PT_KEY_MISS) enter(e0);
                    else popNext();
                }
            }
            PT_MARK("interior_end");
            // the rare lanes whose ray can produce NaN slab distances: the two-wide, NaN-rejecting step over the pair records
            if (__any((int32_t)cur > 0 && special)) {
                if ((int32_t)cur > 0 && special) {
                    if (STATS) (*nsteps)++;
                    const int32_t node = (int32_t)cur;
                    const float4* p = sc.pairs + 2 * (size_t)(node - 1);   // pair (node-1)/2, 4 float4 each
                    const float4 a = p[0], b = p[1], c = p[2], e = p[3];
                    float lt, rt;
                    bool lhit = slabTest<true>(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, &lt);
                    bool rhit = slabTest<true>(c.x, c.y, c.z, c.w, e.x, e.y, o, inv, &rt);
                    if (ANYHIT) { lhit = lhit && lt < tlimit; rhit = rhit && rt < tlimit; }
                    else { lhit = lhit && !(lt > best); rhit = rhit && !(rt > best); }
                    const uint32_t eL = __float_as_uint(b.z), eR = __float_as_uint(e.z);   // entries, formed at upload
                    // nearer surviving child next, the other pushed with its entry distance
                    const bool rightFirst = rhit && (!lhit || rt < lt);
                    const uint32_t nearE = rightFirst ? eR : eL, farE = rightFirst ? eL : eR;
                    const float farT = rightFirst ? lt : rt;
                    if (lhit && rhit) push(farE, farT);
                    if (lhit || rhit) enter(nearE);
                    else popNext();
                }
            }
        }
        // ---- INST: the lanes waiting on a leaf of the top-level tree enter their instances (when enough wait, or nobody is left stepping)
        if (INST) {
            const uint32_t nEnter = (uint32_t)__popcll(__ballot(atInstance()));
            const uint32_t nInterior = (uint32_t)__popcll(__ballot((int32_t)cur > 0));
            if (nEnter != 0u && (nInterior == 0u || nEnter >= PT_ENTER_BATCH)) enterInstances();
        }
        // ---- triangle tests, one per waiting lane and iteration
        for (;;) {
            const unsigned long long leafMask = __ballot((int32_t)cur < 0 && !atInstance());
            if (leafMask == 0ull) break;
            const uint32_t nLeaf = (uint32_t)__popcll(leafMask);
            const uint32_t nInterior = (uint32_t)__popcll(__ballot((int32_t)cur > 0));
            if (nInterior != 0u && nLeaf < PT_LEAF_BATCH) break;
            if (refillable && 64u - nInterior - nLeaf - (INST ? (uint32_t)__popcll(__ballot(atInstance())) : 0u) >= breakAt) break;
            if (STATS) sched[1]++;
            PT_MARK("leaf_begin");
            unsigned long long* const resKey = leafq;                       // [64] best (distance bits, triangle) per owner
            float2* const resUv = (float2*)(leafq + 64);                    // [64] its barycentrics
            uint32_t* const ownerOf = (uint32_t*)(leafq + 128);             // [64] owner lane of each dealt triangle
            const bool owner = (int32_t)cur < 0 && !atInstance();
            if (owner && leafEnd == 0) unpackLeaf();
            const uint32_t left = owner ? (uint32_t)(leafEnd - leafJ) : 0u;
            const uint32_t cnt = left < 64u ? left : 64u;
            const uint32_t incl = waveInclusiveScan(cnt);
            const uint32_t pre = incl - cnt;
            const uint32_t mine = pre >= 64u ? 0u : (cnt < 64u - pre ? cnt : 64u - pre);   // this owner's triangles in this phase
            const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t dealt = all < 64u ? all : 64u;
            if (mine != 0u) resKey[lane] = ~0ull;
            for (uint32_t k = 0; __any(k < mine); k++)
                if (k < mine) ownerOf[pre + k] = lane;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- one triangle per lane
            const bool busy = lane < dealt;
            const uint32_t own = busy ? ownerOf[lane] : lane;
            const uint32_t preO = (uint32_t)__shfl((int)pre, (int)own);
            const int32_t jO = __shfl(leafJ, (int)own);
            const f3 po = mk3(__shfl(o.x, (int)own), __shfl(o.y, (int)own), __shfl(o.z, (int)own));
            const f3 pd = mk3(__shfl(d.x, (int)own), __shfl(d.y, (int)own), __shfl(d.z, (int)own));
            const float limO = ANYHIT ? __shfl(tlimit, (int)own) : 0.0f;
            const uint32_t kth = lane - preO;
            const int32_t j = jO + (int32_t)kth;
            bool hit = false;
            float tt = 0.0f, tu = 0.0f, tv = 0.0f;
            if (STATS) {
                unsigned long long m = __ballot(busy);
                while (m != 0ull) {
                    const int32_t c = __builtin_amdgcn_readlane(j, (int)__ffsll((long long)m) - 1);
                    m &= ~__ballot(busy && j == c);
                    sched[4]++;
                }
            }
            uint32_t cand = (uint32_t)j;   // what identifies the triangle in the result key
            if (INST) {
                // the owner's ray is inside an instance: the triangle is the mesh's, placed by the instance matrix with the host's
                // expression (Vector3.Transform, row vectors: ((x M11 + y M21) + z M31) + M41), which gives the flattened triangle's
                // vertices bit for bit; the test itself is the world-space one.  Owners on the flattened exact walk test flattened triangles.
                const int32_t slotO = __shfl(instSlot, (int)own);
                const bool flatO = __shfl((int)special, (int)own) != 0;
                if (busy) {
                    float4 q0, q1, q2;
                    if (flatO) {
                        const float4* tp = sc.tri_hot + 3 * (size_t)j;
                        q0 = tp[0]; q1 = tp[1]; q2 = tp[2];
                    } else {
                        const float4* tp = sc.itris + 3 * (size_t)j;
                        const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
                        const float4* rec = sc.iinst + 8 * (size_t)slotO;
                        const float4 c0 = rec[0], c1 = rec[1], c2 = rec[2], m6 = rec[6];
#define PT_PLACE(P) mk3(((P.x * c0.x + P.y * c0.y) + P.z * c0.z) + c0.w, ((P.x * c1.x + P.y * c1.y) + P.z * c1.z) + c1.w, ((P.x * c2.x + P.y * c2.y) + P.z * c2.z) + c2.w)
                        const f3 w1 = PT_PLACE(p0), w2 = PT_PLACE(p1), w3 = PT_PLACE(p2);
#undef PT_PLACE
                        q0 = make_float4(w1.x, w1.y, w1.z, 0.0f);
                        q1 = make_float4(w2.x - w1.x, w2.y - w1.y, w2.z - w1.z, 0.0f);     // Triangle.cs:17-18, as the hot records hold them
                        q2 = make_float4(w3.x - w1.x, w3.y - w1.y, w3.z - w1.z, 0.0f);
                        cand = __float_as_uint(m6.y) + __float_as_uint(p0.w);                // the triangle's place in the flattened array as the host listed it
                    }
                    hit = triangleTest(q0, q1, q2, po, pd, &tt, &tu, &tv);
                    if (ANYHIT) hit = hit && tt < limO;
                }
            } else if (QUANT) {
                // the walk came here through widened boxes: the leaf's own box (Node.Bounds, verbatim) decides whether its
                // triangles are eligible, by the reference's slab test and the walk's distance rule.  Owners on the exact
                // two-wide walk have applied both already.
                const f3 pinv = mk3(__shfl(inv.x, (int)own), __shfl(inv.y, (int)own), __shfl(inv.z, (int)own));
                const float bestO = __shfl(best, (int)own);
                const bool exactO = __shfl((int)special, (int)own) != 0;
                if (busy) {
                    const float4* tp = sc.tri_gate + 4 * (size_t)j;
                    const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
                    float tm, tx;
                    slabRange(q1.w, q2.w, q3.x, q3.y, q3.z, q3.w, po, pinv, &tm, &tx);
                    const int32_t tb = __float_as_int(tm);
                    const float tc = __int_as_float(tb > 0 ? tb : 0);
                    const bool open = exactO || (tc < tx && (ANYHIT ? tm < limO : !(tc > bestO)));
                    hit = open && triangleTest(q0, q1, q2, po, pd, &tt, &tu, &tv);
                    if (ANYHIT) hit = hit && tt < limO;
                }
            } else if (busy) {
                const float4* tp = sc.tri_hot + 3 * (size_t)j;
                const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                hit = triangleTest(q0, q1, q2, po, pd, &tt, &tu, &tv);
                if (ANYHIT) hit = hit && tt < limO;
            }
            // closest: smallest (distance, triangle index); any-hit: first occluder in index order
            const unsigned long long key = ANYHIT ? (unsigned long long)kth : (((unsigned long long)__float_as_uint(tt) << 32) | cand);
            if (hit) atomicMin(&resKey[own], key);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!ANYHIT) {
                if (hit && resKey[own] == key) resUv[own] = make_float2(tu, tv);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // ---- owners take their results
            if (mine != 0u) {
                const unsigned long long r = resKey[lane];
                if (ANYHIT) {
                    if (r != ~0ull) occluded = true;
                    if (STATS) (*ntris) += (r != ~0ull) ? (uint32_t)r + 1u : mine;
                } else {
                    if (STATS) (*ntris) += mine;
                    if (r != ~0ull) {
                        const float rt = __uint_as_float((uint32_t)(r >> 32));
                        const int32_t rj = (int32_t)(uint32_t)r;
                        if (rt < best || (rt == best && rj < bestTri)) {
                            const float2 uv = resUv[lane];
                            best = rt; bestTri = rj; bu = uv.x; bv = uv.y;
                        }
                    }
                }
                leafJ += (int32_t)mine;
                if (ANYHIT && occluded) cur = 0;
                else if (leafJ == leafEnd) popNext();
            }
            __builtin_amdgcn_wave_barrier();
            PT_MARK("leaf_end");
        }
    }
}
