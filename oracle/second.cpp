// CPU ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.hpp).
//
// The reference states several pieces of its arithmetic twice: once in the C# host (the contract of this build) and once in
// GLSL, in the live wavefront shaders and again in the unused megakernel Shaders/compute.glsl.  This file restates the GLSL
// statements literally, separately from the product-path restatements in accelerator.cpp / wavefront.cpp, so that tests can
// hold the two against each other (tests/test_oracle.py):
//   * the closest-hit walk of extend.glsl:230-299 (every pierced leaf collected, no ordering, no distance culling) with the
//     triangle test of extend.glsl:171-211 = compute.glsl:144-185 (origin V3, edges V1-V3 and V2-V3) — against the C# walk
//     with the C# test (origin V1, Triangle.cs:15-47);
//   * packNormal / unpackNormal, extend.glsl:23-27 and shade.glsl:29-36 (16+16-bit hit normal);
//   * compute.glsl's schlick (:290-296), diffuse_reflection (:298-313) and the diffuse throughput of Sample (:383-387) —
//     against shade.glsl's schlick, lambertImportanceSample and brdf * cos / pdf as wavefront.cpp restates them.
#include "oracle.hpp"
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace oracle {

static const float INF = std::numeric_limits<float>::infinity();
static const float PI = 3.14159274101257324f;

// extend.glsl:171-211 (identical text in compute.glsl:144-185).  u weighs v1, v weighs v2, w weighs v3.
static bool triangleGlsl(const Triangle& tri, const Ray& ray, float best, float* t_out, float* u_out, float* v_out) {
    const Vec3 edge1 = tri.V1.Position - tri.V3.Position;
    const Vec3 edge2 = tri.V2.Position - tri.V3.Position;
    const Vec3 h = Cross(ray.Direction, edge2);
    const float a = Dot(edge1, h);
    const float f = 1.0f / a;
    const Vec3 s = ray.Origin - tri.V3.Position;
    const float u = f * Dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    const Vec3 q = Cross(s, edge1);
    const float v = f * Dot(ray.Direction, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = f * Dot(edge2, q);
    if (t > 0.0f && t < best) { *t_out = t; *u_out = u; *v_out = v; return true; }
    return false;
}

// extend.glsl:213-228 (= compute.glsl:187-217): GLSL min/max, no NaN propagation
static bool boundsGlsl(const Box& b, const Ray& ray) {
    const float t1 = (b.Min.x - ray.Origin.x) * ray.Inverse.x, t2 = (b.Max.x - ray.Origin.x) * ray.Inverse.x;
    const float t3 = (b.Min.y - ray.Origin.y) * ray.Inverse.y, t4 = (b.Max.y - ray.Origin.y) * ray.Inverse.y;
    const float t5 = (b.Min.z - ray.Origin.z) * ray.Inverse.z, t6 = (b.Max.z - ray.Origin.z) * ray.Inverse.z;
    const float tmin = std::fmax(std::fmax(std::fmin(t1, t2), std::fmin(t3, t4)), std::fmin(t5, t6));
    const float tmax = std::fmin(std::fmin(std::fmax(t1, t2), std::fmax(t3, t4)), std::fmax(t5, t6));
    return tmax > 0.0f && tmin < tmax;
}

// extend.glsl:230-299.  The shader's fixed arrays (stack[32], candidates[128]) are unbounded here; *stack_peak and
// *candidate_count report what the shader would have needed.  A root that is a leaf (fewer than 3 triangles) is tested directly;
// the shader assumes an interior root.
GlslHit IntersectGlsl(const Accelerator& acc, const Ray& ray, int* stack_peak, int* candidate_count) {
    GlslHit hit;
    hit.Distance = std::numeric_limits<float>::max();   // FLT_MAX, extend.glsl:234
    hit.Tri = -1;
    hit.Normal = Vec3{0, 0, 0};
    const auto& nodes = acc.nodes;
    const auto& triangles = acc.triangles;
    if (stack_peak) *stack_peak = 0;
    if (candidate_count) *candidate_count = 0;
    if (nodes.empty()) return hit;
    std::vector<int> candidates;
    if (nodes[0].Count > 0) {
        candidates.push_back(0);
    } else {
        std::vector<int> stack;
        stack.push_back(0);
        size_t peak = 1;
        while (!stack.empty()) {
            const Node& node = nodes[stack.back()];
            const Node& left = nodes[node.LeftFirst];
            const Node& right = nodes[node.LeftFirst + 1];
            const bool lhit = boundsGlsl(left.Bounds, ray), rhit = boundsGlsl(right.Bounds, ray);
            const bool ltraverse = lhit && left.Count < 0, rtraverse = rhit && right.Count < 0;
            if (lhit && !ltraverse) candidates.push_back(node.LeftFirst);
            if (rhit && !rtraverse) candidates.push_back(node.LeftFirst + 1);
            if (!ltraverse && !rtraverse) stack.pop_back();
            else {
                stack.back() = ltraverse ? node.LeftFirst : node.LeftFirst + 1;
                if (ltraverse && rtraverse) stack.push_back(node.LeftFirst + 1);
            }
            if (stack.size() > peak) peak = stack.size();
        }
        if (stack_peak) *stack_peak = (int)peak;
    }
    if (candidate_count) *candidate_count = (int)candidates.size();
    for (int idx : candidates) {
        const Node& node = nodes[idx];
        for (int j = node.LeftFirst; j < node.LeftFirst + node.Count; j++) {
            float t, u, v;
            if (triangleGlsl(triangles[j], ray, hit.Distance, &t, &u, &v)) {
                const Triangle& tri = triangles[j];
                const float w = 1.0f - u - v;
                hit.Distance = t;
                hit.Tri = j;
                hit.Normal = Normalize((tri.V1.Normal * u + tri.V2.Normal * v) + tri.V3.Normal * w);   // extend.glsl:204-209
            }
        }
    }
    return hit;
}

// extend.glsl:23-27
uint32_t PackNormalGlsl(Vec3 n) {
    const float f = 65535.0f / sqrtf(8.0f * n.z + 8.0f);
    return (uint32_t)(n.x * f + 32767.0f) + ((uint32_t)(n.y * f + 32767.0f) << 16);
}
// shade.glsl:29-36
Vec3 UnpackNormalGlsl(uint32_t p) {
    float x = (float)(p & 65535u) * (2.0f / 65535.0f), y = (float)(p >> 16) * (2.0f / 65535.0f), z = 0.0f, w = 0.0f;
    x += -1.0f; y += -1.0f; z += 1.0f; w += -1.0f;
    float l = (x * -x + y * -y) + z * -w;
    z = l;
    l = sqrtf(l);
    x *= l;
    y *= l;
    return Vec3{x * 2.0f + 0.0f, y * 2.0f + 0.0f, z * 2.0f + -1.0f};
}

// ---- compute.glsl's statements of the shading helpers
static inline void xor_shift(uint32_t& seed) { seed ^= seed << 13; seed ^= seed >> 17; seed ^= seed << 5; }            // compute.glsl:108-113
static inline float random_float(uint32_t& seed) { xor_shift(seed); return (float)seed * (float)(1.0 / 4294967295.0); }  // compute.glsl:115-119

// compute.glsl:290-296, pow(x, 5) taken as the product the product path uses (x2 = x*x, x4 = x2*x2, x4*x)
float SecondSchlick(Vec3 direction, Vec3 normal, float n1, float n2) {
    float r0 = (n1 - n2) / (n1 + n2);
    r0 = r0 * r0;
    const float c = 1.0f - Dot(normal, Vec3{-direction.x, -direction.y, -direction.z});
    const float c2 = c * c, c4 = c2 * c2;
    return r0 + (1.0f - r0) * (c4 * c);
}

// compute.glsl:298-313, sin / cos through the same fixed polynomials as everywhere else
Vec3 SecondDiffuseReflection(Vec3 normal, uint32_t& seed) {
    const Vec3 axis = std::fabs(normal.x) > 0.95f ? Vec3{0, 1, 0} : Vec3{1, 0, 0};
    const Vec3 u = Normalize(Cross(normal, axis));
    const Vec3 v = Cross(u, normal);
    const float r0 = random_float(seed);
    const float r1 = random_float(seed);
    const float r = sqrtf(r0);
    const float theta = 2.0f * PI * r1;
    return ((r * det_cos(theta)) * u + (r * det_sin(theta)) * v) + sqrtf(1.0f - r0) * normal;
}

}  // namespace oracle
