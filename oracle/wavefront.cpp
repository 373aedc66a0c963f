// CPU ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// Restates PathTracer/Shaders/Wavefront/{generate,extend,shade}.glsl and the
// host schedule of PathTracer/Wavefront.cs:40-56 as a per-path loop.
//
// Deliberate, documented departures from the reference (each is a reference
// bug or a race that makes its own output non-reproducible):
//  * random streams are keyed on the path's birth index and the bounce, not on
//    the queue slot (shade.glsl:352 reuses one stream for every bounce and the
//    slot depends on a racy compaction, shade.glsl:447-458);
//  * every path is followed to max_bounces; the host's "stop when <= 50 rays
//    remain" (Wavefront.cs:45) is not applied, and truncated paths add nothing;
//  * total internal reflection reflects instead of producing a zero direction
//    (GLSL refract returns vec3(0), shade.glsl:402);
//  * the hit normal is interpolated at full precision instead of going
//    through the 16+16-bit packNormal (extend.glsl:23-27);
//  * the pixel is a sum over samples divided by the sample count at read-out
//    instead of a running mean (shade.glsl:89-96);
//  * pixel groups cover ceil(W/8) x ceil(H/8) and out-of-image threads emit no
//    ray (generate.glsl:177-192 leaves stale rays for sizes not divisible by 8);
//  * sin, cos, exp and pow(x,5) are fixed polynomials / products built from
//    IEEE +,-,*,floor only, so that CPU and GPU agree bit for bit;
//  * optional next-event estimation (not in the reference).
#include "oracle.hpp"
#include <cmath>
#include <cstring>
#include <limits>
#include <algorithm>

namespace oracle {

static const float INF = std::numeric_limits<float>::infinity();
static const float EPSILON = 0.00001f;
static const float PI = 3.14159274101257324f;
static const float INVPI = 0.318309873342514038f;

// ------------------------------------------------ deterministic elementary fns
static inline void det_sincos(float x, float* s, float* c) {
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

float det_sin(float x) { float s, c; det_sincos(x, &s, &c); return s; }
float det_cos(float x) { float s, c; det_sincos(x, &s, &c); return c; }

float det_exp(float x) {
    if (x > 88.0f) return INF;
    if (x < -87.0f) return 0.0f;
    float n = floorf(x * 1.44269504088896341f + 0.5f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float z = r * r;
    float p = ((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r + 1.6666665459e-1f) * r + 5.0000001201e-1f;
    float y = p * z + r + 1.0f;
    uint32_t bits = (uint32_t)((int)n + 127) << 23;
    float scale;
    memcpy(&scale, &bits, 4);
    return y * scale;
}

static inline float pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
static inline float clamp01(float x) { float a = (x < 0.0f) ? 0.0f : x; return (1.0f < a) ? 1.0f : a; }
static inline float max0(float x) { return (0.0f < x) ? x : 0.0f; }

// ----------------------------------------------------------------- random
// generate.glsl:10-21
static inline void xorShift(uint32_t& seed) {
    seed ^= seed << 13;
    seed ^= seed >> 17;
    seed ^= seed << 5;
}
static inline float randomFloat(uint32_t& seed) {
    xorShift(seed);
    return (float)seed * 2.3283064365386963e-10f;   // float(1.0 / 4294967295.0) == 2^-32
}
static inline uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
static const uint32_t STREAM_GENERATE = 2983742873u;   // generate.glsl:174
static const uint32_t STREAM_SHADE = 984524691u;       // shade.glsl:352
static const uint32_t STREAM_NEE = 0x68E31DA4u;
static inline uint32_t streamSeed(uint32_t gidx, uint32_t frames, uint32_t stream, uint32_t bounce, int mode) {
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

// ------------------------------------------------------------ pixel mapping
// generate.glsl:160-168
static inline uint32_t compact(uint32_t x) {
    x &= 0x55555555;
    x = (x ^ (x >> 1)) & 0x33333333;
    x = (x ^ (x >> 2)) & 0x0f0f0f0f;
    x = (x ^ (x >> 4)) & 0x00ff00ff;
    x = (x ^ (x >> 8)) & 0x0000ffff;
    return x;
}

// Local thread index -> pixel.  Groups of 64 threads are 8x8 Morton tiles
// (generate.glsl:177-187); tile g belongs to rank g % world and is that rank's
// local tile g / world.  *gidx_out is the thread index a single-GPU run gives
// the same pixel, which keys the random streams.
static bool pixelOfIndex(uint32_t index, int width, int height, int rank, int world, uint32_t* pixel, uint32_t* gidx_out) {
    uint32_t gcx = (uint32_t)(width + 7) / 8, gcy = (uint32_t)(height + 7) / 8;
    uint32_t lid = index & 63u;
    uint32_t group = (index >> 6) * (uint32_t)world + (uint32_t)rank;
    if (group >= gcx * gcy) return false;
    uint32_t gx = group % gcx, gy = group / gcx;
    uint32_t sx = gx * 8 + compact(lid), sy = gy * 8 + compact(lid >> 1);
    if (sx >= (uint32_t)width || sy >= (uint32_t)height) return false;
    *pixel = sx + sy * (uint32_t)width;
    *gidx_out = group * 64u + lid;
    return true;
}

uint32_t RayPixelOfIndex(uint32_t index, int width, int height, int rank, int world, bool* valid) {
    uint32_t pixel = 0, gidx = 0;
    bool ok = pixelOfIndex(index, width, height, rank, world, &pixel, &gidx);
    if (valid) *valid = ok;
    return pixel;
}

struct Vec4 { float x, y, z, w; };
static inline Vec4 operator*(Vec4 a, Vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline Vec4 operator*(Vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline Vec4 operator/(Vec4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
static inline Vec4 operator+(Vec4 a, Vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline Vec4 ld4(const float* p) { return {p[0], p[1], p[2], p[3]}; }

// generate.glsl:135-156 and 170-200
bool GenerateRay(uint32_t index, const RenderState& rs, const FrameState& fs, const PathConfig& cfg, Vec3* origin, Vec3* dir, uint32_t* pixel) {
    uint32_t gidx;
    if (!pixelOfIndex(index, rs.Width, rs.Height, cfg.tile_rank, cfg.tile_world, pixel, &gidx)) return false;
    uint32_t seed = streamSeed(gidx, (uint32_t)fs.Frames, STREAM_GENERATE, 0, cfg.rng_mode);
    uint32_t sx = *pixel % (uint32_t)rs.Width, sy = *pixel / (uint32_t)rs.Width;

    const float r0 = randomFloat(seed);
    const float r1 = randomFloat(seed);
    const float r2 = randomFloat(seed) - 0.5f;
    const float r3 = randomFloat(seed) - 0.5f;

    const float x = ((float)sx + r0) / (float)rs.Width;
    const float y = ((float)((uint32_t)rs.Height - sy) + r1) / (float)rs.Height;

    const CameraState& cam = fs.Camera;
    const Vec3 t = (cam.P1 + x * (cam.P2 - cam.P1)) + y * (cam.P3 - cam.P1);
    const Vec3 p = cam.Position + 0.05f * (r2 * cam.Right + r3 * cam.Up);

    *dir = Normalize(t - p);
    *origin = p;
    return true;
}

// ------------------------------------------------------------ light table
// Emissive triangles (shade.glsl:366 decides emissive-ness) in triangle order,
// with a running-sum area table for area-proportional selection.
LightTable BuildLightTable(const Scene& s) {
    LightTable lt;
    const auto& tris = s.accelerator.triangles;
    float sum = 0;
    for (int j = 0; j < (int)tris.size(); j++) {
        const Material& m = s.materials[tris[j].Material];
        if (m.Emissive[0] > 0 || m.Emissive[1] > 0 || m.Emissive[2] > 0) {
            Vec3 e1 = tris[j].V2.Position - tris[j].V1.Position;
            Vec3 e2 = tris[j].V3.Position - tris[j].V1.Position;
            Vec3 n = Cross(e1, e2);
            float area = 0.5f * sqrtf(Dot(n, n));
            sum = sum + area;
            lt.tris.push_back(j);
            lt.cdf.push_back(sum);
        }
    }
    lt.totalArea = sum;
    return lt;
}

// ------------------------------------------------------------------- shade
// shade.glsl:150-156
static float schlick(Vec3 direction, Vec3 normal, float n1, float n2) {
    float r0 = (n1 - n2) / (n1 + n2);
    r0 = r0 * r0;
    Vec3 nd{-direction.x, -direction.y, -direction.z};
    return r0 + (1.0f - r0) * pow5(1.0f - Dot(normal, nd));
}
// shade.glsl:158-162
static Vec3 schlick3(Vec3 ks, Vec3 l, Vec3 n) {
    const float ldotn = clamp01(Dot(l, n));
    const float p = pow5(1.0f - ldotn);
    return {ks.x + (1.0f - ks.x) * p, ks.y + (1.0f - ks.y) * p, ks.z + (1.0f - ks.z) * p};
}
// shade.glsl:164-176
static float smithGgxMasking(Vec3 v, Vec3 n, float a2) {
    const float ndotv = clamp01(Dot(n, v));
    const float c = sqrtf(a2 + (1.0f - a2) * ndotv * ndotv) + ndotv;
    if (c == 0) return 0;
    return 2.0f * ndotv / c;
}
// shade.glsl:178-194
static float smithGgxShadowMasking(Vec3 l, Vec3 v, Vec3 n, float a2) {
    const float ndotl = clamp01(Dot(n, l));
    const float ndotv = clamp01(Dot(n, v));
    const float a = ndotv * sqrtf(a2 + (1.0f - a2) * ndotl * ndotl);
    const float b = ndotl * sqrtf(a2 + (1.0f - a2) * ndotv * ndotv);
    const float denom = a + b;
    if (denom == 0) return 0;
    return 2.0f * ndotl * ndotv / denom;
}
static inline Vec3 reflect(Vec3 i, Vec3 n) { float d2 = 2.0f * Dot(n, i); return i - d2 * n; }

// shade.glsl:210-225
static Vec3 lambertImportanceSample(Vec3 normal, uint32_t& seed) {
    const Vec3 axis = fabsf(normal.x) > 0.95f ? Vec3{0, 1, 0} : Vec3{1, 0, 0};
    const Vec3 u = Normalize(Cross(normal, axis));
    const Vec3 v = Cross(u, normal);
    const float r0 = randomFloat(seed);
    const float r1 = randomFloat(seed);
    const float r = sqrtf(r0);
    const float theta = 2.0f * PI * r1;
    float st, ct;
    det_sincos(theta, &st, &ct);
    return ((r * ct) * u + (r * st) * v) + sqrtf(1.0f - r0) * normal;
}

// shade.glsl:238-257
static Vec3 ggxImportance(Vec3 view, float alpha, uint32_t& seed) {
    const float r1 = randomFloat(seed);
    const float r2 = randomFloat(seed);
    const Vec3 wo = Normalize(Vec3{view.x * alpha, view.y * alpha, view.z});
    const Vec3 t1 = (wo.z < 0.999f) ? Normalize(Cross(wo, Vec3{0, 0, 1})) : Vec3{1, 0, 0};
    const Vec3 t2 = Cross(t1, wo);
    float a = 1.0f / (1.0f + wo.z);
    float r = sqrtf(r1);
    float phi = (r2 < a) ? (r2 / a) * PI : PI + (r2 - a) / (1.0f - a) * PI;
    float sphi, cphi;
    det_sincos(phi, &sphi, &cphi);
    float p1 = r * cphi;
    float p2 = r * sphi * ((r2 < a) ? 1.0f : wo.z);
    Vec3 n = (p1 * t1 + p2 * t2) + sqrtf(max0(1.0f - p1 * p1 - p2 * p2)) * wo;
    return Normalize(Vec3{alpha * n.x, alpha * n.y, max0(n.z)});
}

// shade.glsl:296-330.  Returns the weight; *ok = false when the sample is rejected (weight 0).
static Vec4 sampleGgx(const Material& mat, Vec3 world_n, Vec3 world_v, Vec3* world_l, uint32_t& seed) {
    Vec3 W = fabsf(world_n.x) > 0.9f ? Vec3{0, 1, 0} : Vec3{1, 0, 0};
    Vec3 N = world_n;
    Vec3 T = Normalize(Cross(N, W));
    Vec3 B = Cross(N, T);
    Vec3 v{Dot(world_v, T), Dot(world_v, B), Dot(world_v, N)};
    Vec3 m = ggxImportance(v, mat.Roughness, seed);
    Vec3 nv{-v.x, -v.y, -v.z};
    Vec3 l = reflect(nv, m);
    if (Dot(l, m) < 0) return {0, 0, 0, 0};
    *world_l = (l.x * T + l.y * B) + l.z * N;
    float a2 = mat.Roughness * mat.Roughness;
    Vec3 f = schlick3(Vec3{mat.Color[0], mat.Color[1], mat.Color[2]}, m, l);
    float g1 = smithGgxMasking(v, m, a2);
    float g2 = smithGgxShadowMasking(l, v, m, a2);
    if (g1 == 0) return {0, 0, 0, 0};
    float g = g2 / g1;
    return {f.x * g, f.y * g, f.z * g, 0.0f * g};
}

float ProbeSchlick(Vec3 direction, Vec3 normal, float n1, float n2) { return schlick(direction, normal, n1, n2); }
Vec3 ProbeLambertSample(Vec3 normal, uint32_t& seed) { return lambertImportanceSample(normal, seed); }
// shade.glsl:413-426 as RenderSamples applies it: throughput (1,1,1) * (color / pi) * dot(l, n) / (clamp(dot(n, l)) / pi)
void ProbeDiffuseThroughput(Vec3 normal, Vec3 bounce, const float color[3], float out[3]) {
    const float pdf = clamp01(Dot(normal, bounce)) * INVPI;
    for (int k = 0; k < 3; k++) out[k] = pdf == 0 ? 0.0f : 1.0f * (color[k] * INVPI) * Dot(bounce, normal) / pdf;
}

static int pickLight(const LightTable& lt, float r) {
    // smallest i with r * total < cdf[i]
    float target = r * lt.totalArea;
    int lo = 0, hi = (int)lt.cdf.size() - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (target < lt.cdf[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

void RenderSamples(const Scene& scene, const LightTable& lights, const RenderState& rs, FrameState fs, const PathConfig& cfg,
                   int spp, float* sum, uint32_t group0, uint32_t group1, RenderStats* stats, int traversal, uint32_t group_stride) {
    const Accelerator& acc = scene.accelerator;
    const Vec4 sky = ld4(rs.SkyColor);
    if (group_stride == 0) group_stride = 1;
    for (int s = 0; s < spp; s++, fs.Frames++, fs.Samples++) {
        for (uint32_t index = group0 * 64u; index < group1 * 64u; index++) {
            if ((index >> 6) % group_stride != 0) continue;   // sub-sampled runs trace every n-th pixel group
            Vec3 o, d;
            uint32_t pixel, gidx;
            if (!pixelOfIndex(index, rs.Width, rs.Height, cfg.tile_rank, cfg.tile_world, &pixel, &gidx)) continue;
            GenerateRay(index, rs, fs, cfg, &o, &d, &pixel);
            Vec4 T{1, 1, 1, 1};
            Vec4 plane{0, 0, 0, 0};
            bool lastSpecular = true;
            float lastPdf = 0.0f;   // density of the direction that led to the current vertex (nee == 2 only)
            for (int bounce = 0; bounce < cfg.max_bounces; bounce++) {
                // ---- extend (extend.glsl:301-309)
                Ray ray = MakeRay(o, d);
                Intersection hit;
                hit.Distance = INF;
                hit.Tri = -1;
                hit.U = hit.V = 0;
                if (stats) stats->extend_rays++;
                Vec3 glslNormal{0, 0, 0};
                if (cfg.literal & LITERAL_GLSL_EXTEND) {
                    const GlslHit g = IntersectGlsl(acc, ray, nullptr, nullptr);
                    hit.Tri = g.Tri;
                    if (g.Tri >= 0) hit.Distance = g.Distance;
                    glslNormal = g.Normal;
                } else if (traversal == 0) acc.Intersect(ray, hit);
                else acc.IntersectCanonical(ray, hit, stats ? &stats->extend : nullptr);

                // ---- shade (shade.glsl:349-459)
                uint32_t seed = streamSeed(gidx, (uint32_t)fs.Frames, STREAM_SHADE, (uint32_t)bounce, cfg.rng_mode);
                if (hit.Tri < 0) {
                    plane = plane + T * sky;
                    break;
                }
                const Triangle& tri = acc.triangles[hit.Tri];
                const Material& mat = scene.materials[tri.Material];
                if (mat.Emissive[0] > 0 || mat.Emissive[1] > 0 || mat.Emissive[2] > 0) {
                    if (!cfg.nee || lastSpecular) plane = plane + T * ld4(mat.Emissive);
                    else if (cfg.nee == 2) {
                        // multiple importance sampling, balance heuristic: this emitter was reached by BRDF sampling at a
                        // vertex that also sampled the lights; pl = density of the light sampling for this point
                        const Vec3 ng = Cross(tri.V2.Position - tri.V1.Position, tri.V3.Position - tri.V1.Position);
                        const float cosy = fabsf(Dot(d, ng)) / sqrtf(Dot(ng, ng));
                        const float dist2 = hit.Distance * hit.Distance;
                        if (cosy > 0.0f && dist2 > 0.0f) {
                            const float pl = dist2 / (cosy * lights.totalArea);
                            plane = plane + (T * ld4(mat.Emissive)) * (lastPdf / (lastPdf + pl));
                        }
                    }
                    break;
                }
                // interpolated shading normal (extend.glsl:199-209), barycentrics of the C# test:
                // u weighs V2, v weighs V3, 1-u-v weighs V1
                const float w = 1.0f - hit.U - hit.V;
                Vec3 normal = Normalize((tri.V1.Normal * w + tri.V2.Normal * hit.U) + tri.V3.Normal * hit.V);
                if (cfg.literal & LITERAL_GLSL_EXTEND) normal = glslNormal;
                if (cfg.literal & LITERAL_PACKED_NORMAL) normal = UnpackNormalGlsl(PackNormalGlsl(normal));   // storeHit / getHit, extend.glsl:124-131, shade.glsl:115-123
                const float dist = hit.Distance;
                Vec3 bounceDir{0, 0, 0};
                Vec4 throughput = T;
                const Vec4 color = ld4(mat.Color);
                const Vec3 hitP = o + d * dist;

                if (mat.Type == Dielectric) {
                    float n1 = 1.0f, n2 = mat.Index;
                    if (Dot(d, normal) > 0) {
                        n1 = mat.Index;
                        n2 = 1.0f;
                        normal = normal * -1.0f;
                        throughput.x = throughput.x * det_exp(-mat.Color[0] * dist);
                        throughput.y = throughput.y * det_exp(-mat.Color[1] * dist);
                        throughput.z = throughput.z * det_exp(-mat.Color[2] * dist);
                    }
                    const float r = randomFloat(seed);
                    const float sch = schlick(d, normal, n1, n2);
                    if (r < sch) {
                        bounceDir = reflect(d, normal);
                    } else {
                        const float eta = n1 / n2;
                        const float dn = Dot(normal, d);
                        const float k = 1.0f - eta * eta * (1.0f - dn * dn);
                        if (k < 0.0f) bounceDir = reflect(d, normal);
                        else bounceDir = eta * d - (eta * dn + sqrtf(k)) * normal;
                    }
                    throughput = throughput * color;
                }
                if (mat.Type == Mirror) {
                    bounceDir = reflect(d, normal);
                    throughput = throughput * color;
                }
                if (mat.Type == Diffuse) {
                    bounceDir = lambertImportanceSample(normal, seed);
                    const Vec4 brdf = color * INVPI;
                    const float pdf = clamp01(Dot(normal, bounceDir)) * INVPI;
                    // next-event estimation: one emissive triangle, area-proportional
                    if (cfg.nee && !lights.tris.empty()) {
                        uint32_t ns = streamSeed(gidx, (uint32_t)fs.Frames, STREAM_NEE, (uint32_t)bounce, cfg.rng_mode);
                        const float rl = randomFloat(ns);
                        const float ru = randomFloat(ns);
                        const float rv = randomFloat(ns);
                        const int li = pickLight(lights, rl);
                        const Triangle& lt = acc.triangles[lights.tris[li]];
                        const float su = sqrtf(ru);
                        const float b0 = 1.0f - su;
                        const float b1 = rv * su;
                        const float b2 = 1.0f - b0 - b1;
                        const Vec3 yp = (lt.V1.Position * b0 + lt.V2.Position * b1) + lt.V3.Position * b2;
                        const Vec3 wv = yp - hitP;
                        const float dist2 = Dot(wv, wv);
                        const float dl = sqrtf(dist2);
                        const Vec3 wn = wv / dl;
                        const float cosx = Dot(wn, normal);
                        const Vec3 ng = Cross(lt.V2.Position - lt.V1.Position, lt.V3.Position - lt.V1.Position);
                        const float ngl = sqrtf(Dot(ng, ng));
                        const float cosy = fabsf(Dot(wn, ng)) / ngl;
                        if (cosx > 0.0f && cosy > 0.0f && dist2 > 0.0f) {
                            const float geom = cosx * cosy * lights.totalArea / dist2;
                            const Material& lm = scene.materials[lt.Material];
                            Vec4 c = (T * brdf) * ld4(lm.Emissive) * geom;
                            if (cfg.nee == 2) {
                                const float pl = dist2 / (cosy * lights.totalArea);
                                c = c * (pl / (pl + clamp01(cosx) * INVPI));
                            }
                            const Vec3 so = hitP + wn * EPSILON;
                            if (stats) stats->shadow_rays++;
                            if (!acc.Occluded(MakeRay(so, wn), dl * 0.999f, stats ? &stats->shadow : nullptr)) plane = plane + c;
                        }
                    }
                    if (pdf == 0) break;   // shade.glsl:419-423: the path ends without a pixel write
                    throughput = throughput * brdf * Dot(bounceDir, normal) / pdf;
                }
                if (mat.Type == Metal) {
                    Vec3 nd{-d.x, -d.y, -d.z};
                    throughput = throughput * sampleGgx(mat, normal, nd, &bounceDir, seed);
                }

                float roulette = (throughput.x < throughput.y) ? throughput.y : throughput.x;
                roulette = (roulette < throughput.z) ? throughput.z : roulette;
                if (randomFloat(seed) > roulette) break;   // pixel += 0
                throughput = throughput * (1.0f / roulette);

                o = hitP + bounceDir * EPSILON;
                d = bounceDir;
                T = throughput;
                lastSpecular = (mat.Type != Diffuse);
                if (mat.Type == Diffuse && cfg.nee == 2) {   // the queue keeps the density with its low bit cleared (the bit is the specular flag)
                    const float pdf = clamp01(Dot(normal, bounceDir)) * INVPI;
                    uint32_t b;
                    std::memcpy(&b, &pdf, 4);
                    b &= ~1u;
                    std::memcpy(&lastPdf, &b, 4);
                }
            }
            float* px = sum + 4 * (size_t)pixel;
            if (cfg.literal & LITERAL_RUNNING_MEAN) {   // shade.glsl:89-96 with samples = FrameState.Samples of this frame
                const float n = (float)fs.Samples;
                px[0] = (px[0] * n + plane.x) / (n + 1.0f); px[1] = (px[1] * n + plane.y) / (n + 1.0f);
                px[2] = (px[2] * n + plane.z) / (n + 1.0f); px[3] = (px[3] * n + plane.w) / (n + 1.0f);
            } else {
                px[0] += plane.x; px[1] += plane.y; px[2] += plane.z; px[3] += plane.w;
            }
        }
    }
}

}  // namespace oracle
