// CPU ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// extern "C" surface of liboracle.so for ctypes (tests/, smoke(), bench.py's
// cpu_baseline leg).  Every function returns 0 on success, -1 on error with
// the message in orc_last_error().
#include "oracle.hpp"
#include <cstring>
#include <cmath>
#include <string>
#include <thread>
#include <vector>
#include <limits>
#include <exception>
#include <stdexcept>
#include <algorithm>

using namespace oracle;

static thread_local std::string g_err;
#define ORC_TRY try {
#define ORC_CATCH } catch (const std::exception& e) { g_err = e.what(); return -1; } return 0;

struct OrcScene {
    Scene scene;
    LightTable lights;
    bool built = false;
};

template <class F>
static void parallelFor(int n, int threads, F f) {
    if (threads <= 1 || n < 2) { f(0, n, 0); return; }
    std::vector<std::thread> pool;
    int chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        int a = t * chunk, b = std::min(n, a + chunk);
        if (a >= b) break;
        pool.emplace_back([=]() { f(a, b, t); });
    }
    for (auto& th : pool) th.join();
}

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

int orc_random_next(int32_t seed, int n, int32_t* out) {
    DotNetRandom r(seed);
    for (int i = 0; i < n; i++) out[i] = r.Next();
    return 0;
}

int orc_random_doubles(int32_t seed, int n, double* out) {
    DotNetRandom r(seed);
    for (int i = 0; i < n; i++) out[i] = r.NextDouble();
    return 0;
}

// kind: 0 sin, 1 cos, 2 exp
int orc_det_math(int kind, const float* x, int n, float* out) {
    for (int i = 0; i < n; i++) out[i] = kind == 0 ? det_sin(x[i]) : (kind == 1 ? det_cos(x[i]) : det_exp(x[i]));
    return 0;
}

// Parses <obj_dir>/<name>.obj for each name and writes the blob of parsed arrays.
int orc_write_mesh_blob(const char* obj_dir, const char* names_csv, const char* out_path) {
    ORC_TRY
    std::vector<std::pair<std::string, Mesh>> meshes;
    std::string names(names_csv), dir(obj_dir);
    if (!dir.empty() && dir.back() != '/') dir += '/';
    size_t i = 0;
    while (i <= names.size()) {
        size_t j = names.find(',', i);
        if (j == std::string::npos) j = names.size();
        if (j > i) {
            std::string nm = names.substr(i, j - i);
            meshes.push_back({nm, Mesh::LoadFromFile(dir + nm + ".obj")});
        }
        i = j + 1;
    }
    if (!WriteMeshBlob(out_path, meshes)) throw std::runtime_error("cannot write blob");
    ORC_CATCH
}

void* orc_scene_create(const char* mesh_source) {
    OrcScene* s = new OrcScene();
    s->scene.meshSource = mesh_source ? mesh_source : "";
    return s;
}

void orc_scene_destroy(void* h) { delete (OrcScene*)h; }

int orc_scene_clear(void* h) {
    OrcScene* s = (OrcScene*)h;
    s->scene.Clear();
    s->built = false;
    return 0;
}

int orc_scene_initialize(void* h, int seed) {
    ORC_TRY
    ((OrcScene*)h)->scene.InitializeScene(seed);
    ORC_CATCH
}

int orc_scene_soup(void* h, int n, uint32_t seed) {
    ORC_TRY
    ((OrcScene*)h)->scene.GenerateSoup(n, seed);
    ORC_CATCH
}

// matrix16: row-major Matrix4x4 or NULL (identity); material48: Material or NULL (use the MTL groups)
int orc_scene_add_mesh(void* h, const char* name, int normalized, const float* matrix16, const void* material48) {
    ORC_TRY
    Matrix4x4 m = MatIdentity();
    if (matrix16) memcpy(&m, matrix16, 64);
    if (material48) {
        Material mat;
        memcpy(&mat, material48, 48);
        ((OrcScene*)h)->scene.AddMesh(name, normalized != 0, m, mat);
    } else {
        ((OrcScene*)h)->scene.AddMesh(name, normalized != 0, m);
    }
    ORC_CATCH
}

int orc_scene_floor(void* h, const void* a48, const void* b48) {
    ORC_TRY
    Material a, b;
    memcpy(&a, a48, 48);
    memcpy(&b, b48, 48);
    ((OrcScene*)h)->scene.GenerateTiledFloor(a, b);
    ORC_CATCH
}

// raw triangles with material indices into `materials` set by orc_scene_set_materials
int orc_scene_add_triangles(void* h, const void* tris112, int n) {
    ORC_TRY
    const Triangle* t = (const Triangle*)tris112;
    for (int i = 0; i < n; i++) ((OrcScene*)h)->scene.accelerator.AddTriangle(t[i]);
    ORC_CATCH
}

int orc_scene_set_materials(void* h, const void* mats48, int n) {
    ORC_TRY
    auto& m = ((OrcScene*)h)->scene.materials;
    m.resize(n);
    memcpy(m.data(), mats48, (size_t)n * 48);
    ORC_CATCH
}

int orc_scene_build(void* h) {
    ORC_TRY
    OrcScene* s = (OrcScene*)h;
    s->scene.Build();
    s->lights = BuildLightTable(s->scene);
    s->built = true;
    ORC_CATCH
}

// Installs arrays some other builder produced (Material[], Triangle[] in built order, Node[]) without building, so the
// oracle's walks can run over trees too large for its one-thread literal build.  The arrays are taken as they are.
int orc_scene_set_built(void* h, const void* mats48, int nMat, const void* tris112, int nTri, const void* nodes32, int nNodes) {
    ORC_TRY
    OrcScene* s = (OrcScene*)h;
    s->scene.Clear();
    s->scene.materials.resize(nMat);
    memcpy(s->scene.materials.data(), mats48, (size_t)nMat * 48);
    auto& a = s->scene.accelerator;
    a.triangles.resize(nTri);
    memcpy(a.triangles.data(), tris112, (size_t)nTri * 112);
    a.nodes.resize(nNodes);
    memcpy(a.nodes.data(), nodes32, (size_t)nNodes * 32);
    a.quadRecords = a.CountQuadRecords();
    s->lights = BuildLightTable(s->scene);
    s->built = true;
    ORC_CATCH
}

// counts: materials, triangles (post build), nodes, references (pre build), max depth, lights
// which node records the canonical walks restate: 0 exact, 1 quantised, 2 the GPU library's rule (quantised for trees beyond L2)
int orc_scene_set_quantized(void* h, int mode) {
    ORC_TRY
    if (mode < 0 || mode > 2) throw std::runtime_error("quantized mode must be 0, 1 or 2");
    ((OrcScene*)h)->scene.accelerator.quantized = mode;
    ORC_CATCH
}

int orc_scene_counts(void* h, int32_t* out6) {
    OrcScene* s = (OrcScene*)h;
    out6[0] = (int)s->scene.materials.size();
    out6[1] = (int)s->scene.accelerator.triangles.size();
    out6[2] = (int)s->scene.accelerator.nodes.size();
    out6[3] = (int)s->scene.accelerator.References().size();
    out6[4] = s->scene.accelerator.MaxDepth();
    out6[5] = (int)s->lights.tris.size();
    return 0;
}

int orc_scene_copy(void* h, void* mats48, void* tris112, void* nodes32, void* refs112) {
    OrcScene* s = (OrcScene*)h;
    const auto& a = s->scene.accelerator;
    if (mats48) memcpy(mats48, s->scene.materials.data(), s->scene.materials.size() * 48);
    if (tris112) memcpy(tris112, a.triangles.data(), a.triangles.size() * 112);
    if (nodes32) memcpy(nodes32, a.nodes.data(), a.nodes.size() * 32);
    if (refs112) memcpy(refs112, a.References().data(), a.References().size() * 112);
    return 0;
}

int orc_scene_lights(void* h, int32_t* tris, float* cdf, float* total) {
    OrcScene* s = (OrcScene*)h;
    if (tris) memcpy(tris, s->lights.tris.data(), s->lights.tris.size() * 4);
    if (cdf) memcpy(cdf, s->lights.cdf.data(), s->lights.cdf.size() * 4);
    if (total) *total = s->lights.totalArea;
    return 0;
}

// mode: 0 literal C# traversal, 1 canonical near-first, 2 brute force.
// rays: n x 6 floats (origin, direction).  Outputs may be NULL.  stats3 = {steps, tris, rays} (mode 1).
int orc_intersect(void* h, int mode, const float* rays, int n, int32_t* tri, float* t, float* uv, float* point, uint64_t* stats3, int threads) {
    ORC_TRY
    OrcScene* s = (OrcScene*)h;
    const Accelerator& acc = s->scene.accelerator;
    std::vector<TraversalStats> st(std::max(1, threads), TraversalStats{0, 0, 0});
    parallelFor(n, threads, [&](int a, int b, int tid) {
        for (int i = a; i < b; i++) {
            Ray r = MakeRay({rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, {rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]});
            Intersection it;
            it.Distance = std::numeric_limits<float>::infinity();
            it.Point = {0, 0, 0};
            it.Tri = -1;
            it.U = it.V = 0;
            if (mode == 0) acc.Intersect(r, it);
            else if (mode == 1) acc.IntersectCanonical(r, it, &st[tid]);
            else acc.IntersectBrute(r, it);
            if (tri) tri[i] = it.Tri;
            if (t) t[i] = it.Distance;
            if (uv) { uv[2 * i] = it.U; uv[2 * i + 1] = it.V; }
            if (point) { point[3 * i] = it.Point.x; point[3 * i + 1] = it.Point.y; point[3 * i + 2] = it.Point.z; }
        }
    });
    if (stats3) {
        stats3[0] = stats3[1] = stats3[2] = 0;
        for (auto& x : st) { stats3[0] += x.steps; stats3[1] += x.tris; stats3[2] += x.rays; }
    }
    ORC_CATCH
}

int orc_occluded(void* h, const float* rays, const float* tmax, int n, uint8_t* out, int threads) {
    ORC_TRY
    const Accelerator& acc = ((OrcScene*)h)->scene.accelerator;
    parallelFor(n, threads, [&](int a, int b, int) {
        for (int i = a; i < b; i++) {
            Ray r = MakeRay({rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, {rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]});
            out[i] = acc.Occluded(r, tmax[i]) ? 1 : 0;
        }
    });
    ORC_CATCH
}

// The GLSL closest-hit walk + triangle test (extend.glsl:171-299) for n rays: triangle, distance, normalized hit normal;
// peaks2 = { largest traversal stack, largest candidate list } over the rays (the shader's arrays hold 32 and 128).
int orc_intersect_glsl(void* h, const float* rays, int n, int32_t* tri, float* t, float* normal3, int32_t* peaks2, int threads) {
    ORC_TRY
    const Accelerator& acc = ((OrcScene*)h)->scene.accelerator;
    std::vector<int> peakS(std::max(1, threads), 0), peakC(std::max(1, threads), 0);
    parallelFor(n, threads, [&](int a, int b, int tid) {
        for (int i = a; i < b; i++) {
            Ray r = MakeRay({rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, {rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]});
            int sp = 0, cc = 0;
            const GlslHit g = IntersectGlsl(acc, r, &sp, &cc);
            tri[i] = g.Tri;
            t[i] = g.Distance;
            normal3[3 * i] = g.Normal.x; normal3[3 * i + 1] = g.Normal.y; normal3[3 * i + 2] = g.Normal.z;
            peakS[tid] = std::max(peakS[tid], sp);
            peakC[tid] = std::max(peakC[tid], cc);
        }
    });
    if (peaks2) { peaks2[0] = *std::max_element(peakS.begin(), peakS.end()); peaks2[1] = *std::max_element(peakC.begin(), peakC.end()); }
    ORC_CATCH
}

// Two statements of one helper side by side.  kind 0: schlick(direction3, normal3, n1, n2) -> 1 float each;
// kind 1: cosine-weighted direction(normal3, seed bits) -> 3 floats each; kind 2: pack/unpack of a unit normal3 -> 3 floats
// (out_b = the input); kind 3: diffuse throughput for (normal3, color3, seed bits): out_a = shade.glsl's brdf * cos / pdf over
// the direction of kind 1, out_b = compute.glsl's plain `* color`.  in: n rows of 8 floats; out: n rows of 3 floats.
int orc_second_statement(int kind, const float* in, int n, float* out_a, float* out_b) {
    ORC_TRY
    for (int i = 0; i < n; i++) {
        const float* x = in + 8 * i;
        float* a = out_a + 3 * i;
        float* b = out_b + 3 * i;
        uint32_t seedA, seedB;
        memcpy(&seedA, &x[6], 4);
        seedB = seedA;
        if (kind == 0) {
            a[0] = ProbeSchlick({x[0], x[1], x[2]}, {x[3], x[4], x[5]}, x[6], x[7]);
            b[0] = SecondSchlick({x[0], x[1], x[2]}, {x[3], x[4], x[5]}, x[6], x[7]);
            a[1] = a[2] = b[1] = b[2] = 0;
        } else if (kind == 1) {
            const Vec3 va = ProbeLambertSample({x[0], x[1], x[2]}, seedA), vb = SecondDiffuseReflection({x[0], x[1], x[2]}, seedB);
            a[0] = va.x; a[1] = va.y; a[2] = va.z; b[0] = vb.x; b[1] = vb.y; b[2] = vb.z;
        } else if (kind == 2) {
            const Vec3 v = UnpackNormalGlsl(PackNormalGlsl({x[0], x[1], x[2]}));
            a[0] = v.x; a[1] = v.y; a[2] = v.z; b[0] = x[0]; b[1] = x[1]; b[2] = x[2];
        } else {
            const Vec3 nrm{x[0], x[1], x[2]};
            const Vec3 l = ProbeLambertSample(nrm, seedA);
            const float color[3] = {x[3], x[4], x[5]};
            ProbeDiffuseThroughput(nrm, l, color, a);
            b[0] = 1.0f * color[0]; b[1] = 1.0f * color[1]; b[2] = 1.0f * color[2];
        }
    }
    ORC_CATCH
}

// Camera.Set(position, target) for a width x height window; h may be NULL (empty scene, focal = 20).
// Writes FrameState.Camera (the other FrameState fields are left untouched) and the focal distance.
int orc_camera(void* h, const float* pos3, const float* target3, int width, int height, void* frame_state112, float* focal) {
    ORC_TRY
    Camera cam;
    cam.scene = h ? &((OrcScene*)h)->scene : nullptr;
    cam.width = width;
    cam.height = height;
    cam.focal = 1.0f;
    cam.Set({pos3[0], pos3[1], pos3[2]}, {target3[0], target3[1], target3[2]});
    FrameState* fs = (FrameState*)frame_state112;
    cam.SetUniform(*fs);
    if (focal) *focal = cam.focal;
    ORC_CATCH
}

static PathConfig mkcfg(const int32_t* cfg5) {
    PathConfig c;
    c.max_bounces = cfg5[0];
    c.nee = cfg5[1];
    c.rng_mode = cfg5[2];
    c.tile_rank = cfg5[3];
    c.tile_world = cfg5[4];
    c.literal = cfg5[5];
    return c;
}

// Primary rays for thread indices [0, n): o_d n x 6, pixel n (0xFFFFFFFF where the thread emits none)
int orc_generate(const void* rs32, const void* fs112, const int32_t* cfg5, int n, float* o_d, uint32_t* pixel) {
    ORC_TRY
    RenderState rs;
    FrameState fs;
    memcpy(&rs, rs32, 32);
    memcpy(&fs, fs112, 112);
    PathConfig cfg = mkcfg(cfg5);
    for (int i = 0; i < n; i++) {
        Vec3 o{0, 0, 0}, d{0, 0, 0};
        uint32_t px = 0xFFFFFFFFu;
        if (!GenerateRay((uint32_t)i, rs, fs, cfg, &o, &d, &px)) px = 0xFFFFFFFFu;
        o_d[6 * i] = o.x; o_d[6 * i + 1] = o.y; o_d[6 * i + 2] = o.z;
        o_d[6 * i + 3] = d.x; o_d[6 * i + 4] = d.y; o_d[6 * i + 5] = d.z;
        pixel[i] = px;
    }
    ORC_CATCH
}

// Adds `spp` samples (starting at FrameState.Frames) into sum (W*H*4 floats).
// stats8 = {extend_rays, shadow_rays, extend_steps, extend_tris, shadow_steps, shadow_tris, 0, 0}
// traversal: 0 literal C#, 1 canonical.  max_groups > 0 limits the run to the first local pixel groups.
int orc_render(void* h, const void* rs32, const void* fs112, const int32_t* cfg5, int spp, float* sum, uint64_t* stats8, int traversal, int threads, int max_groups, int group_stride) {
    ORC_TRY
    OrcScene* s = (OrcScene*)h;
    RenderState rs;
    FrameState fs;
    memcpy(&rs, rs32, 32);
    memcpy(&fs, fs112, 112);
    PathConfig cfg = mkcfg(cfg5);
    uint32_t gc = (uint32_t)((rs.Width + 7) / 8) * (uint32_t)((rs.Height + 7) / 8);
    int localGroups = (int)((gc + cfg.tile_world - 1 - cfg.tile_rank) / cfg.tile_world);
    if (max_groups > 0 && max_groups < localGroups) localGroups = max_groups;
    int nchunks = std::max(1, threads) * 8;
    std::vector<RenderStats> st(nchunks);
    // interleave chunks over threads for balance (sky rows are cheap)
    parallelFor(std::max(1, threads), threads, [&](int a, int b, int) {
        for (int t = a; t < b; t++) {
            for (int c = t; c < nchunks; c += std::max(1, threads)) {
                uint32_t g0 = (uint32_t)((uint64_t)localGroups * c / nchunks), g1 = (uint32_t)((uint64_t)localGroups * (c + 1) / nchunks);
                RenderSamples(s->scene, s->lights, rs, fs, cfg, spp, sum, g0, g1, &st[c], traversal, (uint32_t)std::max(1, group_stride));
            }
        }
    });
    if (stats8) {
        memset(stats8, 0, 64);
        for (auto& x : st) {
            stats8[0] += x.extend_rays; stats8[1] += x.shadow_rays;
            stats8[2] += x.extend.steps; stats8[3] += x.extend.tris;
            stats8[4] += x.shadow.steps; stats8[5] += x.shadow.tris;
        }
    }
    ORC_CATCH
}

}  // extern "C"
