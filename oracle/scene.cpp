// CPU ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// Restates PathTracer/Scene.cs, the scene recipe of PathTracer/Game.cs:57-136
// and PathTracer/Camera.cs:39-61,89-105.
#include "oracle.hpp"
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <limits>
#include <stdexcept>

namespace oracle {

static const float INF = std::numeric_limits<float>::infinity();

// Scene.cs:28-33
void Scene::Clear() {
    materials.clear();
    accelerator.Clear();
}

// Scene.cs:35-45
Intersection Scene::Intersect(const Ray& ray) const {
    Intersection i;
    i.Distance = INF;
    i.Point = {0, 0, 0};
    i.Tri = -1;
    i.U = i.V = 0;
    accelerator.Intersect(ray, i);
    return i;
}

// Scene.cs:192-204.  Dictionary<Material,int> keyed by value equality of the
// whole struct (GetHashCode only narrows the bucket, Material.cs:16-19).
int Scene::GetMaterialIndex(const Material& m) {
    for (size_t i = 0; i < materials.size(); i++) {
        const Material& o = materials[i];
        bool eq = o.Type == m.Type && o.Index == m.Index && o.Roughness == m.Roughness;
        for (int k = 0; k < 4 && eq; k++) eq = o.Color[k] == m.Color[k] && o.Emissive[k] == m.Emissive[k];
        if (eq) return (int)i;
    }
    materials.push_back(m);
    return (int)materials.size() - 1;
}

static Mesh loadMesh(const std::string& source, const std::string& name) {
    // `source` is a directory holding <name>.obj, or a mesh blob file
    if (!source.empty() && source.back() == '/') return Mesh::LoadFromFile(source + name + ".obj");
    std::ifstream f(source, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open mesh source " + source);
    std::stringstream ss;
    ss << f.rdbuf();
    return Mesh::LoadFromBlob(ss.str(), name);
}

static Vertex mkVertex(Vec3 p, Vec3 n) {
    Vertex v;
    memset(&v, 0, sizeof(v));
    v.Position = p;
    v.Normal = n;
    return v;
}

static Triangle mkTriangle(const Vertex& a, const Vertex& b, const Vertex& c, int material) {
    Triangle t;
    memset(&t, 0, sizeof(t));
    t.V1 = a; t.V2 = b; t.V3 = c;
    t.Material = material;
    return t;
}

// Scene.cs:159-174
static void LoadGroup(Scene& s, const Mesh& mesh, const Group& group, int material) {
    std::vector<GeoVertex> tri;
    for (const Face& face : group.faces) {
        tri.clear();
        mesh.Triangulate(face, tri);
        for (size_t i = 0; i + 2 < tri.size(); i += 3)
            s.accelerator.AddTriangle(mkTriangle(mkVertex(tri[i].Position, tri[i].Normal), mkVertex(tri[i + 1].Position, tri[i + 1].Normal),
                                                 mkVertex(tri[i + 2].Position, tri[i + 2].Normal), material));
    }
}

// Scene.cs:122-141
void Scene::AddMesh(const std::string& name, bool normalized, const Matrix4x4& transform) {
    Mesh mesh = loadMesh(meshSource, name);
    mesh.Normalized = normalized;
    mesh.TransformM = transform;
    for (const Group& group : mesh.groups) {
        Material m;
        memset(&m, 0, sizeof(m));
        m.Color[0] = group.material.Diffuse.x; m.Color[1] = group.material.Diffuse.y; m.Color[2] = group.material.Diffuse.z;
        m.Emissive[0] = group.material.Emissive.x; m.Emissive[1] = group.material.Emissive.y; m.Emissive[2] = group.material.Emissive.z;
        m.Type = Diffuse;
        int material = GetMaterialIndex(m);
        LoadGroup(*this, mesh, group, material);
    }
}

// Scene.cs:143-157
void Scene::AddMesh(const std::string& name, bool normalized, const Matrix4x4& transform, const Material& material) {
    Mesh mesh = loadMesh(meshSource, name);
    mesh.Normalized = normalized;
    mesh.TransformM = transform;
    int m = GetMaterialIndex(material);
    for (const Group& group : mesh.groups) LoadGroup(*this, mesh, group, m);
}

// Scene.cs:87-120
void Scene::GenerateTiledFloor(const Material& a, const Material& b) {
    int i = GetMaterialIndex(a);
    int j = GetMaterialIndex(b);
    Vec3 up{0, 1, 0};
    for (int x = -100; x < 99; x++) {
        for (int y = -100; y < 99; y++) {
            int mat = ((x & 1) ^ (y & 1)) == 0 ? i : j;
            Vertex v4 = mkVertex({(float)x, 0, (float)y}, up);
            Vertex v3 = mkVertex({(float)(x + 1), 0, (float)y}, up);
            Vertex v2 = mkVertex({(float)(x + 1), 0, (float)(y + 1)}, up);
            Vertex v1 = mkVertex({(float)x, 0, (float)(y + 1)}, up);
            accelerator.AddTriangle(mkTriangle(v1, v2, v3, mat));
            accelerator.AddTriangle(mkTriangle(v3, v4, v1, mat));
        }
    }
}

// Game.cs:57-64
static void RandomVector(DotNetRandom& rng, float out[4]) {
    out[0] = (float)rng.NextDouble();
    out[1] = (float)rng.NextDouble();
    out[2] = (float)rng.NextDouble();
    out[3] = 0;
}

// Game.cs:66-136
void Scene::InitializeScene(int seed) {
    Material tile1;
    memset(&tile1, 0, sizeof(tile1));
    tile1.Color[0] = 0.7f; tile1.Color[1] = 0.7f; tile1.Color[2] = 0.9f;
    tile1.Type = Metal;
    tile1.Roughness = 0.1f;
    Material tile2 = tile1;
    tile2.Roughness = 0.16f;

    AddMesh("light", false, MatIdentity());
    GenerateTiledFloor(tile1, tile2);

    DotNetRandom rng(seed);
    const float PI = 3.14159274f;   // MathF.PI

    for (int i = 0; i < 60; i++) {
        Material material;
        memset(&material, 0, sizeof(material));
        RandomVector(rng, material.Color);
        if (rng.Next(100) > 90) {
            RandomVector(rng, material.Emissive);
            for (int k = 0; k < 4; k++) material.Emissive[k] = material.Emissive[k] * 8.0f;
        }
        material.Index = 1.4f;
        material.Type = rng.Next(4);
        material.Roughness = (float)rng.NextDouble();

        float x = (float)rng.NextDouble() * 12 - 6;
        float y = (float)rng.NextDouble() * 5 + 1;
        float z = (float)rng.NextDouble() * 12 - 6;

        float rota = (float)rng.NextDouble() * PI;
        float rotb = (float)rng.NextDouble() * PI;
        float rotc = (float)rng.NextDouble() * PI;

        float scale = (float)rng.NextDouble() + 0.5f;

        Matrix4x4 matrix = MatIdentity();
        matrix = MatMul(matrix, MatRotationX(rota));
        matrix = MatMul(matrix, MatRotationY(rotb));
        matrix = MatMul(matrix, MatRotationZ(rotc));
        matrix = MatMul(matrix, MatTranslation(x, y, z));
        matrix = MatMul(matrix, MatScale(scale));

        switch (rng.Next(5)) {
            case 0: case 1: AddMesh("cube", true, matrix, material); break;
            case 2: case 3: AddMesh("sphere", true, matrix, material); break;
            case 4: AddMesh("bunny", true, matrix, material); break;
        }
    }
}

// Synthetic random triangle soup (not in the reference; the incoherent-traversal
// stress input).  xorshift32 (generate.glsl:10-21) seeded with `seed`;
// per triangle 12 draws: centre in [-50,50)^3, three vertex offsets in
// [-0.5,0.5)^3; geometric normal on all three vertices; material 0 = diffuse
// grey 0.7, material 1 = an emissive 60x60 quad at y = 60 added first.
void Scene::GenerateSoup(int nTriangles, uint32_t seed) {
    Material grey;
    memset(&grey, 0, sizeof(grey));
    grey.Color[0] = grey.Color[1] = grey.Color[2] = 0.7f;
    grey.Type = Diffuse;
    Material lamp;
    memset(&lamp, 0, sizeof(lamp));
    lamp.Color[0] = lamp.Color[1] = lamp.Color[2] = 1.0f;
    lamp.Emissive[0] = lamp.Emissive[1] = lamp.Emissive[2] = 20.0f;
    lamp.Type = Diffuse;
    int g = GetMaterialIndex(grey);
    int l = GetMaterialIndex(lamp);

    Vec3 down{0, -1, 0};
    Vertex q1 = mkVertex({-30, 60, -30}, down), q2 = mkVertex({-30, 60, 30}, down), q3 = mkVertex({30, 60, 30}, down), q4 = mkVertex({30, 60, -30}, down);
    accelerator.AddTriangle(mkTriangle(q1, q2, q3, l));
    accelerator.AddTriangle(mkTriangle(q3, q4, q1, l));

    uint32_t s = seed;
    auto rnd = [&s]() {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        return (float)s * 2.3283064365386963e-10f;   // 2^-32
    };
    for (int i = 0; i < nTriangles - 2; i++) {
        Vec3 c;
        c.x = rnd() * 100.0f - 50.0f; c.y = rnd() * 100.0f - 50.0f; c.z = rnd() * 100.0f - 50.0f;
        Vec3 p[3];
        for (int k = 0; k < 3; k++) {
            Vec3 e;
            e.x = rnd() - 0.5f; e.y = rnd() - 0.5f; e.z = rnd() - 0.5f;
            p[k] = c + e;
        }
        Vec3 n = Normalize(Cross(p[1] - p[0], p[2] - p[0]));
        accelerator.AddTriangle(mkTriangle(mkVertex(p[0], n), mkVertex(p[1], n), mkVertex(p[2], n), g));
    }
}

// ------------------------------------------------------------------ Camera
// Camera.cs:39-44
void Camera::Set(Vec3 pos, Vec3 tgt) {
    position = pos;
    target = tgt;
    Update();
}

// Camera.cs:89-95
float Camera::GetFocalDistance() const {
    Ray r = MakeRay(position, forward);
    Intersection i;
    i.Distance = INF;
    i.Tri = -1;
    if (scene) i = scene->Intersect(r);
    return MathFMin(i.Distance, 20);
}

// Camera.cs:46-61
void Camera::Update() {
    forward = Normalize(target - position);
    right = Normalize(Cross(Vec3{0, 1, 0}, forward));
    up = Normalize(Cross(forward, right));
    focal = GetFocalDistance();

    float ar = width / (float)height;

    Vec3 c = position + focal * forward;
    p1 = c - 0.5f * focal * ar * right + 0.5f * focal * up;
    p2 = c + 0.5f * focal * ar * right + 0.5f * focal * up;
    p3 = c - 0.5f * focal * ar * right - 0.5f * focal * up;

    target = position + forward;
}

// Camera.cs:97-105
void Camera::SetUniform(FrameState& fs) const {
    fs.Camera.P1 = p1;
    fs.Camera.P2 = p2;
    fs.Camera.P3 = p3;
    fs.Camera.Position = position;
    fs.Camera.Right = right;
    fs.Camera.Up = up;
}

}  // namespace oracle
