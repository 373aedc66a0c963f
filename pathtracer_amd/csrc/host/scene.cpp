// libpt_host.so: host mirror of PathTracer/Scene.cs, Library/Geometry/Mesh.cs + Face.cs, the
// scene recipe of Game.cs:57-136 and the Camera math of Camera.cs:46-61,89-105, behind the C
// ABI of include/pt_host.h.  It feeds libpt_hip.so on machines without the C# host.
//
// Arithmetic follows the .NET Core 3.1 BCL conventions the reference relies on (row-vector
// Matrix4x4, Vector3.Dot = (x*x' + y*y') + z*z', true division in Normalize, the seeded
// Knuth subtractive System.Random); compile with -ffp-contract=off.
#include "pt_host.h"
#include "accelerator.hpp"
#include <algorithm>
#include <cmath>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

using namespace pth;

namespace {

thread_local std::string g_error;
const float kInf = std::numeric_limits<float>::infinity();

// ------------------------------------------------------------------ vectors
inline Vec3 sub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 add(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 scale(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 normalize(Vec3 a) { const float l = sqrtf(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }
inline float maxNaN(float x, float y) { if (x > y || x != x) return x; if (x == y) return std::signbit(x) ? y : x; return y; }
inline float minNaN(float x, float y) { if (x < y || x != x) return x; if (x == y) return std::signbit(x) ? x : y; return y; }

// System.Numerics.Matrix4x4, row-major, row-vector convention
struct Mat4 { float m[16]; };
Mat4 identity() { Mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
// operator*: each element (a1*b1 + a2*b2) + (a3*b3 + a4*b4), the SSE form of .NET Core 3.x
Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[4 * i + j] = (a.m[4 * i] * b.m[j] + a.m[4 * i + 1] * b.m[4 + j]) + (a.m[4 * i + 2] * b.m[8 + j] + a.m[4 * i + 3] * b.m[12 + j]);
    return r;
}
Mat4 rotationX(float a) { Mat4 r = identity(); const float c = cosf(a), s = sinf(a); r.m[5] = c; r.m[6] = s; r.m[9] = -s; r.m[10] = c; return r; }
Mat4 rotationY(float a) { Mat4 r = identity(); const float c = cosf(a), s = sinf(a); r.m[0] = c; r.m[2] = -s; r.m[8] = s; r.m[10] = c; return r; }
Mat4 rotationZ(float a) { Mat4 r = identity(); const float c = cosf(a), s = sinf(a); r.m[0] = c; r.m[1] = s; r.m[4] = -s; r.m[5] = c; return r; }
Mat4 translation(float x, float y, float z) { Mat4 r = identity(); r.m[12] = x; r.m[13] = y; r.m[14] = z; return r; }
Mat4 uniformScale(float s) { Mat4 r = identity(); r.m[0] = r.m[5] = r.m[10] = s; return r; }
// Vector3.Transform / TransformNormal
Vec3 transformPoint(Vec3 p, const Mat4& t) {
    return {p.x * t.m[0] + p.y * t.m[4] + p.z * t.m[8] + t.m[12], p.x * t.m[1] + p.y * t.m[5] + p.z * t.m[9] + t.m[13], p.x * t.m[2] + p.y * t.m[6] + p.z * t.m[10] + t.m[14]};
}
Vec3 transformNormal(Vec3 n, const Mat4& t) {
    return {n.x * t.m[0] + n.y * t.m[4] + n.z * t.m[8], n.x * t.m[1] + n.y * t.m[5] + n.z * t.m[9], n.x * t.m[2] + n.y * t.m[6] + n.z * t.m[10]};
}

// ------------------------------------------------------------ System.Random
// Seeded constructor of the .NET Core 3.1 BCL: Knuth's subtractive generator.
class NetRandom {
public:
    explicit NetRandom(int32_t seed) {
        const int32_t big = INT_MAX;
        int32_t mj = 161803398 - (seed == INT_MIN ? INT_MAX : std::abs(seed));
        std::memset(table, 0, sizeof(table));
        table[55] = mj;
        int32_t mk = 1;
        for (int i = 1, ii = 0; i < 55; i++) {
            ii = (ii + 21) % 55;
            table[ii] = mk;
            mk = mj - mk;
            if (mk < 0) mk += big;
            mj = table[ii];
        }
        for (int round = 0; round < 4; round++)
            for (int i = 1; i <= 55; i++) {
                table[i] = (int32_t)((uint32_t)table[i] - (uint32_t)table[1 + (i + 30) % 55]);
                if (table[i] < 0) table[i] += big;
            }
        a = 0;
        b = 21;
    }
    int32_t sample() {
        if (++a >= 56) a = 1;
        if (++b >= 56) b = 1;
        int32_t r = table[a] - table[b];
        if (r == INT_MAX) r--;
        if (r < 0) r += INT_MAX;
        table[a] = r;
        return r;
    }
    double nextDouble() { return sample() * (1.0 / INT_MAX); }
    int32_t next(int32_t maxValue) { return (int32_t)(nextDouble() * maxValue); }
private:
    int32_t table[56];
    int a, b;
};

// ---------------------------------------------------------------------- mesh
struct ObjMaterial { Vec3 diffuse{1, 1, 1}, emissive{0, 0, 0}; };
struct Corner { int p, t, n; };                       // Frame.cs: 0-based, -1 = absent
struct MeshGroup { ObjMaterial material; std::vector<std::vector<Corner>> faces; };
struct MeshData {
    std::vector<Vec3> positions, normals;
    int texcoords = 0;
    std::vector<MeshGroup> groups;
    Vec3 lo{kInf, kInf, kInf}, hi{-kInf, -kInf, -kInf};
    void addPosition(Vec3 p) {
        positions.push_back(p);
        lo = {lo.x < p.x ? lo.x : p.x, lo.y < p.y ? lo.y : p.y, lo.z < p.z ? lo.z : p.z};   // Vector3.Min(min, p)
        hi = {hi.x > p.x ? hi.x : p.x, hi.y > p.y ? hi.y : p.y, hi.z > p.z ? hi.z : p.z};
    }
};

bool isSpace(char c) { return c == ' ' || (c >= 9 && c <= 13); }
std::string strip(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isSpace(s[a])) a++;
    while (b > a && isSpace(s[b - 1])) b--;
    return s.substr(a, b - a);
}
std::vector<std::string> fields(const std::string& s) {
    std::vector<std::string> out;
    std::string cur;
    for (char c : s) {
        if (c == ' ' || c == '\t') { if (!cur.empty()) { out.push_back(cur); cur.clear(); } }
        else cur.push_back(c);
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}
// Mesh.ReadObjLine, Mesh.cs:240-259: key = text before the first blank, arg = the rest
void keyAndArgument(const std::string& line, std::string& key, std::string& arg) {
    key.clear();
    arg.clear();
    std::string s = line;
    for (char& c : s) if (c == '\t') c = ' ';
    s = strip(s);
    const size_t sp = s.find(' ');
    if (sp == std::string::npos) return;
    key = strip(s.substr(0, sp));
    arg = strip(s.substr(sp + 1));
}
Vec3 parseVec3(const std::string& arg) {
    const auto f = fields(strip(arg));
    if (f.size() < 3) throw std::runtime_error("Cannot parse .obj file");   // Mesh.cs:280-283
    return {strtof(f[0].c_str(), nullptr), strtof(f[1].c_str(), nullptr), strtof(f[2].c_str(), nullptr)};
}
std::vector<std::string> textLines(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    std::string text = ss.str();
    if (text.compare(0, 3, "\xEF\xBB\xBF") == 0) text.erase(0, 3);   // StreamReader eats one byte-order mark
    std::vector<std::string> lines;
    std::string cur;
    for (char c : text) {
        if (c == '\n') { if (!cur.empty() && cur.back() == '\r') cur.pop_back(); lines.push_back(cur); cur.clear(); }
        else cur.push_back(c);
    }
    if (!cur.empty()) lines.push_back(cur);
    return lines;
}

// Mesh.ReadMtlFile, Mesh.cs:181-233 (only the fields the scene uses; Ka also lands in emissive, as there)
std::map<std::string, ObjMaterial> readMtl(const std::string& path) {
    std::map<std::string, ObjMaterial> out;
    std::string name, key, arg;
    ObjMaterial cur;
    for (const auto& line : textLines(path)) {
        keyAndArgument(line, key, arg);
        if (key == "newmtl") { out[name] = cur; name = arg; cur = ObjMaterial(); }
        else if (key == "Ka" || key == "Ke") cur.emissive = parseVec3(arg);
        else if (key == "Kd") cur.diffuse = parseVec3(arg);
    }
    out[name] = cur;
    return out;
}

// Mesh.LoadFromFile, Mesh.cs:72-179
MeshData readObj(const std::string& path) {
    MeshData mesh;
    std::map<std::string, ObjMaterial> library;
    ObjMaterial current;
    std::vector<std::vector<Corner>> pending;
    const std::string dir = path.substr(0, path.find_last_of('/') + 1);
    auto closeGroup = [&]() {
        if (pending.empty()) return;
        mesh.groups.push_back(MeshGroup{current, pending});
        pending.clear();
    };
    std::string key, arg;
    for (const auto& line : textLines(path)) {
        keyAndArgument(line, key, arg);
        if (key == "v") mesh.addPosition(parseVec3(arg));
        else if (key == "vt") mesh.texcoords++;
        else if (key == "vn") mesh.normals.push_back(parseVec3(arg));
        else if (key == "f") {
            std::vector<Corner> face;
            for (const auto& part : fields(strip(arg))) {
                int idx[3] = {0, 0, 0};
                int k = 0;
                size_t from = 0;
                std::vector<std::string> pieces;
                for (;;) {
                    const size_t slash = part.find('/', from);
                    pieces.push_back(part.substr(from, slash == std::string::npos ? std::string::npos : slash - from));
                    if (slash == std::string::npos) break;
                    from = slash + 1;
                }
                if (pieces.size() == 1) idx[0] = atoi(pieces[0].c_str());
                else if (pieces.size() == 2) { idx[0] = atoi(pieces[0].c_str()); idx[1] = atoi(pieces[1].c_str()); }
                else if (pieces.size() == 3) { idx[0] = atoi(pieces[0].c_str()); if (!pieces[1].empty()) idx[1] = atoi(pieces[1].c_str()); idx[2] = atoi(pieces[2].c_str()); }
                (void)k;
                // Mesh.cs:116-118: negative = relative to the end, positive = 1-based, 0 = absent (-1)
                Corner c;
                c.p = idx[0] < 0 ? idx[0] + (int)mesh.positions.size() : idx[0] - 1;
                c.t = idx[1] < 0 ? idx[1] + mesh.texcoords : idx[1] - 1;
                c.n = idx[2] < 0 ? idx[2] + (int)mesh.normals.size() : idx[2] - 1;
                face.push_back(c);
            }
            pending.push_back(face);
        } else if (key == "mtllib") {
            for (auto& kv : readMtl(dir + arg)) library[kv.first] = kv.second;
        } else if (key == "g") {
            closeGroup();
        } else if (key == "usemtl") {
            closeGroup();
            auto it = library.find(arg);
            current = it != library.end() ? it->second : ObjMaterial();
        }
    }
    closeGroup();
    return mesh;
}

// Parsed-mesh blob: "PTMESH1\0", u32 count, per mesh { str name, u32 nP, 3f each, u32 nN, 3f each, u32 nT,
// u32 nGroups, per group { str name, 14 f32 material (ambient, diffuse, specular, emissive, Ns, Tr),
// u32 nFaces, per face { u32 nCorners, 3 i32 each } } }
MeshData readBlob(const std::string& path, const std::string& want) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open mesh blob " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string b = ss.str();
    size_t at = 8;
    if (b.size() < 12 || std::memcmp(b.data(), "PTMESH1\0", 8) != 0) throw std::runtime_error(path + " is not a mesh blob");
    auto u32 = [&]() { if (at + 4 > b.size()) throw std::runtime_error("mesh blob truncated"); uint32_t v; std::memcpy(&v, b.data() + at, 4); at += 4; return v; };
    auto f32 = [&]() { const uint32_t v = u32(); float f; std::memcpy(&f, &v, 4); return f; };
    auto str = [&]() { const uint32_t n = u32(); if (at + n > b.size()) throw std::runtime_error("mesh blob truncated"); std::string s = b.substr(at, n); at += n; return s; };
    const uint32_t count = u32();
    for (uint32_t m = 0; m < count; m++) {
        MeshData mesh;
        const std::string name = str();
        for (uint32_t i = 0, n = u32(); i < n; i++) { Vec3 p; p.x = f32(); p.y = f32(); p.z = f32(); mesh.addPosition(p); }
        for (uint32_t i = 0, n = u32(); i < n; i++) { Vec3 p; p.x = f32(); p.y = f32(); p.z = f32(); mesh.normals.push_back(p); }
        mesh.texcoords = (int)u32();
        for (uint32_t g = 0, ng = u32(); g < ng; g++) {
            MeshGroup group;
            (void)str();
            float mat[14];
            for (float& v : mat) v = f32();
            group.material.diffuse = {mat[3], mat[4], mat[5]};
            group.material.emissive = {mat[9], mat[10], mat[11]};
            for (uint32_t f = 0, nf = u32(); f < nf; f++) {
                std::vector<Corner> face;
                for (uint32_t k = 0, nk = u32(); k < nk; k++) { Corner c; c.p = (int)u32(); c.t = (int)u32(); c.n = (int)u32(); face.push_back(c); }
                group.faces.push_back(face);
            }
            mesh.groups.push_back(group);
        }
        if (name == want) return mesh;
    }
    throw std::runtime_error("mesh '" + want + "' is not in " + path);
}

Vertex vertexOf(Vec3 p, Vec3 n) { Vertex v; std::memset(&v, 0, sizeof(v)); v.Position = p; v.Normal = n; return v; }
Triangle triangleOf(const Vertex& a, const Vertex& b, const Vertex& c, int material) {
    Triangle t;
    std::memset(&t, 0, sizeof(t));
    t.V1 = a; t.V2 = b; t.V3 = c; t.Material = material;
    return t;
}
bool sameMaterial(const Material& a, const Material& b) {
    for (int k = 0; k < 4; k++) if (a.Color[k] != b.Color[k] || a.Emissive[k] != b.Emissive[k]) return false;
    return a.Type == b.Type && a.Index == b.Index && a.Roughness == b.Roughness;
}

}  // namespace

struct pth_scene {
    std::string meshSource;
    std::vector<Material> materials;
    Accelerator accelerator;
    std::map<std::string, MeshData> cache;

    // What the scene is made of, for a host that wants the GPU to walk instances instead of the flattened triangles: every
    // addMesh call is one instance of the mesh (name, normalized) under its matrix.  The reference has no such notion
    // (Scene.AddMesh flattens, Scene.cs:122-174); nothing here changes what AddTriangle receives.
    struct MeshRef { std::string name; bool normalized; };
    struct InstanceRef { int mesh; int first; int count; Mat4 transform; };
    std::vector<MeshRef> meshRefs;
    std::vector<InstanceRef> instanceRefs;

    const MeshData& mesh(const std::string& name) {
        auto it = cache.find(name);
        if (it != cache.end()) return it->second;
        MeshData m = (!meshSource.empty() && meshSource.back() == '/') ? readObj(meshSource + name + ".obj") : readBlob(meshSource, name);
        return cache.emplace(name, std::move(m)).first->second;
    }

    // Scene.GetMaterialIndex, Scene.cs:192-204 (Dictionary keyed by value equality)
    int materialIndex(const Material& m) {
        for (size_t i = 0; i < materials.size(); i++) if (sameMaterial(materials[i], m)) return (int)i;
        materials.push_back(m);
        return (int)materials.size() - 1;
    }

    // Scene.LoadGroup + Face.Triangulate + Mesh.GetPosition/GetNormal (Scene.cs:159-174, Face.cs:23-65, Mesh.cs:50-70)
    // sink: where the triangles go instead of the accelerator (the object-space copy of a mesh for instancing)
    void addGroup(const MeshData& mesh, const MeshGroup& group, bool normalized, const Mat4& transform, int material, std::vector<Triangle>* sink = nullptr) {
        const Vec3 extent = sub(mesh.hi, mesh.lo);
        const float largest = maxNaN(maxNaN(extent.x, extent.y), extent.z);
        auto corner = [&](const Corner& c) {
            Vec3 p{0, 0, 0}, n{0, 0, 0};
            if (c.p >= 0) {
                p = mesh.positions[c.p];
                if (normalized) { const Vec3 d = sub(p, mesh.lo); p = {d.x / largest, d.y / largest, d.z / largest}; }
                p = transformPoint(p, transform);
            }
            if (c.n >= 0) n = normalize(transformNormal(mesh.normals[c.n], transform));
            return vertexOf(p, n);
        };
        for (const auto& face : group.faces) {
            // fan; a, b, c are copies that persist across the fan, so a flat normal written in one
            // step stays on a (and on b, the previous c) for the next step
            Vertex a = corner(face[0]), b = corner(face[1]);
            for (size_t i = 2; i < face.size(); i++) {
                Vertex c = corner(face[i]);
                const Vec3 sum = add(add(a.Normal, b.Normal), c.Normal);
                if (dot(sum, sum) < 0.5f) {
                    const Vec3 flat = normalize(cross(sub(a.Position, c.Position), sub(b.Position, c.Position)));
                    a.Normal = b.Normal = c.Normal = flat;
                }
                if (sink) sink->push_back(triangleOf(a, b, c, material));
                else accelerator.AddTriangle(triangleOf(a, b, c, material));
                b = c;
            }
        }
    }

    void addMesh(const std::string& name, bool normalized, const Mat4& transform, const Material* material) {
        const MeshData& m = mesh(name);
        int ref = -1;
        for (size_t i = 0; i < meshRefs.size(); i++) if (meshRefs[i].name == name && meshRefs[i].normalized == normalized) ref = (int)i;
        if (ref < 0) { meshRefs.push_back({name, normalized}); ref = (int)meshRefs.size() - 1; }
        const int firstTriangle = accelerator.Count();
        struct Note {   // records the instance once its triangles are in, on every way out
            pth_scene* s; int ref, first; Mat4 t;
            ~Note() { s->instanceRefs.push_back({ref, first, s->accelerator.Count() - first, t}); }
        } note{this, ref, firstTriangle, transform};
        if (material) {
            const int id = materialIndex(*material);
            for (const auto& g : m.groups) addGroup(m, g, normalized, transform, id);
        } else {
            for (const auto& g : m.groups) {
                Material mat;
                std::memset(&mat, 0, sizeof(mat));
                mat.Color[0] = g.material.diffuse.x; mat.Color[1] = g.material.diffuse.y; mat.Color[2] = g.material.diffuse.z;
                mat.Emissive[0] = g.material.emissive.x; mat.Emissive[1] = g.material.emissive.y; mat.Emissive[2] = g.material.emissive.z;
                addGroup(m, g, normalized, transform, materialIndex(mat));
            }
        }
    }

    // Scene.GenerateTiledFoloor, Scene.cs:87-120
    void tiledFloor(const Material& a, const Material& b) {
        const int ia = materialIndex(a), ib = materialIndex(b);
        const Vec3 up{0, 1, 0};
        for (int x = -100; x < 99; x++)
            for (int y = -100; y < 99; y++) {
                const int mat = ((x & 1) ^ (y & 1)) == 0 ? ia : ib;
                const Vertex v4 = vertexOf({(float)x, 0, (float)y}, up), v3 = vertexOf({(float)(x + 1), 0, (float)y}, up);
                const Vertex v2 = vertexOf({(float)(x + 1), 0, (float)(y + 1)}, up), v1 = vertexOf({(float)x, 0, (float)(y + 1)}, up);
                accelerator.AddTriangle(triangleOf(v1, v2, v3, mat));
                accelerator.AddTriangle(triangleOf(v3, v4, v1, mat));
            }
    }

    // Game.InitializeScene, Game.cs:66-136
    // `extra` further objects, all bunnies, continue the same random stream (a ~260k-triangle variant)
    void bundledScene(int seed, int extra = 0) {
        Material tile;
        std::memset(&tile, 0, sizeof(tile));
        tile.Color[0] = 0.7f; tile.Color[1] = 0.7f; tile.Color[2] = 0.9f;
        tile.Type = 3;
        Material tile1 = tile, tile2 = tile;
        tile1.Roughness = 0.1f;
        tile2.Roughness = 0.16f;
        addMesh("light", false, identity(), nullptr);
        tiledFloor(tile1, tile2);
        NetRandom rng(seed);
        const float pi = 3.14159274f;
        static const char* const shapes[5] = {"cube", "cube", "sphere", "sphere", "bunny"};
        for (int i = 0; i < 60 + extra; i++) {
            Material m;
            std::memset(&m, 0, sizeof(m));
            for (int k = 0; k < 3; k++) m.Color[k] = (float)rng.nextDouble();
            if (rng.next(100) > 90)
                for (int k = 0; k < 4; k++) m.Emissive[k] = (k < 3 ? (float)rng.nextDouble() : 0.0f) * 8.0f;
            m.Index = 1.4f;
            m.Type = rng.next(4);
            m.Roughness = (float)rng.nextDouble();
            const float x = (float)rng.nextDouble() * 12 - 6;
            const float y = (float)rng.nextDouble() * 5 + 1;
            const float z = (float)rng.nextDouble() * 12 - 6;
            const float ra = (float)rng.nextDouble() * pi, rb = (float)rng.nextDouble() * pi, rc = (float)rng.nextDouble() * pi;
            const float s = (float)rng.nextDouble() + 0.5f;
            Mat4 t = identity();
            t = mul(t, rotationX(ra));
            t = mul(t, rotationY(rb));
            t = mul(t, rotationZ(rc));
            t = mul(t, translation(x, y, z));
            t = mul(t, uniformScale(s));
            const char* shape = shapes[rng.next(5)];
            addMesh(i < 60 ? shape : "bunny", true, t, &m);
        }
    }

    // Synthetic many-instance scene (no reference counterpart; the reference has no instancing, so
    // instances are flattened): `count` bunnies on a cubic grid of pitch 2.5 with random rotations,
    // scales in [0.8, 1.8) and materials drawn like the bundled recipe's, plus the light and a floor.
    // The scene as meshes and instances: every triangle of the flattened array belongs to exactly one instance, in order.
    // A mesh's triangles are what addMesh would insert under the identity matrix (Vector3.Transform by the identity returns its
    // argument bit for bit), so Transform(object vertex, instance matrix) reproduces the flattened vertex exactly.  Triangles that
    // did not come from addMesh (the tiled floor, a soup, AddTriangle calls) form runs that become meshes of their own, instanced once
    // under the identity.
    struct Instancing {
        std::vector<Triangle> meshTriangles;
        std::vector<int32_t> meshFirst, meshCount;
        struct Inst { float transform[16]; int32_t mesh, first; };
        std::vector<Inst> instances;
    };
    Instancing instancing() {
        Instancing out;
        std::vector<int> meshSlot(meshRefs.size(), -1);
        const std::vector<Triangle>& flat = accelerator.References();
        auto addLoose = [&](int first, int count) {
            out.meshFirst.push_back((int32_t)out.meshTriangles.size());
            out.meshCount.push_back(count);
            out.meshTriangles.insert(out.meshTriangles.end(), flat.begin() + first, flat.begin() + first + count);
            Instancing::Inst in;
            const Mat4 id = identity();
            std::memcpy(in.transform, id.m, 64);
            in.mesh = (int32_t)out.meshFirst.size() - 1;
            in.first = first;
            out.instances.push_back(in);
        };
        int cursor = 0;
        for (const InstanceRef& r : instanceRefs) {
            if (r.count == 0) continue;
            if (r.first < cursor || r.first + r.count > (int)flat.size()) throw std::runtime_error("the scene's instance records do not match its triangles (cleared or loaded since?)");
            if (r.first > cursor) addLoose(cursor, r.first - cursor);
            if (meshSlot[r.mesh] < 0) {
                const MeshData& m = mesh(meshRefs[r.mesh].name);
                out.meshFirst.push_back((int32_t)out.meshTriangles.size());
                for (const auto& g : m.groups) addGroup(m, g, meshRefs[r.mesh].normalized, identity(), 0, &out.meshTriangles);
                out.meshCount.push_back((int32_t)out.meshTriangles.size() - out.meshFirst.back());
                meshSlot[r.mesh] = (int)out.meshFirst.size() - 1;
            }
            if (out.meshCount[meshSlot[r.mesh]] != r.count) throw std::runtime_error("an instance's triangle count differs from its mesh's");
            Instancing::Inst in;
            std::memcpy(in.transform, r.transform.m, 64);
            in.mesh = meshSlot[r.mesh];
            in.first = r.first;
            out.instances.push_back(in);
            cursor = r.first + r.count;
        }
        if (cursor < (int)flat.size()) addLoose(cursor, (int)flat.size() - cursor);
        return out;
    }

    void instancedBunnies(int count, int seed) {
        addMesh("light", false, identity(), nullptr);
        NetRandom rng(seed);
        const float pi = 3.14159274f;
        int side = 1;
        while (side * side * side < count) side++;
        for (int i = 0; i < count; i++) {
            Material m;
            std::memset(&m, 0, sizeof(m));
            for (int k = 0; k < 3; k++) m.Color[k] = (float)rng.nextDouble();
            if (rng.next(100) > 97)
                for (int k = 0; k < 3; k++) m.Emissive[k] = (float)rng.nextDouble() * 8.0f;
            m.Index = 1.4f;
            m.Type = rng.next(4);
            m.Roughness = (float)rng.nextDouble();
            const int gx = i % side, gy = (i / side) % side, gz = i / (side * side);
            const float ra = (float)rng.nextDouble() * pi, rb = (float)rng.nextDouble() * pi, rc = (float)rng.nextDouble() * pi;
            const float s = (float)rng.nextDouble() + 0.8f;
            Mat4 t = uniformScale(s);
            t = mul(t, rotationX(ra));
            t = mul(t, rotationY(rb));
            t = mul(t, rotationZ(rc));
            t = mul(t, translation((gx - 0.5f * side) * 2.5f, gy * 2.5f + 0.5f, (gz - 0.5f * side) * 2.5f));
            addMesh("bunny", true, t, &m);
        }
    }

    // Random triangle soup: xorshift32 stream, 12 draws per triangle (centre, three offsets)
    void soup(int n, uint32_t seed) {
        Material grey, lamp;
        std::memset(&grey, 0, sizeof(grey));
        std::memset(&lamp, 0, sizeof(lamp));
        for (int k = 0; k < 3; k++) { grey.Color[k] = 0.7f; lamp.Color[k] = 1.0f; lamp.Emissive[k] = 20.0f; }
        const int g = materialIndex(grey), l = materialIndex(lamp);
        const Vec3 down{0, -1, 0};
        const Vertex q1 = vertexOf({-30, 60, -30}, down), q2 = vertexOf({-30, 60, 30}, down), q3 = vertexOf({30, 60, 30}, down), q4 = vertexOf({30, 60, -30}, down);
        accelerator.AddTriangle(triangleOf(q1, q2, q3, l));
        accelerator.AddTriangle(triangleOf(q3, q4, q1, l));
        uint32_t state = seed;
        auto uniform = [&state]() { state ^= state << 13; state ^= state >> 17; state ^= state << 5; return (float)state * 2.3283064365386963e-10f; };
        for (int i = 0; i < n - 2; i++) {
            Vec3 c;
            c.x = uniform() * 100.0f - 50.0f; c.y = uniform() * 100.0f - 50.0f; c.z = uniform() * 100.0f - 50.0f;
            Vec3 p[3];
            for (auto& q : p) { Vec3 e; e.x = uniform() - 0.5f; e.y = uniform() - 0.5f; e.z = uniform() - 0.5f; q = add(c, e); }
            const Vec3 nrm = normalize(cross(sub(p[1], p[0]), sub(p[2], p[0])));
            accelerator.AddTriangle(triangleOf(vertexOf(p[0], nrm), vertexOf(p[1], nrm), vertexOf(p[2], nrm), g));
        }
    }
};

// ---- image files -----------------------------------------------------------------------------------------------------
// What the reference shows is the rgba32f image blitted into an 8-bit window framebuffer (Shaders/Quad/fragment.glsl:9-10):
// row 0 of the image is the bottom row (gl_FragCoord), the value is clamped to [0,1] and stored as round(255 v).
inline unsigned char toByte(float v) {
    if (!(v > 0.0f)) return 0;                       // negatives and NaN
    if (v >= 1.0f) return 255;
    return (unsigned char)(v * 255.0f + 0.5f);
}

inline float srgbEncode(float v) {
    if (!(v > 0.0f)) return 0.0f;
    if (v >= 1.0f) return 1.0f;
    return v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
}

// 8-bit RGB rows, top row first
std::vector<unsigned char> quantize(const float* rgba, int w, int h, int transfer, float exposure) {
    std::vector<unsigned char> out((size_t)w * h * 3);
    for (int y = 0; y < h; y++) {
        const float* row = rgba + (size_t)(h - 1 - y) * w * 4;
        unsigned char* o = out.data() + (size_t)y * w * 3;
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) {
                const float v = row[4 * x + c];
                o[3 * x + c] = toByte(transfer == 0 ? v : srgbEncode(v * exposure));
            }
    }
    return out;
}

struct Crc32 {
    uint32_t table[256];
    Crc32() {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
    }
    uint32_t run(uint32_t crc, const unsigned char* p, size_t n) const {
        for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
        return crc;
    }
};

void putBE32(std::vector<unsigned char>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((unsigned char)(x >> s)); }

void pngChunk(std::vector<unsigned char>& file, const char* type, const std::vector<unsigned char>& body) {
    static const Crc32 crc;
    putBE32(file, (uint32_t)body.size());
    const size_t at = file.size();
    file.insert(file.end(), type, type + 4);
    file.insert(file.end(), body.begin(), body.end());
    putBE32(file, crc.run(0xFFFFFFFFu, file.data() + at, file.size() - at) ^ 0xFFFFFFFFu);
}

// PNG, 8-bit RGB, filter 0 on every row, zlib stream of stored (uncompressed) deflate blocks
std::vector<unsigned char> encodePng(const std::vector<unsigned char>& rgb, int w, int h) {
    std::vector<unsigned char> raw;
    raw.reserve((size_t)h * (3 * w + 1));
    for (int y = 0; y < h; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb.begin() + (size_t)y * w * 3, rgb.begin() + (size_t)(y + 1) * w * 3);
    }
    std::vector<unsigned char> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    size_t at = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - at);
        z.push_back(at + n == raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
        at += n;
    } while (at < raw.size());
    uint32_t a = 1, b = 0;                                                  // Adler-32 of the raw bytes
    for (size_t i = 0; i < raw.size();) {
        const size_t stop = std::min(raw.size(), i + 5552);
        for (; i < stop; i++) { a += raw[i]; b += a; }
        a %= 65521; b %= 65521;
    }
    putBE32(z, (b << 16) | a);
    std::vector<unsigned char> file = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::vector<unsigned char> ihdr;
    putBE32(ihdr, (uint32_t)w); putBE32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    pngChunk(file, "IHDR", ihdr);
    pngChunk(file, "IDAT", z);
    pngChunk(file, "IEND", {});
    return file;
}

// OpenEXR 2, one part, scan lines, no compression, three float32 channels (B, G, R: channels are stored in alphabetical
// order), increasing y with row 0 of the file the TOP row (the read-out's row 0 is the bottom one).  Lossless like the PFM.
std::vector<unsigned char> encodeExr(const float* rgba, int w, int h) {
    std::vector<unsigned char> f;
    auto bytesOf = [&f](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; f.insert(f.end(), b, b + n); };
    auto i32 = [&](int32_t v) { bytesOf(&v, 4); };
    auto f32 = [&](float v) { bytesOf(&v, 4); };
    auto text = [&](const char* t) { bytesOf(t, std::strlen(t) + 1); };
    auto attribute = [&](const char* name, const char* type, int32_t size) { text(name); text(type); i32(size); };
    i32(20000630);                                     // magic
    i32(2);                                            // version 2, no flags: single part, scan lines, short names
    attribute("channels", "chlist", 3 * 18 + 1);
    for (const char* c : {"B", "G", "R"}) { text(c); i32(2 /* FLOAT */); f.push_back(0); f.push_back(0); f.push_back(0); f.push_back(0); i32(1); i32(1); }
    f.push_back(0);
    attribute("compression", "compression", 1);
    f.push_back(0);                                    // NO_COMPRESSION
    attribute("dataWindow", "box2i", 16);
    i32(0); i32(0); i32(w - 1); i32(h - 1);
    attribute("displayWindow", "box2i", 16);
    i32(0); i32(0); i32(w - 1); i32(h - 1);
    attribute("lineOrder", "lineOrder", 1);
    f.push_back(0);                                    // INCREASING_Y
    attribute("pixelAspectRatio", "float", 4);
    f32(1.0f);
    attribute("screenWindowCenter", "v2f", 8);
    f32(0.0f); f32(0.0f);
    attribute("screenWindowWidth", "float", 4);
    f32(1.0f);
    f.push_back(0);                                    // end of header
    const size_t line = 8 + (size_t)w * 12;            // y, byte count, then the three channels of the row
    uint64_t at = f.size() + (size_t)h * 8;
    for (int y = 0; y < h; y++) { bytesOf(&at, 8); at += line; }
    for (int y = 0; y < h; y++) {
        i32(y);
        i32((int32_t)((size_t)w * 12));
        const float* row = rgba + (size_t)(h - 1 - y) * w * 4;
        for (int c = 2; c >= 0; c--)
            for (int x = 0; x < w; x++) f32(row[4 * x + c]);
    }
    return f;
}

void writeImage(const char* path, const float* rgba, int w, int h, int format, int transfer, float exposure) {
    if (!path || !rgba || w < 1 || h < 1) throw std::runtime_error("pth_image_write: bad argument");
    if (format < 0 || format > 3 || transfer < 0 || transfer > 1) throw std::runtime_error("pth_image_write: unknown format or transfer");
    std::vector<unsigned char> bytes;
    if (format == 0) {                                                      // PFM: RGB float, little endian, bottom row first
        char head[64];
        const int n = std::snprintf(head, sizeof head, "PF\n%d %d\n-1.0\n", w, h);
        bytes.assign(head, head + n);
        bytes.resize(n + (size_t)w * h * 12);
        unsigned char* o = bytes.data() + n;
        for (size_t i = 0; i < (size_t)w * h; i++) std::memcpy(o + 12 * i, rgba + 4 * i, 12);
    } else if (format == 3) {
        bytes = encodeExr(rgba, w, h);
    } else {
        const std::vector<unsigned char> rgb = quantize(rgba, w, h, transfer, exposure);
        if (format == 1) {                                                  // PPM P6
            char head[64];
            const int n = std::snprintf(head, sizeof head, "P6\n%d %d\n255\n", w, h);
            bytes.assign(head, head + n);
            bytes.insert(bytes.end(), rgb.begin(), rgb.end());
        } else {
            bytes = encodePng(rgb, w, h);
        }
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) throw std::runtime_error(std::string("cannot write ") + path);
    bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) throw std::runtime_error(std::string("short write to ") + path);
}

#define PTH_GUARD(...) try { __VA_ARGS__; return 0; } catch (const std::exception& e) { g_error = e.what(); return -1; }

extern "C" {

const char* pth_last_error(void) { return g_error.c_str(); }

pth_scene* pth_scene_create(const char* mesh_source) {
    pth_scene* s = new pth_scene();
    s->meshSource = mesh_source ? mesh_source : "";
    return s;
}
void pth_scene_destroy(pth_scene* s) { delete s; }

int pth_scene_clear(pth_scene* s) { PTH_GUARD(s->materials.clear(); s->accelerator.Clear(); s->meshRefs.clear(); s->instanceRefs.clear()) }

int pth_scene_add_mesh(pth_scene* s, const char* name, int normalized, const float* matrix16, const void* material48) {
    PTH_GUARD(
        Mat4 t = identity();
        if (matrix16) std::memcpy(t.m, matrix16, 64);
        Material m;
        if (material48) std::memcpy(&m, material48, 48);
        s->addMesh(name, normalized != 0, t, material48 ? &m : nullptr))
}

int pth_scene_generate_tiled_floor(pth_scene* s, const void* a48, const void* b48) {
    PTH_GUARD(Material a; Material b; std::memcpy(&a, a48, 48); std::memcpy(&b, b48, 48); s->tiledFloor(a, b))
}

int pth_scene_get_material_index(pth_scene* s, const void* material48, int32_t* index) {
    PTH_GUARD(Material m; std::memcpy(&m, material48, 48); *index = s->materialIndex(m))
}

int pth_scene_add_triangles(pth_scene* s, const void* triangles112, int32_t n) {
    PTH_GUARD(const Triangle* t = (const Triangle*)triangles112; for (int i = 0; i < n; i++) s->accelerator.AddTriangle(t[i]))
}

int pth_scene_initialize(pth_scene* s, int32_t seed) { PTH_GUARD(s->bundledScene(seed)) }

int pth_scene_initialize_extended(pth_scene* s, int32_t seed, int32_t extra_bunnies) {
    PTH_GUARD(if (extra_bunnies < 0) throw std::runtime_error("extra_bunnies < 0"); s->bundledScene(seed, extra_bunnies))
}

int pth_scene_generate_instances(pth_scene* s, int32_t count, int32_t seed) {
    PTH_GUARD(if (count < 1) throw std::runtime_error("count < 1"); s->instancedBunnies(count, seed))
}

int pth_scene_generate_soup(pth_scene* s, int32_t n, uint32_t seed) { PTH_GUARD(if (n < 2) throw std::runtime_error("soup needs at least 2 triangles"); s->soup(n, seed)) }

int pth_scene_build(pth_scene* s, int32_t threads) { PTH_GUARD(s->accelerator.Build(threads)) }

// Built-scene cache file: "PTSCENE1", three int32 counts (materials, triangles, nodes), one int32 zero, then the three
// arrays verbatim (Material 48 B, Triangle 112 B in built order, Node 32 B): what Scene.CopyToDevice uploads.
int pth_scene_save(pth_scene* s, const char* path) {
    PTH_GUARD(
        const Accelerator& a = s->accelerator;
        if (a.nodes.empty()) throw std::runtime_error("pth_scene_save: the scene has not been built");
        FILE* f = std::fopen(path, "wb");
        if (!f) throw std::runtime_error(std::string("cannot write ") + path);
        const int32_t head[4] = {(int32_t)s->materials.size(), (int32_t)a.triangles.size(), (int32_t)a.nodes.size(), 0};
        bool ok = std::fwrite("PTSCENE1", 1, 8, f) == 8 && std::fwrite(head, 4, 4, f) == 4;
        ok = ok && std::fwrite(s->materials.data(), sizeof(Material), s->materials.size(), f) == s->materials.size();
        ok = ok && std::fwrite(a.triangles.data(), sizeof(Triangle), a.triangles.size(), f) == a.triangles.size();
        ok = ok && std::fwrite(a.nodes.data(), sizeof(Node), a.nodes.size(), f) == a.nodes.size();
        ok = (std::fclose(f) == 0) && ok;
        if (!ok) throw std::runtime_error(std::string("short write to ") + path))
}

int pth_scene_load(pth_scene* s, const char* path) {
    PTH_GUARD(
        FILE* f = std::fopen(path, "rb");
        if (!f) throw std::runtime_error(std::string("cannot read ") + path);
        char magic[8];
        int32_t head[4];
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PTSCENE1", 8) == 0 && std::fread(head, 4, 4, f) == 4;
        ok = ok && head[0] >= 0 && head[1] >= 0 && head[2] >= 1 && head[2] <= 2 * std::max(head[1], 1);
        std::vector<Material> m;
        std::vector<Triangle> t;
        std::vector<Node> n;
        if (ok) {
            m.resize(head[0]); t.resize(head[1]); n.resize(head[2]);
            ok = std::fread(m.data(), sizeof(Material), m.size(), f) == m.size() && std::fread(t.data(), sizeof(Triangle), t.size(), f) == t.size() &&
                 std::fread(n.data(), sizeof(Node), n.size(), f) == n.size() && std::fgetc(f) == EOF;
        }
        std::fclose(f);
        if (!ok) throw std::runtime_error(std::string(path) + " is not a scene cache file");
        // the arrays go straight into Accelerator.Intersect's loops: every index they hold must be in range
        for (const Triangle& tri : t)
            if (tri.Material < 0 || tri.Material >= head[0]) throw std::runtime_error(std::string(path) + ": a triangle's material index is out of range");
        {
            std::vector<uint8_t> seen(n.size(), 0);
            seen[0] = 1;
            long long covered = 0;
            for (size_t i = 0; i < n.size(); i++) {
                const Node& nd = n[i];
                if (!seen[i]) throw std::runtime_error(std::string(path) + ": a node is not referenced by an earlier node");
                if (nd.Count > 0) {
                    if (nd.LeftFirst < 0 || (long long)nd.LeftFirst + nd.Count > head[1]) throw std::runtime_error(std::string(path) + ": a leaf's triangle range is out of bounds");
                    covered += nd.Count;
                } else if (head[1] > 0) {
                    const long long l = nd.LeftFirst;
                    if (l <= (long long)i || (l & 1) == 0 || l + 1 >= (long long)n.size() || seen[l] || seen[l + 1])
                        throw std::runtime_error(std::string(path) + ": an interior node's children are out of order or out of bounds");
                    seen[l] = seen[l + 1] = 1;
                }
            }
            if (head[1] > 0 && covered != head[1]) throw std::runtime_error(std::string(path) + ": the leaves do not cover the triangles exactly once");
        }
        s->materials.swap(m);
        s->accelerator.Clear();
        s->meshRefs.clear();
        s->instanceRefs.clear();
        for (const Triangle& tri : t) s->accelerator.AddTriangle(tri);   // insertion order := built order (a rebuild keeps working)
        s->accelerator.triangles.swap(t);
        s->accelerator.nodes.swap(n))
}

int pth_scene_counts(pth_scene* s, int32_t* out4) {
    PTH_GUARD(out4[0] = (int32_t)s->materials.size(); out4[1] = (int32_t)s->accelerator.triangles.size();
              out4[2] = (int32_t)s->accelerator.nodes.size(); out4[3] = s->accelerator.Depth())
}

int pth_scene_arrays(pth_scene* s, const void** materials48, const void** triangles112, const void** nodes32) {
    PTH_GUARD(if (materials48) *materials48 = s->materials.data(); if (triangles112) *triangles112 = s->accelerator.triangles.data();
              if (nodes32) *nodes32 = s->accelerator.nodes.data())
}

int pth_scene_references(pth_scene* s, const void** triangles112, const void** materials48, int32_t* n_triangles, int32_t* n_materials) {
    PTH_GUARD(const std::vector<Triangle>& r = s->accelerator.References();
              if (triangles112) *triangles112 = r.data(); if (materials48) *materials48 = s->materials.data();
              if (n_triangles) *n_triangles = (int32_t)r.size(); if (n_materials) *n_materials = (int32_t)s->materials.size())
}

// The scene as meshes + instances (see pth_scene::instancing).  Call with null arrays for the counts, then with arrays of those sizes:
// mesh_triangles112 n_mesh_triangles x 112 B (object space), mesh_first / mesh_count n_meshes each, instances72 n_instances x
// {float transform[16]; int32 mesh; int32 first_triangle} — the layout of pt_instance.
int pth_scene_instancing(pth_scene* s, int32_t* n_meshes, int32_t* n_mesh_triangles, int32_t* n_instances, void* mesh_triangles112, int32_t* mesh_first,
                         int32_t* mesh_count, void* instances72) {
    PTH_GUARD(
        const pth_scene::Instancing in = s->instancing();
        static_assert(sizeof(pth_scene::Instancing::Inst) == 72, "pt_instance layout");
        if (n_meshes) *n_meshes = (int32_t)in.meshFirst.size();
        if (n_mesh_triangles) *n_mesh_triangles = (int32_t)in.meshTriangles.size();
        if (n_instances) *n_instances = (int32_t)in.instances.size();
        if (mesh_triangles112) std::memcpy(mesh_triangles112, in.meshTriangles.data(), in.meshTriangles.size() * sizeof(Triangle));
        if (mesh_first) std::memcpy(mesh_first, in.meshFirst.data(), in.meshFirst.size() * 4);
        if (mesh_count) std::memcpy(mesh_count, in.meshCount.data(), in.meshCount.size() * 4);
        if (instances72) std::memcpy(instances72, in.instances.data(), in.instances.size() * 72))
}

// Camera.Update first half, Camera.cs:48-50
int pth_camera_basis(const float* position3, const float* target3, float* forward3, float* right3, float* up3) {
    PTH_GUARD(
        const Vec3 pos{position3[0], position3[1], position3[2]}; const Vec3 tgt{target3[0], target3[1], target3[2]};
        const Vec3 f = normalize(sub(tgt, pos));
        const Vec3 r = normalize(cross(Vec3{0, 1, 0}, f));
        const Vec3 u = normalize(cross(f, r));
        forward3[0] = f.x; forward3[1] = f.y; forward3[2] = f.z;
        right3[0] = r.x; right3[1] = r.y; right3[2] = r.z;
        up3[0] = u.x; up3[1] = u.y; up3[2] = u.z)
}

// Camera.GetFocalDistance's Min(distance, 20) and the second half of Camera.Update, Camera.cs:51-58,94; SetUniform, Camera.cs:97-105
int pth_camera_state(const float* position3, const float* forward3, const float* right3, const float* up3,
                     float hit_distance, int32_t width, int32_t height, void* camera_state96, float* focal_out) {
    PTH_GUARD(
        const Vec3 pos{position3[0], position3[1], position3[2]}; const Vec3 f{forward3[0], forward3[1], forward3[2]};
        const Vec3 r{right3[0], right3[1], right3[2]}; const Vec3 u{up3[0], up3[1], up3[2]};
        const float focal = minNaN(hit_distance, 20.0f);
        const float ar = (float)width / (float)height;
        const Vec3 c = add(pos, scale(focal, f));
        const Vec3 halfRight = scale(0.5f * focal * ar, r); const Vec3 halfUp = scale(0.5f * focal, u);
        const Vec3 p1 = add(sub(c, halfRight), halfUp); const Vec3 p2 = add(add(c, halfRight), halfUp); const Vec3 p3 = sub(sub(c, halfRight), halfUp);
        float* o = (float*)camera_state96;
        std::memset(o, 0, 96);
        const Vec3 cols[6] = {p1, p2, p3, pos, r, u};
        for (int k = 0; k < 6; k++) { o[4 * k] = cols[k].x; o[4 * k + 1] = cols[k].y; o[4 * k + 2] = cols[k].z; }
        if (focal_out) *focal_out = focal)
}

int pth_image_write(const char* path, const float* rgba, int32_t width, int32_t height, int32_t format, int32_t transfer, float exposure) {
    PTH_GUARD(writeImage(path, rgba, width, height, format, transfer, exposure))
}

}  // extern "C"
