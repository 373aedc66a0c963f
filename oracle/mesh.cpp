// CPU ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// Restates PathTracer/Library/Geometry/{Mesh,Face,Group,Frame,Vertex,Material}.cs.
//
// Two sources of a mesh: OBJ/MTL text (Mesh.LoadFromFile) and a binary blob of
// already parsed arrays, which is what travels to machines that do not have
// the reference's asset files.  Blob layout (little endian):
//   "PTMESH1\0", u32 meshCount, then per mesh:
//     u32 nameLen, name bytes
//     u32 nPositions, 3*f32 each | u32 nNormals, 3*f32 each | u32 nTexcoords (count only)
//     u32 nGroups, per group: u32 nameLen, name, 14*f32 material
//       (ambient, diffuse, specular, emissive, specularExponent, transparency),
//       u32 nFaces, per face: u32 nFrames, 3*i32 (I,J,K; already 0-based, -1 = absent) each
#include "oracle.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <fstream>
#include <sstream>
#include <map>
#include <limits>
#include <stdexcept>

namespace oracle {

static const float INF = std::numeric_limits<float>::infinity();

Mesh::Mesh() : min{INF, INF, INF}, max{-INF, -INF, -INF}, TransformM(MatIdentity()) {}

// Mesh.cs:50-60
Vec3 Mesh::GetPosition(int i) const {
    Vec3 p = positions[i];
    if (Normalized) {
        Vec3 d = max - min;
        float m = MathFMax(MathFMax(d.x, d.y), d.z);
        p = (p - min) / m;
    }
    return Transform(p, TransformM);
}

// Mesh.cs:67-70
Vec3 Mesh::GetNormal(int k) const { return Normalize(TransformNormal(normals[k], TransformM)); }

static bool isDotNetSpace(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

static std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isDotNetSpace((unsigned char)s[a])) a++;
    while (b > a && isDotNetSpace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

// Mesh.cs:240-259
static void ReadObjLine(std::string s, std::string& key, std::string& arg) {
    key.clear();
    arg.clear();
    if (s.empty()) return;
    for (char& c : s) if (c == '\t') c = ' ';
    s = trim(s);
    size_t sp = s.find(' ');
    if (sp == std::string::npos) return;
    key = trim(s.substr(0, sp));
    arg = trim(s.substr(sp + 1));
}

// Mesh.cs:235-238
static std::vector<std::string> ReadParts(std::string s) {
    s = trim(s);
    for (char& c : s) if (c == '\t') c = ' ';
    std::vector<std::string> out;
    size_t i = 0;
    while (i <= s.size()) {
        size_t j = s.find(' ', i);
        if (j == std::string::npos) j = s.size();
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j + 1;
    }
    return out;
}

// float.Parse(InvariantCulture) is correctly rounded on .NET Core 3.x, as is glibc strtof.
static float ReadFloat(const std::string& s) { return strtof(s.c_str(), nullptr); }

static Vec3 ReadVector3(const std::string& s) {
    auto parts = ReadParts(s);
    if (parts.size() < 3) throw std::runtime_error("Cannot parse .obj file");
    return {ReadFloat(parts[0]), ReadFloat(parts[1]), ReadFloat(parts[2])};
}

// Mesh.cs:297-327
static Frame ReadFrame(const std::string& s) {
    std::vector<std::string> parts;
    size_t i = 0;
    while (true) {
        size_t j = s.find('/', i);
        if (j == std::string::npos) { parts.push_back(s.substr(i)); break; }
        parts.push_back(s.substr(i, j - i));
        i = j + 1;
    }
    int a = 0, b = 0, c = 0;
    switch (parts.size()) {
        case 1: a = atoi(parts[0].c_str()); break;
        case 2: a = atoi(parts[0].c_str()); b = atoi(parts[1].c_str()); break;
        case 3:
            a = atoi(parts[0].c_str());
            if (!parts[1].empty()) b = atoi(parts[1].c_str());
            c = atoi(parts[2].c_str());
            break;
    }
    return Frame{a, b, c};
}

static std::vector<std::string> readLines(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string all = ss.str();
    // StreamReader consumes one UTF-8 byte-order mark
    if (all.size() >= 3 && (unsigned char)all[0] == 0xEF && (unsigned char)all[1] == 0xBB && (unsigned char)all[2] == 0xBF) all = all.substr(3);
    std::vector<std::string> lines;
    size_t i = 0;
    while (i < all.size()) {
        size_t j = all.find('\n', i);
        if (j == std::string::npos) j = all.size();
        std::string line = all.substr(i, j - i);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        lines.push_back(line);
        i = j + 1;
    }
    return lines;
}

// Mesh.cs:181-233
static std::map<std::string, ObjMaterial> ReadMtlFile(const std::string& path) {
    std::map<std::string, ObjMaterial> materials;
    std::string name;
    ObjMaterial material;
    std::string key, arg;
    for (const std::string& line : readLines(path)) {
        ReadObjLine(line, key, arg);
        if (key == "newmtl") {
            materials[name] = material;
            name = arg;
            material = ObjMaterial();
        } else if (key == "Ka") material.Emissive = ReadVector3(arg);   // sic: Mesh.cs:203-205
        else if (key == "Kd") material.Diffuse = ReadVector3(arg);
        else if (key == "Ke") material.Emissive = ReadVector3(arg);
        else if (key == "Ks") material.Specular = ReadVector3(arg);
        else if (key == "Ns") material.SpecularExponent = ReadFloat(arg);
        else if (key == "Tr") material.Transparency = ReadFloat(arg);
        else if (key == "d") material.Transparency = 1.0f - ReadFloat(arg);
    }
    materials[name] = material;
    return materials;
}

// Mesh.cs:72-179
Mesh Mesh::LoadFromFile(const std::string& path) {
    Mesh mesh;
    std::vector<Face> faces;
    std::map<std::string, ObjMaterial> materials;
    ObjMaterial material;
    std::string name;
    std::string key, arg;
    std::string dir = path.substr(0, path.find_last_of('/') + 1);

    for (const std::string& line : readLines(path)) {
        ReadObjLine(line, key, arg);
        if (key == "v") {
            Vec3 position = ReadVector3(arg);
            mesh.positions.push_back(position);
            mesh.min = Vec3Min(mesh.min, position);
            mesh.max = Vec3Max(mesh.max, position);
        } else if (key == "vt") {
            mesh.texcoordCount++;
        } else if (key == "vn") {
            mesh.normals.push_back(ReadVector3(arg));
        } else if (key == "f") {
            auto parts = ReadParts(arg);
            Face face;
            for (const std::string& part : parts) {
                Frame f = ReadFrame(part);
                int i = f.I < 0 ? f.I + (int)mesh.positions.size() : f.I - 1;
                int j = f.J < 0 ? f.J + mesh.texcoordCount : f.J - 1;
                int k = f.K < 0 ? f.K + (int)mesh.normals.size() : f.K - 1;
                face.frames.push_back(Frame{i, j, k});
            }
            faces.push_back(face);
        } else if (key == "mtllib") {
            for (auto& kv : ReadMtlFile(dir + arg)) materials[kv.first] = kv.second;
        } else if (key == "g") {
            if (!faces.empty()) {
                mesh.groups.push_back(Group{faces, name, material});
                faces.clear();
            }
            name = arg;
        } else if (key == "usemtl") {
            if (!faces.empty()) {
                mesh.groups.push_back(Group{faces, name, material});
                faces.clear();
            }
            auto it = materials.find(arg);
            material = (it != materials.end()) ? it->second : ObjMaterial();
        }
    }
    if (!faces.empty()) mesh.groups.push_back(Group{faces, name, material});
    return mesh;
}

// Face.cs:23-65.  The three locals a, b, c are value-type copies that live
// across loop iterations, so a flat normal assigned in one fan step stays on
// `a` (and on `b`, which becomes the previous `c`) for the next step.
void Mesh::Triangulate(const Face& face, std::vector<GeoVertex>& out) const {
    auto build = [this](const Frame& fr) {
        GeoVertex v;
        v.Position = fr.I < 0 ? Vec3{0, 0, 0} : GetPosition(fr.I);
        v.Normal = fr.K < 0 ? Vec3{0, 0, 0} : GetNormal(fr.K);
        return v;
    };
    GeoVertex a = build(face.frames[0]);
    GeoVertex b = build(face.frames[1]);
    for (size_t i = 2; i < face.frames.size(); i++) {
        GeoVertex c = build(face.frames[i]);
        Vec3 sum = a.Normal + b.Normal + c.Normal;
        if (Dot(sum, sum) < 0.5f) {
            Vec3 edge1 = a.Position - c.Position;
            Vec3 edge2 = b.Position - c.Position;
            Vec3 normal = Normalize(Cross(edge1, edge2));
            a.Normal = b.Normal = c.Normal = normal;
        }
        out.push_back(a);
        out.push_back(b);
        out.push_back(c);
        b = c;
    }
}

// ------------------------------------------------------------------- blob
static void put32(std::string& o, uint32_t v) { o.append((const char*)&v, 4); }
static void putf(std::string& o, float v) { o.append((const char*)&v, 4); }
static void putstr(std::string& o, const std::string& s) { put32(o, (uint32_t)s.size()); o.append(s); }

bool WriteMeshBlob(const std::string& path, const std::vector<std::pair<std::string, Mesh>>& meshes) {
    std::string o("PTMESH1\0", 8);
    put32(o, (uint32_t)meshes.size());
    for (auto& nm : meshes) {
        const Mesh& m = nm.second;
        putstr(o, nm.first);
        put32(o, (uint32_t)m.positions.size());
        for (auto& p : m.positions) { putf(o, p.x); putf(o, p.y); putf(o, p.z); }
        put32(o, (uint32_t)m.normals.size());
        for (auto& p : m.normals) { putf(o, p.x); putf(o, p.y); putf(o, p.z); }
        put32(o, (uint32_t)m.texcoordCount);
        put32(o, (uint32_t)m.groups.size());
        for (auto& g : m.groups) {
            putstr(o, g.name);
            const ObjMaterial& t = g.material;
            const Vec3* vs[4] = {&t.Ambient, &t.Diffuse, &t.Specular, &t.Emissive};
            for (auto* v : vs) { putf(o, v->x); putf(o, v->y); putf(o, v->z); }
            putf(o, t.SpecularExponent);
            putf(o, t.Transparency);
            put32(o, (uint32_t)g.faces.size());
            for (auto& f : g.faces) {
                put32(o, (uint32_t)f.frames.size());
                for (auto& fr : f.frames) { put32(o, (uint32_t)fr.I); put32(o, (uint32_t)fr.J); put32(o, (uint32_t)fr.K); }
            }
        }
    }
    FILE* fp = fopen(path.c_str(), "wb");
    if (!fp) return false;
    bool ok = fwrite(o.data(), 1, o.size(), fp) == o.size();
    fclose(fp);
    return ok;
}

namespace {
struct Reader {
    const std::string& b;
    size_t p = 0;
    explicit Reader(const std::string& s) : b(s) {}
    uint32_t u32() { if (p + 4 > b.size()) throw std::runtime_error("mesh blob truncated"); uint32_t v; memcpy(&v, b.data() + p, 4); p += 4; return v; }
    float f32() { uint32_t v = u32(); float f; memcpy(&f, &v, 4); return f; }
    std::string str() { uint32_t n = u32(); if (p + n > b.size()) throw std::runtime_error("mesh blob truncated"); std::string s = b.substr(p, n); p += n; return s; }
};
}  // namespace

Mesh Mesh::LoadFromBlob(const std::string& blob, const std::string& want) {
    if (blob.size() < 12 || memcmp(blob.data(), "PTMESH1\0", 8) != 0) throw std::runtime_error("not a mesh blob");
    Reader r(blob);
    r.p = 8;
    uint32_t n = r.u32();
    for (uint32_t mi = 0; mi < n; mi++) {
        Mesh m;
        std::string name = r.str();
        uint32_t np = r.u32();
        for (uint32_t i = 0; i < np; i++) {
            Vec3 p; p.x = r.f32(); p.y = r.f32(); p.z = r.f32();
            m.positions.push_back(p);
            m.min = Vec3Min(m.min, p);
            m.max = Vec3Max(m.max, p);
        }
        uint32_t nn = r.u32();
        for (uint32_t i = 0; i < nn; i++) { Vec3 p; p.x = r.f32(); p.y = r.f32(); p.z = r.f32(); m.normals.push_back(p); }
        m.texcoordCount = (int)r.u32();
        uint32_t ng = r.u32();
        for (uint32_t g = 0; g < ng; g++) {
            Group grp;
            grp.name = r.str();
            Vec3* vs[4] = {&grp.material.Ambient, &grp.material.Diffuse, &grp.material.Specular, &grp.material.Emissive};
            for (auto* v : vs) { v->x = r.f32(); v->y = r.f32(); v->z = r.f32(); }
            grp.material.SpecularExponent = r.f32();
            grp.material.Transparency = r.f32();
            uint32_t nf = r.u32();
            for (uint32_t f = 0; f < nf; f++) {
                Face face;
                uint32_t k = r.u32();
                for (uint32_t q = 0; q < k; q++) { Frame fr; fr.I = (int)r.u32(); fr.J = (int)r.u32(); fr.K = (int)r.u32(); face.frames.push_back(fr); }
                grp.faces.push_back(face);
            }
            m.groups.push_back(grp);
        }
        if (name == want) return m;
    }
    throw std::runtime_error("mesh '" + want + "' not in blob");
}

}  // namespace oracle
