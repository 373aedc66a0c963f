// CPU ORACLE — TEST INFRASTRUCTURE ONLY.
//
// A plain C++17 restatement of the reference's CPU path (C# BVH build and
// closest-hit traversal) and of its GLSL wavefront kernels, used as the
// checker for the HIP kernels.  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load this library; the product
// (pathtracer_amd/, libpt_hip.so, libpt_host.so) never links or calls it.
//
// Parity status: the reference ships no tests, golden vectors or fixtures and
// cannot be built here (C# netcoreapp3.1 + OpenTK), so results are
// "parity unpinned" against the real reference.  The restatement is pinned
// by (1) known answers of .NET's seeded System.Random, (2) structural BVH
// invariants, (3) brute-force closest hit == BVH closest hit, (4) hand-checked
// tiny BVHs.  See tests/test_oracle_*.py.
//
// Build: g++ -O2 -ffp-contract=off -fno-fast-math (x86-64 baseline: SSE2 scalar
// float arithmetic, no FMA), so every float operation rounds exactly once as
// RyuJIT's scalar SSE code does.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <string>

namespace oracle {

// ---------------------------------------------------------------- POD layouts
// PathTracer/Vertex.cs:5-11
struct Vec3 { float x, y, z; };
struct Vertex { Vec3 Position; float pad0; Vec3 Normal; float pad1; };
// PathTracer/Triangle.cs:7-13
struct Triangle { Vertex V1, V2, V3; int32_t Material; int32_t pad[3]; };
// PathTracer/Box.cs:8-12
struct Box { Vec3 Min, Max; };
// PathTracer/Node.cs:5-10
struct Node { Box Bounds; int32_t LeftFirst; int32_t Count; };
// PathTracer/Material.cs:7-28
struct Material { float Color[4]; float Emissive[4]; int32_t Type; float Index; float Roughness; int32_t pad; };
enum MaterialType { Diffuse = 0, Mirror = 1, Dielectric = 2, Metal = 3 };
// PathTracer/State.cs:9-41
struct CameraState { Vec3 P1; float p0; Vec3 P2; float p1; Vec3 P3; float p2; Vec3 Position; float p3; Vec3 Right; float p4; Vec3 Up; float p5; };
struct RenderState { int32_t Width, Height; int32_t pad[2]; float SkyColor[4]; };
struct FrameState { int32_t Frames, Samples; float FocalDistance; int32_t pad; CameraState Camera; };

static_assert(sizeof(Vertex) == 32 && offsetof(Vertex, Normal) == 16, "Vertex layout");
static_assert(sizeof(Triangle) == 112 && offsetof(Triangle, V2) == 32 && offsetof(Triangle, V3) == 64 && offsetof(Triangle, Material) == 96, "Triangle layout");
static_assert(sizeof(Box) == 24 && offsetof(Box, Max) == 12, "Box layout");
static_assert(sizeof(Node) == 32 && offsetof(Node, LeftFirst) == 24 && offsetof(Node, Count) == 28, "Node layout");
static_assert(sizeof(Material) == 48 && offsetof(Material, Emissive) == 16 && offsetof(Material, Type) == 32 && offsetof(Material, Index) == 36 && offsetof(Material, Roughness) == 40, "Material layout");
static_assert(sizeof(CameraState) == 96 && offsetof(CameraState, P2) == 16 && offsetof(CameraState, Position) == 48 && offsetof(CameraState, Up) == 80, "CameraState layout");
static_assert(sizeof(RenderState) == 32 && offsetof(RenderState, SkyColor) == 16, "RenderState layout");
static_assert(sizeof(FrameState) == 112 && offsetof(FrameState, FocalDistance) == 8 && offsetof(FrameState, Camera) == 16, "FrameState layout");

// ---------------------------------------------------------- System.Numerics
inline Vec3 v3(float x, float y, float z) { return Vec3{x, y, z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
float Dot(Vec3 a, Vec3 b);        // (ax*bx + ay*by) + az*bz
Vec3 Cross(Vec3 a, Vec3 b);
Vec3 Normalize(Vec3 a);           // a / sqrt(Dot(a,a))
Vec3 Vec3Min(Vec3 a, Vec3 b);     // (a < b) ? a : b per component
Vec3 Vec3Max(Vec3 a, Vec3 b);     // (a > b) ? a : b per component
float MathFMax(float x, float y); // IEEE 754:2019 maximum (NaN-propagating)
float MathFMin(float x, float y);

struct Matrix4x4 { float m[4][4]; };
Matrix4x4 MatIdentity();
Matrix4x4 MatRotationX(float a);
Matrix4x4 MatRotationY(float a);
Matrix4x4 MatRotationZ(float a);
Matrix4x4 MatTranslation(float x, float y, float z);
Matrix4x4 MatScale(float s);
Matrix4x4 MatMul(const Matrix4x4& a, const Matrix4x4& b);
Vec3 Transform(Vec3 p, const Matrix4x4& m);
Vec3 TransformNormal(Vec3 n, const Matrix4x4& m);

// ------------------------------------------------------------ System.Random
// .NET Core 3.1 BCL, seeded constructor (Knuth subtractive generator).
class DotNetRandom {
public:
    explicit DotNetRandom(int32_t seed);
    int32_t Next();
    int32_t Next(int32_t maxValue);
    double NextDouble();
private:
    int32_t InternalSample();
    int32_t sa[56];
    int inext, inextp;
};

// ----------------------------------------------------------------- geometry
Box BoxNegative();
Vec3 BoxCentroid(const Box& b);
int BoxDomain(const Box& b);
float BoxArea(const Box& b);
Box BoxUnion(const Box& a, const Box& b);
Box BoxFromTriangle(const Triangle& t);

struct Ray { Vec3 Origin, Direction, Inverse; };
Ray MakeRay(Vec3 origin, Vec3 direction);
bool BoxIntersect(const Box& b, const Ray& r, float* tmin, float* tmax);

// PathTracer/Intersection.cs:5-9 plus the fields this build adds (Tri, U, V):
// Tri = index into the post-Build triangle array of the accepted hit, -1 = miss.
struct Intersection { float Distance; Vec3 Point; int32_t Tri; float U, V; };

struct TraversalStats { uint64_t steps; uint64_t tris; uint64_t rays; };

// PathTracer/Accelerator.cs
class Accelerator {
public:
    void AddTriangle(const Triangle& t);
    void Build();
    void Clear();
    int Count() const { return (int)primitives.size(); }
    // literal C# traversal (far-first push order, first tested wins ties)
    void Intersect(const Ray& ray, Intersection& i) const;
    // near-first, distance-culled traversal; ties -> smallest triangle index
    void IntersectCanonical(const Ray& ray, Intersection& i, TraversalStats* st = nullptr) const;
    // every triangle, ties -> smallest triangle index
    void IntersectBrute(const Ray& ray, Intersection& i) const;
    // any hit with 0 < t < tmax (shadow ray)
    bool Occluded(const Ray& ray, float tmax, TraversalStats* st = nullptr) const;
    int MaxDepth() const;
    // which node records the canonical walks restate: 0 the exact ones, 1 the quantised ones, 2 (default) the GPU library's own rule
    bool WalksQuantized() const;
    int quantized = 2;
    long long CountQuadRecords() const;
    long long quadRecords = -1;           // set where a tree is installed (Build, orc_scene_set_built) so that walks on many threads only read it
    const std::vector<Triangle>& References() const { return references; }

    std::vector<Triangle> triangles;  // Buffer<Triangle> CPU mirror
    std::vector<Node> nodes;          // Buffer<Node> CPU mirror
private:
    struct Primitive { int Index; Box Bounds; Vec3 Centroid; };
    void Subdivide(int index);
    int Partition(const Node& node);
    Box Bounds(int first, int count) const;
    Box CentroidBounds(int first, int count) const;
    std::vector<Triangle> references;
    std::vector<Primitive> primitives;
};
bool TriangleIntersect(const Triangle& t, const Ray& ray, Intersection& i, int triIndex);

// ------------------------------------------------------------ mesh / scene
// PathTracer/Library/Geometry/*.cs
struct ObjMaterial { Vec3 Ambient{0, 0, 0}, Diffuse{1, 1, 1}, Specular{1, 1, 1}, Emissive{0, 0, 0}; float SpecularExponent = 0, Transparency = 0; };
struct Frame { int I, J, K; };
struct Face { std::vector<Frame> frames; };
struct Group { std::vector<Face> faces; std::string name; ObjMaterial material; };
struct GeoVertex { Vec3 Position; Vec3 Normal; };
class Mesh {
public:
    std::vector<Vec3> positions, normals;
    int texcoordCount = 0;
    std::vector<Group> groups;
    Vec3 min, max;
    bool Normalized = false;
    Matrix4x4 TransformM;
    Mesh();
    Vec3 GetPosition(int i) const;
    Vec3 GetNormal(int k) const;
    static Mesh LoadFromFile(const std::string& path);           // OBJ + MTL text
    static Mesh LoadFromBlob(const std::string& blob, const std::string& name);  // parsed arrays
    void Triangulate(const Face& f, std::vector<GeoVertex>& out3) const;  // appends 3 vertices per triangle
};
// blob of parsed meshes (tests/golden/meshes.bin); see oracle/mesh.cpp for the format
bool WriteMeshBlob(const std::string& path, const std::vector<std::pair<std::string, Mesh>>& meshes);

class Scene {
public:
    std::vector<Material> materials;
    Accelerator accelerator;
    std::string meshSource;   // a directory of OBJ files, or a mesh blob file
    void Clear();
    Intersection Intersect(const Ray& ray) const;
    void AddMesh(const std::string& name, bool normalized, const Matrix4x4& transform);
    void AddMesh(const std::string& name, bool normalized, const Matrix4x4& transform, const Material& material);
    void GenerateTiledFloor(const Material& a, const Material& b);
    void InitializeScene(int seed);   // Game.cs:66-136
    void GenerateSoup(int nTriangles, uint32_t seed);
    void Build() { accelerator.Build(); }
    int GetMaterialIndex(const Material& m);
};

// ------------------------------------------------------------------ camera
struct Camera {
    Vec3 position, target, forward, right, up, p1, p2, p3;
    float focal;
    int width, height;
    const Scene* scene;
    void Set(Vec3 position, Vec3 target);
    void Update();
    float GetFocalDistance() const;
    void SetUniform(FrameState& fs) const;
};

// --------------------------------------------------------------- wavefront
struct PathConfig {
    int max_bounces = 10;
    int nee = 0;           // next-event estimation at diffuse vertices: 0 off, 1 on, 2 on with multiple importance sampling
    int rng_mode = 0;      // 0 = reference seeding, 1 = decorrelated hash
    int tile_rank = 0, tile_world = 1;
    // Literal behaviours of the reference's GLSL that the product path departs from, switchable for the tests that state how far
    // apart the two are: 1 = closest hit by the walk and triangle test of extend.glsl:171-299 (hit normal from its u, v, w);
    // 2 = the hit normal goes through packNormal / unpackNormal (16+16 bits, extend.glsl:23-27, shade.glsl:29-36);
    // 4 = the pixel is the running mean of shade.glsl:89-96 (the output buffer then holds the image, not a sum).
    int literal = 0;
};
enum { LITERAL_GLSL_EXTEND = 1, LITERAL_PACKED_NORMAL = 2, LITERAL_RUNNING_MEAN = 4 };

// ---- the reference's GLSL statements, restated apart from the product path (second.cpp)
struct GlslHit { float Distance; int32_t Tri; Vec3 Normal; };
GlslHit IntersectGlsl(const Accelerator& acc, const Ray& ray, int* stack_peak, int* candidate_count);
uint32_t PackNormalGlsl(Vec3 n);
Vec3 UnpackNormalGlsl(uint32_t p);
float SecondSchlick(Vec3 direction, Vec3 normal, float n1, float n2);
Vec3 SecondDiffuseReflection(Vec3 normal, uint32_t& seed);
// the product path's statements of the same helpers (wavefront.cpp), exposed for the comparison
float ProbeSchlick(Vec3 direction, Vec3 normal, float n1, float n2);
Vec3 ProbeLambertSample(Vec3 normal, uint32_t& seed);
void ProbeDiffuseThroughput(Vec3 normal, Vec3 bounce, const float color[3], float out[3]);
// deterministic elementary functions shared (by restatement, not by code) with the kernels
float det_sin(float x);
float det_cos(float x);
float det_exp(float x);

uint32_t RayPixelOfIndex(uint32_t index, int width, int height, int rank, int world, bool* valid);
// generate.glsl:135-200 for thread `index`; returns false when the thread writes no ray
bool GenerateRay(uint32_t index, const RenderState& rs, const FrameState& fs, const PathConfig& cfg,
                 Vec3* origin, Vec3* dir, uint32_t* pixel);

struct LightTable { std::vector<int> tris; std::vector<float> cdf; float totalArea = 0; };
LightTable BuildLightTable(const Scene& s);

struct RenderStats { uint64_t extend_rays = 0, shadow_rays = 0; TraversalStats extend{0, 0, 0}, shadow{0, 0, 0}; };
// Renders `spp` samples starting at fs.Frames into sum (rgba, W*H*4 floats, += per sample).
// Rows [row0,row1) of 8x8 pixel groups are rendered (row blocks for threading).
void RenderSamples(const Scene& scene, const LightTable& lights, const RenderState& rs, FrameState fs,
                   const PathConfig& cfg, int spp, float* sum, uint32_t group0, uint32_t group1,
                   RenderStats* stats, int traversal /*0 literal,1 canonical*/, uint32_t group_stride = 1);

}  // namespace oracle
