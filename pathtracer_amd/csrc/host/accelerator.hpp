// Host mirror of PathTracer/Accelerator.cs: binned-SAH BVH build (8 bins, leaves below 3
// primitives) producing the reference's flat Node[] and reordered Triangle[].
//
// Same tree, different machine code: the reference allocates a List per bin per node and
// recurses; this builder keeps one permutation array of 4-byte primitive ids, bins with a
// stable counting sort into a scratch array, walks with an explicit stack, and can build
// disjoint subtrees on several threads.  Every float comparison and min/max is written in
// the reference's operand order, so the output is byte-identical to the C# builder's.
#pragma once
#include <cstdint>
#include <vector>

namespace pth {

struct Vec3 { float x, y, z; };
struct Vertex { Vec3 Position; float pad0; Vec3 Normal; float pad1; };          // Vertex.cs:5-11
struct Triangle { Vertex V1, V2, V3; int32_t Material; int32_t pad[3]; };      // Triangle.cs:7-13
struct Box { Vec3 Min, Max; };                                                  // Box.cs:8-12
struct Node { Box Bounds; int32_t LeftFirst; int32_t Count; };                  // Node.cs:5-10
struct Material { float Color[4]; float Emissive[4]; int32_t Type; float Index; float Roughness; int32_t pad; };  // Material.cs:7-28
static_assert(sizeof(Triangle) == 112 && sizeof(Node) == 32 && sizeof(Material) == 48, "blittable layouts");

class Accelerator {
public:
    void AddTriangle(const Triangle& t);          // Accelerator.cs:38-49
    void Build(int threads = 1);                  // Accelerator.cs:113-134
    void Clear();                                 // Accelerator.cs:179-186
    int Count() const { return (int)references.size(); }
    const std::vector<Triangle>& References() const { return references; }   // the triangles in AddTriangle order
    int Depth() const;

    std::vector<Triangle> triangles;              // Buffer<Triangle> mirror, valid after Build
    std::vector<Node> nodes;                      // Buffer<Node> mirror, valid after Build

private:
    struct Scratch { std::vector<uint32_t> tmp; std::vector<uint8_t> bin; };
    struct Split { int count_left; Box left, right; };
    bool partition(const Box& bounds, int first, int count, Scratch& sc, Split* out);
    void buildSubtree(std::vector<Node>& out, int first, int count, const Box& bounds, Scratch& sc);
    Box boundsOf(int first, int count) const;

    std::vector<Triangle> references;
    std::vector<Box> boxes;        // per reference: padded triangle box (Box.cs:95-102)
    std::vector<Vec3> centroids;   // per reference: box centroid (Box.cs:20-23)
    std::vector<uint32_t> order;   // permutation of reference ids (the reference's `primitives` list)
};

}  // namespace pth
