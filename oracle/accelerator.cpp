// CPU ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// Restates PathTracer/Box.cs, Ray.cs, Triangle.cs and Accelerator.cs.
#include <cstring>
#include "oracle.hpp"
#include <cmath>
#include <limits>
#include <algorithm>

namespace oracle {

static const float Epsilon = 0.00001f;   // Accelerator.cs:10
static const int LeafNodeSize = 3;       // Accelerator.cs:11
static const int BucketCount = 8;        // Accelerator.cs:12
static const float INF = std::numeric_limits<float>::infinity();

// Box.cs:14-18
Box BoxNegative() { return Box{{INF, INF, INF}, {-INF, -INF, -INF}}; }

// Box.cs:20-23
Vec3 BoxCentroid(const Box& b) { return 0.5f * b.Min + 0.5f * b.Max; }

// Box.cs:25-43, Vector.cs:28-31
int BoxDomain(const Box& b) {
    Vec3 dim = b.Max - b.Min;
    float l = MathFMax(MathFMax(dim.x, dim.y), dim.z);
    if (l == dim.x) return 0;
    else if (l == dim.y) return 1;
    else return 2;
}

// Box.cs:45-49
float BoxArea(const Box& b) {
    Vec3 d = b.Max - b.Min;
    return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
}

// Box.cs:81-88
Box BoxUnion(const Box& a, const Box& b) {
    Box r;
    r.Max = Vec3Max(a.Max, b.Max);
    r.Min = Vec3Min(a.Min, b.Min);
    return r;
}

// Box.cs:90-102
Box BoxFromTriangle(const Triangle& t) {
    Vec3 e{0.00001f, 0.00001f, 0.00001f};
    Box r;
    r.Max = Vec3Max(Vec3Max(t.V1.Position, t.V2.Position), t.V3.Position) + e;
    r.Min = Vec3Min(Vec3Min(t.V1.Position, t.V2.Position), t.V3.Position) - e;
    return r;
}

// Ray.cs:14-19
Ray MakeRay(Vec3 origin, Vec3 direction) {
    Ray r;
    r.Origin = origin;
    r.Direction = direction;
    r.Inverse = {1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z};
    return r;
}

// Box.cs:51-79
bool BoxIntersect(const Box& b, const Ray& ray, float* tmin_out, float* tmax_out) {
    float t1 = (b.Min.x - ray.Origin.x) * ray.Inverse.x;
    float t2 = (b.Max.x - ray.Origin.x) * ray.Inverse.x;
    float t3 = (b.Min.y - ray.Origin.y) * ray.Inverse.y;
    float t4 = (b.Max.y - ray.Origin.y) * ray.Inverse.y;
    float t5 = (b.Min.z - ray.Origin.z) * ray.Inverse.z;
    float t6 = (b.Max.z - ray.Origin.z) * ray.Inverse.z;
    float tmin = MathFMax(MathFMax(MathFMin(t1, t2), MathFMin(t3, t4)), MathFMin(t5, t6));
    float tmax = MathFMin(MathFMin(MathFMax(t1, t2), MathFMax(t3, t4)), MathFMax(t5, t6));
    if (tmin_out) *tmin_out = tmin;
    if (tmax_out) *tmax_out = tmax;
    return (tmax > 0.0f && tmin < tmax);
}

// Triangle.cs:15-47.  Returns true when the hit was accepted (closer than i.Distance).
bool TriangleIntersect(const Triangle& t, const Ray& ray, Intersection& i, int triIndex) {
    Vec3 edge1 = t.V2.Position - t.V1.Position;
    Vec3 edge2 = t.V3.Position - t.V1.Position;

    Vec3 h = Cross(ray.Direction, edge2);
    float a = Dot(edge1, h);

    float f = 1.0f / a;
    Vec3 s = ray.Origin - t.V1.Position;
    float u = f * Dot(s, h);

    if (u < 0.0 || u > 1.0) return false;

    Vec3 q = Cross(s, edge1);
    float v = f * Dot(ray.Direction, q);

    if (v < 0.0f || u + v > 1.0f) return false;

    float dist = f * Dot(edge2, q);

    if (dist > 0 && dist < i.Distance) {
        i.Distance = dist;
        i.Point = ray.Origin + ray.Direction * dist;   // Ray.cs:21-24
        i.Tri = triIndex;
        i.U = u;
        i.V = v;
        return true;
    }
    return false;
}

// ------------------------------------------------------------- Accelerator
// Accelerator.cs:38-49
void Accelerator::AddTriangle(const Triangle& triangle) {
    references.push_back(triangle);
    Box box = BoxFromTriangle(triangle);
    primitives.push_back(Primitive{(int)references.size() - 1, box, BoxCentroid(box)});
}

// Accelerator.cs:179-186
void Accelerator::Clear() {
    quadRecords = -1;
    references.clear();
    primitives.clear();
    triangles.clear();
    nodes.clear();
}

// Accelerator.cs:51-111 (literal: far child is popped first, no cull on a single hit)
void Accelerator::Intersect(const Ray& ray, Intersection& i) const {
    if (nodes.empty()) return;
    if (!BoxIntersect(nodes[0].Bounds, ray, nullptr, nullptr)) return;

    std::vector<int> stack;
    stack.push_back(0);

    while (!stack.empty()) {
        const Node& node = nodes[stack.back()];
        stack.pop_back();

        if (node.Count > 0) {
            int to = node.LeftFirst + node.Count;
            for (int j = node.LeftFirst; j < to; j++) TriangleIntersect(triangles[j], ray, i, j);
        } else {
            float lt, rt;
            bool lhit = BoxIntersect(nodes[node.LeftFirst].Bounds, ray, &lt, nullptr);
            bool rhit = BoxIntersect(nodes[node.LeftFirst + 1].Bounds, ray, &rt, nullptr);

            if (lhit && rhit) {
                if (lt < rt && lt < i.Distance) {
                    stack.push_back(node.LeftFirst);
                    stack.push_back(node.LeftFirst + 1);
                } else if (rt < i.Distance) {
                    stack.push_back(node.LeftFirst + 1);
                    stack.push_back(node.LeftFirst);
                }
            } else {
                if (lhit) stack.push_back(node.LeftFirst);
                else if (rhit) stack.push_back(node.LeftFirst + 1);
            }
        }
    }
}

// Canonical closest hit (two-wide walk; see WalkWide for the one most rays take): the minimum over all accepted
// Möller–Trumbore hits, ties broken by
// the smallest triangle index.  The visiting order is the one the GPU kernels use, step for
// step, so that distance culling (which compares box entry distances with triangle
// distances, two differently rounded quantities) takes the same decisions on both sides:
//   a ray holds one current entry and a stack of entries; an entry is an interior reference
//   (index of a left child) or a leaf.  Interior step: slab-test both children; drop a child whose
//   entry distance is beyond the best hit (tmin > best; a child entered exactly AT the best
//   distance is still visited so an equal-distance triangle with a smaller index can win); of two
//   survivors the nearer becomes current and the other is pushed together with its entry
//   distance; with none, pop.  Leaf step: test its triangles in index order, then pop.
//   Pop (closest hit): entries whose recorded entry distance is now beyond the best hit are
//   discarded unvisited.
namespace {
struct Entry { bool leaf; int first; int count; int node; float tmin; };
inline Entry entryOf(const Node& n) {
    if (n.Count > 0) return Entry{true, n.LeftFirst, n.Count, 0, 0.0f};
    return Entry{false, 0, 0, n.LeftFirst, 0.0f};
}
}  // namespace

static inline bool AcceptCanonical(const Triangle& t, const Ray& ray, Intersection& best, int j) {
    Intersection probe;
    probe.Distance = INF;
    probe.Tri = -1;
    if (!TriangleIntersect(t, ray, probe, j)) return false;
    if (probe.Distance < best.Distance || (probe.Distance == best.Distance && j < best.Tri)) {
        best = probe;
        return true;
    }
    return false;
}

// Shared walk.  anyHit: stop at the first accepted hit closer than tlimit and return true.
static bool Walk(const Accelerator& acc, const Ray& ray, Intersection& i, bool anyHit, float tlimit, TraversalStats* st) {
    const auto& nodes = acc.nodes;
    const auto& triangles = acc.triangles;
    if (st) st->rays++;
    if (nodes.empty()) return false;
    if (!BoxIntersect(nodes[0].Bounds, ray, nullptr, nullptr)) return false;
    bool occluded = false;
    auto testTriangle = [&](int j) {
        if (st) st->tris++;
        if (anyHit) {
            Intersection probe;
            probe.Distance = INF;
            probe.Tri = -1;
            if (TriangleIntersect(triangles[j], ray, probe, j) && probe.Distance < tlimit) occluded = true;
        } else {
            AcceptCanonical(triangles[j], ray, i, j);
        }
    };
    std::vector<Entry> stack;
    Entry cur = entryOf(nodes[0]);
    auto pop = [&]() {
        while (!stack.empty()) {
            Entry e = stack.back();
            stack.pop_back();
            if (anyHit || !(e.tmin > i.Distance)) { cur = e; return true; }
        }
        return false;
    };
    for (;;) {
        if (cur.leaf) {
            for (int j = cur.first; j < cur.first + cur.count; j++) {
                testTriangle(j);
                if (occluded) return true;
            }
            if (!pop()) return false;
            continue;
        }
        if (st) st->steps++;
        const Node& L = nodes[cur.node];
        const Node& R = nodes[cur.node + 1];
        float lt, rt;
        bool lhit = BoxIntersect(L.Bounds, ray, &lt, nullptr);
        bool rhit = BoxIntersect(R.Bounds, ray, &rt, nullptr);
        if (anyHit) { lhit = lhit && lt < tlimit; rhit = rhit && rt < tlimit; }
        else { lhit = lhit && !(lt > i.Distance); rhit = rhit && !(rt > i.Distance); }
        Entry eL = entryOf(L), eR = entryOf(R);
        eL.tmin = lt;
        eR.tmin = rt;
        if (lhit && rhit) {
            if (rt < lt) { stack.push_back(eL); cur = eR; }
            else { stack.push_back(eR); cur = eL; }
        } else if (lhit) cur = eL;
        else if (rhit) cur = eR;
        else if (!pop()) return false;
    }
}

// A ray whose slab distances can be NaN (0 * inf): a zero, denormal or non-finite direction
// component, or a non-finite origin.  Such rays take the two-wide walk above; all others the
// wide walk below.
static inline bool RaySpecial(const Ray& ray) {
    const float d[3] = {ray.Direction.x, ray.Direction.y, ray.Direction.z};
    const float o[3] = {ray.Origin.x, ray.Origin.y, ray.Origin.z};
    for (int k = 0; k < 3; k++) {
        if (!(std::fabs(d[k]) >= 1.0e-37f && std::fabs(d[k]) <= 3.0e38f)) return true;
        if (!(std::fabs(o[k]) <= 3.0e38f)) return true;
    }
    return false;
}

// ---- the quantised walk of the GPU kernels, restated for its counters (no reference counterpart: the reference walks exact boxes;
// what the reference fixes is which leaves a ray may test — those whose own box passes Box.Intersect — and that is kept: see below).
// For trees too large for the GPU's L2 caches the kernels walk node records whose child boxes lie on a 16-bit grid over the scene,
// every plane at least one cell outside the exact box, and evaluate a plane's distance as fma(cell number, cell / d, (grid_min - o) / d).
// Such a box only steers the walk; when a leaf becomes current its exact box is tested by Box.Intersect's arithmetic together with
// the walk's distance rule, and only then are its triangles eligible.  Rays that start farther than 16 scene extents from the
// scene's middle, or with a direction component below 1e-20, take the exact two-wide walk.
namespace {
struct QGrid { float mn[3], cell[3], mid[3], reach[3]; };
QGrid GridOf(const Box& root) {
    QGrid g;
    const float rmn[3] = {root.Min.x, root.Min.y, root.Min.z}, rmx[3] = {root.Max.x, root.Max.y, root.Max.z};
    float ext[3], widest = 0.0f;
    for (int a = 0; a < 3; a++) { ext[a] = rmx[a] - rmn[a]; widest = ext[a] > widest ? ext[a] : widest; }
    const bool usable = widest >= 1.0e-10f && widest <= 1.0e15f;
    for (int a = 0; a < 3; a++) {
        float eff = ext[a] > widest * 0.015625f ? ext[a] : widest * 0.015625f;
        if (!(eff >= 1.0e-20f)) eff = 1.0e-20f;
        g.cell[a] = eff / 65000.0f;
        g.mn[a] = rmn[a] - 4.0f * g.cell[a];
        g.mid[a] = 0.5f * rmn[a] + 0.5f * rmx[a];
        g.reach[a] = usable ? 16.0f * eff : -1.0f;
    }
    return g;
}
bool RayOffGrid(const QGrid& g, const Ray& ray) {
    const float d[3] = {ray.Direction.x, ray.Direction.y, ray.Direction.z};
    const float o[3] = {ray.Origin.x, ray.Origin.y, ray.Origin.z};
    for (int k = 0; k < 3; k++) {
        if (!(std::fabs(d[k]) >= 1.0e-20f)) return true;
        if (!(std::fabs(o[k] - g.mid[k]) <= g.reach[k])) return true;
    }
    return false;
}
// entry and exit distance of a box through its grid planes
void GridRange(const QGrid& g, const Box& b, const Ray& ray, float* tmin, float* tmax) {
    const float mn[3] = {b.Min.x, b.Min.y, b.Min.z}, mx[3] = {b.Max.x, b.Max.y, b.Max.z};
    const float o[3] = {ray.Origin.x, ray.Origin.y, ray.Origin.z}, inv[3] = {ray.Inverse.x, ray.Inverse.y, ray.Inverse.z};
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
        const double ql = std::floor(((double)mn[a] - (double)g.mn[a]) / (double)g.cell[a]) - 1.0;
        const double qh = std::ceil(((double)mx[a] - (double)g.mn[a]) / (double)g.cell[a]) + 1.0;
        const float qa = g.cell[a] * inv[a], qb = (g.mn[a] - o[a]) * inv[a];
        const float t1 = std::fmaf((float)(uint32_t)ql, qa, qb), t2 = std::fmaf((float)(uint32_t)qh, qa, qb);
        lo[a] = t1 < t2 ? t1 : t2;
        hi[a] = t1 < t2 ? t2 : t1;
    }
    float m = lo[0] > lo[1] ? lo[0] : lo[1];
    *tmin = m > lo[2] ? m : lo[2];
    float x = hi[0] < hi[1] ? hi[0] : hi[1];
    *tmax = x < hi[2] ? x : hi[2];
}
// Box.Intersect plus the walk's distance rule, on "entry distance clamped at +0" as the kernels evaluate it
bool Admits(float tm, float tx, bool anyHit, float tlimit, float best) {
    const float tc = tm > 0.0f ? tm : 0.0f;
    return tc < tx && (anyHit ? tm < tlimit : !(tc > best));
}
}  // namespace

// The wide walk: the same tree with every other level skipped.  An interior entry stands for an
// interior node at an even depth; a step slab-tests its grandchildren (a child that is a leaf
// stands for itself; slot order: the left child's expansion first), drops the culled ones as in
// the two-wide walk, makes the nearest survivor current and pushes the others, farthest first.
// The order key of a child is its entry distance clamped at zero with the two lowest mantissa bits
// replaced by the slot number; pushed entries carry the key with those bits cleared as distance.
// grid != nullptr: the quantised form (above).
static bool WalkWide(const Accelerator& acc, const Ray& ray, Intersection& i, bool anyHit, float tlimit, TraversalStats* st, const QGrid* grid = nullptr) {
    const auto& nodes = acc.nodes;
    const auto& triangles = acc.triangles;
    if (st) st->rays++;
    if (nodes.empty()) return false;
    if (!BoxIntersect(nodes[0].Bounds, ray, nullptr, nullptr)) return false;
    bool occluded = false;
    auto testTriangle = [&](int j) {
        if (st) st->tris++;
        if (anyHit) {
            Intersection probe;
            probe.Distance = INF;
            probe.Tri = -1;
            if (TriangleIntersect(triangles[j], ray, probe, j) && probe.Distance < tlimit) occluded = true;
        } else {
            AcceptCanonical(triangles[j], ray, i, j);
        }
    };
    std::vector<Entry> stack;
    Entry cur = nodes[0].Count > 0 ? entryOf(nodes[0]) : Entry{false, 0, 0, 0, 0.0f};   // entries name the node itself here (leaves too: node 0 is the root)
    auto pop = [&]() {
        while (!stack.empty()) {
            Entry e = stack.back();
            stack.pop_back();
            if (anyHit || !(e.tmin > i.Distance)) { cur = e; return true; }
        }
        return false;
    };
    for (;;) {
        if (cur.leaf) {
            bool open = true;
            if (grid) {   // the leaf's own box decides; its triangles are dealt (and counted) either way
                float tm, tx;
                BoxIntersect(nodes[cur.node].Bounds, ray, &tm, &tx);
                open = Admits(tm, tx, anyHit, tlimit, i.Distance);
            }
            for (int j = cur.first; j < cur.first + cur.count; j++) {
                if (open) testTriangle(j);
                else if (st) st->tris++;
                if (occluded) return true;
            }
            if (!pop()) return false;
            continue;
        }
        if (st) st->steps++;
        int kids[4];
        int nk = 0;
        for (int ch = nodes[cur.node].LeftFirst; ch <= nodes[cur.node].LeftFirst + 1; ch++) {
            if (nodes[ch].Count > 0) kids[nk++] = ch;
            else { kids[nk++] = nodes[ch].LeftFirst; kids[nk++] = nodes[ch].LeftFirst + 1; }
        }
        uint32_t keys[4];
        int who[4];
        int nh = 0;
        for (int k = 0; k < nk; k++) {
            float tm;
            bool h;
            if (grid) {
                float tx;
                GridRange(*grid, nodes[kids[k]].Bounds, ray, &tm, &tx);
                h = Admits(tm, tx, anyHit, tlimit, i.Distance);
            } else {
                h = BoxIntersect(nodes[kids[k]].Bounds, ray, &tm, nullptr);
                h = h && (anyHit ? tm < tlimit : !(tm > i.Distance));
            }
            if (!h) continue;
            const float tk = tm > 0.0f ? tm : 0.0f;
            uint32_t bits;
            std::memcpy(&bits, &tk, 4);
            keys[nh] = (bits & ~3u) | (uint32_t)k;
            who[nh] = kids[k];
            nh++;
        }
        // ascending by key (keys are unique)
        for (int a = 1; a < nh; a++)
            for (int b = a; b > 0 && keys[b] < keys[b - 1]; b--) { std::swap(keys[b], keys[b - 1]); std::swap(who[b], who[b - 1]); }
        auto entryFor = [&](int slot) {
            const Node& n = nodes[who[slot]];
            Entry e = n.Count > 0 ? entryOf(n) : Entry{false, 0, 0, who[slot], 0.0f};
            e.node = who[slot];
            const uint32_t bits = keys[slot] & ~3u;
            std::memcpy(&e.tmin, &bits, 4);
            return e;
        };
        for (int a = nh - 1; a >= 1; a--) stack.push_back(entryFor(a));
        if (nh > 0) cur = entryFor(0);
        else if (!pop()) return false;
    }
}

// The GPU library's rule for which records its kernels walk: the quantised ones when the exact 128-byte records (one per interior
// node at an even depth) exceed the 32 MiB of L2.
bool Accelerator::WalksQuantized() const {
    if (quantized == 0) return false;
    if (quantized == 1) return !nodes.empty();
    const long long quads = quadRecords >= 0 ? quadRecords : CountQuadRecords();
    return (unsigned long long)quads * 128ull > (32ull << 20);
}

// interior nodes at an even depth
long long Accelerator::CountQuadRecords() const {
    long long quads = 0;
    std::vector<std::pair<int, int>> stack;
    if (!nodes.empty()) stack.push_back({0, 0});
    while (!stack.empty()) {
        auto [n, d] = stack.back();
        stack.pop_back();
        if (nodes[n].Count > 0) continue;
        if ((d & 1) == 0) quads++;
        stack.push_back({nodes[n].LeftFirst, d + 1});
        stack.push_back({nodes[n].LeftFirst + 1, d + 1});
    }
    return quads;
}

void Accelerator::IntersectCanonical(const Ray& ray, Intersection& i, TraversalStats* st) const {
    if (WalksQuantized()) {
        const QGrid g = GridOf(nodes[0].Bounds);
        if (RaySpecial(ray) || RayOffGrid(g, ray)) Walk(*this, ray, i, false, INF, st);
        else WalkWide(*this, ray, i, false, INF, st, &g);
        return;
    }
    if (RaySpecial(ray)) Walk(*this, ray, i, false, INF, st);
    else WalkWide(*this, ray, i, false, INF, st);
}

void Accelerator::IntersectBrute(const Ray& ray, Intersection& i) const {
    for (int j = 0; j < (int)triangles.size(); j++) AcceptCanonical(triangles[j], ray, i, j);
}

// Any accepted hit with 0 < t < tmax, visiting in the same order as IntersectCanonical;
// children whose entry distance is not below tmax are skipped.
bool Accelerator::Occluded(const Ray& ray, float tmax, TraversalStats* st) const {
    Intersection unused;
    unused.Distance = INF;
    unused.Tri = -1;
    if (WalksQuantized()) {
        const QGrid g = GridOf(nodes[0].Bounds);
        if (RaySpecial(ray) || RayOffGrid(g, ray)) return Walk(*this, ray, unused, true, tmax, st);
        return WalkWide(*this, ray, unused, true, tmax, st, &g);
    }
    if (RaySpecial(ray)) return Walk(*this, ray, unused, true, tmax, st);
    return WalkWide(*this, ray, unused, true, tmax, st);
}

int Accelerator::MaxDepth() const {
    if (nodes.empty()) return 0;
    int best = 0;
    std::vector<std::pair<int, int>> stack;
    stack.push_back({0, 1});
    while (!stack.empty()) {
        auto [n, d] = stack.back();
        stack.pop_back();
        best = std::max(best, d);
        if (nodes[n].Count <= 0) {
            stack.push_back({nodes[n].LeftFirst, d + 1});
            stack.push_back({nodes[n].LeftFirst + 1, d + 1});
        }
    }
    return best;
}

// Accelerator.cs:113-134
void Accelerator::Build() {
    nodes.clear();
    triangles.clear();

    nodes.push_back(Node{Bounds(0, (int)primitives.size()), 0, (int)primitives.size()});

    Subdivide(0);

    triangles.reserve(primitives.size());
    for (size_t i = 0; i < primitives.size(); i++) {
        int j = primitives[i].Index;
        triangles.push_back(references[j]);
    }
    quadRecords = CountQuadRecords();
}

// Accelerator.cs:136-177
void Accelerator::Subdivide(int index) {
    if (nodes[index].Count < LeafNodeSize) return;

    int p = Partition(nodes[index]);
    if (p < 0) return;

    Node parent = nodes[index];
    nodes.push_back(Node{Bounds(parent.LeftFirst, p), parent.LeftFirst, p});
    nodes.push_back(Node{Bounds(parent.LeftFirst + p, parent.Count - p), parent.LeftFirst + p, parent.Count - p});

    nodes[index].LeftFirst = (int)nodes.size() - 2;
    // the split axis is stored through a negative count: -1 x, -2 y, -3 z
    nodes[index].Count = BoxDomain(parent.Bounds) * -1 - 1;

    int left = nodes[index].LeftFirst;
    Subdivide(left);
    Subdivide(left + 1);
}

// Accelerator.cs:188-280
int Accelerator::Partition(const Node& node) {
    int domain = BoxDomain(node.Bounds);
    auto axis = [domain](const Vec3& v) { return domain == 0 ? v.x : (domain == 1 ? v.y : v.z); };

    Box bounds = CentroidBounds(node.LeftFirst, node.Count);

    float k0 = axis(bounds.Min);
    float l = axis(bounds.Max) - k0;

    if (l < Epsilon) return -1;

    float k1 = ((float)BucketCount - Epsilon) / l;

    std::vector<Primitive> bins[BucketCount];
    Box domains[BucketCount];
    for (int i = 0; i < BucketCount; i++) domains[i] = BoxNegative();

    // bin primitives
    for (int i = node.LeftFirst; i < node.LeftFirst + node.Count; i++) {
        float c = axis(primitives[i].Centroid);
        float b = (c - k0) * k1;
        int j = (int)b;
        bins[j].push_back(primitives[i]);
        domains[j] = BoxUnion(domains[j], primitives[i].Bounds);
    }

    int partition = -1;
    float cost = BoxArea(node.Bounds) * (float)node.Count;

    int ln = 0, rn = node.Count;

    // select best bin split
    for (int i = 1; i < BucketCount; i++) {
        ln += (int)bins[i - 1].size();
        rn -= (int)bins[i - 1].size();

        Box lb = BoxNegative();
        Box rb = BoxNegative();

        for (int j = 0; j < i; j++) lb = BoxUnion(domains[j], lb);
        for (int j = i; j < BucketCount; j++) rb = BoxUnion(domains[j], rb);

        float c = (float)ln * BoxArea(lb) + (float)rn * BoxArea(rb);

        if (c < cost) {
            partition = ln;
            cost = c;
        }
    }

    if (partition < 0) return -1;

    int p = node.LeftFirst;
    for (int i = 0; i < BucketCount; i++)
        for (size_t j = 0; j < bins[i].size(); j++) primitives[p++] = bins[i][j];

    return partition;
}

// Accelerator.cs:282-292
Box Accelerator::Bounds(int first, int count) const {
    Box bounds = BoxNegative();
    for (int i = first; i < first + count; i++) bounds = BoxUnion(primitives[i].Bounds, bounds);
    return bounds;
}

// Accelerator.cs:294-310
Box Accelerator::CentroidBounds(int first, int count) const {
    Vec3 min{INF, INF, INF};
    Vec3 max{-INF, -INF, -INF};
    for (int i = first; i < first + count; i++) {
        min = Vec3Min(primitives[i].Centroid, min);
        max = Vec3Max(primitives[i].Centroid, max);
    }
    Box b;
    b.Max = max;
    b.Min = min;
    return b;
}

}  // namespace oracle
