// Host mirror of PathTracer/Accelerator.cs (see accelerator.hpp for the design).
#include "accelerator.hpp"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <thread>

namespace pth {

namespace {

const float kEpsilon = 0.00001f;   // Accelerator.cs:10
const int kLeafNodeSize = 3;       // Accelerator.cs:11
const int kBuckets = 8;            // Accelerator.cs:12
const float kInf = std::numeric_limits<float>::infinity();

inline Box negativeBox() { return Box{{kInf, kInf, kInf}, {-kInf, -kInf, -kInf}}; }   // Box.cs:14-18

// Box.Union(a, b), Box.cs:81-88: Vector3.Max(a,b) = (a > b) ? a : b, Vector3.Min(a,b) = (a < b) ? a : b
inline Box unionOf(const Box& a, const Box& b) {
    Box r;
    r.Max.x = a.Max.x > b.Max.x ? a.Max.x : b.Max.x;
    r.Max.y = a.Max.y > b.Max.y ? a.Max.y : b.Max.y;
    r.Max.z = a.Max.z > b.Max.z ? a.Max.z : b.Max.z;
    r.Min.x = a.Min.x < b.Min.x ? a.Min.x : b.Min.x;
    r.Min.y = a.Min.y < b.Min.y ? a.Min.y : b.Min.y;
    r.Min.z = a.Min.z < b.Min.z ? a.Min.z : b.Min.z;
    return r;
}

// MathF.Max (IEEE maximum, NaN-propagating) as used by Vector.MaxValue, Vector.cs:28-31
inline float mathfMax(float x, float y) {
    if (x > y || x != x) return x;
    if (x == y) return std::signbit(x) ? y : x;
    return y;
}

// Box.Domain, Box.cs:25-43: longest axis, ties X before Y before Z
inline int domainOf(const Box& b) {
    const float dx = b.Max.x - b.Min.x, dy = b.Max.y - b.Min.y, dz = b.Max.z - b.Min.z;
    const float l = mathfMax(mathfMax(dx, dy), dz);
    if (l == dx) return 0;
    if (l == dy) return 1;
    return 2;
}

// Box.Area, Box.cs:45-49
inline float areaOf(const Box& b) {
    const float dx = b.Max.x - b.Min.x, dy = b.Max.y - b.Min.y, dz = b.Max.z - b.Min.z;
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}

inline float axisOf(const Vec3& v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

}  // namespace

void Accelerator::AddTriangle(const Triangle& t) {
    references.push_back(t);
    const Vec3 &a = t.V1.Position, &b = t.V2.Position, &c = t.V3.Position;
    // Box.FromVertices, Box.cs:95-102: Max(Max(a,b),c) + 1e-5, Min(Min(a,b),c) - 1e-5
    Box box;
    float m;
    m = a.x > b.x ? a.x : b.x; box.Max.x = (m > c.x ? m : c.x) + 0.00001f;
    m = a.y > b.y ? a.y : b.y; box.Max.y = (m > c.y ? m : c.y) + 0.00001f;
    m = a.z > b.z ? a.z : b.z; box.Max.z = (m > c.z ? m : c.z) + 0.00001f;
    m = a.x < b.x ? a.x : b.x; box.Min.x = (m < c.x ? m : c.x) - 0.00001f;
    m = a.y < b.y ? a.y : b.y; box.Min.y = (m < c.y ? m : c.y) - 0.00001f;
    m = a.z < b.z ? a.z : b.z; box.Min.z = (m < c.z ? m : c.z) - 0.00001f;
    boxes.push_back(box);
    // Box.Centroid, Box.cs:20-23
    centroids.push_back(Vec3{0.5f * box.Min.x + 0.5f * box.Max.x, 0.5f * box.Min.y + 0.5f * box.Max.y, 0.5f * box.Min.z + 0.5f * box.Max.z});
    order.push_back((uint32_t)references.size() - 1);
}

void Accelerator::Clear() {
    references.clear();
    boxes.clear();
    centroids.clear();
    order.clear();
    triangles.clear();
    nodes.clear();
}

// Accelerator.Bounds, Accelerator.cs:282-292: bounds = Union(primitive, bounds) in list order
Box Accelerator::boundsOf(int first, int count) const {
    Box bounds = negativeBox();
    for (int i = first; i < first + count; i++) bounds = unionOf(boxes[order[i]], bounds);
    return bounds;
}

// Accelerator.Parition, Accelerator.cs:188-280.  Returns false where the reference returns -1.
// On success order[first, first+count) is rewritten bin by bin (stable within a bin).
bool Accelerator::partition(const Box& bounds, int first, int count, Scratch& sc, Split* out) {
    const int axis = domainOf(bounds);

    // CentroidBounds along the split axis, Accelerator.cs:294-310
    float cmin = kInf, cmax = -kInf;
    for (int i = first; i < first + count; i++) {
        const float c = axisOf(centroids[order[i]], axis);
        cmin = c < cmin ? c : cmin;
        cmax = c > cmax ? c : cmax;
    }
    const float k0 = cmin;
    const float l = cmax - k0;
    if (l < kEpsilon) return false;
    const float k1 = ((float)kBuckets - kEpsilon) / l;

    int counts[kBuckets] = {0, 0, 0, 0, 0, 0, 0, 0};
    Box domains[kBuckets];
    for (int j = 0; j < kBuckets; j++) domains[j] = negativeBox();
    if ((int)sc.bin.size() < count) sc.bin.resize(count);
    for (int i = 0; i < count; i++) {
        const uint32_t id = order[first + i];
        const float c = axisOf(centroids[id], axis);
        const float b = (c - k0) * k1;
        // a non-finite vertex makes b NaN or infinite: the reference's buckets[j] raises IndexOutOfRangeException there
        // (Accelerator.cs:218-222); here the C boundary turns the exception into an error return
        if (!(b >= 0.0f && b < (float)kBuckets)) throw std::runtime_error("Accelerator.Build: a primitive's centroid is not finite");
        const int j = (int)b;
        sc.bin[i] = (uint8_t)j;
        counts[j]++;
        domains[j] = unionOf(domains[j], boxes[id]);
    }

    int best = -1;
    float cost = areaOf(bounds) * (float)count;
    int ln = 0, rn = count;
    Box lb = negativeBox();
    for (int i = 1; i < kBuckets; i++) {
        ln += counts[i - 1];
        rn -= counts[i - 1];
        lb = unionOf(domains[i - 1], lb);          // the fold over j < i, one step further
        Box rb = negativeBox();
        for (int j = i; j < kBuckets; j++) rb = unionOf(domains[j], rb);
        const float c = (float)ln * areaOf(lb) + (float)rn * areaOf(rb);
        if (c < cost) {
            best = ln;
            cost = c;
        }
    }
    if (best < 0) return false;

    // rewrite the range bin by bin, keeping the order inside each bin
    int offset[kBuckets];
    int acc = 0;
    for (int j = 0; j < kBuckets; j++) { offset[j] = acc; acc += counts[j]; }
    if ((int)sc.tmp.size() < count) sc.tmp.resize(count);
    for (int i = 0; i < count; i++) sc.tmp[offset[sc.bin[i]]++] = order[first + i];
    std::copy(sc.tmp.begin(), sc.tmp.begin() + count, order.begin() + first);

    out->count_left = best;
    out->left = boundsOf(first, best);
    out->right = boundsOf(first + best, count - best);
    return true;
}

// Accelerator.Subdivide, Accelerator.cs:136-177, with an explicit stack.  `out` receives the
// subtree in the reference's order (a node's pair, then everything below the left child, then
// everything below the right child) with indices relative to out[0], the subtree root.
void Accelerator::buildSubtree(std::vector<Node>& out, int first, int count, const Box& bounds, Scratch& sc) {
    out.push_back(Node{bounds, first, count});
    std::vector<int> stack;
    stack.push_back(0);
    while (!stack.empty()) {
        const int idx = stack.back();
        stack.pop_back();
        const Node n = out[idx];
        if (n.Count < kLeafNodeSize) continue;
        Split sp;
        if (!partition(n.Bounds, n.LeftFirst, n.Count, sc, &sp)) continue;
        const int l = (int)out.size();
        out.push_back(Node{sp.left, n.LeftFirst, sp.count_left});
        out.push_back(Node{sp.right, n.LeftFirst + sp.count_left, n.Count - sp.count_left});
        out[idx].LeftFirst = l;
        out[idx].Count = domainOf(n.Bounds) * -1 - 1;   // split axis through a negative count
        stack.push_back(l + 1);
        stack.push_back(l);
    }
}

void Accelerator::Build(int threads) {
    nodes.clear();
    triangles.clear();
    const int n = (int)references.size();
    const Box rootBounds = boundsOf(0, n);

    if (threads <= 1 || n < 65536) {
        Scratch sc;
        nodes.reserve((size_t)n * 2);
        buildSubtree(nodes, 0, n, rootBounds, sc);
    } else {
        // 1. split the top of the tree on this thread until there are enough independent subtrees
        struct Top { int first, count; Box bounds; int left, right; bool closed; std::vector<Node> local; };
        std::vector<Top> top;
        top.push_back(Top{0, n, rootBounds, -1, -1, false, {}});
        Scratch sc;
        const int wantTasks = threads * 8;
        const int minTask = std::max(4096, n / (wantTasks * 4));
        int open = 1;
        while (open < wantTasks) {
            int pick = -1;
            for (int i = 0; i < (int)top.size(); i++)
                if (top[i].left < 0 && !top[i].closed && top[i].count > minTask && (pick < 0 || top[i].count > top[pick].count)) pick = i;
            if (pick < 0) break;
            Split sp;
            const Top t = top[pick];
            if (!partition(t.bounds, t.first, t.count, sc, &sp)) { top[pick].closed = true; continue; }
            top[pick].left = (int)top.size();
            top[pick].right = (int)top.size() + 1;
            top.push_back(Top{t.first, sp.count_left, sp.left, -1, -1, false, {}});
            top.push_back(Top{t.first + sp.count_left, t.count - sp.count_left, sp.right, -1, -1, false, {}});
            open++;
        }
        // 2. build the subtrees concurrently; each works on its own slice of `order`
        std::vector<int> tasks;
        for (int i = 0; i < (int)top.size(); i++) if (top[i].left < 0) tasks.push_back(i);
        std::sort(tasks.begin(), tasks.end(), [&](int a, int b) { return top[a].count > top[b].count; });
        std::atomic<int> next{0};
        auto worker = [&]() {
            Scratch local;
            for (;;) {
                const int k = next.fetch_add(1);
                if (k >= (int)tasks.size()) break;
                Top& t = top[tasks[k]];
                t.local.reserve((size_t)t.count * 2);
                buildSubtree(t.local, t.first, t.count, t.bounds, local);
            }
        };
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; i++) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
        // 3. lay the pieces out in the reference's node order
        nodes.reserve((size_t)n * 2);
        nodes.push_back(Node{});
        struct Emit { int topIdx, at; };
        std::vector<Emit> stack;
        stack.push_back({0, 0});
        while (!stack.empty()) {
            const Emit e = stack.back();
            stack.pop_back();
            Top& t = top[e.topIdx];
            if (t.left < 0) {
                const int base = (int)nodes.size();
                for (size_t k = 0; k < t.local.size(); k++) {
                    Node nd = t.local[k];
                    if (nd.Count < 0) nd.LeftFirst = base + nd.LeftFirst - 1;
                    if (k == 0) nodes[e.at] = nd; else nodes.push_back(nd);
                }
                std::vector<Node>().swap(t.local);
                // everything below the left sibling must be emitted before the right sibling's
                // subtree: guaranteed because the right sibling is deeper in the stack
            } else {
                const int p = (int)nodes.size();
                nodes.push_back(Node{});
                nodes.push_back(Node{});
                nodes[e.at] = Node{t.bounds, p, domainOf(t.bounds) * -1 - 1};
                stack.push_back({t.right, p + 1});
                stack.push_back({t.left, p});
            }
        }
    }

    triangles.resize((size_t)n);
    for (int i = 0; i < n; i++) triangles[i] = references[order[i]];   // Accelerator.cs:129-133
}

int Accelerator::Depth() const {
    if (nodes.empty()) return 0;
    int best = 0;
    std::vector<std::pair<int, int>> stack;
    stack.push_back({0, 1});
    while (!stack.empty()) {
        const auto [i, d] = stack.back();
        stack.pop_back();
        best = std::max(best, d);
        if (nodes[i].Count < 0) {
            stack.push_back({nodes[i].LeftFirst, d + 1});
            stack.push_back({nodes[i].LeftFirst + 1, d + 1});
        }
    }
    return best;
}

}  // namespace pth
