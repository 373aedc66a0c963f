// Binned-SAH BVH construction on the GPU: the same tree, node for node and byte for byte, as
// PathTracer/Accelerator.cs:113-310 builds on one CPU thread.
//
// What makes the parallel result identical:
//   * every quantity the reference folds over a primitive list is a minimum, a maximum or a count
//     (Accelerator.cs:282-310, Box.cs:81-88), so atomics in any order give the same bits (floats go
//     through an order-preserving integer map; bounds carry no negative zeros or NaNs);
//   * the split decision of a node (Accelerator.cs:188-262) is evaluated by one thread with the
//     reference's expression order (this file is compiled without contraction);
//   * the reference's in-place partition leaves the primitives of a node bin by bin, in their previous
//     order inside a bin: a stable sort of the node's range on the 3-bit bin number — rocPRIM's
//     segmented radix sort for the many small nodes, and for nodes of PTB_CHUNK primitives or more a
//     chunked counting sort (count per chunk, scan the chunk counts per node, scatter with stable
//     ranks from wave ballots) that keeps every CU busy on the few huge nodes at the top of the tree;
//   * the tree is built level by level with arbitrary node numbers and renumbered at the end into the
//     reference's order: a node's two children, then everything below the left child, then everything
//     below the right child (Accelerator.cs:136-177);
//   * a node of at most PTB_SMALL primitives leaves the level-by-level scheme: one thread builds everything
//     below it the way the reference's recursion does (Subdivide, left before right), which hands out the
//     reference's relative node numbers as it goes.  The deep levels of a large tree are millions of tiny
//     nodes; per level they cost a pass over all primitives and global atomics per primitive, per thread
//     they cost a few dozen primitives that stay in cache.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <cstring>
#include <vector>
#include <rocprim/device/device_segmented_radix_sort.hpp>

// Every device allocation of the library goes through here, so that the number of allocations a call made can be read back
// (pt_stats.device_allocations): that is how a test shows that an entry point allocates nothing.
inline std::atomic<unsigned long long>& ptAllocCount() { static std::atomic<unsigned long long> n{0}; return n; }
template <class T>
inline hipError_t ptMalloc(T** p, size_t bytes) { ptAllocCount().fetch_add(1, std::memory_order_relaxed); return hipMalloc((void**)p, bytes); }

namespace ptbuild {

#define PTB_BLOCK 256
#define PTB_NONE 0xFFFFFFFFu
#define PTB_EPSILON 0.00001f   // Accelerator.cs:10
#define PTB_LEAF 3             // Accelerator.cs:11
#define PTB_BINS 8             // Accelerator.cs:12
#define PTB_CHUNK 2048         // primitives per block of the chunked partition; nodes at least this large take it
#ifndef PTB_SMALL
#define PTB_SMALL 64           // nodes of at most this many primitives are finished by one thread each
#endif
#define PTB_COPIES 16          // private copies of a block's bin table in LDS (lane & 15 picks one): a sixteenth of the conflicts

struct Tri112 { float v[3][8]; int32_t material; int32_t pad[3]; };   // Triangle.cs:7-13 (three 32-byte vertices)
struct Node32 { float mn[3], mx[3]; int32_t leftFirst, count; };      // Node.cs:5-10

// order-preserving float <-> uint map (no NaNs here)
__host__ __device__ inline uint32_t fkey(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float fval(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f;
    memcpy(&f, &b, 4);
    return f;
}
#define PTB_KEY_POS_INF 0xFF800000u   // fkey(+inf)
#define PTB_KEY_NEG_INF 0x007FFFFFu   // fkey(-inf)

// node under construction (numbered in creation order)
struct BNode {
    float mn[3], mx[3];
    int32_t first, count;
    int32_t left;        // creation index of the left child (right = left + 1), -1 for a leaf
    int32_t axis;        // split axis of an interior node
    uint32_t next_rank;  // rank in the next level's active list, PTB_NONE if it will not be split further
    uint32_t below;      // nodes below this one
    uint32_t block;      // final index of its children pair
    uint32_t index;      // final index of the node
    uint32_t level;      // depth (root 0)
    uint32_t aux;        // PTB_NONE; for the root of a thread-built subtree its own creation index; for the nodes below it that root's
};

// per active node of the running level
struct Split {
    uint32_t node;
    int32_t axis;
    uint32_t cmin, cmax;              // centroid range along the axis (keys)
    float k0, k1;
    int32_t usable;                   // 0: becomes a leaf (Accelerator.cs:203-204 or :262)
    uint32_t split_pos;               // first + primitives on the left
    uint32_t chunk_base, chunks;      // its chunks in the level's chunk list (chunks == 0: sorted by the segmented sort)
    uint32_t cnt[PTB_BINS];
    uint32_t bmn[PTB_BINS][3], bmx[PTB_BINS][3];   // bin domains (keys)
};

struct Counters { uint32_t nodes, next_active, sort_segments, large, chunks, small, max_level, pad; };

// MathF.Max (NaN-propagating IEEE maximum), Vector.cs:28-31
__device__ inline float mathfMax(float x, float y) {
    if (x > y || x != x) return x;
    if (x == y) return (__float_as_uint(x) >> 31) ? y : x;
    return y;
}
// Box.Domain, Box.cs:25-43
__device__ inline int domainOf(const float* mn, const float* mx) {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    const float l = mathfMax(mathfMax(dx, dy), dz);
    if (l == dx) return 0;
    if (l == dy) return 1;
    return 2;
}
// Box.Area, Box.cs:45-49
__device__ inline float areaOf(const float* mn, const float* mx) {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}

// Box.FromVertices + Box.Centroid per primitive (Box.cs:95-102, :20-23); root bounds by atomics
__global__ void __launch_bounds__(PTB_BLOCK) k_prims(const Tri112* tris, uint32_t n, float4* pmin, float4* pmax, float4* cent, uint32_t* order,
                                                     uint32_t* seg, uint32_t* root_keys) {
    __shared__ uint32_t s[6];
    if (threadIdx.x < 3) s[threadIdx.x] = PTB_KEY_POS_INF;
    else if (threadIdx.x < 6) s[threadIdx.x] = PTB_KEY_NEG_INF;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const Tri112& t = tris[i];
        float mn[3], mx[3];
        for (int a = 0; a < 3; a++) {
            const float x = t.v[0][a], y = t.v[1][a], z = t.v[2][a];
            float m = x > y ? x : y;
            mx[a] = (m > z ? m : z) + 0.00001f;
            m = x < y ? x : y;
            mn[a] = (m < z ? m : z) - 0.00001f;
        }
        pmin[i] = make_float4(mn[0], mn[1], mn[2], 0.0f);
        pmax[i] = make_float4(mx[0], mx[1], mx[2], 0.0f);
        cent[i] = make_float4(0.5f * mn[0] + 0.5f * mx[0], 0.5f * mn[1] + 0.5f * mx[1], 0.5f * mn[2] + 0.5f * mx[2], 0.0f);
        order[i] = i;
        seg[i] = 0u;
        for (int a = 0; a < 3; a++) {
            atomicMin(&s[a], fkey(mn[a]));
            atomicMax(&s[3 + a], fkey(mx[a]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&root_keys[threadIdx.x], s[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&root_keys[threadIdx.x], s[threadIdx.x]);
}

__global__ void k_root(BNode* nodes, const uint32_t* root_keys, uint32_t n, Counters* ctr, uint32_t* active, uint32_t* small) {
    BNode r{};
    for (int a = 0; a < 3; a++) {
        r.mn[a] = n ? fval(root_keys[a]) : __builtin_huge_valf();       // Box.Negative, Box.cs:14-18
        r.mx[a] = n ? fval(root_keys[3 + a]) : -__builtin_huge_valf();
    }
    r.first = 0;
    r.count = (int32_t)n;
    r.left = -1;
    r.level = 0;
    const bool split = n >= PTB_LEAF, tiny = split && n <= PTB_SMALL;
    r.next_rank = (split && !tiny) ? 0u : PTB_NONE;
    r.aux = tiny ? 0u : PTB_NONE;
    nodes[0] = r;
    ctr->nodes = 1;
    ctr->next_active = (split && !tiny) ? 1u : 0u;
    ctr->sort_segments = 0;
    ctr->small = tiny ? 1u : 0u;
    ctr->max_level = 0;
    active[0] = 0;
    small[0] = 0;
}

// per active node: split axis, empty centroid range
__global__ void __launch_bounds__(PTB_BLOCK) k_split_begin(const BNode* nodes, const uint32_t* active, uint32_t n_active, Split* sp) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_active) return;
    const BNode& nd = nodes[active[r]];
    Split& s = sp[r];
    s.node = active[r];
    s.axis = domainOf(nd.mn, nd.mx);
    s.cmin = PTB_KEY_POS_INF;
    s.cmax = PTB_KEY_NEG_INF;
}

__device__ inline float axisOf(float4 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// Accelerator.CentroidBounds along the split axis, Accelerator.cs:294-310.  A block walks `tiles` consecutive tiles of 256
// positions and keeps the range of the node its current tile starts in in LDS; it goes to global memory only when that node
// changes or the block ends, so the handful of nodes at the top of the tree do not receive an atomic from every tile of the
// array (same-address atomics serialise: 78,000 of them cost 4 ms per level on 20M primitives).  A wave whose lanes all sit in
// one node reduces its 64 keys with lane exchanges first.
__global__ void __launch_bounds__(PTB_BLOCK) k_centroid_range(const float4* cent, const uint32_t* order, const uint32_t* seg, uint32_t n, Split* sp, uint32_t tiles) {
    __shared__ uint32_t s_min, s_max, s_rank;
    if (threadIdx.x == 0) { s_min = PTB_KEY_POS_INF; s_max = PTB_KEY_NEG_INF; s_rank = PTB_NONE; }
    __syncthreads();
    for (uint32_t tile = 0; tile < tiles; tile++) {
        const uint32_t base = (blockIdx.x * tiles + tile) * PTB_BLOCK;
        if (base >= n) break;
        const uint32_t rank0 = seg[base];
        if (rank0 != s_rank) {            // block-uniform: flush what was gathered for the previous node
            __syncthreads();
            if (threadIdx.x == 0) {
                if (s_rank != PTB_NONE && s_min != PTB_KEY_POS_INF) { atomicMin(&sp[s_rank].cmin, s_min); atomicMax(&sp[s_rank].cmax, s_max); }
                s_min = PTB_KEY_POS_INF; s_max = PTB_KEY_NEG_INF; s_rank = rank0;
            }
            __syncthreads();
        }
        const uint32_t i = base + threadIdx.x;
        const uint32_t r = i < n ? seg[i] : PTB_NONE;
        const bool valid = r != PTB_NONE;
        const uint32_t k = valid ? fkey(axisOf(cent[order[i]], sp[r].axis)) : 0u;
        const unsigned long long vm = __ballot(valid);
        if (vm != 0ull) {
            const int firstLane = (int)__ffsll((long long)vm) - 1;
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)r, firstLane);
            if (__ballot(valid && r == r0) == vm) {
                uint32_t lo = valid ? k : PTB_KEY_POS_INF, hi = valid ? k : PTB_KEY_NEG_INF;
                for (int off = 32; off > 0; off >>= 1) {
                    const uint32_t a = (uint32_t)__shfl_xor((int)lo, off), b2 = (uint32_t)__shfl_xor((int)hi, off);
                    lo = a < lo ? a : lo;
                    hi = b2 > hi ? b2 : hi;
                }
                if ((int)(threadIdx.x & 63u) == firstLane) {
                    if (r0 == rank0) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
                    else { atomicMin(&sp[r0].cmin, lo); atomicMax(&sp[r0].cmax, hi); }
                }
            } else if (valid) {
                if (r == rank0) { atomicMin(&s_min, k); atomicMax(&s_max, k); }
                else { atomicMin(&sp[r].cmin, k); atomicMax(&sp[r].cmax, k); }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_rank != PTB_NONE && s_min != PTB_KEY_POS_INF) {
        atomicMin(&sp[s_rank].cmin, s_min);
        atomicMax(&sp[s_rank].cmax, s_max);
    }
}

// Accelerator.cs:196-213: k0, k1, empty bins
__global__ void __launch_bounds__(PTB_BLOCK) k_split_bins(uint32_t n_active, Split* sp) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_active) return;
    Split& s = sp[r];
    const float k0 = fval(s.cmin);
    const float l = fval(s.cmax) - k0;
    s.k0 = k0;
    s.usable = l < PTB_EPSILON ? 0 : 1;
    s.k1 = ((float)PTB_BINS - PTB_EPSILON) / l;
    for (int j = 0; j < PTB_BINS; j++) {
        s.cnt[j] = 0;
        for (int a = 0; a < 3; a++) { s.bmn[j][a] = PTB_KEY_POS_INF; s.bmx[j][a] = PTB_KEY_NEG_INF; }
    }
}

// Accelerator.cs:215-227: bin of every primitive, bin counts and domains, gathered in LDS before they go to the nodes' global
// counters.  A node's positions are consecutive, so a tile of 256 positions whose first and last position carry the same node
// lies inside it: such tiles (all of them near the top of the tree) add to one table that the block carries from tile to tile,
// kept PTB_COPIES times so that 256 threads do not queue on the same eight counters, and written out when the node changes.
// Any other tile numbers the runs of equal nodes it contains (at most PTB_COPIES of them get a table entry each; a node further
// on in a tile of tiny nodes falls back to global atomics) and writes its entries out when it ends.
__global__ void __launch_bounds__(PTB_BLOCK) k_bin(const float4* cent, const float4* pmin, const float4* pmax, const uint32_t* order, const uint32_t* seg,
                                                   uint32_t n, Split* sp, uint8_t* bin_of, uint32_t tiles) {
    __shared__ uint32_t s_cnt[PTB_COPIES][PTB_BINS], s_mn[PTB_COPIES][PTB_BINS][3], s_mx[PTB_COPIES][PTB_BINS][3];
    __shared__ uint32_t s_entryRank[PTB_COPIES], s_waveRuns[PTB_BLOCK / 64];
    auto clearTable = [&]() {
        for (uint32_t t = threadIdx.x; t < PTB_COPIES * PTB_BINS; t += PTB_BLOCK) {
            const uint32_t c = t / PTB_BINS, j = t % PTB_BINS;
            s_cnt[c][j] = 0;
            for (int a = 0; a < 3; a++) { s_mn[c][j][a] = PTB_KEY_POS_INF; s_mx[c][j][a] = PTB_KEY_NEG_INF; }
        }
    };
    auto flushCarried = [&](uint32_t rank) {   // the PTB_COPIES entries are copies of one node's table
        if (threadIdx.x < PTB_BINS) {
            const int j = threadIdx.x;
            uint32_t cnt = 0, mn[3] = {PTB_KEY_POS_INF, PTB_KEY_POS_INF, PTB_KEY_POS_INF}, mx[3] = {PTB_KEY_NEG_INF, PTB_KEY_NEG_INF, PTB_KEY_NEG_INF};
            for (int c = 0; c < PTB_COPIES; c++) {
                cnt += s_cnt[c][j];
                for (int a = 0; a < 3; a++) { mn[a] = s_mn[c][j][a] < mn[a] ? s_mn[c][j][a] : mn[a]; mx[a] = s_mx[c][j][a] > mx[a] ? s_mx[c][j][a] : mx[a]; }
            }
            if (cnt != 0) {
                Split& g = sp[rank];
                atomicAdd(&g.cnt[j], cnt);
                for (int a = 0; a < 3; a++) { atomicMin(&g.bmn[j][a], mn[a]); atomicMax(&g.bmx[j][a], mx[a]); }
            }
        }
    };
    auto flushRuns = [&]() {                   // entry e is the table of the tile's e-th run
        if (threadIdx.x < PTB_COPIES * PTB_BINS) {
            const uint32_t e = threadIdx.x / PTB_BINS, j = threadIdx.x % PTB_BINS;
            const uint32_t cnt = s_cnt[e][j];
            if (cnt != 0) {
                Split& g = sp[s_entryRank[e]];
                atomicAdd(&g.cnt[j], cnt);
                for (int a = 0; a < 3; a++) { atomicMin(&g.bmn[j][a], s_mn[e][j][a]); atomicMax(&g.bmx[j][a], s_mx[e][j][a]); }
            }
        }
    };
    clearTable();
    uint32_t carried = PTB_NONE;   // block-uniform: the node whose table the block is carrying
    __syncthreads();
    for (uint32_t tile = 0; tile < tiles; tile++) {
        const uint32_t base = (blockIdx.x * tiles + tile) * PTB_BLOCK;
        if (base >= n) break;
        const uint32_t lastPos = base + PTB_BLOCK - 1u < n ? base + PTB_BLOCK - 1u : n - 1u;
        const uint32_t rank0 = seg[base];
        const bool inside = rank0 != PTB_NONE && seg[lastPos] == rank0;   // block-uniform
        if (carried != PTB_NONE && !(inside && rank0 == carried)) {
            __syncthreads();
            flushCarried(carried);
            __syncthreads();
            clearTable();
            __syncthreads();
            carried = PTB_NONE;
        }
        const uint32_t i = base + threadIdx.x;
        const uint32_t r = i < n ? seg[i] : PTB_NONE;
        uint32_t entry = threadIdx.x & (PTB_COPIES - 1u);
        if (inside) {
            carried = rank0;
        } else {
            const bool starts = threadIdx.x == 0u || r != (i - 1u < n ? seg[i - 1u] : PTB_NONE);
            const unsigned long long m = __ballot(starts);
            const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
            if (lane == 0u) s_waveRuns[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            entry = (uint32_t)__popcll(m & ((2ull << lane) - 1ull)) - 1u;
            for (uint32_t w = 0; w < wave; w++) entry += s_waveRuns[w];
            if (starts && entry < PTB_COPIES) s_entryRank[entry] = r;
        }
        if (r != PTB_NONE && sp[r].usable) {
            const Split& s = sp[r];
            const uint32_t id = order[i];
            const float c = axisOf(cent[id], s.axis);
            const float b = (c - s.k0) * s.k1;
            int j = (int)b;
            j = j < 0 ? 0 : (j > PTB_BINS - 1 ? PTB_BINS - 1 : j);   // only non-finite input can leave the range (the C# build throws there)
            bin_of[i] = (uint8_t)j;
            const float4 lo = pmin[id], hi = pmax[id];
            const uint32_t klo[3] = {fkey(lo.x), fkey(lo.y), fkey(lo.z)}, khi[3] = {fkey(hi.x), fkey(hi.y), fkey(hi.z)};
            if (entry < PTB_COPIES) {
                atomicAdd(&s_cnt[entry][j], 1u);
                for (int a = 0; a < 3; a++) { atomicMin(&s_mn[entry][j][a], klo[a]); atomicMax(&s_mx[entry][j][a], khi[a]); }
            } else {
                Split& g = sp[r];
                atomicAdd(&g.cnt[j], 1u);
                for (int a = 0; a < 3; a++) { atomicMin(&g.bmn[j][a], klo[a]); atomicMax(&g.bmx[j][a], khi[a]); }
            }
        }
        if (!inside) {
            __syncthreads();
            flushRuns();
            __syncthreads();
            clearTable();
            __syncthreads();
        }
    }
    __syncthreads();
    if (carried != PTB_NONE) flushCarried(carried);
}

// Box.Union(a, b) with a = a bin's domain, b = the running box (Box.cs:81-88)
__device__ inline void unionInto(float* mn, float* mx, const uint32_t* kmn, const uint32_t* kmx) {
    for (int a = 0; a < 3; a++) {
        const float lo = fval(kmn[a]), hi = fval(kmx[a]);
        mx[a] = hi > mx[a] ? hi : mx[a];
        mn[a] = lo < mn[a] ? lo : mn[a];
    }
}

// Accelerator.cs:229-262 (the sweep over the 7 split planes) and Accelerator.cs:150-176 (the two children)
__global__ void __launch_bounds__(PTB_BLOCK) k_split_decide(BNode* nodes, uint32_t n_active, Split* sp, Counters* ctr, uint32_t* next_active,
                                                            uint32_t* sort_begin, uint32_t* sort_end, uint32_t* large_rank, uint32_t* small) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_active) return;
    Split& s = sp[r];
    BNode& nd = nodes[s.node];
    if (!s.usable) return;
    const float inf = __builtin_huge_valf();
    const int count = nd.count;
    int best = -1, bestPlane = 0;
    float cost = areaOf(nd.mn, nd.mx) * (float)count;
    int ln = 0, rn = count;
    float lmn[3] = {inf, inf, inf}, lmx[3] = {-inf, -inf, -inf};
    for (int i = 1; i < PTB_BINS; i++) {
        ln += (int)s.cnt[i - 1];
        rn -= (int)s.cnt[i - 1];
        unionInto(lmn, lmx, s.bmn[i - 1], s.bmx[i - 1]);
        float rmn[3] = {inf, inf, inf}, rmx[3] = {-inf, -inf, -inf};
        for (int j = i; j < PTB_BINS; j++) unionInto(rmn, rmx, s.bmn[j], s.bmx[j]);
        const float c = (float)ln * areaOf(lmn, lmx) + (float)rn * areaOf(rmn, rmx);
        if (c < cost) {
            best = ln;
            bestPlane = i;
            cost = c;
        }
    }
    if (best < 0) { s.usable = 0; return; }
    // children: bounds of the primitives on each side (Accelerator.cs:264-279) = union of their bins' domains
    BNode L{}, R{};
    for (int a = 0; a < 3; a++) { L.mn[a] = inf; L.mx[a] = -inf; R.mn[a] = inf; R.mx[a] = -inf; }
    for (int j = 0; j < PTB_BINS; j++) {
        if (s.cnt[j] == 0) continue;
        if (j < bestPlane) unionInto(L.mn, L.mx, s.bmn[j], s.bmx[j]);
        else unionInto(R.mn, R.mx, s.bmn[j], s.bmx[j]);
    }
    L.first = nd.first;
    L.count = best;
    R.first = nd.first + best;
    R.count = count - best;
    L.left = R.left = -1;
    L.level = R.level = nd.level + 1u;
    const uint32_t at = atomicAdd(&ctr->nodes, 2u);
    // a child goes on: to the next level's active list, or — small enough — to the list of subtrees one thread finishes
    L.next_rank = R.next_rank = PTB_NONE;
    L.aux = R.aux = PTB_NONE;
    if (L.count >= PTB_LEAF) {
        if (L.count <= PTB_SMALL) { L.aux = at; small[atomicAdd(&ctr->small, 1u)] = at; }
        else L.next_rank = atomicAdd(&ctr->next_active, 1u);
    }
    if (R.count >= PTB_LEAF) {
        if (R.count <= PTB_SMALL) { R.aux = at + 1u; small[atomicAdd(&ctr->small, 1u)] = at + 1u; }
        else R.next_rank = atomicAdd(&ctr->next_active, 1u);
    }
    nodes[at] = L;
    nodes[at + 1] = R;
    if (L.next_rank != PTB_NONE) next_active[L.next_rank] = at;
    if (R.next_rank != PTB_NONE) next_active[R.next_rank] = at + 1;
    nd.left = (int32_t)at;
    nd.axis = s.axis;
    s.split_pos = (uint32_t)(nd.first + best);
    s.chunks = 0;
    if (count >= PTB_CHUNK) {
        s.chunks = ((uint32_t)count + PTB_CHUNK - 1u) / PTB_CHUNK;
        s.chunk_base = atomicAdd(&ctr->chunks, s.chunks);
        large_rank[atomicAdd(&ctr->large, 1u)] = r;   // k_chunk_owner spreads r over the node's chunks
    } else {
        const uint32_t k = atomicAdd(&ctr->sort_segments, 1u);
        sort_begin[k] = (uint32_t)nd.first;
        sort_end[k] = (uint32_t)(nd.first + count);
    }
}

// ---- chunked stable partition of the large nodes
// pass 0: every chunk learns which large node it belongs to (one block per large node)
__global__ void __launch_bounds__(PTB_BLOCK) k_chunk_owner(const Split* sp, const uint32_t* large_rank, uint32_t* chunk_owner) {
    const uint32_t r = large_rank[blockIdx.x];
    const Split& s = sp[r];
    for (uint32_t k = threadIdx.x; k < s.chunks; k += PTB_BLOCK) chunk_owner[s.chunk_base + k] = r;
}
// pass 1: bin histogram of every chunk
__global__ void __launch_bounds__(PTB_BLOCK) k_chunk_count(const BNode* nodes, const Split* sp, const uint32_t* chunk_owner, const uint8_t* bin_of, uint32_t* chunk_cnt) {
    __shared__ uint32_t s_cnt[PTB_BINS];
    if (threadIdx.x < PTB_BINS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const Split& s = sp[chunk_owner[blockIdx.x]];
    const BNode& nd = nodes[s.node];
    const uint32_t lo = (uint32_t)nd.first + (blockIdx.x - s.chunk_base) * PTB_CHUNK;
    const uint32_t end = (uint32_t)(nd.first + nd.count);
    const uint32_t hi = lo + PTB_CHUNK < end ? lo + PTB_CHUNK : end;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += PTB_BLOCK) atomicAdd(&s_cnt[bin_of[i]], 1u);
    __syncthreads();
    if (threadIdx.x < PTB_BINS) chunk_cnt[blockIdx.x * PTB_BINS + threadIdx.x] = s_cnt[threadIdx.x];
}

__device__ inline uint32_t waveScanInclusive(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const uint32_t row = (threadIdx.x & 63u) >> 4;
    return v + (row > 0u ? r0 : 0u) + (row > 1u ? r1 : 0u) + (row > 2u ? r2 : 0u);
}

// pass 2: one block per large node; wave w scans bins 2w and 2w+1 over the node's chunks: chunk_cnt becomes the
// destination of the chunk's first primitive of that bin
__global__ void __launch_bounds__(PTB_BLOCK) k_chunk_scan(const BNode* nodes, const Split* sp, const uint32_t* large_rank, uint32_t* chunk_cnt) {
    const Split& s = sp[large_rank[blockIdx.x]];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t b = wave * 2u; b < wave * 2u + 2u; b++) {
        uint32_t run = (uint32_t)nodes[s.node].first;
        for (uint32_t j = 0; j < b; j++) run += s.cnt[j];
        for (uint32_t k0 = 0; k0 < s.chunks; k0 += 64u) {
            const uint32_t k = k0 + lane;
            const uint32_t c = k < s.chunks ? chunk_cnt[(s.chunk_base + k) * PTB_BINS + b] : 0u;
            const uint32_t incl = waveScanInclusive(c);
            if (k < s.chunks) chunk_cnt[(s.chunk_base + k) * PTB_BINS + b] = run + incl - c;
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
}

// pass 3: scatter; the rank of a primitive among the earlier ones of its bin comes from wave ballots
__global__ void __launch_bounds__(PTB_BLOCK) k_chunk_scatter(const BNode* nodes, const Split* sp, const uint32_t* chunk_owner, const uint8_t* bin_of,
                                                             const uint32_t* chunk_cnt, const uint32_t* order, uint32_t* sorted) {
    __shared__ uint32_t s_run[PTB_BINS], s_wave[PTB_BLOCK / 64][PTB_BINS];
    const Split& s = sp[chunk_owner[blockIdx.x]];
    const BNode& nd = nodes[s.node];
    const uint32_t lo = (uint32_t)nd.first + (blockIdx.x - s.chunk_base) * PTB_CHUNK;
    const uint32_t end = (uint32_t)(nd.first + nd.count);
    const uint32_t hi = lo + PTB_CHUNK < end ? lo + PTB_CHUNK : end;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < PTB_BINS) s_run[threadIdx.x] = chunk_cnt[blockIdx.x * PTB_BINS + threadIdx.x];
    for (uint32_t base = lo; base < hi; base += PTB_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t b = i < hi ? bin_of[i] : PTB_BINS;
        uint32_t mine = 0;
        for (uint32_t bb = 0; bb < PTB_BINS; bb++) {
            const unsigned long long m = __ballot(b == bb);
            if (b == bb) mine = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_wave[wave][bb] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (b < PTB_BINS) {
            uint32_t at = s_run[b] + mine;
            for (uint32_t w = 0; w < wave; w++) at += s_wave[w][b];
            sorted[at] = order[i];
        }
        __syncthreads();
        if (threadIdx.x < PTB_BINS) {
            uint32_t add = 0;
            for (uint32_t w = 0; w < PTB_BLOCK / 64; w++) add += s_wave[w][threadIdx.x];
            s_run[threadIdx.x] += add;
        }
        __syncthreads();
    }
}

// after the partition: take the sorted primitive of every position whose node was split, and note which active
// node of the next level the position belongs to
__global__ void __launch_bounds__(PTB_BLOCK) k_reassign(const BNode* nodes, const Split* sp, uint32_t* seg, uint32_t n, uint32_t* order, const uint32_t* sorted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = seg[i];
    if (r == PTB_NONE) return;
    const Split& s = sp[r];
    uint32_t next = PTB_NONE;
    if (s.usable) {
        const BNode& nd = nodes[s.node];
        next = nodes[nd.left + (i >= s.split_pos ? 1 : 0)].next_rank;
        order[i] = sorted[i];
    }
    seg[i] = next;
}

// ---- small subtrees: Accelerator.Subdivide (Accelerator.cs:136-177) with Parition (:188-280), one wave per subtree.
// Lane k holds the primitive at position first + k of the subtree's root; the wave works on all nodes of one depth at a time.
// The primitives of a node are consecutive lanes, so everything the reference folds over a node's list is a scan over a lane
// range: the centroid range; after the lanes of a node have been put in bin order (the reference's partition, stable inside a
// bin), the boxes left and right of a split plane are a prefix and a suffix union, and the planes worth looking at are the
// boundaries between two occupied bins (a plane next to an empty bin repeats its neighbour's cost, which the reference's strict
// "<" ignores; a plane with an empty side costs 0 * inf = NaN, which it ignores too).  A node that stays a leaf keeps its
// primitives in the order they had (Accelerator.cs:262 returns before the rewrite).
// The subtree takes shape in LDS with local node numbers (the root 0, then two per split node, depth by depth).  When it is
// complete it is numbered the way the recursion would have (pair, everything below the left node, everything below the right
// one) relative to the root's pair (k_small_finish adds the root's base later), the waves of a block take their nodes from the
// shared counter with one atomic between them — a returning atomic on one address is served every dozen nanoseconds, which
// at one per wave and depth was most of this kernel's time — and every wave writes its finished nodes out.
__device__ inline uint32_t segMinU(uint32_t v, uint32_t lane, uint32_t lo, uint32_t hi) {
    for (uint32_t off = 1; off < 64u; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)v, off); if (lane >= lo + off) v = t < v ? t : v; }
    return (uint32_t)__shfl((int)v, (int)hi - 1);
}
__device__ inline uint32_t segMaxU(uint32_t v, uint32_t lane, uint32_t lo, uint32_t hi) {
    for (uint32_t off = 1; off < 64u; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)v, off); if (lane >= lo + off) v = t > v ? t : v; }
    return (uint32_t)__shfl((int)v, (int)hi - 1);
}
#define PTB_SW_WAVES 8     // subtrees (waves) per block of k_small_subtrees
__global__ void __launch_bounds__(PTB_SW_WAVES * 64) k_small_subtrees(BNode* nodes, const uint32_t* small, uint32_t n_small, const float4* cent,
                                                                      const float4* pmin, const float4* pmax, uint32_t* order, Counters* ctr) {
    static_assert(PTB_SMALL <= 64, "one lane per primitive of a small subtree");
    constexpr int WAVES = PTB_SW_WAVES, LOCAL = 128;   // at most 2 * 64 - 1 nodes
    __shared__ float s_box[WAVES][LOCAL][6];
    __shared__ uint8_t s_first[WAVES][LOCAL], s_count[WAVES][LOCAL], s_axis[WAVES][LOCAL], s_depth[WAVES][LOCAL];
    __shared__ uint8_t s_left[WAVES][LOCAL];           // local number of the left child (right = left + 1), 0xFF for a leaf
    __shared__ uint8_t s_below[WAVES][LOCAL], s_block[WAVES][LOCAL], s_index[WAVES][LOCAL];
    __shared__ uint8_t s_levelStart[WAVES][68];        // local numbers are handed out depth by depth
    __shared__ uint8_t s_src[WAVES][64];
    __shared__ uint32_t s_made[WAVES], s_base;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * WAVES + wave;
    const bool live = t < n_small;                     // wave-uniform
    auto waveSync = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    const float inf = __builtin_huge_valf();
    uint32_t root = 0, total = 0, rootLevel = 0, localCount = 1, depth = 0, id = 0;
    int32_t f0 = 0;
    bool holds = false;
    if (live) {
        root = small[t];
        f0 = nodes[root].first;
        total = (uint32_t)nodes[root].count;
        rootLevel = nodes[root].level;
        // the lane's primitive
        holds = lane < total;
        id = holds ? order[f0 + lane] : 0u;
        float c[3] = {0, 0, 0}, lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        if (holds) {
            const float4 cc = cent[id], l4 = pmin[id], h4 = pmax[id];
            c[0] = cc.x; c[1] = cc.y; c[2] = cc.z;
            lo[0] = l4.x; lo[1] = l4.y; lo[2] = l4.z;
            hi[0] = h4.x; hi[1] = h4.y; hi[2] = h4.z;
        }
        // the node the lane's primitive is in
        bool active = holds;                              // false once its node is a leaf
        uint32_t nlo = holds ? 0u : lane, nhi = holds ? total : lane + 1u;
        float nmn[3], nmx[3];
        for (int a = 0; a < 3; a++) { nmn[a] = nodes[root].mn[a]; nmx[a] = nodes[root].mx[a]; }
        uint32_t myLocal = 0;
        if (lane == 0) { s_left[wave][0] = 0xFFu; s_levelStart[wave][0] = 0; s_levelStart[wave][1] = 1; }

        while (__any(active)) {
            const int axis = domainOf(nmn, nmx);
            const float ca = axis == 0 ? c[0] : (axis == 1 ? c[1] : c[2]);
            // CentroidBounds along the axis (Accelerator.cs:294-310), k0 and k1 (:196-213)
            const uint32_t kc = fkey(ca);
            const float k0 = fval(segMinU(kc, lane, nlo, nhi));
            const float l = fval(segMaxU(kc, lane, nlo, nhi)) - k0;
            const bool usable = active && !(l < PTB_EPSILON);
            const float k1 = ((float)PTB_BINS - PTB_EPSILON) / l;
            int j = usable ? (int)((ca - k0) * k1) : 0;
            j = j < 0 ? 0 : (j > PTB_BINS - 1 ? PTB_BINS - 1 : j);
            // position of the primitive once its node is in bin order
            const unsigned long long nodeMask = (nhi >= 64u ? ~0ull : (1ull << nhi) - 1ull) & ~((1ull << nlo) - 1ull);
            const unsigned long long before = nodeMask & ((1ull << lane) - 1ull);
            uint32_t pos = 0;
            for (int b = 0; b < PTB_BINS; b++) {
                const unsigned long long m = __ballot(usable && j == b);
                if (b < j) pos += (uint32_t)__popcll(m & nodeMask);
                else if (b == j) pos += (uint32_t)__popcll(m & before);
            }
            const uint32_t newLane = usable ? nlo + pos : lane;
            s_src[wave][newLane] = (uint8_t)lane;
            waveSync();
            const int src = (int)s_src[wave][lane];
            waveSync();
            const uint32_t pid = (uint32_t)__shfl((int)id, src);
            const int pj = __shfl(j, src);
            float pc[3], plo[3], phi[3];
            for (int a = 0; a < 3; a++) { pc[a] = __shfl(c[a], src); plo[a] = __shfl(lo[a], src); phi[a] = __shfl(hi[a], src); }
            // union of the node's primitives up to this lane, and from this lane on
            float pmn[3], pmx[3], smn[3], smx[3];
            for (int a = 0; a < 3; a++) { pmn[a] = smn[a] = plo[a]; pmx[a] = smx[a] = phi[a]; }
            for (uint32_t off = 1; off < 64u; off <<= 1) {
                const bool up = lane >= nlo + off, down = lane + off < nhi;
                for (int a = 0; a < 3; a++) {
                    const float u0 = __shfl_up(pmn[a], off), u1 = __shfl_up(pmx[a], off), d0 = __shfl_down(smn[a], off), d1 = __shfl_down(smx[a], off);
                    if (up) { pmn[a] = u0 < pmn[a] ? u0 : pmn[a]; pmx[a] = u1 > pmx[a] ? u1 : pmx[a]; }
                    if (down) { smn[a] = d0 < smn[a] ? d0 : smn[a]; smx[a] = d1 > smx[a] ? d1 : smx[a]; }
                }
            }
            // the sweep (Accelerator.cs:229-262): a split after this lane, if the next lane starts another bin
            const int nextJ = __shfl_down(pj, 1);
            float rmn[3], rmx[3];
            for (int a = 0; a < 3; a++) { rmn[a] = __shfl_down(smn[a], 1); rmx[a] = __shfl_down(smx[a], 1); }
            const uint32_t count = nhi - nlo, ln = lane + 1u - nlo, rn = count - ln;
            const float cost0 = areaOf(nmn, nmx) * (float)count;
            const float cost = (float)ln * areaOf(pmn, pmx) + (float)rn * areaOf(rmn, rmx);
            const bool better = usable && lane + 1u < nhi && pj != nextJ && cost < cost0;
            // smallest cost, then the first plane that reaches it
            uint32_t bc = better ? __float_as_uint(cost) : 0xFFFFFFFFu, bl = better ? lane : 0xFFFFFFFFu;
            for (uint32_t off = 1; off < 64u; off <<= 1) {
                const uint32_t tc = (uint32_t)__shfl_up((int)bc, off), tl = (uint32_t)__shfl_up((int)bl, off);
                if (lane >= nlo + off && (tc < bc || (tc == bc && tl < bl))) { bc = tc; bl = tl; }
            }
            const uint32_t splitLane = (uint32_t)__shfl((int)bl, (int)nhi - 1);
            const bool split = usable && splitLane != 0xFFFFFFFFu;
            // the children's bounds (Accelerator.cs:264-279): what lies left and right of the chosen plane
            const int sl = split ? (int)splitLane : (int)lane, sr = split ? (int)splitLane + 1 : (int)lane;
            float Lmn[3], Lmx[3], Rmn[3], Rmx[3];
            for (int a = 0; a < 3; a++) { Lmn[a] = __shfl(pmn[a], sl); Lmx[a] = __shfl(pmx[a], sl); Rmn[a] = __shfl(smn[a], sr); Rmx[a] = __shfl(smx[a], sr); }
            // two new nodes per split node
            const bool starts = split && lane == nlo;
            const unsigned long long sm = __ballot(starts);
            const uint32_t nSplit = (uint32_t)__popcll(sm);
            if (nSplit == 0u) break;
            const uint32_t childLocal = localCount + 2u * (uint32_t)__popcll(sm & ((1ull << nlo) - 1ull));
            if (starts) { s_left[wave][myLocal] = (uint8_t)childLocal; s_axis[wave][myLocal] = (uint8_t)axis; }
            depth++;
            active = false;
            if (split) {
                id = pid;
                for (int a = 0; a < 3; a++) { c[a] = pc[a]; lo[a] = plo[a]; hi[a] = phi[a]; }
                const bool right = lane > splitLane;
                if (right) nlo = splitLane + 1u; else nhi = splitLane + 1u;
                for (int a = 0; a < 3; a++) { nmn[a] = right ? Rmn[a] : Lmn[a]; nmx[a] = right ? Rmx[a] : Lmx[a]; }
                myLocal = childLocal + (right ? 1u : 0u);
                if (lane == nlo) {
                    for (int a = 0; a < 3; a++) { s_box[wave][myLocal][a] = nmn[a]; s_box[wave][myLocal][3 + a] = nmx[a]; }
                    s_first[wave][myLocal] = (uint8_t)nlo;
                    s_count[wave][myLocal] = (uint8_t)(nhi - nlo);
                    s_depth[wave][myLocal] = (uint8_t)depth;
                    s_left[wave][myLocal] = 0xFFu;
                }
                active = nhi - nlo >= PTB_LEAF;           // Accelerator.cs:140-143
            }
            localCount += 2u * nSplit;
            if (lane == 0) s_levelStart[wave][depth + 1u] = (uint8_t)localCount;
        }
        waveSync();
        // nodes below every node, deepest nodes first; then the reference's numbers, root first
        for (uint32_t d = depth + 1u; d-- > 0u;) {
            const uint32_t n = s_levelStart[wave][d] + lane;
            if (n < s_levelStart[wave][d + 1u]) {
                const uint32_t left = s_left[wave][n];
                s_below[wave][n] = left == 0xFFu ? (uint8_t)0 : (uint8_t)(2u + s_below[wave][left] + s_below[wave][left + 1u]);
            }
            waveSync();
        }
        if (lane == 0) s_block[wave][0] = 0u;
        waveSync();
        for (uint32_t d = 0; d <= depth; d++) {
            const uint32_t n = s_levelStart[wave][d] + lane;
            if (n < s_levelStart[wave][d + 1u]) {
                const uint32_t left = s_left[wave][n];
                if (left != 0xFFu) {
                    const uint32_t blk = s_block[wave][n];
                    s_index[wave][left] = (uint8_t)blk;
                    s_index[wave][left + 1u] = (uint8_t)(blk + 1u);
                    s_block[wave][left] = (uint8_t)(blk + 2u);
                    s_block[wave][left + 1u] = (uint8_t)(blk + 2u + s_below[wave][left]);
                }
            }
            waveSync();
        }
        if (holds) order[f0 + lane] = id;
    }
    // this block's nodes: one atomic for all its subtrees
    if (lane == 0) s_made[wave] = live ? localCount - 1u : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (int w = 0; w < WAVES; w++) { const uint32_t m = s_made[w]; s_made[w] = sum; sum += m; }
        s_base = sum ? atomicAdd(&ctr->nodes, sum) : 0u;
    }
    __syncthreads();
    if (!live) return;
    const uint32_t base = s_base + s_made[wave] - 1u;     // local node k >= 1 becomes node base + k
    for (uint32_t k = 1u + lane; k < localCount; k += 64u) {
        BNode nd{};
        for (int a = 0; a < 3; a++) { nd.mn[a] = s_box[wave][k][a]; nd.mx[a] = s_box[wave][k][3 + a]; }
        nd.first = f0 + (int32_t)s_first[wave][k];
        nd.count = (int32_t)s_count[wave][k];
        const uint32_t left = s_left[wave][k];
        nd.left = left == 0xFFu ? -1 : (int32_t)(base + left);
        nd.axis = left == 0xFFu ? 0 : (int32_t)s_axis[wave][k];
        nd.next_rank = PTB_NONE;
        nd.block = left == 0xFFu ? 0u : (uint32_t)s_block[wave][k];
        nd.index = s_index[wave][k];
        nd.level = rootLevel + s_depth[wave][k];
        nd.aux = root;
        nodes[base + k] = nd;
    }
    if (lane == 0) {
        const uint32_t left = s_left[wave][0];
        if (left != 0xFFu) { nodes[root].left = (int32_t)(base + left); nodes[root].axis = (int32_t)s_axis[wave][0]; }
        nodes[root].below = s_below[wave][0];
        atomicMax(&ctr->max_level, rootLevel + depth);
    }
}
// relative numbers of the thread-built nodes -> final ones, once the levels above have numbered the subtree roots
__global__ void __launch_bounds__(PTB_BLOCK) k_small_finish(BNode* nodes, uint32_t lo, uint32_t hi) {
    const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    BNode& nd = nodes[i];
    const uint32_t base = nodes[nd.aux].block;
    nd.index += base;
    if (nd.left >= 0) nd.block += base;
}

// ---- renumbering into the reference's order
__global__ void __launch_bounds__(PTB_BLOCK) k_below(BNode* nodes, uint32_t lo, uint32_t hi) {
    const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    BNode& nd = nodes[i];
    if (nd.aux == i) return;   // root of a thread-built subtree: the thread counted what lies below it
    nd.below = nd.left < 0 ? 0u : 2u + nodes[nd.left].below + nodes[nd.left + 1].below;
}
__global__ void __launch_bounds__(PTB_BLOCK) k_number(BNode* nodes, uint32_t lo, uint32_t hi) {
    const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    BNode& nd = nodes[i];
    if (i == 0) { nd.index = 0; nd.block = 1; }
    if (nd.left < 0 || nd.aux == i) return;   // below the root of a thread-built subtree k_small_finish numbers
    BNode& L = nodes[nd.left];
    BNode& R = nodes[nd.left + 1];
    L.index = nd.block;
    R.index = nd.block + 1;
    L.block = nd.block + 2;
    R.block = nd.block + 2 + L.below;
}
__global__ void __launch_bounds__(PTB_BLOCK) k_emit(const BNode* nodes, uint32_t n_nodes, Node32* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const BNode& nd = nodes[i];
    Node32 o;
    for (int a = 0; a < 3; a++) { o.mn[a] = nd.mn[a]; o.mx[a] = nd.mx[a]; }
    if (nd.left < 0) { o.leftFirst = nd.first; o.count = nd.count; }
    else { o.leftFirst = (int32_t)nd.block; o.count = nd.axis * -1 - 1; }   // split axis through a negative count, Accelerator.cs:158-159
    out[nd.index] = o;
}
__global__ void __launch_bounds__(PTB_BLOCK) k_gather(const Tri112* in, const uint32_t* order, uint32_t n, Tri112* out) {
    // Accelerator.cs:129-133; 7 lanes of 16 bytes per triangle
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = t / 7u, part = t % 7u;
    if (i >= n) return;
    ((float4*)(out + i))[part] = ((const float4*)(in + order[i]))[part];
}

inline int grid(uint32_t n) { return (int)((n + PTB_BLOCK - 1) / PTB_BLOCK); }

// Builds on `stream`.  d_tris: n triangles in insertion order.  On return d_nodes_out (capacity max(1, 2n-1)) and
// d_tris_out (n) hold the reference's arrays.  Returns hipSuccess or the failing call's error.
// keep_nodes (optional): receives the construction-order node array instead of it being freed (hipFree it).
// keep_order (optional): receives the permutation (position in d_tris_out -> position in d_tris), n words (hipFree it).
inline hipError_t build(hipStream_t stream, const Tri112* d_tris, uint32_t n, Node32* d_nodes_out, Tri112* d_tris_out, uint32_t* n_nodes_out,
                        uint32_t* levels_out, BNode** keep_nodes = nullptr, uint32_t** keep_order = nullptr) {
#define PTB_TRY(x) do { err = (x); if (err != hipSuccess) goto done; } while (0)
    hipError_t err = hipSuccess;
    const uint32_t maxNodes = n ? 2u * n - 1u : 1u;
    const uint32_t maxActive = n / PTB_LEAF + 2u;
    float4 *pmin = nullptr, *pmax = nullptr, *cent = nullptr;
    uint32_t *order[2] = {nullptr, nullptr}, *seg = nullptr, *active[2] = {nullptr, nullptr}, *sortb = nullptr, *sorte = nullptr, *rootk = nullptr;
    uint32_t *largeRank = nullptr, *chunkOwner = nullptr, *chunkCnt = nullptr, *small = nullptr;
    const uint32_t maxChunks = 2u * (n / PTB_CHUNK) + 2u, maxLarge = n / PTB_CHUNK + 1u;
    uint8_t* bins[2] = {nullptr, nullptr};
    BNode* nodes = nullptr;
    Split* sp = nullptr;
    Counters* ctr = nullptr;
    void* tmp = nullptr;
    size_t tmpBytes = 0;
    std::vector<uint32_t> levelStart;
    Counters h{};
    const uint32_t nn = n ? n : 1u;
    const int cur = 0;
    int act = 0;
    uint32_t nActive = 0, levelNodes = 0, smallDepth = 0;
    // the per-primitive passes of a level: about 4096 blocks, each over `tiles` consecutive tiles of 256 positions
    const uint32_t nTiles = (nn + PTB_BLOCK - 1u) / PTB_BLOCK, tiles = nTiles > 4096u ? (nTiles + 4095u) / 4096u : 1u, tileGrid = (nTiles + tiles - 1u) / tiles;
    const uint32_t initKeys[6] = {PTB_KEY_POS_INF, PTB_KEY_POS_INF, PTB_KEY_POS_INF, PTB_KEY_NEG_INF, PTB_KEY_NEG_INF, PTB_KEY_NEG_INF};

    PTB_TRY(ptMalloc((void**)&pmin, (size_t)nn * 16));
    PTB_TRY(ptMalloc((void**)&pmax, (size_t)nn * 16));
    PTB_TRY(ptMalloc((void**)&cent, (size_t)nn * 16));
    for (int k = 0; k < 2; k++) {
        PTB_TRY(ptMalloc((void**)&order[k], (size_t)nn * 4));
        PTB_TRY(ptMalloc((void**)&bins[k], (size_t)nn));
        PTB_TRY(ptMalloc((void**)&active[k], (size_t)maxActive * 4));
    }
    PTB_TRY(ptMalloc((void**)&seg, ((size_t)nn + PTB_BLOCK) * 4));
    PTB_TRY(ptMalloc((void**)&sortb, (size_t)maxActive * 4));
    PTB_TRY(ptMalloc((void**)&sorte, (size_t)maxActive * 4));
    PTB_TRY(ptMalloc((void**)&rootk, 6 * 4));
    PTB_TRY(ptMalloc((void**)&small, (size_t)maxActive * 4));
    PTB_TRY(ptMalloc((void**)&largeRank, (size_t)maxLarge * 4));
    PTB_TRY(ptMalloc((void**)&chunkOwner, (size_t)maxChunks * 4));
    PTB_TRY(ptMalloc((void**)&chunkCnt, (size_t)maxChunks * PTB_BINS * 4));
    PTB_TRY(ptMalloc((void**)&nodes, (size_t)maxNodes * sizeof(BNode)));
    PTB_TRY(ptMalloc((void**)&sp, (size_t)maxActive * sizeof(Split)));
    PTB_TRY(ptMalloc((void**)&ctr, sizeof(Counters)));
    PTB_TRY(hipMemcpyAsync(rootk, initKeys, sizeof(initKeys), hipMemcpyHostToDevice, stream));
    PTB_TRY(hipMemsetAsync(seg, 0xFF, ((size_t)nn + PTB_BLOCK) * 4, stream));
    PTB_TRY(hipMemsetAsync(bins[0], 0, nn, stream));
    if (n) hipLaunchKernelGGL(k_prims, dim3(grid(n)), dim3(PTB_BLOCK), 0, stream, d_tris, n, pmin, pmax, cent, order[0], seg, rootk);
    hipLaunchKernelGGL(k_root, dim3(1), dim3(1), 0, stream, nodes, rootk, n, ctr, active[0], small);
    PTB_TRY(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
    PTB_TRY(hipStreamSynchronize(stream));
    levelStart.push_back(0);
    levelStart.push_back(h.nodes);
    nActive = h.next_active;
    if (n) {
        // temporary storage of the segmented sort, sized for the worst case once
        PTB_TRY(rocprim::segmented_radix_sort_pairs(nullptr, tmpBytes, bins[0], bins[1], order[0], order[1], n, maxActive, sortb, sorte, 0, 3, stream));
        PTB_TRY(ptMalloc(&tmp, tmpBytes ? tmpBytes : 16));
    }
    while (nActive > 0) {
        Counters z{};
        z.nodes = h.nodes;
        z.small = h.small;
        PTB_TRY(hipMemcpyAsync(ctr, &z, sizeof(z), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_split_begin, dim3(grid(nActive)), dim3(PTB_BLOCK), 0, stream, nodes, active[act], nActive, sp);
        hipLaunchKernelGGL(k_centroid_range, dim3(tileGrid), dim3(PTB_BLOCK), 0, stream, cent, order[cur], seg, n, sp, tiles);
        hipLaunchKernelGGL(k_split_bins, dim3(grid(nActive)), dim3(PTB_BLOCK), 0, stream, nActive, sp);
        hipLaunchKernelGGL(k_bin, dim3(tileGrid), dim3(PTB_BLOCK), 0, stream, cent, pmin, pmax, order[cur], seg, n, sp, bins[cur], tiles);
        hipLaunchKernelGGL(k_split_decide, dim3(grid(nActive)), dim3(PTB_BLOCK), 0, stream, nodes, nActive, sp, ctr, active[act ^ 1], sortb, sorte, largeRank, small);
        PTB_TRY(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
        PTB_TRY(hipStreamSynchronize(stream));
        if (h.sort_segments > 0)   // sorted copies of the split nodes' ranges land in order[1]; k_reassign takes them over
            PTB_TRY(rocprim::segmented_radix_sort_pairs(tmp, tmpBytes, bins[0], bins[1], order[0], order[1], n, h.sort_segments, sortb, sorte, 0, 3, stream));
        if (h.chunks > 0) {
            hipLaunchKernelGGL(k_chunk_owner, dim3(h.large), dim3(PTB_BLOCK), 0, stream, sp, largeRank, chunkOwner);
            hipLaunchKernelGGL(k_chunk_count, dim3(h.chunks), dim3(PTB_BLOCK), 0, stream, nodes, sp, chunkOwner, bins[0], chunkCnt);
            hipLaunchKernelGGL(k_chunk_scan, dim3(h.large), dim3(PTB_BLOCK), 0, stream, nodes, sp, largeRank, chunkCnt);
            hipLaunchKernelGGL(k_chunk_scatter, dim3(h.chunks), dim3(PTB_BLOCK), 0, stream, nodes, sp, chunkOwner, bins[0], chunkCnt, order[0], order[1]);
        }
        hipLaunchKernelGGL(k_reassign, dim3(grid(n)), dim3(PTB_BLOCK), 0, stream, nodes, sp, seg, n, order[0], order[1]);
        levelStart.push_back(h.nodes);
        nActive = h.next_active;
        act ^= 1;
    }
    // everything below the small nodes, one thread per subtree; their nodes follow the levels' in creation order
    levelNodes = h.nodes;
    if (h.small > 0) {
        hipLaunchKernelGGL(k_small_subtrees, dim3((h.small + PTB_SW_WAVES - 1u) / PTB_SW_WAVES), dim3(PTB_SW_WAVES * 64), 0, stream, nodes, small, h.small, cent, pmin, pmax, order[0], ctr);
        PTB_TRY(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
        PTB_TRY(hipStreamSynchronize(stream));
        smallDepth = h.max_level + 1u;
    }
    // renumber: nodes below (deepest level first), then final indices (root first)
    for (size_t l = levelStart.size() - 1; l-- > 0;) {
        const uint32_t lo = levelStart[l], hi = levelStart[l + 1];
        if (hi > lo) hipLaunchKernelGGL(k_below, dim3(grid(hi - lo)), dim3(PTB_BLOCK), 0, stream, nodes, lo, hi);
    }
    for (size_t l = 0; l + 1 < levelStart.size(); l++) {
        const uint32_t lo = levelStart[l], hi = levelStart[l + 1];
        if (hi > lo) hipLaunchKernelGGL(k_number, dim3(grid(hi - lo)), dim3(PTB_BLOCK), 0, stream, nodes, lo, hi);
    }
    if (h.nodes > levelNodes) hipLaunchKernelGGL(k_small_finish, dim3(grid(h.nodes - levelNodes)), dim3(PTB_BLOCK), 0, stream, nodes, levelNodes, h.nodes);
    hipLaunchKernelGGL(k_emit, dim3(grid(h.nodes)), dim3(PTB_BLOCK), 0, stream, nodes, h.nodes, d_nodes_out);
    if (n) hipLaunchKernelGGL(k_gather, dim3(grid(n * 7u)), dim3(PTB_BLOCK), 0, stream, d_tris, order[cur], n, d_tris_out);
    PTB_TRY(hipGetLastError());
    PTB_TRY(hipStreamSynchronize(stream));
    *n_nodes_out = h.nodes;
    if (levels_out) {   // levels that hold nodes: the last round of the loop adds an empty one when none of its nodes split
        uint32_t levels = 0;
        for (size_t l = 0; l + 1 < levelStart.size(); l++) levels += levelStart[l + 1] > levelStart[l] ? 1u : 0u;
        *levels_out = levels > smallDepth ? levels : smallDepth;
    }
    if (keep_nodes) { *keep_nodes = nodes; nodes = nullptr; }
    if (keep_order) { *keep_order = order[cur]; order[cur] = nullptr; }
done:
    (void)hipFree(pmin); (void)hipFree(pmax); (void)hipFree(cent);
    for (int k = 0; k < 2; k++) { (void)hipFree(order[k]); (void)hipFree(bins[k]); (void)hipFree(active[k]); }
    (void)hipFree(largeRank); (void)hipFree(chunkOwner); (void)hipFree(chunkCnt); (void)hipFree(small);
    (void)hipFree(seg); (void)hipFree(sortb); (void)hipFree(sorte); (void)hipFree(rootk); (void)hipFree(nodes); (void)hipFree(sp); (void)hipFree(ctr); (void)hipFree(tmp);
    return err;
#undef PTB_TRY
}

}  // namespace ptbuild
