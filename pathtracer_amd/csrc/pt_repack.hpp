// The device scene (quad records, pair records, hot/cold triangles) derived on the GPU from a tree that
// ptbuild::build has just constructed, so a scene built on the device never visits the host.
// Records are the ones pt_upload_scene forms on the CPU from the host's arrays (see pt_device.hpp: DevScene);
// node numbers here are construction order (a node's children are adjacent at an odd index, like the host's),
// which the traversal never observes.
#pragma once
#include "pt_build.hpp"
#include "pt_device.hpp"
#include <rocprim/device/device_scan.hpp>

namespace ptrepack {

using ptbuild::BNode;
using ptbuild::Tri112;

__device__ inline bool fitsPacked(int32_t first, int32_t count) { return count <= PT_LEAF_MAX_COUNT && (uint32_t)first < PT_LEAF_MAX_FIRST; }

// hot / cold triangle records (pt_upload_scene does the same two subtractions, Triangle.cs:17-18), emissive flags
__global__ void __launch_bounds__(256) k_triangles(const Tri112* tris, uint32_t n, const float4* mats, int32_t n_mat, float4* hot, float4* cold,
                                                   uint32_t* emissive, uint32_t* bad_material) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Tri112& t = tris[i];
    const float* a = t.v[0]; const float* b = t.v[1]; const float* c = t.v[2];
    hot[3 * (size_t)i] = make_float4(a[0], a[1], a[2], __int_as_float(t.material));
    hot[3 * (size_t)i + 1] = make_float4(b[0] - a[0], b[1] - a[1], b[2] - a[2], 0.0f);
    hot[3 * (size_t)i + 2] = make_float4(c[0] - a[0], c[1] - a[1], c[2] - a[2], 0.0f);
    for (int k = 0; k < 3; k++) cold[3 * (size_t)i + k] = make_float4(t.v[k][4], t.v[k][5], t.v[k][6], 0.0f);
    uint32_t e = 0;
    if (t.material < 0 || t.material >= n_mat) atomicOr(bad_material, 1u);
    else {
        const float4 em = mats[3 * (size_t)t.material + 1];   // Material.Emissive, Material.cs:7-28
        e = (em.x > 0.0f || em.y > 0.0f || em.z > 0.0f) ? 1u : 0u;   // shade.glsl:366
    }
    emissive[i] = e;
}
// Light table rows of the emissive triangles, in triangle order: (V1, running area sum) (V2, material bits) (V3, 0).  The area is
// computed as the host path computes it (same expression, same rounding) and goes to areas[]; k_light_sums fills in the sums.
__global__ void __launch_bounds__(256) k_light_rows(const Tri112* tris, uint32_t n, const uint32_t* emissive, const uint32_t* slot, float4* lights, float* areas) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !emissive[i]) return;
    const Tri112& t = tris[i];
    const float* a = t.v[0]; const float* b = t.v[1]; const float* c = t.v[2];
    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    const float e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
    const uint32_t s = slot[i];
    areas[s] = 0.5f * sqrtf((nx * nx + ny * ny) + nz * nz);
    float4* row = lights + 3 * (size_t)s;
    row[0] = make_float4(a[0], a[1], a[2], 0.0f);
    row[1] = make_float4(b[0], b[1], b[2], __int_as_float(t.material));
    row[2] = make_float4(c[0], c[1], c[2], 0.0f);
}
// The running area sum is a chain of float additions in triangle order (its rounding is part of the light table the host path
// builds), so it cannot be split over threads: one wave walks it, 64 areas per round.  The sum travels along the lanes: every
// step each lane adds its own area to the value of the lane below it (lane 0: to the sum so far), so after step k lanes 0..k
// hold their final sums and repeating the step leaves them as they are — one shifted addition per link of the chain.
__global__ void __launch_bounds__(64) k_light_sums(const float* __restrict__ areas, float4* __restrict__ lights, uint32_t n_lights, float* __restrict__ total) {
    const uint32_t lane = threadIdx.x;
    float carry = 0.0f;
    float next = lane < n_lights ? areas[lane] : 0.0f;
    for (uint32_t base = 0; base < n_lights; base += 64u) {
        const float v = lane == 0u ? carry + next : next;   // lane 0 starts from the sum of the rounds before
        next = base + 64u + lane < n_lights ? areas[base + 64u + lane] : 0.0f;
        float r = v;
#pragma unroll
        for (int step = 0; step < 63; step++) {
            // the value of lane - 1 (wave_shr:1); lane 0 has no such lane and reads +0, which leaves its (non-negative) sum as it is
            const float below = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x138, 0xf, 0xf, true));
            r = below + v;
        }
        if (base + lane < n_lights) lights[3 * (size_t)(base + lane)].w = r;
        const uint32_t last = n_lights - base < 64u ? n_lights - base - 1u : 63u;
        carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), (int)last));
    }
    if (lane == 0) *total = carry;
}

// which nodes own a quad record (interior, even depth) and which leaves need a big-leaf row
__global__ void __launch_bounds__(256) k_flags(const BNode* nodes, uint32_t n_nodes, uint32_t* quad_flag, uint32_t* big_flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const BNode& nd = nodes[i];
    quad_flag[i] = (nd.left >= 0 && (nd.level & 1u) == 0u) ? 1u : 0u;
    big_flag[i] = (nd.left < 0 && nd.count > 0 && !fitsPacked(nd.first, nd.count)) ? 1u : 0u;
}
__device__ inline uint32_t leafEntry(const BNode& nd, uint32_t big_row) {
    if (fitsPacked(nd.first, nd.count)) return PT_LEAF_FLAG | ((uint32_t)(nd.count - 1) << 27) | (uint32_t)nd.first;
    return PT_LEAF_FLAG | (15u << 27) | big_row;
}
__global__ void __launch_bounds__(256) k_records(const BNode* nodes, uint32_t n_nodes, const uint32_t* quad_idx, const uint32_t* big_idx, const uint32_t* big_flag,
                                                 float4* pairs, float4* quads, uint2* big) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const BNode& nd = nodes[i];
    if (big_flag[i]) big[big_idx[i]] = make_uint2((uint32_t)nd.first, (uint32_t)nd.count);
    if (i > 0) {
        // half of pair record (i-1)/2: the node's box, its entry, its count word
        const uint32_t e = nd.left >= 0 ? (uint32_t)nd.left : leafEntry(nd, big_idx[i]);
        pairs[2 * (size_t)(i - 1)] = make_float4(nd.mn[0], nd.mn[1], nd.mn[2], nd.mx[0]);
        pairs[2 * (size_t)(i - 1) + 1] = make_float4(nd.mx[1], nd.mx[2], __uint_as_float(e), __int_as_float(nd.left >= 0 ? -nd.axis - 1 : nd.count));
    }
    if (nd.left >= 0 && (nd.level & 1u) == 0u) {
        int32_t kids[4];
        int nk = 0;
        for (int32_t ch = nd.left; ch <= nd.left + 1; ch++) {
            if (nodes[ch].left < 0) kids[nk++] = ch;
            else { kids[nk++] = nodes[ch].left; kids[nk++] = nodes[ch].left + 1; }
        }
        float box[6][4] = {};
        uint32_t ent[4] = {0, 0, 0, 0};
        for (int k = 0; k < nk; k++) {
            const BNode& kd = nodes[kids[k]];
            for (int a = 0; a < 3; a++) { box[a][k] = kd.mn[a]; box[3 + a][k] = kd.mx[a]; }
            ent[k] = kd.left >= 0 ? quad_idx[kids[k]] + 1u : leafEntry(kd, big_idx[kids[k]]);
        }
        float4* q = quads + 8 * (size_t)quad_idx[i];
        for (int a = 0; a < 6; a++) q[a] = make_float4(box[a][0], box[a][1], box[a][2], box[a][3]);
        q[6] = make_float4(__uint_as_float(ent[0]), __uint_as_float(ent[1]), __uint_as_float(ent[2]), __uint_as_float(ent[3]));
        q[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// ---- quantised quad records and gated triangles (see pt_device.hpp: DevScene.qquads / tri_gate, rayOffGrid)
// The grid: per axis 65000 cells over the root box (an axis much thinner than the widest one gets cells of the widest
// axis / 64, so that no cell is degenerate), four spare cells below.  grid_reach = -1 switches the quantised walk off for
// every ray when the scene is too small or too large for the error bound's assumptions (no overflow or underflow in
// cell / d and (grid_min - o) / d).
struct QGrid { float mn[3], cell[3], mid[3], reach[3]; };
__host__ __device__ inline void gridOf(const float* rmn, const float* rmx, QGrid* g) {
    float ext[3], widest = 0.0f;
    for (int a = 0; a < 3; a++) { ext[a] = rmx[a] - rmn[a]; widest = ext[a] > widest ? ext[a] : widest; }
    const bool usable = widest >= 1.0e-10f && widest <= 1.0e15f;
    for (int a = 0; a < 3; a++) {
        float eff = ext[a] > widest * 0.015625f ? ext[a] : widest * 0.015625f;
        if (!(eff >= 1.0e-20f)) eff = 1.0e-20f;
        g->cell[a] = eff / 65000.0f;
        g->mn[a] = rmn[a] - 4.0f * g->cell[a];
        g->mid[a] = 0.5f * rmn[a] + 0.5f * rmx[a];
        g->reach[a] = usable ? 16.0f * eff : -1.0f;
    }
}
// One 64-byte record per quad record: cell numbers in double arithmetic, one whole cell outward of the cell that holds the plane.
__global__ void __launch_bounds__(256) k_qrecords(const BNode* nodes, uint32_t n_nodes, const uint32_t* quad_idx, const uint32_t* big_idx, uint4* qquads, uint32_t* bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const BNode& nd = nodes[i];
    if (!(nd.left >= 0 && (nd.level & 1u) == 0u)) return;
    QGrid g;
    gridOf(nodes[0].mn, nodes[0].mx, &g);
    int32_t kids[4];
    int nk = 0;
    for (int32_t ch = nd.left; ch <= nd.left + 1; ch++) {
        if (nodes[ch].left < 0) kids[nk++] = ch;
        else { kids[nk++] = nodes[ch].left; kids[nk++] = nodes[ch].left + 1; }
    }
    uint32_t lo[3][4] = {}, hi[3][4] = {}, ent[4] = {0, 0, 0, 0};
    for (int k = 0; k < nk; k++) {
        const BNode& kd = nodes[kids[k]];
        for (int a = 0; a < 3; a++) {
            const double vl = ((double)kd.mn[a] - (double)g.mn[a]) / (double)g.cell[a];
            const double vh = ((double)kd.mx[a] - (double)g.mn[a]) / (double)g.cell[a];
            double ql = floor(vl) - 1.0, qh = ceil(vh) + 1.0;
            if (!(ql >= 0.0) || !(qh <= 65535.0)) { atomicOr(bad, 1u); ql = 0.0; qh = 65535.0; }   // also catches NaN
            lo[a][k] = (uint32_t)ql;
            hi[a][k] = (uint32_t)qh;
        }
        ent[k] = kd.left >= 0 ? quad_idx[kids[k]] + 1u : leafEntry(kd, big_idx[kids[k]]);
    }
#define PT_PK(v, a, b) ((v)[a] | ((v)[b] << 16))
    uint4* q = qquads + 4 * (size_t)quad_idx[i];
    q[0] = make_uint4(PT_PK(lo[0], 0, 1), PT_PK(lo[0], 2, 3), PT_PK(lo[1], 0, 1), PT_PK(lo[1], 2, 3));
    q[1] = make_uint4(PT_PK(lo[2], 0, 1), PT_PK(lo[2], 2, 3), PT_PK(hi[0], 0, 1), PT_PK(hi[0], 2, 3));
    q[2] = make_uint4(PT_PK(hi[1], 0, 1), PT_PK(hi[1], 2, 3), PT_PK(hi[2], 0, 1), PT_PK(hi[2], 2, 3));
    q[3] = make_uint4(ent[0], ent[1], ent[2], ent[3]);
#undef PT_PK
}
// The triangles of every leaf with that leaf's box beside them.
__global__ void __launch_bounds__(256) k_tri_gate(const BNode* nodes, uint32_t n_nodes, const float4* hot, float4* gate) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const BNode& nd = nodes[i];
    if (nd.left >= 0) return;
    for (int32_t j = nd.first; j < nd.first + nd.count; j++) {
        const float4 a = hot[3 * (size_t)j], b = hot[3 * (size_t)j + 1], c = hot[3 * (size_t)j + 2];
        float4* o = gate + 4 * (size_t)j;
        o[0] = a;
        o[1] = make_float4(b.x, b.y, b.z, nd.mn[0]);
        o[2] = make_float4(c.x, c.y, c.z, nd.mn[1]);
        o[3] = make_float4(nd.mn[2], nd.mx[0], nd.mx[1], nd.mx[2]);
    }
}

// ---- a tree that arrives from the host (pt_upload_scene): Node[] as Accelerator.Build leaves it (Accelerator.cs:113-177)
// What the checks guarantee: no kernel can index out of bounds through these nodes, every node hangs below the root exactly
// once, and the leaves hold as many triangles as were uploaded.
using ptbuild::Node32;
enum { UPLOAD_LEAF_RANGE = 1, UPLOAD_CHILD_INDEX = 2, UPLOAD_UNREFERENCED = 3, UPLOAD_TWICE = 4 };
struct UploadCheck {
    unsigned long long first_error;   // (node index << 3 | kind) of the lowest offending node, ~0 if none
    unsigned long long leaf_tris;     // triangles in all leaves
    uint32_t max_level;               // deepest node (root 0)
    uint32_t unresolved;              // nodes whose depth the jumps did not reach: deeper than 127
};
__global__ void __launch_bounds__(256) k_check_nodes(const Node32* nodes, uint32_t n_nodes, uint32_t n_tris, uint32_t* parent, uint32_t* refs, UploadCheck* chk) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t tris = 0;
    if (i < n_nodes) {
        const Node32 nd = nodes[i];
        if (nd.count > 0) {
            if (nd.leftFirst < 0 || (long long)nd.leftFirst + nd.count > (long long)n_tris) atomicMin(&chk->first_error, ((unsigned long long)i << 3) | UPLOAD_LEAF_RANGE);
            else tris = (uint32_t)nd.count;
        } else {
            const long long l = nd.leftFirst;
            if (l <= (long long)i || (l & 1) == 0 || l + 1 >= (long long)n_nodes) atomicMin(&chk->first_error, ((unsigned long long)i << 3) | UPLOAD_CHILD_INDEX);
            else {
                atomicAdd(&refs[l >> 1], 1u);      // pair (l - 1) / 2
                parent[l] = parent[l + 1] = i;
            }
        }
    }
    // one atomic per block: a single counter word takes them one after the other
    __shared__ unsigned long long s_total;
    if (threadIdx.x == 0) s_total = 0ull;
    __syncthreads();
    unsigned long long total = tris;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if ((threadIdx.x & 63u) == 0u && total) atomicAdd(&s_total, total);
    __syncthreads();
    if (threadIdx.x == 0 && s_total) atomicAdd(&chk->leaf_tris, s_total);
}
__global__ void __launch_bounds__(256) k_check_refs(const uint32_t* refs, uint32_t n_nodes, UploadCheck* chk) {
    const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;   // nodes 2 * pair + 1 and 2 * pair + 2
    if (2u * pair + 2u >= n_nodes) return;
    const uint32_t r = refs[pair];
    if (r != 1u) atomicMin(&chk->first_error, ((unsigned long long)(2u * pair + 1u) << 3) | (r == 0u ? UPLOAD_UNREFERENCED : UPLOAD_TWICE));
}
// depth by pointer jumping: after k doublings dep[i] is the distance to anc[i], the 2^k-th ancestor (or the root)
__global__ void __launch_bounds__(256) k_depth_init(const uint32_t* parent, uint32_t n_nodes, uint32_t* anc, uint32_t* dep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    anc[i] = i ? parent[i] : 0u;
    dep[i] = i ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_depth_jump(const uint32_t* anc_in, const uint32_t* dep_in, uint32_t* anc_out, uint32_t* dep_out, uint32_t n_nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t a = anc_in[i];
    dep_out[i] = dep_in[i] + dep_in[a];
    anc_out[i] = anc_in[a];
}
// the host's node in the form the record kernels read (children side by side at `left`, depth in `level`)
__global__ void __launch_bounds__(256) k_convert(const Node32* nodes, const uint32_t* anc, const uint32_t* dep, uint32_t n_nodes, BNode* out, UploadCheck* chk) {
    __shared__ uint32_t s_max;
    if (threadIdx.x == 0) s_max = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes) {
        const Node32 nd = nodes[i];
        BNode b{};
        for (int a = 0; a < 3; a++) { b.mn[a] = nd.mn[a]; b.mx[a] = nd.mx[a]; }
        const bool leaf = nd.count > 0;
        b.first = leaf ? nd.leftFirst : 0;
        b.count = leaf ? nd.count : 0;
        b.left = leaf ? -1 : nd.leftFirst;
        b.axis = leaf ? 0 : -nd.count - 1;          // the split axis travels in an interior node's negative count (Accelerator.cs:158-159)
        b.level = dep[i];
        out[i] = b;
        if (anc[i] != 0u) atomicOr(&chk->unresolved, 1u);
        atomicMax(&s_max, dep[i]);
    }
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&chk->max_level, s_max);   // one atomic per block: a single counter word takes them one after the other
}

inline hipError_t exclusiveScan(hipStream_t stream, const uint32_t* in, uint32_t* out, size_t n, void** tmp, size_t* tmpBytes) {
    size_t need = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, need, in, out, 0u, n, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    if (need > *tmpBytes) {
        (void)hipFree(*tmp);
        *tmp = nullptr;
        e = ptMalloc(tmp, need);
        if (e != hipSuccess) return e;
        *tmpBytes = need;
    }
    return rocprim::exclusive_scan(*tmp, need, in, out, 0u, n, rocprim::plus<uint32_t>(), stream);
}

}  // namespace ptrepack
