// The scene as instances of meshes, derived on the GPU beside the flattened scene (pt_build_scene_instanced): a tree over the
// instances' world boxes and one tree per mesh in the mesh's own space, built by the same builder as the flattened tree
// (ptbuild::build) and written in the quad-record format of pt_device.hpp (DevScene.iquads / itris / iinst / built_of).
// No reference counterpart: Scene.AddMesh flattens every mesh into world-space triangles (PathTracer/Scene.cs:122-174).
#pragma once
#include "pt_repack.hpp"

namespace ptinst {

using ptbuild::BNode;
using ptbuild::Tri112;

struct HostInstance { float m[16]; int32_t mesh, first; };   // pt_instance of pt_hip.h
static_assert(sizeof(HostInstance) == 72, "pt_instance layout");

// Vector3.Transform(p, M) as the host evaluates it (row vectors): ((x M11 + y M21) + z M31) + M41
__device__ inline void place(const float* p, const float* m, float* out) {
    for (int a = 0; a < 3; a++) out[a] = ((p[0] * m[a] + p[1] * m[4 + a]) + p[2] * m[8 + a]) + m[12 + a];
}

// Every flattened triangle must be the instance matrix applied to the mesh triangle, bit for bit: that is what makes the instanced
// walk's hits the flattened walk's.  One thread per (instance, triangle of its mesh).
__global__ void __launch_bounds__(256) k_verify(const Tri112* flat, const Tri112* mesh, const HostInstance* inst, const int32_t* mesh_first, const int32_t* mesh_count,
                                                 const uint32_t* inst_start, uint32_t n_inst, uint32_t total, uint32_t* mismatches) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    // the instance whose range of the concatenated (instance, k) space holds t: binary search over inst_start (n_inst + 1 entries)
    uint32_t lo = 0, hi = n_inst;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (inst_start[mid] <= t) lo = mid; else hi = mid; }
    const HostInstance& in = inst[lo];
    const uint32_t k = t - inst_start[lo];
    if (k >= (uint32_t)mesh_count[in.mesh]) return;
    const Tri112& o = mesh[mesh_first[in.mesh] + k];
    const Tri112& f = flat[in.first + k];
    bool same = true;
    for (int v = 0; v < 3; v++) {
        float w[3];
        place(o.v[v], in.m, w);
        for (int a = 0; a < 3; a++) same = same && (__float_as_uint(w[a]) == __float_as_uint(f.v[v][a]));
    }
    if (!same) atomicAdd(mismatches, 1u);
}

// built_of[position in the host's list] = position in the built flattened array
__global__ void __launch_bounds__(256) k_invert(const uint32_t* order, uint32_t n, uint32_t* built_of) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) built_of[order[i]] = i;
}

// Per instance (one wave each): the world box of its flattened triangles, the inverse of the matrix's linear part (double
// arithmetic) and how much that inverse can stretch a vector (largest absolute row sum), folded per mesh with an atomic maximum.
struct InstWork { float mn[3], mx[3]; float inv[9]; float stretch; };
__global__ void __launch_bounds__(64) k_instance_boxes(const Tri112* flat, const HostInstance* inst, const int32_t* mesh_count, uint32_t n_inst, InstWork* work,
                                                       uint32_t* mesh_stretch_bits) {
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    if (i >= n_inst) return;
    const HostInstance& in = inst[i];
    const uint32_t n = (uint32_t)mesh_count[in.mesh];
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (uint32_t k = lane; k < n; k += 64u) {
        const Tri112& t = flat[in.first + k];
        for (int v = 0; v < 3; v++)
            for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], t.v[v][a]); mx[a] = fmaxf(mx[a], t.v[v][a]); }
    }
    for (int a = 0; a < 3; a++)
        for (int off = 32; off > 0; off >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off)); }
    if (lane != 0) return;
    InstWork w;
    for (int a = 0; a < 3; a++) { w.mn[a] = mn[a]; w.mx[a] = mx[a]; }
    // L = rows 0..2 of the matrix (row vectors: world = p L + T); A = L^-1 by cofactors
    const double a = in.m[0], b = in.m[1], c = in.m[2], d = in.m[4], e = in.m[5], f = in.m[6], g = in.m[8], h = in.m[9], k9 = in.m[10];
    const double det = a * (e * k9 - f * h) - b * (d * k9 - f * g) + c * (d * h - e * g);
    const double r = 1.0 / det;
    const double A[9] = {(e * k9 - f * h) * r, (c * h - b * k9) * r, (b * f - c * e) * r, (f * g - d * k9) * r, (a * k9 - c * g) * r, (c * d - a * f) * r,
                         (d * h - e * g) * r, (b * g - a * h) * r, (a * e - b * d) * r};
    float stretch = 0.0f;
    for (int row = 0; row < 3; row++) {
        float sum = 0.0f;
        for (int col = 0; col < 3; col++) { w.inv[3 * row + col] = (float)A[3 * row + col]; sum += fabsf(w.inv[3 * row + col]); }
    }
    // o' = rel A (row vector times A): component j = sum_i rel_i A[i][j]: a vector of length 1 stretches at most to the largest column sum
    for (int col = 0; col < 3; col++) {
        float sum = 0.0f;
        for (int row = 0; row < 3; row++) sum += fabsf(w.inv[3 * row + col]);
        stretch = fmaxf(stretch, sum);
    }
    if (!(stretch == stretch) || !(fabs(det) > 0.0)) stretch = 3.4e38f;   // a singular matrix: the scene is refused
    w.stretch = stretch;
    work[i] = w;
    atomicMax(&mesh_stretch_bits[in.mesh], __float_as_uint(stretch));   // positive floats order like their bits
}

// One stand-in triangle per instance whose box is the instance's widened world box: the builder makes the top-level tree of them.
__global__ void __launch_bounds__(256) k_instance_prims(const HostInstance* inst, const InstWork* work, const float* mesh_pad_obj, uint32_t n_inst, Tri112* prims) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inst) return;
    const InstWork& w = work[i];
    // the mesh tree's boxes are widened by pad_obj in the mesh's space; in the world that is at most pad_obj * the matrix's stretch
    float grow = 0.0f;
    for (int col = 0; col < 3; col++) {
        float sum = 0.0f;
        for (int row = 0; row < 3; row++) sum += fabsf(inst[i].m[4 * row + col]);
        grow = fmaxf(grow, sum);
    }
    const float pad = mesh_pad_obj[inst[i].mesh] * grow * 1.001f;
    Tri112 t{};
    for (int a = 0; a < 3; a++) { t.v[0][a] = w.mn[a] - pad; t.v[1][a] = w.mx[a] + pad; t.v[2][a] = w.mn[a] - pad; }
    prims[i] = t;
}

// Quad records of one tree into the shared arrays: interior entries shifted by quad_base, leaf ranges by tri_base, big-leaf rows by
// big_base, every box widened by pad.
__global__ void __launch_bounds__(256) k_irecords(const BNode* nodes, uint32_t n_nodes, const uint32_t* quad_idx, const uint32_t* big_idx, const uint32_t* big_flag,
                                                  uint32_t quad_base, uint32_t tri_base, uint32_t big_base, float pad, float4* quads, uint2* big) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const BNode& nd = nodes[i];
    if (big_flag[i]) big[big_base + big_idx[i]] = make_uint2((uint32_t)nd.first + tri_base, (uint32_t)nd.count);
    if (!(nd.left >= 0 && (nd.level & 1u) == 0u)) return;
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
        for (int a = 0; a < 3; a++) { box[a][k] = kd.mn[a] - pad; box[3 + a][k] = kd.mx[a] + pad; }
        if (kd.left >= 0) ent[k] = quad_base + quad_idx[kids[k]] + 1u;
        else if (ptrepack::fitsPacked(kd.first + (int32_t)tri_base, kd.count)) ent[k] = PT_LEAF_FLAG | ((uint32_t)(kd.count - 1) << 27) | ((uint32_t)kd.first + tri_base);
        else ent[k] = PT_LEAF_FLAG | (15u << 27) | (big_base + big_idx[kids[k]]);
    }
    float4* q = quads + 8 * (size_t)(quad_base + quad_idx[i]);
    for (int a = 0; a < 6; a++) q[a] = make_float4(box[a][0], box[a][1], box[a][2], box[a][3]);
    q[6] = make_float4(__uint_as_float(ent[0]), __uint_as_float(ent[1]), __uint_as_float(ent[2]), __uint_as_float(ent[3]));
    q[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
// whether a leaf of a tree needs a big-leaf row once its range is shifted by tri_base (k_flags decides for unshifted ranges)
__global__ void __launch_bounds__(256) k_iflags(const BNode* nodes, uint32_t n_nodes, uint32_t tri_base, uint32_t* quad_flag, uint32_t* big_flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const BNode& nd = nodes[i];
    quad_flag[i] = (nd.left >= 0 && (nd.level & 1u) == 0u) ? 1u : 0u;
    big_flag[i] = (nd.left < 0 && nd.count > 0 && !ptrepack::fitsPacked(nd.first + (int32_t)tri_base, nd.count)) ? 1u : 0u;
}

// Mesh triangles in their tree's order: the three object-space positions, and in the first .w the triangle's place in the mesh as
// the host listed it (order: tree position -> listed position).
__global__ void __launch_bounds__(256) k_itris(const Tri112* sorted, const uint32_t* order, uint32_t n, uint32_t tri_base, float4* itris) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Tri112& t = sorted[i];
    float4* o = itris + 3 * (size_t)(tri_base + i);
    o[0] = make_float4(t.v[0][0], t.v[0][1], t.v[0][2], __uint_as_float(order[i]));
    o[1] = make_float4(t.v[1][0], t.v[1][1], t.v[1][2], 0.0f);
    o[2] = make_float4(t.v[2][0], t.v[2][1], t.v[2][2], 0.0f);
}

// The 128-byte instance records in the top-level tree's order (order: tree slot -> instance as the host listed it).
struct MeshInfo { uint32_t root_entry; float mn[3], mx[3]; };   // entry of the mesh tree's root; its widened box
__global__ void __launch_bounds__(256) k_iinst(const HostInstance* inst, const InstWork* work, const uint32_t* order, const MeshInfo* meshes, uint32_t n_inst, float4* out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_inst) return;
    const uint32_t i = order[s];
    const HostInstance& in = inst[i];
    const InstWork& w = work[i];
    const MeshInfo& m = meshes[in.mesh];
    float4* o = out + 8 * (size_t)s;
    for (int a = 0; a < 3; a++) o[a] = make_float4(in.m[a], in.m[4 + a], in.m[8 + a], in.m[12 + a]);
    for (int a = 0; a < 3; a++) o[3 + a] = make_float4(w.inv[a], w.inv[3 + a], w.inv[6 + a], in.m[12 + a]);
    o[6] = make_float4(__uint_as_float(m.root_entry), __uint_as_float((uint32_t)in.first), m.mn[0], m.mn[1]);
    o[7] = make_float4(m.mn[2], m.mx[0], m.mx[1], m.mx[2]);
}

}  // namespace ptinst
