/*
 * pt_host.h — C ABI of libpt_host.so, the C++ host mirror that stands in for the reference's C# host
 * classes on machines without .NET: Scene, Accelerator (binned-SAH BVH build), Mesh loading, the bundled
 * scene recipe and the Camera math.  It produces exactly the arrays libpt_hip.so consumes (pt_hip.h).
 * A C# host does not need this library: it already owns these classes.
 *
 * All functions return 0 on success and -1 on failure (message in pth_last_error()).
 */
#ifndef PT_HOST_H
#define PT_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene pth_scene;

const char* pth_last_error(void);

/* new Scene(), PathTracer/Scene.cs:18-26.  mesh_source: a directory ending in '/' that holds <name>.obj files,
 * or the path of a parsed-mesh blob (pathtracer_amd/assets/meshes.bin). */
pth_scene* pth_scene_create(const char* mesh_source);
void pth_scene_destroy(pth_scene* s);

/* Scene.Clear, PathTracer/Scene.cs:28-33 */
int pth_scene_clear(pth_scene* s);

/* Scene.AddMesh / AddMeshNormalized (8 overloads), PathTracer/Scene.cs:47-85,122-157.
 * matrix16: row-major System.Numerics.Matrix4x4 or NULL (identity); material48: a Material, or NULL to take
 * one diffuse material per OBJ group from the MTL file. */
int pth_scene_add_mesh(pth_scene* s, const char* name, int normalized, const float* matrix16, const void* material48);

/* Scene.GenerateTiledFoloor, PathTracer/Scene.cs:87-120 */
int pth_scene_generate_tiled_floor(pth_scene* s, const void* material_a48, const void* material_b48);

/* Scene.GetMaterialIndex, PathTracer/Scene.cs:192-204: index of an equal material, appended if new */
int pth_scene_get_material_index(pth_scene* s, const void* material48, int32_t* index);

/* Accelerator.AddTriangle, PathTracer/Accelerator.cs:38-49 (n Triangle records of 112 bytes) */
int pth_scene_add_triangles(pth_scene* s, const void* triangles112, int32_t n);

/* Game.InitializeScene(seed), PathTracer/Game.cs:66-136: light + tiled floor + 60 random meshes */
int pth_scene_initialize(pth_scene* s, int32_t seed);

/* The bundled recipe followed by `extra_bunnies` more objects, all bunnies, drawn from the same random stream
 * (19 extra = 259,600 triangles, the Sponza-scale synthetic input).  No reference counterpart. */
int pth_scene_initialize_extended(pth_scene* s, int32_t seed, int32_t extra_bunnies);

/* `count` randomly rotated bunnies on a cubic grid, flattened to triangles (4,026 = 20.0M triangles), plus the light.
 * No reference counterpart. */
int pth_scene_generate_instances(pth_scene* s, int32_t count, int32_t seed);

/* Synthetic random triangle soup (no reference counterpart): n triangles in [-50,50]^3 with one emissive quad. */
int pth_scene_generate_soup(pth_scene* s, int32_t n_triangles, uint32_t seed);

/* Accelerator.Build, PathTracer/Accelerator.cs:113-310 (reached through Scene.CopyToDevice, Scene.cs:176-183).
 * threads <= 1 builds on the calling thread; more threads build disjoint subtrees concurrently.  The node and
 * triangle arrays are identical for every thread count. */
int pth_scene_build(pth_scene* s, int32_t threads);

/* Cache file of a built scene (no reference counterpart: the reference rebuilds at every start): "PTSCENE1", int32
 * counts of materials, triangles, nodes and a zero, then Material[], Triangle[] (built order) and Node[] verbatim.
 * pth_scene_load replaces the scene's contents; pth_scene_arrays then returns the stored arrays without a build. */
int pth_scene_save(pth_scene* s, const char* path);
int pth_scene_load(pth_scene* s, const char* path);

/* out4 = { materials, triangles, nodes, tree depth } */
int pth_scene_counts(pth_scene* s, int32_t* out4);

/* The CPU mirrors of Buffer<Material>, Buffer<Triangle>, Buffer<Node> (valid until the scene changes). */
int pth_scene_arrays(pth_scene* s, const void** materials48, const void** triangles112, const void** nodes32);

/* The triangles as Accelerator.AddTriangle collected them (PathTracer/Accelerator.cs:38-49: insertion order, before any
 * build) and the materials: the input of pt_build_bvh / pt_build_scene when the tree is to be built on the GPU.
 * Valid until the scene changes. */
int pth_scene_references(pth_scene* s, const void** triangles112, const void** materials48, int32_t* n_triangles, int32_t* n_materials);

/* The scene as meshes and instances, for pt_build_scene_instanced (no reference counterpart: Scene.AddMesh flattens every mesh into
 * world-space triangles, PathTracer/Scene.cs:122-174).  Every pth_scene_add_mesh call (and every object of the scene recipes) is an
 * instance of the mesh (name, normalized) under its matrix; runs of triangles that did not come from a mesh become meshes of their own,
 * instanced once under the identity.  Every triangle of pth_scene_references belongs to exactly one instance, in order, and
 * Vector3.Transform(object-space vertex, instance matrix) reproduces its vertices bit for bit.
 * Call with null arrays for the three counts, then with arrays of those sizes: mesh_triangles112 (object space, 112 B each),
 * mesh_first / mesh_count (one per mesh: its range in mesh_triangles112), instances72 (pt_instance records of pt_hip.h). */
int pth_scene_instancing(pth_scene* s, int32_t* n_meshes, int32_t* n_mesh_triangles, int32_t* n_instances, void* mesh_triangles112,
                         int32_t* mesh_first, int32_t* mesh_count, void* instances72);

/* Camera.Update, PathTracer/Camera.cs:46-61, in two halves around the autofocus ray of
 * Camera.GetFocalDistance (Camera.cs:89-95), which the caller traces with pt_trace_closest:
 * pth_camera_basis gives forward/right/up; pth_camera_state takes the hit distance of the ray
 * (position, forward) — +inf for a miss — and fills CameraState (96 bytes) and the focal distance. */
int pth_camera_basis(const float* position3, const float* target3, float* forward3, float* right3, float* up3);
int pth_camera_state(const float* position3, const float* forward3, const float* right3, const float* up3,
                     float hit_distance, int32_t width, int32_t height, void* camera_state96, float* focal);

/* An image file from the rgba32f read-out of pt_read_image (row 0 = bottom row, as the reference's display reads it,
 * PathTracer/Shaders/Quad/fragment.glsl:9-10; the reference itself never writes a file).
 * format:   PTH_IMAGE_PFM  RGB float32, lossless;  PTH_IMAGE_PPM  binary P6;  PTH_IMAGE_PNG  8-bit RGB;
 *           PTH_IMAGE_EXR  OpenEXR 2 scan lines, float32 R/G/B, uncompressed, lossless (top row first, as the format has it).
 * transfer (8-bit formats): PTH_TRANSFER_DISPLAY  what the reference's window shows: clamp to [0,1], round(255 v);
 *                           PTH_TRANSFER_SRGB     v * exposure, clamp, sRGB encoding. */
enum { PTH_IMAGE_PFM = 0, PTH_IMAGE_PPM = 1, PTH_IMAGE_PNG = 2, PTH_IMAGE_EXR = 3 };
enum { PTH_TRANSFER_DISPLAY = 0, PTH_TRANSFER_SRGB = 1 };
int pth_image_write(const char* path, const float* rgba, int32_t width, int32_t height, int32_t format, int32_t transfer, float exposure);

#ifdef __cplusplus
}
#endif
#endif /* PT_HOST_H */
