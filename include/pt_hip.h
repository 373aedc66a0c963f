/*
 * pt_hip.h — C ABI of libpt_hip.so, the MI355X (gfx950) wavefront path-tracing core.
 *
 * Drop-in boundary: the reference's C# host keeps Scene / Camera / Material /
 * Triangle / Accelerator as they are and replaces its OpenGL compute dispatch
 * by P/Invoke into these entry points.  Every array crosses the boundary in
 * the host's own blittable layout:
 *
 *   Material  48 B  PathTracer/Material.cs:7-28      (Color@0 Emissive@16 Type@32 Index@36 Roughness@40)
 *   Triangle 112 B  PathTracer/Triangle.cs:7-13      (V1@0 V2@32 V3@64 Material@96; Vertex = Position@0 Normal@16)
 *   Node      32 B  PathTracer/Node.cs:5-10          (Bounds.Min@0 Bounds.Max@12 LeftFirst@24 Count@28)
 *   RenderState 32 B, FrameState 112 B               PathTracer/State.cs:20-41
 *
 * One context can drive several GPUs of a node (pt_config.device_count): the host
 * that replaces Wavefront.cs:40-56 gets them from the same create / render /
 * read-image calls; tile sharding and the RCCL gather are inside the library.
 *
 * Every call makes the context's GPU (for a multi-GPU context: one of its GPUs)
 * the calling thread's current HIP device and leaves it so.
 *
 * All functions return PT_OK (0) or a negative pt_status; nothing throws across
 * the boundary.  pt_last_error() returns a message owned by the context (or by
 * the library when ctx is NULL).  A context is used by one thread at a time.
 * The host owns every input array and every output buffer; the library owns
 * all device memory.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_ctx pt_ctx;

typedef enum pt_status {
    PT_OK = 0,
    PT_E_INVALID_ARG = -1,
    PT_E_NO_DEVICE = -2,
    PT_E_OOM = -3,
    PT_E_HIP = -4,
    PT_E_BAD_BVH = -5,
    PT_E_STATE = -6,
    PT_E_RCCL = -7
} pt_status;

/* Constants the reference hard-codes, as fields (reference values are the defaults
 * written by pt_default_config). */
typedef struct pt_config {
    int32_t struct_size;        /* sizeof(pt_config), for versioning */
    int32_t device;             /* HIP device ordinal */
    int32_t nee;                /* next-event estimation at diffuse vertices (not in the reference): 0 off, 1 on,
                                   2 on with multiple importance sampling (balance heuristic with the BRDF sample) */
    int32_t rng_mode;           /* 0 = reference seeding (generate.glsl:174, shade.glsl:352), 1 = decorrelated hash */
    int32_t tile_rank;          /* this context renders pixel groups g with g % tile_world == tile_rank */
    int32_t tile_world;
    int32_t samples_per_batch;  /* samples traced concurrently; 0 = as many as give 256M paths per launch (176 bytes of device memory per path,
                                   at most a quarter of what is free); a number bounds the memory a render call takes */
    int32_t traversal_stats;    /* 1 = count BVH steps / triangle tests (slower instrumented kernels) */
    int32_t stage_timing;       /* 1 = bracket every stage launch with events (pt_get_stats ms_*), stages back to back on one stream;
                                   2 = the extend launches only (a pair of events costs about 5 us of stream time).  With either,
                                   every stage has launches of its own; with 0 the shadow rays of a bounce may be traced inside the
                                   next bounce's extend launch */
    int32_t device_count;       /* GPUs behind this one context: devices device .. device + device_count - 1.  With more than one,
                                   the scene is replicated, the image's 8x8 pixel groups are dealt round-robin to the GPUs, and
                                   pt_read_image gathers the tiles to the first GPU with RCCL (grouped ncclSend / ncclRecv, once per
                                   image); every other call keeps its meaning.  tile_rank / tile_world must then be 0 / 1. */
    int32_t group_flags;        /* PT_GROUP_* */
    int32_t reserved[5];
} pt_config;

/* pt_config.group_flags */
enum { PT_GROUP_SAME_DEVICE = 1 /* rehearsal on a single GPU: all device_count members run on `device`; their tiles are gathered
                                   by device-to-device copies, RCCL having one rank per GPU */,
       PT_GROUP_FORCE_RCCL = 2  /* take the RCCL path even with device_count = 1 (a communicator of one rank that sends to
                                   itself): exercises library loading, communicator set-up and the send/receive calls */ };

typedef struct pt_stats {
    uint32_t struct_size;       /* IN: sizeof(pt_stats) as the caller was compiled (the struct grows at its end from version to
                                   version); pt_get_stats writes no more than this many bytes and stores the count it wrote here */
    uint32_t reserved0;
    uint64_t paths;             /* primary rays generated */
    uint64_t extend_rays;       /* rays traced by extend (primary + bounce) */
    uint64_t shadow_rays;       /* rays traced by connect */
    uint64_t extend_steps;      /* node records fetched in extend: quad records, plus pair records for the rays
                                   on the two-wide walk   (traversal_stats) */
    uint64_t extend_tris;       /* triangle tests in extend      (traversal_stats) */
    uint64_t shadow_steps;
    uint64_t shadow_tris;
    uint64_t stack_overflows;   /* rays dropped because the traversal stack was exhausted (must stay 0) */
    uint32_t extend_launches, shade_launches, connect_launches, generate_launches;
    float ms_generate, ms_extend, ms_shade, ms_connect;   /* summed event time per stage (stage_timing) */
    float ms_render;            /* wall time of the last pt_render + pt_synchronize pair measured by events */
    float ms_extend_primary;    /* the part of ms_extend spent on camera rays (the extend launch of bounce 0; its rays = paths) */
    float reserved[2];
    /* extend stage, traversal_stats only: iterations a wave spent on interior steps, on leaf (triangle) steps and on
     * retiring/refilling lanes; extend_steps / (64 * sched_interior) is the lane utilisation of the interior loop */
    uint64_t sched_interior, sched_leaf, sched_refill;
    uint64_t special_rays;      /* rays (extend + connect) that took the two-wide walk: zero/denormal/non-finite components */
    /* extend stage, traversal_stats only: node records / triangle records fetched, counting the lanes of one wave instruction
     * that ask for the same record once (the vector memory pipeline serves them together): what the stage's loads cost */
    uint64_t extend_unique_steps, extend_unique_tris;
    uint64_t shadow_occluded;    /* shadow rays that found an occluder (traversal_stats) */
    uint64_t device_allocations; /* hipMalloc calls the library has made in this process so far (all contexts): unchanged across a
                                   call means the call allocated no device memory */
} pt_stats;

/* ---- lifetime --------------------------------------------------------------
 * Replaces: new Wavefront() + the three ShaderProgram compiles, PathTracer/Wavefront.cs:22-38. */
int pt_default_config(pt_config* cfg);
int pt_create(const pt_config* cfg, pt_ctx** out);
int pt_destroy(pt_ctx* ctx);
const char* pt_last_error(pt_ctx* ctx);

/* Replaces: Scene.CopyToDevice -> Buffer<T>.CopyToDevice for bindings 10/11/12,
 * PathTracer/Scene.cs:176-183, PathTracer/Library/Graphics/Buffer.cs:94-126.
 * Copies the three arrays to the GPU as they are; there the tree is checked (child indices, leaf ranges, every node
 * referenced once, the leaves' triangle count, the depth against the traversal stack, material indices: PT_E_BAD_BVH /
 * PT_E_INVALID_ARG, the scene installed before stays) and repacked into 128-byte quad records (every other level of
 * the tree), 64-byte sibling-pair records, 48-byte hot/cold triangle records and the emissive-triangle table. */
int pt_upload_scene(pt_ctx* ctx, const void* materials, int32_t n_materials,
                    const void* triangles, int32_t n_triangles,
                    const void* nodes, int32_t n_nodes);

/* Replaces: Game.Resize -> new Image(w,h) + Wavefront.Allocate(w*h), PathTracer/Game.cs:138-151,
 * PathTracer/Wavefront.cs:67-81.  (Re)allocates ray queues and the accumulator, zeroes it. */
int pt_resize(pt_ctx* ctx, int32_t width, int32_t height);

/* Replaces: renderState.CopyToDevice (UBO 0), PathTracer/Game.cs:147-150.  Width/Height must equal pt_resize's. */
int pt_set_render_state(pt_ctx* ctx, const void* render_state32);

/* Replaces: frameState.Data.Samples = 0 (camera moved / key R), PathTracer/Game.cs:157-165. */
int pt_reset_accumulation(pt_ctx* ctx);

/* Replaces: frameState.CopyToDevice + Wavefront.Invoke, PathTracer/Game.cs:187-191, PathTracer/Wavefront.cs:40-56.
 * Traces `spp` samples; sample s uses frames = FrameState.Frames + s.  spp = 1, max_bounces = 10 is one
 * Wavefront.Invoke().  FrameState.Samples == 0 restarts the accumulation first (the reference's running mean,
 * shade.glsl:89-96, forgets the old value when samples is 0); any other value adds to what is there.  Work is enqueued on the context's stream; the call does not wait for it. */
int pt_render(pt_ctx* ctx, const void* frame_state112, int32_t spp, int32_t max_bounces);
int pt_synchronize(pt_ctx* ctx);

/* Replaces: the rgba32f image bound at unit 0 that the quad shader displays, PathTracer/Library/Graphics/Image.cs:7-19,
 * PathTracer/Shaders/Quad/fragment.glsl:7-11.  Blocking; rgba = width*height*4 floats, pixel = x + y*width, row 0 = bottom.
 * Value = sum of sample contributions / samples accumulated since the last reset. */
int pt_read_image(pt_ctx* ctx, float* rgba);

/* GPU twin of Scene.Intersect, PathTracer/Scene.cs:35-45 / PathTracer/Accelerator.cs:51-111 (used by
 * Camera.GetFocalDistance, PathTracer/Camera.cs:89-95).  o_d = n x 6 floats (origin, direction).
 * tri = index into the uploaded triangle array, -1 = miss; t = distance (+inf on a miss); uv may be NULL. */
int pt_trace_closest(pt_ctx* ctx, const float* o_d, int32_t n, int32_t* tri, float* t, float* uv);

/* Any hit with 0 < t < tmax[i] (the connect stage's query).  No reference counterpart: the reference has no shadow rays. */
int pt_trace_occluded(pt_ctx* ctx, const float* o_d, const float* tmax, int32_t n, uint8_t* occluded);

/* pt_trace_closest for callers that cannot allocate device memory: tight serving loops, harnesses that forbid hipMalloc,
 * hosts that are capturing a HIP graph on ANOTHER stream of the process (an allocation there would invalidate the capture).
 * It has no counterpart in the reference.  Same arguments and results as pt_trace_closest, but the rays and hits are staged
 * in a fixed device array of the library (40 bytes per ray, PT_NOALLOC_MAX_RAYS rays = 2.5 MiB held on every GPU the library
 * runs on until the process ends) instead of buffers allocated per call: the call makes no hipMalloc / hipFree.  The call
 * itself copies from and to the caller's host arrays on the context's stream and waits for it: it blocks, and it cannot be
 * recorded into a graph.  n > PT_NOALLOC_MAX_RAYS is PT_E_INVALID_ARG.  All calls on one GPU share that array: calls that
 * overlap on one GPU (two contexts, two threads) race. */
#define PT_NOALLOC_MAX_RAYS 65536
int pt_trace_closest_noalloc(pt_ctx* ctx, const float* o_d, int32_t n, int32_t* tri, float* t, float* uv);

/* Runs the generate stage alone (generate.glsl:170-200) for local thread indices [0,n) of one sample and copies the
 * rays back: o_d n x 6, pixel n (0xFFFFFFFF where the thread emits no ray).  Parity/debug entry. */
int pt_generate_rays(pt_ctx* ctx, const void* frame_state112, int32_t n, float* o_d, uint32_t* pixel);

/* Measurement entry (no reference counterpart): uploads n rays once, runs the extend stage alone over them `iters`
 * times exactly as pt_render launches it, and returns the mean launch time from events on the context's stream.
 * tri / t (may be NULL) receive the hits of the last launch. */
int pt_bench_extend(pt_ctx* ctx, const float* o_d, int32_t n, int32_t iters, float* avg_ms, int32_t* tri, float* t);

/* out->struct_size must hold the caller's sizeof(pt_stats) (at least 16); fields beyond it are not written. */
int pt_get_stats(pt_ctx* ctx, pt_stats* out);
int pt_reset_stats(pt_ctx* ctx);

/* Accelerator.Build on the GPU, PathTracer/Accelerator.cs:113-310: binned SAH (8 bins, leaves below 3 primitives).
 * triangles_in: n_tri Triangle records in Accelerator.AddTriangle order.  triangles_out (n_tri records) and nodes_out
 * (room for node_capacity >= max(1, 2*n_tri-1) Node records) receive what Build leaves in its two Buffer<T> mirrors —
 * the same bytes the one-thread CPU build produces — ready for pt_upload_scene and for the host's own
 * Accelerator.Intersect.  *n_nodes: nodes written.  build_ms (may be NULL): device time of the build itself,
 * without the transfers. */
int pt_build_bvh(pt_ctx* ctx, const void* triangles_in, int32_t n_tri, void* triangles_out, void* nodes_out, int32_t node_capacity,
                 int32_t* n_nodes, float* build_ms);

/* Scene.CopyToDevice with the build on the GPU (Scene.cs:176-183 + Accelerator.Build): builds the tree as pt_build_bvh
 * does and derives the device scene from it without visiting the host — the one call replaces Accelerator.Build +
 * pt_upload_scene.  materials: Material[n_mat]; triangles_in: Accelerator.AddTriangle order.  triangles_out / nodes_out
 * (each may be NULL) receive the built arrays for the host's own mirrors; node_capacity as for pt_build_bvh. */
int pt_build_scene(pt_ctx* ctx, const void* materials, int32_t n_mat, const void* triangles_in, int32_t n_tri, void* triangles_out,
                   void* nodes_out, int32_t node_capacity, int32_t* n_nodes, float* build_ms);

/* pt_build_scene for a scene made of meshes placed by matrices, with a second acceleration structure over the instances (no
 * counterpart in the reference, whose Scene.AddMesh flattens every mesh into world-space triangles, PathTracer/Scene.cs:122-174).
 * The flattened scene is built and installed exactly as by pt_build_scene — triangle ids in results, materials, normals and the
 * default traversal are its — and beside it a tree over the instances' world boxes plus one tree per mesh in the mesh's own space
 * (a 20M-triangle scene of 4,026 bunnies walks 0.7 MB of records instead of 1.6 GB).  pt_set_option(PT_OPT_INSTANCED, 1) makes
 * the traversal kernels walk that structure.  Triangles are still tested in world space on vertices re-derived from the mesh
 * vertex and the instance matrix, which must reproduce the flattened triangle bit for bit (checked: PT_E_INVALID_ARG otherwise),
 * so a hit's distance and barycentrics are the flattened walk's; what differs is eligibility: the flattened walk, like the
 * reference, drops a hit whose leaf box the ray misses by rounding, this walk does not, and exact distance ties go to the triangle
 * listed first rather than to the smaller built index.
 * triangles_in: the flattened list in Accelerator.AddTriangle order.  mesh_triangles: every mesh's triangles in its own space,
 * mesh m at [mesh_first[m], mesh_first[m] + mesh_count[m]).  instances: in any order, together tiling triangles_in — instance i
 * owns the mesh_count[mesh] triangles from first_triangle on, triangle k of them being mesh triangle k under
 * System.Numerics.Vector3.Transform(vertex, transform) (PathTracer/Library/Geometry/Mesh.cs:50-70). */
typedef struct pt_instance {
    float transform[16];        /* Matrix4x4 M11..M44 row by row, row-vector convention (translation in M41..M43) */
    int32_t mesh;
    int32_t first_triangle;
} pt_instance;
int pt_build_scene_instanced(pt_ctx* ctx, const void* materials, int32_t n_mat, const void* triangles_in, int32_t n_tri,
                             const void* mesh_triangles, const int32_t* mesh_first, const int32_t* mesh_count, int32_t n_meshes,
                             const pt_instance* instances, int32_t n_instances, float* build_ms);

/* Runtime switches: key is one of PT_OPT_*; takes effect at the next pt_render. */
enum { PT_OPT_NEE = 1, PT_OPT_RNG_MODE = 2, PT_OPT_TRAVERSAL_STATS = 3, PT_OPT_STAGE_TIMING = 4, PT_OPT_SAMPLES_PER_BATCH = 5,
       PT_OPT_TRAVERSE_BLOCKS_PER_CU = 6 /* resident 256-thread blocks per CU of the traversal kernels; 0 = the default (at most 7).
                                            Once set it holds for every call; left alone, a call that shares the GPU with another
                                            frame launches four per CU (PT_OPT_PIPELINE_CALLS) */,
       PT_OPT_REFILL_PRIMARY = 7 /* idle lanes (1..64) a traversal wave collects before it fetches new camera rays; default 48 */,
       PT_OPT_REFILL_BOUNCE = 8 /* the same for bounce and shadow rays; default 40 */,
       PT_OPT_QUANTIZED = 9 /* which node records the traversal kernels walk: 0 the exact 128-byte records, 1 the 64-byte records
                               with boxes on a 16-bit grid (the exact leaf boxes are then applied at the triangles: hits are the
                               same), 2 (default) the 64-byte records for trees that exceed the L2 caches */,
       PT_OPT_INSTANCED = 10 /* 1: walk the instances of a scene installed by pt_build_scene_instanced (0, the default, and every
                                scene installed another way: the flattened triangles) */,
       PT_OPT_PIPELINE_CALLS = 11 /* 1 (default): batches of up to 64M paths (consecutive pt_render calls, and the batches of one call)
                                     are traced on two alternating sets of streams and queues, so that two frames are in flight
                                     together, fill each other's tails, and one's shade stage runs under the other's traversal (a call
                                     that finds the other set busy launches smaller traversal grids and shade blocks to that end);
                                     their samples still enter the accumulator in call order, on the context's stream.  0: one set */ };
int pt_set_option(pt_ctx* ctx, int32_t key, int32_t value);

/* ---- multi-GPU tiles, one process per GPU --------------------------------------
 * (A single process reaches several GPUs through pt_config.device_count instead and needs none of this.)
 * Each rank accumulates only its own pixel groups.  pt_accum_device exposes the rank-local accumulator
 * (float4 per local slot, slots = local groups * 64, padded to the same length on every rank) so the caller can
 * gather it with RCCL; pt_assemble_tiles de-interleaves `world` gathered accumulators into a row-major image.
 * No reference counterpart (the reference is single-GPU). */
int pt_accum_device(pt_ctx* ctx, void** device_ptr, uint64_t* n_slots_padded, uint32_t* samples);
int pt_assemble_tiles(pt_ctx* ctx, const void* gathered_device, int32_t world, uint32_t samples, float* rgba_host);

/* Stream the context enqueues on (hipStream_t as void*), for callers that order their own work after it.  pt_render also uses a
 * second, internal stream (the shadow rays of a bounce are traced beside the next bounce's rays); it is joined into this one before
 * pt_render returns, so work ordered after this stream is ordered after all of a render call. */
int pt_stream(pt_ctx* ctx, void** stream);

#ifdef __cplusplus
}
#endif
#endif /* PT_HIP_H */
