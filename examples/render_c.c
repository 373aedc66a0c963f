/*
 * A host written in plain C against the two C ABIs, the way the reference's C# host would use them through P/Invoke:
 *   libpt_host.so (pt_host.h)  Scene recipe, BVH build, camera   — stands in for the C# classes that a .NET host keeps
 *   libpt_hip.so  (pt_hip.h)   upload, render, read the image    — what replaces Wavefront.cs and the GL buffers
 *
 *   render_c <meshes.bin> <width> <height> <spp> <bounces> <out.raw|out.png|out.pfm> [gpu-build] [devices=N] [same-device] [force-rccl]
 *
 * devices=N renders on N GPUs through the same calls (pt_config.device_count): tiles are sharded and gathered inside the library,
 * and the image is the same bits as on one GPU.  same-device puts all N members on GPU 0 (a rehearsal for machines with one GPU).
 *
 * Game.Initialize order (Game.cs:29-55): camera set at (0,2,15) -> (0,3,0) before the scene exists (focal distance 20),
 * InitializeScene(11), CopyToDevice, then frames with Frames = Samples = k.  Writes width*height float4 (or, for an output
 * name ending in .png / .pfm, an sRGB PNG / a float PFM) and prints an FNV-1a checksum of the float4 bytes.  With a 7th argument the tree is built on the GPU (pt_build_scene) from the host-built
 * triangle order, which is a valid insertion order of its own.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pt_hip.h"
#include "pt_host.h"

#define HOST(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, pth_last_error()); return 1; } } while (0)
#define GPU(call) do { if ((call) != PT_OK) { fprintf(stderr, "%s: %s\n", #call, pt_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s meshes.bin width height spp bounces out.raw [gpu-build] [devices=N] [same-device] [force-rccl]\n", argv[0]); return 2; }
    int gpu_build = 0, devices = 1, flags = 0;
    for (int a = 7; a < argc; a++) {
        if (strcmp(argv[a], "gpu-build") == 0) gpu_build = 1;
        else if (strncmp(argv[a], "devices=", 8) == 0) devices = atoi(argv[a] + 8);
        else if (strcmp(argv[a], "same-device") == 0) flags |= PT_GROUP_SAME_DEVICE;
        else if (strcmp(argv[a], "force-rccl") == 0) flags |= PT_GROUP_FORCE_RCCL;
        else { fprintf(stderr, "unknown option %s\n", argv[a]); return 2; }
    }
    const int width = atoi(argv[2]), height = atoi(argv[3]), spp = atoi(argv[4]), bounces = atoi(argv[5]);

    pth_scene* scene = pth_scene_create(argv[1]);
    if (!scene) { fprintf(stderr, "pth_scene_create: %s\n", pth_last_error()); return 1; }
    HOST(pth_scene_initialize(scene, 11));
    HOST(pth_scene_build(scene, 4));
    int32_t counts[4];
    const void *materials, *triangles, *nodes;
    HOST(pth_scene_counts(scene, counts));
    HOST(pth_scene_arrays(scene, &materials, &triangles, &nodes));

    pt_ctx* ctx = NULL;
    pt_config cfg;
    pt_default_config(&cfg);
    cfg.nee = 1;
    cfg.rng_mode = 1;
    cfg.device_count = devices;
    cfg.group_flags = flags;
    GPU(pt_create(&cfg, &ctx));
    if (gpu_build) {
        int32_t n_nodes = 0;
        GPU(pt_build_scene(ctx, materials, counts[0], triangles, counts[1], NULL, NULL, 0, &n_nodes, NULL));
    } else {
        GPU(pt_upload_scene(ctx, materials, counts[0], triangles, counts[1], nodes, counts[2]));
    }
    GPU(pt_resize(ctx, width, height));
    struct { int32_t width, height, pad[2]; float sky[4]; } rs = {width, height, {0, 0}, {142.0f / 255.0f, 178.0f / 255.0f, 237.0f / 255.0f, 1.0f}};   /* State.cs:20-27, Game.cs:13 */
    GPU(pt_set_render_state(ctx, &rs));

    /* Camera.Set before the scene exists: the autofocus ray misses, focal distance = the cap (Camera.cs:89-95) */
    const float pos[3] = {0, 2, 15}, target[3] = {0, 3, 0};
    float forward[3], right[3], up[3], focal = 0;
    struct { int32_t frames, samples; float focal; int32_t pad; uint8_t camera[96]; } fs;   /* State.cs:29-41 */
    memset(&fs, 0, sizeof fs);
    HOST(pth_camera_basis(pos, target, forward, right, up));
    HOST(pth_camera_state(pos, forward, right, up, 1.0f / 0.0f, width, height, fs.camera, &focal));
    fs.frames = 1;
    fs.samples = 0;
    fs.focal = focal;
    GPU(pt_render(ctx, &fs, spp, bounces));

    float* image = (float*)malloc((size_t)width * height * 16);
    GPU(pt_read_image(ctx, image));
    uint64_t h = 1469598103934665603ull;
    const unsigned char* bytes = (const unsigned char*)image;
    for (size_t i = 0; i < (size_t)width * height * 16; i++) { h ^= bytes[i]; h *= 1099511628211ull; }
    const char* ext = strrchr(argv[6], '.');
    if (ext && strcmp(ext, ".png") == 0) HOST(pth_image_write(argv[6], image, width, height, PTH_IMAGE_PNG, PTH_TRANSFER_SRGB, 1.0f));
    else if (ext && strcmp(ext, ".pfm") == 0) HOST(pth_image_write(argv[6], image, width, height, PTH_IMAGE_PFM, PTH_TRANSFER_DISPLAY, 1.0f));
    else {
        FILE* f = fopen(argv[6], "wb");
        if (!f || fwrite(image, 16, (size_t)width * height, f) != (size_t)width * height) { fprintf(stderr, "cannot write %s\n", argv[6]); return 1; }
        fclose(f);
    }
    pt_stats st;
    st.struct_size = (uint32_t)sizeof(st);
    GPU(pt_get_stats(ctx, &st));
    printf("%dx%d %d spp on %d GPU context(s): %llu extend rays, %llu shadow rays, checksum %016llx\n", width, height, spp, devices, (unsigned long long)st.extend_rays,
           (unsigned long long)st.shadow_rays, (unsigned long long)h);
    free(image);
    pt_destroy(ctx);
    pth_scene_destroy(scene);
    return 0;
}
