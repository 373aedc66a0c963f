// P/Invoke shim for the reference's C# host over libpt_hip.so (pt_hip.h): Native.cs (new file in PathTracer/).
// Written by tools/extract_csharp.py; not compiled here (no .NET toolchain).
using System;
using System.Runtime.InteropServices;

namespace PathTracer
{
    static class Native
    {
        const string Lib = "pt_hip";   // libpt_hip.so next to the executable or on LD_LIBRARY_PATH

        [StructLayout(LayoutKind.Sequential)]
        public struct Config
        {
            public int StructSize, Device, Nee, RngMode, TileRank, TileWorld, SamplesPerBatch, TraversalStats, StageTiming;
            public int DeviceCount;   // GPUs behind this one context (Device .. Device + DeviceCount - 1); tiles and the RCCL gather are internal
            public int GroupFlags;
            [MarshalAs(UnmanagedType.ByValArray, SizeConst = 5)] public int[] Reserved;
        }

        // pt_stats: sequential, no padding beyond what the field order gives (8-byte fields sit on 8-byte offsets)
        [StructLayout(LayoutKind.Sequential)]
        public struct Stats
        {
            public uint StructSize, Reserved0;           // StructSize = Marshal.SizeOf<Stats>() before pt_get_stats; the library writes no more
            public ulong Paths, ExtendRays, ShadowRays, ExtendSteps, ExtendTris, ShadowSteps, ShadowTris, StackOverflows;
            public uint ExtendLaunches, ShadeLaunches, ConnectLaunches, GenerateLaunches;
            public float MsGenerate, MsExtend, MsShade, MsConnect, MsRender, MsExtendPrimary, Reserved1, Reserved2;
            public ulong SchedInterior, SchedLeaf, SchedRefill, SpecialRays, ExtendUniqueSteps, ExtendUniqueTris, ShadowOccluded, DeviceAllocations;
        }

        // pt_instance: a mesh placed by a System.Numerics.Matrix4x4 (row vectors, translation in M41..M43), 72 bytes
        [StructLayout(LayoutKind.Sequential)]
        public struct Instance
        {
            public System.Numerics.Matrix4x4 Transform;
            public int Mesh, FirstTriangle;
        }

        // PT_OPT_* keys of pt_set_option
        public const int OptNee = 1, OptRngMode = 2, OptTraversalStats = 3, OptStageTiming = 4, OptSamplesPerBatch = 5,
                         OptTraverseBlocksPerCu = 6, OptRefillPrimary = 7, OptRefillBounce = 8, OptQuantized = 9, OptInstanced = 10,
                         OptPipelineCalls = 11;

        [DllImport(Lib)] public static extern int pt_default_config(ref Config cfg);
        [DllImport(Lib)] public static extern int pt_create(ref Config cfg, out IntPtr ctx);
        [DllImport(Lib)] public static extern int pt_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr pt_last_error(IntPtr ctx);
        [DllImport(Lib)] public static extern int pt_upload_scene(IntPtr ctx, Material[] materials, int nMaterials,
                                                                  Triangle[] triangles, int nTriangles, Node[] nodes, int nNodes);
        [DllImport(Lib)] public static extern int pt_resize(IntPtr ctx, int width, int height);
        [DllImport(Lib)] public static extern int pt_set_render_state(IntPtr ctx, ref State.RenderState rs);
        [DllImport(Lib)] public static extern int pt_reset_accumulation(IntPtr ctx);
        [DllImport(Lib)] public static extern int pt_render(IntPtr ctx, ref State.FrameState fs, int spp, int maxBounces);
        [DllImport(Lib)] public static extern int pt_synchronize(IntPtr ctx);
        [DllImport(Lib)] public static extern int pt_read_image(IntPtr ctx, float[] rgba);
        [DllImport(Lib)] public static extern int pt_trace_closest(IntPtr ctx, float[] originDirection, int n, int[] tri, float[] t, float[] uv);
        [DllImport(Lib)] public static extern int pt_build_bvh(IntPtr ctx, Triangle[] trianglesIn, int nTriangles, Triangle[] trianglesOut,
                                                               Node[] nodesOut, int nodeCapacity, out int nNodes, out float buildMs);
        [DllImport(Lib)] public static extern int pt_build_scene(IntPtr ctx, Material[] materials, int nMaterials, Triangle[] trianglesIn, int nTriangles,
                                                                 Triangle[] trianglesOut, Node[] nodesOut, int nodeCapacity, out int nNodes, out float buildMs);
        [DllImport(Lib)] public static extern int pt_build_scene_instanced(IntPtr ctx, Material[] materials, int nMaterials, Triangle[] trianglesIn, int nTriangles,
                                                                           Triangle[] meshTriangles, int[] meshFirst, int[] meshCount, int nMeshes,
                                                                           Instance[] instances, int nInstances, out float buildMs);
        [DllImport(Lib)] public static extern int pt_trace_closest_noalloc(IntPtr ctx, float[] originDirection, int n, int[] tri, float[] t, float[] uv);
        [DllImport(Lib)] public static extern int pt_trace_occluded(IntPtr ctx, float[] originDirection, float[] tmax, int n, byte[] occluded);
        [DllImport(Lib)] public static extern int pt_set_option(IntPtr ctx, int key, int value);
        [DllImport(Lib)] public static extern int pt_get_stats(IntPtr ctx, ref Stats stats);       // set stats.StructSize first
        [DllImport(Lib)] public static extern int pt_reset_stats(IntPtr ctx);
        [DllImport(Lib)] public static extern int pt_generate_rays(IntPtr ctx, ref State.FrameState fs, int n, float[] originDirection, uint[] pixel);
        [DllImport(Lib)] public static extern int pt_bench_extend(IntPtr ctx, float[] originDirection, int n, int iters, out float avgMs, int[] tri, float[] t);
        [DllImport(Lib)] public static extern int pt_stream(IntPtr ctx, out IntPtr hipStream);
        // one process per GPU (TileRank / TileWorld): the rank-local accumulator for a gather the host runs itself, and its assembly
        [DllImport(Lib)] public static extern int pt_accum_device(IntPtr ctx, out IntPtr devicePtr, out ulong nSlotsPadded, out uint samples);
        [DllImport(Lib)] public static extern int pt_assemble_tiles(IntPtr ctx, IntPtr gatheredDevice, int world, uint samples, float[] rgba);

        public static void Check(IntPtr ctx, int status)
        {
            if (status != 0) throw new InvalidOperationException(Marshal.PtrToStringAnsi(pt_last_error(ctx)));
        }
    }
}
