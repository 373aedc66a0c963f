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
        [DllImport(Lib)] public static extern int pt_trace_closest_noalloc(IntPtr ctx, float[] originDirection, int n, int[] tri, float[] t, float[] uv);
        [DllImport(Lib)] public static extern int pt_trace_occluded(IntPtr ctx, float[] originDirection, float[] tmax, int n, byte[] occluded);
        [DllImport(Lib)] public static extern int pt_set_option(IntPtr ctx, int key, int value);

        public static void Check(IntPtr ctx, int status)
        {
            if (status != 0) throw new InvalidOperationException(Marshal.PtrToStringAnsi(pt_last_error(ctx)));
        }
    }
}
