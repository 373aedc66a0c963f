// P/Invoke shim for the reference's C# host over libpt_hip.so (pt_hip.h): Scene.cs (one method changes).
// Written by tools/extract_csharp.py; not compiled here (no .NET toolchain).
public void CopyToDevice(Wavefront wavefront)           // was three Buffer<T>.CopyToDevice calls (Scene.cs:176-183)
{
    this.accelerator.Build();                            // unchanged: the C# builder still produces Node[] / Triangle[]
    Native.Check(wavefront.Context, Native.pt_upload_scene(wavefront.Context,
        this.materials.Data, this.materials.Count, this.triangles.Data, this.triangles.Count, this.nodes.Data, this.nodes.Count));
}

public void Build(IntPtr ctx)                            // was the recursive Subdivide on one CPU thread (Accelerator.cs:113-310)
{
    var input = this.primitives.Select(p => p.Triangle).ToArray();          // Accelerator.AddTriangle order
    var tris = new Triangle[input.Length];
    var nodes = new Node[Math.Max(1, 2 * input.Length - 1)];
    Native.Check(ctx, Native.pt_build_bvh(ctx, input, input.Length, tris, nodes, nodes.Length, out int n, out float ms));
    this.triangles.Assign(tris, input.Length);           // the Buffer<T> mirrors Accelerator.Intersect reads
    this.nodes.Assign(nodes, n);
}
