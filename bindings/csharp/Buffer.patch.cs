// P/Invoke shim for the reference's C# host over libpt_hip.so (pt_hip.h): Buffer.cs (the three members that change; `Library/Graphics/Buffer.cs`).
// Written by tools/extract_csharp.py; not compiled here (no .NET toolchain).
public T[] Data => this.data;                            // the backing array (its first Count elements are live): what P/Invoke pins

public void Assign(T[] items, int count)                 // adopt arrays a native call filled (pt_build_bvh)
{
    this.data = items;
    this.count = count;
}

public void CopyToDevice()                               // was GL.BufferSubData / GL.BufferData (Buffer.cs:94-126)
{
    if (this.binding >= 10 && this.binding <= 12)
    {
        return;                                          // scene arrays: uploaded by Scene.CopyToDevice -> pt_upload_scene
    }

    throw new NotSupportedException($"Buffer<{typeof(T).Name}> (binding={this.binding}) has no GPU side any more");
}
