// P/Invoke shim for the reference's C# host over libpt_hip.so (pt_hip.h): Texture.cs (one method added; `Library/Graphics/Texture.cs`).
// Written by tools/extract_csharp.py; not compiled here (no .NET toolchain).
public void SetData(float[] rgba)                        // beside SetData(uint[]) (Texture.cs:39-52): rgba32f rows, bottom row first
{
    this.Bind();
    GL.TexSubImage2D(
        TextureTarget.Texture2D,
        0,
        0,
        0,
        this.Width,
        this.Height,
        PixelFormat.Rgba,
        PixelType.Float,
        rgba);
}
