// P/Invoke shim for the reference's C# host over libpt_hip.so (pt_hip.h): Wavefront.cs (body replaced; public surface unchanged: `Wavefront()`, `Allocate(n)`, `Invoke()`).
// Written by tools/extract_csharp.py; not compiled here (no .NET toolchain).
class Wavefront : IDisposable
{
    public readonly IntPtr Context;
    private int width, height;

    public Wavefront()                                   // was: 5 SSBOs + 3 ShaderProgram compiles (Wavefront.cs:22-38)
    {
        var cfg = new Native.Config();
        Native.pt_default_config(ref cfg);
        cfg.Nee = 1;                                     // 0 = the reference's estimator, 1 = next-event estimation, 2 = with multiple importance sampling
        Native.Check(IntPtr.Zero, Native.pt_create(ref cfg, out Context));
    }

    public void Allocate(int width, int height, ref State.RenderState rs)   // was Allocate(n) (Wavefront.cs:67-81)
    {
        this.width = width; this.height = height;
        Native.Check(Context, Native.pt_resize(Context, width, height));
        Native.Check(Context, Native.pt_set_render_state(Context, ref rs));
    }

    public void Invoke(ref State.FrameState fs)          // was generate + <=10 x {extend, shade, readback} (Wavefront.cs:40-56)
    {
        Native.Check(Context, Native.pt_render(Context, ref fs, 1, 10));
    }

    public void Dispose() => Native.pt_destroy(Context);
}
