// P/Invoke shim for the reference's C# host over libpt_hip.so (pt_hip.h): Wavefront.cs (body replaced; `Wavefront()`, `Allocate(int n)`, `Invoke()` keep their signatures).
// Written by tools/extract_csharp.py; not compiled here (no .NET toolchain).
class Wavefront : IDisposable
{
    public readonly IntPtr Context;
    private State.FrameState frame;
    private int allocated;

    public Wavefront()                                   // was: 5 SSBOs + 3 ShaderProgram compiles (Wavefront.cs:22-38)
    {
        var cfg = new Native.Config();
        Native.pt_default_config(ref cfg);
        cfg.Nee = 1;                                     // 0 = the reference's estimator, 1 = next-event estimation, 2 = with multiple importance sampling
        // cfg.DeviceCount = 8;                          // all GPUs of the node behind the same three calls
        Native.Check(IntPtr.Zero, Native.pt_create(ref cfg, out Context));
    }

    public void Allocate(int n)                          // Wavefront.cs:67-81: queues for n = width * height rays;
    {                                                    // the library sizes them in SetRenderState, where it learns the two factors
        this.allocated = n;
    }

    public void SetRenderState(ref State.RenderState rs) // replaces renderState.CopyToDevice() (Game.cs:147-150)
    {
        if (rs.Screen.Width * rs.Screen.Height != this.allocated) throw new InvalidOperationException("Allocate(width * height) first");
        Native.Check(Context, Native.pt_resize(Context, rs.Screen.Width, rs.Screen.Height));
        Native.Check(Context, Native.pt_set_render_state(Context, ref rs));
    }

    public void SetFrameState(ref State.FrameState fs)   // replaces frameState.CopyToDevice() (Game.cs:187-189)
    {
        this.frame = fs;
    }

    public void Invoke()                                 // was generate + <=10 x {extend, shade, readback} (Wavefront.cs:40-56)
    {
        // Samples == 0 in the frame state restarts the accumulation inside the library, exactly as the running mean of
        // shade.glsl:89-96 forgets the old value when samples is 0: Game.cs:140,157-165 need no extra call
        Native.Check(Context, Native.pt_render(Context, ref this.frame, 1, 10));
    }

    public void ReadImage(float[] rgba)                  // the rgba32f image the shaders used to write (Image.cs:7-19): width * height * 4
    {                                                    // floats, row 0 = bottom like gl_FragCoord; waits for the frame
        Native.Check(Context, Native.pt_read_image(Context, rgba));
    }

    public Native.Stats Stats()
    {
        var s = new Native.Stats { StructSize = (uint)Marshal.SizeOf<Native.Stats>() };
        Native.Check(Context, Native.pt_get_stats(Context, ref s));
        return s;
    }

    public void Dispose() => Native.pt_destroy(Context);
}
