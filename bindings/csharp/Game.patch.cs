// P/Invoke shim for the reference's C# host over libpt_hip.so (pt_hip.h): Game.cs (`Initialize`, `Resize`, `Update`, `Draw` — everything that touched the GPU; `Game.cs:36-55,138-195`).
// Written by tools/extract_csharp.py; not compiled here (no .NET toolchain).
private float[] pixels;                                  // host copy of the accumulated image, uploaded to `screen` every frame

public void Initialize()
{
    this.quad = new ShaderProgram(
        ShaderArgument.Load(ShaderType.VertexShader, "Shaders/Quad/vertex.glsl"),
        ShaderArgument.Load(ShaderType.FragmentShader, "Shaders/Quad/fragment.glsl"));

    this.screen = new Image(1, 1);
    this.pixels = new float[4];

    this.scene = new Scene();
    this.camera = new Camera(this.window, this.scene);
    this.camera.Set(new Vector3(0, 2, 15), Vector3.UnitY * 3);

    this.renderState = new Uniform<State.RenderState>(0);   // kept as plain holders of the two structs; nothing reads the UBOs
    this.frameState = new Uniform<State.FrameState>(1);

    this.wavefront = new Wavefront();                    // before the upload: the scene goes to its context
    this.InitializeScene(11);
    this.scene.CopyToDevice(this.wavefront);
}

public void Resize(int width, int height)
{
    this.frameState.Data.Samples = 0;                    // a frame state with Samples == 0 restarts the accumulation in the library

    this.screen.Dispose();
    this.screen = new Image(width, height);              // still bound as image unit 0: Quad/fragment.glsl:7-11 reads it unchanged
    this.pixels = new float[width * height * 4];

    this.wavefront.Allocate(width * height);

    this.renderState.Data.Screen.Width = width;
    this.renderState.Data.Screen.Height = height;
    this.renderState.Data.SkyColor = SkyColor;
    this.wavefront.SetRenderState(ref this.renderState.Data);   // was renderState.CopyToDevice()
}

public void Update(float dt, KeyboardState keystate)
{
    bool changed = this.camera.HandleInput(keystate, dt);

    if (changed)
    {
        this.frameState.Data.Samples = 0;
    }

    if (keystate.IsKeyDown(Key.R))
    {
        this.frameState.Data.Samples = 0;
        this.window.ClientSize = new OpenTK.Size(512, 512);
        this.camera.Set(new Vector3(0, 2, 15), Vector3.UnitY * 3);
    }

    if (keystate.IsKeyDown(Key.P))
    {
        this.scene.Clear();

        int seed = random.Next();
        this.InitializeScene(seed);

        this.scene.CopyToDevice(this.wavefront);         // Accelerator.Build + pt_upload_scene
        this.frameState.Data.Samples = 0;

        Console.WriteLine($"Regenerated scene using seed: {seed}");
    }

    this.camera.SetUniform(this.frameState);
}

public void Draw(float dt)
{
    this.wavefront.SetFrameState(ref this.frameState.Data);   // was frameState.CopyToDevice(): the frame sees the counters before the increment
    this.frameState.Data.Frames++;
    this.frameState.Data.Samples++;

    this.wavefront.Invoke();                             // one sample per pixel, up to 10 segments per path
    this.wavefront.ReadImage(this.pixels);               // sum / samples: what the running mean of shade.glsl:89-96 holds
    this.screen.SetData(this.pixels);                    // the image the compute shaders used to write in place

    this.quad.Use();
    GL.DrawArrays(PrimitiveType.TriangleStrip, 0, 4);
}

public void Dispose()
{
    this.quad.Dispose();
    this.screen.Dispose();
    this.wavefront.Dispose();
}
