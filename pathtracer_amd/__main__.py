"""Headless stand-in for running the reference's window: Game.Initialize + a number of Game.Draw calls, image to a file.

  python -m pathtracer_amd --out bundled.png                       # InitializeScene(11) at 1280x720, camera of Game.cs:45
  python -m pathtracer_amd --scene cornell --nee 2 --spp 256 --out cornell.png
  python -m pathtracer_amd --load scene.ptscene --gpu-build --out image.pfm

The scene comes from the bundled recipe (Game.cs:66-136), a bundled mesh, a random triangle soup, or a cache file written
by Scene.save; the tree is built by the host mirror of Accelerator.Build or on the GPU.
"""
import argparse
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m pathtracer_amd", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scene", default="bundled", help="bundled | cornell | a mesh name of the asset blob | soup:<triangles>")
    ap.add_argument("--seed", type=int, default=11, help="Game.InitializeScene seed of the bundled recipe")
    ap.add_argument("--load", help="a scene cache file written by Scene.save (overrides --scene)")
    ap.add_argument("--width", type=int, default=1280)      # Game.cs:31: the reference's window
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel = number of Game.Draw frames accumulated")
    ap.add_argument("--bounces", type=int, default=10, help="path segments per sample (Wavefront.cs:44)")
    ap.add_argument("--nee", type=int, default=1, choices=(0, 1, 2), help="0 the reference's estimator, 1 next-event estimation, 2 with multiple importance sampling")
    ap.add_argument("--position", type=float, nargs=3, help="camera position (default: Game.cs:45, or a view of the mesh)")
    ap.add_argument("--target", type=float, nargs=3)
    ap.add_argument("--sky", type=float, nargs=3, help="sky colour (default: Game.cs:13)")
    ap.add_argument("--gpu-build", action="store_true", help="build the tree on the GPU (pt_build_scene) instead of the host mirror")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", required=True, help="image file: .png / .ppm (8 bit) or .pfm (float)")
    ap.add_argument("--transfer", default="srgb", choices=("srgb", "display"), help="8-bit formats: sRGB curve, or what the reference's window shows (no gamma)")
    ap.add_argument("--exposure", type=float, default=1.0)
    args = ap.parse_args(argv)

    from . import Wavefront
    from .host import Camera, Scene, write_image

    t0 = time.perf_counter()
    scene = Scene()
    position, target = (0, 2, 15), (0, 3, 0)        # Game.cs:45
    autofocus = False                                # the reference sets the camera before the scene exists: focal distance 20
    if args.load:
        scene.load(args.load)
    elif args.scene == "bundled":
        scene.initialize_scene(args.seed)
    elif args.scene.startswith("soup:"):
        scene = Scene("")
        scene.generate_soup(int(args.scene[5:]))
        position, target = (0, 0, 120), (0, 0, 0)
    else:
        scene.add_mesh(args.scene)
        position, target, autofocus = (0, 1, 3.4), (0, 1, 0), True
    if args.position:
        position = tuple(args.position)
    if args.target:
        target = tuple(args.target)

    wf = Wavefront(device=args.device, nee=args.nee, rng_mode=1)
    if args.gpu_build:
        mats, tris = scene.arrays()[:2] if args.load else scene.references()   # a cache file's order is an insertion order too
        build_ms = wf.build_scene(mats, tris)
        how = f"GPU build {build_ms:.1f} ms"
        n_triangles = int(tris.size)
    else:
        if not args.load:
            scene.build()
        wf.upload_scene(*scene.arrays())
        how = "host build"
        n_triangles = scene.counts()["triangles"]
    t1 = time.perf_counter()

    sky = tuple(args.sky) + (1.0,) if args.sky else None
    wf.allocate(args.width, args.height, sky=sky)
    fs = Camera(args.width, args.height, wf if autofocus else None).frame_state(position, target, frames=1, samples=1)
    wf.invoke(fs, spp=args.spp, max_bounces=args.bounces)
    image = wf.read_image()
    st = wf.stats()
    t2 = time.perf_counter()
    wf.close()
    write_image(args.out, image, transfer=args.transfer, exposure=args.exposure)
    rays = st["extend_rays"] + st["shadow_rays"]
    print(f"{n_triangles} triangles ({how}, scene ready in {t1 - t0:.2f} s); {args.width}x{args.height} x {args.spp} spp: "
          f"{rays / 1e6:.1f} Mrays in {t2 - t1:.3f} s; wrote {args.out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
