import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import pathtracer_amd as pa
from pathtracer_amd.host import Scene, Camera
for name in ("soup", "bundled"):
    s = Scene("") if name == "soup" else Scene()
    if name == "soup": s.generate_soup(1000000); view=((0,0,120),(0,0,0))
    else: s.initialize_scene(11); view=((0,2,15),(0,3,0))
    s.build(8)
    wf = pa.Wavefront(nee=True, rng_mode=1, traversal_stats=True)
    wf.upload_scene(*s.arrays())
    w,h=960,540
    wf.allocate(w,h)
    fs=Camera(w,h).frame_state(view[0],view[1],frames=1)
    wf.invoke(fs, spp=4, max_bounces=4)
    st=wf.stats()
    print(name, "shadow rays", st["shadow_rays"], "occluded", st["shadow_occluded"], "frac %.3f"%(st["shadow_occluded"]/st["shadow_rays"]), "steps/ray %.1f tris/ray %.1f"%(st["shadow_steps"]/st["shadow_rays"], st["shadow_tris"]/st["shadow_rays"]), "extend steps/ray %.1f"%(st["extend_steps"]/st["extend_rays"]))
    wf.close()
