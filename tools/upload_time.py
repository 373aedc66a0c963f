import sys, time
sys.path.insert(0, "/root/repo")
import pathtracer_amd as pa
from pathtracer_amd.host import Scene
s = Scene(); s.generate_instances(4026, 11); s.build(16)
m, t, n = s.arrays()
wf = pa.Wavefront()
for i in range(2):
    t0 = time.perf_counter(); wf.upload_scene(m, t, n); print("upload_scene %.0f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
