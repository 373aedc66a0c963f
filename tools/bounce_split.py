"""Roofline fractions of the extend launches of camera rays (b = 0) and of bounce rays (b >= 1) apart: an instrumented and a timed
run with 1 bounce, the same with 4, and the difference.  python tools/bounce_split.py [refill-bounce]"""
import sys, json, numpy as np
sys.path.insert(0, '.')
import bench, argparse
import pathtracer_amd as pa
args = argparse.Namespace(width=0,height=0,spp=0,bounces=0,batch=0,blocks=-1,refill_primary=0,refill_bounce=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cost = bench.isa_cost()["k_extend"]
for wl in ("bundled", "c3"):
    res = {}
    for nb in (1, 4):
        a = argparse.Namespace(**vars(args)); a.bounces = nb
        run = bench.Run(a, wl, 0, 1, 0)
        counted, staged = run.instrumented()
        run.wf.set_option(4, 1)
        dt, st = bench.timed_steps(run, 3, 1, run.wf.synchronize)
        res[nb] = (counted, st)
        run.close()
    c1, s1 = res[1]; c4, s4 = res[4]
    def diff(a, b): return {k: (a[k] - b[k]) if isinstance(a[k], (int, float)) else a[k] for k in a}
    cb, sb = diff(c4, c1), diff(s4, s1)
    for name, c, s in (("camera", c1, s1), ("bounce", cb, sb)):
        rays = c["extend_rays"]; ms = s["ms_extend"]; trays = s["extend_rays"]
        loads = 7 * c["extend_unique_steps"] / rays + 3 * c["extend_unique_tris"] / rays + 2
        vm = 16 * loads * trays / (ms * 1e-3) / 1e9 / bench.VMEM_PEAK_GBS
        lane = (c["extend_steps"] * cost["interior_iteration"]["valu"] + c["extend_tris"] * cost["leaf_phase"]["valu"])
        va = lane / rays * trays / (ms * 1e-3) / 1e12 / bench.VALU_PEAK_TLANEOPS
        issued = (c["sched_interior"] * cost["interior_iteration"]["valu"] + c["sched_leaf"] * cost["leaf_phase"]["valu"]) * 64
        print(f"{wl} {name}: {trays / ms / 1e3:.0f} Mrays/s, steps/ray {c['extend_steps'] / rays:.2f} distinct {c['extend_unique_steps'] / rays:.2f}, tris/ray {c['extend_tris'] / rays:.2f} distinct {c['extend_unique_tris'] / rays:.2f}, "
              f"vmem frac {vm:.3f}, valu useful frac {va:.3f}, lane use {lane / issued:.2f}, model busy {va / (lane / issued):.2f}; per 64 rays: interior iterations {64 * c['sched_interior'] / rays:.1f} (lane use {c['extend_steps'] / (64 * c['sched_interior']):.2f}), "
              f"leaf phases {64 * c['sched_leaf'] / rays:.1f} (lane use {c['extend_tris'] / (64 * c['sched_leaf']):.2f}), refills {64 * c['sched_refill'] / rays:.1f}")
