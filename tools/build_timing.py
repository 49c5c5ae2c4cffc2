"""Debug (GPU box): phase durations inside k_bin_build from the instrumented library (libgs3d_hip_tm.so)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GS3D_HIP_LIB"] = os.path.join(ROOT, "3dgs.cpp_amd", "libgs3d_hip_tm.so")
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
n, w, h = int(os.environ.get("N", 1_000_000)), int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
rec = pkg.synth.synth_records(n, seed=0, kind="S")
scene = pkg.Scene.from_records(rec, device=0)
rend = pkg.Renderer(scene)
u = pkg.camera_uniforms(pkg.make_camera(), w, h)
for _ in range(5):
    rend.render_host(u, want_rgba=True)
st = rend.stats()
t = np.zeros((1024, 10), np.uint64)
L = pkg.binding.lib()
assert L.gs_debug_build_timing(t.ctypes.data_as(C.c_void_p)) == 0
t = t[t[:, 7] > 0].astype(np.float64)
extra = t[:, 8:10]
t = t[:, :8]
d = np.diff(t, axis=1) / 100.0  # 100 MHz clock -> us
names = ["bin offset", "load ids+depth", "sort + ties", "boxes + counts", "chunk prefix", "ranges + alloc", "fill"]
print(f"bins {len(t)} max_bin {st.max_bin_entries} E1 {st.num_bin_entries} D {st.num_instances}; kernel span {(t[:,7].max()-t[:,0].min())/100:.1f} us; per-bin total mean {(t[:,7]-t[:,0]).mean()/100:.1f} max {(t[:,7]-t[:,0]).max()/100:.1f} us")
for k in range(7):
    print(f"  {names[k]:16s} mean {d[:,k].mean():7.2f} us   max {d[:,k].max():7.2f}")
print(f"  inside sort + ties: key sort {(extra[:,0]-t[:,2]).mean()/100:.2f} us, tie scan + id gather + ids in place {(extra[:,1]-extra[:,0]).mean()/100:.2f} us, tie fix-up {(t[:,3]-extra[:,1]).mean()/100:.2f} us")
print("  start spread (first to last bin start)", (t[:, 0].max() - t[:, 0].min()) / 100.0, "us")
s0 = (t[:, 0] - t[:, 0].min()) / 100.0
tot = (t[:, 7] - t[:, 0]) / 100.0
print("  start offsets us: quantiles", [round(float(np.quantile(s0, q)), 1) for q in (0, .1, .25, .5, .6, .75, .9, 1)])
late = s0 > 5
print(f"  bins starting later than 5 us: {int(late.sum())} of {len(s0)}; mean total early {tot[~late].mean():.1f} us, late {tot[late].mean() if late.any() else 0:.1f} us")
