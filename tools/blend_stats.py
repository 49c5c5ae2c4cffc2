"""Blend-kernel work statistics from the instrumented build (libgs3d_hip_stats.so, -DGS_BLEND_STATS).

    GS3D_HIP_LIB=3dgs.cpp_amd/libgs3d_hip_stats.so python tools/blend_stats.py [N] [W] [H]
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
rec = pkg.synth.synth_records(n, seed=0, kind="S")
scene = pkg.Scene.from_records(rec)
rend = pkg.Renderer(scene)
u = pkg.camera_uniforms(pkg.make_camera(), w, h)
L = pkg.binding.lib()
out = (ctypes.c_ulonglong * 8)()
rend.render_host(u)
L.gs_debug_blend_stats(out, 1)
rend.render_host(u)
L.gs_debug_blend_stats(out, 1)
st = rend.stats()
names = ["wave_batches", "entries_any_quadrant", "pairs_evaluated", "lanes_alive", "pairs_reaching_exp",
         "lanes_in_exp", "entries_staged", "-"]
vals = dict(zip(names, [int(x) for x in out]))
print(f"N={st.num_gaussians} V={st.num_visible} D={st.num_instances} render={st.ms_render:.3f} ms")
for k, v in vals.items():
    print(f"  {k:24s} {v:14d}")
d = st.num_instances
print(f"  staged/D={vals['entries_staged']/d:.3f}  pairs/4D={vals['pairs_evaluated']/(4*d):.3f} "
      f"alive/pair={vals['lanes_alive']/max(vals['pairs_evaluated'],1):.1f} "
      f"exp_pairs/pairs={vals['pairs_reaching_exp']/max(vals['pairs_evaluated'],1):.3f} "
      f"exp_lanes/exp_pair={vals['lanes_in_exp']/max(vals['pairs_reaching_exp'],1):.1f}")
