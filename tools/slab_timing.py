"""Debug (GPU box): phase durations inside k_bin_slabs / k_slab_work (depth-order level 4) at T(6e6), 1920x1080, from the
instrumented library:  make -C 3dgs.cpp_amd/csrc variant TAG=tm DEFS=-DGS_BUILD_TIMING ; gpurun -- python tools/slab_timing.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GS3D_HIP_LIB"] = os.path.join(ROOT, "3dgs.cpp_amd", "libgs3d_hip_tm.so")
import __graft_entry__ as entry
pkg = entry.load_package()
rec = pkg.synth.synth_records(6_000_000, seed=0, kind="T")
scene = pkg.Scene.from_records(rec, device=0)
rend = pkg.Renderer(scene)
u = pkg.camera_uniforms(pkg.make_camera(), 1920, 1080)
for _ in range(3):
    rend.render_host(u, want_rgba=True)
for _ in range(3):
    rend.render_host(u, want_rgba=True)
st = rend.stats()
print("level", st.sort_level, "max_bin", st.max_bin_entries, "E1", st.num_bin_entries, "D", st.num_instances, "ms_sort", st.ms_sort)
t = np.zeros((1024, 10), np.uint64)
L = pkg.binding.lib()
assert L.gs_debug_build_timing(t.ctypes.data_as(C.c_void_p)) == 0
t = t.astype(np.float64) / 100.0
a = t[:512]; a = a[a[:, 7] > 0]
multi = a[:, 2] < a[:, 0]  # (stamps of an earlier frame at another level: this frame only planned the bin)
print(f"k_bin_slabs: {len(a)} bins, {int(multi.sum())} planned (multi), span {a[:,7].max()-a[:,0].min():.1f} us; sum of WG times {(a[:,7]-a[:,0]).sum():.0f} us = {(a[:,7]-a[:,0]).sum()/256/(a[:,7].max()-a[:,0].min()):.2f} of 256 CUs x span")
s0 = a[:, 0] - a[:, 0].min()
print("  start offsets quantiles", [round(float(np.quantile(s0, q)), 1) for q in (0, .25, .5, .75, .9, 1)])
pm = a[multi]
print(f"  planned bins: offset scan {(pm[:,1]-pm[:,0]).mean():.1f}, plan (3 passes + descriptors) mean {(pm[:,7]-pm[:,1]).mean():.1f} max {(pm[:,7]-pm[:,1]).max():.1f} us")
ps = a[~multi]
order = [0, 1, 2, 8, 9, 3, 4, 5, 6, 7]
names = ["offset scan", "load", "key sort", "tie scan+id gather", "tie fix", "boxes+counts", "chunk prefix", "ranges", "fill"]
tot = ps[:, 7] - ps[:, 0]
print(f"  whole bins (<= 12288): total mean {tot.mean():.1f} max {tot.max():.1f} us")
for k in range(9):
    d = ps[:, order[k + 1]] - ps[:, order[k]]
    print(f"    {names[k]:20s} mean {d.mean():7.2f} max {d.max():7.2f}")
b = t[512:]; b = b[b[:, 7] > 0]
tot = b[:, 7] - b[:, 1]
print(f"k_slab_work: {len(b)} slabs, span {b[:,7].max()-b[:,1].min():.1f} us, per slab mean {tot.mean():.1f} max {tot.max():.1f}; sum {tot.sum():.0f} us = {tot.sum()/256/(b[:,7].max()-b[:,1].min()):.2f} of 256 CUs x span")
names2 = ["compaction pass", "key sort", "tie scan+id gather", "tie fix", "boxes+counts", "chunk prefix", "ranges", "fill"]
order2 = [1, 2, 8, 9, 3, 4, 5, 6, 7]
for k in range(8):
    d = b[:, order2[k + 1]] - b[:, order2[k]]
    print(f"    {names2[k]:20s} mean {d.mean():7.2f} max {d.max():7.2f}")
s1 = b[:, 1] - b[:, 1].min()
print("  slab start quantiles", [round(float(np.quantile(s1, q)), 1) for q in (0, .25, .5, .75, .9, 1)])
