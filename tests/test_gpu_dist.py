"""The native multi-GPU entry points (gs_dist_*, include/gs3d_hip.h) on the one GPU a test box has: a world of
one rank goes through the same code -- RCCL loaded on first use, ncclCommInitRank, the count + blob broadcasts,
adoption of the received scene -- and pose_shard_host, the C++ consumer that shards poses over ranks, renders every
pose of config D's kind bit-identically to the oracle.  World sizes > 1 need one GPU per rank (RCCL refuses two ranks
on one device); the pose split and the blob layout are covered at world size 2 by tests/test_dist_gloo.py on CPU.
"""
import os
import subprocess

import numpy as np
import pytest

from test_gpu_viewer import oracle_rgb8, read_ppm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3dgs.cpp_amd")


def test_native_broadcast_world_of_one(pkg, oracle, gpu):
    rec = pkg.synth.synth_records(5000, seed=51, kind="A")
    scene = pkg.Scene.from_records(rec, device=0)
    d = pkg.Dist(pkg.Dist.unique_id(), rank=0, world=1, device=0)
    assert (d.rank, d.world) == (0, 1) and d.pose_count(8) == 8
    got = d.broadcast_scene(scene, root=0)
    assert got is scene  # the root keeps its own handle
    rend = pkg.Renderer(got)
    w, h = 320, 200
    img, _ = rend.render_host(pkg.camera_uniforms(pkg.make_camera(), w, h))
    ref = oracle.stages(oracle.activate_records(rec), oracle.camera_uniforms(oracle.default_camera(), w, h))
    np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
    rend.close()
    d.close()
    scene.close()


def test_pose_count_is_round_robin(pkg, gpu):
    # pure bookkeeping of gs_dist_pose_count, checked against dist.poses_for_rank for the world sizes of config D
    d = pkg.Dist(pkg.Dist.unique_id(), rank=0, world=1, device=0)
    for poses in (0, 1, 7, 8, 9):
        assert d.pose_count(poses) == len(pkg.dist.poses_for_rank(poses, 0, 1))
    d.close()


def test_pose_shard_host_renders_every_pose(pkg, oracle, gpu, tmp_path):
    exe = os.path.join(PKG, "pose_shard_host")
    rec = pkg.synth.synth_records(6000, seed=52, kind="A")
    ply = str(tmp_path / "scene.ply")
    pkg.synth.write_ply(ply, rec)
    w, h, poses = 256, 160, 3
    out = subprocess.run([exe, ply, str(w), str(h), str(poses), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert f"{poses} of {poses} poses" in out.stdout
    verts = oracle.activate_records(rec)
    for k in range(poses):
        cam = oracle.default_camera(rotation=pkg.dist.pose_quaternion(k))
        got = read_ppm(tmp_path / f"pose_{k:03d}.ppm").astype(int)
        ref = oracle_rgb8(oracle, verts, cam, w, h).astype(int)
        if k == 0:
            np.testing.assert_array_equal(got, ref)
        else:  # the host computes sin/cos of the yaw in C++ double -> float; numpy may differ in the last ulp
            assert np.mean(np.abs(got - ref) > 1) < 0.01
