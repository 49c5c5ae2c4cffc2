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

import helpers

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


def test_receiving_branch_and_storage_flags_at_world_one(pkg, oracle, gpu):
    """gs_dist_broadcast_scene_ex(GS_DIST_COPY_ON_ROOT): the root receives into a NEW scene through the code every
    non-root rank runs (allocate, out-of-place ncclBroadcast, cov3D recomputed locally, storage flags applied) -- on the one
    GPU of a test box.  The received replica must render the root's frame bit for bit, with fp32 SH and, when the root was
    quantised first, with binary16 SH (the header carries the flag: every replica renders from the same coefficients)."""
    rec = pkg.synth.synth_records(7000, seed=53, kind="A")
    w, h = 320, 200
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    for quantised in (False, True):
        scene = pkg.Scene.from_records(rec, device=0)
        if quantised:
            scene.quantize_sh()
        d = pkg.Dist(pkg.Dist.unique_id(), rank=0, world=1, device=0)
        replica = d.broadcast_scene(scene, root=0, copy_on_root=True)
        assert replica is not scene and replica.num_vertices == scene.num_vertices
        assert replica.sh_bits == (16 if quantised else 32)
        np.testing.assert_array_equal(replica.download_vertices().view(np.uint32), scene.download_vertices().view(np.uint32))
        np.testing.assert_array_equal(replica.download_cov3d().view(np.uint32), scene.download_cov3d().view(np.uint32))
        r0, r1 = pkg.Renderer(scene), pkg.Renderer(replica)
        a, _ = r0.render_host(u)
        b, _ = r1.render_host(u)
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
        verts = oracle.activate_records(rec)
        if quantised:
            verts["sh"] = verts["sh"].astype(np.float16).astype(np.float32)
        ref = oracle.stages(verts, oracle.camera_uniforms(oracle.default_camera(), w, h))
        np.testing.assert_array_equal(b.view(np.uint32), ref["image"].view(np.uint32))
        for x in (r0, r1, d, replica, scene):
            x.close()


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
    # gs_dist_verify's evidence, printed by the native host too (one rank here: RCCL refuses two ranks on one device): the
    # all-reduce saw one rank, the replica's checksum equals itself, the broadcast was timed and sized
    import json
    rep = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{"rccl"')][0])["rccl"]
    assert rep["ranks"] == 1 and rep["world_size"] == 1 and rep["blob_checksums_equal"] is True and rep["version"] > 20000
    assert rep["blob_MB"] == round(pkg.dist.blob_floats(6000) * 4 / 1e6, 1) and rep["broadcast_ms"] > 0
    verts = oracle.activate_records(rec)
    for k in range(poses):
        cam = oracle.default_camera(rotation=pkg.dist.pose_quaternion(k))
        got = read_ppm(tmp_path / f"pose_{k:03d}.ppm").astype(int)
        ref = oracle_rgb8(oracle, verts, cam, w, h).astype(int)
        # both hosts evaluate the yaw with the same double operations (libm cos / sin of 5k * (pi / 180) / 2): same float
        # quaternion, same uniforms, same frame
        np.testing.assert_array_equal(got, ref)


def test_bench_two_ranks_on_one_gpu(pkg, oracle, gpu, tmp_path):
    """bench.py's N > 1 path as the driver launches it (torch.distributed.run, one process per rank, RANK / WORLD_SIZE from
    the environment), rehearsed on the one GPU of a test box with the gloo backend (RCCL refuses two ranks on one device):
    rank 0 builds the scene, the blob is broadcast, each rank adopts its copy and renders ITS pose.  Both ranks' frames must
    equal the checker's for poses 0 and 1, and the line must report two GPUs' worth of frames."""
    import json
    import sys
    n, w, h = 50_000, 640, 360
    env = dict(os.environ, GS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GS_SORT_PATH", None)
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--gaussians", str(n), "--width", str(w), "--height", str(h), "--dump-frames", str(tmp_path), "--exact"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["gaussians"] == n
    helpers.assert_value_is_frames_over_time(line)  # whole-job frames: both ranks' K steps
    assert "cpu_baseline" not in line and "other_configs" not in line  # rank 0 at N = 1 only
    # what the collective saw (bench.py: rccl_evidence): every rank took part, every rank holds rank 0's blob, and the line
    # carries each rank's own rate (the driver's 8-GPU run prints the same block over RCCL)
    r = line["rccl"]
    assert r["ranks"] == 2 and r["world_size"] == 2 and r["backend"].startswith("gloo") and r["blob_checksums_equal_rank0"] is True
    assert r["blob_MB"] == round(pkg.dist.blob_floats(n) * 4 / 1e6, 1) and r["broadcast_ms"] > 0
    assert len(r["per_rank_frames_per_s"]) == 2 and all(x > 0 for x in r["per_rank_frames_per_s"]) and r["slowest_rank"] in (0, 1)
    assert line["value"] <= sum(r["per_rank_frames_per_s"]) * 1.001  # the job's rate is bounded by its slowest rank
    verts = oracle.activate_records(pkg.synth.synth_records(n, seed=0, kind="S"))
    for rank in (0, 1):
        cam = oracle.default_camera(rotation=pkg.dist.pose_quaternion(rank))
        ref = oracle.stages(verts, oracle.camera_uniforms(cam, w, h))
        got = np.load(tmp_path / f"frame_rank{rank}.npy")
        np.testing.assert_array_equal(got.view(np.uint32), ref["image"].view(np.uint32))
