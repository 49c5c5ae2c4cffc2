"""world_size-2 test of the multi-GPU path on CPU (gloo): scene-blob broadcast + pose sharding.

The HIP kernels cannot run here, so each rank renders its poses with the oracle from the blob it
*received*; the parent compares them with a single-process rendering of the same poses."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, W, H, POSES = 600, 96, 64, 4


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    pkg, oracle = entry.load_package(), entry.load_oracle()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = torch.zeros(pkg.dist.blob_floats(N), dtype=torch.float32)
    if rank == 0:
        rec = pkg.synth.synth_records(N, seed=21, kind="A")
        blob.copy_(torch.from_numpy(pkg.dist.pack_blob(pkg.activate_records(rec))))
    pkg.dist.broadcast_blob(blob, src=0)
    verts = pkg.dist.unpack_blob(blob.numpy(), N).view(oracle.VERTEX_DT).reshape(-1)
    for k in pkg.dist.poses_for_rank(POSES, rank, world):
        u = oracle.camera_uniforms(oracle.default_camera(rotation=pkg.dist.pose_quaternion(k)), W, H)
        img, st = oracle.render_frame(verts, oracle.cov3d(verts), u)
        np.save(os.path.join(out_dir, f"pose{k}.npy"), img)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_and_pose_sharding(pkg, oracle, tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    rec = pkg.synth.synth_records(N, seed=21, kind="A")
    verts = oracle.activate_records(rec)
    imgs = []
    for k in range(POSES):
        u = oracle.camera_uniforms(oracle.default_camera(rotation=pkg.dist.pose_quaternion(k)), W, H)
        ref, _ = oracle.render_frame(verts, oracle.cov3d(verts), u)
        got = np.load(tmp_path / f"pose{k}.npy")
        np.testing.assert_array_equal(got, ref)
        imgs.append(ref)
    assert np.abs(imgs[0] - imgs[1]).max() > 1e-3  # poses really differ


def test_blob_roundtrip_and_sharding(pkg):
    v = pkg.activate_records(pkg.synth.synth_records(50, seed=1, kind="A"))
    np.testing.assert_array_equal(pkg.dist.unpack_blob(pkg.dist.pack_blob(v), 50), v)
    assert pkg.dist.poses_for_rank(8, 3, 8) == [3]
    assert pkg.dist.poses_for_rank(8, 1, 2) == [1, 3, 5, 7]
    assert sorted(sum((pkg.dist.poses_for_rank(7, r, 3) for r in range(3)), [])) == list(range(7))
    assert pkg.dist.pose_quaternion(0) == (1.0, 0.0, 0.0, 0.0)
