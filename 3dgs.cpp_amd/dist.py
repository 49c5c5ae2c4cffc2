"""Multi-GPU plumbing: replicate the scene, shard camera poses.

The path shards by frame: the scene is read-only, every pose is an independent unit, so there is no
per-frame collective.  The only exchange is one broadcast of the packed SoA scene blob (59 floats per
Gaussian) from rank 0 at load time -- torch.distributed backend "nccl" (RCCL over xGMI) on GPUs, "gloo"
in the CPU tests.  Pure layout / bookkeeping; no arithmetic of the hot path lives here.
"""
import math

import numpy as np

BLOB_PLANES = 59  # floats per Gaussian: 11 SoA planes (pos 3, scale 3, rot 4, opacity 1) + AoS SH block of 48


def blob_stride(n):
    """Plane stride in floats: n rounded up to 16, so that every plane and every SH block starts on a 64-byte line."""
    return (n + 15) & ~15


def blob_floats(n):
    """gs_scene_blob_floats(n): 11 padded planes + the n x 48 SH block."""
    return 11 * blob_stride(n) + 48 * n


def pack_blob(vertices):
    """(n, 60) activated GSScene::Vertex rows -> packed SoA blob (blob_floats(n) floats), plane-major."""
    v = np.ascontiguousarray(vertices).view(np.float32).reshape(-1, 60)
    n = len(v)
    st = blob_stride(n)
    blob = np.zeros(blob_floats(n), np.float32)
    planes = blob[:11 * st].reshape(11, st)[:, :n]
    planes[0:3] = v[:, 0:3].T      # position xyz (w == 1 is implicit)
    planes[3:6] = v[:, 4:7].T      # exp(scale)
    planes[6:10] = v[:, 8:12].T    # rotation w x y z
    planes[10] = v[:, 7]           # sigmoid(opacity)
    blob[11 * st:] = v[:, 12:60].reshape(-1)  # SH block, AoS: 16 RGB triples per Gaussian, contiguous
    return blob


def unpack_blob(blob, n):
    """Inverse of pack_blob -> (n, 60) float32."""
    blob = np.asarray(blob, np.float32).reshape(-1)
    st = blob_stride(n)
    b = blob[:11 * st].reshape(11, st)[:, :n]
    v = np.zeros((n, 60), np.float32)
    v[:, 0:3] = b[0:3].T
    v[:, 3] = 1.0
    v[:, 4:7] = b[3:6].T
    v[:, 7] = b[10]
    v[:, 8:12] = b[6:10].T
    v[:, 12:60] = blob[11 * st:].reshape(n, 48)
    return v


def broadcast_blob(blob_tensor, src=0):
    """One collective per scene: rank `src` -> everyone.  No-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob_tensor, src=src)
    return blob_tensor


def pose_quaternion(k, step_deg=5.0):
    """Pose k of BASELINE config D: the default camera yawed by k * 5 degrees about world y (w, x, y, z)."""
    a = math.radians(step_deg * k) / 2.0
    return (math.cos(a), 0.0, math.sin(a), 0.0)


def poses_for_rank(num_poses, rank, world):
    """Pose i goes to rank i mod world."""
    return [i for i in range(num_poses) if i % world == rank]
