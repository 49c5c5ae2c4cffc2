"""Independent numpy (float64) restatement of the splat pipeline in conventional math form.

Purpose: catch transcription errors in oracle/gs_oracle.c (a transposed matrix, a wrong SH sign, a
swapped conic entry).  It is written from the GLSL shaders via the "conventional math" reading in
SURVEY.md Appendix A -- standard rotation matrix, Sigma = R S^2 R^T, cov2d = (J V) Sigma (J V)^T --
NOT by following the oracle's column-major emulation, and it evaluates in float64 with numpy's own
exp.  So it agrees with the oracle to ~1e-6 relative, not bit for bit; discrete decisions may flip for
the rare value sitting on a threshold, which the tests allow for explicitly.
"""
import numpy as np

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def activate(records):
    """GSScene.cpp:36-59 -> dict of float64 arrays."""
    r = records.astype(np.float64)
    sh_planar = r[:, 6:54]
    sh = np.zeros((len(r), 16, 3))
    sh[:, 0, :] = sh_planar[:, 0:3]
    for c in range(3):
        sh[:, 1:, c] = sh_planar[:, 3 + 15 * c: 3 + 15 * (c + 1)]
    rot = r[:, 58:62]
    return dict(pos=r[:, 0:3], scale=np.exp(r[:, 55:58]), opacity=1.0 / (1.0 + np.exp(-r[:, 54])),
                rot=rot / np.linalg.norm(rot, axis=1, keepdims=True), sh=sh)


def rotation_std(q):
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - z * w)
    R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w)
    R[:, 2, 1] = 2 * (y * z + x * w)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def cov3d(scene):
    R = rotation_std(scene["rot"])
    S2 = scene["scale"] ** 2
    Sigma = np.einsum("nij,nj,nkj->nik", R, S2, R)  # R S^2 R^T
    return Sigma


def camera(position, quat, fov_deg, near, far, width, height):
    """Renderer.cpp:719-754 in conventional form: world->camera, then the row flips."""
    w, x, y, z = quat
    Rc = rotation_std(np.array([[w, x, y, z]], dtype=np.float64))[0]
    M = np.eye(4)
    M[:3, :3] = Rc
    M[:3, 3] = position
    V0 = np.linalg.inv(M)
    tan_fovx = np.tan(np.radians(fov_deg) / 2.0)
    tan_fovy = tan_fovx * height / width
    P = np.zeros((4, 4))
    P[0, 0] = 1.0 / ((width / height) * tan_fovy)
    P[1, 1] = 1.0 / tan_fovy
    P[2, 2] = -(far + near) / (far - near)
    P[2, 3] = -(2.0 * far * near) / (far - near)
    P[3, 2] = -1.0
    proj = np.diag([1.0, -1.0, 1.0, 1.0]) @ P @ V0
    view = np.diag([1.0, -1.0, -1.0, 1.0]) @ V0
    return dict(proj=proj, view=view, tan_fovx=tan_fovx, tan_fovy=tan_fovy, cam=np.asarray(position, float),
                width=width, height=height)


def preprocess(scene, cam):
    W, H = cam["width"], cam["height"]
    n = len(scene["pos"])
    ph = np.concatenate([scene["pos"], np.ones((n, 1))], axis=1)
    p_hom = ph @ cam["proj"].T
    p_view = ph @ cam["view"].T
    ndc = p_hom[:, :2] / p_hom[:, 3:4]
    tz = p_view[:, 2]
    vis = tz > 0.2
    tzs = np.where(vis, tz, 1.0)
    limx, limy = 1.3 * cam["tan_fovx"], 1.3 * cam["tan_fovy"]
    tx = np.clip(p_view[:, 0] / tzs, -limx, limx) * tzs
    ty = np.clip(p_view[:, 1] / tzs, -limy, limy) * tzs
    fx, fy = W / (2 * cam["tan_fovx"]), H / (2 * cam["tan_fovy"])
    Jac = np.zeros((n, 2, 3))
    Jac[:, 0, 0] = fx / tzs
    Jac[:, 0, 2] = -fx * tx / tzs ** 2
    Jac[:, 1, 1] = fy / tzs
    Jac[:, 1, 2] = -fy * ty / tzs ** 2
    A = Jac @ cam["view"][:3, :3]
    cov2 = A @ cov3d(scene) @ np.transpose(A, (0, 2, 1))
    a, b, c = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    vis &= det > 0
    dets = np.where(vis, det, 1.0)
    conic = np.stack([c / dets, -b / dets, a / dets], axis=1)
    mid = 0.5 * (a + c)
    lam = mid + np.sqrt(np.maximum(0.1, mid * mid - det))
    radius = np.ceil(3.0 * np.sqrt(np.maximum(lam, 0)))
    uv = ((ndc + 1.0) * np.array([W, H]) - 1.0) * 0.5
    tw, th = (W + 15) // 16, (H + 15) // 16
    with np.errstate(invalid="ignore"):
        x0 = np.clip(np.trunc((uv[:, 0] - radius) / 16), 0, tw)
        y0 = np.clip(np.trunc((uv[:, 1] - radius) / 16), 0, th)
        x1 = np.clip(np.trunc((uv[:, 0] + radius + 15) / 16), 0, tw)
        y1 = np.clip(np.trunc((uv[:, 1] + radius + 15) / 16), 0, th)
    box = np.nan_to_num(np.stack([x0, y0, x1, y1], axis=1)).astype(np.int64)
    tiles = (box[:, 2] - box[:, 0]) * (box[:, 3] - box[:, 1])
    tiles = np.where(vis, tiles, 0)
    # SH
    d = scene["pos"] - cam["cam"]
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    sh = scene["sh"]
    rgb = SH_C0 * sh[:, 0] - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    rgb += SH_C2[0] * x * y * sh[:, 4] + SH_C2[1] * y * z * sh[:, 5] + SH_C2[2] * (2 * z * z - x * x - y * y) * sh[:, 6]
    rgb += SH_C2[3] * z * x * sh[:, 7] + SH_C2[4] * (x * x - y * y) * sh[:, 8]
    rgb += SH_C3[0] * (3 * x * x - y * y) * y * sh[:, 9] + SH_C3[1] * x * y * z * sh[:, 10]
    rgb += SH_C3[2] * (4 * z * z - x * x - y * y) * y * sh[:, 11]
    rgb += SH_C3[3] * z * (2 * z * z - 3 * x * x - 3 * y * y) * sh[:, 12]
    rgb += SH_C3[4] * x * (4 * z * z - x * x - y * y) * sh[:, 13]
    rgb += SH_C3[5] * (x * x - y * y) * z * sh[:, 14] + SH_C3[6] * x * (x * x - 3 * y * y) * sh[:, 15]
    rgb += 0.5
    rgb[:, 0] = np.maximum(rgb[:, 0], 0.0)
    return dict(tiles=tiles, box=box, conic=conic, radius=radius, uv=uv, depth=tz, rgb=rgb, opacity=scene["opacity"])


def render(pre, width, height):
    """Straight per-pixel blend over depth-sorted Gaussians (tile membership through the boxes)."""
    vis = np.nonzero(pre["tiles"])[0]
    order = vis[np.lexsort((vis, pre["depth"][vis].astype(np.float32).view(np.uint32)))]
    img = np.zeros((height, width, 4))
    img[..., 3] = 1.0
    T = np.ones((height, width))
    alive = np.ones((height, width), bool)
    ys, xs = np.mgrid[0:height, 0:width]
    for g in order:
        x0, y0, x1, y1 = pre["box"][g]
        sl = (slice(y0 * 16, min(y1 * 16, height)), slice(x0 * 16, min(x1 * 16, width)))
        dx = pre["uv"][g, 0] - xs[sl]
        dy = pre["uv"][g, 1] - ys[sl]
        c00, c01, c11 = pre["conic"][g]
        power = -0.5 * (c00 * dx * dx + c11 * dy * dy) - c01 * dx * dy
        alpha = np.minimum(0.99, pre["opacity"][g] * np.exp(np.minimum(power, 0)))
        ok = alive[sl] & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = T[sl] * (1 - alpha)
        kill = ok & (test_T < 1e-4)
        upd = ok & ~kill
        for k in range(3):
            img[sl + (k,)] += np.where(upd, pre["rgb"][g, k] * alpha * T[sl], 0.0)
        T[sl] = np.where(upd, test_T, T[sl])
        alive[sl] &= ~kill
    return img
