"""Deterministic synthetic scenes and ray generators.

The reference has no camera: callers hand `trace_rays` flat arrays of 16.16 fixed-point
start positions and directions (cuda_volume_raytracer.h:84-97).  "Orthographic" and
"perspective" in BASELINE.json's configs are ray *generators*; they live here.

Two coordinate frames are used:
  * volume frame  -- what RaytraceScene::trace_rays takes (image_util.cpp:645): voxel
    coordinates of the ior volume, valid range [1, N-1).
  * field frame   -- what TraceRaysCu::trace_rays_cu takes (the seam): the derivative
    field is the volume cropped by one voxel per side (valid 3x3x3 stencil), so
    field = volume - 1 voxel, and directions are already multiplied by the local ior
    (image_util.cpp:690-709).

All generators are pure numpy and seeded; nothing here touches the GPU.
"""
import numpy as np

UNIT_POS = 0x10000          # 16.16 fixed point (types.h:5)
UNIT_DIR_I16 = 0x100        # 8.8 fixed point   (types.h:46)
UNIT_IOR_U32 = 0x10000      # (types.h:19)
DIFF_PER_DLOGN = float(0x420000) / 256.0   # = 16896: diff value per unit d(log n)/dvoxel
#   ior_log = log(n) * 0x420000 (image_util.cpp:533/611), stencil gain 812, divided by 812*0x100 (:441)


def blob_centres(N, seed=1234, nblobs=4):
    rng = np.random.RandomState(seed)
    return (0.25 + 0.5 * rng.rand(nblobs, 3)) * N


def ior_and_grad(px, py, pz, N, centres, amp=0.5):
    """n = 1 + amp * sum_k exp(-|p-c_k|^2 * 16/N^2) and grad(log n), float64, broadcast over inputs."""
    a = 16.0 / (N * N)
    n = np.ones(np.broadcast(px, py, pz).shape)
    gx = np.zeros_like(n)
    gy = np.zeros_like(n)
    gz = np.zeros_like(n)
    for c in centres:
        ddx, ddy, ddz = px - c[0], py - c[1], pz - c[2]
        e = amp * np.exp(-(ddx * ddx + ddy * ddy + ddz * ddz) * a)
        n = n + e
        gx = gx - 2 * a * ddx * e
        gy = gy - 2 * a * ddy * e
        gz = gz - 2 * a * ddz * e
    return n, gx / n, gy / n, gz / n


def ior_volume(N, seed=1234, nblobs=4, amp=0.5, dtype=np.float32):
    """Smooth "sum of Gaussians" ior volume, shape [N,N,N] (x major, z minor).
    dtype float32 -> ior as float; uint32 -> 16.16 fixed (ior_t)."""
    c = blob_centres(N, seed, nblobs)
    out = np.empty((N, N, N), np.float32 if dtype == np.float32 else np.uint32)
    y = np.arange(N, dtype=np.float64)[:, None]
    z = np.arange(N, dtype=np.float64)[None, :]
    for x in range(N):
        n, _, _, _ = ior_and_grad(float(x), y, z, N, c, amp)
        out[x] = n.astype(np.float32) if dtype == np.float32 else np.round(n * UNIT_IOR_U32).astype(np.uint32)
    return out


def seam_field(N, dtype=np.float32, seed=1234, nblobs=4, amp=0.5, opaque_radius=None):
    """Synthetic inputs for the TraceRaysCu seam for an N^3 volume: the derivative field has
    bounds (N-2)^3 and holds diff = 16896 * d(log n)/dvoxel sampled analytically at volume
    voxel (x+1,y+1,z+1).  Returns (bounds, dx, dy, dz, transl_cropped), arrays flat, z minor.
    opaque_radius: if set, voxels within that radius (voxels) of any blob centre get
    translucency 0 (opaque => the w>0 test terminates rays, cuda_volume_raytracer.cu:343)."""
    B = N - 2
    c = blob_centres(N, seed, nblobs)
    dt = np.float32 if dtype == np.float32 else np.int16
    d = [np.empty((B, B, B), dt) for _ in range(3)]
    tr = np.full((B, B, B), 0xFFFFFFFF, np.uint32)
    y = np.arange(1, N - 1, dtype=np.float64)[:, None]
    z = np.arange(1, N - 1, dtype=np.float64)[None, :]
    for x in range(B):
        _, gx, gy, gz = ior_and_grad(float(x + 1), y, z, N, c, amp)
        for k, g in enumerate((gx, gy, gz)):
            v = g * DIFF_PER_DLOGN
            d[k][x] = v.astype(np.float32) if dt == np.float32 else np.round(v).astype(np.int16)
        if opaque_radius is not None:
            m = np.zeros((B, B), bool)
            for cc in c:
                m |= ((x + 1 - cc[0]) ** 2 + (y - cc[1]) ** 2 + (z - cc[2]) ** 2) < opaque_radius ** 2
            tr[x][m] = 0
    return [B, B, B], d[0].reshape(-1), d[1].reshape(-1), d[2].reshape(-1), tr.reshape(-1)


def _finish_seam_rays(pv, dirs, N, dir_dtype, seed, nblobs, amp):
    """pv: float64 [n,3] start points in VOLUME voxel coords; dirs: unit vectors.
    -> (start_pos uint32 [3n] in FIELD frame, start_dir [3n] scaled by local ior)."""
    c = blob_centres(N, seed, nblobs)
    n, _, _, _ = ior_and_grad(pv[:, 0], pv[:, 1], pv[:, 2], N, c, amp)
    pos = np.round((pv - 1.0) * UNIT_POS).astype(np.int64)
    assert pos.min() >= 0 and pos.max() < (N - 3) * UNIT_POS, "ray start outside the field"
    d = dirs * n[:, None]
    if dir_dtype == np.float32:
        sd = d.astype(np.float32)
    else:
        sd = np.round(d * UNIT_DIR_I16).astype(np.int16)
    return pos.astype(np.uint32).reshape(-1), sd.reshape(-1)


def _lattice(lo, hi, R, rows_total, row_range):
    """(row, col) coordinates of a rows_total x R lattice over [lo, hi]^2, restricted to the
    contiguous band of rows row_range = (r0, r1) (one band per GPU)."""
    rows_total = R if rows_total is None else rows_total
    r0, r1 = (0, rows_total) if row_range is None else row_range
    tr = np.linspace(lo, hi, rows_total)[r0:r1]
    tc = np.linspace(lo, hi, R)
    return np.meshgrid(tr, tc, indexing="ij")


def ortho_rays(N, R, dir_dtype=np.float32, seed=1234, nblobs=4, amp=0.5, rows_total=None, row_range=None):
    """R x R axis-parallel rays: start x = 1.5, (y,z) on a regular lattice over [1.5, N-2.5]
    (volume voxels), direction +x.  Ray index = iy * R + iz (z minor)."""
    yy, zz = _lattice(1.5, N - 2.5, R, rows_total, row_range)
    pv = np.stack([np.full(yy.size, 1.5), yy.reshape(-1), zz.reshape(-1)], 1)
    dirs = np.zeros_like(pv)
    dirs[:, 0] = 1.0
    return _finish_seam_rays(pv, dirs, N, dir_dtype, seed, nblobs, amp)


def perspective_rays(N, R, fov_deg=40.0, dir_dtype=np.float32, seed=1234, nblobs=4, amp=0.5, rows_total=None,
                     row_range=None):
    """R x R pinhole rays: eye at (-N, N/2, N/2), lattice over a fov_deg cone, clipped to the
    entry plane x = 1.5 (volume voxels)."""
    eye = np.array([-float(N), N / 2.0, N / 2.0])
    h = np.tan(np.radians(fov_deg) / 2.0)
    uu, vv = _lattice(-h, h, R, rows_total, row_range)
    dirs = np.stack([np.ones(uu.size), uu.reshape(-1), vv.reshape(-1)], 1)
    dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    s = (1.5 - eye[0]) / dirs[:, 0]
    pv = eye[None, :] + dirs * s[:, None]
    return _finish_seam_rays(pv, dirs, N, dir_dtype, seed, nblobs, amp)


def random_rays(N, nrays, dir_dtype=np.float32, seed=5678, vol_seed=1234, nblobs=4, amp=0.5):
    """Random in-bounds starts and random unit directions (fixture F2)."""
    rng = np.random.RandomState(seed)
    pv = 1.5 + rng.rand(nrays, 3) * (N - 4.0)
    d = rng.randn(nrays, 3)
    d /= np.linalg.norm(d, axis=1)[:, None]
    return _finish_seam_rays(pv, d, N, dir_dtype, vol_seed, nblobs, amp)


def uint8_density_and_lut(N, seed=1234, nblobs=4):
    """BASELINE config 5 (SURVEY §0 reading A): uint8 density = the same Gaussians quantised to
    0..255, and a 256-entry transfer function d -> (ior = 1 + d/510, translucency = 0xFFFFFFFF if
    d < 160 else 0).  Returns (density uint8 [N,N,N], lut_ior float32[256], lut_transl uint32[256])."""
    c = blob_centres(N, seed, nblobs)
    dens = np.empty((N, N, N), np.uint8)
    y = np.arange(N, dtype=np.float64)[:, None]
    z = np.arange(N, dtype=np.float64)[None, :]
    for x in range(N):
        n, _, _, _ = ior_and_grad(float(x), y, z, N, c, 0.5)
        dens[x] = np.clip(np.round((n - 1.0) / (0.5 * nblobs) * 255.0 * 2.0), 0, 255).astype(np.uint8)
    d = np.arange(256, dtype=np.float64)
    lut_ior = (1.0 + d / 510.0).astype(np.float32)
    lut_tr = np.where(d < 160, 0xFFFFFFFF, 0).astype(np.uint32)
    return dens, lut_ior, lut_tr
