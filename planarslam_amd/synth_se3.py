"""Synthetic RGB-D streams rendered from SE3 camera motion in a box room: depth AND gray of every frame come from the
same geometry and the same camera pose, so what the tracker sees from frame to frame is what a moving camera produces - parallax that depends on depth, planes
whose coefficients change with the rotation, a Manhattan frame that turns against the camera - and the true pose of every frame is known.

The room: floor, ceiling and four walls (the world frame IS the room's Manhattan frame: x right, y down, z forward) and a table-sized box on the floor.
Every face carries the scene's texture (one of synth.gray_image's canvases, mirror-repeated, shifted and scaled per face, with a per-face gain so that the
room's edges are image edges).  A frame = ray casting against the eleven faces + a bilinear texture fetch; torch tensors, so the bench renders its streams on
the GPU before the timed region and the CPU tests render small ones with the same code.  Test / bench infrastructure: nothing in the product path imports it."""
from __future__ import annotations

import numpy as np

TEX_PX_PER_M = 190.0          # texture scale: one metre of wall = 190 texels (about one texel per pixel at 2.8 m with fx = 535)


def _rodrigues(w):
    th = float(np.linalg.norm(w))
    if th < 1e-12:
        return np.eye(3)
    k = np.asarray(w, np.float64) / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def room_scene(seed):
    """Faces of one room: arrays n [F,3], d [F] (n.X + d = 0, n towards the camera side), a, b [F,3] (in-plane axes: texture u, v and the rectangle test),
    lo, hi [F,2] (bounds of (a.X, b.X); +-inf for the room's own faces), toff [F,2] (texture offset in texels), gain [F]."""
    rng = np.random.default_rng(1000003 * seed + 17)
    hf, hc = rng.uniform(0.9, 1.4), rng.uniform(1.0, 1.5)
    zb, zf = rng.uniform(2.6, 4.0), rng.uniform(1.0, 2.0)
    xl, xr = rng.uniform(1.3, 2.5), rng.uniform(1.3, 2.5)
    bx0, bx1 = sorted(rng.uniform(-0.8, 0.8, 2))
    if bx1 - bx0 < 0.35:
        bx1 = bx0 + 0.35
    by = hf - rng.uniform(0.45, 0.8)
    bz0 = rng.uniform(1.3, 2.0)
    bz1 = min(bz0 + rng.uniform(0.4, 0.9), zb - 0.2)
    X, Y, Z = np.eye(3)
    inf = np.inf
    F = [  # n, d, a, b, lo, hi
        (-Y, hf, X, Z, (-inf, -inf), (inf, inf)),          # floor y = hf
        (Y, hc, X, Z, (-inf, -inf), (inf, inf)),           # ceiling y = -hc
        (-Z, zb, X, Y, (-inf, -inf), (inf, inf)),          # back wall z = zb
        (Z, zf, X, Y, (-inf, -inf), (inf, inf)),           # wall behind the camera z = -zf
        (X, xl, Z, Y, (-inf, -inf), (inf, inf)),           # left wall x = -xl
        (-X, xr, Z, Y, (-inf, -inf), (inf, inf)),          # right wall x = xr
        (-Y, by, X, Z, (bx0, bz0), (bx1, bz1)),            # box top y = by
        (-Z, bz0, X, Y, (bx0, by), (bx1, hf)),             # box front z = bz0
        (X, -bx0, Z, Y, (bz0, by), (bz1, hf)),             # box left side x = bx0 (seen from x < bx0)
        (-X, bx1, Z, Y, (bz0, by), (bz1, hf)),             # box right side x = bx1
        (Z, -bz1, X, Y, (bx0, by), (bx1, hf)),             # box back z = bz1 (seen from behind)
    ]
    nF = len(F)
    out = dict(n=np.array([f[0] for f in F], np.float64), d=np.array([f[1] for f in F], np.float64), a=np.array([f[2] for f in F], np.float64),
               b=np.array([f[3] for f in F], np.float64), lo=np.array([f[4] for f in F], np.float64), hi=np.array([f[5] for f in F], np.float64),
               toff=rng.uniform(0, 4000, (nF, 2)), gain=rng.uniform(0.62, 1.0, nF))
    out["gain"][0] = rng.uniform(0.85, 1.0)
    # the camera's resting pose: near the origin, turned towards a corner of the back wall (two walls, floor and box in view) and pitched down, a few degrees of roll
    yaw, pitch, roll = rng.uniform(-0.55, 0.55), rng.uniform(0.08, 0.30), rng.normal() * 0.04
    out["R0"] = _rodrigues(Y * yaw) @ _rodrigues(X * -pitch) @ _rodrigues(Z * roll)
    out["p0"] = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.15, 0.15), rng.uniform(-0.3, 0.3)])
    return out


def camera_path(seed, K, step_t=0.012, step_r=np.deg2rad(0.25)):
    """K camera poses Twc [K,4,4] (camera -> world) of a stream: constant-velocity sweep along a random direction with a rotation about a random axis close to
    one of the Manhattan axes (<= step_t metres and step_r radians per frame); played forwards and backwards (frame_index) it is a continuous motion."""
    rng = np.random.default_rng(7919 * seed + 5)
    dirt = rng.normal(size=3); dirt /= np.linalg.norm(dirt)
    axis = np.eye(3)[rng.integers(0, 3)] + rng.normal(size=3) * 0.15
    axis /= np.linalg.norm(axis)
    T = np.tile(np.eye(4), (K, 1, 1))
    for j in range(K):
        s = j - (K - 1) / 2.0
        T[j, :3, :3] = _rodrigues(axis * step_r * s)
        T[j, :3, 3] = dirt * step_t * s
    return T


def frame_index(i, K, hold=1):
    """Frame of the K-frame loop shown at step i: 0 for the first 1 + hold steps (the tracker's first tracked frame is its third: the camera rests until its local
    map - the previous two frames - exists), then 1, .., K-1, K-2, .., 1, 0, 1, .."""
    i = max(i - hold, 0)
    if K <= 1:
        return 0
    m = i % (2 * K - 2)
    return m if m < K else 2 * K - 2 - m


def stream_poses(scene, seed, K):
    """Twc [K,4,4] of the stream `seed` in `scene`: the scene's resting pose times the stream's path."""
    P = camera_path(seed, K)
    T0 = np.eye(4); T0[:3, :3] = scene["R0"]; T0[:3, 3] = scene["p0"]
    return np.einsum("ij,kjl->kil", T0, P)


def relative_pose(Twc, j, ref=0):
    """Tcw of frame j in the frame of camera `ref` (the tracker's world when its map was built from frame `ref` at the identity pose): inv(Twc[j]) Twc[ref]."""
    return np.linalg.inv(Twc[j]) @ Twc[ref]


def render(torch, scenes, tex, Twc, cam, W=640, H=480, noise_seed=0, factor=5000.0, depth_noise=True, holes=True, pixel_noise=2.0):
    """scenes: list of S room_scene dicts; tex: uint8 tensor [S, Ht, Wt] on the target device; Twc: float64 array [S, 4, 4] (one pose per scene).
    -> gray uint8 [S, H, W], depth int16 (the bits of TUM's uint16: metres * factor) [S, H, W], on tex.device."""
    dev = tex.device
    S = len(scenes)
    f32 = torch.float32
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
    nF = len(scenes[0]["d"])
    n = t(np.stack([s["n"] for s in scenes])); d = t(np.stack([s["d"] for s in scenes]))
    a = t(np.stack([s["a"] for s in scenes])); b = t(np.stack([s["b"] for s in scenes]))
    lo = t(np.stack([s["lo"] for s in scenes])); hi = t(np.stack([s["hi"] for s in scenes]))
    toff = t(np.stack([s["toff"] for s in scenes])); gain = t(np.stack([s["gain"] for s in scenes]))
    R = t(Twc[:, :3, :3]); o = t(Twc[:, :3, 3])
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=f32), torch.arange(W, device=dev, dtype=f32), indexing="ij")
    rc = torch.stack([(xs - cam["cx"]) / cam["fx"], (ys - cam["cy"]) / cam["fy"], torch.ones_like(xs)], 0)      # [3, H, W], z = 1: t is the depth
    dot3 = lambda v, F: v[:, 0, None, None] * F[:, 0] + v[:, 1, None, None] * F[:, 1] + v[:, 2, None, None] * F[:, 2]       # [S, 3] . [S, 3, H, W] (no BLAS call: the
    rw = torch.stack([R[:, i, 0, None, None] * rc[0] + R[:, i, 1, None, None] * rc[1] + R[:, i, 2, None, None] * rc[2] for i in range(3)], 1)   # profiles stay ours)
    best_t = torch.full((S, H, W), float("inf"), device=dev, dtype=f32)
    best_u = torch.zeros((S, H, W), device=dev, dtype=f32); best_v = torch.zeros_like(best_u); best_g = torch.zeros_like(best_u)
    for f in range(nF):
        nf = n[:, f]                                                                                               # [S, 3]
        den = dot3(nf, rw)
        num = -((nf * o).sum(1) + d[:, f])[:, None, None]
        tt = torch.where(den.abs() > 1e-9, num / den, torch.full_like(den, float("inf")))
        Xh = o[:, :, None, None] + tt[:, None] * rw                                                                # hit points [S, 3, H, W]
        u = dot3(a[:, f], Xh); v = dot3(b[:, f], Xh)
        ok = (tt > 0.05) & (tt < best_t) & (u >= lo[:, f, 0, None, None]) & (u <= hi[:, f, 0, None, None]) & (v >= lo[:, f, 1, None, None]) & (v <= hi[:, f, 1, None, None])
        best_t = torch.where(ok, tt, best_t)
        best_u = torch.where(ok, u * TEX_PX_PER_M + toff[:, f, 0, None, None], best_u)
        best_v = torch.where(ok, v * TEX_PX_PER_M + toff[:, f, 1, None, None], best_v)
        best_g = torch.where(ok, gain[:, f, None, None].expand(S, H, W), best_g)
    hit = torch.isfinite(best_t)
    Ht, Wt = tex.shape[1:]

    def mirror(x, size):                                  # mirror-repeat onto [0, size - 1]
        p = 2.0 * (size - 1)
        x = torch.remainder(x, p)
        return torch.where(x > size - 1, p - x, x)
    u = mirror(torch.where(hit, best_u, torch.zeros_like(best_u)), Wt); v = mirror(torch.where(hit, best_v, torch.zeros_like(best_v)), Ht)
    u0 = u.floor().clamp(0, Wt - 2); v0 = v.floor().clamp(0, Ht - 2)
    fu = u - u0; fv = v - v0
    si = torch.arange(S, device=dev)[:, None, None].expand(S, H, W)
    u0 = u0.long(); v0 = v0.long()
    texf = tex
    g00 = texf[si, v0, u0].to(f32); g01 = texf[si, v0, u0 + 1].to(f32); g10 = texf[si, v0 + 1, u0].to(f32); g11 = texf[si, v0 + 1, u0 + 1].to(f32)
    val = (g00 * (1 - fu) + g01 * fu) * (1 - fv) + (g10 * (1 - fu) + g11 * fu) * fv
    gen = torch.Generator(device=dev); gen.manual_seed(int(noise_seed))
    val = val * best_g
    if pixel_noise:
        val = val + torch.randn(val.shape, generator=gen, device=dev, dtype=f32) * pixel_noise
    gray = torch.where(hit, val, torch.zeros_like(val)).round().clamp(0, 255).to(torch.uint8)
    z = torch.where(hit, best_t, torch.zeros_like(best_t))
    if depth_noise:
        z = z + torch.randn(z.shape, generator=gen, device=dev, dtype=f32) * 0.0012 * z * z
    dq = (z * factor).round().clamp(0, 65535)
    if holes:                                             # ~3 % of the pixels zeroed in blobs, as synth.depth_image
        hc = torch.rand((S, 12, 3), generator=gen, device=dev, dtype=f32)
        cx = hc[:, :, 0] * W; cy = hc[:, :, 1] * H; r = 8 + hc[:, :, 2] * 32
        for k in range(12):
            m = (xs[None] - cx[:, k, None, None]) ** 2 + (ys[None] - cy[:, k, None, None]) ** 2 < (r[:, k, None, None]) ** 2
            dq = torch.where(m, torch.zeros_like(dq), dq)
    depth = dq.to(torch.int32)
    depth = torch.where(depth > 32767, depth - 65536, depth).to(torch.int16)       # the uint16's bits
    return gray, depth


def render_streams(torch, tex, B, K, cam, seed=0, W=640, H=480, chunk=32, **kw):
    """B streams x K frames: stream s lives in scene s mod S (S = tex.shape[0]) on its own camera path.  -> gray uint8 [B, K, H, W], depth int16 [B, K, H, W] (both on
    tex.device), Twc float64 [B, K, 4, 4]."""
    S = tex.shape[0]
    scenes = [room_scene(seed * 4096 + s) for s in range(S)]
    dev = tex.device
    gray = torch.empty((B, K, H, W), dtype=torch.uint8, device=dev); depth = torch.empty((B, K, H, W), dtype=torch.int16, device=dev)
    Twc = np.stack([stream_poses(scenes[s % S], seed * 65536 + s, K) for s in range(B)])
    for s0 in range(0, B, chunk):
        s1 = min(B, s0 + chunk)
        idx = [s % S for s in range(s0, s1)]
        tsel = tex[torch.as_tensor(idx, device=dev)]
        for j in range(K):
            g, d = render(torch, [scenes[i] for i in idx], tsel, Twc[s0:s1, j], cam, W, H, noise_seed=(seed * 1000003 + s0) * 64 + j, **kw)
            gray[s0:s1, j] = g; depth[s0:s1, j] = d
    return gray, depth, Twc
