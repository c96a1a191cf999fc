"""Seeded synthetic clouds of the bunny / dragon shape: a closed bumpy surface
r(dir) = 1 + sum_k a_k * exp(kappa_k * (dir . c_k - 1)), scaled to an anisotropic box.  The target
samples the whole surface; the source is an independent sample of a partial region of the same
surface, moved by a ground-truth rigid motion and perturbed by Gaussian noise, so that
R_gt @ src + t_gt lies on the target surface."""
import numpy as np

WORKLOADS = {
    # name: (nt, ns, box extents, seed) — vertex counts of data/bunny/bun000.ply / bun045.ply and of
    # the Stanford dragon_vrip model; boxes = AABBs of bun000.ply / dragonClearSpace2_0.ply
    "bunny": dict(nt=40256, ns=40097, box=(0.156, 0.152, 0.118), seed=1),
    "dragon": dict(nt=437645, ns=437645, box=(0.22, 0.22, 0.18), seed=2),
    "synthetic1m": dict(nt=1_000_000, ns=1_000_000, box=(0.2, 0.2, 0.2), seed=3),
    # BASELINE config 5: 20 % of the source replaced by uniform outliers in 1.5x the box (use with trim_fraction=0.2)
    "synthetic1m_outliers": dict(nt=1_000_000, ns=1_000_000, box=(0.2, 0.2, 0.2), seed=3, outlier_frac=0.2),
    # BASELINE configs[0] shape: test/bunny.toml subsamples data/bunny to ~0.5 * 35 947 target and ~0.1 * 30 379 source points
    # and builds the LUT at resolution 0.002 (923 x 906 x 711 nodes on the real clouds)
    "bunny_toml": dict(nt=17973, ns=3037, box=(0.156, 0.152, 0.118), seed=5),
    "mid": dict(nt=150_000, ns=150_000, box=(0.156, 0.152, 0.118), seed=13),  # between the two: 0.36 source points per LUT face voxel
    "tiny": dict(nt=1500, ns=1200, box=(0.156, 0.152, 0.118), seed=7),
    "small": dict(nt=6000, ns=5000, box=(0.156, 0.152, 0.118), seed=11),
}


def random_rotation(rng, max_angle_deg=None, min_angle_deg=0.0):
    """Rotation about a uniform random axis; uniform on SO(3) when max_angle_deg is None."""
    if max_angle_deg is None:
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(min_angle_deg, max_angle_deg))
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


class BumpySurface:
    def __init__(self, rng, box, n_bumps=24):
        self.box = np.asarray(box, dtype=np.float64) / 2.0
        c = rng.normal(size=(n_bumps, 3))
        self.centers = c / np.linalg.norm(c, axis=1, keepdims=True)
        self.amps = rng.uniform(-0.25, 0.45, size=n_bumps)
        self.kappas = rng.uniform(4.0, 30.0, size=n_bumps)

    def sample(self, rng, n, cap_dir=None, cap_fraction=1.0):
        out = np.empty((0, 3))
        cos_cut = 1.0 - 2.0 * cap_fraction  # spherical cap holding `cap_fraction` of the sphere
        while len(out) < n:
            d = rng.normal(size=(int((n - len(out)) / max(cap_fraction, 0.05) * 1.2) + 16, 3))
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            if cap_dir is not None and cap_fraction < 1.0:
                d = d[d @ cap_dir >= cos_cut]
            out = np.concatenate([out, d])
        d = out[:n]
        r = 1.0 + (self.amps[None, :] * np.exp(self.kappas[None, :] * (d @ self.centers.T - 1.0))).sum(axis=1)
        return d * r[:, None] * self.box[None, :]


def make_pair(nt, ns, box, seed, overlap=0.7, noise=1e-3, angle_deg=None, min_angle_deg=0.0, t_frac=0.25, outlier_frac=0.0):
    """Returns (target (nt,3) f32, source (ns,3) f32, R_gt (3,3) f64, t_gt (3,) f64)."""
    rng = np.random.default_rng(seed)
    surf = BumpySurface(rng, box)
    tgt = surf.sample(rng, nt)
    cap = rng.normal(size=3)
    cap /= np.linalg.norm(cap)
    on_target = surf.sample(rng, ns, cap_dir=cap, cap_fraction=overlap)
    on_target = on_target + rng.normal(scale=noise * float(np.max(box)), size=on_target.shape)
    R_gt = random_rotation(rng, angle_deg, min_angle_deg)
    t_gt = rng.uniform(-t_frac, t_frac, size=3) * np.asarray(box)
    src = (on_target - t_gt[None, :]) @ R_gt  # rows: R_gt^T (x - t)
    if outlier_frac > 0:
        n_out = int(ns * outlier_frac)
        idx = rng.choice(ns, n_out, replace=False)
        src[idx] = rng.uniform(-0.75, 0.75, size=(n_out, 3)) * np.asarray(box)
    return tgt.astype(np.float32), src.astype(np.float32), R_gt, t_gt


def workload(name, **over):
    cfg = dict(WORKLOADS[name])
    cfg.update(over)
    return make_pair(cfg.pop("nt"), cfg.pop("ns"), cfg.pop("box"), cfg.pop("seed"), **cfg)


def preprocess(tgt, src):
    """numpy restatement of the driver's pre-processing for tests that need scaled clouds without
    a solver: centre both, scale by 1/max|src| (fgoicp.cpp:176-287).  fp32 throughout, serial sums."""
    def center(pc):
        c = np.zeros(3, np.float32)
        for k in range(3):
            c[k] = np.cumsum(pc[:, k], dtype=np.float32)[-1]  # serial fp32 sum, in order (a running sum cannot be re-associated)
        c = (c / np.float32(len(pc))).astype(np.float32)
        return (pc - c[None, :]).astype(np.float32), (-c).astype(np.float32)
    s_c, off_s = center(src.astype(np.float32))
    t_c, off_t = center(tgt.astype(np.float32))
    scale = np.float32(1.0) / np.float32(np.max(np.abs(s_c)))
    s_c = (s_c * scale).astype(np.float32)
    t_c = (t_c * scale).astype(np.float32)
    bounds = np.array([[t_c[:, k].min(), t_c[:, k].max()] for k in range(3)], dtype=np.float32)
    return t_c, s_c, off_t, off_s, scale, bounds
