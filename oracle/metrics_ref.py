"""Oracle: teacher-vs-student trajectory metrics on torch-CPU / numpy / scipy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
  * ``analysis/metrics/trajectory_metrics.py:12-325`` compute_trajectory_metrics
  * ``analysis/metrics/time_dependent.py:10-120``     analyze_time_dependent_distances (numeric part)
  * ``utils/metric_transformations.py:3-38``          transform_metrics
organised around the five per-step reductions the device kernel produces
(D_i, Vt_i, Vs_i, <dX_i,dY_i>, W1_i) but evaluated with the same torch/numpy/scipy
calls and dtypes as the reference, so results agree to the last bit on one machine.
"""
import numpy as np
import torch
from scipy.interpolate import interp1d
from scipy.stats import wasserstein_distance


def _images(traj):
    """trajectory_metrics.py:29-37: entries may be tensors or (tensor, t) tuples."""
    return [e[0] for e in traj] if isinstance(traj[0], tuple) else list(traj)


def _fro(a):
    return torch.norm(a).item()


def compute_trajectory_metrics(teacher_trajectory, student_trajectory, config=None):
    X, Y = _images(teacher_trajectory), _images(student_trajectory)
    if X[-1].shape != Y[-1].shape and X[-1].shape[2:] != Y[-1].shape[2:]:      # :40-52
        Y = [torch.nn.functional.interpolate(y, size=X[0].shape[2:], mode="bilinear", align_corners=True) for y in Y]
    m = {}
    n = min(len(X), len(Y))
    pixels = X[0].shape[2] * X[0].shape[3]

    # endpoint / final-image terms (:55-59)
    m["endpoint_distance"] = _fro(X[-1] - Y[-1])
    mse = torch.mean((X[-1] - Y[-1]) ** 2).item()
    m["mse"] = mse

    # trajectory MSE similarity (:62-86); NaN when 1000*mean > 2
    acc = 0.0
    for i in range(n):
        acc += torch.mean((X[i] - Y[i]) ** 2).item()
    m["trajectory_mse"] = np.log1p(1.0 - (acc / n) * 1000)

    # point-by-point similarity (:90-101)
    D = [_fro(X[i] - Y[i]) for i in range(n)]
    m["point_by_point_similarity"] = np.exp(-5.0 * (np.mean(D) if D else float("inf")))

    # log-MSE similarity (:106-108)
    m["log_mse_similarity"] = max(0, 1.0 - np.log1p(mse * 5000) / np.log1p(5000))

    # path lengths (:111-131)
    tl = sl = 0
    for i in range(1, n):
        tl += _fro(X[i] - X[i - 1]) / pixels
        sl += _fro(Y[i] - Y[i - 1]) / pixels
    tl /= (n - 1)
    sl /= (n - 1)
    m["teacher_path_length"], m["student_path_length"] = tl, sl
    m["path_length_similarity"] = np.log1p(min(tl, sl) / max(tl, sl) if max(tl, sl) > 0 else 1.0)   # :134-137

    # efficiency (:140-153)
    te = _fro(X[-1] - X[0]) / tl if tl > 0 else 0
    se = _fro(Y[-1] - Y[0]) / sl if sl > 0 else 0
    m["teacher_efficiency"], m["student_efficiency"] = te, se
    m["efficiency_similarity"] = np.log1p(min(te, se) / max(te, se) if max(te, se) > 0 else 1.0)

    # velocity profiles over each trajectory's own length (:156-177)
    Vt = [_fro(X[i] - X[i - 1]) for i in range(1, len(X))]
    Vs = [_fro(Y[i] - Y[i - 1]) for i in range(1, len(Y))]
    m["teacher_velocities"], m["student_velocities"] = Vt, Vs
    vsim = [(min(a, b) / max(a, b) if max(a, b) > 0 else 1.0) for a, b in zip(Vt, Vs)]
    m["velocity_similarities"] = vsim
    m["mean_velocity_similarity"] = np.mean(vsim) if vsim else 0.0

    # position differences (:180-187)
    m["position_differences"] = list(D)
    m["mean_position_difference"] = np.mean(D) if D else 0.0
    m["max_position_difference"] = np.max(D) if D else 0.0

    # directional consistency (:190-231); steps where either move is zero are skipped
    cos, wcos = [], []
    for i in range(n - 1):
        dx, dy = X[i + 1] - X[i], Y[i + 1] - Y[i]
        nx, ny = torch.norm(dx), torch.norm(dy)
        if nx > 0 and ny > 0:
            fx, fy = dx.flatten(), dy.flatten()
            c = (torch.sum(fx * fy) / (torch.norm(fx) * torch.norm(fy))).item()
            cos.append(c)
            wcos.append(c * ((nx.item() + ny.item()) / 2))
    m["directional_consistency"] = cos
    m["mean_directional_consistency"] = np.mean(cos) if cos else 0.0
    if wcos:
        total_w = sum((Vt[i] + Vs[i]) / 2 for i in range(min(len(Vt), len(Vs))))
        m["weighted_directional_consistency"] = (sum(wcos) / total_w if total_w > 0 else 0) ** 2
    else:
        m["weighted_directional_consistency"] = 0.0

    # path alignment (:239-293); the longer trajectory is resampled onto the shorter's grid
    fx = [x.flatten().cpu().numpy() for x in X]
    fy = [y.flatten().cpu().numpy() for y in Y]
    if len(X) != len(Y):
        longer, shorter = (fx, fy) if len(fx) > len(fy) else (fy, fx)
        lt, st = np.linspace(0, 1, len(longer)), np.linspace(0, 1, len(shorter))
        funcs = [interp1d(lt, [v[d] for v in longer]) for d in range(len(longer[0]))]
        res = [np.array([f(t) for f in funcs]) for t in st]
        fx, fy = (res, shorter) if len(X) > len(Y) else (shorter, res)
    pd = [np.linalg.norm(a - b) for a, b in zip(fx, fy)]
    m["path_alignment"] = np.exp(-10.0 * np.sum(pd) / len(pd))

    # Wasserstein per zipped step on <=1000 sub-sampled coordinates (:295-323)
    fx = [x.flatten().cpu().numpy() for x in X]
    fy = [y.flatten().cpu().numpy() for y in Y]
    W = []
    for a, b in zip(fx, fy):
        idx = np.random.choice(len(a), min(1000, len(a)), replace=False)
        W.append(wasserstein_distance(a[idx], b[idx]))
    m["wasserstein_distances"] = W
    m["mean_wasserstein"] = np.mean(W)
    m["distribution_similarity"] = np.log1p(np.exp(-m["mean_wasserstein"]))
    return m


def time_dependent_distances(teacher_trajectories, student_trajectories):
    """analysis/metrics/time_dependent.py:46-120 without printing/plotting."""
    def per_traj(trajs):
        out = []
        for tr in trajs:
            im = _images(tr)
            d = [_fro(im[i] - im[i - 1]) for i in range(1, len(im))]
            if d:
                out.append(d)
        return out

    def stats(all_d):
        if not all_d:
            return [], 0, 0
        L = min(len(d) for d in all_d)
        avg = [sum(d[t] for d in all_d) / len(all_d) for t in range(L)]
        mean = sum(avg) / len(avg) if avg else 0
        std = (sum((a - mean) ** 2 for a in avg) / len(avg)) ** 0.5 if avg else 0
        return avg, mean, std

    res = {"teacher_distances": [], "student_distances": [], "teacher_avg_distance": 0, "student_avg_distance": 0,
           "teacher_std_distance": 0, "student_std_distance": 0}
    if not teacher_trajectories or not student_trajectories:
        return res
    td, sd = per_traj(teacher_trajectories), per_traj(student_trajectories)
    res["teacher_distances"], res["student_distances"] = td, sd
    if td and sd:
        ta, tm, ts = stats(td)
        sa, sm, ss = stats(sd)
    else:
        ta, tm, ts, sa, sm, ss = [], 0, 0, [], 0, 0
    res.update(teacher_avg_per_timestep=ta, student_avg_per_timestep=sa, teacher_avg_distance=tm,
               student_avg_distance=sm, teacher_std_distance=ts, student_std_distance=ss)
    return res


def transform_metrics(path_length_similarity, trajectory_mse, directional_consistency, distribution_similarity):
    """utils/metric_transformations.py:3-38."""
    mse = np.clip(1 - np.log1p(np.clip(trajectory_mse, 0, None)) / np.log1p(1.0), 0, 1)
    dist = np.clip(np.log1p(distribution_similarity) / np.log1p(1.0), 0, 1)
    return {
        "path_length_similarity": path_length_similarity,
        "trajectory_mse": mse,
        "mean_directional_consistency": np.abs(directional_consistency),
        "distribution_similarity": dist,
    }


def trajectory_divergence(trajectory1, trajectory2):
    """evaluation/metrics.py:118-183 (entries are (image, timestep) pairs)."""
    from sklearn.metrics.pairwise import cosine_similarity
    im1 = [item[0] for item in trajectory1]
    im2 = [item[0] for item in trajectory2]
    distances = [torch.norm(a.flatten() - b.flatten()).item() for a, b in zip(im1, im2)]
    sims = [cosine_similarity(a.flatten().cpu().numpy().reshape(1, -1), b.flatten().cpu().numpy().reshape(1, -1))[0, 0]
            for a, b in zip(im1, im2)]
    l1 = l2 = 0
    for i in range(1, len(im1)):
        l1 += torch.norm(im1[i] - im1[i - 1]).item()
    for i in range(1, len(im2)):
        l2 += torch.norm(im2[i] - im2[i - 1]).item()
    return {"distances": distances, "similarities": sims, "avg_distance": np.mean(distances),
            "max_distance": np.max(distances), "avg_similarity": np.mean(sims), "min_similarity": np.min(sims),
            "length_ratio": l2 / l1 if l1 > 0 else float("inf")}


def noise_metrics(teacher_noise, student_noise):
    """analysis/noise_prediction/noise_analysis.py:43-85 for equal shapes."""
    mse = torch.mean((teacher_noise - student_noise) ** 2).item()
    mae = torch.mean(torch.abs(teacher_noise - student_noise)).item()
    tf = torch.nn.functional.normalize(teacher_noise.view(teacher_noise.size(0), -1), p=2, dim=1)
    sf = torch.nn.functional.normalize(student_noise.view(student_noise.size(0), -1), p=2, dim=1)
    return {"mse": mse, "mae": mae, "cosine_similarity": torch.mean(torch.sum(tf * sf, dim=1)).item()}


def calculate_fid(features_1, features_2):
    """analysis/metrics/fid_score.py:61-93."""
    from scipy import linalg
    if len(features_1) < 2 or len(features_2) < 2:
        return 999.0
    mu1, s1 = features_1.mean(axis=0), np.cov(features_1, rowvar=False)
    mu2, s2 = features_2.mean(axis=0), np.cov(features_2, rowvar=False)
    covmean = linalg.sqrtm(s1.dot(s2))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return np.sum((mu1 - mu2) ** 2.0) + np.trace(s1 + s2 - 2.0 * covmean)
