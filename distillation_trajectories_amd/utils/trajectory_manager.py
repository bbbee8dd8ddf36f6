"""Drop-in for the reference's ``utils/trajectory_manager.py`` on the HIP path.

Same class, method names, return shapes, pickle format and error behaviour as reference
``utils/trajectory_manager.py:9-581``: trajectories are lists of ``(x[1,C,H,W], t)`` recorded BEFORE
each update, the update rule is the reference's placeholder ``x = (x - 0.1*eps)/sqrt(0.9) +
0.1*(t/teacher_steps)*z`` (:167-205, also for the student), pairs are pickled to
``{trajectory_dir}/trajectory_size_{sf}_sample_{i}.pkl`` and per-sample failures are printed and
skipped.  The model passes and updates run in csrc/ kernels (rule MANAGER); Gaussian noise is drawn
on the host from the CPU generator in the reference's order.
"""
import os
import pickle

import numpy as np
import torch

from .. import engine
from .._hip import COND_NONE, RULE_MANAGER
from ..analysis.metrics.trajectory_metrics import compute_trajectory_metrics, compute_trajectory_metrics_many


def timestep_list(sample_steps, steps):
    """Evenly spaced indices plus the forced last one (reference :88-98), in visiting (descending) order."""
    stride = sample_steps // steps
    idx = [i * stride for i in range(steps)]
    if idx[-1] != sample_steps - 1:
        idx.append(sample_steps - 1)
    return list(reversed(idx))


class TrajectoryManager:
    """Generate, store, reload and score (teacher, student) trajectory pairs."""

    def __init__(self, teacher_model, student_model, config, size_factor=1.0, fixed_samples=None):
        self.teacher_model = teacher_model
        self.student_model = student_model
        self.config = config
        self.size_factor = size_factor
        self.fixed_samples = self._ensure_tensor_compatibility(fixed_samples) if fixed_samples is not None else None
        os.makedirs(config.trajectory_dir, exist_ok=True)
        self.device = next(teacher_model.parameters()).device

    def _ensure_tensor_compatibility(self, tensor_or_batch):
        """Pass-through with the reference's diagnostics (:41-63)."""
        if tensor_or_batch is None:
            return None
        if not isinstance(tensor_or_batch, torch.Tensor):
            print(f"Warning: Input is not a tensor but {type(tensor_or_batch)}")
            return tensor_or_batch
        print(f"Tensor shape: {tensor_or_batch.shape}")
        return tensor_or_batch

    # ------------------------------------------------------------------ one model, one loop
    def _manager_coefficients(self, indices):
        """(beta, sqrt(alpha), noise_scale_t) with the reference's dtype path (:181-203)."""
        beta = 1 - 0.9
        sqrt_alpha = float(torch.sqrt(torch.tensor(0.9)))
        steps = float(self.config.teacher_steps)        # the student also divides by teacher_steps
        return [(beta, sqrt_alpha, 0.1 * (float(t) / steps)) for t in indices]

    def _run_states(self, model, x0, indices, z):
        """Device states [n_upd+1, S, E] of one model for S start images x0[S,C,H,W]: record, predict, update
        while t > 0 (reference :100-112).  z[n_upd, S, E] holds the step noise in draw order."""
        h = engine.UNetHandle.for_module(model)
        S, C, H, W = x0.shape
        E = C * H * W
        n_upd = sum(1 for t in indices if t > 0)        # every visited t except a trailing 0 is followed by an update
        upd = indices[:n_upd]
        traj = torch.empty(n_upd + 1, S, E, dtype=torch.float32, device=self.device)
        traj[0].copy_(x0.reshape(S, E))
        if n_upd:
            tb = h.time_bias(upd, [COND_NONE] * n_upd)
            h.sample(RULE_MANAGER, traj, H, W, tb, 1, self._manager_coefficients(upd), [True] * n_upd,
                     z=z.reshape(-1, E).to(self.device), z_shift=[k * S for k in range(n_upd)])
        return traj, n_upd

    def _as_pairs(self, states, n_upd, indices, sample, shape):
        """[(x[1,C,H,W] on self.device, t)] of one sample from the batched device states."""
        C, H, W = shape
        out = [(states[i, sample].reshape(1, C, H, W).clone(), t) for i, t in enumerate(indices[: n_upd + 1])]
        # indices after the first t == 0 (never produced by timestep_list) would repeat the last state
        for t in indices[n_upd + 1:]:
            out.append((states[n_upd, sample].reshape(1, C, H, W).clone(), t))
        return out

    def _run(self, model, x0, indices):
        """[(x[1,C,H,W] on self.device, t)] for one model and one start image; draws the step noise from
        the CPU generator in the reference's order (one draw per update)."""
        _, C, H, W = x0.shape
        n_upd = sum(1 for t in indices if t > 0)
        zs = [torch.randn(1, C, H, W) for _ in range(n_upd)]
        z = torch.stack(zs).reshape(n_upd, 1, C * H * W) if zs else None
        states, n_upd = self._run_states(model, x0, indices, z)
        return self._as_pairs(states, n_upd, indices, 0, (C, H, W))

    def generate_trajectories_batched(self, seeds):
        """[(teacher_trajectory, student_trajectory)] for many seeds with ONE batched loop per model
        (what generate_and_save_trajectories uses): per seed the CPU generator is re-seeded, x_T drawn,
        then one draw per update -- the student re-seeds and therefore reuses the same stream (:114-116)."""
        cfg = self.config
        size = self._student_size()
        if size != cfg.image_size:
            return [self.generate_trajectory(seed=s) for s in seeds]     # different student resolution: no shared noise
        self.teacher_model.eval()
        self.student_model.eval()
        shape = (cfg.channels, cfg.image_size, cfg.image_size)
        t_idx = timestep_list(cfg.sample_steps, cfg.teacher_steps)
        s_idx = timestep_list(cfg.sample_steps, cfg.student_steps)
        n_t, n_s = sum(1 for t in t_idx if t > 0), sum(1 for t in s_idx if t > 0)
        x0, z = [], []
        for seed in seeds:
            torch.manual_seed(seed)
            np.random.seed(seed)
            x0.append(torch.randn(1, *shape))
            z.append(torch.stack([torch.randn(1, *shape) for _ in range(max(n_t, n_s))]) if max(n_t, n_s) else None)
        x0 = torch.cat(x0)
        zz = torch.stack(z, dim=1).reshape(max(n_t, n_s), len(seeds), -1) if max(n_t, n_s) else None
        t_states, _ = self._run_states(self.teacher_model, x0, t_idx, None if zz is None else zz[:n_t])
        s_states, _ = self._run_states(self.student_model, x0, s_idx, None if zz is None else zz[:n_s])
        # global generators as after the reference's last per-seed call (student pass of the last seed)
        last = seeds[-1]
        torch.manual_seed(last)
        np.random.seed(last)
        for _ in range(1 + n_s):
            torch.randn(1, *shape)
        return [(self._as_pairs(t_states, n_t, t_idx, i, shape), self._as_pairs(s_states, n_s, s_idx, i, shape))
                for i in range(len(seeds))]

    def _student_size(self):
        return getattr(self.student_model, "image_size", self.config.image_size)

    def _resize_student(self, student_trajectory, size):
        """reference :153-163: every stored student state resized to the teacher's resolution (bilinear, align_corners=True);
        one dt_resize_bilinear launch over the whole trajectory."""
        if size == self.config.image_size or not student_trajectory:
            return student_trajectory
        imgs = torch.cat([img for img, _ in student_trajectory])
        out = engine.resize_bilinear(imgs, (self.config.image_size, self.config.image_size))
        return [(out[i:i + 1].clone(), t) for i, (_, t) in enumerate(student_trajectory)]

    def generate_trajectory(self, seed=None):
        """One (teacher, student) pair from seeded noise (reference :65-165)."""
        cfg = self.config
        if seed is not None:
            torch.manual_seed(seed)
            np.random.seed(seed)
        self.teacher_model.eval()
        self.student_model.eval()
        x_teacher = torch.randn(1, cfg.channels, cfg.image_size, cfg.image_size)
        teacher_trajectory = self._run(self.teacher_model, x_teacher, timestep_list(cfg.sample_steps, cfg.teacher_steps))
        if seed is not None:
            torch.manual_seed(seed)
            np.random.seed(seed)
        size = self._student_size()
        x_student = torch.randn(1, cfg.channels, size, size)
        student_trajectory = self._run(self.student_model, x_student, timestep_list(cfg.sample_steps, cfg.student_steps))
        return teacher_trajectory, self._resize_student(student_trajectory, size)

    def _update_x(self, x, noise_pred, t, noise):
        """Reference's placeholder update (:167-205) as one fused kernel launch."""
        if noise_pred.shape != x.shape:
            noise_pred = engine.resize_bilinear(noise_pred, x.shape[2:])
        coef = self._manager_coefficients([t])[0]
        return engine.cfg_update(RULE_MANAGER, x.contiguous().float(), noise_pred.contiguous().float(), None,
                                 noise.contiguous().float(), coef, True)

    def generate_trajectory_from_sample(self, sample, seed=None):
        """Pair starting from a given noise sample (reference :265-387); ([], []) / (teacher, []) on failure."""
        cfg = self.config
        if seed is not None:
            torch.manual_seed(seed)
            np.random.seed(seed)
        self.teacher_model.eval()
        self.student_model.eval()
        sample = self._ensure_tensor_compatibility(sample)
        print(f"Sample shape for teacher: {sample.shape}")
        try:
            teacher_trajectory = self._run(self.teacher_model, sample.detach().cpu().float(),
                                           timestep_list(cfg.sample_steps, cfg.teacher_steps))
        except Exception as e:
            print(f"Error generating teacher trajectory: {e}")
            return [], []
        size = self._student_size()
        start = sample.detach().cpu().float()
        if sample.shape[2] != size or sample.shape[3] != size:
            start = engine.resize_bilinear(start, (size, size)).cpu()
        print(f"Sample shape for student: {start.shape}")
        try:
            student_trajectory = self._run(self.student_model, start, timestep_list(cfg.sample_steps, cfg.student_steps))
        except Exception as e:
            print(f"Error generating student trajectory: {e}")
            return teacher_trajectory, []
        return teacher_trajectory, self._resize_student(student_trajectory, size)

    # ------------------------------------------------------------------ persistence
    def _path(self, i, size_factor=None):
        sf = self.size_factor if size_factor is None else size_factor
        return os.path.join(self.config.trajectory_dir, f"trajectory_size_{sf}_sample_{i}.pkl")

    def generate_and_save_trajectories(self, num_samples=10):
        """Generate ``num_samples`` pairs and pickle each (reference :207-263); returns the file paths."""
        file_paths = []
        use_fixed = self.fixed_samples is not None and num_samples <= len(self.fixed_samples)
        if use_fixed:
            print(f"Using {num_samples} fixed samples for consistent comparison")
        batched = None
        if not use_fixed:
            try:      # all seeds in one device-resident batch; per-seed fallback keeps the reference's error behaviour
                batched = self.generate_trajectories_batched(list(range(num_samples)))
            except Exception as e:
                print(f"Batched trajectory generation failed ({e}); generating one by one")
        for i in range(num_samples):
            try:
                if use_fixed:
                    pair = self.generate_trajectory_from_sample(self.fixed_samples[:num_samples][i], i)
                elif batched is not None:
                    pair = batched[i]
                else:
                    pair = self.generate_trajectory(seed=i)
            except Exception as e:
                print(f"Error generating trajectory {i}{' from fixed sample' if use_fixed else ''}: {e}")
                continue
            path = self._path(i)
            with open(path, "wb") as f:
                pickle.dump(pair, f)
            file_paths.append(path)
        return file_paths

    def _files(self, size_factor):
        prefix = f"trajectory_size_{size_factor}_sample_"
        files = [f for f in os.listdir(self.config.trajectory_dir) if f.startswith(prefix) and f.endswith(".pkl")]
        files.sort(key=lambda x: int(x.split("_sample_")[1].split(".")[0]))
        return files

    def load_trajectories(self, size_factor=None, indices=None):
        """(teacher_trajectories, student_trajectories) from disk, sorted by sample index (reference :389-432).

        The files are this class's own pickles of (list[(Tensor, int)], list[(Tensor, int)])."""
        size_factor = self.size_factor if size_factor is None else size_factor
        files = self._files(size_factor)
        if indices is not None:
            files = [f for f in files if int(f.split("_sample_")[1].split(".")[0]) in indices]
        teachers, students = [], []
        for name in files:
            with open(os.path.join(self.config.trajectory_dir, name), "rb") as f:
                t_traj, s_traj = pickle.load(f)
            teachers.append(t_traj)
            students.append(s_traj)
        return teachers, students

    def compute_trajectory_metrics_batch(self, size_factor=None, batch_size=10):
        """Metric lists + ``_avg`` keys over all stored pairs (reference :434-548)."""
        size_factor = self.size_factor if size_factor is None else size_factor
        files = self._files(size_factor)
        all_metrics = {k: [] for k in (
            "wasserstein_distances", "wasserstein_distances_per_timestep", "endpoint_distances", "teacher_path_lengths",
            "student_path_lengths", "teacher_efficiency", "student_efficiency", "path_length_similarity",
            "efficiency_similarity", "mean_velocity_similarity", "mean_directional_consistency",
            "mean_position_difference", "distribution_similarity", "architecture_type")}
        renamed = (("wasserstein_distances", "mean_wasserstein"),
                   ("wasserstein_distances_per_timestep", "wasserstein_distances"),
                   ("endpoint_distances", "endpoint_distance"), ("teacher_path_lengths", "teacher_path_length"),
                   ("student_path_lengths", "student_path_length"), ("teacher_efficiency", "teacher_efficiency"),
                   ("student_efficiency", "student_efficiency"))
        same = ("path_length_similarity", "efficiency_similarity", "mean_velocity_similarity",
                "mean_directional_consistency", "mean_position_difference", "distribution_similarity")
        for start in range(0, len(files), batch_size):
            pairs = []
            for name in files[start:start + batch_size]:
                with open(os.path.join(self.config.trajectory_dir, name), "rb") as f:
                    pairs.append(pickle.load(f))
            # the pairs of a batch are reduced together: one launch of each metric kernel and one device-to-host copy per
            # group of equal trajectory lengths instead of three launches and three syncs per pair
            for metrics in compute_trajectory_metrics_many(pairs, self.config):
                for dst, src in renamed:
                    all_metrics[dst].append(metrics[src])
                for k in same:
                    if k in metrics:
                        all_metrics[k].append(metrics[k])
                if hasattr(self, "architecture_type"):
                    all_metrics["architecture_type"].append(self.architecture_type)
        for k in ("endpoint_distances", "teacher_path_lengths", "student_path_lengths", "teacher_efficiency",
                  "student_efficiency", "wasserstein_distances") + same:
            if all_metrics.get(k):
                all_metrics[k + "_avg"] = sum(all_metrics[k]) / len(all_metrics[k])
        return all_metrics


def generate_trajectories_with_disk_storage(teacher_model, student_model, config, size_factor=1.0, num_samples=10,
                                            fixed_samples=None):
    """Create the manager and top the on-disk pairs up to ``num_samples`` (reference :550-581)."""
    manager = TrajectoryManager(teacher_model, student_model, config, size_factor, fixed_samples)
    prefix = f"trajectory_size_{size_factor}_sample_"
    existing = [f for f in os.listdir(config.trajectory_dir) if f.startswith(prefix) and f.endswith(".pkl")]
    if len(existing) < num_samples:
        print(f"Generating {num_samples - len(existing)} new trajectories...")
        manager.generate_and_save_trajectories(num_samples - len(existing))
    else:
        print(f"Using {num_samples} existing trajectories...")
    return manager
