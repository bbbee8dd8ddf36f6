"""Host-side configuration object for the MI355X trajectory hot path.

Mirrors the attribute names of the reference's single ``Config`` class
(reference ``config/config.py:5-95``) that the hot path reads --
``channels, image_size, timesteps, sample_steps, teacher_steps,
student_steps, beta_start, beta_end, dropout, trajectory_dir`` and the
progress-bar fields ``p_sample_loop`` looks up with ``getattr`` -- without
the reference's module-level ``torchvision`` import (``config/config.py:2-3``),
so the analysis callers can be used on a box that has no torchvision.
Dataset access (``get_test_dataset``) is out of scope.
"""
import os


class Config:
    """Field-compatible stand-in for the reference ``Config`` (same defaults)."""

    def __init__(self, base_dir=None):
        # dataset shape (reference config/config.py:9-12)
        self.dataset = "CIFAR10"
        self.image_size = 32
        self.channels = 3
        self.batch_size = 128

        # model (reference config/config.py:15-19); only ``dropout`` is read by the U-Net
        self.latent_dim = 128
        self.hidden_dims = [128, 256, 256, 256]
        self.dropout = 0.3

        # diffusion process (reference config/config.py:22-26). The code path uses
        # the *linear* beta schedule regardless of ``noise_schedule``.
        self.sample_steps = 100
        self.timesteps = 100
        self.beta_start = 1e-4
        self.beta_end = 0.02
        self.noise_schedule = "cosine"

        # directory layout (reference config/config.py:37-62), rooted at ``base_dir``
        root = os.path.abspath(base_dir) if base_dir else os.getcwd()
        self.base_dir = root
        self.output_dir = os.path.join(root, "output")
        self.results_dir = os.path.join(self.output_dir, "results")
        self.models_dir = os.path.join(self.output_dir, "models")
        self.teacher_models_dir = os.path.join(self.models_dir, "teacher")
        self.student_models_dir = os.path.join(self.models_dir, "students")
        self.data_dir = os.path.join(root, "data")
        self.trajectory_dir = os.path.join(self.data_dir, "trajectories")
        self.analysis_dir = os.path.join(self.output_dir, "analysis")
        self.metrics_dir = os.path.join(self.analysis_dir, "metrics")
        # the remaining analysis sub-directories of reference config/config.py:52-62 (names only; nothing in this
        # package writes there, the unchanged plotting callers do)
        for name in ("model_comparisons", "time_dependent", "size_dependent", "dimensionality", "latent_space",
                     "attention", "noise_prediction", "denoising", "fid"):
            setattr(self, f"{name}_dir", os.path.join(self.analysis_dir, name))

        # distillation step counts (reference config/config.py:65-70)
        self.distill = True
        self.teacher_steps = self.timesteps
        self.student_steps = self.timesteps
        self.student_size_factors = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]

        # progress bars (reference config/config.py:81-83); kept so that
        # getattr(config, 'progress_bar_leave', ...) behaves the same
        self.progress_bar_leave = False
        self.progress_bar_position = 0
        self.progress_bar_ncols = 100
        self.mps_enabled = False

    def create_directories(self):
        """Create the directories the hot path writes to (trajectory pickles)."""
        for d in (self.output_dir, self.data_dir, self.trajectory_dir):
            os.makedirs(d, exist_ok=True)
        return self
