"""MI355X-native (gfx950) hot path of distillation_trajectories.

Public surface mirrors the reference's module paths:
  distillation_trajectories_amd.models                          <- models.py
  distillation_trajectories_amd.utils.diffusion                 <- utils/diffusion.py
  distillation_trajectories_amd.analysis.trajectory_engine      <- analysis/trajectory_engine.py
  distillation_trajectories_amd.analysis.metrics.trajectory_metrics
  distillation_trajectories_amd.utils.trajectory_manager
  distillation_trajectories_amd.utils.metric_transformations
Device work is done by csrc/ (hand-written HIP behind the C-ABI in include/dt_hip.h).
"""
__version__ = "0.1.0"
