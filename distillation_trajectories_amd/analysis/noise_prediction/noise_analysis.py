"""Teacher-vs-student noise-prediction metrics (reference analysis/noise_prediction/noise_analysis.py:11-85).

``predict_noise`` is one U-Net forward on the HIP path; ``calculate_noise_metrics`` (MSE, MAE, mean
per-sample cosine similarity) is one dt_pair_stats launch over the two prediction batches.  The dataset
driver and the plots of the reference module are out of scope (they need the image data).
"""
import numpy as np
import torch

from ... import engine


def generate_noise_samples(batch_size, channels, image_size, device):
    """reference :11-24 (CPU generator draw, then moved to ``device``)."""
    return torch.randn(batch_size, channels, image_size, image_size).to(device)


def predict_noise(model, noisy_images, timesteps, device):
    """reference :26-41: eps = model(noisy_images, timesteps) in eval mode."""
    model.eval()
    with torch.no_grad():
        return model(noisy_images, timesteps)


def calculate_noise_metrics(teacher_noise, student_noise):
    """{'mse', 'mae', 'cosine_similarity'} as python floats (reference :43-85)."""
    if teacher_noise.shape != student_noise.shape:
        print(f"  Resizing student noise from {student_noise.shape} to {teacher_noise.shape}")
        student_noise = engine.resize_bilinear(student_noise, (teacher_noise.shape[2], teacher_noise.shape[3]))
    if not teacher_noise.is_cuda:
        from ..metrics.trajectory_metrics import _metrics_device
        dev = _metrics_device()
        teacher_noise, student_noise = teacher_noise.to(dev), student_noise.to(dev)
    B = teacher_noise.shape[0]
    X = teacher_noise.detach().float().reshape(1, B, -1).contiguous()
    Y = student_noise.detach().float().reshape(1, B, -1).contiguous()
    E = X.shape[2]
    f32 = np.float32
    st = engine.device_pair_stats(X, Y)[:, 0].cpu().numpy()                  # [B,5] float64
    mse = float(f32(st[:, 0].sum()) / f32(B * E))
    mae = float(f32(st[:, 1].sum()) / f32(B * E))
    eps = f32(1e-12)                                                          # F.normalize's clamp
    nx = np.maximum(np.sqrt(st[:, 3].astype(f32)), eps)
    ny = np.maximum(np.sqrt(st[:, 4].astype(f32)), eps)
    cos = st[:, 2].astype(f32) / (nx * ny)
    return {"mse": mse, "mae": mae, "cosine_similarity": float(np.mean(cos, dtype=f32))}
