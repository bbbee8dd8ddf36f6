"""Heat-map normalisation of the four headline metrics (reference utils/metric_transformations.py:3-38).
Pure host scalar math, kept so that the analysis scripts find the same function at the same place."""
import numpy as np


def transform_metrics(path_length_similarity, trajectory_mse, directional_consistency, distribution_similarity):
    """Map raw metric values to [0, 1] scores; keys match the heat-map code of the reference."""
    unit = np.log1p(1.0)
    mse_score = np.clip(1 - np.log1p(np.clip(trajectory_mse, 0, None)) / unit, 0, 1)
    dist_score = np.clip(np.log1p(distribution_similarity) / unit, 0, 1)
    return {
        "path_length_similarity": path_length_similarity,
        "trajectory_mse": mse_score,
        "mean_directional_consistency": np.abs(directional_consistency),
        "distribution_similarity": dist_score,
    }
