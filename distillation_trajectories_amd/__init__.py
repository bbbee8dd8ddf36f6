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


# reference top-level module name -> module of this package that stands in for it
_ALIASES = {
    "models": "models",
    "config.config": "config",
    "utils.diffusion": "utils.diffusion",
    "utils.trajectory_manager": "utils.trajectory_manager",
    "utils.metric_transformations": "utils.metric_transformations",
    "analysis.trajectory_engine": "analysis.trajectory_engine",
    "analysis.metrics.trajectory_metrics": "analysis.metrics.trajectory_metrics",
    "analysis.metrics.time_dependent": "analysis.metrics.time_dependent",
    "analysis.metrics.fid_score": "analysis.metrics.fid_score",
    "analysis.noise_prediction.noise_analysis": "analysis.noise_prediction.noise_analysis",
    "evaluation.metrics": "evaluation.metrics",
}


def install_aliases(force=False):
    """Register this package's modules in ``sys.modules`` under the reference's top-level names, so that the
    reference's callers run unchanged (``from config.config import Config``, ``from models import DiffusionUNet``,
    ``from analysis.trajectory_engine import compare_trajectories`` ... as in
    scripts/analysis/analyze_trajectory_metrics.py:22-26 and analyze_trajectories.py:20-23).

    Call it before the caller's imports run.  A name that is already imported from somewhere else (the reference's
    own module, say) is an error unless ``force`` is given, because a process that mixes both would be confusing.
    Returns the list of names registered; ``remove_aliases()`` undoes it.
    """
    import importlib
    import sys
    import types
    done = []
    for ref_name, own in _ALIASES.items():
        target = importlib.import_module(f"{__name__}.{own}")
        parts = ref_name.split(".")
        for depth in range(1, len(parts)):                 # parent packages: plain namespace shells
            pkg = ".".join(parts[:depth])
            cur = sys.modules.get(pkg)
            if cur is None or (force and not getattr(cur, "__dt_alias__", False)):
                shell = types.ModuleType(pkg)
                shell.__path__ = []                        # a package, but with nothing importable from disk
                shell.__dt_alias__ = True
                sys.modules[pkg] = shell
                done.append(pkg)
                if depth > 1:
                    setattr(sys.modules[".".join(parts[:depth - 1])], parts[depth - 1], shell)
            elif not getattr(cur, "__dt_alias__", False):
                raise ImportError(f"install_aliases: '{pkg}' is already imported from {getattr(cur, '__file__', '?')}; "
                                  "install the aliases before the reference's modules are imported (or pass force=True)")
        cur = sys.modules.get(ref_name)
        if cur is not None and cur is not target and not force:
            raise ImportError(f"install_aliases: '{ref_name}' is already imported from {getattr(cur, '__file__', '?')}")
        sys.modules[ref_name] = target
        if len(parts) > 1:
            setattr(sys.modules[".".join(parts[:-1])], parts[-1], target)
        done.append(ref_name)
    return done


def remove_aliases():
    """Drop every name ``install_aliases`` registered (test hygiene)."""
    import sys
    own = {f"{__name__}.{v}" for v in _ALIASES.values()}
    for name in list(sys.modules):
        mod = sys.modules[name]
        top = name.split(".")[0]
        if top in {k.split(".")[0] for k in _ALIASES} and (getattr(mod, "__dt_alias__", False) or getattr(mod, "__name__", "") in own):
            del sys.modules[name]
