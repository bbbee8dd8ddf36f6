"""Build libdt_hip.so in-tree with hipcc for gfx950 (no torch headers, pure HIP runtime + C ABI).

``python -m distillation_trajectories_amd.csrc.build`` or ``build()`` from ``__graft_entry__``.
The library is linked against ``libamdhip64.so.7`` by SONAME only (no rpath), so inside a Python
process that has already imported torch it binds to the HIP runtime torch itself loaded and shares
its streams and allocations.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["dt_conv.hip", "dt_conv_bf16.hip", "dt_conv_strip.hip", "dt_layers.hip", "dt_update.hip", "dt_metrics.hip", "dt_fused.hip", "dt_unet.hip"]
HEADERS = ["dt_internal.h", "dt_conv_epilogue.h", "dt_update_math.h", "dt_fused.h", os.path.join("..", "..", "include", "dt_hip.h")]
LIB = os.path.join(HERE, "libdt_hip.so")
ARCH = "gfx950"


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


OBJ_DIR = os.path.join(HERE, "_build")
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def is_stale():
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return _newer(LIB, deps)


TOOLS_MARK = os.path.join(OBJ_DIR, "tools_build")


def build(force=False, verbose=False, tools=False):
    """Compile every HIP source for gfx950 (one object per source, in parallel, only what changed unless
    ``force``) and link the shared library.  Returns its path.

    ``tools=True`` (``--tools``) defines DT_TOOLS: the ablation instantiations of the convolution kernels and the
    DT_ABLATE switch of dt_unet_time_conv (timing experiments that produce WRONG results by design, tools/ablate.py,
    tools/tap_phases.py, tools/block_timeline.py).  The default library contains none of that; switching between the
    two kinds rebuilds everything."""
    if bool(tools) != os.path.exists(TOOLS_MARK):
        force = True
    if not force and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = hipcc_path()
    common = [os.path.join(HERE, f) for f in HEADERS] + [os.path.abspath(__file__)]
    if tools:
        open(TOOLS_MARK, "w").close()
    elif os.path.exists(TOOLS_MARK):
        os.remove(TOOLS_MARK)

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        path = os.path.join(HERE, src)
        if not force and not _newer(obj, [path] + common):
            return obj, None
        cmd = [hipcc] + FLAGS + (["-DDT_TOOLS"] if tools else []) + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        return obj, subprocess.run(cmd, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_one, SOURCES))
    for obj, res in results:
        if res is None:
            continue
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError(f"hipcc failed with exit code {res.returncode} on {obj}")
        if verbose and res.stderr:
            sys.stderr.write(res.stderr)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-rtlib-add-rpath", "-o", LIB] + [o for o, _ in results]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError(f"hipcc link failed with exit code {res.returncode}")
    return LIB


SAN_DRIVER = os.path.join(OBJ_DIR, "dt_host_sanitize")


def build_sanitizer_driver(verbose=False):
    """tests/host_sanitize/driver.cpp + every library source, HOST code instrumented with AddressSanitizer and
    UndefinedBehaviorSanitizer (``-Xarch_host -fsanitize=...``; device code is left alone: GPU ASan is unavailable on this
    pool), one executable csrc/_build/dt_host_sanitize.  Cross-compiles without a GPU; tests/test_host_sanitize.py runs it
    on the GPU box."""
    hipcc = hipcc_path()
    os.makedirs(OBJ_DIR, exist_ok=True)
    root = os.path.dirname(os.path.dirname(HERE))
    driver = os.path.join(root, "tests", "host_sanitize", "driver.cpp")
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [driver, os.path.abspath(__file__)]
    if not _newer(SAN_DRIVER, deps):
        return SAN_DRIVER
    san = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host", "-fno-omit-frame-pointer"]
    cmd = ([hipcc, f"--offload-arch={ARCH}", "-O1", "-g", "-std=c++17", "-Wno-unused-function", "-x", "hip"] + san +
           [os.path.join(HERE, f) for f in SOURCES] + [driver, "-o", SAN_DRIVER])
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError(f"sanitizer driver build failed with exit code {res.returncode}")
    return SAN_DRIVER


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, tools="--tools" in sys.argv))
    if "--sanitize" in sys.argv:
        print(build_sanitizer_driver(verbose=True))
