"""Build the native libraries of the package in-tree (no JIT cache).

  lib/libsvslam_hip.so      HIP kernels + C ABI (include/svslam.h), gfx950
  lib/libsvslam_synth.so    CPU generator of the synthetic stereo stream
  lib/libsvslam_pipeline.so C++ host pipeline (Frontend/Backend/Map mirror of the
                            reference) bound to libsvslam_hip.so

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(HERE, "lib")
ROOT = os.path.dirname(HERE)

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
             "-Wno-unused-value"]
# development: extra compiler flags for an A/B variant (tools/ab.sh), output directory of the variant
HIP_FLAGS += os.environ.get("SVS_EXTRA_HIP_FLAGS", "").split()
if os.environ.get("SVS_LIB_DIR"):
    LIB = os.path.abspath(os.environ["SVS_LIB_DIR"])
if os.environ.get("SVS_IEEE_DIV"):      # parity-debugging build: IEEE division / sqrt instead of estimate + Newton in the LM kernels
    HIP_FLAGS.append("-DSVS_IEEE_DIV")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(d, exts):
    out = []
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(root, f))
    return out


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def source_hash():
    """sha256 (first 12 hex digits) over the sources and flags libsvslam_hip.so is compiled from: svslam_build_info() carries
    it, the PMC summaries under profiles/ are stamped with it and bench.py says whether the stamp is the loaded library's"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(_sources(CSRC, (".hip", ".h")) + [os.path.join(ROOT, "include", "svslam.h")]):
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    h.update(" ".join(HIP_FLAGS).encode())
    return h.hexdigest()[:12]


def build_hip(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libsvslam_hip.so")
    srcs = _sources(CSRC, (".hip", ".h")) + [os.path.join(ROOT, "include", "svslam.h")]
    if force or _newer(out, srcs):
        _run([HIPCC] + HIP_FLAGS + ['-DSVS_SRC_HASH="%s"' % source_hash(),
                                     os.path.join(CSRC, "svslam_hip.hip"), os.path.join(CSRC, "synth.hip"),
                                     "-o", out], verbose)
    return out


def build_synth(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libsvslam_synth.so")
    srcs = [os.path.join(CSRC, f) for f in ("synth_cpu.c", "synth_scene.h", "synth_traj.h")]
    if force or _newer(out, srcs):
        _run(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu99", srcs[0], "-o", out, "-lm"], verbose)
    return out


def build_pipeline(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libsvslam_pipeline.so")
    main = os.path.join(HOST, "pipeline_capi.cpp")
    if not os.path.exists(main):
        return None
    srcs = _sources(HOST, (".cpp", ".h")) + [os.path.join(ROOT, "include", "svslam.h")]
    hip_so = build_hip(force=False, verbose=verbose)
    if force or _newer(out, srcs + [hip_so]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"),
              "-I", CSRC, main, "-o", out, "-L", LIB, "-lsvslam_hip", "-Wl,-rpath,$ORIGIN"], verbose)
    return out


def build_all(force=False, verbose=False):
    return [build_hip(force, verbose), build_synth(force, verbose), build_pipeline(force, verbose)]


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose=True):
        print("built", p)
