"""Builds zklaim_amd/libzkg.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libzkg.so")
SOURCES = ["ntt.hip", "msm.hip", "prover.hip", "codec.hip", "zklaim_circuit.hip", "setup_verify.hip", "compat.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(dp, f) for dp, _, fs in os.walk(CSRC) for f in fs] + [os.path.join(HERE, "..", "include", f) for f in ("zkg.h", "zklaim_abi.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    # shared inputs (headers, generated streams, the C ABI): a change there recompiles everything; otherwise only the sources that changed
    private = {"msm_ba.inc": "msm.hip"}      # included by one source only
    shared = [os.path.join(dp, f) for dp, _, fs in os.walk(CSRC) for f in fs if not f.endswith(".hip") and f not in private] + \
             [os.path.join(HERE, "..", "include", f) for f in ("zkg.h", "zklaim_abi.h")] + [os.path.abspath(__file__)]
    shared_t = max(os.path.getmtime(f) for f in shared)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        own = [os.path.join(CSRC, src)] + [os.path.join(CSRC, f) for f, owner in private.items() if owner == src and os.path.exists(os.path.join(CSRC, f))]
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max([shared_t] + [os.path.getmtime(f) for f in own]):
            continue
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs])
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
