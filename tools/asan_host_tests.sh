#!/bin/bash
# Host code of libzkg.so (circuit builder, gadgets, codec, pairing / verifier, C ABI glue) under AddressSanitizer on the CPU: builds an
# instrumented copy of the library (host side only: -Xarch_host; device code is not instrumented), puts it in the package's place, runs the
# CPU tests that call into it with the ASan runtime preloaded, and restores the release build.  No GPU needed, none used.
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/zkg_asan; mkdir -p "$OUT"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
python3 - "$OUT" <<'PY'
import os, subprocess, sys
sys.path.insert(0, "zklaim_amd")
import build as zb
out = sys.argv[1]; objs = []; procs = []
flags = [f for f in zb.FLAGS if f != "-O3"] + ["-O1", "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fno-omit-frame-pointer", "-gline-tables-only"]
for src in zb.SOURCES:
    obj = os.path.join(out, src.replace(".hip", ".o")); objs.append(obj)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(zb.CSRC, src), "-o", obj]))
assert all(p.wait() == 0 for p in procs)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", "-o", os.path.join(out, "libzkg.so"), *objs])
PY
cp -p zklaim_amd/libzkg.so "$OUT/libzkg_release.so"
trap 'cp -p "$OUT/libzkg_release.so" zklaim_amd/libzkg.so' EXIT
cp "$OUT/libzkg.so" zklaim_amd/libzkg.so
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python3 -m pytest tests/test_zklaim_circuit.py tests/test_verifier.py tests/test_abi.py tests/test_pk_blob_host.py -x -q
