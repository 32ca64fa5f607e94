"""Build liblunaris_hip.so (gfx950) in-tree with hipcc.  `python -m lunaris_orion_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblunaris_hip.so")
SOURCES = ["lo_util.cpp", "lo_conv.hip", "lo_conv3.hip", "lo_wgrad3.hip", "lo_wgrad2.hip", "lo_norm.hip", "lo_edge.hip", "lo_train.hip", "lo_lowrank.hip", "lo_attn.hip", "lo_teacher.hip", "lo_api.hip"]
# lo_teacher_bwd.inc is textually included at the end of lo_teacher.hip (the teacher's full backward; same translation unit)
HEADERS = ["lo_common.h", "lo_internal.h", "lo_teacher_bwd.inc", os.path.join("..", "..", "include", "lunaris_hip.h")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for f in SOURCES:
        src = os.path.join(CSRC, f)
        obj = os.path.join(HERE, "build", f + ".o")
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    fail = False
    for f, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            fail = True
            sys.stderr.write(f"[build] {f} failed:\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if fail:
        raise RuntimeError("hipcc failed")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
