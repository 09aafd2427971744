"""The DIAGNOSTIC twin of the product library: experiments/libcbdock_diag.so = the product sources compiled with -DCBD_DIAG
-DCBD_EXPERIMENTS plus experiments/csrc/*.hip.

The product library (confidence_bootstrapping_amd/libcbdock.so, built by __graft_entry__.build()) contains only kernels with correct
results and reads none of the diagnostic environment variables.  Everything that exists for MEASUREMENT only lives here:
  * CBD_CONV_VARIANT = 8 / 14 (phase stamps, non-temporal gathers: correct results), 9 .. 13 (timing-only bounds, WRONG results)
  * CBD_BF16_DIAG = 4 / 5 / 6 (phase clocks: correct results), bits 1 / 2 / 8 / 16 / 32 (timing-only bounds, WRONG results)
  * the bf16 role split (option `bf16_roles`, CBD_BF16_ROLES, CBD_BF16P_WGS; experiments/csrc/tp_conv_bf16p.hip): correct, slower, kept
    as the record of DESIGN.md section 5
Usage from a tool (BEFORE any engine is created):   from tools.diag_lib import use_diag_library; use_diag_library()
Command line:                                       python tools/diag_lib.py [--force]      (build only; CBD_DIAG_EXTRA_FLAGS="-DX=0 ..." adds
                                                    compile flags, e.g. to A/B a compile-time kernel switch against the product library)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT_DIR = os.path.join(ROOT, "experiments")
LIB = os.path.join(OUT_DIR, "libcbdock_diag.so")
OBJ_DIR = os.path.join(OUT_DIR, "build")


def build(force: bool = False) -> str:
    import __graft_entry__ as ge
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = [os.path.join(ge.CSRC, s) for s in ge.SOURCES]
    exp_dir = os.path.join(OUT_DIR, "csrc")
    srcs += sorted(os.path.join(exp_dir, f) for f in os.listdir(exp_dir) if f.endswith(".hip"))
    headers = [os.path.join(ge.CSRC, h) for h in os.listdir(ge.CSRC) if h.endswith(".h")] + [os.path.join(ROOT, "include", "cbdock.h"), os.path.abspath(__file__)]
    objs = []
    for s in srcs:
        o = os.path.join(OBJ_DIR, os.path.basename(s).replace(".hip", ".o"))
        if force or ge._stale(o, [s] + headers):
            subprocess.check_call([ge.HIPCC] + ge.FLAGS + ge.EXTRA_FLAGS.get(os.path.basename(s), []) +
                                  ["-DCBD_DIAG", "-DCBD_EXPERIMENTS"] + os.environ.get("CBD_DIAG_EXTRA_FLAGS", "").split() + ["-I", ge.CSRC, "-c", s, "-o", o])
        objs.append(o)
    if force or ge._stale(LIB, objs):
        subprocess.check_call([ge.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def use_diag_library(build_if_missing: bool = True) -> str:
    """Make confidence_bootstrapping_amd.engine bind the diagnostic library for the rest of this process."""
    if build_if_missing and not os.path.exists(LIB):
        build()
    from confidence_bootstrapping_amd import engine
    engine.load_library(LIB)
    return LIB


if __name__ == "__main__":
    print("diagnostic library:", build(force="--force" in sys.argv))
