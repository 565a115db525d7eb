"""Build tests/cabi/consumer.cpp -- a HIP-runtime-only consumer of libresdepth_hip.so (no PyTorch) -- against the in-tree
library.  Used by the CPU suite (does it compile and link against include/resdepth_hip.h) and the GPU suite (does it run)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build_consumer(out_dir) -> str:
    lib_dir = os.path.join(ROOT, "resdepth_amd")
    if not os.path.exists(os.path.join(lib_dir, "libresdepth_hip.so")):
        raise RuntimeError("libresdepth_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    exe = os.path.join(str(out_dir), "cabi_consumer")
    cmd = [HIPCC, "-O1", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi", "consumer.cpp"),
           "-L", lib_dir, "-lresdepth_hip", "-Wl,-rpath," + lib_dir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    return exe
