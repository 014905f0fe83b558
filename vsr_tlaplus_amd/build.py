"""Builds libvsrmc.so (HIP, gfx950) and the `vsrmc` CLI in-tree with hipcc.  No GPU is needed to build."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvsrmc.so")
LIB_HOOKS = os.path.join(HERE, "libvsrmc_hooks.so")     # the same sources with -DVSRMC_TEST_HOOKS: test hooks the product library does not contain
CLI = os.path.join(HERE, "vsrmc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _newer(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def sources():
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".cpp"))]
    out.append(os.path.join(os.path.dirname(HERE), "include", "vsrmc.h"))
    return out


def build(force=False, verbose=False):
    srcs = sources()
    if force or _newer(LIB, srcs):
        cmd = [HIPCC] + FLAGS + ["-shared", "-o", LIB, os.path.join(CSRC, "vsrmc.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    if force or _newer(LIB_HOOKS, srcs):
        cmd = [HIPCC] + FLAGS + ["-DVSRMC_TEST_HOOKS", "-shared", "-o", LIB_HOOKS, os.path.join(CSRC, "vsrmc.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    cli_src = os.path.join(CSRC, "vsrmc_cli.cpp")
    if os.path.exists(cli_src) and (force or _newer(CLI, srcs + [LIB])):
        cmd = [HIPCC, "-O2", "-std=c++17", "-o", CLI, cli_src, "-L" + HERE, "-lvsrmc", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
