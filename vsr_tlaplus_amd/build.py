"""Builds libvsrmc.so (HIP, gfx950) and the `vsrmc` CLI in-tree with hipcc.  No GPU is needed to build."""
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvsrmc.so")
LIB_HOOKS = os.path.join(HERE, "libvsrmc_hooks.so")     # the same sources with -DVSRMC_TEST_HOOKS: test hooks the product library does not contain
CLI = os.path.join(HERE, "vsrmc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -disable-machine-licm (round 6): MachineLICM hoists the materialisation of constants (position salts, masks, small integers) out of k_expand's tile loop into
# VGPRs that then live across the whole loop and get spilled — with it off the three hot instantiations compile to 107 / 119 / 120 VGPRs and NO scratch
# (were 128 + 12 / 34 / 27 spilled, reloaded in the middle of every successor's hash chain): k_expand -4.4 % (config 2), -6.4 % (README); DESIGN.md §8.5
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-disable-machine-licm"]


def _newer(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def sources():
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".cpp"))]
    out.append(os.path.join(os.path.dirname(HERE), "include", "vsrmc.h"))
    return out


def _snapshot():
    """hipcc maps the files it compiles for minutes: the sources are compiled from a private copy, so that an edit meanwhile cannot end the build with a
    bus error or produce a mixture of two states."""
    snap = tempfile.mkdtemp(prefix="vsrmc_build.")
    shutil.copytree(CSRC, os.path.join(snap, "vsr_tlaplus_amd", "csrc"))
    os.makedirs(os.path.join(snap, "include"))
    shutil.copy(os.path.join(os.path.dirname(HERE), "include", "vsrmc.h"), os.path.join(snap, "include", "vsrmc.h"))
    return snap, os.path.join(snap, "vsr_tlaplus_amd", "csrc")


def build(force=False, verbose=False):
    srcs = sources()
    need_lib = force or _newer(LIB, srcs)
    need_hooks = force or _newer(LIB_HOOKS, srcs)
    cli_src = os.path.join(CSRC, "vsrmc_cli.cpp")
    if need_lib or need_hooks:
        t_start = max(os.path.getmtime(s) for s in srcs)
        snap, csrc = _snapshot()
        try:
            procs = []
            for need, lib, extra in ((need_lib, LIB, []), (need_hooks, LIB_HOOKS, ["-DVSRMC_TEST_HOOKS"])):
                if not need:
                    continue
                cmd = [HIPCC] + FLAGS + extra + ["-shared", "-o", lib + ".tmp", os.path.join(csrc, "vsrmc.hip")]
                if verbose:
                    print(" ".join(cmd))
                procs.append((lib, cmd, subprocess.Popen(cmd)))             # the two libraries side by side
            for lib, cmd, p in procs:
                if p.wait() != 0:
                    raise subprocess.CalledProcessError(p.returncode, cmd)
            for lib, _cmd, _p in procs:
                os.replace(lib + ".tmp", lib)
                os.utime(lib, (t_start + 1, t_start + 1))                     # as old as the snapshot: an edit made DURING the build makes the library stale again
        finally:
            shutil.rmtree(snap, ignore_errors=True)
    if os.path.exists(cli_src) and (force or _newer(CLI, srcs + [LIB])):
        cmd = [HIPCC, "-O2", "-std=c++17", "-o", CLI, cli_src, "-L" + HERE, "-lvsrmc", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
