"""Builds libreinlife_hip.so (hipcc, gfx950 only) in-tree: reinlife_amd/lib/libreinlife_hip.so."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
PROFILE = bool(os.environ.get("RL_PHASE_PROFILE"))  # tuning build with in-kernel phase stamps
# The TUNING library (RL_TUNE=1 -> lib/libreinlife_hip_tune.so): the product's kernels plus the measurement switches the product does not
# carry -- rl_debug_set_run_mask (one half of every tick skipped: bench.py's tick_half / policy_half figures, tools/) and k_run<1024>
# (DESIGN.md 5.10).  Results under a mask are WRONG by design; nothing in the product path loads this library.
TUNE = bool(os.environ.get("RL_TUNE"))
# A/B builds (tuning): RL_LIB_TAG=x RL_EXTRA_HIPCC_FLAGS=-D... -> lib/libreinlife_hip_x.so next to the product; load it with REINLIFE_HIP_LIB
TAG = "_prof" if PROFILE else "_tune" if TUNE else ("_" + os.environ["RL_LIB_TAG"] if os.environ.get("RL_LIB_TAG") else "")
TUNE_LIB_PATH = os.path.join(LIB_DIR, "libreinlife_hip_tune.so")
LIB_PATH = os.path.join(LIB_DIR, "libreinlife_hip%s.so" % TAG)
SOURCES = ["rl_world.hip", "rl_run.hip", "rl_policy.hip", "rl_capi.hip"]
# (object suffix, extra flags) per source: rl_run.hip is compiled as TWO units side by side (RL_RUN_UNIT, see the file) -- one compiler for all
# of k_run's instantiations is what a forced build waits for
UNITS = {"rl_run.hip": [("", ["-DRL_RUN_UNIT=0"]), ("_all", ["-DRL_RUN_UNIT=1"])]}
HEADERS = ["rl_common.h", "rl_policy_dev.h", "rl_world_dev.h", os.path.join("..", "..", "include", "reinlife_hip.h")]
# -ffp-contract=off: the world kernels' float64 reward / fitness arithmetic must round exactly like the CPU path
# -fvisibility=hidden: the export list is include/reinlife_hip.h (its declarations sit inside a visibility push(default))
# --offload-compress: the gfx950 code objects are stored zstd-compressed in the fat binary (2.8 MB -> 0.7 MB; the runtime inflates them once, at load)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "--offload-compress", "-Wall",
         "-Wno-unused-function"]
FLAGS += os.environ.get("RL_EXTRA_HIPCC_FLAGS", "").split()


def source_hash():
    """sha256[:16] over the kernel sources and headers: stamps measurements that are only valid for the code they were taken on
    (profiles/run_traffic.json; bench.py drops a stamped figure when the sources have changed since)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES) + sorted(HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _digest(paths, extra=""):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for f in paths:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _stamp_ok(target, want):
    """A built file is current when it exists and the stamp beside it (<file>.srchash) names the digest of what it was built from.
    Content, not mtimes: prebuilt objects travel to the GPU box next to freshly copied sources, and a stale object with a newer
    mtime must not survive there."""
    try:
        with open(target + ".srchash") as fh:
            return os.path.exists(target) and fh.read().strip() == want
    except OSError:
        return False


def _write_stamp(target, digest):
    with open(target + ".srchash", "w") as fh:
        fh.write(digest + "\n")


def variant_flags():
    """The compile flags of the variant this process builds / loads (product, RL_TUNE, RL_PHASE_PROFILE)."""
    return FLAGS + (["-DRL_PHASE_PROFILE", "-DRL_TUNING", "-DRL_RUN_1024", "-DRL_RUN_256"] if PROFILE else ["-DRL_TUNING", "-DRL_RUN_1024", "-DRL_RUN_256"] if TUNE else [])


def library_is_current():
    """True when LIB_PATH exists and was built from the current sources with the current flags (the stamps build() writes)."""
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    flags = variant_flags()
    digests = [_digest([os.path.join(CSRC, src)] + hdrs, " ".join(flags + extra)) for src in SOURCES for _, extra in UNITS.get(src, [("", [])])]
    return _stamp_ok(LIB_PATH, _digest([], " ".join(digests)))


class _BuildLock:
    """One builder at a time per library directory (an flock on lib/.build.lock): N ranks that find the library stale at the same moment
    (torchrun, `bench.py --gpus N` on a tree whose sources were just edited) take turns -- the first one compiles, the others find the
    stamps current when they get the lock and do nothing.  The lock is advisory and dies with its process."""

    def __enter__(self):
        import fcntl
        os.makedirs(LIB_DIR, exist_ok=True)
        self.fh = open(os.path.join(LIB_DIR, ".build.lock"), "a+")
        fcntl.flock(self.fh, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.fh, fcntl.LOCK_UN)
        self.fh.close()
        return False


def build(force=False, verbose=False):
    """Compile what is stale and link LIB_PATH.  Safe to call from several processes at once: the whole of it runs under _BuildLock, and every
    object and the library itself are written under a temporary name and moved into place with os.replace() -- a process that dlopens
    LIB_PATH while another one rebuilds it maps either the old file or the new one, never a half-written one."""
    with _BuildLock():
        return _build_locked(force, verbose)


def _build_locked(force, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    flags = variant_flags()
    tmp_tag = ".tmp%d" % os.getpid()
    objs, digests, jobs = [], [], []
    for src in SOURCES:   # the translation units compile side by side (rl_world.hip alone takes over half a minute)
        sp = os.path.join(CSRC, src)
        for suffix, extra in UNITS.get(src, [("", [])]):
            obj = os.path.join(LIB_DIR, src.replace(".hip", suffix + TAG + ".o"))
            want = _digest([sp] + hdrs, " ".join(flags + extra))   # (every unit includes every header)
            objs.append(obj); digests.append(want)
            if force or not _stamp_ok(obj, want):
                cmd = [hipcc] + flags + extra + ["-c", sp, "-o", obj + tmp_tag]
                if verbose:
                    print(" ".join(cmd))
                jobs.append((cmd, subprocess.Popen(cmd), obj, want))
    failed = None
    for cmd, job, obj, want in jobs:   # wait for EVERY compiler before reporting the first failure (no orphaned hipcc behind an exception)
        if job.wait() != 0:
            failed = failed or subprocess.CalledProcessError(job.returncode, cmd)
            if os.path.exists(obj + tmp_tag):
                os.remove(obj + tmp_tag)
        else:
            if os.path.exists(obj + ".srchash"):
                os.remove(obj + ".srchash")
            os.replace(obj + tmp_tag, obj)
            _write_stamp(obj, want)
    if failed is not None:
        raise failed
    lib_want = _digest([], " ".join(digests))
    if force or jobs or not _stamp_ok(LIB_PATH, lib_want):
        vs = os.path.join(CSRC, "exports.map")   # global: rl_*; everything else (libstdc++ template instances, toolchain markers) local
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vs, "-o", LIB_PATH + tmp_tag] + objs
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
        except BaseException:
            if os.path.exists(LIB_PATH + tmp_tag):
                os.remove(LIB_PATH + tmp_tag)
            raise
        if os.path.exists(LIB_PATH + ".srchash"):
            os.remove(LIB_PATH + ".srchash")
        os.replace(LIB_PATH + tmp_tag, LIB_PATH)
        _write_stamp(LIB_PATH, lib_want)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
