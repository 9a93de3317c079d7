"""Build libsvc_hip.so (gfx950 only) with hipcc.  Incremental: a .hip is recompiled when it or a header is newer
than its object.  Usage: python so-vits-svc_amd/csrc/build.py [--force] [--asm]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libsvc_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result", "-ffp-contract=off"]


def _sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "svc_hip.h"))
    return hs


def source_hash():
    """sha256 over the kernel sources and headers (sorted by name): identifies the build a measurement was taken on — PMC
    summaries are stamped with it and bench.py refuses to quote one taken on other sources."""
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(HERE, s) for s in _sources()] + sorted(_headers()):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=True, save_temps=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = max(os.path.getmtime(h) for h in _headers())
    jobs, objs = [], []
    for src in _sources():
        s = os.path.join(HERE, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if save_temps:
                cmd += ["-save-temps=obj"]
            jobs.append((src, cmd))

    def run(job):
        src, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
        return src, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for src, r in ex.map(run, jobs):
                if verbose and (r.stderr.strip() or r.returncode):
                    sys.stderr.write(r.stderr)
                if r.returncode:
                    raise RuntimeError(f"hipcc failed on {src}")
                if verbose:
                    print(f"[build] compiled {src}")
    need_link = bool(jobs) or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs)
    if need_link:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[build] linked {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--asm" in sys.argv)
