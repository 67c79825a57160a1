"""Identity of the kernel sources a measurement was taken on: sha256 over loam_livox_amd/csrc/* (sorted by name), first 16 hex digits.
tools/summarize_rocprof.py stamps it into every profiles/*.csv it writes ("# build <id> [commit <hash>]"); bench*.py compute it for the
tree they run on and say whether a committed counter summary still belongs to these kernels (roofline.traffic_is_current).  Works on
the GPU box, where there is no .git."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_id() -> str:
    h = hashlib.sha256()
    d = os.path.join(ROOT, "loam_livox_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def stamp_line() -> str:
    commit = os.environ.get("LL_GIT_COMMIT", "")
    return f"# build {build_id()}" + (f" commit {commit}" if commit else "")


def read_stamp(path: str):
    """(build id, commit) from the first comment line of a summary, (None, None) for summaries written before round 6"""
    try:
        with open(path) as f:
            first = f.readline().split()
    except OSError:
        return None, None
    if len(first) >= 3 and first[0] == "#" and first[1] == "build":
        return first[2], (first[4] if len(first) >= 5 and first[3] == "commit" else None)
    return None, None


if __name__ == "__main__":
    print(stamp_line())
