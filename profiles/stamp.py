"""Which kernel sources a counter set belongs to: one SHA-256 over the library's sources (pointcloudhookup_amd/csrc/*,
include/*.h).  profiles/summarize.py stamps every entry of traffic.json with it, bench.py compares it with the sources it
runs on and drops (and says so) every counter figure whose stamp differs - a kernel change without a re-collect must not
pair new times with old bytes."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha():
    files = sorted(glob.glob(os.path.join(ROOT, "pointcloudhookup_amd", "csrc", "*.hip"))
                   + glob.glob(os.path.join(ROOT, "pointcloudhookup_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "pointcloudhookup_amd", "csrc", "*.cpp"))
                   + glob.glob(os.path.join(ROOT, "pointcloudhookup_amd", "csrc", "Makefile"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha())
