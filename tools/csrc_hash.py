"""Hash of the kernel sources (rust-bio_amd/csrc): profiles/*.json written by tools/pmc_summary.py carry it, and bench.py
reports a committed counter only when the sources it was collected with are the sources that run."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha(root=ROOT):
    h = hashlib.sha256()
    d = os.path.join(root, "rust-bio_amd", "csrc")
    files = sorted(f for pat in ("*.hip", "*.inc", "*.h", "*.cpp", "Makefile") for f in glob.glob(os.path.join(d, pat)))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha())
