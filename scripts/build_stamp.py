"""Which build a counter file under profiles/ describes: hash of the kernel sources + flags (irmv_detection_amd/_build.py
source_hash: survives a rebuild of the same tree), hash of the library binary that was loaded, hash of the tile table in use.

    python3 scripts/build_stamp.py            -> one JSON object on stdout
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _sha(path):
    if not path or not os.path.exists(path):
        return None
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def build_stamp():
    from irmv_detection_amd import _build
    tc = os.environ.get("IRMV_TUNE_CACHE")
    return dict(src_sha256=_build.source_hash(), lib_sha256=_sha(os.environ.get("IRMV_LIB_PATH") or _build.LIB_PATH),
                tune_cache_sha256=_sha(tc), tune_cache=os.path.basename(tc) if tc else None)


if __name__ == "__main__":
    print(json.dumps(build_stamp()))
