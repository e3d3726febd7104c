"""Stride-2 layers of the LDS family at a 416 net: LDS kernel vs chunk-major direct kernel vs its deep variant, tap by tap."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine
blob = weights.synthetic_blob(0)
rng = np.random.default_rng(11)
img = rng.integers(0, 256, (1024, 1280, 3), dtype=np.uint8)
res = {}
for kind in ("lds", "ct", "deep"):
    os.environ["IRMV_FORCE_S2"] = kind
    # IRMV_FORCE_S2 is read once per process (static): run each kind in a child
    import subprocess
    code = f"""
import os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np
from irmv_detection_amd import weights
from irmv_detection_amd.engine import YoloEngine
img = np.random.default_rng(11).integers(0, 256, (1024, 1280, 3), dtype=np.uint8)
with YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), net_size=int(os.environ.get('NET', '416'))) as e:
    names = [(s['layer'], s['name']) for s in e.profile(0, 1) if 's2' in s['name']]
    e.get_src_image_buffer()[:] = img
    e.detect()
    np.savez('/tmp/s2_{kind}.npz', head=e.read_head(0), **{{'t' + t: e.read_tap(t, 0) for t in ('3', '5', '7', '16', '18', '19', '21')}})
print(names)
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ))
    print(kind, out.stdout.strip()[-600:], out.stderr.strip()[-300:], flush=True)
    res[kind] = dict(np.load(f"/tmp/s2_{kind}.npz"))
for k in ("ct", "deep"):
    for name in res["lds"]:
        d = np.abs(res["lds"][name] - res[k][name])
        print(f"lds vs {k}: {name:6s} differs {int((d > 0).sum()):7d} max {d.max():.5f}", flush=True)
