# which of the pointwise / direct 1x1 kernels reproduces model.6.cv1 = fp16(S * silu(W a + b)) computed in float64 from the layer's own input?
import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine
blob = weights.synthetic_blob(0)
hdr, layers = weights.parse_blob(blob)
L = {sp.name: (sp, w, b) for sp, w, b in layers}
S = 1.44269504088896341
img = frames.synthetic_frame(42)
res = {}
for mode in ("IRMV_FORCE_PW", "IRMV_NO_PW"):
    for m in ("IRMV_FORCE_PW", "IRMV_FORCE_PWN", "IRMV_NO_PW", "IRMV_NO_PWN"):
        os.environ.pop(m, None)
    os.environ[mode] = "1"
    os.environ["IRMV_NO_PWN"] = "1"
    with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=3, num_streams=1) as e:
        for s in range(3):
            e.get_src_image_buffer(s)[:] = img
        e.submit(0, 3); e.wait()
        x = e.read_tap("5", 2).astype(np.float64)            # unscaled by the read-back: a = a' * ln 2 (float32 product)
        cat = e.read_tap("model.6.cat", 2).astype(np.float64)
        res[mode] = (x, cat)
sp, w, b = L["model.6.cv1"]
x = res["IRMV_FORCE_PW"][0]
assert np.array_equal(x, res["IRMV_NO_PW"][0])
xs = x * S                                                    # the stored scaled activations (to ~1e-7 relative)
y = xs.reshape(-1, sp.cin) @ w.reshape(sp.cout, sp.cin).astype(np.float64).T + b.astype(np.float64) * S
ref = (y / (1 + np.exp2(-y))).reshape(x.shape[0], x.shape[1], sp.cout)   # scaled output, exact
for mode in res:
    got = res[mode][1][:, :, :sp.cout] * S                   # back to the stored scale
    ulp = np.maximum(2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - 10), 2.0 ** -24)
    err = np.abs(got - ref) / ulp
    print(f"{mode:14s} model.6.cv1: max err {err.max():.3f} fp16 ulp, mean {err.mean():.4f}, > 0.51 ulp: {(err > 0.51).sum()} of {err.size}, > 1.0: {(err > 1.0).sum()}")
a, c = res["IRMV_FORCE_PW"][1][:, :, :128], res["IRMV_NO_PW"][1][:, :, :128]
print("pw vs direct on model.6.cv1 output: differ", int((a != c).sum()), "of", a.size)
