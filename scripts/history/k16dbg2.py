import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine
blob = weights.synthetic_blob(0)
f = frames.synthetic_frame(0)
for slots in (1, 4):
    with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=slots) as e:
        for s in range(slots): e.get_src_image_buffer(s)[:] = f
        e.write_head(np.full((8400, 86), 7.0, np.float32), 0)
        e.submit(0, slots); e.wait()
        h = e.read_head(0)
        print(slots, "after batched submit: kpt==7 count", int((h[:, 78:] == 7.0).sum()), "of", h[:, 78:].size, "box==7", int((h[:, :64] == 7.0).sum()), "cls==7", int((h[:, 64:78] == 7.0).sum()))
        e.write_head(np.full((8400, 86), 7.0, np.float32), 0)
        e.detect(0)
        h = e.read_head(0)
        print(slots, "after detect: kpt==7 count", int((h[:, 78:] == 7.0).sum()), "per level", [int((h[a:b, 78:] == 7.0).sum()) for a, b in ((0, 6400), (6400, 8000), (8000, 8400))])
        print("   row0", h[0, 78:], "row 6400", h[6400, 78:])
