import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["IRMV_AUTOTUNE_VERBOSE"] = "1"
from irmv_detection_amd import weights
from irmv_detection_amd.engine import YoloEngine
e = YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), num_slots=int(os.environ.get("SLOTS", "1")))
e.close()
