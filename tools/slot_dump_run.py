import os, sys
sys.path.insert(0, "3d-reconstruction-from-multi-view-exp_amd"); sys.path.insert(0, ".")
from lib.synthetic import make_scene
from lib.bundle_adjustment import BundleAdjuster
sc = make_scene(1_000_000, 100, vis_p=0.1)
ba = BundleAdjuster.from_observations(sc.n_points, 100, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t, axis=sc.axis)
print("created")
