import sys, numpy as np
sys.path[:0]=["3d-reconstruction-from-multi-view-exp_amd","."]
import bench
d=bench.svd_config5(5_000_000); print({k:(round(float(v),4) if not isinstance(v,(str,list)) else v) for k,v in d.items()})
