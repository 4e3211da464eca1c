"""exit-code probe: an LM run on a (loopback-)sharded or plain context, then a clean exit -- `python tools/exp_crash.py W F n_devices max_iter; echo $?`"""
import sys, numpy as np
sys.path.insert(0, '.')
from balm_amd import capi, scene
W, F, nd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sc = scene.generate(5, W, F, 6, mode=1)
c = capi.Context(W, 0, capi.FLAG_LOOPBACK_SHARDS, n_devices=nd) if nd > 0 else capi.Context(W)
c.set_features(sc.clusters, None, sc.coeffs)
out, lg = c.damping_iter(sc.poses_init, form=0, u0=0.01, max_iter=int(sys.argv[4]))
c.close()
print("done", W, F, nd, len(lg), flush=True)
