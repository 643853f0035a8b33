import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench as B
dev = torch.device("cuda")
t0 = time.time()
log = lambda m: print(f"[{time.time() - t0:6.1f}] {m}", flush=True)
model = B.build_model(dev)
out = B.recons_leg(model, dev, 8, log)
for k in ("full_batch", "config5_full_batch"):
    d = out[k]
    print(k, json.dumps({kk: vv for kk, vv in d.items() if kk not in ("note", "reference_arithmetic", "synthetic_prior")}, indent=None)[:1500])
print({k: v for k, v in out.items() if not isinstance(v, (dict, list))})
