import sys, importlib, json, time
sys.path.insert(0, "/root/repo")
import bench
tfx = importlib.import_module("tomofast-x_amd")
t0=time.time()
out = bench.cpu_baseline_large(tfx, 19894211024, 9.53, 26800000, lambda m: print("[log]", m, flush=True), wall_budget_s=60.0)
print(json.dumps(out, indent=1)); print("total", time.time()-t0)
