import importlib, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
tfx = importlib.import_module("tomofast-x_amd")
nx, ny, nz = 64, 64, 32
grid = tfx.synthetic.grid(nx, ny, nz)
xs, ys, zs = tfx.synthetic.observations(nx, ny, 8, 8)
free0 = None
for it in range(12):
    ctx = tfx.Context(0)
    ctx.set_grid(nx, ny, nz, *grid)
    cw = ctx.calculate_depth_weight()
    ctx.debug_set("band_select_min_cells", 0)
    ctx.calculate_sensit(xs, ys, zs, cw, 2, 0.05)
    ctx.profile_enable(True)
    b = np.random.default_rng(0).standard_normal(xs.size)
    ctx.lsqr_solve_sensit(b, 20, 1e-13, 0.0, 0.0, [np.full(nx * ny * nz, 1e-3, np.float32)], [np.zeros(nx * ny * nz)])
    ctx.profile_get(0)
    ctx.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(0)
    if it == 1:
        free0 = free
    print(it, "free MB", free // (1 << 20))
assert free0 - free < 64 << 20, (free0, free)
print("no leak")
