"""Deterministic synthetic gravity problems: uniform 100 m cells, a lattice of observations 1 m above
the surface (never on a cell face), a 300 kg/m3 block.  Used by bench.py, smoke() and the parity tests."""
import numpy as np


def grid(nx, ny, nz, h=100.0):
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    i, j, k = i.ravel().astype(np.float64), j.ravel().astype(np.float64), k.ravel().astype(np.float64)   # i fastest
    return i * h, (i + 1) * h, j * h, (j + 1) * h, k * h, (k + 1) * h


def observations(nx, ny, ox, oy, h=100.0):
    a, b = np.meshgrid(np.arange(ox), np.arange(oy), indexing="xy")
    xs = (a.ravel() + 0.5) * nx * h / ox + 0.37
    ys = (b.ravel() + 0.5) * ny * h / oy + 0.41
    return xs, ys, np.full(xs.size, -1.0)


def true_model(nx, ny, nz, rho=300.0):
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    inside = (k >= nz // 4) & (k < nz // 2) & (j >= ny // 3) & (j < 2 * ny // 3) & (i >= nx // 3) & (i < 2 * nx // 3)
    return np.where(inside, rho, 0.0)
