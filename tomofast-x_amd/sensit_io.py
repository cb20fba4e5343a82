"""Reference-compatible SENSIT files: the kernel checkpoint that `sensit.readFromFiles = 1` re-uses
(src/problem_joint_gravmag.F90:172-202).  Big-endian streams, as written by the reference's
`-fconvert=big-endian` build (Makefile:51):

  sensit_{grav|magn}_{nbproc}_{rank}   header 5 x int32: ndata_loc, ndata, nelements_total, myrank, nbproc
                                       per row 4 x int32: idata, nel, model_component, data_component; int32 cols[nel]
                                       (1-based, ascending); float32 vals[nel]      (sensitivity_gravmag.F90:183, :306-309)
  sensit_{grav|magn}_meta.txt          5 text lines                                 (:360-375)
  sensit_{grav|magn}_nnz               int32 N, int32 nnz[N]                        (:380-392)
  sensit_{grav|magn}_weight            int32 N, float64 weight[N]                   (:415-464)

Host-side file I/O on top of Context.matrix_download_csr / matrix_upload_csr; no GPU work here."""
import os

import numpy as np

SUFFIX = {1: "grav", 2: "magn"}


def write_sensit(folder, problem_type, ctx_csr, nelements_total, grid_dims, column_weight, compression_type, comp_error,
                 depth_weighting_type=1, nbproc=1, rank=0, row_begin=0, ndata_total=None, ndata_components=1,
                 nmodel_components=1, problem_weight=1.0, data_weight=None, nnz_hist_total=None, nnz_total=None):
    """ctx_csr = (rowptr, cols, vals) of the matrix rows of the data [row_begin, row_begin + ndata_loc) over ALL columns
    (1-based cols).  With several components the matrix row idata*ndata_components + d holds model component k in columns
    k*nelements_total + cell; the file stores one line per (idata, d, k) with cell columns (sensitivity_gravmag.F90:222-311).
    The files hold the UNSCALED kernel (the reference scales by problem_weight * data_weight on reload, :834-843): a matrix that was
    built scaled must come with the factors it was built with, and is refused otherwise; the hosts build unscaled and scale
    afterwards (Context.matrix_scale_rows), which keeps the files bit-identical to the reference's.
    Several writer ranks (nbproc > 1): every rank passes its own rows; rank 0 also needs the per-cell counts and the entry count of
    the WHOLE kernel (nnz_hist_total, nnz_total = the all-reduced values) for the metadata files."""
    rp, cols, vals = ctx_csr
    if problem_weight != 1.0 or data_weight is not None:
        raise ValueError("write_sensit: pass the unscaled kernel (build with problem_weight 1 and no data weights, scale the device "
                         "matrix afterwards with matrix_scale_rows) - dividing the factors out again is not bit-exact in fp32")
    if nbproc > 1 and rank == 0 and (nnz_hist_total is None or nnz_total is None):
        raise ValueError("write_sensit: with several writer ranks rank 0 needs nnz_hist_total and nnz_total of the whole kernel")
    ncd, ncm, N = int(ndata_components), int(nmodel_components), int(nelements_total)
    nrows = rp.size - 1
    assert nrows % ncd == 0
    ndata_loc = nrows // ncd
    ndata_total = ndata_loc if ndata_total is None else ndata_total
    sfx = SUFFIX[problem_type]
    os.makedirs(folder, exist_ok=True)
    cols = np.asarray(cols, np.int64)
    with open(os.path.join(folder, "sensit_%s_%d_%d" % (sfx, nbproc, rank)), "wb") as f:
        f.write(np.array([ndata_loc, ndata_total, N, rank, nbproc], ">i4").tobytes())
        for r in range(nrows):
            a, b = int(rp[r]), int(rp[r + 1])
            comp = (cols[a:b] - 1) // N
            for k in range(ncm):
                sel = np.nonzero(comp == k)[0] + a
                f.write(np.array([row_begin + r // ncd + 1, sel.size, k + 1, r % ncd + 1], ">i4").tobytes())
                if sel.size:
                    f.write((cols[sel] - k * N).astype(">i4").tobytes())
                    f.write(np.asarray(vals)[sel].astype(">f4").tobytes())
    if rank == 0:
        nx, ny, nz = grid_dims
        hist = (np.bincount((cols - 1) % N, minlength=N) if nnz_hist_total is None else np.asarray(nnz_hist_total)).astype(np.int32)
        with open(os.path.join(folder, "sensit_%s_meta.txt" % sfx), "w") as f:
            f.write(" %d %d %d %d\n" % (nx, ny, nz, ndata_total))
            f.write(" %d %d %d\n" % (nbproc, 4, depth_weighting_type))
            f.write(" %d %.17g\n" % (compression_type, comp_error))
            f.write(" %d %d\n" % (ncm, ncd))
            f.write(" %d\n" % int(rp[-1] if nnz_total is None else nnz_total))
        with open(os.path.join(folder, "sensit_%s_nnz" % sfx), "wb") as f:
            f.write(np.array([nelements_total], ">i4").tobytes() + hist.astype(">i4").tobytes())
        with open(os.path.join(folder, "sensit_%s_weight" % sfx), "wb") as f:
            f.write(np.array([nelements_total], ">i4").tobytes() + np.asarray(column_weight, ">f8").tobytes())


def read_sensit(folder, problem_type):
    """Reads every rank file of a SENSIT set (any nbproc, like the reference: sensitivity_gravmag.F90:1016-1030).
    Returns dict(rowptr, cols, vals, nnz_hist, column_weight, meta)."""
    sfx = SUFFIX[problem_type]
    meta_txt = open(os.path.join(folder, "sensit_%s_meta.txt" % sfx)).read().split()
    nx, ny, nz, ndata = [int(v) for v in meta_txt[:4]]
    nbproc, precision, dw_type = [int(v) for v in meta_txt[4:7]]
    ctype, comp_error = int(meta_txt[7]), float(meta_txt[8])
    ncm, ncd = int(meta_txt[9]), int(meta_txt[10])
    nnz_total = int(meta_txt[11])
    N = nx * ny * nz
    if precision != 4:
        raise ValueError("SENSIT matrix precision %d is not the 4-byte real this path stores" % precision)
    rows = {}
    for rank in range(nbproc):
        raw = open(os.path.join(folder, "sensit_%s_%d_%d" % (sfx, nbproc, rank)), "rb").read()
        hdr = np.frombuffer(raw, ">i4", 5, 0)
        if int(hdr[1]) != ndata or int(hdr[2]) != nx * ny * nz or int(hdr[4]) != nbproc:
            raise ValueError("SENSIT file header is inconsistent with the metadata")
        off = 20
        for _ in range(int(hdr[0]) * ncd * ncm):
            idata, nel, k, d = [int(v) for v in np.frombuffer(raw, ">i4", 4, off)]
            off += 16
            c = np.frombuffer(raw, ">i4", nel, off).astype(np.int32)
            off += 4 * nel
            v = np.frombuffer(raw, ">f4", nel, off).astype(np.float32)
            off += 4 * nel
            # matrix row (idata, d); model component k in columns (k-1)*N + cell (sensitivity_gravmag.F90:829-846)
            key = (idata - 1) * ncd + d
            pc, pv = rows.get(key, (np.zeros(0, np.int32), np.zeros(0, np.float32)))
            rows[key] = (np.concatenate([pc, c + (k - 1) * N]).astype(np.int32), np.concatenate([pv, v]))
        if off != len(raw):
            raise ValueError("trailing bytes in SENSIT file")
    if sorted(rows) != list(range(1, ndata * ncd + 1)):
        raise ValueError("SENSIT row set is incomplete")
    order = [rows[i] for i in range(1, ndata * ncd + 1)]
    rowptr = np.concatenate([[0], np.cumsum([c.size for c, _ in order])]).astype(np.int64)
    if int(rowptr[-1]) != nnz_total:
        raise ValueError("nnz_total in the metadata differs from the rows read")
    z = open(os.path.join(folder, "sensit_%s_nnz" % sfx), "rb").read()
    w = open(os.path.join(folder, "sensit_%s_weight" % sfx), "rb").read()
    return dict(rowptr=rowptr, cols=np.concatenate([c for c, _ in order]), vals=np.concatenate([v for _, v in order]),
                nnz_hist=np.frombuffer(z, ">i4", offset=4).astype(np.int32),
                column_weight=np.frombuffer(w, ">f8", offset=4).astype(np.float64),
                meta=dict(nx=nx, ny=ny, nz=nz, ndata=ndata, nbproc=nbproc, depth_weighting_type=dw_type,
                          nmodel_components=ncm, ndata_components=ncd,
                          compression_type=ctype, comp_error=comp_error, nnz_total=nnz_total))
