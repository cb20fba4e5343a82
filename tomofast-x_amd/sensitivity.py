"""Host-side mirror of the reference's interface for the sensitivity-kernel hot path.

Names follow the reference (Tomofast-x): t_sparse_matrix's products (src/inversion/sparse_matrix.f90), lsqr_solve_sensit
(src/inversion/lsqr_solver2.F90), calculate_and_write_sensit / get_load_balancing_nelements
(src/forward/gravmag/sensitivity_gravmag.F90), model_calculate_data (src/inversion/model.F90),
forward_wavelet / inverse_wavelet (src/utils/wavelet_transform.F90), graviprism_z (src/forward/gravmag/grav/
gravity_field.f90), calculate_depth_weight (src/forward/gravmag/weights_gravmag.f90).  Everything executes in the
hand-written HIP kernels of libtfx.so; this module only marshals arguments."""
import ctypes as C

import numpy as np

from . import lib as L
from .lib import TfxError, check, f64, ptr


class Context:
    """One MI355X.  Owns the device-resident grid and sensitivity matrix (tfx_ctx)."""

    def __init__(self, device=0, stream=None):
        self._lib = L.load()
        h = C.c_void_p()
        check(self._lib.tfx_create(int(device), C.c_void_p(stream or 0), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.dims = None
        self._hook = None          # keeps the ctypes callbacks alive
        self._ghook = None
        self.rank, self.nranks = 0, 1

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tfx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- facts
    def device_info(self):
        name = C.create_string_buffer(256)
        mem = C.c_int64()
        cus = self._lib.tfx_device_info(self._h, name, 256, C.byref(mem))
        return dict(name=name.value.decode(), cus=cus, hbm_bytes=mem.value)

    # ---- multi-GPU hook
    def set_allreduce(self, fn, rank, nranks):
        """fn(dev_ptr:int, n:int, stream:int) -> None : in-place fp64 sum all-reduce of a device buffer."""
        if fn is None:
            self._hook = None
            check(self._lib.tfx_set_allreduce(self._h, C.cast(None, L.ALLREDUCE_FN), None, int(rank), int(nranks)))
        else:
            def tramp(user, buf, n, stream):
                try:
                    fn(int(buf or 0), int(n), int(stream or 0))
                    return 0
                except Exception as e:       # never let an exception cross the C frame
                    import sys
                    sys.stderr.write("all-reduce hook failed: %r\n" % (e,))
                    return 1
            self._hook = L.ALLREDUCE_FN(tramp)
            check(self._lib.tfx_set_allreduce(self._h, self._hook, None, int(rank), int(nranks)))
        self.rank, self.nranks = int(rank), int(nranks)

    def set_allgatherv(self, fn):
        """Companion hook with MPI_Allgatherv semantics on device buffers:
        fn(send_ptr:int, nsend:int, recv_ptr:int, counts:list, displs:list, stream:int) -> None."""
        if fn is None:
            self._ghook = None
            check(self._lib.tfx_set_allgatherv(self._h, C.cast(None, L.ALLGATHERV_FN)))
            return
        def tramp(user, send, nsend, recv, counts, displs, stream):
            try:
                nranks = self.nranks            # read at call time: the hook may be set before the rank count is known
                fn(int(send or 0), int(nsend), int(recv or 0), [int(counts[r]) for r in range(nranks)],
                   [int(displs[r]) for r in range(nranks)], int(stream or 0))
                return 0
            except Exception as e:
                import sys
                sys.stderr.write("all-gather hook failed: %r\n" % (e,))
                return 1
        self._ghook = L.ALLGATHERV_FN(tramp)
        check(self._lib.tfx_set_allgatherv(self._h, self._ghook))

    # ---- RCCL inside the library (the production multi-GPU path)
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        check(self._lib.tfx_comm_unique_id(buf))
        return buf.raw

    def comm_init_rccl(self, unique_id, rank, nranks):
        """Collective: every rank calls it with the 128 bytes rank 0 got from comm_unique_id()."""
        if len(unique_id) != 128:
            raise ValueError("unique id must be 128 bytes")
        if getattr(self, "_rccl_poisoned", False):
            raise RuntimeError("a thread that timed out inside an RCCL call of this context never returned: no further RCCL start-up on it")
        check(self._lib.tfx_comm_init_rccl(self._h, C.c_char_p(bytes(unique_id)), int(rank), int(nranks)))
        self.rank, self.nranks = int(rank), int(nranks)

    def comm_destroy(self):
        check(self._lib.tfx_comm_destroy(self._h))
        if self._hook is None:
            self.rank, self.nranks = 0, 1

    def comm_abort(self):
        """Drops a (possibly half-open) communicator without the peers' hand-shake: the failure leg of the start-up ladder."""
        check(self._lib.tfx_comm_abort(self._h))
        if self._hook is None:
            self.rank, self.nranks = 0, 1

    def comm_info(self):
        """dict(rccl_ranks, rccl_rank, rccl_device, rccl_version, librccl): what the communicator itself reports + which RCCL serves it."""
        n, r, d, v = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        path = C.create_string_buffer(1024)
        check(self._lib.tfx_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d), C.byref(v), path, 1024))
        return dict(rccl_ranks=n.value, rccl_rank=r.value, rccl_device=d.value, rccl_version=v.value, librccl=path.value.decode())

    def comm_allreduce(self, dev_buf, n, dtype="f64"):
        check(self._lib.tfx_comm_allreduce(self._h, ptr(dev_buf), C.c_int64(n), {"f64": 0, "i32": 1, "i64": 2}[dtype]))

    def comm_allgatherv(self, dev_send, dev_recv, counts, displs):
        """MPI_Allgatherv on device buffers of doubles (tfx_comm_allgatherv): counts / displs per rank, in doubles."""
        cnt = np.ascontiguousarray(counts, dtype=np.int64)
        dsp = np.ascontiguousarray(displs, dtype=np.int64)
        check(self._lib.tfx_comm_allgatherv(self._h, ptr(dev_send), ptr(dev_recv), cnt.ctypes.data_as(C.POINTER(C.c_int64)),
                                            dsp.ctypes.data_as(C.POINTER(C.c_int64))))

    def comm_group_begin(self):
        check(self._lib.tfx_comm_group_begin(self._h))

    def comm_group_end(self):
        check(self._lib.tfx_comm_group_end(self._h))

    def comm_send(self, dev_buf, nbytes, peer):
        check(self._lib.tfx_comm_send(self._h, ptr(dev_buf), C.c_int64(nbytes), int(peer)))

    def comm_recv(self, dev_buf, nbytes, peer):
        check(self._lib.tfx_comm_recv(self._h, ptr(dev_buf), C.c_int64(nbytes), int(peer)))

    def comm_barrier(self):
        check(self._lib.tfx_comm_barrier(self._h))

    # ---- grid (t_grid, src/inversion/grid.F90:30-50)
    def set_grid(self, nx, ny, nz, X1, X2, Y1, Y2, Z1, Z2):
        arrs = [f64(a) for a in (X1, X2, Y1, Y2, Z1, Z2)]
        n = nx * ny * nz
        for a in arrs:
            if a.size != n:
                raise ValueError("grid array size %d != nx*ny*nz = %d" % (a.size, n))
        check(self._lib.tfx_set_grid(self._h, nx, ny, nz, *[ptr(a) for a in arrs]))
        self.dims = (int(nx), int(ny), int(nz))
        # cell sizes along the axes of the structured grid (t_grad_grid, src/inversion/grid.F90:371-391) for host-side
        # constraint builders
        sh = (nz, ny, nx)
        self.spacing = (np.abs(arrs[1] - arrs[0]).reshape(sh)[0, 0, :].copy(), np.abs(arrs[3] - arrs[2]).reshape(sh)[0, :, 0].copy(),
                        np.abs(arrs[5] - arrs[4]).reshape(sh)[:, 0, 0].copy())

    @property
    def nelements_total(self):
        return int(np.prod(self.dims))

    # ---- calculate_depth_weight type 1 + columnWeightMultiplier
    def calculate_depth_weight(self, power=2.0, Z0=0.0, multiplier=4.0e3):
        cw = np.empty(self.nelements_total)
        check(self._lib.tfx_column_weight_type1(self._h, C.c_double(power), C.c_double(Z0), C.c_double(multiplier), ptr(cw)))
        return cw

    def calculate_distance_weight(self, Xdata, Ydata, Zdata, power=2.0, beta=1.0, multiplier=4.0e3):
        """forward.depthWeighting.type = 2 (weights_gravmag.f90:81-138)."""
        xd, yd, zd = f64(Xdata), f64(Ydata), f64(Zdata)
        cw = np.empty(self.nelements_total)
        check(self._lib.tfx_column_weight_type2(self._h, C.c_int64(xd.size), ptr(xd), ptr(yd), ptr(zd), C.c_double(power),
                                                C.c_double(beta), C.c_double(multiplier), ptr(cw)))
        return cw

    def calculate_mindist_weight(self, Xdata, Ydata, Zdata, power=2.0, multiplier=4.0e3):
        """forward.depthWeighting.type = 3 (weights_gravmag.f90:140-162): distance from the cell centre to the nearest datum."""
        xd, yd, zd = f64(Xdata), f64(Ydata), f64(Zdata)
        cw = np.empty(self.nelements_total)
        check(self._lib.tfx_column_weight_type3(self._h, C.c_int64(xd.size), ptr(xd), ptr(yd), ptr(zd), C.c_double(power),
                                                C.c_double(multiplier), ptr(cw)))
        return cw

    # ---- graviprism_z
    def graviprism_z(self, Xdata, Ydata, Zdata):
        xd, yd, zd = f64(np.atleast_1d(Xdata)), f64(np.atleast_1d(Ydata)), f64(np.atleast_1d(Zdata))
        rows = np.empty((xd.size, self.nelements_total))
        check(self._lib.tfx_prism_rows_gz(self._h, C.c_int64(xd.size), ptr(xd), ptr(yd), ptr(zd), ptr(rows)))
        return rows

    # ---- graviprism_full (gravity_field.f90:41-126): rows[ndata, 3, N] = LineX, LineY, LineZ of every observation
    def graviprism_full(self, Xdata, Ydata, Zdata):
        return self.sensit_lines(1, Xdata, Ydata, Zdata, data_type=1, ndata_components=3)[:, :, 0, :]

    # ---- magprism (TMI, scalar susceptibility); field = (inclination, declination, XaxisDeclination, intensity_nT)
    def magprism(self, Xdata, Ydata, Zdata, field):
        xd, yd, zd = f64(np.atleast_1d(Xdata)), f64(np.atleast_1d(Ydata)), f64(np.atleast_1d(Zdata))
        rows = np.empty((xd.size, self.nelements_total))
        incl, decl, azim, inten = [float(v) for v in field]
        check(self._lib.tfx_prism_rows_mag(self._h, C.c_int64(xd.size), ptr(xd), ptr(yd), ptr(zd), C.c_double(incl), C.c_double(decl),
                                           C.c_double(azim), C.c_double(inten), ptr(rows)))
        return rows

    # ---- any row generator of the build loop (graviprism_z / _full, gradiprism_zz / _full, magprism with 1|3 components)
    def sensit_lines(self, problem_type, Xdata, Ydata, Zdata, data_type=1, ndata_components=1, nmodel_components=1, mag_field=None):
        """-> rows[ndata, ndata_components, nmodel_components, N] = the reference's sensit_line_full(:, k, d) per observation
        (sensitivity_gravmag.F90:193-220)."""
        xd, yd, zd = f64(np.atleast_1d(Xdata)), f64(np.atleast_1d(Ydata)), f64(np.atleast_1d(Zdata))
        rows = np.empty((xd.size, ndata_components, nmodel_components, self.nelements_total))
        mf = None if mag_field is None else f64(mag_field)
        check(self._lib.tfx_prism_rows(self._h, int(problem_type), int(data_type), int(ndata_components), int(nmodel_components),
                                       C.c_int64(xd.size), ptr(xd), ptr(yd), ptr(zd), ptr(mf), ptr(rows)))
        return rows

    # ---- forward_wavelet / inverse_wavelet
    def forward_wavelet(self, s, n1, n2, n3, wavelet_type):
        return self._wavelet(s, n1, n2, n3, wavelet_type, 1)

    def inverse_wavelet(self, s, n1, n2, n3, wavelet_type):
        return self._wavelet(s, n1, n2, n3, wavelet_type, 2)

    def _wavelet(self, s, n1, n2, n3, wtype, direction):
        if wtype not in (1, 2):
            raise ValueError("Unknown wavelet type!")
        a = f64(s).copy()
        n = n1 * n2 * n3
        if a.size % n != 0:
            raise ValueError("array size is not a multiple of n1*n2*n3")
        check(self._lib.tfx_wavelet(self._h, ptr(a), n1, n2, n3, C.c_int64(a.size // n), wtype, direction))
        return a

    def wavelet_device(self, dev_ptr, n1, n2, n3, nvec, wtype, direction):
        """tfx_wavelet in place on a DEVICE buffer (raw address or torch tensor): kernels on the ctx stream, no host copy."""
        check(self._lib.tfx_wavelet(self._h, ptr(dev_ptr), n1, n2, n3, C.c_int64(nvec), int(wtype), int(direction)))

    def compress_row(self, row, K):
        row = f64(row)
        N = row.size
        cap = max(1, min(N, K))
        cols = np.empty(cap, np.int32)
        vals = np.empty(cap, np.float32)
        nel = C.c_int64()
        thr = C.c_double()
        cd = C.c_double()
        check(self._lib.tfx_compress_row(self._h, ptr(row), C.c_int64(N), C.c_int64(K), ptr(cols), ptr(vals), C.byref(nel),
                                         C.byref(thr), C.byref(cd)))
        return cols[:nel.value].copy(), vals[:nel.value].copy(), thr.value, cd.value

    # ---- calculate_and_write_sensit + read_sensitivity_kernel (no disk round trip)
    def calculate_sensit(self, Xdata, Ydata, Zdata, column_weight, compression_type, compression_rate, problem_weight=1.0,
                         data_weight=None, col_range=None, want_hist=False, mag_field=None, data_type=1, ndata_components=1,
                         nmodel_components=1):
        """mag_field = None: gravity (graviprism_z, or gradiometry with data_type = 2 and 1 | 6 data components);
        (incl, decl, azim, intensity_nT): magnetic kernel (magprism; 1 | 3 data components, 1 | 3 model components).
        With several components the matrix has ndata*ndata_components rows (d fastest) and nmodel_components column blocks;
        data_weight is then [ndata, ndata_components]."""
        xd, yd, zd = f64(Xdata), f64(Ydata), f64(Zdata)
        cw = f64(column_weight)
        N = self.nelements_total
        c0, c1 = (0, N) if col_range is None else col_range
        dw = None if data_weight is None else f64(data_weight)
        nnz = C.c_int64()
        err = C.c_double()
        hist = np.zeros(N, np.int32) if want_hist else None
        if data_type != 1 or ndata_components != 1 or nmodel_components != 1:
            mf = None if mag_field is None else f64(mag_field)
            check(self._lib.tfx_build_kernel(self._h, 1 if mag_field is None else 2, int(data_type), int(ndata_components),
                                             int(nmodel_components), C.c_int64(xd.size), ptr(xd), ptr(yd), ptr(zd), ptr(cw), ptr(mf),
                                             int(compression_type), C.c_double(compression_rate), C.c_double(problem_weight), ptr(dw),
                                             C.c_int64(c0), C.c_int64(c1), C.byref(nnz), C.byref(err), ptr(hist)))
            nlines = xd.size * ndata_components * nmodel_components
            return dict(nnz=nnz.value, error_sum=err.value, comp_error=err.value / nlines, nnz_hist=hist)
        if mag_field is None:
            check(self._lib.tfx_build_kernel_grav(self._h, C.c_int64(xd.size), ptr(xd), ptr(yd), ptr(zd), ptr(cw), int(compression_type),
                                                  C.c_double(compression_rate), C.c_double(problem_weight), ptr(dw), C.c_int64(c0),
                                                  C.c_int64(c1), C.byref(nnz), C.byref(err), ptr(hist)))
        else:
            incl, decl, azim, inten = [float(v) for v in mag_field]
            check(self._lib.tfx_build_kernel_mag(self._h, C.c_int64(xd.size), ptr(xd), ptr(yd), ptr(zd), ptr(cw), C.c_double(incl),
                                                 C.c_double(decl), C.c_double(azim), C.c_double(inten), int(compression_type),
                                                 C.c_double(compression_rate), C.c_double(problem_weight), ptr(dw), C.c_int64(c0),
                                                 C.c_int64(c1), C.byref(nnz), C.byref(err), ptr(hist)))
        return dict(nnz=nnz.value, error_sum=err.value, comp_error=err.value / xd.size, nnz_hist=hist)

    # ---- joint inversion: which of the two problems the matrix-level calls act on (LSQR uses both once slot 1 holds a matrix)
    def select_problem(self, slot):
        check(self._lib.tfx_select_problem(self._h, int(slot)))
        self._slot = int(slot)

    def system_dims(self):
        """(rows, columns) of the LSQR system matrix S = blockdiag(problem 0, problem 1)."""
        cur = getattr(self, "_slot", 0)
        nr = nc = 0
        try:
            for slot in (0, 1):
                self.select_problem(slot)
                try:
                    info = self.matrix_info()
                except TfxError:
                    if slot == 0:
                        raise
                    continue
                nr += info["nrows"]
                nc += info["ncols"]
        finally:
            self.select_problem(cur)
        return nr, nc

    # ---- row store / piecewise assembly (multi-GPU build, distributed.build_partitioned)
    ROW_BLOCK = 2048

    def rowstore_build(self, Xdata, Ydata, Zdata, column_weight, compression_type, compression_rate, problem_weight=1.0,
                       data_weight=None, mag_field=None, data_type=1, ndata_components=1, nmodel_components=1):
        xd, yd, zd, cw = f64(Xdata), f64(Ydata), f64(Zdata), f64(column_weight)
        dw = None if data_weight is None else f64(data_weight)
        mf = None if mag_field is None else f64(mag_field)
        nnz, err = C.c_int64(), C.c_double()
        hist = np.zeros(self.nelements_total, np.int32)
        if nmodel_components != 1:
            check(self._lib.tfx_rowstore_build_comp(self._h, 1 if mag_field is None else 2, int(data_type), int(ndata_components),
                                                    int(nmodel_components), C.c_int64(xd.size), ptr(xd), ptr(yd), ptr(zd), ptr(cw), ptr(mf),
                                                    int(compression_type), C.c_double(compression_rate), C.c_double(problem_weight), ptr(dw),
                                                    C.byref(nnz), C.byref(err), ptr(hist)))
            return dict(nnz=nnz.value, error_sum=err.value, nnz_hist=hist)
        if data_type != 1 or ndata_components != 1:
            check(self._lib.tfx_rowstore_build_ex(self._h, 1 if mag_field is None else 2, int(data_type), int(ndata_components),
                                                  C.c_int64(xd.size), ptr(xd), ptr(yd), ptr(zd), ptr(cw), ptr(mf), int(compression_type),
                                                  C.c_double(compression_rate), C.c_double(problem_weight), ptr(dw), C.byref(nnz),
                                                  C.byref(err), ptr(hist)))
            return dict(nnz=nnz.value, error_sum=err.value, nnz_hist=hist)
        check(self._lib.tfx_rowstore_build(self._h, 1 if mag_field is None else 2, C.c_int64(xd.size), ptr(xd), ptr(yd), ptr(zd), ptr(cw),
                                           ptr(mf), int(compression_type), C.c_double(compression_rate), C.c_double(problem_weight),
                                           ptr(dw), C.byref(nnz), C.byref(err), ptr(hist)))
        return dict(nnz=nnz.value, error_sum=err.value, nnz_hist=hist)

    def rowstore_counts(self, nrows_local, bounds):
        b = np.ascontiguousarray(bounds, np.int64)
        out = np.zeros((nrows_local, b.size - 1), np.int32)
        check(self._lib.tfx_rowstore_counts(self._h, int(b.size - 1), ptr(b), ptr(out)))
        return out

    def rowstore_pack(self, row_begin, nrows, col_begin, col_end, cols_dev, vals_dev, capacity):
        n = C.c_int64()
        check(self._lib.tfx_rowstore_pack(self._h, C.c_int64(row_begin), C.c_int64(nrows), C.c_int64(col_begin), C.c_int64(col_end),
                                          ptr(cols_dev), ptr(vals_dev), C.c_int64(capacity), C.byref(n)))
        return n.value

    def rowstore_free(self):
        check(self._lib.tfx_rowstore_free(self._h))

    def matrix_begin(self, nrows, ncols, nnz_upper):
        check(self._lib.tfx_matrix_begin(self._h, C.c_int64(nrows), C.c_int64(ncols), C.c_int64(nnz_upper)))

    def matrix_append_rows(self, row_begin, cols_dev, vals_dev, nel):
        nel = np.ascontiguousarray(nel, np.int32)
        check(self._lib.tfx_matrix_append_rows(self._h, C.c_int64(row_begin), C.c_int64(nel.size), ptr(cols_dev), ptr(vals_dev), ptr(nel)))

    def matrix_finish(self):
        check(self._lib.tfx_matrix_finish(self._h))

    # ---- t_sparse_matrix
    def matrix_upload_csr(self, nrows, ncols, rowptr, cols, vals):
        rp = np.ascontiguousarray(rowptr, np.int64)
        c = np.ascontiguousarray(cols, np.int32)
        v = np.ascontiguousarray(vals, np.float32)
        if rp.size != nrows + 1:
            raise ValueError("rowptr size")
        check(self._lib.tfx_matrix_upload_csr(self._h, C.c_int64(nrows), C.c_int64(ncols), ptr(rp), ptr(c), ptr(v)))

    def matrix_info(self):
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        check(self._lib.tfx_matrix_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(nrows=a.value, ncols=b.value, nnz=c.value, device_bytes=d.value)

    def matrix_format(self):
        b, n, sb, adj = C.c_double(), C.c_int64(), C.c_int64(), C.c_int()
        check(self._lib.tfx_matrix_format(self._h, C.byref(b), C.byref(n), C.byref(sb), C.byref(adj)))
        return dict(bytes_per_entry=b.value, stored_entries=n.value, stream_bytes=sb.value, adjoint_copy=bool(adj.value))

    def matrix_download_csr(self):
        info = self.matrix_info()
        rp = np.empty(info["nrows"] + 1, np.int64)
        cols = np.empty(max(1, info["nnz"]), np.int32)
        vals = np.empty(max(1, info["nnz"]), np.float32)
        check(self._lib.tfx_matrix_download_csr(self._h, ptr(rp), ptr(cols), ptr(vals)))
        return rp, cols[:rp[-1]], vals[:rp[-1]]

    def cons_upload_csr(self, rowptr, cols, vals, rhs):
        """General constraint rows (matrix_cons) + their right-hand side; used by the next lsqr_* calls."""
        rp = np.ascontiguousarray(rowptr, np.int64)
        c = np.ascontiguousarray(cols, np.int32)
        v = np.ascontiguousarray(vals, np.float32)
        b = f64(rhs)
        if b.size != rp.size - 1:
            raise ValueError("rhs size != number of constraint rows")
        check(self._lib.tfx_cons_upload_csr(self._h, C.c_int64(rp.size - 1), ptr(rp), ptr(c), ptr(v), ptr(b)))

    def cons_clear(self):
        check(self._lib.tfx_cons_clear(self._h))

    def matrix_free(self):
        check(self._lib.tfx_matrix_free(self._h))

    def matrix_scale_rows(self, scale):
        """Row r of the selected matrix times float32(scale[r]) in fp32: the problem_weight * data_weight scaling that
        read_sensitivity_kernel applies to a kernel stored unscaled (sensitivity_gravmag.F90:834-843)."""
        sc = f64(scale)
        if sc.size != self.matrix_info()["nrows"]:
            raise ValueError("scale size != number of matrix rows")
        check(self._lib.tfx_matrix_scale_rows(self._h, ptr(sc)))

    def matrix_reserve(self, nnz_upper):
        """Entry bound of the next kernel build into the selected slot (tfx_matrix_reserve): the sum of the per-column histogram over
        the column range a rank builds."""
        check(self._lib.tfx_matrix_reserve(self._h, C.c_int64(int(nnz_upper))))

    def normalize_columns(self):
        """t_sparse_matrix%normalize_columns (sparse_matrix.f90:414-443): scales the columns of the selected matrix to unit length
        (zero columns stay) and returns their original norms."""
        norm = np.empty(self.matrix_info()["ncols"], np.float64)
        check(self._lib.tfx_matrix_normalize_columns(self._h, ptr(norm)))
        return norm

    def mult_vector(self, x, b=None):
        """b = S x (mult_vector) or b += S x when b is given (add_mult_vector)."""
        info = self.matrix_info()
        x = f64(x)
        if x.size != info["ncols"]:
            raise ValueError("x size %d != ncols %d" % (x.size, info["ncols"]))
        add = b is not None
        out = f64(b).copy() if add else np.empty(info["nrows"])
        check(self._lib.tfx_spmv(self._h, ptr(x), ptr(out), int(add)))
        return out

    def trans_mult_vector(self, x, b=None):
        info = self.matrix_info()
        x = f64(x)
        if x.size != info["nrows"]:
            raise ValueError("x size %d != nrows %d" % (x.size, info["nrows"]))
        add = b is not None
        out = f64(b).copy() if add else np.empty(info["ncols"])
        check(self._lib.tfx_spmtv(self._h, ptr(x), ptr(out), int(add)))
        return out

    # ---- lsqr_solve_sensit
    def _blocks(self, diag_blocks, rhs_blocks, ncols):
        nb = len(diag_blocks)
        if nb != len(rhs_blocks):
            raise ValueError("diag / rhs block count mismatch")
        keep = []
        dp = (C.c_void_p * max(1, nb))()
        rp = (C.c_void_p * max(1, nb))()
        for i, (d, r) in enumerate(zip(diag_blocks, rhs_blocks)):
            d = np.ascontiguousarray(d, np.float32)
            r = f64(r)
            if d.size != ncols or r.size != ncols:
                raise ValueError("constraint block size != ncols")
            keep += [d, r]
            dp[i] = d.ctypes.data
            rp[i] = r.ctypes.data
        return nb, dp, rp, keep

    def lsqr_solve_sensit(self, b_data, niter, rmin=1e-13, gamma=0.0, target_misfit=0.0, diag_blocks=(), rhs_blocks=()):
        nrows, ncols = self.system_dims()
        b = f64(b_data)
        if b.size != nrows:
            raise ValueError("Wrong matrix sizes in lsqr_solve_sensit!")
        nb, dp, rp, keep = self._blocks(diag_blocks, rhs_blocks, ncols)
        x = np.empty(ncols)
        it = C.c_int()
        r = C.c_double()
        check(self._lib.tfx_lsqr_solve(self._h, int(niter), C.c_double(rmin), C.c_double(gamma), C.c_double(target_misfit), ptr(b),
                                       nb, dp, rp, ptr(x), C.byref(it), C.byref(r)))
        return x, it.value, r.value

    def lsqr_set_wavelet_domain(self, wavelet_domain, wavelet_type=0):
        """WAVELET_DOMAIN = False: spatial unknowns, S applied through the wavelet transform (single rank)."""
        n1, n2, n3 = self.dims if self.dims else (0, 0, 0)
        check(self._lib.tfx_lsqr_set_wavelet_domain(self._h, int(bool(wavelet_domain)), n1, n2, n3, int(wavelet_type)))

    def lsqr_set_partition(self, col_begin, ncomponents=1):
        """Multi-rank WAVELET_DOMAIN = False: first cell of this rank's column range and the number of model components."""
        check(self._lib.tfx_lsqr_set_partition(self._h, C.c_int64(col_begin), int(ncomponents)))

    def lsqr_begin(self, b_data, rmin=1e-13, gamma=0.0, target_misfit=0.0, diag_blocks=(), rhs_blocks=()):
        b = f64(b_data)
        nb, dp, rp, keep = self._blocks(diag_blocks, rhs_blocks, self.system_dims()[1])
        check(self._lib.tfx_lsqr_begin(self._h, C.c_double(rmin), C.c_double(gamma), C.c_double(target_misfit), ptr(b), nb, dp, rp))

    def lsqr_iterate(self, k):
        done = C.c_int()
        r = C.c_double()
        check(self._lib.tfx_lsqr_iterate(self._h, int(k), C.byref(done), C.byref(r)))
        return done.value, r.value

    def lsqr_end(self):
        x = np.empty(self.system_dims()[1])
        check(self._lib.tfx_lsqr_end(self._h, ptr(x)))
        return x

    # ---- model_calculate_data (after un-weighting + wavelet)
    def calc_data(self, xw_local, problem_weight=1.0, data_weight=None):
        info = self.matrix_info()
        x = f64(xw_local)
        dw = None if data_weight is None else f64(data_weight)
        out = np.empty(info["nrows"])
        check(self._lib.tfx_calc_data(self._h, ptr(x), C.c_double(problem_weight), ptr(dw), ptr(out)))
        return out

    def fastmath_eval(self, a, b):
        """Diagnostics: the device log(a) and atan2(a, b) of the prism kernels (csrc/fastmath.h)."""
        a = f64(a)
        b = f64(b)
        out_log = np.empty(a.size)
        out_atan = np.empty(a.size)
        check(self._lib.tfx_fastmath_eval(self._h, C.c_int64(a.size), ptr(a), ptr(b), ptr(out_log), ptr(out_atan)))
        return out_log, out_atan

    def debug_set(self, key, value=0):
        rc = self._lib.tfx_debug_set(self._h, key.encode(), int(value))
        if rc < 0:
            check(rc)
        return rc

    # ---- timing
    def timer_start(self):
        check(self._lib.tfx_timer_start(self._h))

    def timer_stop_ms(self):
        ms = C.c_double()
        check(self._lib.tfx_timer_stop_ms(self._h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        check(self._lib.tfx_profile_enable(self._h, int(bool(on))))

    def profile_get(self, which):
        ms = C.c_double()
        n = C.c_int64()
        check(self._lib.tfx_profile_get(self._h, int(which), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def get_load_balancing_nelements(sensit_nnz, nbproc):
    """src/forward/gravmag/sensitivity_gravmag.F90:470-524: returns (nelements_at_cpu, nnz_at_cpu)."""
    lib = L.load()
    h = np.ascontiguousarray(sensit_nnz, np.int32)
    nel = np.zeros(nbproc, np.int32)
    nnz = np.zeros(nbproc, np.int64)
    check(lib.tfx_partition_columns(ptr(h), C.c_int64(h.size), int(nbproc), ptr(nel), ptr(nnz)))
    return nel, nnz
