"""ctypes binding of libdqc_amd.so (C ABI declared in include/dqc_amd.h).

The product path has no CPU fallback: if the HIP library cannot be loaded this module raises.
PyTorch is used only for device memory and streams (tensor.data_ptr(), current stream handle).
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("DQC_AMD_LIB") or os.path.join(_HERE, "libdqc_amd.so")  # override: perf-bisection variants
_lib = None

XC_IDS = {"lda_x": 1, "lda_c_vwn": 7, "lda_c_pw": 12, "lda_c_pw_mod": 13, "gga_x_pbe": 101, "gga_x_pbe_r": 102, "gga_x_b88": 106,
          "gga_x_pbe_sol": 116, "gga_x_rpbe": 117, "gga_c_pbe": 130, "gga_c_lyp": 131, "gga_c_pbe_sol": 133,
          "mgga_x_scan": 263, "mgga_c_scan": 267, "mgga_x_tpss": 202, "mgga_c_tpss": 231,
          "lda_c_pz": 9, "gga_x_b86": 103, "gga_x_g96": 107, "gga_x_pw86": 108, "gga_x_pw91": 109, "gga_x_optx": 110, "gga_x_wc": 118,
          "gga_c_p86": 132}


class DqcAmdError(RuntimeError):
    pass


def libpath():
    return _LIBPATH


def load():
    """Load libdqc_amd.so; raises (loudly) when it is missing -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIBPATH) and os.environ.get("DQC_AMD_AUTOBUILD", "0") == "1" and "DQC_AMD_LIB" not in os.environ:
        from . import build as _build  # fresh clone: compile the HIP sources in-tree (hipcc, ~1 min)
        _build.build()
    if not os.path.exists(_LIBPATH):
        raise DqcAmdError(
            "libdqc_amd.so not found at %s -- build it with `python -m dqc_amd.build` "
            "(hipcc --offload-arch=gfx950) or set DQC_AMD_AUTOBUILD=1; the MI355X path has no CPU fallback" % _LIBPATH)
    lib = ctypes.CDLL(_LIBPATH)
    c_int, c_sz, c_vp, c_dp = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p
    ip, dp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)
    lib.dqc_last_error.restype = ctypes.c_char_p
    lib.dqc_version.restype = c_int
    lib.dqc_nao.argtypes = [ip, c_int]
    lib.dqc_padded_nao.argtypes = [c_int]
    lib.dqc_ao_stride.argtypes = [c_int]
    lib.dqc_ao_doubles.argtypes = [c_int, c_int, c_int]
    lib.dqc_ao_doubles.restype = c_sz
    lib.dqc_eri_tile_count.argtypes = [c_int]
    lib.dqc_eri_tile_count.restype = c_sz
    if hasattr(lib, "dqc_eri_store_doubles"):  # (absent from the pre-packing A/B build tools/gpu_jk_ab.py loads)
        lib.dqc_eri_store_doubles.argtypes = [c_int]
        lib.dqc_eri_store_doubles.restype = c_sz
    lib.dqc_eri_tile_offset.argtypes = [c_int, ctypes.c_longlong]
    lib.dqc_eri_tile_offset.restype = ctypes.c_longlong
    lib.dqc_eri_fill_tiles_part.argtypes = [c_dp, ip, c_int, ip, c_int, dp, c_int, ctypes.c_longlong, ctypes.c_longlong, c_vp]
    lib.dqc_jk_from_tiles_part.argtypes = [c_dp, c_dp, c_dp, c_dp, c_int, c_dp, ctypes.c_longlong, ctypes.c_longlong, c_vp]
    lib.dqc_jk_work_doubles.argtypes = [c_int]
    lib.dqc_jk_work_doubles.restype = c_sz
    tab = [ip, c_int, ip, c_int, dp, c_int]
    lib.dqc_int1e.argtypes = [c_int, c_dp] + tab + [dp, c_vp]
    lib.dqc_eri_fill_tiles.argtypes = [c_dp] + tab + [c_vp]
    lib.dqc_jk_direct.argtypes = [c_dp, c_dp, c_dp] + tab + [c_vp]
    lib.dqc_direct_create.argtypes = [ctypes.POINTER(c_vp)] + tab + [c_vp]
    lib.dqc_direct_jk.argtypes = [c_vp, c_dp, c_dp, c_dp, ctypes.c_double, c_vp]
    lib.dqc_direct_jk_part.argtypes = [c_vp, c_dp, c_dp, c_dp, ctypes.c_double, c_int, c_int, c_vp]
    lib.dqc_direct_stats.argtypes = [c_vp, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong), dp]
    lib.dqc_direct_npairs.argtypes = [c_vp]
    lib.dqc_eri_pair_stats.argtypes = [ip, c_int, ip, c_int, dp, c_int, c_int, ctypes.POINTER(ctypes.c_longlong)]
    lib.dqc_direct_bounds.argtypes = [c_vp, dp, ip]
    lib.dqc_direct_bounds_groups.argtypes = [c_vp, dp, ip, ip]
    lib.dqc_direct_destroy.argtypes = [c_vp]
    lib.dqc_int3c2e.argtypes = [c_dp] + tab + [c_int, c_int, c_int, c_int, c_vp]
    lib.dqc_int2c2e.argtypes = [c_dp] + tab + [c_int, c_int, c_vp]
    lib.dqc_ncart.argtypes = [ip, c_int]
    lib.dqc_cart2sph_matrix.argtypes = [dp, ip, c_int]
    lib.dqc_int1e_grad.argtypes = [c_dp, c_dp, c_dp] + tab + [dp, c_vp]
    lib.dqc_eri_grad.argtypes = [c_dp, c_dp, ctypes.c_double, ctypes.c_double] + tab + [c_vp]
    lib.dqc_df_grad.argtypes = [c_dp, c_dp, c_dp] + tab + [c_int, c_int, c_int, c_int, c_vp]
    lib.dqc_becke_weights.argtypes = [c_dp, c_dp, c_vp, c_dp, c_dp, c_dp, c_int, c_int, ctypes.c_double, c_vp]
    lib.dqc_becke_weights_grad.argtypes = [c_dp, c_dp, c_dp, c_dp, c_dp, c_vp, c_dp, c_dp, c_dp, c_int, c_int, ctypes.c_double, c_vp]
    lib.dqc_purify_tc2.argtypes = [c_dp, c_dp, c_int, ctypes.c_double, c_int, ctypes.c_double, c_dp, c_vp]
    lib.dqc_orth_factor.argtypes = [c_dp, c_dp, c_dp, c_int, c_int, c_vp]
    lib.dqc_purify_tc2_batched.argtypes = [c_dp, c_dp, c_int, c_int, ctypes.c_double, c_int, ctypes.c_double, c_dp, c_vp]
    lib.dqc_orth_factor_batched.argtypes = [c_dp, c_dp, c_dp, c_int, c_int, c_int, c_vp]
    lib.dqc_diis_solve.argtypes = [c_dp, c_dp, c_int, c_int, c_int, c_vp]
    lib.dqc_diis_solve_dev.argtypes = [c_dp, c_dp, c_int, c_int, c_vp, c_vp]
    lib.dqc_projector_work_doubles.argtypes = [c_int, c_int]
    lib.dqc_projector_work_doubles.restype = c_sz
    lib.dqc_projector_tc2.argtypes = [c_dp, c_dp, c_dp, c_int, ctypes.c_double, c_int, ctypes.c_double, c_dp, c_vp]
    lib.dqc_purify_tc2_persist.argtypes = [c_dp, c_dp, c_int, ctypes.c_double, c_int, ctypes.c_double, c_dp, c_vp, c_vp]
    lib.dqc_df_coulomb.argtypes = [c_dp, c_dp, c_dp, c_dp, c_int, c_int, c_dp, c_vp]
    lib.dqc_eri_tiles_to_dense.argtypes = [c_dp, c_dp, c_int, c_vp]
    lib.dqc_jk_from_tiles.argtypes = [c_dp, c_dp, c_dp, c_dp, c_int, c_dp, c_vp]
    lib.dqc_jk_multi_work_doubles.argtypes = [c_int, c_int, c_int]
    lib.dqc_jk_multi_work_doubles.restype = c_sz
    lib.dqc_jk_from_tiles_multi.argtypes = [c_dp, c_dp, c_int, c_dp, c_dp, c_int, c_dp, c_int, c_dp, c_vp]
    lib.dqc_eval_gto.argtypes = [c_int, c_dp, c_dp, c_int] + tab + [c_vp]
    lib.dqc_grid_density.argtypes = [c_dp, c_dp, c_dp, c_int, c_int, c_int, c_dp, c_vp]
    lib.dqc_xc_eval.argtypes = [c_dp, c_dp, c_dp, c_dp, c_dp, c_int, ip, dp, c_int, c_vp]
    lib.dqc_xc_eval_quad.argtypes = [c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_int, ip, dp, c_int, c_vp]
    lib.dqc_xc_eval_pol.argtypes = [c_dp] * 9 + [c_int, ip, dp, c_int, c_vp]
    lib.dqc_xc_eval_mgga.argtypes = [c_dp] * 7 + [c_int, ip, dp, c_int, c_vp]
    lib.dqc_xc_eval_mgga_pol.argtypes = [c_dp] * 11 + [c_int, ip, dp, c_int, c_vp]
    lib.dqc_xc_eval_mgga_pol2.argtypes = [c_dp] * 12 + [c_int, ip, dp, c_int, c_vp]
    lib.dqc_stream_create_partition.argtypes = [ctypes.POINTER(c_vp), c_int, c_int, c_int]
    lib.dqc_stream_destroy.argtypes = [c_vp]
    lib.dqc_stream_cus.argtypes = [c_vp]
    lib.dqc_set_vxc_cus.argtypes = [c_int]
    lib.dqc_fock_factor.argtypes = [c_dp, c_dp, c_dp, c_dp, c_int, c_dp, c_int, c_int, c_int, c_int, c_int, c_vp]
    lib.dqc_fock_orb2dm.argtypes = [c_dp, c_dp, c_dp, c_dp, c_dp, c_int, c_dp, c_int, c_int, c_int, c_int, c_int, c_vp]
    lib.dqc_grid_vxc_raw.argtypes = [c_dp, c_dp, c_int, c_int, c_int, c_dp, c_dp, c_dp, dp, c_vp]
    lib.dqc_fock_finish_vraw.argtypes = [c_dp, c_dp, c_dp, c_dp, c_int, ctypes.c_double, c_dp, c_dp, c_int, c_int, c_vp]
    lib.dqc_fock_prep.argtypes = [c_dp, c_dp, c_dp, c_dp, c_int, c_int, c_int, c_int, c_vp]
    lib.dqc_jk_stream_prepared.argtypes = [c_dp, c_int, c_dp, c_int, c_vp]
    lib.dqc_fock_finish.argtypes = [c_dp, c_dp, c_dp, c_dp, c_dp, c_int, c_dp, c_dp, c_int, c_int, c_int, c_vp]
    lib.dqc_padded_norb.argtypes = [c_int]
    lib.dqc_padded_norb.restype = c_int
    lib.dqc_grid_density_lr.argtypes = [c_dp, c_dp, c_dp, c_int, c_int, c_int, c_dp, c_dp, c_int, c_vp]
    lib.dqc_grid_density_lr_pol.argtypes = [c_dp, c_dp, c_dp, c_int, c_int, c_int, c_dp, c_dp, c_int, c_vp]
    lib.dqc_grid_xc_gradient_terms.argtypes = [c_dp, c_dp, c_dp, c_int, c_int, c_dp, c_dp, c_dp, c_dp, c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_vp]
    lib.dqc_grid_density_lr_tau.argtypes = [c_dp, c_dp, c_dp, c_dp, c_int, c_int, c_int, c_dp, c_int, c_vp]
    lib.dqc_grid_density_pair.argtypes = [c_dp, c_dp, c_dp, c_int, c_int, c_dp, c_vp]
    lib.dqc_grid_vxc_pair.argtypes = [c_dp, c_dp, c_dp, c_int, c_int, c_dp, c_dp, c_vp]
    lib.dqc_grid_vxc.argtypes = [c_dp, c_dp, c_int, c_int, c_int, c_dp, c_dp, c_dp, c_vp]
    lib.dqc_probe_stream_read.argtypes = [c_dp, c_sz, c_dp, c_vp]
    lib.dqc_probe_mfma_f64.argtypes = [c_dp, c_int, c_vp]
    _lib = lib
    if os.environ.get("DQC_AMD_DETERMINISTIC", "0") == "1":
        lib.dqc_set_deterministic(1)
    return lib


def set_generic_eri(on=True):
    """every shell-quartet class through the runtime-angular-momentum integral kernel (include/dqc_amd.h:
    dqc_set_generic_eri) -- normally only the classes with a g shell take it.  Returns the previous setting."""
    return bool(load().dqc_set_generic_eri(1 if on else 0))


def set_deterministic(on=True):
    """bit-reproducible Fock builds: the cross-block sums (J / K accumulators, split-K Vxc, purification trace) use fixed-point
    integer atomics instead of fp64 atomics (include/dqc_amd.h: dqc_set_deterministic).  Returns the previous setting.
    Environment: DQC_AMD_DETERMINISTIC=1."""
    return bool(load().dqc_set_deterministic(1 if on else 0))


_TRACE = None      # a list while `call_trace` is active: (entry point, start event, end event) per library call
_TRACE_OPEN = None


class call_trace:
    """measurement aid: `with lib.call_trace() as tr:` brackets every library call made inside with two HIP events on the stream
    the call is enqueued on; `tr.ms()` -> {entry point: (calls, mean ms per call)} (synchronises).  bench.py's per-kernel
    rooflines of the legs that have no hand-unrolled event chain come from here; nothing is recorded outside the block."""

    def __enter__(self):
        global _TRACE
        self.rows = _TRACE = []
        return self

    def __exit__(self, *a):
        global _TRACE, _TRACE_OPEN
        _TRACE = _TRACE_OPEN = None
        return False

    def ms(self):
        torch.cuda.synchronize()
        out = {}
        for what, e0, e1 in self.rows:
            n, t = out.get(what, (0, 0.0))
            out[what] = (n + 1, t + e0.elapsed_time(e1))
        return {k: (n, t / n) for k, (n, t) in out.items()}


def _check(rc, what):
    global _TRACE_OPEN
    if _TRACE is not None and _TRACE_OPEN is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(torch.cuda.current_stream(_TRACE_OPEN[1]))
        _TRACE.append((what, _TRACE_OPEN[0], e1))
        _TRACE_OPEN = None
    if rc != 0:
        raise DqcAmdError("%s failed (%d): %s" % (what, rc, load().dqc_last_error().decode()))


class _on:
    """Run a C entry point on the device that owns the tensors: the library launches on (and hipMallocs from) the CURRENT
    HIP device and on the stream it is handed, so both are taken from `dev`, not from whatever device happens to be
    current (Mol(..., device="cuda:1") while cuda:0 is current)."""

    def __init__(self, dev):
        dev = torch.device(dev)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.idx = idx
        self.guard = None if idx == torch.cuda.current_device() else torch.cuda.device(idx)

    def __enter__(self):
        global _TRACE_OPEN
        if self.guard is not None:
            self.guard.__enter__()
        st = torch.cuda.current_stream(self.idx)
        if _TRACE is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            _TRACE_OPEN = (e0, self.idx)
        return ctypes.c_void_p(st.cuda_stream)

    def __exit__(self, *a):
        if self.guard is not None:
            self.guard.__exit__(*a)
        return False


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous(), "need contiguous float64 device tensor"
    return ctypes.c_void_p(t.data_ptr())


class Tables:
    """Host-side libcint-style tables (atm, bas, env) -- see dqc_amd.basis.make_tables."""

    def __init__(self, atm, bas, env):
        self.atm = np.ascontiguousarray(atm, dtype=np.int32)
        self.bas = np.ascontiguousarray(bas, dtype=np.int32)
        self.env = np.ascontiguousarray(env, dtype=np.float64)
        self.natm, self.nbas = self.atm.shape[0], self.bas.shape[0]
        self.nao = int(sum(2 * int(b[1]) + 1 for b in self.bas))

    def args(self):
        ip, dp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)
        return (self.atm.ctypes.data_as(ip), self.natm, self.bas.ctypes.data_as(ip), self.nbas,
                self.env.ctypes.data_as(dp), self.env.shape[0])


def padded_nao(nao):
    """rows / columns of the zero-padded AO-indexed square matrices (D, V, the factor pair): whole 16 x 16 tiles"""
    return int(load().dqc_padded_nao(int(nao)))


def ao_stride(nao):
    """row stride of the AO-on-grid arrays (nao rounded up to 8 doubles: 64-byte rows, no tile padding in HBM)"""
    return int(load().dqc_ao_stride(int(nao)))


def ao_empty(ncomp, ngrid, nao, device, zero=False):
    """an AO-on-grid array for the grid kernels: (ngrid, lda) for ncomp = 0, else (ncomp, ngrid, lda), a view of a flat buffer
    of dqc_ao_doubles doubles -- the kernels read whole 16-column tiles, up to ld - lda doubles past the end of the last row
    (that slack is zeroed here)"""
    nc = max(int(ncomp), 1)
    lda = ao_stride(nao)
    tot = int(load().dqc_ao_doubles(nc, int(ngrid), int(nao)))
    flat = (torch.zeros if zero else torch.empty)(tot, dtype=torch.float64, device=device)
    body = nc * int(ngrid) * lda
    if not zero and tot > body:
        flat[body:].zero_()
    out = flat[:body].view(nc, int(ngrid), lda)
    return out[0] if ncomp == 0 else out


def ao_from(values, nao=None):
    """values (ngrid, n) or (ncomp, ngrid, n) with n >= nao (any device tensor / view, e.g. the sum of two AO arrays) -> a
    fresh array in the kernels' layout (row stride lda, zero padding columns, zeroed slack) holding values[..., :nao]"""
    nao = values.shape[-1] if nao is None else nao
    out = ao_empty(0 if values.dim() == 2 else values.shape[0], values.shape[-2], nao, values.device, zero=True)
    out[..., :nao] = values[..., :nao]
    return out


def int1e(which, tab, device, zs=None):
    """which: 'ovlp' | 'kin' | 'nuc' -> (nao, nao) device tensor; 'r0' -> (3, nao, nao), 'r0r0' -> (9, nao, nao): multipole
    moments about the origin (the reference's intor.int1e("r0" * n), hcgto.py:117-125)"""
    if which in ("r0", "r0r0"):
        return torch.stack([int1e(c, tab, device) for c in (range(3, 6) if which == "r0" else range(6, 15))])
    code = which if isinstance(which, int) else {"ovlp": 0, "kin": 1, "nuc": 2}[which]
    out = torch.zeros((tab.nao, tab.nao), dtype=torch.float64, device=device)
    zp = None
    if zs is not None:
        zs = np.ascontiguousarray(zs, dtype=np.float64)
        zp = zs.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    with _on(device) as st_:
        _check(load().dqc_int1e(code, _ptr(out), *tab.args(), zp, st_), "dqc_int1e")
    return out


def eri_store_doubles(nao):
    """doubles of the packed ERI tile store of `nao` functions"""
    return int(load().dqc_eri_store_doubles(int(nao)))


def eri_tiles(tab, device):
    tiles = torch.empty(eri_store_doubles(tab.nao), dtype=torch.float64, device=device)  # packed tiles
    with _on(device) as st_:
        _check(load().dqc_eri_fill_tiles(_ptr(tiles), *tab.args(), st_), "dqc_eri_fill_tiles")
    return tiles


def _range_nao(tab, s0, s1):
    return int(sum(2 * int(b[1]) + 1 for b in tab.bas[s0:s1]))


def int3c2e(tab, orb_range, aux_range, device):
    """(ij|k) over concatenated tables: orbital shells [s0, s1), auxiliary shells [k0, k1) -> (nao, nao, naux)"""
    (s0, s1), (k0, k1) = orb_range, aux_range
    nao, naux = _range_nao(tab, s0, s1), _range_nao(tab, k0, k1)
    out = torch.zeros((nao, nao, naux), dtype=torch.float64, device=device)
    with _on(device) as st_:
        _check(load().dqc_int3c2e(_ptr(out), *tab.args(), s0, s1, k0, k1, st_), "dqc_int3c2e")
    return out


def int2c2e(tab, aux_range, device):
    """(k|l) over the auxiliary shells [k0, k1) of concatenated tables -> (naux, naux)"""
    k0, k1 = aux_range
    naux = _range_nao(tab, k0, k1)
    out = torch.zeros((naux, naux), dtype=torch.float64, device=device)
    with _on(device) as st_:
        _check(load().dqc_int2c2e(_ptr(out), *tab.args(), k0, k1, st_), "dqc_int2c2e")
    return out


def df_coulomb(j3c, inv_j2c, dm_ao, work=None):
    """J_ao (nao, nao) = sum_k (ij|k) [inv_j2c (kl|D)]_k from the stored DF integrals; dm_ao (nao, nao) contiguous"""
    nao, _, naux = j3c.shape
    if work is None:
        work = torch.empty(2 * naux, dtype=torch.float64, device=j3c.device)
    out = torch.empty((nao, nao), dtype=torch.float64, device=j3c.device)
    with _on(j3c.device) as st_:
        _check(load().dqc_df_coulomb(_ptr(out), _ptr(j3c), _ptr(inv_j2c), _ptr(dm_ao), nao, naux, _ptr(work), st_),
               "dqc_df_coulomb")
    return out


def cart2sph_matrix(tab, device):
    """T (nao, ncart) block diagonal, chi_m = sum_c T[m, c] g_c (solid harmonics of the Cartesian Gaussians)"""
    ip = ctypes.POINTER(ctypes.c_int)
    ncart = int(load().dqc_ncart(tab.bas.ctypes.data_as(ip), tab.nbas))
    out = np.zeros((tab.nao, ncart))
    _check(load().dqc_cart2sph_matrix(out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), tab.bas.ctypes.data_as(ip), tab.nbas),
           "dqc_cart2sph_matrix")
    return torch.as_tensor(out, device=device)


def int1e_grad(grad, dcart, wcart, tab, zs=None):
    """grad (natm, 3) += one-electron derivative terms; dcart / wcart (ncart, ncart) contiguous device tensors"""
    zp = None
    if zs is not None:
        zs = np.ascontiguousarray(zs, dtype=np.float64)
        zp = zs.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    with _on(grad.device) as st_:
        _check(load().dqc_int1e_grad(_ptr(grad), _ptr(dcart), _ptr(wcart), *tab.args(), zp, st_), "dqc_int1e_grad")
    return grad


def eri_grad(grad, dcart, kscale, tab, jscale=1.0):
    """grad (natm, 3) += two-electron derivative term  sum (d_A a b|c d) [2 jscale D_ab D_cd - kscale D_ac D_bd]"""
    with _on(grad.device) as st_:
        _check(load().dqc_eri_grad(_ptr(grad), _ptr(dcart), float(jscale), float(kscale), *tab.args(), st_), "dqc_eri_grad")
    return grad


def becke_weights(xyz, atom_off, pos, inv_rij, aij, cut):
    """xyz (ngrid, 3) the atoms' grids concatenated, atom_off (natm + 1,) int32 device tensor -> Becke partition weights (ngrid,)"""
    ngrid, natm = xyz.shape[0], pos.shape[0]
    assert atom_off.dtype == torch.int32 and atom_off.is_cuda and atom_off.numel() == natm + 1
    w = torch.empty(ngrid, dtype=torch.float64, device=xyz.device)
    with _on(xyz.device) as st_:
        _check(load().dqc_becke_weights(_ptr(w), _ptr(xyz), ctypes.c_void_p(atom_off.data_ptr()), _ptr(pos.contiguous()),
                                        _ptr(inv_rij.contiguous()), _ptr(aij.contiguous()), natm, ngrid, float(cut), st_),
               "dqc_becke_weights")
    return w


def becke_weights_grad(cw, xyz, atom_off, pos, inv_rij, aij, cut):
    """backward of becke_weights: cw (ngrid,) = dL/dw -> (gpos (natm, 3) explicit nuclear part, gxyz (ngrid, 3) point part)"""
    ngrid, natm = xyz.shape[0], pos.shape[0]
    gpos = torch.zeros((natm, 3), dtype=torch.float64, device=xyz.device)
    gxyz = torch.empty((ngrid, 3), dtype=torch.float64, device=xyz.device)
    scratch = torch.empty((natm, ngrid), dtype=torch.float64, device=xyz.device)
    with _on(xyz.device) as st_:
        _check(load().dqc_becke_weights_grad(_ptr(gpos), _ptr(gxyz), _ptr(scratch), _ptr(cw.contiguous()), _ptr(xyz.contiguous()),
                                             ctypes.c_void_p(atom_off.data_ptr()), _ptr(pos.contiguous()), _ptr(inv_rij.contiguous()),
                                             _ptr(aij.contiguous()), natm, ngrid, float(cut), st_), "dqc_becke_weights_grad")
    return gpos, gxyz


def xc_eval_mgga_pol2(terms, rho_u, rho_d, grho_u, grho_d, tau_u, tau_d, want_e=True, want_v=True):
    """spin-polarised meta-GGA correlation terms with one gradient potential per spin (mgga_c_scan, mgga_c_tpss)
    -> edens, (vrho_u, vrho_d), (vgrad_u, vgrad_d) (3, n) each, vtau shared"""
    n = rho_u.shape[0]
    ids = (ctypes.c_int * len(terms))(*[XC_IDS[nm] for _, nm in terms])
    cfs = (ctypes.c_double * len(terms))(*[float(c) for c, _ in terms])
    e = torch.empty_like(rho_u) if want_e else None
    vu = torch.empty_like(rho_u) if want_v else None
    vd = torch.empty_like(rho_u) if want_v else None
    gu = torch.empty((3, n), dtype=torch.float64, device=rho_u.device) if want_v else None
    gd = torch.empty((3, n), dtype=torch.float64, device=rho_u.device) if want_v else None
    vt = torch.empty_like(rho_u) if want_v else None
    with _on(rho_u.device) as st_:
        _check(load().dqc_xc_eval_mgga_pol2(_ptr(e), _ptr(vu), _ptr(vd), _ptr(gu), _ptr(gd), _ptr(vt), _ptr(rho_u), _ptr(rho_d),
                                            _ptr(grho_u), _ptr(grho_d), _ptr(tau_u), _ptr(tau_d), n, ids, cfs, len(terms), st_),
               "dqc_xc_eval_mgga_pol2")
    return e, (vu, vd), (gu, gd), vt


def purify_tc2(x_pad, tmp, nocc, iters, tol, state):
    """in-place TC2 purification of the zero-padded (ld, ld) matrix x_pad (spectrum in [0, 1]); state: 2 (iters + 2)"""
    with _on(x_pad.device) as st_:
        _check(load().dqc_purify_tc2(_ptr(x_pad), _ptr(tmp), x_pad.shape[-1], float(nocc), int(iters), float(tol), _ptr(state),
                                     st_), "dqc_purify_tc2")
    return x_pad


def purify_tc2_persist(x_pad, tmp, nocc, iters, tol, state, ctl):
    """purify_tc2 as ONE persistent launch on one XCD (ld <= 256); ctl: 4-element int32 device scratch, ctl[2] != 0 afterwards =
    the kernel gave up and x_pad is not a projector (the caller's idempotency error shows it)"""
    assert ctl.dtype == torch.int32 and ctl.is_cuda and ctl.numel() >= 4
    with _on(x_pad.device) as st_:
        _check(load().dqc_purify_tc2_persist(_ptr(x_pad), _ptr(tmp), x_pad.shape[-1], float(nocc), int(iters), float(tol), _ptr(state),
                                             ctypes.c_void_p(ctl.data_ptr()), st_), "dqc_purify_tc2_persist")
    return x_pad


def projector_tc2(fock, nocc, iters, tol):
    """fock (n, n) symmetric, n <= 256 -> (P (n, n), err (0-dim)): ONE persistent launch (dqc_projector_tc2)"""
    n = fock.shape[-1]
    p = torch.empty((n, n), dtype=torch.float64, device=fock.device)
    err = torch.empty(1, dtype=torch.float64, device=fock.device)
    work = torch.empty(int(load().dqc_projector_work_doubles(n, int(iters))), dtype=torch.float64, device=fock.device)
    with _on(fock.device) as st_:
        _check(load().dqc_projector_tc2(_ptr(p), _ptr(err), _ptr(fock.contiguous()), n, float(nocc), int(iters), float(tol), _ptr(work), st_),
               "dqc_projector_tc2")
    return p, err[0]


def purify_tc2_batched(x_pad, tmp, nocc, iters, tol, state):
    """in-place TC2 purification of the (nmol, ld, ld) batch x_pad; state (nmol, 2 (iters + 2))"""
    nmol, ld = x_pad.shape[0], x_pad.shape[-1]
    with _on(x_pad.device) as st_:
        _check(load().dqc_purify_tc2_batched(_ptr(x_pad), _ptr(tmp), ld, nmol, float(nocc), int(iters), float(tol), _ptr(state),
                                             st_), "dqc_purify_tc2_batched")
    return x_pad


def orth_factor_batched(y, g, out=None):
    """y (nmol, n, r), g (nmol, r, r) = y^T y -> (nmol, n, r) orthonormal bases (one launch)"""
    nmol, n, r = y.shape
    q = torch.empty_like(y) if out is None else out
    with _on(y.device) as st_:
        _check(load().dqc_orth_factor_batched(_ptr(q), _ptr(y.contiguous()), _ptr(g.contiguous()), n, r, nmol, st_),
               "dqc_orth_factor_batched")
    return q


def diis_solve(gram, m, out=None):
    """gram (nmol, H, H) Gram matrices of the stored error vectors (first m slots valid) -> Pulay coefficients (nmol, H)"""
    nmol, H, _ = gram.shape
    c = torch.empty((nmol, H), dtype=torch.float64, device=gram.device) if out is None else out
    with _on(gram.device) as st_:
        _check(load().dqc_diis_solve(_ptr(c), _ptr(gram), nmol, H, int(m), st_), "dqc_diis_solve")
    return c


def diis_solve_dev(gram, count, out):
    """diis_solve with the number of valid slots min(count, H) read from the device (count: 0-dim / 1-element int64 device tensor)"""
    nmol, H, _ = gram.shape
    assert count.dtype == torch.int64 and count.is_cuda
    with _on(gram.device) as st_:
        _check(load().dqc_diis_solve_dev(_ptr(out), _ptr(gram), nmol, H, ctypes.c_void_p(count.data_ptr()), st_), "dqc_diis_solve_dev")
    return out


def orth_factor(y, g):
    """y (n, r) = P Omega, g (r, r) = y^T y -> (n, r) orthonormal basis of range(P) (Cholesky QR in one launch)"""
    n, r = y.shape
    q = torch.empty_like(y)
    with _on(y.device) as st_:
        _check(load().dqc_orth_factor(_ptr(q), _ptr(y.contiguous()), _ptr(g.contiguous()), n, r, st_), "dqc_orth_factor")
    return q


def df_grad(grad, dcart, ccart, tab, orb_range, aux_range):
    """grad (natm, 3) += gradient of the density-fitted Coulomb energy; dcart (ncart, ncart), ccart (ncart) over the
    Cartesian basis of the whole concatenated table"""
    (s0, s1), (k0, k1) = orb_range, aux_range
    with _on(grad.device) as st_:
        _check(load().dqc_df_grad(_ptr(grad), _ptr(dcart), _ptr(ccart), *tab.args(), s0, s1, k0, k1, st_), "dqc_df_grad")
    return grad


def eri_dense(tiles, nao):
    out = torch.empty((nao,) * 4, dtype=torch.float64, device=tiles.device)
    with _on(tiles.device) as st_:
        _check(load().dqc_eri_tiles_to_dense(_ptr(out), _ptr(tiles), nao, st_), "dqc_eri_tiles_to_dense")
    return out


def jk_workspace(nao, device):
    return torch.empty(load().dqc_jk_work_doubles(nao), dtype=torch.float64, device=device)


def jk(tiles, dm_ao, work, with_k=True):
    """dm_ao (nao,nao) -> J, K (K = None if not with_k); AO basis, symmetrised"""
    nao = dm_ao.shape[-1]
    J = torch.empty((nao, nao), dtype=torch.float64, device=dm_ao.device)
    K = torch.empty_like(J) if with_k else None
    with _on(dm_ao.device) as st_:
        _check(load().dqc_jk_from_tiles(_ptr(J), _ptr(K), _ptr(tiles), _ptr(dm_ao.contiguous()), nao, _ptr(work),
                                        st_), "dqc_jk_from_tiles")
    return J, K


def fock_max_nao():
    return int(load().dqc_fock_max_nao())


def fock_factor(x, c, w, nao, ld):
    """the padded AO-basis factor pair (orb (ld, rp), orbt (rp, ld)) of D = C diag(w) C^T: L = X (C sqrt(w)) in one launch; None when
    the factor is wider than the density kernel's widest instantiation.  c: (north, r) with unit column stride, w: (r,) >= 0"""
    r = c.shape[1]
    rp = padded_norb(r)
    if rp == 0:
        return None
    assert (c.shape[1] == 1 or c.stride(1) == 1) and c.dtype == torch.float64 and w.is_contiguous()  # (rows c.stride(0) apart)
    orb = torch.empty((ld, rp), dtype=torch.float64, device=x.device)
    orbt = torch.empty((rp, ld), dtype=torch.float64, device=x.device)
    with _on(x.device) as st_:
        _check(load().dqc_fock_factor(_ptr(orb), _ptr(orbt), _ptr(x), ctypes.c_void_p(c.data_ptr()), int(c.stride(0)), _ptr(w), int(nao),
                                      int(x.shape[1]), int(r), int(ld), int(rp), st_), "dqc_fock_factor")
    return orb, orbt


def fock_orb2dm(x, c, w, nao, ld):
    """ao_orb2dm and its AO-basis factor in one launch -> dm (north, north) = C diag(w) C^T and the padded factor pair (orb, orbt) of
    L = X (C sqrt(w)); None when the factor is wider than the density kernel's widest instantiation"""
    r = c.shape[1]
    rp = padded_norb(r)
    if rp == 0:
        return None
    assert (c.shape[1] == 1 or c.stride(1) == 1) and c.dtype == torch.float64 and w.is_contiguous()  # (rows c.stride(0) apart)
    north = x.shape[1]
    dm = torch.empty((north, north), dtype=torch.float64, device=x.device)
    orb = torch.empty((ld, rp), dtype=torch.float64, device=x.device)
    orbt = torch.empty((rp, ld), dtype=torch.float64, device=x.device)
    with _on(x.device) as st_:
        _check(load().dqc_fock_orb2dm(_ptr(dm), _ptr(orb), _ptr(orbt), _ptr(x), ctypes.c_void_p(c.data_ptr()), int(c.stride(0)), _ptr(w),
                                      int(nao), int(north), int(r), int(ld), int(rp), st_), "dqc_fock_orb2dm")
    return dm, (orb, orbt)


def fock_prep(work, x, nao, with_k, dm=None, orb=None):
    """work <- symmetric AO density + zeroed accumulators: from the orthogonal-basis `dm` (north, north) and the orthogonaliser `x`
    (nao, north), or from the AO-basis factor `orb` (rows >= nao zero, rp columns): D_ao = orb orb^T"""
    north = x.shape[1]
    with _on(work.device) as st_:
        _check(load().dqc_fock_prep(_ptr(work), _ptr(dm), _ptr(x), _ptr(orb), 0 if orb is None else orb.shape[1], int(nao), int(north),
                                    1 if with_k else 0, st_), "dqc_fock_prep")


def jk_stream_prepared(tiles, nao, work, with_k):
    with _on(work.device) as st_:
        _check(load().dqc_jk_stream_prepared(_ptr(tiles), int(nao), _ptr(work), 1 if with_k else 0, st_),
               "dqc_jk_from_tiles" if with_k else "dqc_jk_from_tiles[J only]")


def fock_finish(work, x, nao, with_k, vxc_ao=None, core=None, want_j=False):
    """-> fock (north, north) = sym(X^T (J - K / 2 + V) X) + core, energies (2,) = [tr D J / 2, -tr D K / 4], J_ao or None"""
    north = x.shape[1]
    fock = torch.empty((north, north), dtype=torch.float64, device=work.device)
    en = torch.empty(2, dtype=torch.float64, device=work.device)
    jao = torch.empty((nao, nao), dtype=torch.float64, device=work.device) if want_j else None
    with _on(work.device) as st_:
        _check(load().dqc_fock_finish(_ptr(fock), _ptr(en), _ptr(jao), _ptr(work), _ptr(vxc_ao), 0 if vxc_ao is None else vxc_ao.shape[-1],
                                      _ptr(core), _ptr(x), int(nao), int(north), 1 if with_k else 0, st_), "dqc_fock_finish")
    return fock, en, jao


def jk_direct(tab, dm_ao, with_k=True):
    """direct SCF: J, K (K = None if not with_k) of one AO density straight from the shell quartets (no tile store)"""
    nao = dm_ao.shape[-1]
    J = torch.empty((nao, nao), dtype=torch.float64, device=dm_ao.device)
    K = torch.empty_like(J) if with_k else None
    with _on(dm_ao.device) as st_:
        _check(load().dqc_jk_direct(_ptr(J), _ptr(K), _ptr(dm_ao.contiguous()), *tab.args(), st_), "dqc_jk_direct")
    return J, K


def eri_pair_stats(tab, merge=True):
    """host-side size of the ERI pair tables (no GPU needed): dict(groups, pairs, primitive_pairs, primitive_quartets)"""
    out = (ctypes.c_longlong * 4)()
    _check(load().dqc_eri_pair_stats(*tab.args(), 1 if merge else 0, out), "dqc_eri_pair_stats")
    return {"groups": out[0], "pairs": out[1], "primitive_pairs": out[2], "primitive_quartets": out[3]}


class DirectContext:
    """screened direct SCF (include/dqc_amd.h: dqc_direct_*): pair tables and Schwarz bounds resident on the device.
    `jk(dm, with_k, tau)` skips the shell quartets whose contribution is bounded by `tau` (0: none skipped)."""

    def __init__(self, tab, device):
        self.device = torch.device(device)
        self.nao = tab.nao
        self._h = ctypes.c_void_p()
        with _on(self.device) as st_:
            _check(load().dqc_direct_create(ctypes.byref(self._h), *tab.args(), st_), "dqc_direct_create")

    def jk(self, dm_ao, with_k=True, tau=0.0, part=(0, 1)):
        """J, K (K = None if not with_k) of one AO density; part = (r, n): only every n-th block of shell quartets, starting
        with the r-th -- the partial sums of rank r when one molecule is spread over n GPUs (the caller all_reduces)"""
        J = torch.empty((self.nao, self.nao), dtype=torch.float64, device=dm_ao.device)
        K = torch.empty_like(J) if with_k else None
        with _on(dm_ao.device) as st_:
            _check(load().dqc_direct_jk_part(self._h, _ptr(J), _ptr(K), _ptr(dm_ao.contiguous()), float(tau), int(part[0]), int(part[1]),
                                             st_), "dqc_direct_jk_part")
        return J, K

    def stats(self):
        """(unique shell quartets, quartets launched, max |D|) of the last jk call"""
        a, b, d = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_double()
        _check(load().dqc_direct_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(d)), "dqc_direct_stats")
        return a.value, b.value, d.value

    def bounds(self, groups=False):
        """(Q (npairs,), shells (npairs, 2)): Schwarz bound sqrt(max |(ab|ab)|) of every shell pair, table order; groups=True adds
        the index of the GROUP pair each shell pair is screened with (s shells of one atom over the same exponents are evaluated
        together, with the largest bound of their members)"""
        n = int(load().dqc_direct_npairs(self._h))
        q, sh, gr = np.zeros(n), np.zeros((n, 2), dtype=np.int32), np.zeros(n, dtype=np.int32)
        _check(load().dqc_direct_bounds_groups(self._h, q.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                               sh.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), gr.ctypes.data_as(ctypes.POINTER(ctypes.c_int))),
               "dqc_direct_bounds_groups")
        return (q, sh, gr) if groups else (q, sh)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            with _on(self.device):
                load().dqc_direct_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tile_slice(nao, part, nparts):
    """(tile_begin, tile_end, doubles) of the `part`-th of `nparts` contiguous slices of the packed tile store, equal in DOUBLES to
    within a tile (the tiles differ in size: diagonal block pairs, the truncated last block row), i.e. equal streaming work"""
    L = load()
    nao = int(nao)
    nt = int(L.dqc_eri_tile_count(nao))
    tot = int(L.dqc_eri_tile_offset(nao, nt))

    def cut(k):  # first tile whose offset reaches k / nparts of the store (bisection on the monotone offset function)
        if k <= 0:
            return 0
        if k >= nparts:
            return nt
        want, lo, hi = tot * k // nparts, 0, nt
        while lo < hi:
            mid = (lo + hi) // 2
            if int(L.dqc_eri_tile_offset(nao, mid)) < want:
                lo = mid + 1
            else:
                hi = mid
        return lo

    t0, t1 = cut(part), cut(part + 1)
    return t0, t1, int(L.dqc_eri_tile_offset(nao, t1) - L.dqc_eri_tile_offset(nao, t0))


def eri_tiles_part(tab, device, t0, t1):
    """the tiles [t0, t1) of the packed store (one rank's slice when the store is spread over several GPUs)"""
    L = load()
    n = int(L.dqc_eri_tile_offset(tab.nao, t1) - L.dqc_eri_tile_offset(tab.nao, t0))
    tiles = torch.empty(max(n, 1), dtype=torch.float64, device=device)
    with _on(device) as st_:
        _check(L.dqc_eri_fill_tiles_part(_ptr(tiles), *tab.args(), int(t0), int(t1), st_), "dqc_eri_fill_tiles_part")
    return tiles


def jk_part(tiles_part, dm_ao, work, with_k, t0, t1):
    """PARTIAL J, K (the contributions of the tiles [t0, t1) held in `tiles_part`): the caller sums the ranks' parts"""
    nao = dm_ao.shape[-1]
    J = torch.empty((nao, nao), dtype=torch.float64, device=dm_ao.device)
    K = torch.empty_like(J) if with_k else None
    with _on(dm_ao.device) as st_:
        _check(load().dqc_jk_from_tiles_part(_ptr(J), _ptr(K), _ptr(tiles_part), _ptr(dm_ao.contiguous()), nao, _ptr(work),
                                             int(t0), int(t1), st_), "dqc_jk_from_tiles_part")
    return J, K


def jk_multi(tiles, dms_j, dms_k, work=None):
    """ONE pass over the tiles for several density matrices: dms_j (nj, nao, nao) -> J (nj, nao, nao), dms_k (nk, nao, nao)
    -> K (nk, nao, nao) (plain K, not -K/2); either may be None.  AO basis, symmetrised."""
    ref = dms_j if dms_j is not None else dms_k
    nao, dev = ref.shape[-1], ref.device
    nj = 0 if dms_j is None else dms_j.shape[0]
    nk = 0 if dms_k is None else dms_k.shape[0]
    need = load().dqc_jk_multi_work_doubles(nao, nj, nk)
    if work is None or work.numel() < need:
        work = torch.empty(need, dtype=torch.float64, device=dev)
    J = torch.empty((nj, nao, nao), dtype=torch.float64, device=dev) if nj else None
    K = torch.empty((nk, nao, nao), dtype=torch.float64, device=dev) if nk else None
    with _on(dev) as st_:
        _check(load().dqc_jk_from_tiles_multi(_ptr(J), _ptr(None if dms_j is None else dms_j.contiguous()), nj,
                                              _ptr(K), _ptr(None if dms_k is None else dms_k.contiguous()), nk,
                                              _ptr(tiles), nao, _ptr(work), st_), "dqc_jk_from_tiles_multi")
    return J, K


def eval_gto(tab, rgrid, deriv):
    """rgrid (ngrid,3) device -> (ngrid, lda) [deriv 0], (4, ngrid, lda) [deriv 1], (5, ngrid, lda) [deriv 2: + laplacian]
    or (10, ngrid, lda) [deriv 3: + xx xy xz yy yz zz]; lda = ao_stride(nao) (the array carries the kernels' slack, see ao_empty)"""
    ngrid = rgrid.shape[0]
    ncomp = {0: 0, 1: 4, 2: 5, 3: 10}[deriv]
    nc = max(ncomp, 1)
    lda = ao_stride(tab.nao)
    flat = torch.empty(int(load().dqc_ao_doubles(nc, int(ngrid), int(tab.nao))), dtype=torch.float64, device=rgrid.device)
    out = flat[:nc * ngrid * lda].view(nc, ngrid, lda)
    out = out[0] if ncomp == 0 else out
    with _on(rgrid.device) as st_:
        _check(load().dqc_eval_gto(deriv, _ptr(out), _ptr(rgrid.contiguous()), ngrid, *tab.args(), st_),
               "dqc_eval_gto")
    return out


def pad_matrix(m, ld):
    n = m.shape[-1]
    if n == ld:
        return m.contiguous()
    out = torch.zeros((ld, ld), dtype=m.dtype, device=m.device)
    out[:n, :n] = m
    return out


def grid_density(ao, nao, dm_pad, gga):
    """ao (ngrid, ld) or (4, ngrid, ld); dm_pad (ld, ld) -> rho (ngrid,), grho (3,ngrid) or None"""
    ncomp = 1 if ao.dim() == 2 else ao.shape[0]
    ngrid = ao.shape[-2]
    rho = torch.empty(ngrid, dtype=torch.float64, device=ao.device)
    grho = torch.empty((3, ngrid), dtype=torch.float64, device=ao.device) if gga else None
    with _on(ao.device) as st_:
        _check(load().dqc_grid_density(_ptr(rho), _ptr(grho), _ptr(ao), ncomp, ngrid, nao, _ptr(dm_pad), st_),
               "dqc_grid_density" if gga else "dqc_grid_density[value only]")
    return rho, grho


def padded_norb(norb):
    """factor width the low-rank density kernel is instantiated for (0: too wide, use grid_density)"""
    return int(load().dqc_padded_norb(int(norb)))


def pad_factor(l_ao, ld):
    """l_ao (nao, r) -> (orb (ld, rp), orbt (rp, ld)) zero padded, or None when r is too wide"""
    nao, r = l_ao.shape
    rp = padded_norb(r)
    if rp == 0:
        return None
    orb = torch.zeros((ld, rp), dtype=torch.float64, device=l_ao.device)
    orb[:nao, :r] = l_ao
    return orb, orb.t().contiguous()


def grid_density_lr(ao, nao, factor, gga):
    """density of D = L L^T from the padded factor pair of `pad_factor` -> rho, grho (same as grid_density)"""
    orb, orbt = factor
    ncomp = 1 if ao.dim() == 2 else ao.shape[0]
    ngrid = ao.shape[-2]
    rho = torch.empty(ngrid, dtype=torch.float64, device=ao.device)
    grho = torch.empty((3, ngrid), dtype=torch.float64, device=ao.device) if gga else None
    with _on(ao.device) as st_:
        _check(load().dqc_grid_density_lr(_ptr(rho), _ptr(grho), _ptr(ao), ncomp, ngrid, nao, _ptr(orb), _ptr(orbt),
                                          orb.shape[1], st_), "dqc_grid_density_lr" if gga else "dqc_grid_density_lr[value only]")
    return rho, grho


def grid_density_lr_pol(ao, nao, factor_u, factor_d):
    """both spin densities of D_u = L_u L_u^T, D_d = L_d L_d^T from ONE pass over the AO matrix (GGA form): padded factor pairs of
    `pad_factor`, brought to the same padded width (<= 64 columns each) -> rho (2, ngrid), grho (2, 3, ngrid); None when the
    widths do not fit"""
    rp = max(factor_u[0].shape[1], factor_d[0].shape[1])
    if rp > 64 or ao.dim() != 3 or ao.shape[0] < 4:
        return None

    def widen(f):
        orb, orbt = f
        if orb.shape[1] == rp:
            return orb, orbt
        o2 = torch.zeros((orb.shape[0], rp), dtype=orb.dtype, device=orb.device)
        o2[:, :orb.shape[1]] = orb
        return o2, o2.t().contiguous()
    (ou, otu), (od, otd) = widen(factor_u), widen(factor_d)
    orb = torch.cat([ou, od], dim=1).contiguous()
    orbt = torch.cat([otu, otd], dim=0).contiguous()
    ngrid = ao.shape[-2]
    rho = torch.empty((2, ngrid), dtype=torch.float64, device=ao.device)
    grho = torch.empty((2, 3, ngrid), dtype=torch.float64, device=ao.device)
    with _on(ao.device) as st_:
        _check(load().dqc_grid_density_lr_pol(_ptr(rho), _ptr(grho), _ptr(ao), ao.shape[0], ngrid, nao, _ptr(orb), _ptr(orbt), rp, st_),
               "dqc_grid_density_lr_pol")
    return rho, grho


def grid_xc_gradient_terms(ao, nao, b, c, w, vrho, u, grho, vtau=None):
    """per-point (ngrid, 3) and per-basis-function (nao, 3) grid sums of the XC nuclear gradient (dqc_amd/gradient.py) from one
    pass over the deriv-3 AO array `ao` (10, ngrid, lda); b, c[0..2]: (ngrid, ldb) = Phi D, d_i Phi D"""
    ngrid = ao.shape[-2]
    q = torch.empty((ngrid, 3), dtype=torch.float64, device=ao.device)
    per_ao = torch.empty((nao, 3), dtype=torch.float64, device=ao.device)
    b, c0, c1, c2 = b.contiguous(), c[0].contiguous(), c[1].contiguous(), c[2].contiguous()
    u, grho, w, vrho = u.contiguous(), grho.contiguous(), w.contiguous(), vrho.contiguous()
    vt = None if vtau is None else vtau.contiguous()
    with _on(ao.device) as st_:
        _check(load().dqc_grid_xc_gradient_terms(_ptr(q), _ptr(per_ao), _ptr(ao), ngrid, nao, _ptr(b), _ptr(c0), _ptr(c1), _ptr(c2),
                                                 b.shape[-1], _ptr(w), _ptr(vrho), _ptr(u), _ptr(grho), _ptr(vt), st_),
               "dqc_grid_xc_gradient_terms")
    return q, per_ao


def grid_density_lr_tau(ao, nao, factor):
    """rho, grad rho (3, ngrid), tau of D = L L^T from ONE pass over the four AO components (factor width <= 96 columns)"""
    orb = factor[0]
    ngrid = ao.shape[-2]
    rho = torch.empty(ngrid, dtype=torch.float64, device=ao.device)
    grho = torch.empty((3, ngrid), dtype=torch.float64, device=ao.device)
    tau = torch.empty(ngrid, dtype=torch.float64, device=ao.device)
    with _on(ao.device) as st_:
        _check(load().dqc_grid_density_lr_tau(_ptr(rho), _ptr(grho), _ptr(tau), _ptr(ao), ao.shape[0], ngrid, nao, _ptr(orb),
                                              orb.shape[1], st_), "dqc_grid_density_lr_tau")
    return rho, grho, tau


def xc_eval(terms, rho, grho, want_e=True, want_v=True):
    """terms: list of (coef, name).  -> edens, vrho, vgrad(3,n) (None where not requested/applicable)"""
    n = rho.shape[0]
    ids = (ctypes.c_int * len(terms))(*[XC_IDS[nm] for _, nm in terms])
    cfs = (ctypes.c_double * len(terms))(*[float(c) for c, _ in terms])
    e = torch.empty_like(rho) if want_e else None
    v = torch.empty_like(rho) if want_v else None
    vg = torch.empty((3, n), dtype=torch.float64, device=rho.device) if (want_v and grho is not None) else None
    with _on(rho.device) as st_:
        _check(load().dqc_xc_eval(_ptr(e), _ptr(v), _ptr(vg), _ptr(rho), _ptr(grho), n, ids, cfs, len(terms), st_),
               "dqc_xc_eval")
    return e, v, vg


def xc_eval_quad(terms, rho, grho, w, want_v=True):
    """potentials AND the quadrature E_xc = sum_i w_i e_i from one pass over the grid -> exc (result in exc[0]), vrho, vgrad (3, n) or None"""
    n = rho.shape[0]
    ids = (ctypes.c_int * len(terms))(*[XC_IDS[nm] for _, nm in terms])
    cfs = (ctypes.c_double * len(terms))(*[float(c) for c, _ in terms])
    exc = torch.empty(1025, dtype=torch.float64, device=rho.device)  # DQC_XC_QUAD_DOUBLES: result + per-block partials
    v = torch.empty_like(rho) if want_v else None
    vg = torch.empty((3, n), dtype=torch.float64, device=rho.device) if (want_v and grho is not None) else None
    with _on(rho.device) as st_:
        _check(load().dqc_xc_eval_quad(_ptr(exc), None, _ptr(v), _ptr(vg), _ptr(rho), _ptr(grho), _ptr(w), n, ids, cfs, len(terms),
                                       st_), "dqc_xc_eval_quad")
    return exc, v, vg


def xc_eval_pol(terms, rho_u, rho_d, grho_u, grho_d, want_e=True, want_v=True):
    """-> edens, (vrho_u, vrho_d), (vgrad_u, vgrad_d)   (None where not requested / LDA)"""
    n = rho_u.shape[0]
    ids = (ctypes.c_int * len(terms))(*[XC_IDS[nm] for _, nm in terms])
    cfs = (ctypes.c_double * len(terms))(*[float(c) for c, _ in terms])
    gga = grho_u is not None
    e = torch.empty_like(rho_u) if want_e else None
    vu = torch.empty_like(rho_u) if want_v else None
    vd = torch.empty_like(rho_u) if want_v else None
    gu = torch.empty((3, n), dtype=torch.float64, device=rho_u.device) if (want_v and gga) else None
    gd = torch.empty((3, n), dtype=torch.float64, device=rho_u.device) if (want_v and gga) else None
    with _on(rho_u.device) as st_:
        _check(load().dqc_xc_eval_pol(_ptr(e), _ptr(vu), _ptr(vd), _ptr(gu), _ptr(gd), _ptr(rho_u), _ptr(rho_d),
                                      _ptr(grho_u), _ptr(grho_d), n, ids, cfs, len(terms), st_), "dqc_xc_eval_pol")
    return e, (vu, vd), (gu, gd)


def xc_eval_mgga(terms, rho, grho, tau, want_e=True, want_v=True):
    """-> edens, vrho, vgrad (3,n), vtau"""
    n = rho.shape[0]
    ids = (ctypes.c_int * len(terms))(*[XC_IDS[nm] for _, nm in terms])
    cfs = (ctypes.c_double * len(terms))(*[float(c) for c, _ in terms])
    e = torch.empty_like(rho) if want_e else None
    v = torch.empty_like(rho) if want_v else None
    vg = torch.empty((3, n), dtype=torch.float64, device=rho.device) if want_v else None
    vt = torch.empty_like(rho) if want_v else None
    with _on(rho.device) as st_:
        _check(load().dqc_xc_eval_mgga(_ptr(e), _ptr(v), _ptr(vg), _ptr(vt), _ptr(rho), _ptr(grho), _ptr(tau), n, ids, cfs,
                                       len(terms), st_), "dqc_xc_eval_mgga")
    return e, v, vg, vt


def xc_eval_mgga_pol(terms, rho_u, rho_d, grho_u, grho_d, tau_u, tau_d, want_e=True, want_v=True):
    """spin-polarised meta-GGA correlation terms -> edens, (vrho_u, vrho_d), vgrad (3,n) shared by both spins, vtau shared"""
    n = rho_u.shape[0]
    ids = (ctypes.c_int * len(terms))(*[XC_IDS[nm] for _, nm in terms])
    cfs = (ctypes.c_double * len(terms))(*[float(c) for c, _ in terms])
    e = torch.empty_like(rho_u) if want_e else None
    vu = torch.empty_like(rho_u) if want_v else None
    vd = torch.empty_like(rho_u) if want_v else None
    vg = torch.empty((3, n), dtype=torch.float64, device=rho_u.device) if want_v else None
    vt = torch.empty_like(rho_u) if want_v else None
    with _on(rho_u.device) as st_:
        _check(load().dqc_xc_eval_mgga_pol(_ptr(e), _ptr(vu), _ptr(vd), _ptr(vg), _ptr(vt), _ptr(rho_u), _ptr(rho_d), _ptr(grho_u),
                                           _ptr(grho_d), _ptr(tau_u), _ptr(tau_d), n, ids, cfs, len(terms), st_),
               "dqc_xc_eval_mgga_pol")
    return e, (vu, vd), vg, vt


def grid_density_pair(ao_a, ao_b, nao, dm_pad):
    """sum_ij a_gi D_ij b_gj on single-component (ngrid, ld) arrays"""
    ngrid = ao_a.shape[0]
    out = torch.empty(ngrid, dtype=torch.float64, device=ao_a.device)
    with _on(ao_a.device) as st_:
        _check(load().dqc_grid_density_pair(_ptr(out), _ptr(ao_a), _ptr(ao_b), ngrid, nao, _ptr(dm_pad), st_),
               "dqc_grid_density_pair")
    return out


def grid_vxc_pair(ao_a, ao_b, nao, w, v, what="dqc_grid_vxc_pair"):
    """sym( sum_g w_g v_g a_ga b_gb ) -> (ld, ld)   (`what`: the name this call is listed under by call_trace)"""
    ngrid, ld = ao_a.shape[0], padded_nao(nao)
    vm = torch.empty((ld, ld), dtype=torch.float64, device=ao_a.device)
    with _on(ao_a.device) as st_:
        _check(load().dqc_grid_vxc_pair(_ptr(vm), _ptr(ao_a), _ptr(ao_b), ngrid, nao, _ptr(w), _ptr(v), st_), what)
    return vm


def grid_vxc(ao, nao, w, vrho, vgrad):
    """-> (ld, ld) symmetric AO-basis Vxc matrix (zero padded)"""
    ncomp = 1 if ao.dim() == 2 else ao.shape[0]
    ngrid = ao.shape[-2]
    ld = padded_nao(nao)
    vm = torch.empty((ld, ld), dtype=torch.float64, device=ao.device)
    with _on(ao.device) as st_:
        _check(load().dqc_grid_vxc(_ptr(vm), _ptr(ao), ncomp, ngrid, nao, _ptr(w), _ptr(vrho), _ptr(vgrad), st_),
               "dqc_grid_vxc" if vgrad is not None else "dqc_grid_vxc[no gradient term]")
    return vm


class PartitionStream:
    """a HIP stream whose kernels run on compute units [cu_begin, cu_end) of every XCD only (dqc_stream_create_partition), wrapped
    as a torch stream (`.stream`) so that `with torch.cuda.stream(p.stream)` sends library calls and torch ops to it"""

    def __init__(self, device, cu_begin, cu_end):
        dev = torch.device(device)
        dev = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.device = dev
        self.cu_begin, self.cu_end = int(cu_begin), int(cu_end)
        h = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _check(load().dqc_stream_create_partition(ctypes.byref(h), self.cu_begin, self.cu_end, 0), "dqc_stream_create_partition")
            self.cus = int(load().dqc_stream_cus(h))
        self._handle = h
        self.stream = torch.cuda.ExternalStream(h.value, device=dev)

    def close(self):
        """hand the stream back to the process-wide pool (partition_stream): PyTorch's caching allocator keeps events and block
        records that name a stream for as long as the process lives, so a stream it has seen is never destroyed"""
        if self._handle is not None:
            self.stream.synchronize()
            _PART_POOL.setdefault((self.device.index, self.cu_begin, self.cu_end), []).append(self)


_PART_POOL = {}


def partition_stream(device, cu_begin, cu_end):
    """a PartitionStream from the pool of closed ones, or a new one"""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    free = _PART_POOL.get((idx, int(cu_begin), int(cu_end)))
    if free:
        return free.pop()
    return PartitionStream(torch.device("cuda", idx), cu_begin, cu_end)


def set_vxc_cus(ncu):
    """cap on the CUs the one-block-per-CU Vxc kernels occupy (0: none); returns the previous setting"""
    return int(load().dqc_set_vxc_cus(int(ncu)))


def device_cu_count(device=None):
    with torch.cuda.device(device):
        return int(load().dqc_device_cu_count())


def grid_vxc_raw(ao, nao, w, vrho, vgrad):
    """dqc_grid_vxc without the closing symmetrisation launch -> (raw (ld, ld) cross-block sums, their fixed-point scale or 0.0);
    for fock_finish_vraw"""
    ncomp = 1 if ao.dim() == 2 else ao.shape[0]
    ngrid = ao.shape[-2]
    ld = padded_nao(nao)
    vm = torch.empty((ld, ld), dtype=torch.float64, device=ao.device)
    sc = ctypes.c_double(0.0)
    with _on(ao.device) as st_:
        _check(load().dqc_grid_vxc_raw(_ptr(vm), _ptr(ao), ncomp, ngrid, nao, _ptr(w), _ptr(vrho), _ptr(vgrad), ctypes.byref(sc), st_),
               "dqc_grid_vxc" if vgrad is not None else "dqc_grid_vxc[no gradient term]")
    return vm, float(sc.value)


def fock_finish_vraw(work, x, nao, vxc_raw, vscale, core=None):
    """the Kohn-Sham finish on the raw sums of grid_vxc_raw -> fock (north, north), energies (2,)"""
    north = x.shape[1]
    fock = torch.empty((north, north), dtype=torch.float64, device=work.device)
    en = torch.empty(2, dtype=torch.float64, device=work.device)
    with _on(work.device) as st_:
        _check(load().dqc_fock_finish_vraw(_ptr(fock), _ptr(en), _ptr(work), _ptr(vxc_raw), int(vxc_raw.shape[-1]), float(vscale), _ptr(core),
                                           _ptr(x), int(nao), int(north), st_), "dqc_fock_finish")
    return fock, en


def probe_stream_read(buf):
    out = torch.zeros(1, dtype=torch.float64, device=buf.device)
    with _on(buf.device) as st_:
        _check(load().dqc_probe_stream_read(_ptr(buf), buf.numel(), _ptr(out), st_), "dqc_probe_stream_read")
    return out


def probe_mfma_f64_tflops(device, iters=4000):
    """measured fp64 MFMA ceiling of this GPU (TFLOP/s): 2048 waves x 8 independent accumulators"""
    out = torch.empty(512 * 256, dtype=torch.float64, device=device)
    L = load()
    with _on(device) as st_:
        _check(L.dqc_probe_mfma_f64(_ptr(out), 100, st_), "dqc_probe_mfma_f64")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with _on(device) as st_:
        _check(L.dqc_probe_mfma_f64(_ptr(out), iters, st_), "dqc_probe_mfma_f64")
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * 16 * 16 * 4 * 8 * iters * 2048 / (e0.elapsed_time(e1) * 1e-3) / 1e12


def probe_hbm_read_gbs(device, nbytes=2 << 30):
    """measured streaming-read bandwidth (GB/s) over a buffer larger than the Infinity Cache"""
    buf = torch.empty(nbytes // 8, dtype=torch.float64, device=device).normal_()
    for _ in range(5):  # (the first launches after an idle spell run at a lower clock: 6.05 then 6.25 TB/s)
        probe_stream_read(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        probe_stream_read(buf)
    e1.record()
    torch.cuda.synchronize()
    return 10.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9
