"""HamiltonMI355 -- the drop-in boundary: same constructor, method names, argument meaning, basis
convention (everything in the orthogonalised basis), return types and error behaviour as the reference's
HamiltonCGTO (dqc/hamilton/hcgto.py:19-558) / BaseHamilton (dqc/hamilton/base_hamilton.py:11-279), with the
arithmetic done by the gfx950 library:

    build()        S, T, V_nuc         -> dqc_int1e            (hcgto.py:108-114)
                   (ij|kl)             -> dqc_eri_fill_tiles   (hcgto.py:129; stays on the device as
                                          8-fold-unique tiles instead of a dense, orthogonalised nao^4 tensor)
    setup_grid()   AO (+gradient)      -> dqc_eval_gto         (hcgto.py:152-186)
    get_elrep / get_exchange           -> dqc_jk_from_tiles    (hcgto.py:204-241)
    get_vxc / get_e_xc                 -> dqc_grid_density, dqc_xc_eval, dqc_grid_vxc   (hcgto.py:260-269, 320-328)

The reference orthogonalises the ERI tensor once (convert4, O(nao^5)); here the ERIs stay in the AO basis
and the density / J / K / Vxc matrices are transformed instead (D_ao = X D X^T, J = X^T J_ao X), which is
algebraically identical.
"""
from typing import List, Optional

import os

import torch

from . import lib
from .basis import make_tables
from .linop import LinearOperator
from .utils.datastruct import AtomCGTOBasis, SpinParam, ValGrad
from .xc import LibXC

# Coulomb side streams (round 6, dqc_amd.batch.CuPartition): a Fock build that runs on a registered "grid" stream sends its
# Coulomb pass over the ERI tiles to the stream registered beside it -- a HIP stream confined to OTHER compute units
# (dqc_stream_create_partition), so that the HBM-bound tile stream of this molecule runs beside the matrix-core-bound grid pass
# instead of before it.  Key: (device index, stream handle).
_COULOMB_SIDE = {}


def register_coulomb_side_stream(grid_stream, side_stream):
    """builds whose CURRENT stream is `grid_stream` run their Coulomb tile pass on `side_stream` (None: forget the pair)"""
    key = (grid_stream.device.index, grid_stream.cuda_stream)
    if side_stream is None:
        _COULOMB_SIDE.pop(key, None)
    else:
        _COULOMB_SIDE[key] = side_stream



try:  # inside a DQC installation the class IS a BaseHamilton (isinstance checks of dqc.qccalc / dqc.system pass)
    from dqc.hamilton.base_hamilton import BaseHamilton as _Base
except Exception:  # dqc (or one of its dependencies: xitorch, dqclibs) is absent: stand alone
    _Base = object


_SIDE_STREAMS = {}


def _side_stream(dev):
    """one side stream per device for setup work that overlaps the main stream (ERI tile fill)"""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


class HamiltonMI355(_Base):
    def __init__(self, atombases: List[AtomCGTOBasis], spherical: bool = True, df=None, efield=None,
                 vext: Optional[torch.Tensor] = None, cache=None, orthozer: bool = True,
                 aoparamzer: str = "qr", device="cuda"):
        if not spherical:
            raise NotImplementedError("only spherical AOs (the reference default) are implemented on MI355X")
        self._dfoptions = df
        # efield: tuple of flattened tensors (E_d,), (E_d, dE_{d1 d2}) as Mol passes them (mol.py:456-474); hcgto.py:117-125
        if efield is not None:
            if isinstance(efield, torch.Tensor):
                efield = (efield,)
            if len(efield) > 2:
                raise NotImplementedError("electric-field terms beyond the field gradient (int1e r0r0r0...) are not implemented")
            for i, ef in enumerate(efield):
                assert ef.numel() == 3 ** (i + 1), "The %d-th tuple element of efield must have %d elements" % (i, 3 ** (i + 1))
        self._efield = efield
        if aoparamzer not in ("qr", "matexp"):
            raise RuntimeError("Unknown ao parameterizer: %s. Available options are: ['qr', 'matexp']" % aoparamzer)
        lib.load()  # fail loudly if the HIP library is missing
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise lib.DqcAmdError("HamiltonMI355 needs a ROCm device ('cuda'); there is no CPU path")
        self.dtype = torch.float64
        self.atombases = atombases
        self.spherical = spherical
        atm, bas, env, zs = make_tables(atombases)
        self._tab = lib.Tables(atm, bas, env)
        self._zs = zs
        self._nao_ao = self._tab.nao
        self._ld = lib.padded_nao(self._nao_ao)    # rows / columns of the zero-padded square matrices the kernels take
        self._lda = lib.ao_stride(self._nao_ao)   # row stride of the AO-on-grid arrays

        self._ovlp_ao = lib.int1e("ovlp", self._tab, self.device)
        # the orthogonaliser (eigh of S: 5 ms of small rocSOLVER kernels with the host waiting) is formed at first use, so that
        # build() can put the ERI tile fill on a side stream BEFORE it and the two overlap
        self._X = None
        self._fill_done = None
        # where the two-electron integrals live: "tiles" (8-fold-unique tiles resident in HBM, streamed per Fock build), "direct"
        # (nothing stored, the shell quartets are re-evaluated per build: dqc_jk_direct) or "auto" (tiles when they fit)
        self._eri_mode = os.environ.get("DQC_AMD_ERI", "auto")
        if self._eri_mode not in ("auto", "tiles", "direct"):
            raise RuntimeError("DQC_AMD_ERI must be auto, tiles or direct")
        self._direct = False
        self.orthogonalized = bool(orthozer)  # False: the API's matrices live in the raw AO basis (overlap != 1)
        if df is None:
            self._df = None
        else:  # hcgto.py:60-64
            from .df import DFMI355
            self._df = DFMI355(df, atombases, self._orthozer, self.device)
        self._vext = vext
        self.is_grid_set = False
        self.is_ao_set = False
        self.is_grad_ao_set = False
        self.is_built = False
        self.xc = None
        self.xcfamily = 1
        self._fuse_k = False
        self._jk_cache = None
        self._dm_factor = None
        self._w_checked = None
        # which density kernel the grid passes took: "factor" (rank-n_occ kernel, D = ao_orb2dm(...) recognised), "dense" (anonymous
        # full matrix: 0.84 instead of 0.50 ms on a 20-atom molecule).  A caller that forms more than two density matrices before
        # using them falls out of the two-entry memo silently otherwise -- read `grid_path_counts` to see it.
        self.grid_path_counts = {"factor": 0, "dense": 0}
        # DQC_AMD_DENSITY=dense forces the full-matrix density kernel (A/B timing, parity tests)
        self._lowrank_density = os.environ.get("DQC_AMD_DENSITY", "lr") != "dense"

    # ------------------------------------------------------------------ properties
    @property
    def _orthozer(self):
        """X with X^T S X = 1 (dqc/hamilton/orbconverter.py:67-116: eigh(S), eigenvalues below 1e-6 dropped)"""
        if self._X is None:
            if self.orthogonalized:
                ev, evec = torch.linalg.eigh(self._ovlp_ao)
                acc = ev > 1e-6  # dqc/hamilton/orbconverter.py:73
                self._X = evec[:, acc] * ev[acc] ** (-0.5)
            else:
                self._X = torch.eye(self._nao_ao, dtype=self.dtype, device=self.device)
        return self._X

    @property
    def _tiles(self):
        """the ERI tile store; a stream that reaches for it while the side-stream fill is still running waits for the fill"""
        ev = self._fill_done
        if ev is not None and not torch.cuda.is_current_stream_capturing():  # (a capture's warm-up pass has waited already)
            if ev.query():
                self._fill_done = None
            else:
                torch.cuda.current_stream(self.device).wait_event(ev)
        t = self._tiles_store
        if t is not None and not torch.cuda.is_current_stream_capturing():
            # the store was allocated on the setup side stream's pool: every stream that streams it (run_concurrent / lockstep side
            # streams) is recorded, so that the caching allocator does not hand the block to the next molecule's fill while a J / K
            # kernel of a dropped Hamiltonian is still reading it
            cur = torch.cuda.current_stream(self.device)
            seen = self.__dict__.setdefault("_tile_streams", set())
            if cur.cuda_stream not in seen:
                seen.add(cur.cuda_stream)
                t.record_stream(cur)
        return t

    @property
    def nao(self) -> int:
        return self._orthozer.shape[-1]

    @property
    def kpts(self):
        raise TypeError("Isolated molecule Hamiltonian does not have kpts property")

    @property
    def df(self):
        return self._df

    # ------------------------------------------------------------------ orbital converter
    def _convert2(self, mat):
        return self._orthozer.transpose(-2, -1) @ mat @ self._orthozer

    def _unconvert_dm(self, dm):
        return self._orthozer @ dm @ self._orthozer.transpose(-2, -1)

    # ------------------------------------------------------------------ setups
    def build(self):
        if self.is_built:  # idempotent: the integrals of a Hamiltonian do not change (a second caller finds them resident)
            return self
        tab, dev = self._tab, self.device
        fill_done = None
        if self._df is None:
            # exact J/K keeps the 8-fold-unique 8^4 tiles resident: ~nao^4 bytes (2 GB at nao 208, 31 GB at 412, > 288 GB
            # near nao 740).  Fail with a message instead of an allocator OOM deep inside the fill.
            need = lib.eri_store_doubles(tab.nao) * 8
            if self._pworld > 1 and self._eri_mode == "tiles":  # this rank's slice of the store
                self._tile_slice = lib.tile_slice(tab.nao, self._prank, self._pworld)
                need = self._tile_slice[2] * 8
            free, _total = torch.cuda.mem_get_info(dev)
            free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)  # blocks cached by the allocator are reusable
            mode = self._eri_mode
            if mode == "auto":
                mode = "tiles" if need <= free else "direct"
            if mode == "tiles" and need > free:
                raise lib.DqcAmdError(
                    "the exact-J/K ERI tile store of this basis needs %.1f GB (nao = %d) but only %.1f GB of device memory "
                    "are free: use eri='direct' (DQC_AMD_ERI=direct: the integrals are re-evaluated in every Fock build) or the "
                    "density-fitted Coulomb operator (mol.densityfit(method='coulomb', auxbasis=...))" % (need / 1e9, tab.nao, free / 1e9))
            self._direct = mode == "direct"
            if self._eri_mode == "auto" and self._direct:
                import warnings
                warnings.warn("the ERI tile store of this basis (nao = %d: %.1f GB) does not fit the %.1f GB of device memory that are "
                              "free: direct SCF (integrals re-evaluated in every Fock build, no hipGraph / lockstep step); "
                              "DQC_AMD_ERI=tiles insists on the store" % (tab.nao, need / 1e9, free / 1e9))
            self.eri_mode_used = mode
        if self._df is None and self._direct:
            # direct SCF: no tile store; every J / K call re-evaluates the shell quartets -- those the
            # Schwarz bounds do not rule out (dqc_direct_*: tables and bounds stay on the device)
            self._tiles_store, self._jkwork = None, None
            self._dctx = lib.DirectContext(tab, dev)
            self._dinc = {}
        elif self._df is None:
            # the fill (VALU-bound, 18 ms for a 20-atom molecule) runs on a side stream while this stream does the small
            # latency-bound setup work -- eigh of S, T, V, their conversions; the streams join at the end of build()
            main = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if self._tile_slice is not None:
                    self._tiles_store = lib.eri_tiles_part(tab, dev, self._tile_slice[0], self._tile_slice[1])
                else:
                    self._tiles_store = lib.eri_tiles(tab, dev)
                fill_done = torch.cuda.Event()
                fill_done.record(side)
            self._tiles_store.record_stream(main)
            self._fill_done = fill_done  # whoever touches the tiles first waits for it (the `_tiles` property)
            self._jkwork = lib.jk_workspace(self._nao_ao, dev)
        kin = lib.int1e("kin", tab, dev)
        nuc = lib.int1e("nuc", tab, dev, self._zs)
        self.olp_mat = self._convert2(self._ovlp_ao)
        if self._efield is not None:  # hcgto.py:117-125: sum_n  (1/n!) E^(n) . <mu| r0^n |nu>
            fac = 1.0
            for i, ef in enumerate(self._efield):
                fac *= i + 1
                mats = lib.int1e("r0" * (i + 1), tab, dev)  # (3^(i+1), nao, nao)
                kin = kin + torch.einsum("dab,d->ab", mats, ef.reshape(-1).to(device=dev, dtype=self.dtype)) / fac
        self.kinnucl_mat = self._convert2(kin + nuc)
        self.nucl_mat = self._convert2(nuc)
        if self._df is not None:  # hcgto.py:133-135
            self._df.build()
        self.is_built = True
        if self._vext is not None:  # hcgto.py:144-146
            self.kinnucl_mat = self.kinnucl_mat + self.get_vext(self._vext.to(self.device)).fullmatrix()
        return self

    def setup_grid(self, grid, xc=None) -> None:
        family = 1 if xc is None else xc.family
        if family not in (1, 2, 4):
            raise RuntimeError("unknown xc family %s" % family)
        if self.is_grid_set and getattr(self, "grid", None) is grid and family == self.xcfamily:
            if xc is not self.xc:  # another functional on the resident AO values: what was remembered of the old one goes
                self._energy_memo = None
                self._jk_cache = None
            self.xc = xc  # same grid, same derivative level: the AO values are already resident
            return
        self.xc = xc
        self.xcfamily = family
        self.grid = grid
        assert grid.coord_type == "cart"
        self.rgrid = grid.get_rgrid().to(self.device)
        self.dvolume = grid.get_dvolume().to(self.device).contiguous()
        # Points of weight EXACTLY zero are dropped from the resident arrays (round 6).  The reference's Becke partition zeroes the
        # weight of every point that fails its `mu < 0.74` cut (multiatoms_grid.py:233-262: p[ia] stays 0) -- 3.0 % of the sg3 grid
        # of a 20-atom molecule -- and such a point adds exactly nothing to any quadrature this class forms (E_xc, Vxc, vext, the
        # electron count): same sums, fewer AO rows to evaluate, store and stream in every grid pass.  `rgrid`, `dvolume`, `basis`,
        # `grad_basis` and the per-point arrays of `_dm2densinfo` then hold the LIVE points only; `ngrid_full` and `live_index`
        # (None: nothing dropped) relate them to the caller's grid, and get_vext indexes its argument.  DQC_AMD_PRUNE_GRID=0 keeps all.
        self.ngrid_full = int(self.rgrid.shape[0])
        self.live_index = None
        if os.environ.get("DQC_AMD_PRUNE_GRID", "1") != "0" and self.ngrid_full > 0:
            cached = getattr(grid, "_live_index_cache", None)
            if cached is None or cached[0].device != self.dvolume.device:
                idx = torch.nonzero(self.dvolume != 0).squeeze(1)
                cached = (idx, int(idx.shape[0]))  # (one device -> host read per grid object)
                try:
                    grid._live_index_cache = cached
                except AttributeError:
                    pass
            if cached[1] < self.ngrid_full:
                self.live_index = cached[0]
                self.rgrid = self.rgrid[self.live_index].contiguous()
                self.dvolume = self.dvolume[self.live_index].contiguous()
        if self._pworld > 1:  # this rank's contiguous slab of the points (equal work per point: equal slabs)
            g = self.rgrid.shape[0]
            lo, hi = g * self._prank // self._pworld, g * (self._prank + 1) // self._pworld
            self.rgrid, self.dvolume = self.rgrid[lo:hi].contiguous(), self.dvolume[lo:hi].contiguous()
        deriv = {1: 0, 2: 1, 4: 2}[self.xcfamily]
        self._ao = lib.eval_gto(self._tab, self.rgrid, deriv)  # (ngrid, ld), (4, ngrid, ld) or (5, ngrid, ld)
        self.is_grid_set = True
        self.is_ao_set = True
        self.is_grad_ao_set = deriv >= 1
        self.is_lapl_ao_set = deriv == 2
        self._ao_lapl_pm = None

    def grid_select(self, x):
        """a per-point array on the caller's grid (last axis ngrid_full) -> the same on the resident (live) points"""
        return x if self.live_index is None else x[..., self.live_index]

    def grid_scatter(self, x):
        """a per-point array on the resident points (last axis) -> on the caller's grid, zero at the dropped (zero-weight) points"""
        if self.live_index is None:
            return x
        out = torch.zeros(tuple(x.shape[:-1]) + (self.ngrid_full,), dtype=x.dtype, device=x.device)
        out[..., self.live_index] = x
        return out

    @property
    def basis(self):
        """(ngrid, nao) AO values, the attribute name of the reference (hcgto.py:168)"""
        a = self._ao if self._ao.dim() == 2 else self._ao[0]
        return a[:, :self._nao_ao]

    @property
    def grad_basis(self):
        """(3, ngrid, nao) AO gradients (hcgto.py:179)"""
        return self._ao[1:4, :, :self._nao_ao]

    # ------------------------------------------------------------------ Fock components
    def get_nuclattr(self):
        return LinearOperator.m(self.nucl_mat, is_hermitian=True)

    def get_kinnucl(self):
        return LinearOperator.m(self.kinnucl_mat, is_hermitian=True)

    def get_overlap(self):
        return LinearOperator.m(self.olp_mat, is_hermitian=True)

    def _batched(self, fcn, dm):
        if dm.dim() == 2:
            return fcn(dm)
        bshape = dm.shape[:-2]
        res = [fcn(d) for d in dm.reshape(-1, *dm.shape[-2:])]
        return torch.stack(res).reshape(*bshape, *res[0].shape)

    def _jk_orth(self, dm, need_k):
        """(J, K) in the orthogonalised basis for one (nao,nao) dm; one fused pass over the ERI tiles.
        The reference calls get_elrep(dm) and get_exchange(dm) back to back (dqc/qccalc/hf.py:198-199);
        the pair is memoised on the identity + version of the dm tensor so the tiles are streamed once."""
        # the cache holds `dm` itself (identity + in-place version), never a raw pointer: the caching
        # allocator reuses addresses of freed tensors
        c = self._jk_cache
        if c is not None and c[0] is dm and c[3] == dm._version and (c[2] is not None or not need_k):
            return c[1], c[2]
        dao = self._unconvert_dm(dm)
        with_k = need_k or self._fuse_k
        J, K = self._jk_ao(dao, with_k)
        J = self._convert2(J)
        J = (J + J.transpose(-2, -1)) * 0.5
        if K is not None:
            K = self._convert2(K)
            K = (K + K.transpose(-2, -1)) * 0.5
        self._jk_cache = (dm, J, K, dm._version)
        return J, K

    def use_direct_eri(self, on: bool = True):
        """before build(): keep no ERI tile store and re-evaluate the integrals in every Fock build (direct SCF) -- for bases
        whose ~nao^4 bytes of tiles do not fit the GPU; `on=False` insists on the tile store"""
        if self.is_built:
            raise RuntimeError("use_direct_eri must be called before build()")
        self._eri_mode = "direct" if on else "tiles"
        return self

    # ------------------------------------------------------------------ one molecule over several GPUs
    _pg, _prank, _pworld = None, 0, 1
    _tile_slice = None  # (tile_begin, tile_end, doubles) when the tile store is spread over the ranks

    def shard_over(self, group=None, eri="direct"):
        """before build() / setup_grid(), on EVERY rank of `group` (default: the world group): this one molecule is spread
        over the ranks' GPUs -- rank r evaluates every world-th block of shell quartets of the screened direct J / K pass
        (dqc_direct_jk_part) and the r-th contiguous slab of the grid (its AO matrix is 1 / world of the whole), and the
        partial J, K, Vxc matrices and the E_xc quadrature are summed with ONE all_reduce each per Fock build (RCCL; n^2
        doubles: latency-bound).  The SCF driver runs on every rank (SPMD); rank 0's convergence scalars are broadcast once
        per iteration so that all ranks take the same decisions.  Tile store and density fitting are not sharded: the
        Hamiltonian switches to direct SCF.  Nuclear gradients of a sharded Hamiltonian are not provided.
        eri="tiles": the STORED path instead -- the packed tile store itself is spread over the ranks' memories (rank r keeps
        and streams the r-th contiguous slice of the tiles: 8 x 288 GB hold the store of ~1200 basis functions;
        dqc_eri_fill_tiles_part / dqc_jk_from_tiles_part), same all_reduce of the partial J / K."""
        import torch.distributed as dist
        if self.is_built or self.is_grid_set:
            raise RuntimeError("shard_over must be called before build() and setup_grid()")
        if self._df is not None or self._vext is not None:
            raise RuntimeError("a density-fitted Hamiltonian / one with an external potential on the grid cannot be sharded")
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("shard_over needs an initialised torch.distributed process group")
        self._pg = group
        self._prank, self._pworld = dist.get_rank(group), dist.get_world_size(group)
        if eri not in ("direct", "tiles"):
            raise RuntimeError("shard_over: eri must be 'direct' or 'tiles'")
        if lib.load().dqc_get_deterministic():
            # (tiles: the fixed-point scale reads the diagonal tiles of the WHOLE store; direct: fp64 atomics only)
            raise RuntimeError("shard_over: the deterministic mode (dqc_set_deterministic) covers unsharded Hamiltonians only")
        self._eri_mode = eri
        self._tile_slice = None
        return self

    @property
    def sharded(self):
        return self._pworld > 1

    _deferred = None  # a list while a build collects its partial results for ONE all_reduce (_allsum_deferred)

    def _allsum(self, t):
        """sum of the ranks' partial results, in place (no-op for an unsharded Hamiltonian)"""
        if self._pworld > 1 and t is not None:
            if self._deferred is not None:
                self._deferred.append(t)
                return t
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._pg)
        return t

    def _allsum_flush(self):
        """the partial results collected since `_deferred = []` summed over the ranks with one all_reduce of their concatenation
        (J, Vxc and the E_xc quadrature of a Fock build are each latency-bound on their own)"""
        ts, self._deferred = self._deferred, None
        if ts:
            import torch.distributed as dist
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self._pg)
            o = 0
            for t in ts:
                t.copy_(flat[o:o + t.numel()].reshape(t.shape))
                o += t.numel()

    def sync_scalars(self, t):
        """rank 0's copy of a small device tensor on every rank (the SCF driver's per-iteration host read): all ranks decide alike"""
        if self._pworld > 1:
            import torch.distributed as dist
            src = dist.get_global_rank(self._pg, 0) if self._pg is not None else 0
            t = t.contiguous()
            dist.broadcast(t, src=src, group=self._pg)
        return t

    # direct SCF: shell quartets bounded by this much are skipped (DQC_AMD_DIRECT_TAU; 0 = none), and a build adds
    # G[D - D_prev] to the previous one for at most this many calls in a row before it is redone from D itself
    _DIRECT_TAU = float(os.environ.get("DQC_AMD_DIRECT_TAU", "1e-13"))
    _DIRECT_RESET = int(os.environ.get("DQC_AMD_DIRECT_RESET", "12"))

    def _jk_direct(self, dao, with_k, slot):
        """J, K of one AO density from the shell quartets (screened, dqc_direct_jk).  J and K are linear in D: the builds of
        one `slot` (one density stream of the SCF loop) are INCREMENTAL -- G[D] = G[D_prev] + G[D - D_prev] -- so that the
        density-weighted screening sees max |D - D_prev|, which falls as the SCF converges; every _DIRECT_RESET-th build of a
        slot (or a change of shape / of the K request, or a difference that is no smaller than the density) starts again from D"""
        tau = self._DIRECT_TAU
        part = (self._prank, self._pworld)
        st = self._dinc.get(slot)
        if tau > 0 and st is not None and st["n"] < self._DIRECT_RESET and st["k"] == with_k and st["d"].shape == dao.shape:
            dd = dao - st["d"]
            J, K = self._dctx.jk(dd, with_k, tau, part)
            self._allsum(J), self._allsum(K)
            J += st["J"]
            if with_k:
                K += st["K"]
            n = st["n"] + 1
        else:
            J, K = self._dctx.jk(dao, with_k, tau, part)
            self._allsum(J), self._allsum(K)
            n = 0
        self._direct_stats = self._dctx.stats()
        if tau > 0:
            self._dinc[slot] = {"d": dao.clone(), "J": J.clone(), "K": None if K is None else K.clone(), "k": with_k, "n": n}
        return J, K

    def _jk_ao(self, dao, with_k, slot=0):
        """(J, K or None) of one AO-basis density: from the resident tiles, or directly from the shell quartets"""
        if self._direct:
            return self._jk_direct(dao, with_k, slot)
        if self._tile_slice is not None:  # the store is spread over the ranks: stream this rank's slice, add the parts up
            J, K = lib.jk_part(self._tiles, dao, self._jkwork, with_k, self._tile_slice[0], self._tile_slice[1])
            return self._allsum(J), self._allsum(K)
        return lib.jk(self._tiles, dao, self._jkwork, with_k)

    def _fused_fock_ok(self, dm):
        """the small-matrix ends of this build through the fused kernels of csrc/fock.hip (dqc_fock_prep / dqc_fock_finish)?"""
        if self._df is not None or self._direct or self._tile_slice is not None or not isinstance(dm, torch.Tensor):
            return False
        if dm.dim() != 2 or not dm.is_cuda or dm.dtype != torch.float64 or os.environ.get("DQC_AMD_FUSED_FOCK", "1") == "0":
            return False
        x = self._orthozer
        return self._nao_ao <= lib.fock_max_nao() and x.is_contiguous() and x.shape[0] == self._nao_ao

    def _sym_orth(self, m_ao):
        m = self._convert2(m_ao)
        return (m + m.transpose(-2, -1)) * 0.5

    def _multi_work(self, nj, nk):
        need = lib.load().dqc_jk_multi_work_doubles(self._nao_ao, nj, nk)
        w = getattr(self, "_jkwork_multi", None)
        if w is None or w.numel() < need:
            w = self._jkwork_multi = torch.empty(need, dtype=torch.float64, device=self.device)
        return w

    def _jk_many(self, dms_j, dms_k, j_is_sum_of_k=False):
        """Coulomb matrices of the stacked (nj, nao, nao) `dms_j` and exchange matrices K of `dms_k`, all from ONE pass over
        the ERI tiles (dqc_jk_from_tiles_multi); orthogonalised basis in and out."""
        dj = None if dms_j is None else self._unconvert_dm(dms_j)
        dk = None if dms_k is None else self._unconvert_dm(dms_k)
        if self._direct:  # one pass over the shell quartets per density
            if j_is_sum_of_k and dj is not None and dk is not None and dj.shape[0] == 1:
                # unrestricted Hartree-Fock: J[D_u + D_d] = J[D_u] + J[D_d], and a J + K pass yields J of its density anyway:
                # two passes over the shell quartets instead of three
                jk = [self._jk_direct(d, True, ("k", i)) for i, d in enumerate(dk)]
                J = sum(x[0] for x in jk).unsqueeze(0)
                K = torch.stack([x[1] for x in jk])
                return self._sym_orth(J), self._sym_orth(K)
            J = None if dj is None else torch.stack([self._jk_direct(d, False, ("j", i))[0] for i, d in enumerate(dj)])
            K = None if dk is None else torch.stack([self._jk_direct(d, True, ("k", i))[1] for i, d in enumerate(dk)])
            return (None if J is None else self._sym_orth(J)), (None if K is None else self._sym_orth(K))
        if self._tile_slice is not None:  # (the several-densities-per-pass kernel streams the whole store only)
            if j_is_sum_of_k and dj is not None and dk is not None and dj.shape[0] == 1:
                jk = [self._jk_ao(d, True) for d in dk]
                return self._sym_orth(sum(x[0] for x in jk).unsqueeze(0)), self._sym_orth(torch.stack([x[1] for x in jk]))
            J = None if dj is None else torch.stack([self._jk_ao(d, False)[0] for d in dj])
            K = None if dk is None else torch.stack([self._jk_ao(d, True)[1] for d in dk])
            return (None if J is None else self._sym_orth(J)), (None if K is None else self._sym_orth(K))
        J, K = lib.jk_multi(self._tiles, dj, dk, self._multi_work(0 if dj is None else dj.shape[0], 0 if dk is None else dk.shape[0]))
        return (None if J is None else self._sym_orth(J)), (None if K is None else self._sym_orth(K))

    def get_elrep(self, dm):
        if not self.is_built:
            raise RuntimeError("Please call `build()` before `get_elrep`")
        if self._df is not None:  # hcgto.py:212-214
            return self._df.get_elrep(dm)
        if dm.dim() > 2:  # batch dimensions (base_hamilton.py:92-93): every density rides on the same tile pass
            flat = dm.reshape(-1, *dm.shape[-2:])
            mat = self._jk_many(flat, None)[0].reshape(dm.shape)
        else:
            mat = self._jk_orth(dm, False)[0]
        return LinearOperator.m(mat, is_hermitian=True)

    def get_exchange(self, dm):
        """returns -K/2 (hcgto.py:234); SpinParam input uses K(2 D_sigma) per spin (hcgto.py:238-241)"""
        if self._df is not None:  # hcgto.py:229-230
            raise RuntimeError("Exact exchange cannot be computed with density fitting")
        if not self.is_built:
            raise RuntimeError("Please call `build()` before `get_exchange`")
        if isinstance(dm, SpinParam):
            if dm.u.dim() == 2 and dm.d.dim() == 2:  # both spins from one pass (memoised by get_elrep_exchange_pol)
                ku, kd = self._jk_pol(dm)[1]
                return SpinParam(u=LinearOperator.m(ku, is_hermitian=True), d=LinearOperator.m(kd, is_hermitian=True))
            return SpinParam(u=self.get_exchange(2 * dm.u), d=self.get_exchange(2 * dm.d))
        self._fuse_k = True
        if dm.dim() > 2:
            flat = dm.reshape(-1, *dm.shape[-2:])
            mat = -0.5 * self._jk_many(None, flat)[1].reshape(dm.shape)
        else:
            mat = -0.5 * self._jk_orth(dm, True)[1]
        return LinearOperator.m(mat, is_hermitian=True)

    def _jk_pol(self, dm):
        """unrestricted Hartree-Fock: J[D_u + D_d], -K[2 D_u]/2, -K[2 D_d]/2 (hcgto.py:238-241, hf.py:93-103, 198-199) from
        one pass over the tiles; memoised on the identity + version of the two spin matrices so that the reference's call
        sequence get_elrep(dm.u + dm.d), get_exchange(dm) streams the tiles once when routed through here"""
        c = getattr(self, "_jkpol_cache", None)
        if c is not None and c[0] is dm.u and c[1] is dm.d and c[2] == (dm.u._version, dm.d._version):
            return c[3], c[4]
        J, K = self._jk_many((dm.u + dm.d).unsqueeze(0), torch.stack([dm.u, dm.d]), j_is_sum_of_k=True)
        # K[2 D] = 2 K[D] and the operator is -K/2: the two factors cancel
        out = (J[0], (-K[0], -K[1]))
        self._jkpol_cache = (dm.u, dm.d, (dm.u._version, dm.d._version), out[0], out[1])
        return out

    def get_elrep_exchange_pol(self, dm: SpinParam):
        """(J[D_u + D_d], SpinParam(-K[2 D_u]/2, -K[2 D_d]/2)) as plain tensors in the orthogonalised basis: the three
        operators an unrestricted Hartree-Fock Fock build needs (hf.py:93-103), one tile pass"""
        J, (ku, kd) = self._jk_pol(dm)
        return J, SpinParam(u=ku, d=kd)

    def get_vext(self, vext):
        if not self.is_ao_set:
            raise RuntimeError("Please call `setup_grid(grid, xc)` to call this function")
        ao = self._ao if self._ao.dim() == 2 else self._ao[0]
        zero = None
        if vext.shape[-1] == self.ngrid_full and self._pworld == 1 and self.live_index is not None:
            vext = vext.to(self.device)[..., self.live_index]  # (the caller's array lives on the caller's grid)
        mat = self._batched1(lambda v: lib.grid_vxc(ao, self._nao_ao, self.dvolume, v.contiguous(), zero), vext)
        mat = self._convert2(mat[..., :self._nao_ao, :self._nao_ao])
        mat = (mat + mat.transpose(-2, -1)) * 0.5
        return LinearOperator.m(mat, is_hermitian=True)

    def _batched1(self, fcn, v):
        if v.dim() == 1:
            return fcn(v)
        res = [fcn(x) for x in v.reshape(-1, v.shape[-1])]
        return torch.stack(res).reshape(*v.shape[:-1], *res[0].shape)

    def get_vxc(self, dm):
        assert self.xc is not None, "Please call .setup_grid with the xc object"
        if isinstance(dm, SpinParam):  # polarised branch of hcgto.py:260-269
            assert dm.u.dim() == 2, "batched polarised densities are not supported"
            densinfo = self._dm2densinfo_pol(dm)
            potinfo = self.xc.get_vxc(densinfo)
            return SpinParam(u=LinearOperator.m(self._get_vxc_from_potinfo(potinfo.u), is_hermitian=True),
                             d=LinearOperator.m(self._get_vxc_from_potinfo(potinfo.d), is_hermitian=True))

        def one(d):
            densinfo = self._dm2densinfo(d)
            potinfo = self.xc.get_vxc(densinfo)
            return self._get_vxc_from_potinfo(potinfo)

        return LinearOperator.m(self._batched(one, dm), is_hermitian=True)

    # ------------------------------------------------------------------ interface to dm
    def ao_orb2dm(self, orb, orb_weight):
        """hcgto.py:272-281.  The factor L = orb sqrt(w) of the returned matrix is remembered (keyed on the identity
        + in-place version of the result) so that the grid pass can use the rank-n_occ density kernel."""
        if (orb.dim() == 2 and orb_weight.dim() == 1 and self._lowrank_density and orb.is_cuda and orb.dtype == torch.float64
                and (orb.shape[1] == 1 or orb.stride(1) == 1) and orb_weight.is_contiguous() and 0 < orb.shape[1] <= 128 and self._X is not None
                and self._X.is_contiguous() and orb.shape[0] == self._X.shape[1] and self._nao_ao <= lib.fock_max_nao()
                and os.environ.get("DQC_AMD_FUSED_FOCK", "1") != "0"):
            # D and its AO-basis factor L = X C sqrt(w) from ONE launch (csrc/fock.hip) instead of multiply + GEMM here and
            # sqrt + multiply + GEMM + padding copies at the factor's first use; the factor is only USED after the occupations were
            # checked to be >= 0 (_factor_of), as before
            dm, pair = lib.fock_orb2dm(self._X, orb, orb_weight, self._nao_ao, self._ld)
            self._dm_factor = ([[dm, dm._version, orb, orb_weight, pair]] + (self._dm_factor or []))[:2]
            return dm
        orb_w = orb * orb_weight.unsqueeze(-2)
        dm = torch.matmul(orb, orb_w.transpose(-2, -1))
        if orb.dim() == 2 and self._lowrank_density:
            # occupations are >= 0 in every SCF caller; a negative weight simply disables the factor path.
            # Two entries are kept: the spin-up and spin-down matrices of an unrestricted iteration.
            self._dm_factor = ([[dm, dm._version, orb, orb_weight]] + (self._dm_factor or []))[:2]
        return dm

    def _weights_nonneg(self, w):
        """occupations >= 0?  Checked ONCE per weight tensor (identity + version): the check reads the device, and a
        device->host sync in every SCF iteration would drain the launch queue."""
        for c in self._w_checked or []:
            if c[0] is w and c[1] == w._version:
                return c[2]
        ok = not bool((w < 0).any())
        self._w_checked = ([(w, w._version, ok)] + (self._w_checked or []))[:2]
        return ok

    def _factor_of(self, dm):
        """list of padded AO-basis factor pairs (column panels of L) of `dm` if it came out of ao_orb2dm unmodified, else None"""
        for c in self._dm_factor or []:
            if c[0] is dm and c[1] == dm._version:
                if len(c) >= 4 and not isinstance(c[2], list):  # first use: orthogonal basis -> AO basis (X . orb sqrt(w)), padded for the kernel
                    orb, w = c[2], c[3]
                    if not self._weights_nonneg(w):
                        self._dm_factor.remove(c)
                        return None
                    if len(c) == 5:  # (ao_orb2dm's launch has formed the padded pair already)
                        c[2:] = [[c[4]]]
                        return c[2]
                    x = self._orthozer
                    if (orb.dim() == 2 and w.dim() == 1 and orb.is_cuda and (orb.shape[1] == 1 or orb.stride(1) == 1) and w.is_contiguous() and x.is_contiguous()
                            and 0 < orb.shape[1] <= 128 and self._nao_ao <= lib.fock_max_nao() and os.environ.get("DQC_AMD_FUSED_FOCK", "1") != "0"):
                        c[2:] = [[lib.fock_factor(x, orb, w, self._nao_ao, self._ld)]]  # (one launch: csrc/fock.hip)
                        return c[2]
                    l_ao = x @ (orb * torch.sqrt(w).unsqueeze(-2))
                    r = l_ao.shape[-1]
                    if lib.padded_norb(r) > 0:
                        c[2:] = [[lib.pad_factor(l_ao, self._ld)]]
                    else:  # wider than the kernel's widest instantiation: D = sum_p L_p L_p^T over column panels of L
                        npan = (r + 127) // 128
                        wid = (r + npan - 1) // npan
                        c[2:] = [[lib.pad_factor(l_ao[:, i:i + wid].contiguous(), self._ld) for i in range(0, r, wid)]]
                return c[2]
        return None

    def aodm2dens(self, dm, xyz):
        """density at arbitrary points (hcgto.py:283-299): dm (*BD, nao, nao), xyz (*BR, ndim) -> (*BRD), the batch dimensions
        of the two broadcast against each other as in the reference"""
        dao = self._unconvert_dm(dm.to(self.device))
        pts = xyz.reshape(-1, xyz.shape[-1]).to(self.device).contiguous()
        ao = lib.eval_gto(self._tab, pts, 0)  # (npts, ld)
        if dao.dim() == 2:
            dp = lib.pad_matrix((dao + dao.transpose(-2, -1)) * 0.5, self._ld)
            rho, _ = lib.grid_density(ao, self._nao_ao, dp, False)
            return rho.reshape(xyz.shape[:-1])
        basis = ao[:, :self._nao_ao].reshape(*xyz.shape[:-1], self._nao_ao)  # (*BR, nao)
        dens = torch.matmul(dao, basis.unsqueeze(-1))  # (*BRD, nao, 1)
        return torch.matmul(basis.unsqueeze(-2), dens).squeeze(-1).squeeze(-1)

    # ------------------------------------------------------------------ energies
    @staticmethod
    def _trdot(a, b):
        """tr(a b) over the last two dimensions as multiply + reduce (torch.einsum turns it into a 1 x n^2 x 1 GEMM, which
        takes a slow rocBLAS path for fp64: ~0.3 ms for n = 208)"""
        return (a * b.transpose(-2, -1)).sum((-2, -1))

    def get_e_hcore(self, dm):
        return self._trdot(self.kinnucl_mat, dm)

    def get_e_elrep(self, dm):
        e = self._memo_energy(dm, 2)
        if e is not None:
            return e
        return 0.5 * self._trdot(self.get_elrep(dm).fullmatrix(), dm)

    def get_e_exchange(self, dm):
        e = self._memo_energy(dm, 4)
        if e is not None:
            return e
        exc = self.get_exchange(dm)
        ene = SpinParam.apply_fcn(lambda e, d: 0.5 * self._trdot(e.fullmatrix(), d), exc, dm)
        return SpinParam.sum(ene)

    def get_e_xc(self, dm):
        assert self.xc is not None, "Please call .setup_grid with the xc object"
        if isinstance(dm, SpinParam):  # hcgto.py:320-328 with SpinParam densinfo
            densinfo = self._dm2densinfo_pol(dm)
            return self._allsum(torch.sum(self.dvolume * self.xc.get_edensityxc(densinfo), dim=-1))

        e = self._memo_energy(dm, 3)
        if e is not None:
            return e

        def one(d):
            edens = self.xc.get_edensityxc(self._dm2densinfo(d))
            return self._allsum(torch.sum(self.dvolume * edens, dim=-1))

        return self._batched(one, dm)

    # ------------------------------------------------------------------ variational-solver hooks (not in scope)
    def ao_orb_params2dm(self, *a, **k):
        raise NotImplementedError("the variational orbital parametrisation is out of scope (SURVEY.md 2, row 3)")

    def dm2ao_orb_params(self, *a, **k):
        raise NotImplementedError("the variational orbital parametrisation is out of scope (SURVEY.md 2, row 3)")

    # ------------------------------------------------------------------ grid passes
    def _dm2densinfo(self, dm) -> ValGrad:
        if not self.is_ao_set:
            raise RuntimeError("Please call `setup_grid(grid, xc)` first")
        gga = self.xcfamily in (2, 4)
        if gga and not self.is_grad_ao_set:
            raise RuntimeError("Please call `setup_grid(grid, gradlevel>=1)` to calculate the density gradient")
        fac = self._factor_of(dm)
        # meta-GGA: the laplacian of the density is only formed for functionals that may use it (none of the kernel set does:
        # LibXC objects; a user-supplied BaseXC gets it, as in the reference, through the full-matrix path below)
        from .xc import LibXC
        # meta-GGA: LibXC objects of the kernel set never read the laplacian of the density (lapl=None); any other BaseXC gets it,
        # as in the reference, from the factor too (cross term by the polarisation identity, see below)
        use_factor = fac is not None and (self.xcfamily != 4 or self.is_lapl_ao_set)
        self.grid_path_counts["factor" if use_factor else "dense"] += 1
        if use_factor and self.xcfamily == 4 and isinstance(self.xc, LibXC) and all(f[0].shape[1] <= 96 for f in fac):
            # meta-GGA functional of the kernel set (no laplacian): rho, grad rho and tau from ONE pass over the AO components
            rho, grho, kin = lib.grid_density_lr_tau(self._ao, self._nao_ao, fac[0])
            for f in fac[1:]:
                r2, g2, k2 = lib.grid_density_lr_tau(self._ao, self._nao_ao, f)
                rho, grho, kin = rho + r2, grho + g2, kin + k2
            return ValGrad(value=rho, grad=grho, lapl=None, kin=kin)
        if use_factor:
            # D = L L^T known: two chained rank-n_occ GEMMs instead of Phi . D
            rho, grho = lib.grid_density_lr(self._ao, self._nao_ao, fac[0], gga)
            for f in fac[1:]:
                r2, g2 = lib.grid_density_lr(self._ao, self._nao_ao, f, gga)
                rho = rho + r2
                grho = None if grho is None else grho + g2
            if self.xcfamily != 4:
                return ValGrad(value=rho, grad=grho)
            # tau = 1/2 sum_d sum_r (d_d Phi . L)_r^2 (hcgto.py:420-438 grad_grad term): the factor kernel on each gradient
            # component, phase 1 only
            gg = sum(lib.grid_density_lr(self._ao[d], self._nao_ao, f, False)[0] for d in (1, 2, 3) for f in fac)
            if isinstance(self.xc, LibXC):
                return ValGrad(value=rho, grad=grho, lapl=None, kin=gg * 0.5)
            # lapl rho = 2 (sum_ij phi_i D_ij lapl phi_j + gg) (hcgto.py:427-436); with D = L L^T the cross term is
            # sum_r (Phi L)_r (lapl Phi L)_r = 1/4 [ |(Phi + lapl Phi) L|^2 - |(Phi - lapl Phi) L|^2 ]: two more phase-1 passes of the
            # factor kernel on the arrays Phi +- lapl Phi (formed on first use: 2 x ngrid x ld doubles)
            if getattr(self, "_ao_lapl_pm", None) is None:
                self._ao_lapl_pm = (lib.ao_from(self._ao[0] + self._ao[4], self._nao_ao), lib.ao_from(self._ao[0] - self._ao[4], self._nao_ao))
            pl = sum(lib.grid_density_lr(self._ao_lapl_pm[0], self._nao_ao, f, False)[0] for f in fac)
            mi = sum(lib.grid_density_lr(self._ao_lapl_pm[1], self._nao_ao, f, False)[0] for f in fac)
            return ValGrad(value=rho, grad=grho, lapl=2.0 * (0.25 * (pl - mi) + gg), kin=gg * 0.5)
        dmdmt = (dm + dm.transpose(-2, -1)) * 0.5
        dao = lib.pad_matrix(self._unconvert_dm(dmdmt), self._ld)
        rho, grho = lib.grid_density(self._ao, self._nao_ao, dao, gga)
        if self.xcfamily != 4:
            return ValGrad(value=rho, grad=grho)
        # meta-GGA extras (hcgto.py:420-438): grad_grad = sum_d dphi_d D dphi_d, lapl_basis = phi D lapl(phi)
        if not self.is_lapl_ao_set:
            raise RuntimeError("Please call `setup_grid(grid, gradlevel>=2)` to calculate the density gradient")
        gg = sum(lib.grid_density_pair(self._ao[d], self._ao[d], self._nao_ao, dao) for d in (1, 2, 3))
        lb = lib.grid_density_pair(self._ao[0], self._ao[4], self._nao_ao, dao)
        return ValGrad(value=rho, grad=grho, lapl=(lb + gg) * 2, kin=gg * 0.5)

    def _dm2densinfo_pol(self, dm: SpinParam) -> SpinParam:
        """both spin channels' densities (SpinParam.apply_fcn over _dm2densinfo in the reference, hcgto.py:260-269).  GGA with both
        factors known: ONE pass over the AO matrix for the two spins (dqc_grid_density_lr_pol) instead of one per spin"""
        if self.xcfamily == 2 and self.is_ao_set and self.is_grad_ao_set and os.environ.get("DQC_AMD_POL_DENSITY", "fused") != "split":
            fu, fd = self._factor_of(dm.u), self._factor_of(dm.d)
            if fu is not None and fd is not None and len(fu) == 1 and len(fd) == 1:
                out = lib.grid_density_lr_pol(self._ao, self._nao_ao, fu[0], fd[0])
                if out is not None:
                    self.grid_path_counts["factor"] += 2
                    self.grid_path_counts["factor_spin_fused"] = self.grid_path_counts.get("factor_spin_fused", 0) + 1
                    rho, grho = out
                    return SpinParam(u=ValGrad(value=rho[0], grad=grho[0]), d=ValGrad(value=rho[1], grad=grho[1]))
        return SpinParam(u=self._dm2densinfo(dm.u), d=self._dm2densinfo(dm.d))

    def _get_vxc_from_potinfo(self, potinfo: ValGrad):
        vm = self._vxc_ao_from_potinfo(potinfo)
        mat = self._convert2(vm[:self._nao_ao, :self._nao_ao])
        return (mat + mat.transpose(-2, -1)) * 0.5

    def _vxc_ao_from_potinfo(self, potinfo: ValGrad):
        """(ld, ld) AO-basis Vxc matrix (hcgto.py:445-489 before the conversion to the orthogonal basis)"""
        vg = potinfo.grad if self.xcfamily in (2, 4) else None
        vm = lib.grid_vxc(self._ao, self._nao_ao, self.dvolume, potinfo.value.contiguous(),
                          None if vg is None else vg.contiguous())
        if self.xcfamily == 4:  # hcgto.py:473-489
            lapl, kin = potinfo.lapl, potinfo.kin  # lapl None: no laplacian dependence (no device read to find out)
            if lapl is not None:
                vm = vm + lib.grid_vxc_pair(self._ao[0], self._ao[4], self._nao_ao, self.dvolume, (2.0 * lapl).contiguous())
            lk = ((2.0 * lapl if lapl is not None else 0.0) + 0.5 * kin).contiguous()
            # sum_d (d_d Phi)^T diag(w lk) (d_d Phi) = G^T diag(w lk, w lk, w lk) G with G the three gradient arrays stacked along the
            # point axis (they are consecutive in memory): ONE symmetric rank update over 3 ngrid rows instead of three launches
            g3 = self._ao[1:4].reshape(-1, self._ao.shape[-1])
            vm = vm + lib.grid_vxc_pair(g3, g3, self._nao_ao, self.dvolume.repeat(3), lk.repeat(3), what="dqc_grid_vxc_pair[three gradient components]")
        return self._allsum(vm)

    def get_elrep_plus_vxc(self, dm, core=None):
        """J[D] + Vxc[D] of ONE restricted density matrix as a plain tensor in the orthogonalised basis -- the sum
        `_KSEngine.__dm2fock` forms (ks.py:176-187) -- with a single AO -> orthogonal conversion X^T (J_ao + V_ao) X instead
        of one per operator.  Same numbers as get_elrep(dm) + get_vxc(dm) up to round-off."""
        assert self.xc is not None and dm.dim() == 2
        fac = self._factor_of(dm)
        if self._fused_fock_ok(dm) and not (_COULOMB_SIDE and not torch.cuda.is_current_stream_capturing()):
            return self._elrep_plus_vxc_fused(dm, fac, core)
        if fac is not None and len(fac) == 1:  # D_ao = L_ao L_ao^T: one thin GEMM instead of X D X^T
            n = self._nao_ao
            dao = (fac[0][0] @ fac[0][1])[:n, :n].contiguous()
        else:
            dao = self._unconvert_dm(dm)
        if self._tile_slice is not None:  # tile store spread over the ranks: J, Vxc and E_xc parts travel in one all_reduce
            self._deferred = []
        side = main = None
        if _COULOMB_SIDE and dm.is_cuda and self._df is None and not self._direct and self._tile_slice is None \
                and not torch.cuda.is_current_stream_capturing():
            main = torch.cuda.current_stream(dm.device)
            side = _COULOMB_SIDE.get((dm.device.index, main.cuda_stream))
        try:
            if side is not None:  # the tile stream on its own compute units, beside this molecule's grid pass
                side.wait_stream(main)
                dao.record_stream(side)
                with torch.cuda.stream(side):
                    jao, _ = self._jk_ao(dao, False)
            elif self._df is not None:
                jao = self._df.coulomb_ao(dao)
            else:
                jao, _ = self._jk_ao(dao, False)
            densinfo = self._dm2densinfo(dm)
            if hasattr(self.xc, "get_vxc_and_exc"):  # potentials and the E_xc quadrature from one pass over the grid
                potinfo, exc = self.xc.get_vxc_and_exc(densinfo, self.dvolume)
                if exc is not None:
                    exc = exc[:1]  # (the result; the rest of the buffer is the kernel's per-block scratch)
                    self._allsum(exc)  # (sharded: the quadrature of this rank's slab)
            else:
                potinfo, exc = self.xc.get_vxc(densinfo), None
            vm = self._vxc_ao_from_potinfo(potinfo)
            if side is not None:
                main.wait_stream(side)
                jao.record_stream(main)  # (allocated on the side stream, read and later freed on this one)
            if self._deferred is not None:
                self._allsum_flush()
        finally:
            self._deferred = None  # (an exception in between must not leave later _allsum calls queued for ever)
        # the two-electron energies of THIS density fall out of the build (tr D J = tr D_ao J_ao): remembered under the
        # identity + version of `dm`, so that dm2energy(dm) right after dm2scp(dm) streams neither the tiles nor the grid again
        e_j = 0.5 * (dao * jao).sum()
        self._energy_memo = (dm, dm._version, e_j, None if exc is None else exc[0])
        mat = self._convert2(jao + vm[:self._nao_ao, :self._nao_ao])
        mat = (mat + mat.transpose(-2, -1)) * 0.5
        return mat if core is None else core + mat

    def _elrep_plus_vxc_fused(self, dm, fac, core=None):
        """get_elrep_plus_vxc with the small-matrix ends fused (csrc/fock.hip): AO density (from the orbital factor when it is known,
        else X D X^T) + zeroed accumulators in ONE launch, the tile pass, the grid pass, then J's symmetrisation, tr D J / 2,
        X^T (J + V_ao) X and its symmetrisation in ONE launch"""
        n, x, work = self._nao_ao, self._orthozer, self._jkwork
        tiles = self._tiles
        if fac is not None and len(fac) == 1 and fac[0][0].is_contiguous():
            lib.fock_prep(work, x, n, False, orb=fac[0][0])
        else:
            lib.fock_prep(work, x, n, False, dm=dm.contiguous())
        lib.jk_stream_prepared(tiles, n, work, False)
        densinfo = self._dm2densinfo(dm)
        if hasattr(self.xc, "get_vxc_and_exc"):
            potinfo, exc = self.xc.get_vxc_and_exc(densinfo, self.dvolume)
            if exc is not None:
                exc = exc[:1]
        else:
            potinfo, exc = self.xc.get_vxc(densinfo), None
        if self.xcfamily != 4 and self._pworld == 1:
            # LDA / GGA: the Vxc kernel's raw cross-block sums go straight into the finish (their symmetrisation is part of its
            # combine kernel: one launch less)
            vg = potinfo.grad if self.xcfamily == 2 else None
            vraw, vsc = lib.grid_vxc_raw(self._ao, n, self.dvolume, potinfo.value.contiguous(), None if vg is None else vg.contiguous())
            mat, en = lib.fock_finish_vraw(work, x, n, vraw, vsc, core=core)
        else:
            vm = self._vxc_ao_from_potinfo(potinfo)
            mat, en, _ = lib.fock_finish(work, x, n, False, vxc_ao=vm, core=core)  # (core: the one-electron part, added in the same launch)
        self._energy_memo = (dm, dm._version, en[0], None if exc is None else exc[0])
        return mat

    def get_elrep_plus_vxc_pol(self, dm: SpinParam, core=None):
        """J[D_u + D_d] + Vxc_s[D_u, D_d] of an unrestricted pair of density matrices as a stacked (2, nao, nao) tensor in the
        orthogonalised basis -- the sums the polarised `_KSEngine.__dm2fock` forms (ks.py:176-187, hf.py:93-103) -- with ONE batched
        AO -> orthogonal conversion X^T (J_ao + V_s,ao) X of the two sums instead of one per operator (three), the AO-basis total
        density from the two orbital factors when they are known, and the Coulomb stream over the tile store (HBM-bound) enqueued on
        a second stream BESIDE the two-spin grid pass (its Vxc products are bound by the matrix cores; DQC_AMD_J_OVERLAP=0 or a
        graph capture: one stream).  Same numbers as get_elrep(dm.u + dm.d) + get_vxc(dm) up to round-off."""
        assert self.xc is not None and dm.u.dim() == 2 and self._df is None and not self._direct and self._tile_slice is None
        n = self._nao_ao
        fu, fd = self._factor_of(dm.u), self._factor_of(dm.d)
        if (self._fused_fock_ok(dm.u) and fu is not None and fd is not None and len(fu) == 1 and len(fd) == 1
                and not (_COULOMB_SIDE and not torch.cuda.is_current_stream_capturing())):
            return self._elrep_plus_vxc_pol_fused(dm, fu[0], fd[0], core)

        def coulomb():  # (the AO-basis total density and its Coulomb matrix: nothing the grid pass waits for)
            if fu is not None and fd is not None and len(fu) == 1 and len(fd) == 1:
                dao = (fu[0][0] @ fu[0][1] + fd[0][0] @ fd[0][1])[:n, :n].contiguous()
            else:
                dao = self._unconvert_dm(dm.u + dm.d)
            return self._jk_ao(dao, False)[0]

        side = None
        if dm.u.is_cuda and os.environ.get("DQC_AMD_J_OVERLAP", "1") != "0" and not torch.cuda.is_current_stream_capturing():
            side = getattr(self, "_j_stream", None)
            if side is None:
                side = self._j_stream = torch.cuda.Stream(device=dm.u.device)
        if side is not None:
            main = torch.cuda.current_stream(dm.u.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                jao = coulomb()
        else:
            jao = coulomb()
        potinfo = self.xc.get_vxc(self._dm2densinfo_pol(dm))
        vu = self._vxc_ao_from_potinfo(potinfo.u)[:n, :n]
        vd = self._vxc_ao_from_potinfo(potinfo.d)[:n, :n]
        if side is not None:
            main.wait_stream(side)
            jao.record_stream(main)  # (allocated on the side stream, read and later freed on this one)
        x = self._orthozer
        mat = x.transpose(-2, -1) @ torch.stack([jao + vu, jao + vd]) @ x
        mat = (mat + mat.transpose(-2, -1)) * 0.5
        return mat if core is None else core + mat

    def _elrep_plus_vxc_pol_fused(self, dm, fu, fd, core=None):
        """get_elrep_plus_vxc_pol through the fused build ends (csrc/fock.hip): the total AO density L L^T of the stacked factor
        [L_u | L_d] in one launch, the Coulomb stream beside the two-spin grid pass (second stream, as in the torch form), then per
        spin M_s = J + V_s, X^T M_s X, the symmetrisation and the core Hamiltonian in three launches"""
        n, x, work = self._nao_ao, self._orthozer, self._jkwork
        tiles = self._tiles
        lib.fock_prep(work, x, n, False, orb=torch.cat([fu[0], fd[0]], dim=1))
        side = None
        if dm.u.is_cuda and os.environ.get("DQC_AMD_J_OVERLAP", "1") != "0" and not torch.cuda.is_current_stream_capturing():
            side = getattr(self, "_j_stream", None)
            if side is None:
                side = self._j_stream = torch.cuda.Stream(device=dm.u.device)
        if side is not None:
            main = torch.cuda.current_stream(dm.u.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                lib.jk_stream_prepared(tiles, n, work, False)
        else:
            lib.jk_stream_prepared(tiles, n, work, False)
        potinfo = self.xc.get_vxc(self._dm2densinfo_pol(dm))
        vu = self._vxc_ao_from_potinfo(potinfo.u)
        vd = self._vxc_ao_from_potinfo(potinfo.d)
        if side is not None:
            main.wait_stream(side)
        # (the two finishes share the scratch regions of the work buffer: stream order keeps them apart)
        f_u, _, _ = lib.fock_finish(work, x, n, False, vxc_ao=vu, core=core)
        f_d, _, _ = lib.fock_finish(work, x, n, False, vxc_ao=vd, core=core)
        return torch.stack([f_u, f_d])

    def _memo_energy(self, dm, k):
        c = getattr(self, "_energy_memo", None)
        if c is not None and isinstance(dm, torch.Tensor) and c[0] is dm and c[1] == dm._version and k < len(c):
            return c[k]
        return None

    def get_elrep_plus_exchange(self, dm, core=None):
        """J[D] - K[D] / 2 of ONE restricted density matrix as a plain tensor in the orthogonalised basis -- the sum a restricted
        Hartree-Fock Fock build forms from get_elrep(dm) and get_exchange(dm) (hf.py:198-199, hcgto.py:204-241) -- with a single
        AO -> orthogonal conversion X^T (J_ao - K_ao / 2) X instead of one per operator (two rocBLAS GEMMs of ~10 us each at
        nao ~ 100, where the whole tile pass takes 58 us).  Same numbers as the operators' sum up to round-off."""
        if self._df is not None:  # hcgto.py:229-230
            raise RuntimeError("Exact exchange cannot be computed with density fitting")
        assert dm.dim() == 2
        if self._fused_fock_ok(dm):
            # D_ao = X D X^T + zeroed accumulators, the tile pass, then symmetrised J / K, the two traces and X^T (J - K / 2) X: three
            # launches (csrc/fock.hip) instead of sixteen
            n, x, work = self._nao_ao, self._orthozer, self._jkwork
            tiles = self._tiles
            fac = self._factor_of(dm)
            if fac is not None and len(fac) == 1 and fac[0][0].is_contiguous():  # D_ao = L L^T from the orbital factor (dqc_fock_factor)
                lib.fock_prep(work, x, n, True, orb=fac[0][0])
            else:
                lib.fock_prep(work, x, n, True, dm=dm.contiguous())
            lib.jk_stream_prepared(tiles, n, work, True)
            mat, en, _ = lib.fock_finish(work, x, n, True, core=core)
            self._energy_memo = (dm, dm._version, en[0], None, en[1])
            return mat
        dao = self._unconvert_dm(dm)
        J, K = self._jk_ao(dao, True)
        # the two-electron energies of THIS density fall out of the build: remembered like get_elrep_plus_vxc's
        self._energy_memo = (dm, dm._version, 0.5 * (dao * J).sum(), None, -0.25 * (dao * K).sum())
        mat = self._convert2(J - 0.5 * K)
        mat = (mat + mat.transpose(-2, -1)) * 0.5
        return mat if core is None else core + mat

    def timed_fock_kernels(self, dm, core):
        """measurement aid (bench.py): the restricted KS Fock build `core + get_elrep_plus_vxc(dm)` unrolled -- the same
        library calls in the same order -- with a HIP event on the launch stream between its kernels.
        Returns (names, events) with len(events) == len(names) + 1."""
        assert self.xc is not None and dm.dim() == 2 and self.xcfamily == 2
        n = self._nao_ao
        fac = self._factor_of(dm)
        ev = []

        def mark():
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.device))
            ev.append(e)

        if self._fused_fock_ok(dm):  # the calls of _elrep_plus_vxc_fused, in its order
            x, work, tiles = self._orthozer, self._jkwork, self._tiles
            mark()
            if fac is not None and len(fac) == 1 and fac[0][0].is_contiguous():
                lib.fock_prep(work, x, n, False, orb=fac[0][0])
            else:
                lib.fock_prep(work, x, n, False, dm=dm.contiguous())
            mark()
            lib.jk_stream_prepared(tiles, n, work, False)
            mark()
            if fac is not None and len(fac) == 1:
                rho, grho = lib.grid_density_lr(self._ao, n, fac[0], True)
            else:
                dmdmt = (dm + dm.transpose(-2, -1)) * 0.5
                rho, grho = lib.grid_density(self._ao, n, lib.pad_matrix(self._unconvert_dm(dmdmt), self._ld), True)
            mark()
            _, v, vg = lib.xc_eval(self.xc.terms, rho, grho, want_e=False, want_v=True)
            mark()
            vraw, vsc = lib.grid_vxc_raw(self._ao, n, self.dvolume, v, vg)
            mark()
            fock, _ = lib.fock_finish_vraw(work, x, n, vraw, vsc, core=core.contiguous())  # noqa: F841
            mark()
            return ["orth_transforms", "jk_tiles", "grid_density", "xc_eval", "grid_vxc", "fock_assemble"], ev
        dao_n = self._unconvert_dm((dm + dm.transpose(-2, -1)) * 0.5).contiguous()

        mark()
        if self._df is None:
            jao, _ = self._jk_ao(dao_n, False)
        else:
            jao = lib.df_coulomb(self._df.j3c, self._df._inv_j2c, dao_n, self._df._work)
        mark()
        dao = lib.pad_matrix(dao_n, self._ld)
        mark()
        names = ["jk_tiles", "orth_transforms"]
        if fac is not None and len(fac) == 1:
            rho, grho = lib.grid_density_lr(self._ao, n, fac[0], True)
        else:
            rho, grho = lib.grid_density(self._ao, n, dao, True)
        mark()
        _, v, vg = lib.xc_eval(self.xc.terms, rho, grho, want_e=False, want_v=True)
        mark()
        vm = lib.grid_vxc(self._ao, n, self.dvolume, v, vg)
        mark()
        names += ["grid_density", "xc_eval", "grid_vxc"]
        mat = self._convert2(jao + vm[:n, :n])
        fock = core + (mat + mat.transpose(-2, -1)) * 0.5  # noqa: F841
        mark()
        names.append("fock_assemble")
        return names, ev

    def getparamnames(self, methodname: str, prefix: str = "") -> List[str]:
        table = {
            "get_kinnucl": ["kinnucl_mat"], "get_nuclattr": ["nucl_mat"], "get_overlap": ["olp_mat"],
            "get_elrep": ["_tiles"], "get_exchange": ["_tiles"], "ao_orb2dm": [],
            "get_e_hcore": ["kinnucl_mat"], "get_e_elrep": ["_tiles"], "get_e_exchange": ["_tiles"],
            "get_e_xc": ["_ao", "_orthozer", "dvolume"], "get_vxc": ["_ao", "_orthozer", "dvolume"],
            "get_vext": ["_ao", "_orthozer", "dvolume"], "_dm2densinfo": ["_ao", "_orthozer"],
            "_get_vxc_from_potinfo": ["_ao", "_orthozer", "dvolume"],
        }
        if methodname not in table:
            raise KeyError("getparamnames has no %s method" % methodname)
        return [prefix + n for n in table[methodname]]
