"""Molecule batches across the GPUs of one node.

The reference has no batching (one Mol -> one Hamiltonian, dqc/system/mol.py:77-121); a batch is N
independent SCF problems, so the path shards with NO data-path collective: every rank
(one process per GPU) takes a cost-balanced share of the molecules, and one all_gather of
(energy, iterations, fock seconds) per molecule closes the run.  Backend "nccl" is RCCL on ROCm;
the CPU tests drive the same code with "gloo"."""
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def molecule_cost(nao: int, ngrid: int) -> float:
    """relative per-iteration cost: ERI-tile stream ~ nao^4 bytes, grid passes ~ ngrid*nao^2 flops"""
    return float(nao) ** 4 + 12.0 * float(ngrid) * float(nao) ** 2


def molecule_bytes(nao: int, ngrid: int, ncomp: int = 4) -> int:
    """device memory one resident molecule needs: 8-fold-unique ERI tiles (~nao^4 bytes), the AO matrix on the grid
    (ncomp = 1 LDA, 4 GGA, 5 meta-GGA components) and the per-point work arrays of a Fock build"""
    nb = (nao + 7) // 8
    npair = nb * (nb + 1) // 2
    ld = (nao + 15) // 16 * 16 + 16
    # (an upper bound of the packed store: full 8^4 tiles)
    return npair * (npair + 1) // 2 * 4096 * 8 + 8 * ncomp * ngrid * ld + 8 * 16 * ngrid + 64 * ld * ld


def reserve_device_memory(nbytes: int, device) -> float:
    """Take `nbytes` of device memory from the driver in ONE request and hand it to PyTorch's caching allocator (the block
    is freed into the cache, later tensors are carved out of it).  Fresh VRAM is not free on this platform: the kernel
    driver clears it at ~35 GB/s (measured, tools/gpu_alloc_cost*.py: the first ~84 GiB of a freshly booted MI355X are
    instant, beyond that -- or while memory released by a previous process is still being wiped -- a 2 GB hipMalloc blocks
    for 60 ms), which used to be spread over the ERI-tile and AO allocations of every molecule of a batch.  Returns the
    seconds the request took.  A long-lived process pays this once, not once per batch."""
    import time
    dev = torch.device(device)
    free, _ = torch.cuda.mem_get_info(dev)
    cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    nbytes = int(min(nbytes, free - (2 << 30)))
    if nbytes <= cached:
        return 0.0
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    block = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    del block
    return time.perf_counter() - t0


class CuPartition:
    """The chip split in two for a batch of restricted Kohn-Sham builds (round 6): `grid_streams` -- HIP streams confined to the
    first 32 - k compute units of every XCD, on which the caller runs its Fock builds (density, functional and Vxc kernels: bound
    by the matrix cores, most registers and LDS of their CUs taken) -- and ONE Coulomb stream on the other k CUs per XCD, to which
    `HamiltonMI355.get_elrep_plus_vxc` sends the pass over the ERI tiles (HBM-bound, no matrix-core work) of every build that runs
    on one of the grid streams.  The tile stream of molecule m then rides in the HBM bandwidth the grid pass of molecule m leaves;
    with two or more grid streams the next molecule's Coulomb pass starts while this one's build still waits for its own.
    k = 0 or DQC_AMD_J_CUS=0: no partition (`grid_streams` are ordinary streams, Coulomb pass in line).  k must be a multiple of 4:
    a mask bit is (CU slot, XCD) with the slots dealt to the four shader engines of an XCD in turn, and the workgroup dispatcher
    feeds the engines round-robin -- a partition with unequal engines runs at the pace of its smallest (30 CUs per XCD: Vxc 2 x
    slower than on 28).  MEASURED (profiles/r06a_cu_partition_curves.txt, r06b_*): the Coulomb stream is bound per CU (36 GB/s each:
    64 CUs for 2.2 TB/s), the Vxc kernel by the chip's power (192 CUs: 0.64 ms against 0.56 on 256), and the batch's throughput is
    the same 715-739 it/s for every split, stream count and Vxc CU cap -- opt-in, not what bench.py runs.
        part = CuPartition(dev); ...; with torch.cuda.stream(part.grid_streams[i % len(part.grid_streams)]): eng.dm2scp(dm)"""

    def __init__(self, device, j_cus_per_xcd=None, n_grid_streams=2):
        import os
        from . import lib
        from .hamilton import register_coulomb_side_stream
        dev = torch.device(device)
        self.device = dev
        if j_cus_per_xcd is None:
            j_cus_per_xcd = int(os.environ.get("DQC_AMD_J_CUS", "8"))
        ncu = lib.device_cu_count(dev)
        per = ncu // 8 if (ncu >= 64 and ncu % 8 == 0) else ncu
        k = max(0, min(int(j_cus_per_xcd), per - 1))
        self.j_cus_per_xcd, self.cus_per_xcd = k, per
        self._parts = []
        if k == 0:
            self.coulomb = None
            self.grid_streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, n_grid_streams))]
            return
        pj = lib.partition_stream(dev, per - k, per)
        self._parts.append(pj)
        self.coulomb = pj.stream
        self.grid_streams = []
        for _ in range(max(1, n_grid_streams)):
            pg = lib.partition_stream(dev, 0, per - k)
            self._parts.append(pg)
            self.grid_streams.append(pg.stream)
            register_coulomb_side_stream(pg.stream, pj.stream)

    def close(self):
        from .hamilton import register_coulomb_side_stream
        torch.cuda.synchronize(self.device)
        if self.coulomb is not None:
            for g in self.grid_streams:
                register_coulomb_side_stream(g, None)
        for p in self._parts:
            p.close()
        self._parts = []


def prepare_orthogonalisers(hams) -> int:
    """The orthogonalisers X (X^T S X = 1: eigh(S), eigenvalues below 1e-6 dropped, dqc/hamilton/orbconverter.py:67-116) of a
    batch of Hamiltonians from ONE batched eigh per matrix size instead of one rocSOLVER call per molecule: 6.5 ms for 32
    matrices of 208 x 208 against 4.7 ms each in a loop (tools/ubench/eigh_batch.py) -- and in a molecule-by-molecule setup the
    host WAITS for each of them (11 ms per molecule with the ERI fill queued beside it).  Call it on freshly constructed
    Hamiltonians (`mol.get_hamiltonian()`) before their engines are built; returns the number of matrices it handled."""
    groups = {}
    for h in hams:
        if getattr(h, "_X", "no") is None and getattr(h, "orthogonalized", False):
            groups.setdefault((int(h._nao_ao), str(h.device)), []).append(h)
    done = 0
    for hs in groups.values():
        if len(hs) < 2:
            continue
        ev, evec = torch.linalg.eigh(torch.stack([h._ovlp_ao for h in hs]))
        accs = ev > 1e-6
        full = bool(accs.all())  # one host read for the group: no function is dropped anywhere (the usual case)
        for k, h in enumerate(hs):
            if full:
                h._X = (evec[k] * ev[k] ** (-0.5)).contiguous()
            else:
                acc = accs[k]
                h._X = (evec[k][:, acc] * ev[k][acc] ** (-0.5)).contiguous()
            done += 1
    return done


def shard_lpt(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """longest-processing-time-first assignment of molecule indices to ranks (deterministic)"""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    for r in range(world_size):
        out[r].sort()
    return out


def run_sharded(nmol: int, costs: Sequence[float], runner: Callable[[int], Sequence[float]],
                nvals: int = 3, device=None) -> torch.Tensor:
    """Run `runner(i) -> nvals floats` for this rank's molecules and return the (nmol, nvals) table on every
    rank.  Works without an initialised process group (single process)."""
    ws = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if ws > 1 else 0
    mine = shard_lpt(costs, ws)[rank]
    dev = torch.device("cpu") if device is None else torch.device(device)
    table = torch.zeros((nmol, nvals), dtype=torch.float64, device=dev)
    for i in mine:
        table[i] = torch.as_tensor(list(runner(i)), dtype=torch.float64, device=dev)
    if ws > 1:
        # every row is owned by exactly one rank and zero elsewhere: a sum is a gather
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
    return table


def run_concurrent(qcs, max_inflight: int = 16, **run_kwargs):
    """Run the SCF loops of many (small) molecules AT ONCE on one GPU: one HIP stream per molecule in flight, the
    per-iteration host read of each loop (max|[F,D]| + the DIIS Gram row, a few hundred bytes) as an asynchronous copy
    into pinned memory behind an event, and a round-robin over the generators (`SCF_QCCalc._run_gen`).  While the host
    does the DIIS algebra of one molecule the hipGraph steps of the others are running, so molecules whose kernels fill
    only a fraction of the 256 CUs (H2O ... benzene: 0.1-0.4 ms of launch-bound work per iteration) share the chip.
    SURVEY.md 7 step 6 / 8e ("each rank: own HIP streams").  `qcs`: built HF / KS objects on one device; returns them."""
    if not qcs:
        return qcs
    for q in qcs:
        # every rank of a sharded Hamiltonian must take the same branches (its Fock builds hold collectives): this driver reads each
        # generator's scalars on the local rank only -- use qc.run(), which broadcasts rank 0's
        for e in getattr(q, "engines", None) or [q._engine]:
            if getattr(getattr(e, "hamilton", None), "sharded", False):
                raise RuntimeError("run_concurrent / run_lockstep cannot drive a Hamiltonian sharded over several GPUs (shard_over): "
                                   "call qc.run() on every rank instead")
    dev = qcs[0]._engine.device
    n_in = max(1, min(max_inflight, len(qcs)))
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_in)]
    main = torch.cuda.current_stream(dev)
    for s in streams:
        s.wait_stream(main)

    class _Slot:
        pass

    def post(slot, req):
        # the request tensor -> pinned host memory, asynchronously on the molecule's stream
        if slot.pinned is None or slot.pinned.numel() < req.numel():
            slot.pinned = torch.empty(max(64, req.numel()), dtype=req.dtype, pin_memory=True)
        slot.n = req.numel()
        slot.pinned[:slot.n].copy_(req.reshape(-1), non_blocking=True)
        slot.ev.record(slot.stream)

    todo = list(qcs)
    active = []
    free = list(streams)

    def start():
        while todo and free:
            sl = _Slot()
            sl.qc, sl.stream, sl.pinned, sl.ev = todo.pop(0), free.pop(0), None, torch.cuda.Event()
            with torch.cuda.stream(sl.stream):
                sl.gen = sl.qc._run_gen(**run_kwargs)
                post(sl, next(sl.gen))
            active.append(sl)

    start()
    while active:
        for sl in list(active):
            sl.ev.synchronize()
            host = sl.pinned[:sl.n].numpy().copy()
            with torch.cuda.stream(sl.stream):
                try:
                    post(sl, sl.gen.send(host))
                except StopIteration:
                    active.remove(sl)
                    free.append(sl.stream)
        start()
    for s in streams:
        main.wait_stream(s)
    return qcs


def run_lockstep(qcs, group_size: int = None, inflight: int = 2, nstreams: int = None, **run_kwargs):
    """Run the SCF of a batch on one GPU with everything but the Fock builds batched across molecules
    (dqc_amd/lockstep.py): calculations are bucketed by (nao, n_occ), every bucket is cut into groups of at most
    `group_size` that advance in lockstep, and `inflight` groups are driven at once on their own streams so that one
    group's latency-bound phase (DIIS, purification: ~1 ms of small launches) overlaps another's Fock builds.
    Calculations that do not qualify (unrestricted, non-uniform occupations, raw AO basis) run through
    `run_concurrent`.  Returns `qcs`; every element is in the state `qc.run()` leaves it in.
    Defaults (measured on MI355X, tools/gpu_lockstep_check.py): groups of 128 / 32 / 16 molecules for nao <= 48 / <= 160 /
    larger, 4 streams per group where the Fock builds replay as hipGraphs (nao <= 160), 3 where they are issued eagerly."""
    from .lockstep import LockstepSCF, signature
    buckets, rest = {}, []
    for q in qcs:
        s = signature(q)
        (buckets.setdefault(s, []) if s is not None else rest).append(q)
    groups = []
    for s, members in buckets.items():
        if len(members) == 1:
            rest.extend(members)
            continue
        nbas = s[1]
        gsz = group_size if group_size is not None else (128 if nbas <= 48 else (32 if nbas <= 160 else 16))
        nst = nstreams if nstreams is not None else (4 if nbas <= 160 else 3)
        # equal-sized groups, at least `inflight` of them when there are enough molecules
        ng = max((len(members) + gsz - 1) // gsz, min(inflight, len(members) // 2))
        size = (len(members) + ng - 1) // ng
        for i in range(0, len(members), size):
            groups.append(LockstepSCF(members[i:i + size], nstreams=nst))
    if groups:
        run_concurrent(groups, max_inflight=inflight, **run_kwargs)
    if rest:
        run_concurrent(rest, **run_kwargs)
    return qcs
