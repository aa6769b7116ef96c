"""bench.py -- SCF iterations/sec (Fock build + XC grid) per GPU, cc-pVDZ 20-atom  (BASELINE.json metric).

A "step" is one pass of the hot path over this rank's batch: for every local molecule one evaluation of
engine.dm2scp(D) = J (ERI tiles -> Coulomb matrix) + density on the grid + XC functional + Vxc matrix, i.e.
exactly the per-SCF-iteration Fock build of the reference's _KSEngine (dqc/qccalc/ks.py:176-187), float64,
with the one-off setup (ERI fill, AO-on-grid, Becke grid) excluded and reported separately.  Inputs are the
C5 set of SURVEY.md 8(d): vitamin C (the reference's own 20-atom cc-pVDZ benchmark molecule,
dqc/test/benchmark.py:7-26) plus seeded 0.05-Bohr jitters, RKS PBE, grid sg3; each GPU holds
--molecules-per-gpu of them (4 x 8 GPUs = the 32-molecule batch), so per-GPU work is fixed as N grows
(weak scaling) and there is no data-path collective -- only the closing barrier / max-reduce of the timing.

Usage: python bench.py [--gpus N --steps K --warmup W --molecules-per-gpu M --no-cpu-baseline]
For N>1 launch with torch.distributed.run (one rank per GPU, RCCL); prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
F64_MFMA_PEAK_TF = 78.6  # MI355X dense fp64 matrix peak (vendor figure; the guide lists no fp64 row)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--molecules-per-gpu", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--df", default=None, metavar="AUXBASIS",
                    help="density-fitted Coulomb operator (Mol.densityfit, e.g. --df etb) instead of the exact-J tile stream; "
                         "a different formulation, labelled as such in config")
    ap.add_argument("--dense-dm", action="store_true",
                    help="feed dm2scp an anonymous full density matrix (no ao_orb2dm factor): full-matrix density kernel")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo for self-tests")
    ap.add_argument("--all-ranks-on-gpu0", action="store_true",
                    help="self-test only: map every rank to cuda:0 (exercise the N>1 code path on a 1-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
    import torch.distributed as dist
    if args.all_ranks_on_gpu0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import dqc_amd
    from dqc_amd import lib
    from dqc_amd.batch import shard_lpt, molecule_cost
    from tests import molecules as M

    M_per = args.molecules_per_gpu
    nmol = M_per * world
    costs = [molecule_cost(208, 353400)] * nmol
    mine = shard_lpt(costs, world)[rank]

    # ---------------- one-off setup (not timed as part of the metric) ----------------
    t0 = time.perf_counter()
    engines, dms, orbs = [], [], []
    for i in mine:
        zs, pos = M.c5_molecule(i)
        mol = dqc_amd.Mol((zs, pos), basis="cc-pvdz", grid="sg3", device=dev)
        if args.df:
            mol.densityfit(method="coulomb", auxbasis=args.df)
        eng = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")._engine
        n = eng.shape[-1]
        # density of the core-Hamiltonian guess ("1e", reference scf_qccalc.py:88-91) after one SCF update
        dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device=dev)))
        orb = eng.scp2orb(eng.dm2scp(dm)).contiguous()  # occupied orbitals of the second SCF iterate
        engines.append(eng)
        orbs.append(orb)
        dms.append(eng.hamilton.ao_orb2dm(orb, eng.orb_weight))
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    h0 = engines[0].hamilton
    nao, ngrid, ld = h0._nao_ao, h0.rgrid.shape[0], h0._ld

    def step(record=None, dense_dm=False):
        for eng, dm, orb in zip(engines, dms, orbs):
            # a fresh density-matrix tensor every step (defeats the J/K memoisation: everything is recomputed).
            # Default: D = ao_orb2dm(C_occ, n) exactly as scp2dm produces it in every SCF iteration (hf.py:105-113), so
            # the Hamiltonian knows its rank-n_occ factor; --dense-dm hands over an anonymous full matrix instead.
            d = dm.clone() if dense_dm else eng.hamilton.ao_orb2dm(orb, eng.orb_weight)
            if record is None:
                eng.dm2scp(d)
            else:
                h = eng.hamilton
                fac = h._factor_of(d)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
                dmdmt = (d + d.transpose(-2, -1)) * 0.5
                dao_n = h._unconvert_dm(dmdmt).contiguous()
                ev[0].record()
                if h.df is None:
                    Jao, _ = lib.jk(h._tiles, dao_n, h._jkwork, False)   # the same calls get_elrep / get_vxc make,
                else:                                                    # unrolled so each kernel gets its own events
                    Jao = lib.df_coulomb(h.df.j3c, h.df._inv_j2c, dao_n, h.df._work)
                ev[1].record()
                J = h._convert2(Jao)
                J = (J + J.transpose(-2, -1)) * 0.5
                dao = lib.pad_matrix(dao_n, h._ld)
                ev[2].record()
                if fac is not None:
                    rho, grho = lib.grid_density_lr(h._ao, h._nao_ao, fac[0], True)  # C5: one panel (r = 46 -> 48)
                else:
                    rho, grho = lib.grid_density(h._ao, h._nao_ao, dao, True)
                ev[3].record()
                _, v, vg = lib.xc_eval(h.xc.terms, rho, grho, want_e=False, want_v=True)
                ev[4].record()
                vm = lib.grid_vxc(h._ao, h._nao_ao, h.dvolume, v, vg)
                ev[5].record()
                mat = h._convert2(vm[:h._nao_ao, :h._nao_ao])
                fock = eng.knvext.fullmatrix() + J + (mat + mat.transpose(-2, -1)) * 0.5
                ev[6].record()
                record.append(ev)

    dense = args.dense_dm
    for _ in range(args.warmup):
        step(dense_dm=dense)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- timed region: exactly K steps ----------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(dense_dm=dense)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)

    # ---------------- per-kernel HIP-event timing over K more steps (same stream as the launches) ----------------
    rec = []
    for _ in range(args.steps):
        step(rec, dense_dm=dense)
    torch.cuda.synchronize()
    # the same K steps with an anonymous full density matrix (no factor): the rate a caller that bypasses ao_orb2dm gets
    elapsed_other = None
    if not dense:
        step(dense_dm=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(dense_dm=True)
        barrier()
        elapsed_other = time.perf_counter() - t0
    norb_pad = 0 if dense else lib.padded_norb(orbs[0].shape[1])
    # SURVEY.md 8(d) metric (ii): the full SCF iteration F -> eigh -> ao_orb2dm -> dm2scp (scp2scp), same K steps
    focks = [eng.dm2scp(eng.hamilton.ao_orb2dm(orb, eng.orb_weight)) for eng, orb in zip(engines, orbs)]
    for eng, f in zip(engines, focks):
        eng.scp2scp(f)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for eng, f in zip(engines, focks):
            eng.scp2scp(f)
    barrier()
    elapsed_full_eigh = time.perf_counter() - t0
    # the same with the eigensolver-free step: purification + Fock build replayed as one hipGraph (dqc_amd/graph.py)
    from dqc_amd.graph import GraphedSCFStep
    steps_g = [GraphedSCFStep(eng) for eng in engines]
    for st, f in zip(steps_g, focks):
        st(f)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for st, f in zip(steps_g, focks):
            st(f)
    barrier()
    elapsed_full = time.perf_counter() - t0
    names = ["jk_tiles", "orth_transforms", "grid_density", "xc_eval", "grid_vxc", "fock_assemble"]
    ktime = {nm: sum(e[i].elapsed_time(e[i + 1]) for e in rec) / len(rec) for i, nm in enumerate(names)}  # ms / launch

    # the Hartree-Fock flavour of the J/K pass (get_elrep + get_exchange, hf.py:198-199): one fused J + K launch over the same
    # tiles, timed on molecule 0 (reported next to the kernel times; not part of `value`)
    jk_hf_ms = None
    if h0.df is None:
        dao0 = h0._unconvert_dm(dms[0]).contiguous()
        lib.jk(h0._tiles, dao0, h0._jkwork, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            lib.jk(h0._tiles, dao0, h0._jkwork, True)
        e1.record()
        torch.cuda.synchronize()
        jk_hf_ms = e0.elapsed_time(e1) / args.steps

    if rank == 0:
        c = 4  # GGA: phi + 3 gradient components
        alg_bytes = {
            # AO read once per pass + per-point in/outs + the (n,n) matrix; see DESIGN.md
            "grid_density": 8.0 * c * ngrid * nao + 8.0 * ngrid * 4 + 8.0 * nao * nao,
            "grid_vxc": 8.0 * c * ngrid * nao + 8.0 * ngrid * 5 + 8.0 * nao * nao,
            "jk_tiles": float(nao) ** 4 + 3 * 8.0 * nao * nao,
        }
        if args.df:  # two passes over the i >= j rows of j3c (dqc_df_coulomb) + inv_j2c
            naux = int(h0.df.j2c.shape[0])
            alg_bytes["jk_tiles"] = 2 * 4.0 * nao * (nao + 1) * naux + 8.0 * naux * naux + 2 * 8.0 * nao * nao
        alg_flops = {
            # 2 G n^2 per GEMM pass (+ the row dots / Psi combination); J: 2 n^4 dense-equivalent
            # density: Phi . D (full matrix) or the two chained rank-n_occ GEMMs Phi . L, (Phi L) . L^T (factor form)
            "grid_density": (2.0 * ngrid * ld * ld if dense else 4.0 * ngrid * ld * norb_pad) + 2.0 * c * ngrid * nao,
            "grid_vxc": 2.0 * ngrid * ld * ld + 2.0 * c * ngrid * nao,
            "jk_tiles": 2.0 * float(nao) ** 4,
        }
        # HBM bytes per launch from the committed rocprofv3 PMC pass of this same command (bench.py cannot sample
        # counters itself); only used when the workload matches
        traffic = {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if tj["workload"] == {"nao": nao, "ngrid": ngrid, "xc": "gga"}:
                traffic = dict(tj["hbm_read_bytes_per_launch"])
                if not dense and "grid_density_lr" in traffic:  # the factor-form density kernel was profiled separately
                    traffic["grid_density"] = traffic["grid_density_lr"]
        except Exception:
            pass
        mfma_ceiling = lib.probe_mfma_f64_tflops(dev)
        hbm_ceiling = lib.probe_hbm_read_gbs(dev)

        def roof(k):
            t = ktime[k] * 1e-3
            gbs, tfs = alg_bytes[k] / t / 1e9, alg_flops[k] / t / 1e12
            # the binding roof is the one the kernel sits closer to
            if k != "jk_tiles" and tfs / F64_MFMA_PEAK_TF > gbs / HBM_PEAK_GBS:
                return {"bound": "mfma", "kernel": k, "achieved": tfs, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "frac": tfs / F64_MFMA_PEAK_TF, "traffic": traffic.get(k), "algorithmic_flops_per_launch": alg_flops[k],
                        "algorithmic_bytes_per_launch": alg_bytes[k], "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                        "measured_mfma_f64_ceiling_tflops": mfma_ceiling, "frac_of_measured_ceiling": tfs / mfma_ceiling}
            return {"bound": "hbm", "kernel": k, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": traffic.get(k), "algorithmic_bytes_per_launch": alg_bytes[k],
                    "measured_hbm_read_ceiling_gbs": hbm_ceiling, "frac_of_measured_ceiling": gbs / hbm_ceiling}

        dom = max(alg_bytes, key=lambda k: ktime[k])
        out = {
            "metric": "SCF iterations/sec (Fock build + XC grid) per GPU, cc-pVDZ 20-atom",
            "value": nmol * args.steps / elapsed,
            "unit": "SCF Fock-build iterations/s (whole job)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C5: %d x 20-atom vitamin-C-like organics (nao 208, 353400 grid pts) RKS PBE/cc-pVDZ sg3, "
                                   "%d per GPU" % (nmol, M_per),
                       "coulomb": ("density-fitted J, auxbasis %s (naux %d)" % (args.df, int(h0.df.j2c.shape[0]))) if args.df
                                  else "exact J from stored ERI tiles",
                       "molecules_per_gpu": M_per, "global_batch": nmol, "nao": nao, "ngrid": ngrid,
                       "parallelism": "molecule-sharded x%d, no data-path collective" % world},
            "per_gpu_value": M_per * args.steps / elapsed,
            "density_matrix_input": "full matrix (no factor)" if dense else
                                    "ao_orb2dm(C_occ, n): rank-%d factor known to the Hamiltonian" % norb_pad,
            "value_full_matrix_dm": None if elapsed_other is None else nmol * args.steps / elapsed_other,
            # SURVEY 8(d) metric (ii), rank 0's clock: F -> D -> F'.  "purify": GEMM-only projector + Fock build in one
            # hipGraph; "eigh": torch.linalg.eigh (rocSOLVER) + ao_orb2dm + eager Fock build
            "full_scf_iterations_per_s": nmol * args.steps / elapsed_full,
            "full_scf_iterations_per_s_eigh": nmol * args.steps / elapsed_full_eigh,
            "setup_s_per_rank": setup_s,
            "kernel_ms_per_molecule": ktime,
            "jk_with_exchange_ms": jk_hf_ms,
            "roofline": roof(dom),
            "roofline_other_kernels": [roof(k) for k in alg_bytes if k != dom],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(nsteps):
    """the oracle (CPU restatement of DQC's algorithm, kind = "port") timed on this box's host cores on a bounded
    sample of the same workload: molecule 0 of the C5 set, `nsteps` dm2scp evaluations after one warm-up"""
    from oracle import basis as ob, hamilton as oh, natives as nat
    from tests import molecules as M
    t = ob.make_tables(M.c5_molecule(0), "cc-pvdz")
    t0 = time.perf_counter()
    eng = oh.Engine(t, xc="gga_x_pbe+gga_c_pbe", grid="sg3", eri_mode="s4")
    setup = time.perf_counter() - t0
    n = eng.h.nao
    dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64)))
    # torch's intra-op pool collapses when oversubscribed (256 threads: 70 s per call on the 64-core EPYC box),
    # so the thread count is calibrated: one call each at 16/32/64 threads (capped by the core count), best kept
    best = None
    for nt in sorted({min(c, os.cpu_count() or 1) for c in (16, 32, 64)}):
        torch.set_num_threads(nt)
        eng.dm2scp(dm)
        t0 = time.perf_counter()
        eng.dm2scp(dm)
        dt1 = time.perf_counter() - t0
        if best is None or dt1 < best[0]:
            best = (dt1, nt)
    torch.set_num_threads(best[1])
    t0 = time.perf_counter()
    for _ in range(nsteps):
        eng.dm2scp(dm)
    dt = time.perf_counter() - t0
    return {"value": nsteps / dt, "unit": "SCF Fock-build iterations/s", "cores": torch.get_num_threads(),
            "kind": "port", "setup_s": setup,
            "sample": "molecule 0 of the C5 set, %d dm2scp calls after 1 warm-up; J from the packed-s4 ERI matrix "
                      "(3.8 GB) instead of the reference's dense 15 GB einsum (faster than the reference shape), "
                      "density/Vxc passes chunked at 16 MiB like the reference" % nsteps}


if __name__ == "__main__":
    main()
