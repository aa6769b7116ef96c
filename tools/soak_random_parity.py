"""extended run of tests/test_gpu_parity.py::test_random_molecules_fock_and_energy_vs_oracle over more seeds (not part of the suite)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest, torch
import tests.test_gpu_parity as T
bad = 0
for seed in range(8, 72):
    try:
        T.test_random_molecules_fock_and_energy_vs_oracle(torch.device("cuda:0"), seed)
        print("seed", seed, "ok", flush=True)
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED", repr(e)[:300], flush=True)
print("failures:", bad)
