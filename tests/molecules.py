"""Synthetic inputs shared by tests and bench.py: geometries in Bohr."""
import numpy as np

ANG = 1.0 / 0.52917721092

H2O = ([8, 1, 1], [[0, 0, 0.2156], [0, 1.4749, -0.8625], [0, -1.4749, -0.8625]])  # reference test_properties.py:21-27
CH4 = ([6, 1, 1, 1, 1], [[0, 0, 0], [1.186, 1.186, 1.186], [-1.186, -1.186, 1.186], [-1.186, 1.186, -1.186],
                         [1.186, -1.186, -1.186]])

# vitamin C, 20 atoms, exactly the reference's 20-atom cc-pVDZ benchmark input (dqc/test/benchmark.py:7-26)
# nitroxyl HNO (N-O 1.212 A, N-H 1.063 A, 108.6 deg; Bohr): the smallest molecule with N, O and their cc-pVTZ f shells
HNO = ([7, 8, 1], [[0, 0, 0], [2.2903, 0, 0], [-0.6408, 1.9040, 0]])
FORMAMIDE = ([6, 8, 7, 1, 1, 1], [[0, 0, 0], [2.2960, 0, 0], [-1.2420, 2.2490, 0], [-0.9950, -1.8230, 0], [-3.1150, 2.0880, 0], [-0.3560, 3.9420, 0]])
VITC = ([8] * 6 + [6] * 6 + [1] * 8,
        [[0.1761, -2.0912, 1.2179], [-2.5390, 0.1686, -2.6197], [1.5859, 4.5166, 0.5374], [-7.3565, -0.3855, -0.6285],
         [5.7108, 1.0762, -1.1444], [3.7634, -4.0742, -0.2600], [-0.6419, 0.4947, 1.4840], [-3.0347, 0.8664, -0.0624],
         [1.5518, 1.9398, 0.5456], [-5.1899, -0.7714, 0.9190], [3.4161, 0.4295, -0.1982], [2.5579, -2.1696, 0.2094],
         [-0.9740, 0.8710, 3.4930], [-3.6211, 2.8520, -0.0816], [-4.7222, -2.7845, 0.8222], [-5.6716, -0.2763, 2.8684],
         [-2.0785, -1.6104, -2.6430], [0.0074, 5.0941, 1.2812], [-8.6971, -1.4445, 0.0486], [6.6186, -0.4860, -1.4846]])


def benzene():
    rcc, rch = 1.397 * ANG, 1.084 * ANG
    zs, pos = [], []
    for k in range(6):
        a = np.pi / 3 * k
        zs.append(6)
        pos.append([rcc * np.cos(a), rcc * np.sin(a), 0.0])
    for k in range(6):
        a = np.pi / 3 * k
        zs.append(1)
        pos.append([(rcc + rch) * np.cos(a), (rcc + rch) * np.sin(a), 0.0])
    return zs, pos


def naphthalene():
    """idealised D2h naphthalene: all r_CC = 1.40 A, r_CH = 1.09 A, 120 degree angles"""
    a, h = 1.40 * ANG, 1.09 * ANG
    s3 = np.sqrt(3.0) / 2
    c = [(0.0, 0.5 * a), (0.0, -0.5 * a)]
    for sx in (1, -1):
        c += [(sx * s3 * a, a), (sx * s3 * a, -a), (sx * 2 * s3 * a, 0.5 * a), (sx * 2 * s3 * a, -0.5 * a)]
    hh = []
    for sx in (1, -1):
        hh += [(sx * s3 * a, a + h), (sx * s3 * a, -a - h),
               (sx * (2 * s3 * a + s3 * h), 0.5 * a + 0.5 * h), (sx * (2 * s3 * a + s3 * h), -0.5 * a - 0.5 * h)]
    zs = [6] * 10 + [1] * 8
    pos = [[x, y, 0.0] for x, y in c + hh]
    return zs, pos


def c5_molecule(i):
    """molecule i of the 32-molecule C5 set: vitamin C + seeded sigma = 0.05 Bohr jitter (i = 0: unperturbed)"""
    zs, pos = VITC
    pos = np.array(pos, dtype=np.float64)
    if i > 0:
        pos = pos + np.random.default_rng(20260928 + i).normal(0.0, 0.05, pos.shape)
    return zs, pos.tolist()


def seeded_dm_ao(nao, nel, S, seed):
    """symmetric PSD pseudo-density with tr(D S) = nel (the probe densities of tools/make_golden.py)"""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((nao, max(nel // 2, 1)))
    D = A @ A.T
    return D * (nel / np.trace(D @ S))
