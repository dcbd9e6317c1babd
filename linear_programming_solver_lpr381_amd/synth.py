"""Synthetic workloads of BASELINE.json / SURVEY.md section 8(d), seed 20251003, NumPy PCG64.

Everything here is host-side input generation; nothing is solved here.
"""
from __future__ import annotations

import numpy as np

SEED = 20251003
MAX, MIN = 0, 1
LE, GE, EQ = 0, 1, 2


def _rng(seed: int, stream: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, stream]))


def dense_lp(m: int, n: int, seed: int = SEED):
    """Config 2/3: A ~ U(0,1) dense, b_i = 0.5*n*U(0.9,1.1) > 0, c ~ U(0.5,1.5), Max, all <=.
    Returns (c[n], A[m,n], b[m])."""
    g = _rng(seed, 1)
    A = g.random((m, n))
    b = 0.5 * n * g.uniform(0.9, 1.1, size=m)
    c = g.uniform(0.5, 1.5, size=n)
    return c, A, b


def primal_tableau_from(c, A, b):
    """BuildTableau (Models/PrimalSimplex.cs:179-203) for an all-<= Max model:
    T[i,0:n]=A_i, T[i,n+i]=1, T[i,n+m]=b_i, T[m,0:n]=-c, basis=n..n+m-1."""
    m, n = A.shape
    T = np.zeros((m + 1, n + m + 1), dtype=np.float64)
    T[:m, :n] = A
    T[np.arange(m), n + np.arange(m)] = 1.0
    T[:m, n + m] = b
    T[m, :n] = -np.asarray(c, dtype=np.float64)
    basis = np.arange(n, n + m, dtype=np.int32)
    return T, basis


def raw_tableau(R: int, C: int, seed: int = SEED):
    """Headline rank-1-update shape: values U(-1,1)."""
    g = _rng(seed, 2)
    return g.uniform(-1.0, 1.0, size=(R, C))


def forced_pivot_list(R: int, C: int, count: int, seed: int = SEED):
    g = _rng(seed, 3)
    rows = g.integers(0, R, size=count, dtype=np.int32)
    cols = g.integers(0, C, size=count, dtype=np.int32)
    return rows, cols


def binary_ip(n: int = 512, m: int = 256, seed: int = SEED):
    """Config 4: A_ij in {0..9}, b_i = floor(0.5*sum_j A_ij), c_j in {1..20}, Max, plus n rows
    x_j <= 1 (the reference has no bounds syntax). Returns (c[n], A[m+n,n], rel[m+n], b[m+n])."""
    g = _rng(seed, 4)
    A0 = g.integers(0, 10, size=(m, n)).astype(np.float64)
    b0 = np.floor(0.5 * A0.sum(axis=1))
    c = g.integers(1, 21, size=n).astype(np.float64)
    A = np.vstack([A0, np.eye(n)])
    b = np.concatenate([b0, np.ones(n)])
    rel = np.zeros(m + n, dtype=np.int32)
    return c, A, rel, b


def knapsack(n: int = 100_000, seed: int = SEED):
    """Config 5: integer w in [1,1000], p = w + U{0..100}, cap = floor(0.5*sum w)."""
    g = _rng(seed, 5)
    w = g.integers(1, 1001, size=n).astype(np.float64)
    p = w + g.integers(0, 101, size=n).astype(np.float64)
    cap = float(np.floor(0.5 * w.sum()))
    return p, w, cap
