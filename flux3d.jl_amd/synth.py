"""Deterministic synthetic inputs (BASELINE.md "Inputs"): a documented counter-based generator so
that every process / rank / language regenerates bit-identical clouds without any file.

  state_k = seed + (k+1) * 0x9E3779B97F4A7C15          (SplitMix64, Steele et al. 2014)
  z = state_k; z = (z ^ (z>>30)) * 0xBF58476D1CE4E5B9; z = (z ^ (z>>27)) * 0x94D049BB133111EB; z ^= z>>31
  u_k = (z >> 40) * 2^-24                               (Float32 in [0,1), 24 bits)
"""
import numpy as np

SEED_A = 0x5EED0001
SEED_B = 0x5EED0002


def splitmix_uniform(seed, count, offset=0):
    k = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def uniform_cloud(seed, D, N, B, batch_offset=0):
    """(D,N,B) Float32 U[0,1)^D, F-ordered; batch element b uses stream positions
    [(batch_offset+b)*D*N, ...), so a shard is a slice of the global cloud."""
    u = splitmix_uniform(seed, D * N * B, offset=batch_offset * D * N)
    return np.asfortranarray(u.reshape((D, N, B), order="F"))


def reference_bench_cloud(npoints):
    """generate_pcloud of the reference harness (benchmarks/metrics.jl:11-15): p_i = (i,i,i)/n."""
    p = np.ones((3, npoints), np.float32)
    return np.asfortranarray(np.cumsum(p, axis=1, dtype=np.float32) / np.float32(npoints))
