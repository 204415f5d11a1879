"""Sketches of large tensors for golden fixtures: instead of 2 million gradient values, store the 2-norm, 8 dot products
with fixed +-1 vectors and a strided sample.  For a difference d = test - reference the projections estimate |d|_2
(E[(r.d)^2] = |d|^2 for a +-1 vector r), so a test gets the relative L2 error of the whole tensor from 8 numbers.
Shared by the generators (tests/golden/make_golden_*.py) and the tests."""
import numpy as np

WHOLE_MAX = 4096      # tensors up to this many elements are stored whole
N_PROJ = 8
N_SAMPLE = 2048


def sketch_signs(n: int, k: int) -> np.ndarray:
    """+-1 vector number k of length n: one bit of a 32-bit integer hash of (index, k).  Pure integer arithmetic, no RNG, so
    generator and test build identical vectors on any machine."""
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(k) * np.uint64(0x9E3779B1) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    h = ((h ^ (h >> np.uint64(15))) * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h = h ^ (h >> np.uint64(13))
    return ((h >> np.uint64(16)) & np.uint64(1)).astype(np.float64) * 2.0 - 1.0


def sample_index(n: int) -> np.ndarray:
    return np.linspace(0, n - 1, N_SAMPLE).astype(np.int64)


def sketch(t) -> dict:
    """t: torch tensor (any device) -> {'norm', 'proj' [N_PROJ], 'sample' [N_SAMPLE]}"""
    v = t.detach().double().reshape(-1).cpu().numpy()
    n = v.size
    proj = np.array([float(np.dot(sketch_signs(n, k), v)) for k in range(N_PROJ)])
    return dict(norm=np.float64(np.linalg.norm(v)), proj=proj, sample=v[sample_index(n)].astype(np.float32))


def store(out: dict, key: str, t) -> None:
    """out[key + '/whole'] for small tensors, out[key + '/norm|proj|sample'] for large ones"""
    if t.numel() <= WHOLE_MAX:
        out[key + '/whole'] = t.detach().float().cpu().numpy()
    else:
        for k, v in sketch(t).items():
            out[key + '/' + k] = v


def rel_err_vs(z, key: str, t) -> float:
    """Relative L2 error of tensor t against the fixture entry `key` of the npz z (exact for whole entries; the RMS of the
    projection differences over the reference norm -- an unbiased estimate of the same ratio -- for sketched ones, combined
    with the exact error of the strided sample so that a localised defect cannot hide behind a lucky projection)."""
    v = t.detach().double().reshape(-1).cpu().numpy()
    if key + '/whole' in z.files:
        ref = z[key + '/whole'].astype(np.float64).reshape(-1)
        return float(np.linalg.norm(v - ref) / (np.linalg.norm(ref) + 1e-30))
    norm = float(z[key + '/norm'])
    proj = np.array([float(np.dot(sketch_signs(v.size, k), v)) for k in range(N_PROJ)])
    est = float(np.sqrt(np.mean((proj - z[key + '/proj']) ** 2)) / (norm + 1e-30))
    smp_ref = z[key + '/sample'].astype(np.float64)
    smp = v[sample_index(v.size)]
    smp_err = float(np.linalg.norm(smp - smp_ref) / (np.linalg.norm(smp_ref) + 1e-30))
    return max(est, smp_err)
