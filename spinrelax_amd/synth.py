"""
Deterministic synthetic bond-vector trajectories (BASELINE.json configs).

The generator uses only integer hashing and IEEE-exact float64 ``+ - * / sqrt`` (no libm
transcendentals, no numpy.random), so the same float32 ``(frames, vectors, 3)`` array is produced
bit-for-bit on the build container and on the GPU box.  Each vector is a fixed mean direction plus
three Ornstein-Uhlenbeck (AR(1)) vector processes with correlation times of roughly 50, 800 and
6000 ps, renormalised to unit length -- i.e. a genuinely multi-exponential internal C(t) with a
plateau, which exercises the model-order selection of the C(t) fit
(reference: fitting_Ct_functions.py:278-304).

Config table (dt = 10 ps, tau_memory = 2*L*dt):

    cfg  frames   V     L     F     R
    1    1 000    32    50    100   10
    2    10 000   128   512   1024  9
    3    100 000  512   2048  4096  24
    4    100 000  2048  2048  4096  24   (sharded, 256 vectors per GPU on 8 GPUs)
"""
import numpy as np
from scipy.signal import lfilter

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_K1 = np.uint64(0x9E3779B97F4A7C15)
_K2 = np.uint64(0xBF58476D1CE4E5B9)
_K3 = np.uint64(0x94D049BB133111EB)

CONFIGS = {
    1: dict(frames=1000, V=32, L=50, seed=1),
    2: dict(frames=10000, V=128, L=512, seed=2),
    3: dict(frames=100000, V=512, L=2048, seed=3),
    4: dict(frames=100000, V=2048, L=2048, seed=4),
}
DT_PS = 10.0
TAUS_PS = (50.0, 800.0, 6000.0)
# README.md:155 example parameters of the reference (ubiquitin)
DISO = 3.7383e-5
DANI = 1.26006
Q_EXT = (0.866165, 0.392069, -0.308123, -0.033159)
ZETA = 0.890023
FIELD_MHZ = 600.133


def _mix(x):
    """splitmix64 finaliser on a uint64 array (in place)."""
    with np.errstate(over='ignore'):
        x ^= x >> np.uint64(30)
        x *= _K2
        x ^= x >> np.uint64(27)
        x *= _K3
        x ^= x >> np.uint64(31)
    return x


def _hash(counter, stream):
    with np.errstate(over='ignore'):
        x = counter.astype(np.uint64) * _K1 + np.uint64((stream * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF)
    return _mix(x)


def _uniform(counter, stream):
    """uniform in [0,1) with 53 bits."""
    return (_hash(counter, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _noise(counter, stream):
    """approximately N(0,1): Irwin-Hall sum of the four 16-bit fields of one 64-bit hash."""
    h = _hash(counter, stream)
    m = np.uint64(0xFFFF)
    s = (h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + (h >> np.uint64(48))
    # each field uniform on 0..65535: mean 32767.5, variance (65536^2-1)/12
    return (s.astype(np.float64) - 131070.0) * (1.0 / 37837.22761670906)


def vector_params(nvec, seed, v0=0):
    """Per-vector mean direction (nvec,3) and OU amplitudes (nvec,3 timescales)."""
    idx = np.arange(v0, v0 + nvec, dtype=np.uint64)
    m = np.stack([2.0 * _uniform(idx * np.uint64(8) + np.uint64(k), seed * 1000 + 1) - 1.0 for k in range(3)], axis=-1)
    m[np.abs(m).sum(axis=1) < 1e-3] = (0.0, 0.0, 1.0)
    m = m / np.sqrt((m * m).sum(axis=1))[:, None]
    base = np.array([0.10, 0.12, 0.14])
    amp = np.stack([base[k] * (0.5 + _uniform(idx * np.uint64(8) + np.uint64(3 + k), seed * 1000 + 1)) for k in range(3)], axis=-1)
    return m, amp


def synth_vectors(nframes, nvec, seed, v0=0, dt=DT_PS, taus=TAUS_PS, vchunk=64):
    """float32 (nframes, nvec, 3) unit vectors.  Vector ``v0+i`` of any call with the same seed is
    identical regardless of nvec/v0, so shards of one trajectory can be generated independently."""
    out = np.empty((nframes, nvec, 3), dtype=np.float32)
    nT = len(taus)
    burn = 256                              # frames discarded so the AR(1) processes are stationary
    T = nframes + burn
    t_idx = np.arange(T, dtype=np.uint64)
    for c0 in range(0, nvec, vchunk):
        c1 = min(nvec, c0 + vchunk)
        nv = c1 - c0
        m, amp = vector_params(nv, seed, v0 + c0)
        acc = np.zeros((nv, 3, T))
        for k, tau in enumerate(taus):
            a = tau / (tau + dt)
            b = np.sqrt(1.0 - a * a)
            lane = ((np.arange(v0 + c0, v0 + c1, dtype=np.uint64)[:, None] * np.uint64(3)
                     + np.arange(3, dtype=np.uint64)[None, :]) * np.uint64(nT) + np.uint64(k))
            ctr = lane[:, :, None] * np.uint64(1 << 24) + t_idx[None, None, :]
            xi = lfilter([b], [1.0, -a], _noise(ctr, seed * 1000 + 2), axis=-1)
            acc += amp[:, k][:, None, None] * xi
        u = m[:, :, None] + acc[:, :, burn:]
        u /= np.sqrt((u * u).sum(axis=1))[:, None, :]
        out[:, c0:c1, :] = np.transpose(u, (2, 0, 1)).astype(np.float32)
    return out


def config_shapes(cfg):
    c = CONFIGS[cfg]
    L = c['L']
    F = 2 * L
    R = c['frames'] // F
    return dict(frames=c['frames'], V=c['V'], L=L, F=F, R=R, N=R * F, tau_memory=F * DT_PS, dt=DT_PS,
                seed=c['seed'])


def exact_triples(R, F, V):
    """R*V*sum_{d=1..L}(F-d): P2 evaluations of one C(t) call."""
    L = F // 2
    return R * V * (L * F - L * (L + 1) // 2)


def synth_config(cfg, nvec=None, v0=0):
    s = config_shapes(cfg)
    return synth_vectors(s['frames'], s['V'] if nvec is None else nvec, s['seed'], v0=v0)
