"""Seeded synthetic event streams.

Per read r: seed = splitmix64(0x6E616E6F ^ r); the read's random numbers are the splitmix64 stream
of that seed (output n = mix(seed + (n+1)*GAMMA)), which is counter-based and so vectorises over
reads and events.  Each event consumes 6 outputs: [move u, base bits, bm u1, bm u2, length u, spare].

Hidden path: k0 = bits & 4095; then u < .10 stay, u < .70 step k = ((k<<2)|b) & 4095, else skip-1
k = ((k<<4)|bb) & 4095.  Emission from the UNSCALED builtin table: mean = mu_k + sigma_k*z1,
stdv = max(.05, eta_k + .3*sd_stdv_k*z2) (Box-Muller in double), length = .01 + .02*u, start =
cumulative length; all cast to float32.  Host prep (corrected_mean, log_stdv) is NOT done here --
use nanocall_amd.api.events_prepare, which mirrors the reference.
"""
import numpy as np

GAMMA = np.uint64(0x9E3779B97F4A7C15)
PER_EVENT = 6


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _u01(x):
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def read_seeds(read_ids):
    with np.errstate(over="ignore"):
        r = np.asarray(read_ids, dtype=np.uint64)
        return _mix((np.uint64(0x6E616E6F) ^ r) + GAMMA)


def generate(table_Sx4, n_reads, n_events, first_read=0, return_path=False):
    """Returns dict(mean, stdv, start, length: float32 [n_reads, n_events]) (+ path uint16)."""
    t = np.asarray(table_Sx4, dtype=np.float32).reshape(4096, 4).astype(np.float64)
    mu, sigma, eta, sd_stdv = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    with np.errstate(over="ignore"):
        seeds = read_seeds(np.arange(first_read, first_read + n_reads, dtype=np.uint64))[:, None]
        ctr = (np.arange(n_events, dtype=np.uint64) * np.uint64(PER_EVENT))[None, :]

        def draw(c):
            return _mix(seeds + (ctr + np.uint64(c + 1)) * GAMMA)

        u_move = _u01(draw(0))
        bits = draw(1)
        u1 = _u01(draw(2))
        u2 = _u01(draw(3))
        u_len = _u01(draw(4))
    # hidden k-mer walk: sequential in the event index, vectorised over reads
    path = np.empty((n_reads, n_events), np.uint16)
    k = (bits[:, 0] & np.uint64(4095)).astype(np.int64)
    path[:, 0] = k
    b2 = (bits & np.uint64(3)).astype(np.int64)
    b4 = (bits & np.uint64(15)).astype(np.int64)
    for i in range(1, n_events):
        u = u_move[:, i]
        step = ((k << 2) | b2[:, i]) & 4095
        skip = ((k << 4) | b4[:, i]) & 4095
        k = np.where(u < 0.10, k, np.where(u < 0.70, step, skip))
        path[:, i] = k
    # Box-Muller (double)
    rad = np.sqrt(-2.0 * np.log(1.0 - u1))
    z1 = rad * np.cos(2.0 * np.pi * u2)
    z2 = rad * np.sin(2.0 * np.pi * u2)
    p = path.astype(np.int64)
    mean = (mu[p] + sigma[p] * z1).astype(np.float32)
    stdv = np.maximum(0.05, eta[p] + 0.3 * sd_stdv[p] * z2).astype(np.float32)
    length = (0.01 + 0.02 * u_len)
    start = (np.cumsum(length, axis=1) - length).astype(np.float32)
    out = dict(mean=mean, stdv=stdv, start=start, length=length.astype(np.float32))
    if return_path:
        out["path"] = path
    return out


def flat_batch(ev):
    """[n_reads, n_events] arrays -> (off uint64[n_reads+1], flat mean, stdv, start)."""
    n_reads, n_events = ev["mean"].shape
    off = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(n_events))
    return off, ev["mean"].reshape(-1), ev["stdv"].reshape(-1), ev["start"].reshape(-1)
