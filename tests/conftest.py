import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: long CPU oracle runs')
    # the CPU oracles run on torch's thread pool: sized by the visible cores (256 on the GPU box) it spins the container's 16-CPU CFS
    # quota away and every oracle forward runs throttled (4-5x slower): keep the pool inside the quota
    try:
        from ctrlhair_amd.hostutil import cap_threads_to_cpu_quota
        cap_threads_to_cpu_quota()
    except Exception:
        pass


@pytest.fixture(scope='session')
def hip_lib():
    """The C-ABI library; GPU tests must run on the real HIP path or fail (no fallback)."""
    from ctrlhair_amd import lib
    return lib.load()


def _memoize_weights():
    """Procedural weights are pure functions of their arguments and the ngf = 64 generator dict alone is 1 GB of Philox draws plus
    spectral power iterations (3 s per call on the GPU box, 30 s on a small CPU container): build each set once per pytest session.
    Every call returns a fresh (shallow) dict, so tests may replace entries -- they must not write INTO the arrays."""
    import functools
    from ctrlhair_amd import procedural as P

    def wrap(fn):
        @functools.lru_cache(maxsize=None)
        def cached(*a, **k_items):
            out = fn(*a, **dict(k_items))
            return out

        @functools.wraps(fn)
        def call(*a, **k):
            out = cached(*a, **{kk: vv for kk, vv in sorted(k.items())})
            return {kk: (dict(vv) if isinstance(vv, dict) else vv) for kk, vv in out.items()}
        return call

    for name in ('sean_state_dict', 'shape_state_dict', 'color_state_dicts', 'bisenet_state_dict'):
        setattr(P, name, wrap(getattr(P, name)))


_memoize_weights()
