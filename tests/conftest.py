import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


_SPAWNER = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on 8 CPU cores")
    # GPU sessions: start the rank launcher NOW, while this process has not touched the GPU (collection may:
    # torch.cuda.is_available() in a skipif).  Multi-rank GPU tests ask it for fresh processes (tools/dp_spawner.py).
    global _SPAWNER
    expr = config.getoption("-m") or ""
    if "gpu" in expr and "not gpu" not in expr and _SPAWNER is None:
        import subprocess
        _SPAWNER = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "dp_spawner.py")],
                                    stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)


def pytest_unconfigure(config):
    global _SPAWNER
    if _SPAWNER is not None:
        try:
            _SPAWNER.stdin.close()
            _SPAWNER.wait(timeout=30)
        except Exception:
            _SPAWNER.kill()
        _SPAWNER = None


@pytest.fixture(scope="session")
def rank_launcher():
    """run(argv, ranks=2, env=None, timeout=600) -> {"rc": [...], "tail": [...]}: ranks as fresh processes."""
    if _SPAWNER is None:
        pytest.skip("rank launcher only runs in `-m gpu` sessions")
    import json

    def run(argv, ranks=2, env=None, timeout=600):
        _SPAWNER.stdin.write(json.dumps({"argv": argv, "ranks": ranks, "env": env or {}, "timeout": timeout}) + "\n")
        _SPAWNER.stdin.flush()
        line = _SPAWNER.stdout.readline()
        assert line, "rank launcher died"
        return json.loads(line)
    return run


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import ops
    ops.build()
    return ops
