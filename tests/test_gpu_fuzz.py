"""A short run of tools/fuzz_parity.py: random pool sizes across the dispatch thresholds (1 .. 16001 candidates, 1 .. 32
sentences, 1 .. 5 queries, ragged and uniform, a candidate sharing a sentence with the query): sampled otAspire /
tsAspire scores against the oracle at 1e-4 and every query's rank against the stable sort of its own scores; the padded
caching_score pattern; batched jobs (every query against its own pool, ragged pools, a few long documents among short ones)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_parity_short():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_parity.py'), '14', '11'], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert '14 cases ok' in out.stdout and 'caching_score cases ok' in out.stdout and 'batched cases ok' in out.stdout
