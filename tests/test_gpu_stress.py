"""GPU: adversarial differential test (tools/stress_compare.py at a reduced size) — poles, the theta seam, tiny and
huge boxes, identical / near-identical pairs, integer degrees, |gamma| up to 360: the closed-form core must stay in
range and at least as close to the f64 oracle as the reference-order kernels on every set."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_adversarial_sets_fast_vs_reference_order_vs_truth():
    os.environ['SPH2POB_STRESS_N'] = '40000'
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import stress_compare
    bad, rows = stress_compare.run()
    assert bad == 0
    for dim, name, v, df, dr, dd in rows:
        assert np.quantile(df, 0.999) <= 3 * np.quantile(dr, 0.999) + 1e-4, (dim, name, v)
        assert df.mean() <= 3 * dr.mean() + 2e-6, (dim, name, v, df.mean(), dr.mean())
        assert df.max() <= max(3 * dr.max(), 0.1), (dim, name, v, df.max(), dr.max())
