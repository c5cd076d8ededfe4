import numpy as np

from pylrbms_amd.online_enrichment import doerfler_marking


def test_doerfler_marking():
    """online_enrichment.py:9-22."""
    ind = [0.1, 0.5, 0.2, 0.4]
    assert doerfler_marking(ind, 1.0) == [1, 3, 2, 0]
    assert doerfler_marking(ind, 0.5) == [1]               # 0.25 > 0.5 * 0.46
    assert doerfler_marking(ind, 0.8) == [1, 3]            # 0.25 + 0.16 = 0.41 > 0.368
    assert sorted(doerfler_marking(np.ones(5), 0.99)) == [0, 1, 2, 3, 4]
