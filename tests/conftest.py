import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
# the cached activation geometry of gcc_amd.ops.geom is re-derived and compared on every use under the tests
os.environ.setdefault('GCC_DEBUG_GEOM', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """One line over every logged-loss scalar the session's model tests compared (tests/_updates.LossBars): how many passed against the
    reference / fp32 oracle, how many only against the bf16-emulating oracle -- appended to GCC_TEST_REPORT"""
    try:
        from tests import _updates
    except Exception:
        return
    log = _updates.LOSS_BAR_LOG
    if not log:
        return
    only = [(f, l) for f, l, e_ref, e_emul in log if e_ref > 1.0 and e_emul <= 1.0]
    neither = [(f, l) for f, l, e_ref, e_emul in log if e_ref > 1.0 and e_emul > 1.0]
    _updates._report('SESSION logged losses: %d scalars over %d families; %d (%.1f %%) pass only against the bf16-emulating oracle: %s; '
                     '%d within neither bar (judged by their tests\' own stated bars): %s' % (
                         len(log), len({f for f, _, _, _ in log}), len(only), 100.0 * len(only) / len(log),
                         ', '.join('%s/%s' % fl for fl in only) or '-', len(neither), ', '.join('%s/%s' % fl for fl in neither) or '-'))
