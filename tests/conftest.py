import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'timeout_s(n): watchdog limit of this test in seconds (default YOLO2_TEST_TIMEOUT or 300)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


# A hung GPU kernel blocks the host inside hipDeviceSynchronize, where no Python-level time-out can interrupt it; on the GPU box that costs the
# whole lease, not one test.  Every test therefore runs under a watchdog: after YOLO2_TEST_TIMEOUT seconds (default 300; a marker
# @pytest.mark.timeout_s(n) overrides it per test) faulthandler dumps every thread's stack and the process exits.  The library's own
# device-side waits are bounded as well (csrc/conv_shared.h y2_sk_wait_and_clear), so this is the second line of defence.
import faulthandler

_DEFAULT_TIMEOUT_S = float(os.environ.get('YOLO2_TEST_TIMEOUT', 300))


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_protocol(item, nextitem):
    marker = item.get_closest_marker('timeout_s')
    limit = float(marker.args[0]) if marker and marker.args else _DEFAULT_TIMEOUT_S
    if limit > 0:
        sys.stderr.flush()
        faulthandler.dump_traceback_later(limit, exit=True)
    try:
        yield
    finally:
        faulthandler.cancel_dump_traceback_later()
