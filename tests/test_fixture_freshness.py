"""The committed fixtures of tests/golden/ are what their committed generators make (build container only: the generators execute
the reference where it lies under /root/reference; on the GPU box these tests skip).

Round 4 shipped a core_cases.npz that its generator no longer reproduced — the reader had changed underneath it — and nothing
noticed.  The cheap generators are re-run whole into a scratch directory and compared byte for byte; of the slow one
(make_golden_core.py: minutes) the section that depends on this repository's own reader is re-run and compared array by array."""
import filecmp
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden')
pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/arpeggio'), reason='the generators execute /root/reference')

CHEAP = {
    'make_golden_reader.py': ['reader.json'],
    'make_golden_struct_conn.py': ['struct_conn.json'],
    'make_golden_typing.py': ['typing.json'],
    'make_golden_prepare.py': ['prepare_cases.npz'],
    'make_golden.py': ['angles.npz', 'group_angles.npz', 'norm_f32.npz', 'hbond.npz', 'contact_type.json', 'sift_updates.json',
                       'planes_input.npz', 'planes_expected.json', 'selection_parser.json'],
}


@pytest.mark.parametrize('script', sorted(CHEAP))
def test_cheap_generators_reproduce_the_committed_fixtures(script, tmp_path):
    env = dict(os.environ, ARP_GOLDEN_OUT=str(tmp_path), PYTHONHASHSEED='0')
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    for name in CHEAP[script]:
        assert filecmp.cmp(os.path.join(GOLDEN, name), str(tmp_path / name), shallow=False), f'{name} is not what {script} makes'


def test_reader_cases_of_the_core_fixture_are_what_the_generator_makes():
    sys.path.insert(0, GOLDEN)
    try:
        import make_golden_core
        assert make_golden_core.check_reader() > 100
    finally:
        sys.path.remove(GOLDEN)
