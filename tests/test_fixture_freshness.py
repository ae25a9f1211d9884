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


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get('ARP_RUN_SLOW') != '1', reason='4 - 5 minutes of CPU: ARP_RUN_SLOW=1 python -m pytest tests/test_fixture_freshness.py -m slow')
def test_the_whole_core_generator_reproduces_the_committed_fixture_without_the_hip_library(tmp_path):
    """make_golden_core.py, whole, in a copy of the tracked files with NO built library in it (the mmCIF reader it needs comes from
    libarpeggio_host.so, g++ only): all arrays of core_cases.npz, core_cases.json and the exports byte for byte."""
    import shutil
    import numpy as np
    root = os.path.dirname(HERE)
    copy = tmp_path / 'checkout'
    files = subprocess.run(['git', 'ls-files', '-z'], cwd=root, capture_output=True, check=True).stdout.decode().split('\0')
    for f in filter(None, files):
        dst = copy / f
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(root, f), dst)
    assert not list(copy.rglob('*.so'))
    out = tmp_path / 'out'
    out.mkdir()
    env = dict(os.environ, ARP_GOLDEN_OUT=str(out), PYTHONHASHSEED='0')
    env.pop('ARP_LIB_PATH', None)
    r = subprocess.run([sys.executable, str(copy / 'tests' / 'golden' / 'make_golden_core.py')], env=env, capture_output=True, text=True, timeout=3600, cwd=str(copy))
    assert r.returncode == 0, r.stderr[-2000:]
    assert not (copy / 'arpeggio_amd' / 'csrc' / 'libarpeggio_hip.so').exists()
    for name in ('core_cases.json', 'core_exports.json.gz'):
        assert filecmp.cmp(os.path.join(GOLDEN, name), str(out / name), shallow=False), name
    x, y = np.load(out / 'core_cases.npz'), np.load(os.path.join(GOLDEN, 'core_cases.npz'))
    assert sorted(x.files) == sorted(y.files) and len(x.files) > 8000
    for k in x.files:
        assert x[k].dtype == y[k].dtype and x[k].shape == y[k].shape and np.array_equal(x[k], y[k]), k
