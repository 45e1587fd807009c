"""The fused sweep's update schedule (elfi_amd/csrc/sweep_sched.hpp) replayed on the host for every block-column count it
serves: each tile half receives exactly the k range of the factor it must, contiguously and in order, never from a
panel that is not solved yet, created once (L^-T tiles), complete at its deadline, and never from two workgroups in one
step.  The header is host-only C++; the checker (tests/native/sweep_sched_check.cpp) is compiled with g++ here."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def checker(tmp_path_factory):
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('no g++')
    exe = str(tmp_path_factory.mktemp('sched') / 'sweep_sched_check')
    subprocess.run([gxx, '-O2', '-std=c++17', '-I', os.path.join(ROOT, 'elfi_amd', 'csrc'), '-o', exe,
                    os.path.join(ROOT, 'tests', 'native', 'sweep_sched_check.cpp')], check=True)
    return exe


@pytest.mark.parametrize('nwg', [255, 248, 96, 8])
def test_every_schedule_is_complete_and_in_order(checker, nwg):
    hi = 96 if nwg == 255 else (64 if nwg == 248 else 40)     # the library runs 255 workgroups up to 96 block columns
    r = subprocess.run([checker, '2', str(hi), str(nwg)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('nb')]
    assert len(lines) == hi - 1 and all(l.rstrip().endswith('ok') for l in lines)


def test_the_schedule_levels_the_steps(checker):
    """n = 4096 (32 block columns) on 255 workgroups: every step's update is about as long as the diagonal block (35 us)
    -- the plain right-looking order has three units per workgroup (63 us) in the first third and idles in the last."""
    r = subprocess.run([checker, '32', '32', '255', 'v'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    steps = [float(x.split('/')[0]) for x in r.stdout.splitlines()[1].split()]
    med = sorted(steps)[15]
    # level from the fourth step on; what does not fit lands in the FIRST steps (backward construction), never at the end
    assert len(steps) == 31 and 34.0 <= med <= 40.0 and max(steps[3:]) <= 1.05 * med and max(steps[:3]) <= 64.0
    assert sum(max(s_, 35.0) for s_ in steps) <= 31 * 39.5
