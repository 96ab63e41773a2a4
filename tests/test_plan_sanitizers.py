"""The host-side plan builder (csrc/plan.cpp: index tables, use lists, workspace layouts, compose geometry) under AddressSanitizer
and UBSan.  GPU sanitizers are not available on the MI355X pool, so the CPU build is where the native host code is checked."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='no g++')
def test_plan_builder_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / 'plan_asan')
    cmd = ['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-o', exe,
           os.path.join(ROOT, 'tests', 'native', 'plan_asan_driver.cpp'), os.path.join(ROOT, 'cliora_amd', 'csrc', 'plan.cpp')]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1'))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'plans ok' in r.stdout
