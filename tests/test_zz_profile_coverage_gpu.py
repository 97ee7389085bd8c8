"""Runs LAST (file name): every conv-family kernel instantiation that the committed rocprofv3 profile of the benchmark
names must have been launched inside a passing, parity-checked test of this session (tests/common.py::traced records
them through tdvc_debug_trace). This is the "every template name in profiles/*kernel_stats.csv appears in a passing
test" gate of VERDICT r1: the benchmark may not time a kernel instance that no parity test has executed.

Only meaningful when the whole GPU suite ran in this session; with a partial selection (-k, single files) it skips.
"""
import csv
import glob
import os

import pytest

from common import SESSION_KERNELS, pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRACED_FAMILIES = ('conv_lean_kernel', 'conv_wgrad_pipe_kernel', 'conv_wgrad_tile_kernel', 'conv_wgrad_lean_kernel', 'conv_gemm_kernel',
                   'conv_wgrad_kernel', 'conv_wgrad_x6_kernel', 'conv_fwd_x6_kernel', 'film_cond_fwd_x6_kernel', 'conv_scalar_kernel', 'conv_wgrad_scalar_kernel', 'film_cond0_bwd_kernel', 'film_cond_bwd_kernel', 'film_block_kernel', 'film_block_fwd_kernel',
                   'wn_gate_kernel', 'small_group_fwd_kernel', 'small_group_dgrad_kernel', 'small_group_wgrad_kernel')


def latest_profile():
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_kernel_stats.csv')))
    assert files, 'no rocprofv3 kernel stats committed under profiles/'
    return files[-1]


def profiled_kernels(path):
    norm = pkg()._lib.normalize_kernel_name
    names = set()
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            n = norm(row['Name'])
            if n.split('<')[0] in TRACED_FAMILIES:
                names.add(n)
    return names


def test_every_profiled_kernel_instance_ran_in_a_parity_test(request):
    ran = {item.fspath.basename for item in request.session.items}
    needed = {'test_kernel_instances_gpu.py', 'test_step_launch_shape_gpu.py', 'test_model_gpu.py', 'test_conv_ops_gpu.py'}
    if not needed <= ran or request.session.testsfailed:
        pytest.skip('partial test selection: the coverage gate needs the whole GPU suite')
    path = latest_profile()
    prof = profiled_kernels(path)
    assert prof, f'{path}: no tdvc conv-family kernels found'
    missing = sorted(prof - SESSION_KERNELS)
    assert not missing, f'{os.path.basename(path)}: kernel instances timed by the benchmark but never run by a parity test: {missing}'
    print(f'\n{len(prof)} kernel instances of {os.path.basename(path)} all covered ({len(SESSION_KERNELS)} instances run by tests)')
