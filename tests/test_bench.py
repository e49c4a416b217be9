"""bench.py contract checks: the multi-GPU launcher logic (CPU) and, on the GPU box, the RCCL code path with one rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env.pop('LOCAL_RANK', None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_gpus_n_without_enough_devices_fails_loudly():
    """`python bench.py --gpus N` (how the driver starts it) must never run a smaller job under the label n_gpus=N."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(['--gpus', str(have + 2), '--steps', '1', '--warmup', '0'], timeout=300)
    assert r.returncode != 0
    assert f'only {have} GPU(s) visible' in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_force_dist_runs_the_rccl_path_on_one_gpu():
    """1-rank NCCL (= RCCL) process group: PipelinedGather's all_gather_into_tensor on a side stream, slot check, JSON line."""
    r = _run(['--force-dist', '--only-headline', '--no-cpu-baseline', '--steps', '3', '--warmup', '1', '--batch', '2', '--size', '256'])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and 'RCCL all-gather' in line['config']['parallelism']
    assert line['dtype'].startswith('f32 (every product and sum an IEEE f32') and line['step_ms']['n'] == 3
    rf = line['roofline']
    assert rf['bound'] == 'mfma' and 0 < rf['frac'] < 1 and rf['flops_executed_per_launch_avg'] <= rf['flops_dense_per_launch_avg'] * 1.001


@pytest.mark.gpu
def test_default_line_carries_every_block():
    """The one driver-run line: exact-f32 top level + f16x3 block + face-like labels + Config 3 pipeline (small sizes here)."""
    r = _run(['--no-cpu-baseline', '--steps', '3', '--warmup', '1', '--batch', '2', '--size', '256'])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'f32_class_f16x3', 'face_like_labels', 'pipeline', 'sustained_peaks'):
        assert k in line, k
    assert line['config']['conv_path'] == 'f32' and line['f32_class_f16x3']['value'] > 0
    assert set(line['pipeline']) >= {'f16x3', 'f32'} and 'stages' in line['pipeline']['f16x3']
    assert line['face_like_labels']['f32']['value'] > 0
    assert line['host_enqueue_ms_per_step']['n'] == 3 and line['host_enqueue_ms_per_step']['median'] > 0
    for leg in ('f16x3', 'f32'):         # stage rows: executed FLOPs against the peak of the unit that ran them -- never above it
        for name, row in line['pipeline'][leg]['stages'].items():
            if row.get('bound') == 'mfma':
                assert 0 <= row['frac'] < 1 and 0 < row['executed_tflops'] <= row['peak_tflops'], (leg, name, row)      # (tiny here: 2 x 256^2)
