"""CPU: `python bench.py --gpus N` starts its own ranks (the driver's contract command, no launcher in front) and rank 0
prints ONE JSON line.  Rehearsed with `--dry-run`: gloo instead of RCCL, a stand-in operator instead of the HIP kernel —
the launch, shard-plan, pipelined all-gather, barrier / max-over-ranks timing and reporting logic are the real ones."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, 'bench.py', '--dry-run', *args], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_self_launched_strong_scaling_with_gather():
    d = _run('--gpus', '2', '--steps', '6', '--warmup', '2', '--total-pairs', '50000')
    assert d['n_gpus'] == 2 and d['steps'] == 6 and d['warmup'] == 2 and d['unit'] == 'pairs/s'
    assert d['scaling'] == 'strong' and d['config']['total_pairs'] == 50000 and d['config']['pairs_per_gpu'] == 25000
    assert d['config']['gather'] is True and d['config']['parallelism'] == 'shard2'
    assert abs(d['value'] - 50000 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert d['no_gather']['value'] > 0 and d['gather_only']['ms_per_step'] > 0 and d['strong_scaling']['total_pairs'] == 50000
    assert d['vs_baseline'] is None and d['higher_is_better'] is True and 'model' not in d['config']


def test_two_ranks_default_is_configs1_per_gpu_without_collective_and_the_north_star_exchange_beside_it():
    """The driver's command for N = 2.  `value` = the ranks' own batches, no data-path collective ("weak"); the north star's
    configs[4] (8 M pairs split over the ranks + one all-gather per step) is measured in the same job and reported beside it."""
    d = _run('--gpus', '2', '--steps', '4', '--warmup', '1')
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['gather'] is False
    assert d['config']['total_pairs'] == 2_000_000 and d['config']['pairs_per_gpu'] == 1_000_000 and 'configs[1]' in d['config']['workload']
    assert abs(d['value'] - 2_000_000 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    ns = d['north_star_configs4']
    assert ns['total_pairs'] == 8_000_000 and ns['pairs_per_gpu'] == 4_000_000 and ns['own_shard_intact_in_gathered'] is True
    for key in ('with_gather', 'no_gather', 'gather_only'):
        assert ns[key]['ms_per_step'] > 0
    assert ns['gather_only']['bytes_received_per_rank'] == 16_000_000
    assert ns['with_gather']['ms_per_step'] >= 0.5 * max(ns['no_gather']['ms_per_step'], ns['gather_only']['ms_per_step'])


def test_default_batches_are_configs_1_per_gpu_and_configs_4_on_request():
    sys.path.insert(0, ROOT)
    import bench
    for n in (1, 2, 4, 8):
        counts, total, label = bench.shard_plan(bench.parse_args(['--gpus', str(n)]), n)
        assert counts == [1_000_000] * n and total == n * 1_000_000 and label == 'weak'
    for n in (2, 4, 8):
        counts, total, label = bench.shard_plan(bench.parse_args(['--gpus', str(n), '--scaling', 'strong']), n)
        assert total == 8_000_000 and sum(counts) == total and max(counts) - min(counts) == 0 and label == 'strong'
    counts, total, label = bench.shard_plan(bench.parse_args(['--gpus', '3', '--total-pairs', '1000000']), 3)
    assert label == 'strong' and sum(counts) == 1_000_000 and max(counts) - min(counts) <= 1      # ragged shards: first ranks get the extra pair


def test_weak_scaling_without_gather_and_one_rank_plain():
    d = _run('--gpus', '2', '--steps', '4', '--warmup', '1', '--scaling', 'weak', '--pairs', '7000', '--no-gather')
    assert d['scaling'] == 'weak' and d['config']['total_pairs'] == 14000 and d['config']['gather'] is False and 'no_gather' not in d
    d = _run('--steps', '4', '--warmup', '1', '--pairs', '3000')
    assert d['n_gpus'] == 1 and d['config']['total_pairs'] == 3000 and d['config']['gather'] is False
