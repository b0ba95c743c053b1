"""Multi-process (world_size 2, gloo, CPU) test of the read-sharding path used by
bench.py --gpus N and by multi-GPU callers (psi_amd/dist.py)."""
import os
import socket
import subprocess
import sys

from psi_amd.dist import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_gather(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = tmp_path / 'result.txt'
    env = dict(os.environ, PSI_AMD_NO_TORCH='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'tests', '_dist_worker.py'), str(out)]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert out.read_text().startswith('OK 5010')


def _run_two_ranks(script_args, env_extra, timeout=900):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PSI_AMD_NO_TORCH='0', **env_extra)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port)] + script_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_one_gpu_through_the_hip_path(tmp_path):
    """Both ranks share the box's one GPU (collectives on gloo): every rank answers its read range on the
    device (patched index, sort-unique on the device), rank 0 gathers; the gathered array is the golden
    hit set in sorted order."""
    out = tmp_path / 'result.txt'
    p = _run_two_ranks([os.path.join(ROOT, 'tests', '_dist_worker.py'), str(out)], {'PSI_DIST_HIP': '1'})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert out.read_text().startswith('OK 5010')


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu():
    """bench.py's N > 1 path end to end (PSI_BENCH_BACKEND=gloo: ranks share the GPU): one JSON line, weak
    scaling, hit lists gathered on rank 0 in read-id order."""
    import json
    p = _run_two_ranks([os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--wg', 'off',
                        '--reads', '20000', '--backbone', '3000000', '--snvs', '60000', '--nblock', '200000'],
                       {'PSI_BENCH_BACKEND': 'gloo'})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.split('\n') if l.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['scaling'] == 'weak' and j['steps'] == 3
    assert j['config']['seeds_per_step_per_gpu'] == 140000 and j['value'] > 0
    assert 'fell back to configs[1]' in j['config']['workload'] and j['config']['whole_genome'] is False
    g = j['gather_hits']
    assert g['records'] >= 2 * 140000 and g['sorted_by_read_id'] and g['backend'] == 'gloo'


@pytest.mark.gpu
def test_bench_two_ranks_whole_genome_path_at_reduced_size(tmp_path):
    """bench.py --gpus N, N > 1, as the driver runs it: BASELINE.json configs[3] -- rank 0 builds ONE host index and
    writes the arrays behind the views to a shared directory, every rank maps them (psi_amd/shared.py), simulates its
    own contiguous range of reads, and the line carries per-GPU rates, the end-to-end rate through every rank's host
    entry at once, both gathers and the found-where-sampled property.  Here at 3 Mbp with two ranks on the box's one
    GPU (PSI_BENCH_BACKEND=gloo), the same code path; a second run finds the directory and builds nothing."""
    import json
    env = {'PSI_BENCH_BACKEND': 'gloo', 'PSI_BENCH_SHARE_DIR': str(tmp_path)}
    cmd = [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--wg', 'force',
           '--wg-backbone', '3000000', '--wg-snvs', '60000', '--wg-nblock', '200000', '--wg-reads', '20000',
           '--full-out', str(tmp_path / 'full.json')]
    for cached in (False, True):
        p = _run_two_ranks(cmd, env)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        lines = [l for l in p.stdout.split('\n') if l.startswith('{')]
        assert len(lines) == 1 and len(lines[0]) < 6100           # (the line the driver parses: small, round 5)
        line = json.loads(lines[0])
        assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0 and line['roofline'] and 'cpu_baseline' in line
        assert line['config']['whole_genome'] is True and line['multi_gpu']['n_ranks'] == 2 and line['value_end_to_end'] > 0
        j = json.load(open(tmp_path / 'full.json'))               # ... and the full report beside it
        assert j['n_gpus'] == 2 and j['scaling'] == 'weak' and j['value'] == pytest.approx(line['value'], rel=1e-4)
        c = j['config']
        assert c['whole_genome'] is True and 'configs[3]' in c['workload'] and c['shared_index_cached'] is cached
        assert c['seeds_per_step_per_gpu'] == 140000 and c['reads_per_gpu'] == 20000
        m = j['multi_gpu']
        assert len(m['per_gpu']) == 2 and all(x['seeds_per_s'] > 0 for x in m['per_gpu'])
        assert m['properties']['every_seed_of_the_first_reads_found_on_every_rank'] is True
        e = m['end_to_end_all_links']
        assert e['value'] > 0 and len(e['per_gpu_ms']) == 2 and min(e['records_per_gpu']) >= 140000
        g = j['gather_hits']
        assert g['records'] >= 2 * 140000 and g['sorted_by_read_id']


@pytest.mark.gpu
def test_config3_harness_three_contexts_one_gpu():
    """tools/wg_scale.py --devices: the harness for BASELINE.json configs[3] (whole-genome graph, 100 M reads over
    the 8 GPUs of a node) at reduced size, the box's one GPU listed three times -- ONE process, ONE host index,
    one context + host thread per listed device, contiguous read ranges with global read ids; psikt's default
    indexing (3 patched walks) cut into several index parts, answered from the FM index of every part
    (locus-table mode), compared with the k-mer table mode's records; MEM mode over the parts; the host entry
    point through every context."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'wg_scale.py'), '--backbone', '3000000', '--snvs', '60000',
           '--nblock', '200000', '--reads', '30000', '--devices', '0,0,0', '--paths', '3', '--patched',
           '--max-part-text', '1500000', '--mode', 'locus-table', '--compare-modes', '--mems', '1500', '--host-entry',
           '--steps', '2']
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    j = json.loads([l for l in p.stdout.split('\n') if l.startswith('{')][-1])
    assert j['n_devices'] == 3 and j['index_parts'] >= 2 and j['query_mode'] == 'locus-table'
    assert [d['reads'] for d in j['per_device']] == [10000, 10000, 10000]
    assert j['all_seeds_found'] and j['first_base_agrees'] and j['same_records_as_kmer_table_mode']
    assert j['counters']['n_seeds'] == 70000 and j['seeds_per_s'] > 0
    assert j['mems_records'] >= 1500 and j['mems_first_pattern_subset_of_seed_hits'] and j['mems_min_len'] >= 21
    assert j['host_entry_equals_device_entry'] and len(j['host_entry_ms_per_device']) == 3
