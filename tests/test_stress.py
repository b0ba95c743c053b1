"""Hit sets under GPU SHARING (round-3 review, item 1): several contexts + host threads in one process and several
processes on the one GPU, against the brute-force definition; zero mismatches and zero abnormal exits required.
The reference is single-threaded and deterministic (include/psi/seed_finder.hpp:1724-1732): so must every context be,
whatever runs beside it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


N_GRAPHS = int(os.environ.get('PSI_STRESS_GRAPHS', '320'))


def test_contexts_threads_and_processes_share_the_gpu():
    """4 processes x 4 threads (a context per thread at a time, finders made and destroyed all along) over random
    graphs: device entry and host entry (pageable / pinned / packed reads, raw / sorted, 8- / 16- / 32-byte wire
    records, sub-batches of 16 bytes to one piece, transfers queued ahead or not) all equal to the definition.
    A finder lifetime costs seconds when sixteen threads make and drop tables on one device (every table build waits
    for the device to be idle), so the collected test runs 320 graphs (~2 minutes; 640 until round 6 -- the GPU suite has 1 200 s); the campaigns ran the same
    tool over 1 200 and 2 000 (PSI_STRESS_GRAPHS=2000; profiles/r04_load_campaign.json)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'stress.py'), '--procs', '4', '--threads', '4',
                        '--graphs', str(N_GRAPHS), '--lifetimes', '1', '--calls', '3', '--timeout', '900'],
                       capture_output=True, text=True, timeout=1100)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else '{}'
    res = json.loads(line)
    assert r.returncode == 0 and res.get('mismatches') == 0 and res.get('abnormal_exits') == [] and not res.get('errors'), \
        (r.returncode, line[:4000], r.stderr[-2000:])
    assert res['graphs'] == N_GRAPHS and res['calls'] >= N_GRAPHS * 4 and res.get('stale_handbacks', 0) == 0


@pytest.mark.parametrize('seed', [147003, 163032])
def test_seeds_that_once_mismatched_under_load(seed):
    """The two fuzz seeds whose hit sets differed once under eight processes sharing the GPU in round 3 (one record
    missing: 147003, k-mer table without a path index; two extra under a gocc threshold: 163032) -- the whole fuzz
    configuration of each seed, four processes at once, twice."""
    for _ in range(2):
        ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_modes.py'), str(seed), str(seed + 1)],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(4)]
        for p in ps:
            out, _ = p.communicate(timeout=600)
            assert p.returncode == 0 and 'MISMATCH' not in out, out[-2000:]
