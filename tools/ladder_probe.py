"""Which kernel does the dispatch (csrc/capi.hip run_down / run_up / run_wgrad) land every geometry on?

    python tools/ladder_probe.py [out.json]

Walks tests/test_gpu_kernels.py's named cases plus the ladder cases of tests/ladder_cases.py through the three
roles of the Conv2d entry points with the profiling hook armed and writes {case: {role: kernel name}} -- the table
tests/golden/dispatch_ladder.json pins (tests/test_gpu_kernels.py::test_dispatch_ladder_is_pinned)."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from behavenet_amd import _hip  # noqa: E402
from tests.ladder_cases import ladder_table  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, 'gpurun_out', 'dispatch_ladder.json')
    table = ladder_table()
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, 'w') as f:
        json.dump(table, f, indent=0, sort_keys=True)
    n_detour = sum(1 for roles in table.values() for name in roles.values()
                   if any(t in name for t in ('im2col', 'col2im', 'generic')))
    print('%d cases, %d roles, %d on im2col / col2im / generic rungs -> %s' % (
        len(table), sum(len(r) for r in table.values()), n_detour, out))


if __name__ == '__main__':
    main()
