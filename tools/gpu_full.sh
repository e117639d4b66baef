#!/bin/bash
# the driver's round-end GPU tier, rehearsed: the whole -m gpu suite + smoke
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/${1:-full}; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python -c "import __graft_entry__ as e; e.smoke()"
