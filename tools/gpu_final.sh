#!/bin/bash
# the round's last GPU call: the whole -m gpu suite, smoke(), then tools/profile_round.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gputest_all.log 2>&1; echo "pytest rc $?" >> $O/gputest_all.log
tail -4 $O/gputest_all.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round.sh ${1:-r04} > $O/profile_round.log 2>&1
tail -3 $O/profile_round.log
cat $O/bench.json
